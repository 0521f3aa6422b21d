"""MyEpochBasedRunnerLambda (mmdet/utils/Epoch_Based_Runner_Lambda.py:18-169): per iteration the main
forward/backward/step THEN the MEH forward/backward/step (two optimizers), run_SSL workflow loop,
save_checkpoint.  Distributed (one process per MI355X) gradient averaging hooks in between backward and
step through `parallel.GradSync` (RCCL all-reduce of flat buckets) when torch.distributed is initialised."""
import os.path as osp
import platform
import shutil

import torch

from ..mmcv_lite import RUNNERS, BaseRunner, get_host_info, save_checkpoint
from ..parallel import GradSync


@RUNNERS.register_module()
class MyEpochBasedRunnerLambda(BaseRunner):
    def _module(self):
        return self.model.module if hasattr(self.model, 'module') else self.model

    def _sync_start(self, optimizer):
        if not hasattr(self, '_gsync'):
            self._gsync = GradSync()
        return self._gsync.start([p for g in optimizer.param_groups for p in g['params']])

    def _sync(self, optimizer):
        self._sync_start(optimizer).wait()

    def run_iter(self, data_batch, train_mode, **kwargs):
        """Epoch_Based_Runner_Lambda.py:20-38."""
        if self.batch_processor is not None:
            outputs = self.batch_processor(self.model, data_batch, train_mode=train_mode, **kwargs)
        elif train_mode and self._graphed_iter(data_batch, kwargs):
            outputs = self.outputs
        elif train_mode:
            from .. import functional as AF
            from ..parallel import backward_and_sync, is_dist
            if is_dist():
                # data parallelism: the backward pass runs in segments (head + neck, then the backbone stages, deepest first) and each
                # segment's gradient buckets are all-reduced while the next segment computes (parallel.GradSync)
                if not hasattr(self, '_gsync'):
                    self._gsync = GradSync()
                params = [p for g in self.optimizer.param_groups for p in g['params']]
                self._gsync.attach(params, segments=self._module().grad_segments(params))
                with AF.grad_cuts() as cuts:
                    loss, head_out, feat_out, prev_loss = self.model.train_step(data_batch, **kwargs)
                self.optimizer.zero_grad()
                pending = backward_and_sync(self._gsync, params, loss['loss'], cuts)
            else:
                loss, head_out, feat_out, prev_loss = self.model.train_step(data_batch, **kwargs)
                self.optimizer.zero_grad()
                loss['loss'].backward()
                pending = self._sync_start(self.optimizer)
            # The MEH step reads only detached features / losses and its own parameters (train_step_L), so the main update may be
            # applied after it: the main network's last all-reduce buckets then run under the whole MEH forward/backward.  Same values as
            # the reference order (optimizer.step() before train_step_L, Epoch_Based_Runner_Lambda.py:27-35).
            loss_L = self._module().train_step_L(prev_loss, head_out, feat_out, _data=data_batch, **kwargs)
            self.optimizer_L.zero_grad()
            loss_L['loss'].backward()
            pending.wait()
            self.optimizer.step()
            self._sync(self.optimizer_L)
            self.optimizer_L.step()
            loss['log_vars'].update(loss_L['log_vars'])
            # keep only the value: a loss that still owns its autograd graph keeps the parameters' AccumulateGrad nodes (bound to
            # this stream) alive, which makes a later HIP-graph capture of the iteration illegal -- and pins the activations
            loss['loss'] = loss['loss'].detach()
            del loss_L, head_out, feat_out, prev_loss
            outputs = loss
        else:
            outputs = self.model.val_step(data_batch, self.optimizer, **kwargs)
        if 'log_vars' in outputs:
            self.log_buffer.update(outputs['log_vars'], outputs['num_samples'])
        self.outputs = outputs

    def _graphed_iter(self, data_batch, kwargs):
        """HIP-graph replay of the iteration (graphs.GraphedTrainStep) once the input shape repeats; False -> run it eagerly.
        Disabled by `runner.hip_graph = False` or AOD_HIP_GRAPH=0."""
        import os
        if not getattr(self, 'hip_graph', True) or os.environ.get('AOD_HIP_GRAPH', '1') == '0' or not hasattr(self, 'optimizer_L'):
            return False
        key = tuple(sorted((k, v) for k, v in kwargs.items() if isinstance(v, (bool, int, str))))
        gs = getattr(self, '_graph_step', None)
        if gs is None or gs[0] != key:
            from ..graphs import GraphedTrainStep
            from ..parallel import is_dist
            if not hasattr(self, '_gsync'):
                self._gsync = GradSync()
            sync = self._gsync if is_dist() else None
            gs = self._graph_step = (key, GraphedTrainStep(self.model, self.optimizer, self.optimizer_L, grad_sync=sync, **dict(key)))
        out = gs[1].maybe(data_batch)
        if out is None:
            return False
        self.outputs = out
        return True

    def train(self, data_loader, **kwargs):
        """:40-75 (labeled-only branch; the unlabeled/pseudo loader list is never passed by the AL driver)."""
        assert not isinstance(data_loader, list), 'pseudo-label loaders are dead code in the reference driver'
        self.model.train()
        self.mode = 'train'
        self.data_loader = data_loader
        self._max_iters = self._max_epochs * len(data_loader)
        sampler = getattr(data_loader, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):          # mmcv DistSamplerSeedHook: a different share / order every epoch under data parallelism
            sampler.set_epoch(self.epoch)
        self.call_hook('before_train_epoch')
        for i, data_batch_L in enumerate(data_loader):
            if kwargs.get('onlyEval'):
                break
            self._inner_iter = i
            self.call_hook('before_train_iter')
            self.run_iter(data_batch_L, train_mode=True, Labeled=True, Pseudo=False, **kwargs)
            self.call_hook('after_train_iter')
            self._iter += 1
        eval_res = self.call_hook('after_train_epoch')
        self._epoch += 1
        return eval_res

    @torch.no_grad()
    def val(self, data_loader, **kwargs):
        self.model.eval()
        self.mode = 'val'
        self.data_loader = data_loader
        self.call_hook('before_val_epoch')
        for i, data_batch in enumerate(self.data_loader):
            self._inner_iter = i
            self.call_hook('before_val_iter')
            self.run_iter(data_batch, train_mode=False)
            self.call_hook('after_val_iter')
        self.call_hook('after_val_epoch')

    def run_SSL(self, data_loaders, workflow, max_epochs=None, **kwargs):
        """:115-142."""
        self._max_epochs = max_epochs
        self._max_iters = self._max_epochs * len(data_loaders[0])
        self.logger.info('Start running, host: %s, work_dir: %s', get_host_info(), self.work_dir or 'NONE')
        self.logger.info('workflow: %s, max: %d epochs', workflow, self._max_epochs)
        self.call_hook('before_run')
        eval_res = None
        while self.epoch < self._max_epochs:
            for i, (mode, epochs) in enumerate(workflow):
                epoch_runner = getattr(self, mode)
                for _ in range(epochs):
                    if mode == 'train' and self.epoch >= self._max_epochs:
                        break
                    eval_res = epoch_runner(data_loaders[i], **kwargs)
        self.call_hook('after_run')
        return eval_res

    run = run_SSL

    def save_checkpoint(self, out_dir, filename_tmpl='epoch_{}.pth', save_optimizer=True, meta=None, create_symlink=True):
        """:144-169."""
        meta = dict(meta or {}, epoch=self.epoch + 1, iter=self.iter)
        if self.meta is not None:
            meta.update(self.meta)
        filename = filename_tmpl.format(self.epoch + 1)
        filepath = osp.join(out_dir, filename)
        save_checkpoint(self.model, filepath, optimizer=self.optimizer if save_optimizer else None, meta=meta)
        if create_symlink:
            dst_file = osp.join(out_dir, 'latest.pth')
            if platform.system() != 'Windows':
                if osp.lexists(dst_file):
                    import os
                    os.remove(dst_file)
                import os
                os.symlink(filename, dst_file)
            else:
                shutil.copy(filepath, dst_file)


@RUNNERS.register_module()
class MyEpochBasedRunnerLSSD(MyEpochBasedRunnerLambda):
    """mmdet/utils/Epoch_Based_Runner_SSD_L.py (byte-identical to the Lambda runner apart from the class name)."""
