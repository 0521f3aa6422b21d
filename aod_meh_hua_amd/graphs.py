"""HIP-graph replay of the two fixed-shape inner loops of the active-learning cycle.

A training iteration (`MyEpochBasedRunnerLambda.run_iter`, Epoch_Based_Runner_Lambda.py:20-38) enqueues ~700 kernels; on MI355X they
take ~20 ms while the Python that launches them takes about as long, so the GPU work is captured once per input shape
(`torch.cuda.CUDAGraph` == hipGraph on ROCm) and replayed:

  GraphedTrainStep   main forward / backward, MEH forward / backward, the two SGD steps.  One graph on a single GPU; with data
                     parallelism four graphs with the eager gradient all-reduces (parallel.GradSync) between them, the main
                     network's all-reduce overlapping the whole MEH segment.
  GraphedScore       one batch of the HUA scoring pass (`calculate_uncertainty`, apis/test.py:65-135).

Both keep STATIC input buffers (image batch, packed ground truth, image ids); a call copies the new batch in and replays.  Rules the
captured code obeys: no host<->device copies, no `.item()`, every accumulator is zeroed inside the graph, learning rates live in device
memory (FusedSGD.device_lr), and after a replay the versions of the updated parameters are bumped so that eager code (evaluation,
checkpoints) rebuilds its packed weights."""
import torch

from . import functional as AF
from . import hipops as ho

torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)      # warm-up runs on a side stream by design
from .models.dense_heads.L_anchor_head import PackedGT


def _pin_caches():
    """References to every cached device buffer a captured kernel may point at (row tables, the wgrad scratch, parameter-preparation
    buffers and table, scoring meta tensors): the host-side caches evict / regrow, a graph must keep what it captured alive."""
    from . import scoring
    return (dict(ho._ROW_TABLES), dict(ho._DW), dict(ho._SLABS), list(AF.PREP.items.values()), AF.PREP.table, dict(scoring._META))


def _low_memory(dev, frac=0.25):
    """less than `frac` of the device memory is free (captured graphs keep their activations in private pools)"""
    try:
        free, total = torch.cuda.mem_get_info(dev)
        return free < frac * total
    except Exception:      # noqa: BLE001
        return False


def _unwrap(model):
    return model.module if hasattr(model, 'module') else model


def _plain(data, dev):
    """DataContainer batch (runner) or plain dict (bench / tests) -> plain dict; tensors stay where they are (the image is copied into the
    static device buffer, ground truth is packed on the host)."""
    from .mmcv_lite import DataContainer

    def un(x):
        if isinstance(x, DataContainer):
            d = x.data
            return d[0] if isinstance(d, list) and len(d) == 1 else d
        return x
    return {k: un(v) for k, v in data.items()}


class GraphedTrainStep:
    """Capture-once / replay of run_iter.  Captured graphs are cached per input signature (a few shapes alternate with aspect-ratio
    grouped VOC batches).  Building a graph needs eager warm-up iterations (allocator pools, row tables, parameter-preparation
    registration): parameters, momentum buffers and BN buffers are snapshotted before and restored after, so the batch that triggers a
    capture is applied exactly ONCE (by the first replay), like every other batch.  Under data parallelism gradients live in the static
    slices of GradSync's flat buffer and the all-reduces run eagerly between the captured segments; building a graph issues NO collective
    (warm-up runs without communication), so capture / replay / eager are interchangeable per rank and per iteration."""
    MAX_GRAPHS = 16

    def __init__(self, model, optimizer, optimizer_L, grad_sync=None, gmax=64, warmup=2, **step_kwargs):
        self.model, self.module = model, _unwrap(model)
        self.opt, self.opt_L, self.sync = optimizer, optimizer_L, grad_sync
        self.gmax, self.warmup, self.kw = gmax, warmup, step_kwargs
        self.cache = {}                   # signature -> dict(graphs, static, live, live_L, keep, src, src_L), insertion order = LRU
        self.cur = None
        self.last_sig = None              # signature of the previous call to maybe()
        self.dev = next(self.module.parameters()).device

    # ------------------------------------------------------------------ input staging
    def _signature(self, d):
        """(batch tensor shape, per-image pad shapes).  Heads that read their valid-anchor flags from the static buffer
        (L_AnchorHead.get_targets_batch) make the second part irrelevant: __call__ then matches on the tensor shape alone, so the
        aspect-ratio-grouped keep-ratio VOC batches -- a dozen padded shapes, thousands of per-image shape combinations -- replay."""
        return (tuple(d['img'].shape), tuple(tuple(int(v) for v in m['pad_shape'][:2]) for m in d['img_metas']), AF.mode_key())

    def _lookup(self, sig):
        ent = self.cache.pop(sig, None)
        if ent is None:
            for k in list(self.cache):
                if k[0] == sig[0] and k[2] == sig[2] and self.cache[k]['static'].get('valid_fn') is not None:
                    return self.cache.pop(k)
        return ent

    def _known(self, sig):
        return sig in self.cache or any(k[0] == sig[0] and k[2] == sig[2] and e['static'].get('valid_fn') is not None for k, e in self.cache.items())

    def _load(self, d):
        st = self.cur['static']
        if d['img'].data_ptr() != st['img'].data_ptr():        # (a loader that filled static_image() directly has nothing to copy: 50 MB at 16 x 512^2)
            st['img'].copy_(d['img'], non_blocking=True)
        B, G = st['gts'].shape[:2]
        # fresh pinned staging every call: the copy below is asynchronous, and torch's caching host allocator only recycles a pinned
        # block once the copy that read it has completed (re-using one fixed buffer races with the next call's host writes).  Boxes, labels
        # and counts travel as ONE upload: the three static tensors are views of one byte buffer (_alloc), three 5-us copy launches per step became one
        host = torch.zeros(st['gt_pack'].numel(), dtype=torch.uint8, pin_memory=True)
        hg, hl, hc = (host[o:o + n].view(dt).view(sh) for o, n, dt, sh in st['gt_views'])
        for b, (bb, ll) in enumerate(zip(d['gt_bboxes'], d['gt_labels'])):
            n = int(bb.shape[0])
            if n > G:
                raise ValueError(f'{n} ground-truth boxes in one image exceed the graph capacity gmax={G}')
            hc[b] = n
            if n:
                hg[b, :n] = bb.detach().float().cpu() if bb.device.type != 'cpu' else bb.float()
                hl[b, :n] = ll.detach().long().cpu() if ll.device.type != 'cpu' else ll.long()
        st['gt_pack'].copy_(host, non_blocking=True)
        if st.get('valid_fn') is not None:            # valid-anchor flags of THIS batch's per-image pad shapes (cached per shape on the device)
            flags, key = st['valid_fn'](d['img_metas'])
            if st.get('valid_key') != key:            # (fixed-size data: every batch has the same pad shapes, the buffer already holds them)
                st['valid'].copy_(flags, non_blocking=True)
                st['valid_key'] = key

    def _alloc(self, d):
        B, dev, G = d['img'].shape[0], self.dev, self.gmax
        # boxes [B, G, 4] fp32 | labels [B, G] int64 | counts [B] int32 in one byte buffer (16-B aligned pieces), see _load
        nb, nl, nc = B * G * 16, B * G * 8, (B * 4 + 15) // 16 * 16
        pack = torch.zeros(nb + nl + nc, dtype=torch.uint8, device=dev)
        views = [(0, nb, torch.float32, (B, G, 4)), (nb, nl, torch.int64, (B, G)), (nb + nl, B * 4, torch.int32, (B,))]
        gts, labs, counts = (pack[o:o + n].view(dt).view(sh) for o, n, dt, sh in views)
        return dict(img=torch.empty(tuple(d['img'].shape), dtype=torch.float32, device=dev), gt_pack=pack, gt_views=views,
                    gts=gts, counts=counts, labs=labs, metas=[dict(m) for m in d['img_metas']])

    # ------------------------------------------------------------------ the segments of run_iter
    def _seg_a(self, cuts=False):
        st = self.cur['static']
        gt = PackedGT((st['gts'], st['counts'], st['labs']))
        gt.static = st                                  # (the head parks its valid-anchor flags here: a static input like the boxes)
        data = dict(img=st['img'], img_metas=st['metas'], gt_bboxes=gt, gt_labels=None)
        if cuts:            # data parallelism: the backward pass falls into segments at the backbone's stage outputs (functional.grad_cuts)
            with AF.grad_cuts() as cl:
                out, head_out, feat_out, prev = self.module.train_step(data, **self.kw)
            self.cur['cuts'] = list(reversed(cl))
            # the cuts the forward actually produced decide: when they do not match the bucket segments (a precision mode or model without
            # cut points) the pass runs as whatever segments exist and ONE unsegmented all-reduce follows the last of them -- what the eager
            # path (parallel.backward_and_sync) does in the same situation
            self.cur['segmented'] = len(cl) == self.nseg - 1
        else:
            out, head_out, feat_out, prev = self.module.train_step(data, **self.kw)
        self.opt.zero_grad()
        self.cur['live'] = (out, head_out, feat_out, prev)
        out['loss'].backward()          # (with cuts: segment 0 -- heads and neck; it ends at the cut copies)
        # every 0-dim value the caller gets back, packed into ONE static vector inside the graph: a step then hands out copies with one
        # device copy per segment instead of one per log entry (~60 launches of 4 us each, serialised behind the replay)
        self.cur['pack'] = torch.stack([out['loss'].detach().float().reshape(())] + [v.detach().float().reshape(()) for v in out['log_vars'].values()])

    def _seg_ak(self, k):
        """backward segment k >= 1: the backbone stage behind the k-th deepest cut"""
        if k - 1 >= len(self.cur['cuts']):
            return
        x, xc = self.cur['cuts'][k - 1]
        g, xc.grad = xc.grad, None
        if g is not None:
            x.backward(g)

    def _seg_b(self):
        # the MEH step only reads detached features / losses and its own parameters, so the main update (segment C) may follow it:
        # with data parallelism the main gradients' last all-reduce buckets then run under this whole segment
        out, head_out, feat_out, prev = self.cur['live']
        loss_L = self.module.train_step_L(prev, head_out, feat_out, **self.kw)
        self.opt_L.zero_grad()
        loss_L['loss'].backward()
        self.cur['live_L'] = loss_L
        self.cur['pack_L'] = torch.stack([v.detach().float().reshape(()) for v in loss_L['log_vars'].values()])

    def _seg_c(self):
        self.opt.step()

    def _seg_d(self):
        self.opt_L.step()

    def _params(self, opt):
        return [p for g in opt.param_groups for p in g['params']]

    def _segments(self, dist_mode):
        """(tag, callable) in execution order.  One process: everything is ONE graph.  Data parallelism: a0 (forward + heads / neck
        backward), a1 .. aK (backbone stages, deepest first), b (MEH forward / backward), c, d (the two SGD steps) are graphs of their
        own with the eager bucket all-reduces between them."""
        if not dist_mode:
            return [('a', self._seg_a), ('b', self._seg_b), ('c', self._seg_c), ('d', self._seg_d)]
        segs = [('a0', lambda: self._seg_a(cuts=True))]
        segs += [(f'a{k}', (lambda k=k: self._seg_ak(k))) for k in range(1, self.nseg)]
        return segs + [('b', self._seg_b), ('c', self._seg_c), ('d', self._seg_d)]

    def _between(self, tag, capturing=False):
        """Communication behind segment `tag` (data parallelism only).  While CAPTURING nothing is sent: only the hand-over the later
        segments' captured pointers depend on happens (.grad -> the slices of the flat buffer the reduced values arrive in)."""
        if self.sync is None:
            return
        cur = self.cur
        main, meh = self._params(self.opt), self._params(self.opt_L)
        if tag.startswith('a'):
            k = int(tag[1:] or 0)
            ent = self.sync.attach(main)
            whole = ent['seg_of'] is None or not cur.get('segmented', True)     # no per-segment buckets to send: one all-reduce at the end
            if whole and ent['seg_of'] is not None and k < self.nseg - 1:
                return
            idx = [i for i in range(len(main)) if whole or ent['seg_of'][i] == k]
            if capturing:
                src = cur.setdefault('src', [None] * len(main))
                for i in idx:
                    src[i] = main[i].grad                                     # what this segment's captured kernels write
            else:
                self.inflight.append(self.sync.start(main, sources=cur.get('src'), segment=None if whole else k))
        elif tag == 'b':
            if capturing:
                cur['src_L'] = [p.grad for p in meh]
                for opt in (self.opt, self.opt_L):                              # segments C / D read the reduced slices
                    ent = self.sync.attach(self._params(opt))
                    for p, v in zip(ent['params'], ent['views']):
                        if p.grad is not None:
                            p.grad = v
            else:
                for h in self.inflight:
                    h.wait()
                self.inflight = []
        elif tag == 'c' and not capturing:
            self.sync.start(meh, sources=cur.get('src_L')).wait()

    def _run_eager(self, comm=True):
        """comm=False (warm-up before a capture): NO collective is issued -- the iteration is undone afterwards anyway (_restore), and a
        rank that captures must issue exactly the collectives of a rank that replays or runs eagerly (ranks see different batch shapes
        with keep-ratio VOC data and keep their own graph caches: one rank may capture while its peers replay)."""
        self.inflight = []
        for tag, f in self._segments(getattr(self, 'dist_mode', False)):
            f()
            if comm:
                self._between(tag)
            elif self.sync is not None and tag in ('b', 'c'):       # hand the flat-buffer slices back to the next backward (what start() does)
                self.sync.release(self._params(self.opt if tag == 'b' else self.opt_L))

    # ------------------------------------------------------------------ state snapshot around warm-up
    def _snapshot(self):
        ps = self._params(self.opt) + self._params(self.opt_L)
        snap = dict(p=[p.detach().clone() for p in ps], m=[], b=[b.detach().clone() for b in self.module.buffers()])
        for opt in (self.opt, self.opt_L):
            for p in self._params(opt):
                mb = opt.state.get(p, {}).get('momentum_buffer')
                snap['m'].append(None if mb is None else mb.detach().clone())
        return snap

    def _restore(self, snap):
        ps = self._params(self.opt) + self._params(self.opt_L)
        with torch.no_grad():
            torch._foreach_copy_(ps, snap['p'])
            for b, v in zip(self.module.buffers(), snap['b']):
                b.copy_(v)
            i = 0
            for opt in (self.opt, self.opt_L):
                for p in self._params(opt):
                    mb = opt.state.get(p, {}).get('momentum_buffer')
                    if mb is not None:
                        # a buffer that did not exist before warm-up restarts from zero: buf = m * 0 + d == the first-step rule buf = d
                        mb.zero_() if snap['m'][i] is None else mb.copy_(snap['m'][i])
                    i += 1
        torch._C._autograd._unsafe_set_version_counter(ps, [p._version + 1 for p in ps])

    def _build(self, d):
        from .parallel import is_dist
        dist_mode = self.dist_mode = self.sync is not None and is_dist()
        self.nseg = 1
        if dist_mode:                      # gradients must land in the flat buffer's slices from the first captured backward on
            main = self._params(self.opt)
            self.nseg = self.sync.attach(main, segments=self.module.grad_segments(main) if hasattr(self.module, 'grad_segments') else None)['nseg']
            self.sync.attach(self._params(self.opt_L))
        self.cur = dict(static=self._alloc(d), live=None, live_L=None)
        self._load(d)
        self.opt.device_lr(), self.opt_L.device_lr()
        self.module.train()
        snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self._run_eager(comm=False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ho.reset_zero_arena()                     # accumulators handed out during capture must be zeroed INSIDE the graph
        # packed weights / folded BN are re-derived EAGERLY before every replay (one launch, only when a parameter changed): the graph
        # itself must not contain the refresh, or its position would depend on which layers happened to be stale at capture time
        AF.PREP.refresh_if_stale()
        segs = self._segments(dist_mode)
        graphs = []
        if not dist_mode:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):      # other threads (RCCL watchdog) may touch the runtime
                for _, f in segs:
                    f()
            graphs.append(g)
        else:
            pool = torch.cuda.graph_pool_handle()
            for tag, f in segs:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode='thread_local'):
                    f()
                graphs.append(g)
                self._between(tag, capturing=True)
            self.cur['tags'] = [tag for tag, _ in segs]
        ho.reset_zero_arena()
        self._restore(snap)                       # warm-up iterations were real updates: undo them, the first replay applies this batch
        AF.PREP.refresh_if_stale()
        self.cur.update(graphs=graphs, keep=_pin_caches(), touched=self._params(self.opt) + self._params(self.opt_L))
        return self.cur

    # ------------------------------------------------------------------ call
    def maybe(self, data_batch):
        """Replay when this input shape is cached, capture when it was also the previous call's shape (second consecutive appearance);
        otherwise return None and let the caller run the iteration eagerly (multi-scale data would re-capture on every batch).
        The decision is RANK-LOCAL under data parallelism: an eager iteration, a replay and a capture + first replay all issue the same
        sequence of bucket all-reduces (main buckets after the main backward, MEH buckets after the MEH backward; building a graph sends
        nothing), so ranks need not agree and no per-iteration flag all-reduce / host sync is paid."""
        d = _plain(data_batch, self.dev)
        sig = self._signature(d)
        want = self._known(sig) or sig[0] == (self.last_sig or (None,))[0]
        self.last_sig = sig
        if not want:
            return None
        return self(data_batch)

    def __call__(self, data_batch):
        d = _plain(data_batch, self.dev)
        sig = self._signature(d)
        ent = self._lookup(sig)
        if ent is None:
            # least recently used graphs go when the cache is full -- or when their private memory pools leave too little room for one more
            while self.cache and (len(self.cache) >= self.MAX_GRAPHS or _low_memory(self.dev)):
                self.cache.pop(next(iter(self.cache)))
                import gc
                gc.collect()
                torch.cuda.empty_cache()
            ent = self._build(d)
        self.cache[sig] = self.cur = ent
        self._load(d)
        self.opt.device_lr(), self.opt_L.device_lr()
        AF.PREP.refresh_if_stale()
        self.inflight = []
        for i, g in enumerate(ent['graphs']):
            g.replay()
            if len(ent['graphs']) > 1:
                self._between(ent['tags'][i])
        tp = [p for p in ent['touched'] if p.grad is not None]
        torch._C._autograd._unsafe_set_version_counter(tp, [p._version + 1 for p in tp])
        out, loss_L = ent['live'][0], ent['live_L']
        # static tensors: hand out copies (two device copies: the packed vectors of segments A and B)
        pa, pb = self.cur['pack'].clone(), self.cur['pack_L'].clone()
        log_vars = type(out['log_vars'])(zip(out['log_vars'].keys(), pa[1:].unbind(0)))
        log_vars.update(zip(loss_L['log_vars'].keys(), pb.unbind(0)))
        return dict(loss=pa[0], log_vars=log_vars, num_samples=out['num_samples'])


class GraphedScore:
    """One HUA scoring batch: model(img=[img], img_metas=[metas], image_ids=ids, **kw) under no_grad -> unc [B] (a copy).  One captured
    graph set per batch tensor shape (LRU): the per-image sizes and scale factors the decode step needs are STATIC device buffers refreshed
    per call (scoring.static_meta), so keep-ratio pool batches of one padded shape share a graph.

    Two graphs, two streams (AOD_SCORE_PIPELINE=0: one graph).  A scoring batch is a conv half -- backbone, neck, towers, prediction convs:
    thousands of workgroups per launch -- and a selection half -- softmax / top-k / decode / NMS / pairs / Dirichlet sampler / reduce: one
    workgroup per image for most of its 0.5 ms, 240 of 256 CUs idle.  The halves are captured separately (detector.extract_feat +
    head.test_heads | head.simple_test(_preds=...)) into TWO slots that alternate, so that the selection half of batch k runs on a second
    stream beside the conv half of batch k + 1 (it reads slot k % 2's prediction tensors, batch k + 1 writes the other slot's).  The
    overlap happens between calls with `defer=True` (the pool loop of apis/test.py, which collects the scores and calls sync() once at the
    end); a plain call makes the caller's stream wait for its scores, which also orders the next call behind them."""
    MAX_GRAPHS = 16

    def __init__(self, model, warmup=2, pipeline=None, **score_kwargs):
        import os
        import weakref
        # the model is held WEAKLY: single_gpu_uncertainty caches this object on the model itself, and a strong reference back would be a
        # cycle that keeps a dead cycle's graph memory pool + static image buffers alive until the cyclic collector happens to run
        self._model = weakref.ref(_unwrap(model))
        self.kw, self.warmup = score_kwargs, warmup
        self.cache, self.pending = {}, None          # shape -> entry (insertion order = LRU); shape seen once
        self.dev = next(_unwrap(model).parameters()).device
        m = _unwrap(model)
        two_phase = (hasattr(m, 'extract_feat') and hasattr(getattr(m, 'bbox_head', None), 'test_heads') and not score_kwargs.get('isEval')
                     and getattr(m.test_cfg, 'uncertainty_pool', None) in ('Entropy_NMS', 'Entropy_ALL'))
        self.pipe = two_phase and (os.environ.get('AOD_SCORE_PIPELINE', '1') != '0' if pipeline is None else bool(pipeline))
        self.s_conv = self.s_tail = None

    @property
    def module(self):
        m = self._model()
        if m is None:
            raise RuntimeError('GraphedScore outlived its model')
        return m

    # ------------------------------------------------------------------ what the graphs contain
    def _run(self, sl):
        from . import scoring
        with torch.no_grad(), scoring.static_meta(sl['hw'], sl['sc']):
            sl['out'] = self.module(img=[sl['img']], img_metas=[sl['metas']], return_loss=False, image_ids=sl['ids'], **self.kw)

    def _run_a(self, sl):
        m = self.module
        with torch.no_grad():
            sl['preds'] = m.bbox_head.test_heads(m.extract_feat(sl['img']))

    def _run_b(self, sl):
        """the selection half as SSL_L_SingleStageDetector.simple_test / forward_test run it for isEval=False"""
        from . import scoring
        m = self.module
        kw = dict(self.kw)
        rescale = kw.pop('rescale', False)
        for meta in sl['metas']:
            meta['batch_input_shape'] = tuple(sl['img'].shape[-2:])
        with torch.no_grad(), scoring.static_meta(sl['hw'], sl['sc']):
            results_list, *unc = m.bbox_head.simple_test(None, sl['metas'], rescale=rescale, _preds=sl['preds'], image_ids=sl['ids'],
                                                         _data=sl['img'], _meta=sl['metas'], **kw)
        sl['out'] = (results_list, *unc)

    def static_image(self, shape):
        """The input buffer the NEXT call with image batches of `shape` replays on (None while no graph of that shape exists): a producer that
        writes the batch there -- the on-device pool generator, a loader's H2D copy -- saves the device-to-device copy into it.  The caller's
        stream is made to wait until the slot's previous conv half has read it."""
        ent = self.cache.get((tuple(shape), AF.mode_key()))
        if ent is None:
            return None
        sl = ent['slots'][ent['n'] % len(ent['slots'])]
        if sl.get('ev_a') is not None:
            torch.cuda.current_stream().wait_event(sl['ev_a'])
        return sl['img']

    def maybe(self, img, img_metas, image_ids, defer=False):
        """Replay if this batch shape is captured, capture if it repeats the previous batch's shape, else None (caller scores eagerly)."""
        shape = (tuple(img.shape), AF.mode_key())
        if shape not in self.cache and shape != self.pending:
            self.pending = shape
            return None
        return self(img, img_metas, image_ids, defer=defer)

    def sync(self):
        """the caller's stream waits for every deferred selection half (call before reading scores returned with defer=True)"""
        if self.s_tail is not None:
            torch.cuda.current_stream().wait_stream(self.s_tail)

    def _fill_img(self, sl, img):
        if img.data_ptr() != sl['img'].data_ptr():            # (see static_image)
            sl['img'].copy_(img, non_blocking=True)

    def _fill_meta(self, sl, img_metas, image_ids):
        from . import scoring
        # (value-keyed device copies of the sizes / scale factors: a pool re-uses a handful of them; device-to-device into the static buffers)
        hw, sc = scoring._meta_tensors([m['img_shape'] for m in img_metas], [m['scale_factor'] for m in img_metas], self.dev)
        sl['ids'].copy_(image_ids, non_blocking=True)
        # (the size / scale tensors are value-cached objects: a pool of one image size hands over the same two tensors batch after batch --
        # the slot already holds their values then, two 5-us copy launches at the head of every conv half saved)
        if sl.get('meta_src') != (id(hw), id(sc)):
            sl['hw'].copy_(hw, non_blocking=True), sl['sc'].copy_(sc, non_blocking=True)
            sl['meta_src'], sl['meta_keep'] = (id(hw), id(sc)), (hw, sc)          # (keep the objects alive: ids are only unique among live objects)

    def _capture(self, sl, fns):
        """warm-up runs on a side stream, then one graph per function of `fns` (they share nothing but the slot's tensors)"""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                for f in fns:
                    f(sl)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graphs = []
        for f in fns:
            ho.reset_zero_arena()
            AF.PREP.refresh_if_stale()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                f(sl)
            graphs.append(g)
        ho.reset_zero_arena()
        return graphs

    def __call__(self, img, img_metas, image_ids, defer=False):
        """defer=True: the scores are complete only after sync().  The caller may drop or re-allocate `img` / `image_ids` at once (both are
        recorded on the stream that copies them) and may overwrite them in place on its own stream (which is made to wait for those copies)."""
        shape = (tuple(img.shape), AF.mode_key())          # (the key: a graph serves one batch shape in one arithmetic / determinism mode)
        ent = self.cache.pop(shape, None)
        if ent is None:
            while self.cache and (len(self.cache) >= self.MAX_GRAPHS or _low_memory(self.dev)):
                self.cache.pop(next(iter(self.cache)))
                import gc
                gc.collect()
                torch.cuda.empty_cache()
            if self.s_tail is not None:                          # (a capture synchronises the device anyway)
                torch.cuda.current_stream().wait_stream(self.s_tail)
            B = img.shape[0]
            self.module.eval()
            slots = []
            for _ in range(2 if self.pipe else 1):
                sl = dict(img=torch.empty(tuple(img.shape), dtype=torch.float32, device=self.dev), ids=torch.zeros(B, dtype=torch.int64, device=self.dev),
                          hw=torch.zeros(B, 2, device=self.dev), sc=torch.ones(B, 4, device=self.dev), metas=[dict(m) for m in img_metas],
                          ev_a=None, ev_b=None)
                self._fill_img(sl, img)
                self._fill_meta(sl, img_metas, image_ids)
                if self.pipe:
                    sl['ga'], sl['gb'] = self._capture(sl, [self._run_a, self._run_b])
                else:
                    sl['g'], = self._capture(sl, [self._run])
                slots.append(sl)
            ent = dict(slots=slots, n=0, keep=_pin_caches())
        self.cache[shape] = ent
        sl = ent['slots'][ent['n'] % len(ent['slots'])]
        ent['n'] += 1
        AF.PREP.refresh_if_stale()        # e.g. a training replay or a checkpoint load since the last scoring batch
        if not self.pipe:
            self._fill_img(sl, img)
            self._fill_meta(sl, img_metas, image_ids)
            sl['g'].replay()
            unc = sl['out'][1]
            return sl['out'][0], (unc.clone() if torch.is_tensor(unc) else unc)
        if self.s_conv is None:
            self.s_conv, self.s_tail = torch.cuda.Stream(), torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        self.s_conv.wait_stream(cur)                 # the batch, its ids (and, for a non-deferred predecessor, its scores) are produced on the caller's stream
        with torch.cuda.stream(self.s_conv):
            if sl['ev_b'] is not None:
                self.s_conv.wait_event(sl['ev_b'])   # this slot's previous selection half has read the prediction tensors, the ids and the sizes
            copied = img.data_ptr() != sl['img'].data_ptr() or (torch.is_tensor(image_ids) and image_ids.is_cuda)
            self._fill_img(sl, img)
            self._fill_meta(sl, img_metas, image_ids)            # (on THIS stream: the selection half starts behind ev_a, an id copy queued there would
            ev_fill = None                                       #  read the caller's tensor a whole conv half later -- ADVICE r4)
            if copied:
                ev_fill = torch.cuda.Event()
                ev_fill.record(self.s_conv)
                if img.is_cuda:
                    img.record_stream(self.s_conv)
                if torch.is_tensor(image_ids) and image_ids.is_cuda:
                    image_ids.record_stream(self.s_conv)
            sl['ga'].replay()
            sl['ev_a'] = torch.cuda.Event()
            sl['ev_a'].record(self.s_conv)
        if ev_fill is not None:
            cur.wait_event(ev_fill)                  # the caller's inputs have been read: it may overwrite them in place from here on
        with torch.cuda.stream(self.s_tail):
            self.s_tail.wait_event(sl['ev_a'])
            sl['gb'].replay()
            unc = sl['out'][1]
            unc = unc.clone() if torch.is_tensor(unc) else unc
            sl['ev_b'] = torch.cuda.Event()
            sl['ev_b'].record(self.s_tail)
        if torch.is_tensor(unc):
            unc.record_stream(cur)
        if not defer:
            cur.wait_stream(self.s_tail)
        return sl['out'][0], unc
