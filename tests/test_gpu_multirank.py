"""GPU: the multi-rank bench path (per-segment HIP graphs, asynchronous GradSync between them, sharded scoring + all-gather) runs end to
end.  A 1-GPU box cannot host two RCCL ranks, so both ranks share device 0 over gloo (AOD_BENCH_ONE_GPU=1): same code path, slower
collectives.  The collective semantics themselves are covered by the world-size-2 CPU tests (tests/test_distributed_cpu.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_on_one_gpu():
    env = dict(os.environ, AOD_BENCH_ONE_GPU='1', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29541', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '2',
           '--size', '128', '--no-cpu-baseline']
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 4 and out['value'] > 0
    assert out['config']['launch'] == 'hip-graph replay' and out['roofline'] is not None
    assert out['config']['collective_ranks'] == 2            # counted by an all-reduce over the group, not taken from the launcher
    assert out['config']['scores_read'].startswith('after the loop')


def test_two_rank_pool_mode_on_one_gpu_equals_one_rank_in_both_partitions():
    """`bench.py --mode pool` (BASELINE configs[3]: the product's pool loop, sharded, scores all-gathered) end to end at
    world size 2 through the code path the RCCL run takes (only the backend string differs), with the contiguous AND the interleaved
    partition (the indexed all-gather): the gathered score vector must be the one a single rank computes, bit for bit."""
    sha = {}
    for tag, world, shard, port in (('one', 1, 'contiguous', '29561'), ('contig', 2, 'contiguous', '29563'), ('inter', 2, 'interleaved', '29565')):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', AOD_POOL_SHARD=shard)
        if world > 1:
            env['AOD_BENCH_ONE_GPU'] = '1'
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
               '--master-port', port, os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--mode', 'pool', '--pool', '44', '--batch', '4',
               '--size', '128', '--warmup', '1', '--no-cpu-baseline']
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        out = json.loads([l for l in p.stdout.splitlines() if l.startswith('{"metric"')][-1])
        assert out['n_gpus'] == world and out['config']['collective_ranks'] == world and out['value'] > 0
        assert out['config']['pool_partition'] == shard and out['pool']['nonzero_scores'] > 0, out
        sha[tag] = out['config']['scores_sha16']
    assert sha['one'] == sha['contig'] == sha['inter'], sha


def test_two_rank_replicas_stay_equal_and_match_a_mean_gradient_run():
    """after 3 data-parallel iterations rank 0 and rank 1 hold bit-equal parameters (graph mode and eager mode), .grad
    is the slice of GradSync's flat buffer (all-reduce in place), and the result equals a 1-process run fed the mean gradient.  (The fast mode,
    whatever the suite's default: its 8-bit operands round the column sums' arrival-order noise away; the reference-precision mode has the
    test below.)"""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', AOD_CONV_PREC='bf16')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29547', os.path.join(ROOT, 'tests', 'multirank_worker.py')]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith('MULTIRANK ')][-1]
    out = json.loads(line[len('MULTIRANK '):])
    assert out['graph_replicas_equal'] and out['eager_replicas_equal'], out
    assert out['graph_grad_is_flat_slice'] and out['eager_grad_is_flat_slice'], out
    assert out['moved'] > 0
    # weight gradients are slab-deterministic; only the bias / BN column sums still go through fp32 atomics (arrival order), so the
    # data-parallel result equals the one-process mean-gradient run to the last few bits of three small SGD steps
    assert out['graph_vs_mean_gradient_run'] < 1e-5 and out['eager_vs_mean_gradient_run'] < 1e-5, out
    # per-rank shape sequences differ: the ranks mixed eager / capture / replay differently and still agree (graphs.py)
    assert out['mixed_replicas_equal'], out
    assert out['mixed_modes'][0] != out['mixed_modes'][1] and any(m == 'graph' for ms in out['mixed_modes'] for m in ms), out


def test_two_rank_replicas_in_the_reference_precision_mode():
    """The same worker in the bf16x3 mode.  Replicas stay bit-equal over three iterations in every launch mode.  Against the one-process
    mean-gradient run only the FIRST iteration is compared tightly: in this mode a 1e-7 difference in a bias (the arrival order of the fp32
    column-sum atomics) reaches the 16-bit head + tail images of activations and filters, and on the worker's 128 x 128 images -- a 4 x 4
    map in layer 4 -- a single flipped ReLU bit moves a filter's gradient by percents (tools/dbg/drift_where.py: two IDENTICAL one-process
    runs drift apart the same way from iteration 3 on; the bf16 mode rounds the difference away, which is why its runs repeat to 1e-7)."""
    # (third / fourth case: the deterministic mode -- ordered column sums AND ungrouped weight gradients, functional.set_deterministic -- removes
    # both noise sources: the data-parallel run and its one-process emulation then issue the same launches with the same summation orders, and
    # three iterations at the worker's full step size agree to the rounding of the gradient mean)
    for steps, det, bound, lr in ((1, '0', 1e-5, '2e-4'), (3, '0', 5e-2, '2e-4'), (1, '1', 1e-6, '2e-4'), (3, '1', 1e-6, '2e-4')):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', AOD_CONV_PREC='bf16x3', MULTIRANK_STEPS=str(steps), MULTIRANK_DETERMINISTIC=det,
                   MULTIRANK_LR=lr)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', '29557', os.path.join(ROOT, 'tests', 'multirank_worker.py')]
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        line = [l for l in p.stdout.splitlines() if l.startswith('MULTIRANK ')][-1]
        out = json.loads(line[len('MULTIRANK '):])
        assert out['graph_replicas_equal'] and out['eager_replicas_equal'] and out['mixed_replicas_equal'], out
        assert out['graph_grad_is_flat_slice'] and out['eager_grad_is_flat_slice'], out
        assert out['moved'] > 0
        assert out['graph_vs_mean_gradient_run'] < bound and out['eager_vs_mean_gradient_run'] < bound, (steps, det, out)


def test_two_rank_bench_over_rccl_when_two_devices_are_visible():
    """The first multi-GPU lease must exercise RCCL before the driver's scaling bench does: with >= 2 visible devices, two ranks on two
    GPUs over backend 'nccl' (= RCCL over xGMI) run the data-parallel bench step -- bucketed gradient all-reduces between the per-segment
    graphs, score all-gather -- and the pool mode (sharded scoring + one all-gather).  Skipped on a 1-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:            # (counting devices does not initialise the GPU)
        pytest.skip('needs two visible devices')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('AOD_BENCH_ONE_GPU', None)
    for extra, port in ((['--steps', '3', '--warmup', '1'], '29551'), (['--mode', 'pool', '--pool', '64'], '29553')):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', port, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--batch', '2', '--size', '128', '--no-cpu-baseline', *extra]
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        out = json.loads([l for l in p.stdout.splitlines() if l.startswith('{"metric"')][-1])
        assert out['n_gpus'] == 2 and out['value'] > 0 and out['config']['collective_backend'] == 'nccl (RCCL)', out['config']
