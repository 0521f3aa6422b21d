import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aod_meh_hua_amd._C import call, ptr, stream
def timeit(fn, name, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:28s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us')
for B in (16, 1):
    for A in (36864, 9216, 2304):
        for dist in ('rand', 'const'):
            x = torch.rand(B, A, device='cuda') if dist == 'rand' else torch.full((B, A), 0.05, device='cuda')
            idx = torch.empty(B, 1000, dtype=torch.int32, device='cuda')
            timeit(lambda: call('aod_topk_stable', ptr(x), B, A, 1000, ptr(idx), 1000, stream()), f'topk B={B} A={A} {dist}')
