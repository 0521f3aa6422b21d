import torch, sys
def timeit(fn, name, nbytes, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f'{name:24s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s')
for mb in (33.5, 134, 536):
    n = int(mb * 1e6 / 2)
    x = torch.randn(n, device='cuda').bfloat16(); y = torch.empty_like(x)
    timeit(lambda: y.copy_(x), f'copy {mb} MB', 2 * n * 2)
    timeit(lambda: torch.relu(x, out=y) if False else torch.clamp_min(x, 0, out=y), f'relu {mb} MB', 2 * n * 2)
    timeit(lambda: y.zero_(), f'fill {mb} MB', n * 2)
    timeit(lambda: x.sum(), f'sum(read) {mb} MB', n * 2)
