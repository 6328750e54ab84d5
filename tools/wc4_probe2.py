import torch
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return round(s.elapsed_time(e) * 1000 / n, 1)
for parts in (512, 256, 128):
    slab = torch.randn(parts, 14336, device='cuda')
    print(parts, 'sum(0) us', t(lambda: slab.sum(0)), ' copy+sum us', t(lambda: (slab.add_(1.0), slab.sum(0))), ' add_ alone', t(lambda: slab.add_(1.0)))
