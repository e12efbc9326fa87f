"""What does a plain device copy reach on this box?  (Reference point for K1's 3.1 TB/s.)"""
import torch
for mb in (64, 256, 1024, 4096):
    n = mb << 20
    a = torch.empty(n, dtype=torch.uint8, device='cuda'); b = torch.empty_like(a)
    a.fill_(1)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%5d MB copy: %.3f ms  -> %.2f TB/s (read + write)" % (mb, ms, 2 * n / ms / 1e9))
    r = torch.empty(1, device='cuda')
    e0.record()
    for _ in range(10): s = a.sum(dtype=torch.int64)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%5d MB read (sum): %.3f ms  -> %.2f TB/s" % (mb, ms, n / ms / 1e9))
