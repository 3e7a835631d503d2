"""Pure-write and copy bandwidth of the device (torch fill_ / copy_) for comparison with the write-bound kernels."""
import torch
dev = torch.device('cuda:0')
x = torch.empty(1 << 29, device=dev)          # 2 GiB of floats
y = torch.empty_like(x)
for name, fn, bytes_ in (('fill', lambda: x.fill_(1.0), x.numel() * 4), ('copy', lambda: y.copy_(x), 2 * x.numel() * 4)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print('%s: %.3f ms  %.2f TB/s' % (name, ms, bytes_ / 1e9 / ms))
