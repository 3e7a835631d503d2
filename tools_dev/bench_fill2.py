import torch
dev = torch.device('cuda:0')
for n in (1 << 27, 1 << 29):
    x = torch.empty(n, device=dev)
    for _ in range(3): x.fill_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): x.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print('fill %d MB: %.3f ms %.2f TB/s' % (n * 4 >> 20, ms, n * 4 / 1e9 / ms))
