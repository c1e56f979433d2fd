"""dev aid: achievable HBM read / copy bandwidth on this box for buffers the size of the sample store (torch ops)."""
import torch, time
dev = torch.device("cuda:0")
for mb in (64, 520, 1040, 4096):
    x = torch.empty(mb * 1024 * 1024 // 4, dtype=torch.float32, device=dev).normal_()
    y = torch.empty_like(x)
    for name, fn, nbytes in (("sum (read)", lambda: x.sum(), x.numel() * 4), ("copy (read+write)", lambda: y.copy_(x), 2 * x.numel() * 4),
                             ("fill (write)", lambda: y.fill_(1.0), x.numel() * 4)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%5d MB  %-18s %7.3f ms  %6.2f TB/s" % (mb, name, ms, nbytes / ms / 1e9))
