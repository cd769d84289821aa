import torch, time
n = 1 << 28   # 2 GiB of f64
a = torch.empty(n, dtype=torch.float64, device="cuda"); b = torch.ones(n, dtype=torch.float64, device="cuda")
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: a.copy_(b)); print("copy  read+write %.0f GB/s" % (2 * n * 8 / ms / 1e6))
ms = t(lambda: b.sum()); print("sum   read only  %.0f GB/s" % (n * 8 / ms / 1e6))
ms = t(lambda: a.fill_(1.0)); print("fill  write only %.0f GB/s" % (n * 8 / ms / 1e6))
ms = t(lambda: torch.add(a, b, out=a)); print("a+=b  2 reads 1 write %.0f GB/s" % (3 * n * 8 / ms / 1e6))
