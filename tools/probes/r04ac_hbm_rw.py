"""What the memory system sustains for pure writes, pure reads and copies (torch kernels; sizes beyond the 256 MB Infinity Cache and a 84 MB tensor that fits it)."""
import torch
dev = torch.device("cuda:0")
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (84, 336, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device=dev)
    y = torch.empty(n, dtype=torch.bfloat16, device=dev)
    tw = timeit(lambda: x.fill_(1.0))
    tc = timeit(lambda: y.copy_(x))
    tr = timeit(lambda: x.view(torch.int32).sum())
    print(f"{mb:5d} MB: fill {mb / 1024 / tw / 1e3 * 1.0737:6.2f} TB/s   copy {2 * mb / 1024 / tc / 1e3 * 1.0737:6.2f} TB/s (read + write)   read (sum) {mb / 1024 / tr / 1e3 * 1.0737:6.2f} TB/s", flush=True)
