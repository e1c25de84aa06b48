"""HBM streaming rates of this part as seen by simple kernels (torch fill / copy / sum on 4 GiB): what can a write-heavy kernel expect?"""
import torch

dev = torch.device("cuda:0")
n = 1 << 30  # floats = 4 GiB
x = torch.empty(n, device=dev)
y = torch.empty(n, device=dev)


def timed(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


gb = n * 4 / 1e9
t = timed(lambda: x.zero_()); print(f"fill  (write only)      {t:7.3f} ms  {gb / t:7.2f} TB/s")
t = timed(lambda: x.sum()); print(f"sum   (read only)       {t:7.3f} ms  {gb / t:7.2f} TB/s")
t = timed(lambda: y.copy_(x)); print(f"copy  (read + write)    {t:7.3f} ms  {2 * gb / t:7.2f} TB/s")
t = timed(lambda: x.mul_(1.0001)); print(f"scale (read + write, in place) {t:7.3f} ms  {2 * gb / t:7.2f} TB/s")
