"""Known-byte-count streams for calibrating FETCH_SIZE / WRITE_SIZE (run under rocprofv3 --pmc)."""
import torch
x = torch.rand(64 * 1024 * 1024, device="cuda")          # 256 MiB of fp32
for _ in range(3):
    y = x.clone()                                         # reads 256 MiB, writes 256 MiB (elementwise copy kernel)
    s = x.sum()                                           # reads 256 MiB (reduce kernel)
    z = torch.zeros_like(x)                               # writes 256 MiB (fill kernel)
torch.cuda.synchronize()
print("done", float(s))
