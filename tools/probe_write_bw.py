"""What a pure streaming write reaches on this GPU (the matrix build's roof in practice): torch fill_ of 1 GiB, 2 GiB."""
import time
import torch
for gib in (1, 2):
    x = torch.empty(gib * 2 ** 27, dtype=torch.float64, device='cuda')
    for _ in range(3):
        x.fill_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        x.fill_(2.0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"fill {gib} GiB: {ms:.3f} ms  {x.numel() * 8 / ms / 1e9:.2f} TB/s")
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"copy {gib} GiB: {ms:.3f} ms  {2 * x.numel() * 8 / ms / 1e9:.2f} TB/s (read+write)")
    del x, y
