"""Calibration: what plain streaming gets out of this GPU's HBM (context for the roofline fractions in DESIGN.md) -- a device-to-device
copy (read + write) and torch reductions (read only) over multi-GB buffers."""
import time
import torch
dev = torch.device("cuda", 0)
n = 2 << 30   # 2 Gi elements
for dtype, name in ((torch.int64, "int64 sum (read 16 GiB)"), (torch.float64, "float64 sum (read 16 GiB)")):
    x = torch.ones(n, dtype=dtype, device=dev)
    for _ in range(2):
        x.sum()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        x.sum()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {x.numel() * x.element_size() / dt / 1e12:.2f} TB/s")
    del x
a = torch.empty(1 << 33, dtype=torch.uint8, device=dev)
b = torch.empty_like(a)
for _ in range(2):
    b.copy_(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    b.copy_(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"device-to-device copy of 8 GiB: {2 * a.numel() / dt / 1e12:.2f} TB/s (read + write)")
