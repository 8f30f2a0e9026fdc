"""Kernel study: SerializedPage decode / encode rates (host bytes <-> HBM columns; PCIe-inclusive by construction)."""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("presto-1_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
rng = np.random.default_rng(0)
ctx = pkg.Context(0)
ctx.profile_enable(True)
for null_frac in (0.0, 0.1):
    nulls = lambda: (rng.random(n) < null_frac).astype(np.uint8) if null_frac else None
    blocks = [pkg.Block(pkg.BIGINT, rng.integers(0, 2**40, n), nulls()), pkg.Block(pkg.DOUBLE, rng.random(n), nulls()), pkg.Block(pkg.DOUBLE, rng.random(n), nulls()),
              pkg.Block(pkg.DATE, rng.integers(8000, 10500, n).astype(np.int32), nulls()),
              pkg.Block(pkg.VARCHAR, rng.integers(65, 91, n).astype(np.uint8), None, np.arange(n + 1, dtype=np.int32))]
    types = [pkg.BIGINT, pkg.DOUBLE, pkg.DOUBLE, pkg.DATE, pkg.VARCHAR]
    page = pkg.Page(*blocks, position_count=n)
    data = ctx.serialize_page(page)
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = ctx.deserialize_page(data, types)
        ctx.sync() if hasattr(ctx, "sync") else torch.cuda.synchronize()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dev = out.as_device_page()
        t2 = time.perf_counter()
        if it == 0:
            buf = np.zeros(len(data) + 64, dtype=np.uint8)   # a reused (already touched) exchange buffer
        t2 = time.perf_counter()
        got = ctx.serialize_page(dev, into=buf)
        t3 = time.perf_counter()
        again = buf[:got].tobytes()
        out.release()
    assert again == data
    print(f"null_frac {null_frac}: {len(data) / 1e6:.1f} MB, {n} rows; decode {(t1 - t0) * 1e3:.2f} ms = {len(data) / (t1 - t0) / 1e9:.2f} GB/s; "
          f"encode {(t3 - t2) * 1e3:.2f} ms = {len(data) / (t3 - t2) / 1e9:.2f} GB/s", flush=True)
print({k: (round(v["total_ms"] / v["count"], 3), v["count"]) for k, v in ctx.profile().items() if v["count"] and not k.startswith("__")})
