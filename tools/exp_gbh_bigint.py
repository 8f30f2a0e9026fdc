"""Kernel study (round 3): the BenchmarkGroupByHash shapes (T/operator/BenchmarkGroupByHash.java:65-74) through GroupByHash.addPage +
appendValuesTo, per-kernel times; TGPU_GBH_INTEGER_TABLE=0 selects the generic (tag word + key store) table for comparison."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("presto-1_amd")
dev = torch.device("cuda", 0)
B = pkg.BIGINT
shapes = [(10_000_000, 3_000_000), (100_000_000, 40_000_000), (100_000_000, 1000), (50_000_000, 50_000_000)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("/")) for a in sys.argv[1:]]
for rows, groups in shapes:
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    keys = torch.randint(0, groups, (rows,), device=dev, dtype=torch.int64, generator=g)
    if groups == rows:
        keys = torch.randperm(rows, device=dev, dtype=torch.int64)
    page = pkg.Page(pkg.DeviceBlock(B, rows, keys))
    ctx = pkg.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.profile_enable(True)
    want = int(torch.unique(keys).numel())
    times = []
    for it in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gb = pkg.GroupByHash(ctx, [B], [0], expected_size=10_000)
        gb.addPage(page)
        out = gb.appendValuesDevice()
        n_groups = out.position_count
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        out.release()
        gb.close()
        if it == 0:
            ctx.profile_reset()
    prof = ctx.profile()
    steps = 5
    print(f"{rows}/{groups}: groups {n_groups} (want {want}) step {min(times[1:]) * 1e3:.3f} ms min, {sum(times[1:]) / steps * 1e3:.3f} ms avg",
          {k: (round(x["total_ms"] / steps, 3), x["count"] // steps) for k, x in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}, flush=True)
    ctx.close()
    del keys, page
    torch.cuda.empty_cache()
