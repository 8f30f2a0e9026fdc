"""Kernel study: the orders-shaped fused probe against a customer table of N ranks' worth of keys (what every rank of the replicated
plan probes at N GPUs): 150 M probe rows, custkey domain 15 M x N, 20 % of the domain in the build side."""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("presto-1_amd")
entry = importlib.import_module("__graft_entry__")
dev = torch.device("cuda", 0)
B, DT, I = pkg.BIGINT, pkg.DATE, pkg.INTEGER
n = 150_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
orderkey = torch.arange(n, dtype=torch.int64, device=dev)
orderdate = torch.randint(8000, 10500, (n,), dtype=torch.int32, device=dev, generator=g)
ship = torch.zeros(n, dtype=torch.int32, device=dev)
for N in (1, 2, 4, 8):
    dom = 15_000_000 * N
    custkey = torch.randint(1, dom + 1, (n,), dtype=torch.int64, device=dev, generator=g)
    build = torch.nonzero(torch.rand(dom, device=dev, generator=g) < 0.2).flatten().to(torch.int64) + 1
    ctx = pkg.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.profile_enable(True)
    bf = pkg.HashBuilderOperatorFactory(ctx, 1, [B], [], [0])
    b = bf.createOperator()
    b.addInput(pkg.Page(pkg.DeviceBlock(B, build.numel(), build)))
    b.finish()
    pp = entry.bench_page_processors(pkg)
    jf = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 2, bf.lookup_source_factory, *pp["q3_orders"], [1], probe_output_channels=[0, 2, 3])
    page = pkg.Page(pkg.DeviceBlock(B, n, orderkey), pkg.DeviceBlock(B, n, custkey), pkg.DeviceBlock(DT, n, orderdate), pkg.DeviceBlock(I, n, ship))
    for it in range(4):
        op = jf.createOperator()
        op.addInput(page)
        o = op.getOutput()
        rows = o.position_count if o is not None else 0
        if o is not None:
            o.release()
        op.close()
        if it == 0:
            ctx.profile_reset()
    prof = ctx.profile()
    print(f"N={N} build {build.numel()} out {rows}", {k: round(x["total_ms"] / max(x["count"], 1), 3) for k, x in prof.items() if k.startswith(("fused", "join_build"))}, flush=True)
    b.close(); ctx.close()
    del custkey, build
