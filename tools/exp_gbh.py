"""Kernel study: the generic group-by probe (gbh_insert) on the Q3 aggregation's key shape, clustered vs shuffled rows."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("presto-1_amd")
dev = torch.device("cuda", 0)
B, DT, I, D = pkg.BIGINT, pkg.DATE, pkg.INTEGER, pkg.DOUBLE
g = 1_131_128
torch.manual_seed(1)
cnt = torch.randint(1, 5, (g,), device=dev)
oi = torch.repeat_interleave(torch.arange(g, device=dev), cnt)
n = oi.numel()
for label in ("clustered", "shuffled", "distinct"):
    idx = oi if label != "shuffled" else oi[torch.randperm(n, device=dev)]
    if label == "distinct":
        idx = torch.arange(n, device=dev)
    key = (idx * 4 + 1).to(torch.int64)
    date = (8000 + idx % 2000).to(torch.int32)
    prio = torch.zeros(n, dtype=torch.int32, device=dev)
    val = torch.rand(n, dtype=torch.float64, device=dev)
    page = pkg.Page(pkg.DeviceBlock(B, n, key), pkg.DeviceBlock(DT, n, date), pkg.DeviceBlock(I, n, prio), pkg.DeviceBlock(D, n, val))
    ctx = pkg.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.profile_enable(True)
    fac = pkg.HashAggregationOperatorFactory(ctx, 14, [B, DT, I], [0, 1, 2], [(pkg.SUM_DOUBLE, 3)], expected_groups=1 << 20)
    for it in range(3):
        op = fac.createOperator()
        op.addInput(page)
        op.finish()
        o = op.getOutput()
        groups = o.position_count
        o.release()
        op.close()
        if it == 0:
            ctx.profile_reset()
    prof = ctx.profile()
    print(label, n, groups, {k: round(x["total_ms"] / x["count"], 3) for k, x in prof.items()}, flush=True)
    ctx.close()
