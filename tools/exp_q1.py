"""Kernel study: the fused Q1 pipeline under experiment switches of the group-probe kernel."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["BASE", "NOKEYS", "NOTABLE"]
pkg = importlib.import_module("presto-1_amd")
entry = importlib.import_module("__graft_entry__")
dev = torch.device("cuda", 0)
n = int(6_000_379.02 * sf)
t = bench.gen_q1(dev, n)
V, D, DT = pkg.VARCHAR, pkg.DOUBLE, pkg.DATE
db = lambda ty, v, off=None: pkg.DeviceBlock(ty, v.numel() if off is None else off.numel() - 1, v, None, off)
page = pkg.Page(db(V, t["returnflag"], t["off"]), db(V, t["linestatus"], t["off"]), db(D, t["quantity"]), db(D, t["extendedprice"]), db(D, t["discount"]),
                db(D, t["tax"]), db(DT, t["shipdate"]))
pp = entry.bench_page_processors(pkg)
for v in variants:
    os.environ.pop("TGPU_FG_EXP", None)
    if "=" in v:  # NAME=VALUE: an environment switch of the library
        k, val = v.split("=", 1)
        os.environ[k] = val
    elif v != "BASE":
        os.environ["TGPU_FG_EXP"] = v
    ctx = pkg.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.profile_enable(True)
    fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 21, *pp["q1"], [V, V], [0, 1], entry.q1_aggregates(pkg), expected_groups=16)
    for it in range(3):
        op = fac.createOperator()
        op.addInput(page)
        op.finish()
        o = op.getOutput()
        rows = o.to_host().rows() if o is not None else []
        op.close()
        if it == 0:
            ctx.profile_reset()
    prof = ctx.profile()
    print(v, len(rows), {k: (round(x["total_ms"] / 2, 3), round(x["min_ms"], 3), round(x["max_ms"], 3), x["count"]) for k, x in prof.items()}, flush=True)
    ctx.close()
