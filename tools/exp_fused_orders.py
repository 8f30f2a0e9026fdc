"""Kernel study: the fused filter+probe kernel on the Q3 ORDERS shape (random custkey probes into the customer table) under
experiment switches."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["BASE", "NOPROBE"]
pkg = importlib.import_module("presto-1_amd")
entry = importlib.import_module("__graft_entry__")
dev = torch.device("cuda", 0)
t = bench.gen_q3(dev, sf)
B, D, DT, I = pkg.BIGINT, pkg.DOUBLE, pkg.DATE, pkg.INTEGER
seg = t["c_seg_bytes"][t["c_seg_off"][:-1].to(torch.int64)]
ckeys = t["c_custkey"][seg == ord("B")].contiguous()
n = t["o_orderkey"].numel()
print("build rows", ckeys.numel(), "orders rows", n)
for v in variants:
    os.environ.pop("TGPU_FJ_EXP", None)
    if "=" in v:
        k_, val_ = v.split("=", 1)
        os.environ[k_] = val_
    elif v != "BASE":
        os.environ["TGPU_FJ_EXP"] = v
    ctx = pkg.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.profile_enable(True)
    bf = pkg.HashBuilderOperatorFactory(ctx, 1, [B], [], [0])
    b = bf.createOperator()
    b.addInput(pkg.Page(pkg.DeviceBlock(B, ckeys.numel(), ckeys)))
    b.finish()
    pp = entry.bench_page_processors(pkg)
    jf = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 2, bf.lookup_source_factory, *pp["q3_orders"], [1], probe_output_channels=[0, 2, 3])
    page = pkg.Page(pkg.DeviceBlock(B, n, t["o_orderkey"]), pkg.DeviceBlock(B, n, t["o_custkey"]), pkg.DeviceBlock(DT, n, t["o_orderdate"]),
                    pkg.DeviceBlock(I, n, t["o_shippriority"]))
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
    print(v, "rows", rows, {k: round(x["total_ms"] / x["count"], 3) for k, x in prof.items() if k.startswith("fused") or k.startswith("join")}, flush=True)
    b.close()
    ctx.close()
