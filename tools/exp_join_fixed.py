"""Kernel study: per-launch time of the Q3 lineitem fused filter + probe / emit kernels as a function of the probe page's row count --
the intercept is the launch's fixed cost (DESIGN.md "Page granularity").  Prints one line per page size.

  python tools/exp_join_fixed.py [rows ...]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [2048, 131072, 1 << 20, 1 << 22, 1 << 24]
    args = argparse.Namespace(gpus=1, steps=1, warmup=0, sf=10.0)
    b = bench.Bench(args)
    b.setup_q3(10.0)
    p, ctx, f, pages, t = b.pkg, b.ctx, b.q3_fac, b.q3_pages, b.q3
    B, D, DT, I = p.BIGINT, p.DOUBLE, p.DATE, p.INTEGER
    pp = b.entry.bench_page_processors(p)
    cb = p.HashBuilderOperatorFactory(ctx, 10, [B], [], [0])
    cbuild = cb.createOperator()
    for o in b.drive(f["cust_fp"].createOperator(), pages["customer"]):
        cbuild.addInput(o)
    cbuild.finish()
    oj = p.FilterProjectLookupJoinOperatorFactory(ctx, 11, cb.lookup_source_factory, *pp["q3_orders"], [1], probe_output_channels=[0, 2, 3])
    ob = p.HashBuilderOperatorFactory(ctx, 12, [B, DT, I], [1, 2], [0])
    obuild, ojoin = ob.createOperator(), oj.createOperator()
    for j in b.drive(ojoin, pages["orders"]):
        obuild.addInput(j)
    obuild.finish()
    lj = p.FilterProjectLookupJoinOperatorFactory(ctx, 13, ob.lookup_source_factory, *pp["q3_lineitem"], [0], probe_output_channels=[0, 1])
    ljoin = lj.createOperator()
    ctx.profile_enable(True)
    for n in sizes:
        fixed = lambda ty, key: p.DeviceBlock(ty, n, t[key][:n])
        page = p.Page(fixed(B, "l_orderkey"), fixed(D, "l_extendedprice"), fixed(D, "l_discount"), fixed(DT, "l_shipdate"), position_count=n)
        for _ in range(3):
            ljoin.addInput(page)
            o = ljoin.getOutput()
            if o is not None:
                o.release()
        ctx.profile_reset()
        reps = 30
        for _ in range(reps):
            ljoin.addInput(page)
            o = ljoin.getOutput()
            if o is not None:
                o.release()
        prof = ctx.profile()
        line = {k: round(v["total_ms"] / max(v["count"], 1) * 1e3, 1) for k, v in prof.items() if not k.startswith("__")}
        print(n, line, flush=True)


if __name__ == "__main__":
    main()
