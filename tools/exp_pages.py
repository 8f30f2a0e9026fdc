"""Study: the fused Q1 pipeline fed with pages of different sizes (the operators synchronise the stream a few times per page)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
pkg = importlib.import_module("presto-1_amd")
entry = importlib.import_module("__graft_entry__")
dev = torch.device("cuda", 0)
n = int(6_000_379.02 * sf)
t = bench.gen_q1(dev, n)
V, D, DT = pkg.VARCHAR, pkg.DOUBLE, pkg.DATE
pp = entry.bench_page_processors(pkg)
ctx = pkg.Context(0, stream=torch.cuda.current_stream().cuda_stream)
fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 21, *pp["q1"], [V, V], [0, 1], entry.q1_aggregates(pkg), expected_groups=16)


def page(a, z):
    m = z - a
    vb = lambda key: pkg.DeviceBlock(V, m, t[key], None, t["off"][a:z + 1])
    db = lambda ty, key: pkg.DeviceBlock(ty, m, t[key][a:z])
    return pkg.Page(vb("returnflag"), vb("linestatus"), db(D, "quantity"), db(D, "extendedprice"), db(D, "discount"), db(D, "tax"), db(DT, "shipdate"), position_count=m)


types = [V, V, D, D, D, D, DT]
merge = os.environ.get("EXP_MERGE_MB")   # put a MergePagesOperator with this many MB of min/max page size in front
mfac = pkg.MergePagesOperatorFactory(ctx, 20, types, int(merge) << 20, 1 << 27, (int(merge) << 20) * 2) if merge else None
for rows in (n, 1 << 26, 1 << 24, 1 << 22, 1 << 20):
    pages = [page(a, min(a + rows, n)) for a in range(0, n, rows)]
    best = None
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        op = fac.createOperator()
        if mfac is None:
            for pg in pages:
                op.addInput(pg)
        else:
            m = mfac.createOperator()

            def drain():
                while True:
                    o = m.getOutput()
                    if o is None:
                        break
                    op.addInput(o.as_device_page())
                    o.release()
            for pg in pages:
                m.addInput(pg)
                drain()
            m.finish()
            drain()
            m.close()
        op.finish()
        o = op.getOutput()
        res = o.to_host().rows()
        op.close()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(("merged " + merge + " MB  " if merge else "") + f"page rows {rows:>10} pages {len(pages):>5} best {best * 1e3:8.2f} ms  {n / best / 1e9:7.2f} G rows/s  groups {len(res)}", flush=True)
ctx.close()
