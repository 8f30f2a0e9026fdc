"""Ad-hoc large-size checks against numpy (needs a GPU): PartitionedOutputOperator with 1024 partitions and replicated null-key rows
(5 M rows), byte-exact serde of a page with nulls, a FULL_OUTER join of 2 x 10 M probe rows against 5 M build rows + LookupOuterOperator."""
import argparse, importlib, sys, numpy as np
sys.path.insert(0, '/root/repo')
import bench                       # (torch first: its HIP runtime has to be the one the process starts with)
bb = bench.Bench(argparse.Namespace())
pkg = importlib.import_module("presto-1_amd")
from oracle import oracle
ctx = pkg.Context(0)
rng = np.random.default_rng(5)
n, P = 5_000_000, 1024
keys = rng.integers(0, 10**9, n).astype(np.int64)
nulls = (rng.random(n) < 0.0005).astype(np.uint8)
pay = np.arange(n, dtype=np.int64)
page = pkg.Page(pkg.Block(pkg.BIGINT, keys, nulls), pkg.Block(pkg.BIGINT, pay))
fac = pkg.PartitionedOutputOperatorFactory(ctx, 1, [pkg.BIGINT, pkg.BIGINT], [0], P, null_channel=0)
op = fac.createOperator()
op.addInput(page)
raw = ctx.hash_page(page, [0])
pid = (raw & 0x7fffffffffffffff) % P
rep = nulls != 0
tot = 0
seen = 0
while True:
    e = op.poll()
    if e is None: break
    p, out = e
    h = out.to_host()
    got = np.asarray(h.blocks[1].values[:h.position_count])
    want = pay[(pid == p) | rep]
    assert np.array_equal(got, want), p
    tot += len(got); seen += 1
    out.release()
print("partitions", seen, "rows", tot, "expected", int((~rep).sum() + rep.sum() * P))
assert tot == int((~rep).sum() + rep.sum() * P)
# serde round trip of a big page with nulls
blocks = [pkg.Block(pkg.BIGINT, keys, nulls), pkg.Block(pkg.DOUBLE, rng.random(n), (rng.random(n) < 0.3).astype(np.uint8))]
pg = pkg.Page(*blocks)
data = ctx.serialize_page(pg)
want = oracle.serialize_page([oracle.Col(oracle.BIGINT, keys, nulls), oracle.Col(oracle.DOUBLE, blocks[1].values, blocks[1].nulls)])
assert data == want, (len(data), len(want))
print("serde ok", len(data))

# FULL_OUTER join at size: 5 M unique build keys (hash layout: sparse keys), 20 M probe rows in two probe operators, counts vs numpy
nb, npb = 5_000_000, 10_000_000
bkeys = rng.permutation(np.arange(nb, dtype=np.int64) * 1000 + 7)          # sparse domain -> hash table + Bloom filter, not DIRECT
bf = pkg.HashBuilderOperatorFactory(ctx, 2, [pkg.BIGINT], [0], [0])
jf = pkg.LookupJoinOperatorFactory(ctx, 3, bf.lookup_source_factory, [pkg.BIGINT], [0], join_type=pkg.FULL_OUTER)
of = pkg.LookupOuterOperatorFactory(ctx, 4, bf.lookup_source_factory, [pkg.BIGINT])
b = bf.createOperator(); b.addInput(pkg.Page(pkg.Block(pkg.BIGINT, bkeys))); b.finish()
outer = of.createOperator()
matched = np.zeros(nb, dtype=bool)
for part in range(2):
    pk = rng.integers(0, nb * 2, npb).astype(np.int64) * 1000 + 7        # half of the probe keys exist in the build side
    op = jf.createOperator()
    op.addInput(pkg.Page(pkg.Block(pkg.BIGINT, pk)))
    out = op.getOutput()
    h = out.to_host()
    got_build_null = int(np.asarray(h.blocks[1].nulls[:h.position_count]).sum())
    assert h.position_count == npb and got_build_null == int((pk >= nb * 1000).sum()), (h.position_count, got_build_null)
    matched[np.unique((pk[pk < nb * 1000] - 7) // 1000)] = True
    out.release(); op.close()
jf.noMoreOperators()
o = outer.getOutput()
want_unmatched = int((~matched).sum())
assert o.position_count == want_unmatched, (o.position_count, want_unmatched)
hv = np.asarray(o.to_host().blocks[1].values[:o.position_count])
inv = np.empty(nb, dtype=np.int64); inv[(bkeys - 7) // 1000] = np.arange(nb)
assert np.array_equal(hv, bkeys[np.sort(inv[~matched])])                 # unmatched build rows in build-position order
o.release()
print("full outer ok: unmatched", want_unmatched)

# fused filter/project -> hash aggregation with many groups (ORDERED mode behind the JIT kernels), several pages, vs numpy
n2, groups = 6_000_000, 500_000
f = pkg.field
fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 5, [pkg.BIGINT, pkg.DOUBLE, pkg.BIGINT], f(2, pkg.BIGINT) < 90, [f(0, pkg.BIGINT), f(1, pkg.DOUBLE) * pkg.constant(2.0, pkg.DOUBLE)],
                                                      [pkg.BIGINT], [0], [(pkg.SUM_DOUBLE, 1), (pkg.COUNT_ALL, -1)], expected_groups=1 << 19)
op = fac.createOperator()
want_sum, want_cnt, first_seen = np.zeros(groups), np.zeros(groups, dtype=np.int64), []
seen = np.zeros(groups, dtype=bool)
for part in range(3):
    k = rng.integers(0, groups, n2).astype(np.int64)
    v = rng.integers(-1000, 1000, n2).astype(np.float64)          # integer-valued: the sums are exact in any order
    sel = rng.integers(0, 100, n2).astype(np.int64)
    op.addInput(pkg.Page(pkg.Block(pkg.BIGINT, k), pkg.Block(pkg.DOUBLE, v), pkg.Block(pkg.BIGINT, sel)))
    m = sel < 90
    want_sum += np.bincount(k[m], weights=2.0 * v[m], minlength=groups)
    want_cnt += np.bincount(k[m], minlength=groups)
    ks = k[m]
    u, idx = np.unique(ks, return_index=True)
    new = u[~seen[u]]
    first_seen.append(new[np.argsort(idx[~seen[u]], kind="stable")])
    seen[u] = True
op.finish()
rows = []
while True:
    o = op.getOutput()
    if o is None:
        if op.isFinished():
            break
        continue
    h = o.to_host()
    rows.append((np.asarray(h.blocks[0].values[:h.position_count]), np.asarray(h.blocks[1].values[:h.position_count]), np.asarray(h.blocks[2].values[:h.position_count])))
    o.release()
gk = np.concatenate([r[0] for r in rows]); gs = np.concatenate([r[1] for r in rows]); gc = np.concatenate([r[2] for r in rows])
order = np.concatenate(first_seen)
assert np.array_equal(gk, order), "group output order is not first-seen order"
assert np.array_equal(gs, want_sum[order]) and np.array_equal(gc, want_cnt[order])
print("fused aggregation with", len(gk), "groups ok")

# GroupByHash: 40 M distinct BIGINT keys in one page (several sub-batches, table growth), then the same page again (pure lookups)
nk = 40_000_000
keys40 = rng.permutation(nk).astype(np.int64) * 3 + 1
gbh = pkg.GroupByHash(ctx, [pkg.BIGINT], [0], expected_size=1 << 20)
ids = gbh.getGroupIds(pkg.Page(pkg.Block(pkg.BIGINT, keys40)))
assert np.array_equal(ids, np.arange(nk)), "first-seen ids of distinct keys must be 0..n-1"
ids2 = gbh.getGroupIds(pkg.Page(pkg.Block(pkg.BIGINT, keys40[::-1].copy())))
assert np.array_equal(ids2, np.arange(nk)[::-1]) and gbh.getGroupCount() == nk
print("group-by hash with", nk, "groups ok")

# INNER join with repeated build keys (position links): 9 M build rows over 3 M keys, 12 M probe rows; pair count and order vs numpy
nbk, nbr, npr = 3_000_000, 9_000_000, 12_000_000
bk = rng.integers(0, nbk, nbr).astype(np.int64) * 11 + 3
bf = pkg.HashBuilderOperatorFactory(ctx, 6, [pkg.BIGINT, pkg.BIGINT], [1], [0])
jf = pkg.LookupJoinOperatorFactory(ctx, 7, bf.lookup_source_factory, [pkg.BIGINT, pkg.BIGINT], [0], probe_output_channels=[1])
b = bf.createOperator(); b.addInput(pkg.Page(pkg.Block(pkg.BIGINT, bk), pkg.Block(pkg.BIGINT, np.arange(nbr, dtype=np.int64)))); b.finish()
pk = rng.integers(0, nbk * 2, npr).astype(np.int64) * 11 + 3
op = jf.createOperator()
op.addInput(pkg.Page(pkg.Block(pkg.BIGINT, pk), pkg.Block(pkg.BIGINT, np.arange(npr, dtype=np.int64))))
out = op.getOutput()
h = out.to_host()
m = h.position_count
probe_pos = np.asarray(h.blocks[0].values[:m]); build_pos = np.asarray(h.blocks[1].values[:m])
mult = np.bincount((bk - 3) // 11, minlength=nbk * 2)
assert m == int(mult[(pk - 3) // 11].sum()), (m, int(mult[(pk - 3) // 11].sum()))
assert np.all(np.diff(probe_pos) >= 0), "probe positions must ascend (LookupJoinPageBuilder)"
assert np.array_equal(bk[build_pos], pk[probe_pos]), "every pair joins equal keys"
same = np.diff(probe_pos) == 0
assert np.all(np.diff(build_pos)[same] < 0), "matches of one probe row come newest build position first (ArrayPositionLinks)"
out.release(); op.close()
print("inner join with duplicates ok:", m, "pairs")

# the fused join over MANY small pages (collected into multi-page launches, two launches in flight, paired launches): 90 pages of
# 150 K - 450 K rows against 5 M sparse-domain build keys (DIRECT layout) with a build output channel; rows and order vs numpy
dom, nb5 = 20_000_000, 5_000_000
bk5 = rng.permutation(dom)[:nb5].astype(np.int64)
pos_of = np.full(dom, -1, dtype=np.int64); pos_of[bk5] = np.arange(nb5)
f, c = pkg.field, pkg.constant
bf = pkg.HashBuilderOperatorFactory(ctx, 8, [pkg.BIGINT, pkg.BIGINT], [1], [0])
b = bf.createOperator(); b.addInput(pkg.Page(pkg.Block(pkg.BIGINT, bk5), pkg.Block(pkg.BIGINT, np.arange(nb5, dtype=np.int64) * 7))); b.finish()
jf = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 9, bf.lookup_source_factory, [pkg.BIGINT, pkg.DATE], f(1, pkg.DATE) > c(9200, pkg.DATE), [f(0, pkg.BIGINT), f(1, pkg.DATE)], [0],
                                                 probe_output_channels=[0, 1])
pages = []
for i in range(90):
    n = int(rng.integers(150_000, 450_000))
    pages.append(pkg.Page(pkg.Block(pkg.BIGINT, rng.integers(0, dom, n).astype(np.int64)), pkg.Block(pkg.DATE, rng.integers(9000, 9400, n).astype(np.int32))))
outs = pkg.to_pages(jf.createOperator(), pages)
gk = np.concatenate([o.getBlock(0).values for o in outs]); gd = np.concatenate([o.getBlock(1).values for o in outs]); gb = np.concatenate([o.getBlock(2).values for o in outs])
wk, wd, wb = [], [], []
for pg in pages:
    k, d = pg.getBlock(0).values, pg.getBlock(1).values
    m = (d > 9200) & (pos_of[k] >= 0)
    wk.append(k[m]); wd.append(d[m]); wb.append(pos_of[k[m]] * 7)
assert np.array_equal(gk, np.concatenate(wk)) and np.array_equal(gd, np.concatenate(wd)) and np.array_equal(gb, np.concatenate(wb))
assert len(outs) < len(pages) // 4, len(outs)       # pages shared launches
print("fused join over", len(pages), "small pages ok:", len(gk), "pairs in", len(outs), "output pages")

# the strict Java order at size: Q1's program over 30 M rows (bench.py's generator and operator, TGPU_SUM_ORDER_JAVA: one workgroup per group, the
# sums as chains fed from LDS) -- the four sums and three averages of every group BIT FOR BIT the sequential (numpy cumsum) sums; then a
# skewed 400 K-group input whose heavy group's lane hands over to a chain
n1 = 30_000_000
bb.ctx.set_double_sum_order(pkg.SUM_ORDER_JAVA)
try:
    bb.setup_q1(n1)
    bb.step_q1()
finally:
    bb.ctx.set_double_sum_order(pkg.SUM_ORDER_EXACT)
rows = [r for pg in bb.q1_result for r in pg]
t = {k: v.cpu().numpy() for k, v in bb.q1.items()}
sel = t["shipdate"] <= 10471
bits = lambda x: np.float64(x).view(np.int64)
for r in rows:
    m = sel & (t["returnflag"] == ord(r[0])) & (t["linestatus"] == ord(r[1]))
    q, e, d, x = t["quantity"][m], t["extendedprice"][m], t["discount"][m], t["tax"][m]
    dp = e * (1.0 - d)
    ch = dp * (1.0 + x)
    want = [np.cumsum(q)[-1], np.cumsum(e)[-1], np.cumsum(dp)[-1], np.cumsum(ch)[-1]]
    cnt = int(m.sum())
    want += [want[0] / cnt, want[1] / cnt, np.cumsum(d)[-1] / cnt]
    assert [int(bits(v)) for v in r[2:9]] == [int(bits(v)) for v in want] and r[9] == cnt, (r[:2], r[2:9], want)
print("Q1 in the Java order over", n1, "rows: 7 double aggregates x", len(rows), "groups bit-identical to the sequential sums")
bb.q1_result = None
n2, g2 = 20_000_000, 400_000
k2 = rng.integers(0, g2, n2).astype(np.int64)
k2[rng.random(n2) < 0.25] = 7                        # one heavy group: 5 M rows
v2 = rng.standard_normal(n2) * 10.0 ** rng.integers(-6, 7, n2)
rows2 = [r for p_ in pkg.to_pages(pkg.HashAggregationOperatorFactory(ctx, 0, [pkg.BIGINT], [0], [(pkg.SUM_DOUBLE, 1), (pkg.COUNT_ALL, -1), (pkg.MAX_BIGINT, 0)],
                                                                     expected_groups=g2).createOperator(),
                                  [pkg.Page(pkg.Block(pkg.BIGINT, k2[a:a + 5_000_000]), pkg.Block(pkg.DOUBLE, v2[a:a + 5_000_000])) for a in range(0, n2, 5_000_000)])
         for r in p_.rows()]
order = np.argsort(k2, kind="stable")
ks, vs = k2[order], v2[order]
starts = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
ends = np.r_[starts[1:], len(ks)]
want2 = {int(ks[s]): (float(np.cumsum(vs[s:e])[-1]), int(e - s)) for s, e in zip(starts, ends) if e - s > 3000 or ks[s] % 1000 == 0}
got2 = {r[0]: r for r in rows2}
assert len(rows2) == len(starts)
for k_, (s_, c_) in want2.items():
    assert int(bits(got2[k_][1])) == int(bits(s_)) and got2[k_][2] == c_ and got2[k_][3] == k_, (k_, got2[k_], s_, c_)
print("skewed ORDERED aggregation:", len(rows2), "groups, heavy group", want2[7][1], "rows, sums in row order bit-identical (", len(want2), "groups checked )")
