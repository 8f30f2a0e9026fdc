"""Ad-hoc large-size checks against numpy (needs a GPU): PartitionedOutputOperator with 1024 partitions and replicated null-key rows
(5 M rows), byte-exact serde of a page with nulls, a FULL_OUTER join of 2 x 10 M probe rows against 5 M build rows + LookupOuterOperator."""
import importlib, sys, numpy as np
sys.path.insert(0, '/root/repo')
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
