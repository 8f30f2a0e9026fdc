"""Ad-hoc large-size check (5 M rows): PartitionedOutputOperator with 1024 partitions and replicated null-key rows against numpy, and a
byte-exact serde comparison with the oracle on a page with nulls.  python tools/big_check.py (needs a GPU)."""
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
