"""GPU parity tests (run with -m gpu on an MI355X): every case goes through the C ABI (include/tgpu.h) and is checked
against the CPU oracle on the same inputs -- bit-exact for hashes, group ids, join matches and BIGINT results; DOUBLE
aggregates against the exactly rounded sum (tolerance stated in each test)."""
import importlib
import json
import os

import numpy as np
import pytest

from gpu_common import ocol, rand_block, ulp_diff
from seqpages import sequence_page, sequence_values

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


def blocks_of(pkg, types, columns):
    return [pkg.Block(t, v) for t, v in zip(types, columns)]


# ---------------------------------------------------------------------------------------------------------------------
# K3 hashes
# ---------------------------------------------------------------------------------------------------------------------
def test_hash_page_all_types(pkg, ctx, oracle):
    rng = np.random.default_rng(1)
    n = 5000
    types = [pkg.BIGINT, pkg.INTEGER, pkg.DATE, pkg.DOUBLE, pkg.BOOLEAN, pkg.VARCHAR]
    blocks = [rand_block(pkg, rng, t, n, null_frac=0.1) for t in types]
    blocks[3].values[:4] = [0.0, -0.0, np.nan, np.inf]
    page = pkg.Page(*blocks)
    for chans in ([0], [1], [2], [3], [4], [5], [0, 5], list(range(6))):
        got = ctx.hash_page(page, chans)
        want = oracle.hash_rows([ocol(oracle, blocks[c]) for c in chans])
        assert np.array_equal(got, want), chans


def test_hash_page_long_strings_and_encodings(pkg, ctx, oracle):
    rng = np.random.default_rng(2)
    strs = ["", "a", "hashme"] + ["".join(chr(97 + int(x)) for x in rng.integers(0, 26, k)) for k in (7, 8, 9, 31, 32, 33, 64, 100, 257)]
    flat = pkg.Block(pkg.VARCHAR, strs)
    want = oracle.hash_rows([ocol(oracle, flat)])
    assert np.array_equal(ctx.hash_page(pkg.Page(flat), [0]), want)
    # reference literal (T/operator/scalar/TestVarbinaryFunctions.java:334-335) straight off the GPU
    assert int(ctx.hash_page(pkg.Page(flat), [0])[2]) & (2**64 - 1) == 0xF9D96E0E1165E892
    ids = rng.integers(0, len(strs), 500).astype(np.int32)
    dic = pkg.DictionaryBlock(flat, ids)
    assert np.array_equal(ctx.hash_page(pkg.Page(dic), [0]), want[ids])
    rle = pkg.RunLengthEncodedBlock(pkg.Block(pkg.VARCHAR, ["hashme"]), 17)
    assert np.array_equal(ctx.hash_page(pkg.Page(rle), [0]), np.full(17, want[2]))


# ---------------------------------------------------------------------------------------------------------------------
# K4/K5 GroupByHash
# ---------------------------------------------------------------------------------------------------------------------
def test_group_by_hash_reference_cases(pkg, ctx, oracle):
    # T/operator/TestGroupByHash.java:69-96,121-140 (one-row pages, value == group id)
    gbh = pkg.GroupByHash(ctx, [pkg.BIGINT], [0], input_hash_channel=1, expected_size=100)
    for tries in range(2):
        for value in range(0, 500, 7):
            blk = pkg.Block(pkg.BIGINT, np.array([value], dtype=np.int64))
            page = pkg.Page(blk, pkg.Block(pkg.BIGINT, oracle.hash_rows([ocol(oracle, blk)])))
            ids = gbh.getGroupIds(page)
            assert ids[0] == value // 7
    # :180-202 values i % 50
    vals = np.arange(100, dtype=np.int64) % 50
    blk = pkg.Block(pkg.BIGINT, vals)
    gbh = pkg.GroupByHash(ctx, [pkg.BIGINT], [0], expected_size=100)
    assert np.array_equal(gbh.getGroupIds(pkg.Page(blk)), vals)
    assert gbh.getGroupCount() == 50
    assert gbh.appendValues().getBlock(0).to_list() == list(range(50))


def test_group_by_hash_null_group_and_capacity(pkg, ctx, oracle):
    # :98-118 null group + forced rehash, then contains(0) is false
    gbh = pkg.GroupByHash(ctx, [pkg.BIGINT], [0], expected_size=100)
    o = oracle.BigintGroupByHash(100)
    for blk in (pkg.Block(pkg.BIGINT, [None]), pkg.Block(pkg.BIGINT, sequence_values(pkg.BIGINT, 1, 132748))):
        got = gbh.getGroupIds(pkg.Page(blk))
        assert np.array_equal(got, o.get_group_ids(ocol(oracle, blk)))
    assert not gbh.contains(0, pkg.Page(pkg.Block(pkg.BIGINT, np.array([0], dtype=np.int64))))
    assert gbh.contains(0, pkg.Page(pkg.Block(pkg.BIGINT, [None])))
    assert gbh.getGroupCount() == o.group_count == 132748
    assert gbh.getCapacity() == o.capacity
    assert gbh.getRehashCount() == o.rehash_count


def test_group_by_hash_varchar_append_to(pkg, ctx, oracle):
    # :150-178
    blk = pkg.Block(pkg.VARCHAR, sequence_values(pkg.VARCHAR, 0, 100))
    hashes = oracle.hash_rows([ocol(oracle, blk)])
    gbh = pkg.GroupByHash(ctx, [pkg.VARCHAR], [0], input_hash_channel=1, expected_size=100)
    ids = gbh.getGroupIds(pkg.Page(blk, pkg.Block(pkg.BIGINT, hashes)))
    assert list(ids) == list(range(100))
    out = gbh.appendValues()
    assert out.getBlock(0).to_list() == blk.to_list()
    assert np.array_equal(out.getBlock(1).values, hashes)
    # :237-252 forced rehash from expectedSize 4
    g2 = pkg.GroupByHash(ctx, [pkg.VARCHAR], [0], input_hash_channel=1, expected_size=4)
    page = pkg.Page(blk, pkg.Block(pkg.BIGINT, hashes))
    g2.getGroupIds(page)
    assert all(g2.contains(i, page) for i in range(0, 100, 9))
    assert g2.getCapacity() == 256


def test_group_by_hash_contains_multiple_columns_golden(pkg, ctx, oracle):
    """T/operator/TestGroupByHash.java:221-235 testContainsMultipleColumns: (DOUBLE, VARCHAR) keys with the precomputed hash as channel 2"""
    d = pkg.Block(pkg.DOUBLE, sequence_values(pkg.DOUBLE, 0, 10))
    v = pkg.Block(pkg.VARCHAR, sequence_values(pkg.VARCHAR, 0, 10))
    h = pkg.Block(pkg.BIGINT, oracle.hash_rows([ocol(oracle, d), ocol(oracle, v)]))
    gbh = pkg.GroupByHash(ctx, [pkg.DOUBLE, pkg.VARCHAR], [0, 1], input_hash_channel=2, expected_size=100)
    assert list(gbh.getGroupIds(pkg.Page(d, v, h))) == list(range(10))
    td, tv = pkg.Block(pkg.DOUBLE, [3.0]), pkg.Block(pkg.VARCHAR, ["3"])
    th = pkg.Block(pkg.BIGINT, oracle.hash_rows([ocol(oracle, td), ocol(oracle, tv)]))
    assert gbh.contains(0, pkg.Page(td, tv, th)) is GOLD["group_by_hash"]["testContainsMultipleColumns"]["expect_contains"]
    td = pkg.Block(pkg.DOUBLE, [3.0])
    tv = pkg.Block(pkg.VARCHAR, ["4"])                                    # (one column off: not a group)
    th = pkg.Block(pkg.BIGINT, oracle.hash_rows([ocol(oracle, td), ocol(oracle, tv)]))
    assert not gbh.contains(0, pkg.Page(td, tv, th))


def test_aggregation_operator_golden(pkg, ctx):
    """T/operator/TestAggregationOperator.java: the aggregation without group-by channels.  testMaskWithDirtyNulls (:90-118): a mask whose null
    rows carry non-zero bytes and whose true is any non-zero byte; testAggregation (:119-157) without its max(varchar) / REAL columns;
    testMemoryTracking (:158-198)"""
    G = GOLD["aggregation_operator"]
    case = G["testMaskWithDirtyNulls"]
    mask = pkg.Block(pkg.BOOLEAN, np.array(case["mask_bytes"], dtype=np.uint8), np.array(case["mask_nulls"], dtype=np.uint8))
    page = pkg.Page(pkg.Block(pkg.BIGINT, np.array([1, 2, 3, 4], dtype=np.int64)), mask)
    rows = run_agg(pkg, ctx, [page], [], [], [(pkg.COUNT_COLUMN, 0, 1)])
    assert [list(r) for r in rows] == case["expected_rows"]
    case = G["testAggregation"]
    types = [pkg.VARCHAR, pkg.BIGINT, pkg.VARCHAR, pkg.BIGINT, pkg.DOUBLE, pkg.DOUBLE, pkg.VARCHAR]     # (channel 4 is REAL in the reference: not read here)
    page = pkg.Page(*blocks_of(pkg, types, sequence_page(types, case["rows"], 0, 0, 300, 500, 400, 500, 500)))
    aggs = [(pkg.COUNT_COLUMN, 0), (pkg.SUM_BIGINT, 1), (pkg.AVG_BIGINT, 1), (pkg.COUNT_COLUMN, 0), (pkg.SUM_BIGINT, 3), (pkg.SUM_DOUBLE, 5)]
    (row,) = run_agg(pkg, ctx, [page], [], [], aggs)
    e = case["expected_row_subset"]
    assert list(row) == [e["count"], e["long_sum_1"], e["long_average_1"], e["count_varchar"], e["long_sum_3"], e["double_sum_5"]]
    case = G["testMemoryTracking"]
    op = pkg.HashAggregationOperatorFactory(ctx, 0, [], [], [(pkg.SUM_BIGINT, 0)]).createOperator()
    assert op.needsInput()
    op.addInput(pkg.Page(pkg.Block(pkg.BIGINT, np.arange(case["rows"], dtype=np.int64))))
    assert op.memoryBytes() > 0
    op.finish()
    out = []
    while not op.isFinished():
        o = op.getOutput()
        if o is not None:
            out += o.to_host().rows()
            o.release()
    assert out == [(4950,)] and op.memoryBytes() == 0
    op.close()


@pytest.mark.parametrize("types,domains", [
    (["BIGINT"], [(0, 3000)]),
    (["VARCHAR"], [(0, 500)]),
    (["BIGINT", "DATE", "INTEGER"], [(0, 50), (8000, 8040), (0, 3)]),
    (["DOUBLE", "VARCHAR"], [(0, 30), (0, 30)]),
    (["BOOLEAN", "BIGINT"], [None, (-5, 5)]),
])
@pytest.mark.parametrize("with_hash", [False, True])
def test_group_by_hash_random_vs_oracle(pkg, ctx, oracle, types, domains, with_hash):
    rng = np.random.default_rng(hash((tuple(types), with_hash)) % 2**32)
    tids = [getattr(pkg, t) for t in types]
    gbh = pkg.GroupByHash(ctx, tids, list(range(len(tids))), input_hash_channel=len(tids) if with_hash else None, expected_size=16)
    single_bigint = tids == [pkg.BIGINT]
    o = oracle.BigintGroupByHash(16) if single_bigint else oracle.MultiChannelGroupByHash(tids, 16)
    for n in (1, 257, 20_000, 3, 50_000):
        blocks = [rand_block(pkg, rng, t, n, null_frac=0.05, domain=d) for t, d in zip(tids, domains)]
        ocols = [ocol(oracle, b) for b in blocks]
        hashes = oracle.hash_rows(ocols)
        page = pkg.Page(*(blocks + ([pkg.Block(pkg.BIGINT, hashes)] if with_hash else [])))
        got = gbh.getGroupIds(page)
        want = o.get_group_ids(ocols[0]) if single_bigint else o.get_group_ids(ocols, hashes if with_hash else None)
        assert np.array_equal(got, want)
        assert gbh.getGroupCount() == o.group_count
        assert gbh.getCapacity() == o.capacity


def test_group_by_hash_sub_batches(pkg, oracle, monkeypatch):
    # the device implementation processes big pages in sub-batches; ids must not depend on the split
    monkeypatch.setenv("TGPU_GBH_SUBBATCH", "1000")
    c = pkg.Context(0)
    rng = np.random.default_rng(5)
    blk = rand_block(pkg, rng, pkg.BIGINT, 12_345, null_frac=0.01, domain=(0, 4000))
    gbh = pkg.GroupByHash(c, [pkg.BIGINT], [0], expected_size=10)
    o = oracle.BigintGroupByHash(10)
    assert np.array_equal(gbh.getGroupIds(pkg.Page(blk)), o.get_group_ids(ocol(oracle, blk)))
    gbh.close()
    c.close()


@pytest.mark.parametrize("type_name", ["BIGINT", "INTEGER", "DATE"])
@pytest.mark.parametrize("layout", ["integer_table", "generic_table"])
def test_group_by_hash_single_integer_key_table(pkg, oracle, monkeypatch, type_name, layout):
    """the inline-key table of groupby_bigint.hip (one BIGINT / INTEGER / DATE key, BigintGroupByHash.java) against the oracle and against the
    generic table: the key that equals the table's fill pattern (-1), the NULL group, keys clustered in runs (wave-level followers), keys
    that collide into long probe sequences, several pages, lookups of absent keys"""
    monkeypatch.setenv("TGPU_GBH_INTEGER_TABLE", "1" if layout == "integer_table" else "0")
    t = getattr(pkg, type_name)
    c = pkg.Context(0)
    rng = np.random.default_rng(77)
    dt = np.int64 if t == pkg.BIGINT else np.int32
    gbh = pkg.GroupByHash(c, [t], [0], expected_size=16)
    o = oracle.BigintGroupByHash(16) if t == pkg.BIGINT else oracle.MultiChannelGroupByHash([t], 16)
    pages = []
    runs = np.repeat(rng.integers(-3, 40, 3000), rng.integers(1, 9, 3000)).astype(dt)               # clustered runs incl. -1
    pages.append((runs, rng.random(len(runs)) < 0.02))
    pages.append((rng.integers(-2, 2, 50_000).astype(dt), rng.random(50_000) < 0.3))                  # four keys + many nulls: hot slots
    stride = (1 << 40) if t == pkg.BIGINT else (1 << 20)
    pages.append(((rng.integers(0, 5000, 70_000) * stride).astype(dt), None))                         # multiples of a power of two
    pages.append((rng.integers(-2**31, 2**31 - 1, 100_000).astype(dt), rng.random(100_000) < 0.001))  # mostly distinct
    pages.append((np.array([-1, -1, 0, -1], dtype=dt), np.array([False, True, False, False])))
    for vals, nulls in pages:
        blk = pkg.Block(t, vals, None if nulls is None else nulls.astype(np.uint8))
        oc = ocol(oracle, blk)
        want = o.get_group_ids(oc) if t == pkg.BIGINT else o.get_group_ids([oc], None)
        got = gbh.getGroupIds(pkg.Page(blk))
        assert np.array_equal(got, want)
        assert gbh.getGroupCount() == o.group_count and gbh.getCapacity() == o.capacity
    out = gbh.appendValues()
    if t == pkg.BIGINT:
        v, nl, _ = o.values()
        want_keys = [None if isnull else int(x) for x, isnull in zip(v, nl)]
    else:
        first, _ = o.group_rows()
        want_keys = None
    got_keys = out.getBlock(0).to_list()
    assert len(got_keys) == o.group_count and len(set(got_keys)) == o.group_count
    if want_keys is not None:
        assert got_keys == want_keys
    probe = pkg.Page(pkg.Block(t, np.array([-1, 123456789, 0], dtype=dt)))
    assert gbh.contains(0, probe) and gbh.contains(2, probe)
    assert gbh.contains(1, probe) == (123456789 in set(k for k in got_keys if k is not None))
    gbh.close()
    c.close()


def test_group_by_hash_integer_table_output_hash_is_recomputed_from_the_value(pkg, ctx, oracle):
    """BigintGroupByHash.appendValuesTo (:150-157) writes AbstractLongType.hash(value) / NULL_HASH_CODE -- not the input hash channel"""
    vals = np.array([5, 7, 5, -1, 9, 7], dtype=np.int64)
    nulls = np.array([0, 0, 0, 0, 1, 0], dtype=np.uint8)
    blk = pkg.Block(pkg.BIGINT, vals, nulls)
    hashes = oracle.hash_rows([ocol(oracle, blk)])
    gbh = pkg.GroupByHash(ctx, [pkg.BIGINT], [0], input_hash_channel=1, expected_size=10)
    assert list(gbh.getGroupIds(pkg.Page(blk, pkg.Block(pkg.BIGINT, hashes)))) == [0, 1, 0, 2, 3, 1]
    out = gbh.appendValues()
    assert out.getBlock(0).to_list() == [5, 7, -1, None]
    assert list(out.getBlock(1).values) == [int(hashes[0]), int(hashes[1]), int(hashes[3]), 0]


# ---------------------------------------------------------------------------------------------------------------------
# hash aggregation operator
# ---------------------------------------------------------------------------------------------------------------------
def run_agg(pkg, ctx, pages, group_types, group_channels, aggs, step=0, hash_channel=-1, expected=100):
    f = pkg.HashAggregationOperatorFactory(ctx, 0, group_types, group_channels, aggs, step=step, hash_channel=hash_channel, expected_groups=expected)
    op = f.createOperator()
    out = pkg.to_pages(op, pages)
    op.close()
    rows = []
    for p in out:
        rows.extend(p.rows())
    return rows


def test_hash_aggregation_golden(pkg, ctx, oracle):
    # T/operator/TestHashAggregationOperator.java:161-220 (count, sum, avg, count(col) columns; max(varchar) is out of scope)
    n = GOLD["hash_aggregation"]["testHashAggregation"]["rows"]
    types = [pkg.VARCHAR, pkg.VARCHAR, pkg.VARCHAR, pkg.BIGINT, pkg.BOOLEAN]
    pages = []
    for base in (100_000, 200_000, 300_000):
        cols = sequence_page(types, n, 100, 0, base, 0, 500)
        blocks = blocks_of(pkg, types, cols)
        hashes = oracle.hash_rows([ocol(oracle, blocks[1])])
        pages.append(pkg.Page(*(blocks + [pkg.Block(pkg.BIGINT, hashes)])))
    aggs = [(pkg.COUNT_ALL, -1), (pkg.SUM_BIGINT, 3), (pkg.AVG_BIGINT, 3), (pkg.COUNT_COLUMN, 0), (pkg.COUNT_COLUMN, 4)]
    rows = run_agg(pkg, ctx, pages, [pkg.VARCHAR], [1], aggs, hash_channel=5, expected=100_000)
    assert len(rows) == n
    for i, r in enumerate(rows):  # output is in group-id order = first-seen order = i
        assert r[0] == str(i) and r[2] == 3 and r[3] == 3 * i and r[4] == float(i) and r[5] == 3 and r[6] == 3, (i, r)
    hk = oracle.hash_rows([oracle.Col(pkg.VARCHAR, [str(i) for i in range(0, n, 997)])])
    assert [rows[i][1] for i in range(0, n, 997)] == list(hk)


def test_hash_aggregation_doubles_exact(pkg, ctx, oracle):
    """DOUBLE policy: the GPU sum is the correctly rounded exact sum -> 0 ULP vs the exact oracle; and it equals the Java
    left-to-right sum whenever that sum is exact (integer-valued inputs)."""
    rng = np.random.default_rng(11)
    n, g = 200_000, 7
    keys = rng.integers(0, g, n).astype(np.int64)
    vals = rng.standard_normal(n) * 10.0 ** rng.integers(-12, 13, n)
    ints = rng.integers(-1000, 1000, n).astype(np.float64)
    nulls = (rng.random(n) < 0.1).astype(np.uint8)
    mask = rng.integers(0, 2, n).astype(np.uint8)
    page = pkg.Page(pkg.Block(pkg.BIGINT, keys), pkg.Block(pkg.DOUBLE, vals, nulls), pkg.Block(pkg.DOUBLE, ints), pkg.Block(pkg.BOOLEAN, mask))
    aggs = [(pkg.SUM_DOUBLE, 1), (pkg.AVG_DOUBLE, 1), (pkg.SUM_DOUBLE, 2), (pkg.SUM_DOUBLE, 1, 3), (pkg.COUNT_ALL, -1, 3)]
    rows = run_agg(pkg, ctx, [page], [pkg.BIGINT], [0], aggs)
    o = oracle.BigintGroupByHash(100)
    gids = o.get_group_ids(oracle.Col(pkg.BIGINT, keys))
    ng = o.group_count
    cnt, exact = oracle.agg_double_sum_exact(gids, vals, ng, nulls=nulls)
    _, java_ints = oracle.agg_double_sum(gids, ints, ng)
    cnt_m, exact_m = oracle.agg_double_sum_exact(gids, vals, ng, nulls=nulls, mask=mask)
    count_m = oracle.agg_count(gids, n, ng, mask=mask)
    got = np.array([[r[1], r[2], r[3], r[4]] for r in rows], dtype=np.float64)
    assert ulp_diff(got[:, 0], exact).max() == 0           # tolerance: 0 ULP vs the exact sum
    assert ulp_diff(got[:, 1], exact / cnt).max() == 0     # avg = exact sum / count, one IEEE division
    assert np.array_equal(got[:, 2], java_ints)            # bit-equal to the Java order when that order is exact
    assert ulp_diff(got[:, 3], exact_m).max() == 0
    assert [r[5] for r in rows] == list(count_m)
    # vs the Java-order oracle: differs only by the Java order's own rounding drift, bounded by n * eps * sum|v|
    _, java = oracle.agg_double_sum(gids, vals, ng, nulls=nulls)
    bound = n * np.finfo(np.float64).eps * np.abs(vals).sum()
    assert np.all(np.abs(got[:, 0] - java) <= bound)


def test_hash_aggregation_special_doubles_and_overflow(pkg, ctx, oracle):
    keys = np.array([0, 0, 1, 1, 2, 2, 3, 3, 4], dtype=np.int64)
    vals = np.array([np.inf, 1.0, np.inf, -np.inf, np.nan, 1.0, 1e308, 1e308, 5e-324])
    rows = run_agg(pkg, ctx, [pkg.Page(pkg.Block(pkg.BIGINT, keys), pkg.Block(pkg.DOUBLE, vals))], [pkg.BIGINT], [0], [(pkg.SUM_DOUBLE, 1)])
    got = [r[1] for r in rows]
    assert got[0] == np.inf and np.isnan(got[1]) and np.isnan(got[2]) and got[3] == np.inf and got[4] == 5e-324
    # sum(bigint) overflow -> NUMERIC_VALUE_OUT_OF_RANGE (M/type/BigintOperators.java:47-57)
    big = np.array([2**62, 2**62, 1], dtype=np.int64)
    with pytest.raises(pkg.TgpuError) as e:
        run_agg(pkg, ctx, [pkg.Page(pkg.Block(pkg.BIGINT, np.zeros(3, dtype=np.int64)), pkg.Block(pkg.BIGINT, big))], [pkg.BIGINT], [0], [(pkg.SUM_BIGINT, 1)])
    assert e.value.code == -2
    neg = np.array([-(2**62), -(2**62), -5, 7], dtype=np.int64)
    rows = run_agg(pkg, ctx, [pkg.Page(pkg.Block(pkg.BIGINT, np.zeros(4, dtype=np.int64)), pkg.Block(pkg.BIGINT, neg))], [pkg.BIGINT], [0], [(pkg.SUM_BIGINT, 1)])
    assert rows[0][1] == int(neg.sum())


def test_hash_aggregation_partial_final(pkg, ctx, oracle):
    rng = np.random.default_rng(13)
    n = 30_000
    pages = []
    for _ in range(3):
        pages.append(pkg.Page(rand_block(pkg, rng, pkg.VARCHAR, n, 0.02, (0, 300)), rand_block(pkg, rng, pkg.DOUBLE, n, 0.1),
                              rand_block(pkg, rng, pkg.BIGINT, n, 0.1, (-10**6, 10**6))))
    aggs = [(pkg.COUNT_ALL, -1), (pkg.SUM_DOUBLE, 1), (pkg.AVG_DOUBLE, 1), (pkg.SUM_BIGINT, 2), (pkg.AVG_BIGINT, 2), (pkg.COUNT_COLUMN, 2)]
    single = run_agg(pkg, ctx, pages, [pkg.VARCHAR], [0], aggs)
    # two partial operators over disjoint page sets, one final
    partial_rows = []
    fp = pkg.HashAggregationOperatorFactory(ctx, 0, [pkg.VARCHAR], [0], aggs, step=pkg.PARTIAL)
    partial_pages = []
    for chunk in (pages[:1], pages[1:]):
        op = fp.createOperator()
        partial_pages.extend(pkg.to_pages(op, chunk))
        op.close()
    # intermediate layout: key, then count | (count,sum) per aggregate
    final_aggs = [(pkg.COUNT_ALL, 1), (pkg.SUM_DOUBLE, 2), (pkg.AVG_DOUBLE, 4), (pkg.SUM_BIGINT, 6), (pkg.AVG_BIGINT, 8), (pkg.COUNT_COLUMN, 10)]
    final = run_agg(pkg, ctx, partial_pages, [pkg.VARCHAR], [0], final_aggs, step=pkg.FINAL)
    a = {r[0]: r[1:] for r in single}
    b = {r[0]: r[1:] for r in final}
    assert a.keys() == b.keys()
    # per-group sum of |v|: with this many groups both plans sum in row order (ORDERED mode, like the Java loop) and the partial
    # sums cross the partial/final boundary as rounded doubles (LongDoubleState): the two association orders differ by at most
    # (rows in the group) * eps * sum|v|
    abs_sum = {}
    for pg in pages:
        ks, ds, ls = pg.getBlock(0).to_list(), pg.getBlock(1).to_list(), pg.getBlock(2).to_list()
        for k, d, l in zip(ks, ds, ls):
            e = abs_sum.setdefault(k, [0.0, 0.0])
            e[0] += abs(d) if d is not None else 0.0
            e[1] += abs(l) if l is not None else 0.0
    eps = np.finfo(np.float64).eps
    for k in a:
        assert a[k][0] == b[k][0] and a[k][3] == b[k][3] and a[k][5] == b[k][5]   # counts and bigint sums: exact
        m = a[k][0] + 2   # rows of the group (count(*)) + the roundings at the partial/final boundary
        assert abs(a[k][1] - b[k][1]) <= m * eps * abs_sum[k][0]
        assert abs(a[k][2] - b[k][2]) <= m * eps * abs_sum[k][0] / max(a[k][5], 1) + abs(a[k][2]) * eps
        assert abs(a[k][4] - b[k][4]) <= m * eps * abs_sum[k][1] / max(a[k][5], 1) + abs(a[k][4]) * eps


def test_global_aggregation_and_default_output(pkg, ctx):
    f = pkg.HashAggregationOperatorFactory(ctx, 0, [], [], [(pkg.COUNT_ALL, -1), (pkg.SUM_DOUBLE, 0)], produce_default_output=True)
    op = f.createOperator()
    out = pkg.to_pages(op, [])
    assert [p.rows() for p in out] == [[(0, None)]]
    op = f.createOperator()
    out = pkg.to_pages(op, [pkg.Page(pkg.Block(pkg.DOUBLE, [1.5, None, 2.5]))])
    assert out[0].rows() == [(3, 4.0)]


# ---------------------------------------------------------------------------------------------------------------------
# hash join
# ---------------------------------------------------------------------------------------------------------------------
def run_join(pkg, ctx, build_pages, probe_pages, types_b, types_p, key_b, key_p, hash_b=-1, hash_p=-1, join_type=0, out_b=None, out_p=None):
    out_b = list(range(len(types_b))) if out_b is None else out_b
    bf = pkg.HashBuilderOperatorFactory(ctx, 1, types_b, out_b, key_b, precomputed_hash_channel=hash_b)
    jf = pkg.LookupJoinOperatorFactory(ctx, 2, bf.lookup_source_factory, types_p, key_p, probe_hash_channel=hash_p, probe_output_channels=out_p, join_type=join_type)
    build = bf.createOperator()
    probe = jf.createOperator()
    assert probe.isBlocked() and not probe.needsInput()  # waits on the lookup source future (LookupJoinOperator.java:235-243)
    for p in build_pages:
        assert build.needsInput()
        build.addInput(p)
    build.finish()
    assert not probe.isBlocked()
    out = pkg.to_pages(probe, probe_pages)
    stats = bf.lookup_source_factory.stats()
    probe.close()
    jf.noMoreOperators()
    assert build.isFinished()
    build.close()
    rows = []
    for p in out:
        rows.extend(p.rows())
    return rows, stats


def test_hash_join_empty_probe_and_blocking_lookup_source_golden(pkg, ctx):
    """T/operator/TestHashJoinOperator.java:1200-1238 testInnerJoinWithNonEmptyLookupSourceAndEmptyProbe (build rows a, b, null, c; no probe page:
    no output), :1241-1259 testInnerJoinWithBlockingLookupSourceAndEmptyProbe (the build never finishes: the probe operator does not need input;
    finish() -> no output, not blocked any more, finished) and :1261-1277 testInnerJoinWithBlockingLookupSource (without finish(): no output,
    blocked on the build side, not finished)"""
    V = pkg.VARCHAR
    rows, _ = run_join(pkg, ctx, [pkg.Page(pkg.Block(V, ["a", "b", None, "c"]))], [], [V], [V], [0], [0])
    assert rows == []
    for finish_probe in (True, False):
        bf = pkg.HashBuilderOperatorFactory(ctx, 1, [V], [0], [0])
        jf = pkg.LookupJoinOperatorFactory(ctx, 2, bf.lookup_source_factory, [V], [0])
        build = bf.createOperator()                      # (fed nothing and never finished: the lookup source future stays open)
        probe = jf.createOperator()
        jf.noMoreOperators()
        assert not probe.needsInput()
        if finish_probe:
            probe.finish()
            assert probe.getOutput() is None and probe.getOutput() is None
            assert not probe.isBlocked() and probe.isFinished()
        else:
            assert probe.getOutput() is None
            assert probe.isBlocked() and not probe.isFinished()
        probe.close()
        build.close()


@pytest.mark.parametrize("probe_hash,build_hash", [(False, False), (True, True), (True, False)])
def test_hash_join_inner_golden(pkg, ctx, oracle, probe_hash, build_hash):
    # T/operator/TestHashJoinOperator.java:164-199
    types = [pkg.VARCHAR, pkg.BIGINT, pkg.BIGINT]
    b = blocks_of(pkg, types, sequence_page(types, 10, 20, 30, 40))
    p = blocks_of(pkg, types, sequence_page(types, 1000, 0, 1000, 2000))
    tb, tp = list(types), list(types)
    if build_hash:
        b.append(pkg.Block(pkg.BIGINT, oracle.hash_rows([ocol(oracle, b[0])])))
        tb.append(pkg.BIGINT)
    if probe_hash:
        p.append(pkg.Block(pkg.BIGINT, oracle.hash_rows([ocol(oracle, p[0])])))
        tp.append(pkg.BIGINT)
    rows, stats = run_join(pkg, ctx, [pkg.Page(*b)], [pkg.Page(*p)], tb, tp, [0], [0], hash_b=3 if build_hash else -1, hash_p=3 if probe_hash else -1,
                           out_b=[0, 1, 2], out_p=[0, 1, 2])
    assert [list(r) for r in rows] == GOLD["hash_join"]["testInnerJoin"]["expect_rows"]
    assert stats["positions"] == 10 and stats["link_count"] == 0


@pytest.mark.parametrize("name", ["testInnerJoinWithNullProbe", "testInnerJoinWithNullBuild", "testInnerJoinWithNullOnBothSides"])
def test_hash_join_nulls_golden(pkg, ctx, oracle, name):
    case = GOLD["hash_join"][name]
    rows, _ = run_join(pkg, ctx, [pkg.Page(pkg.Block(pkg.VARCHAR, case["build"]))], [pkg.Page(pkg.Block(pkg.VARCHAR, case["probe"]))],
                       [pkg.VARCHAR], [pkg.VARCHAR], [0], [0])
    assert sorted(list(r) for r in rows) == sorted(case["expect_rows"])


def test_hash_join_probe_outer_golden(pkg, ctx):
    types = [pkg.VARCHAR, pkg.BIGINT, pkg.BIGINT]
    b = blocks_of(pkg, types, sequence_page(types, 10, 20, 30, 40))
    p = blocks_of(pkg, types, sequence_page(types, 15, 20, 1020, 2020))
    rows, _ = run_join(pkg, ctx, [pkg.Page(*b)], [pkg.Page(*p)], types, types, [0], [0], join_type=pkg.PROBE_OUTER)
    assert len(rows) == 15
    for i, r in enumerate(rows):
        want = (str(20 + i), 1020 + i, 2020 + i) + ((str(20 + i), 30 + i, 40 + i) if i < 10 else (None, None, None))
        assert r == want


@pytest.mark.parametrize("types,domains", [
    (["BIGINT"], [(0, 2000)]),
    (["INTEGER"], [(0, 300)]),
    (["VARCHAR"], [(0, 400)]),
    (["BIGINT", "VARCHAR"], [(0, 40), (0, 40)]),
    (["DOUBLE"], [(0, 500)]),
])
def test_hash_join_random_vs_oracle(pkg, ctx, oracle, types, domains):
    """duplicates on both sides, nulls, several build pages: the (probe, build) pair list must equal the Java order
    (probe ascending, matches newest build position first)."""
    rng = np.random.default_rng(abs(hash(tuple(types))) % 2**32)
    tids = [getattr(pkg, t) for t in types]
    nk = len(tids)
    build_pages, bcols = [], [[] for _ in tids]
    for n in (700, 1, 1500):
        blocks = [rand_block(pkg, rng, t, n, 0.03, d) for t, d in zip(tids, domains)]
        payload = pkg.Block(pkg.BIGINT, rng.integers(0, 10**9, n).astype(np.int64))
        build_pages.append(pkg.Page(*(blocks + [payload])))
    probe_blocks = [rand_block(pkg, rng, t, 5000, 0.03, d) for t, d in zip(tids, domains)]
    probe_pay = pkg.Block(pkg.BIGINT, np.arange(5000, dtype=np.int64))
    rows, stats = run_join(pkg, ctx, build_pages, [pkg.Page(*(probe_blocks + [probe_pay]))], tids + [pkg.BIGINT], tids + [pkg.BIGINT],
                           list(range(nk)), list(range(nk)), out_b=[nk], out_p=[nk])
    # oracle over the concatenated build side
    cat = []
    for c in range(nk + 1):
        vals = []
        for pg in build_pages:
            vals.extend(pg.getBlock(c).to_list())
        cat.append(pkg.Block((tids + [pkg.BIGINT])[c], vals))
    ph = oracle.PagesHash([ocol(oracle, b) for b in cat[:nk]])
    op, ob = ph.probe([ocol(oracle, b) for b in probe_blocks])
    want = [(int(i), int(cat[nk].values[j])) for i, j in zip(op, ob)]
    assert rows == want
    assert stats["link_count"] == ph.link_count


def test_hash_join_empty_sides(pkg, ctx):
    rows, stats = run_join(pkg, ctx, [], [pkg.Page(pkg.Block(pkg.BIGINT, np.arange(10, dtype=np.int64)))], [pkg.BIGINT], [pkg.BIGINT], [0], [0])
    assert rows == [] and stats["positions"] == 0
    rows, _ = run_join(pkg, ctx, [], [pkg.Page(pkg.Block(pkg.BIGINT, np.arange(3, dtype=np.int64)))], [pkg.BIGINT], [pkg.BIGINT], [0], [0], join_type=pkg.PROBE_OUTER)
    assert rows == [(0, None), (1, None), (2, None)]
    rows, _ = run_join(pkg, ctx, [pkg.Page(pkg.Block(pkg.BIGINT, np.arange(3, dtype=np.int64)))], [], [pkg.BIGINT], [pkg.BIGINT], [0], [0])
    assert rows == []


# ---------------------------------------------------------------------------------------------------------------------
# filter + project
# ---------------------------------------------------------------------------------------------------------------------
def run_fp(pkg, ctx, pages, types, filt, projs):
    f = pkg.FilterAndProjectOperatorFactory(ctx, 0, types, filt, projs)
    op = f.createOperator()
    out = pkg.to_pages(op, pages)
    op.close()
    return f, out


def oracle_fp(pkg, oracle, factory, page):
    prog = factory.program
    cols = [ocol(oracle, b) for b in page.blocks]
    pos = oracle.filter_positions(prog.nodes, prog.filter_root, bytes(prog.pool), cols) if prog.filter_root >= 0 else np.arange(page.position_count, dtype=np.int32)
    outs = []
    for r in prog.projection_roots:
        if prog.nodes[r]["kind"] == 0:  # identity projection
            vals = page.getBlock(prog.nodes[r]["op"]).flatten().to_list()
            outs.append([vals[i] for i in pos])
        else:
            v, nl = oracle.project(prog.nodes, r, bytes(prog.pool), cols, pos)
            t = prog.nodes[r]["type"]
            conv = float if t == pkg.DOUBLE else (bool if t == pkg.BOOLEAN else int)
            outs.append([None if nl[i] else conv(v[i]) for i in range(len(pos))])
    return pos, outs


def test_filter_project_cfg2_shape(pkg, ctx, oracle):
    # BASELINE config 2 at test size: col0 > 899 (10 %), project col1 * col2, input order preserved
    rng = np.random.default_rng(42)
    n = 300_000
    cols = [rng.integers(0, 1000, n), rng.integers(0, 2**20, n), rng.integers(0, 2**20, n)]
    page = pkg.Page(*[pkg.Block(pkg.BIGINT, c.astype(np.int64)) for c in cols])
    f = pkg.field
    fac, out = run_fp(pkg, ctx, [page], [pkg.BIGINT] * 3, f(0, pkg.BIGINT) > 899, [f(1, pkg.BIGINT) * f(2, pkg.BIGINT), f(0, pkg.BIGINT)])
    pos, want = oracle_fp(pkg, oracle, fac, page)
    assert len(out) == 1 and out[0].position_count == len(pos)
    assert np.array_equal(out[0].getBlock(0).values, np.array(want[0], dtype=np.int64))
    assert np.array_equal(out[0].getBlock(1).values, cols[0][pos])


def test_filter_project_null_protocol_and_short_circuit(pkg, ctx, oracle):
    rng = np.random.default_rng(7)
    n = 20_000
    T = [pkg.BIGINT, pkg.BIGINT, pkg.DOUBLE, pkg.DATE, pkg.BOOLEAN, pkg.VARCHAR, pkg.INTEGER]
    doms = [(-50, 50), (-3, 4), None, (9000, 9100), None, (0, 5), (-100, 100)]
    page = pkg.Page(*[rand_block(pkg, rng, t, n, 0.15, d) for t, d in zip(T, doms)])
    f, c = pkg.field, pkg.constant
    a, b, d, dt, bo, s, i = (f(k, t) for k, t in enumerate(T))
    cases = [
        (pkg.and_(a > 0, b.ne(0)), [a / b, a % b, a + b, a - b, -a]),
        (pkg.or_(a < -10, pkg.and_(bo, dt <= 9050)), [d * (c(1.0, pkg.DOUBLE) - d), d / c(0.0, pkg.DOUBLE), pkg.cast(a, pkg.DOUBLE) + d]),
        (pkg.or_(pkg.is_null(a), s.eq("k3")), [pkg.coalesce(a, b, c(7, pkg.BIGINT)), pkg.if_(bo, a, b), i * i + i, pkg.cast(i, pkg.BIGINT) * a]),
        (pkg.between(a, -5, b), [pkg.not_(bo), a.eq(b), d < c(0.5, pkg.DOUBLE), s < c("k2", pkg.VARCHAR), s, dt]),
        (pkg.and_(pkg.or_(b.eq(0), (a / b) > 1), pkg.not_(pkg.is_null(d))), [a]),   # division guarded by the short circuit
        (None, [a + c(1, pkg.BIGINT), s, pkg.and_(bo, pkg.is_null(a)), c(None, pkg.BIGINT) + a]),
        (c(None, pkg.BOOLEAN), [a]),
    ]
    for filt, projs in cases:
        fac, out = run_fp(pkg, ctx, [page], T, filt, projs)
        pos, want = oracle_fp(pkg, oracle, fac, page)
        if len(pos) == 0:
            assert out == []
            continue
        got = out[0]
        assert got.position_count == len(pos)
        for k in range(len(projs)):
            g = got.getBlock(k).to_list()
            w = want[k]
            if projs[k].type == pkg.DOUBLE:
                gi = [None if x is None else np.float64(x).view(np.int64) for x in g]
                wi = [None if x is None else np.float64(x).view(np.int64) for x in w]
                nan_ok = all((x is None) == (y is None) and (x == y or (np.isnan(np.int64(x).view(np.float64)) and np.isnan(np.int64(y).view(np.float64))))
                             for x, y in zip(gi, wi) if not (x is None and y is None))
                assert nan_ok, k
            else:
                assert g == w, k


def test_filter_project_errors_only_on_selected_rows(pkg, ctx, oracle):
    # multiplyExact overflow is raised only for rows the filter selects (PageProcessor.java:120-136)
    a = np.array([1, 2**62, 3, 2**62], dtype=np.int64)
    sel = np.array([1, 0, 1, 0], dtype=np.int64)
    page = pkg.Page(pkg.Block(pkg.BIGINT, a), pkg.Block(pkg.BIGINT, sel))
    f = pkg.field
    fac, out = run_fp(pkg, ctx, [page], [pkg.BIGINT] * 2, f(1, pkg.BIGINT).eq(1), [f(0, pkg.BIGINT) * 4])
    assert out[0].getBlock(0).to_list() == [4, 12]
    with pytest.raises(pkg.TgpuError) as e:
        run_fp(pkg, ctx, [page], [pkg.BIGINT] * 2, f(1, pkg.BIGINT).eq(0), [f(0, pkg.BIGINT) * 4])
    assert e.value.code == -2
    with pytest.raises(oracle.OracleError) as oe:
        prog = pkg.expressions.FlatProgram(f(1, pkg.BIGINT).eq(0), [f(0, pkg.BIGINT) * 4])
        cols = [ocol(oracle, b) for b in page.blocks]
        oracle.project(prog.nodes, prog.projection_roots[0], b"", cols, oracle.filter_positions(prog.nodes, prog.filter_root, b"", cols))
    assert oe.value.code == -2
    with pytest.raises(pkg.TgpuError) as e:
        run_fp(pkg, ctx, [page], [pkg.BIGINT] * 2, (f(0, pkg.BIGINT) / (f(1, pkg.BIGINT) - 1)) > 0, [])
    assert e.value.code == -7


def test_filter_project_select_all_none_and_encodings(pkg, ctx):
    f = pkg.field
    blk = pkg.Block(pkg.BIGINT, np.arange(1000, dtype=np.int64))
    dic = pkg.DictionaryBlock(pkg.Block(pkg.VARCHAR, ["x", "y", None]), np.arange(1000, dtype=np.int32) % 3)
    page = pkg.Page(blk, dic)
    _, out = run_fp(pkg, ctx, [page], [pkg.BIGINT, pkg.VARCHAR], f(0, pkg.BIGINT) >= 0, [f(1, pkg.VARCHAR), f(0, pkg.BIGINT) + 1])
    assert out[0].position_count == 1000 and out[0].getBlock(0).to_list()[:4] == ["x", "y", None, "x"]
    _, out = run_fp(pkg, ctx, [page], [pkg.BIGINT, pkg.VARCHAR], f(0, pkg.BIGINT) < 0, [f(0, pkg.BIGINT)])
    assert out == []
    _, out = run_fp(pkg, ctx, [page, pkg.Page(pkg.Block(pkg.BIGINT, []), pkg.Block(pkg.VARCHAR, []))], [pkg.BIGINT, pkg.VARCHAR], f(1, pkg.VARCHAR).eq("y"), [])
    assert len(out) == 1 and out[0].position_count == 333 and out[0].getChannelCount() == 0


# ---------------------------------------------------------------------------------------------------------------------
# K10 partition
# ---------------------------------------------------------------------------------------------------------------------
def test_partition_page_vs_oracle(pkg, ctx, oracle):
    rng = np.random.default_rng(3)
    n = 40_000
    blocks = [rand_block(pkg, rng, pkg.BIGINT, n, 0.02, (0, 10**6)), rand_block(pkg, rng, pkg.VARCHAR, n, 0.02, (0, 50)), rand_block(pkg, rng, pkg.DOUBLE, n, 0.1)]
    page = pkg.Page(*blocks)
    for parts in (1, 2, 8, 7):
        counts, out = ctx.partition_page(page, [0, 1], parts)
        host = out.to_host()
        out.release()
        raw = oracle.hash_rows([ocol(oracle, blocks[0]), ocol(oracle, blocks[1])])
        pid = oracle.partition_remote(raw, parts)
        order = np.argsort(pid, kind="stable")
        assert list(counts) == [int((pid == p).sum()) for p in range(parts)]
        rows = page.rows()
        assert host.rows() == [rows[i] for i in order]


def test_lowcard_and_general_accumulators_agree_bitwise(pkg, oracle, monkeypatch):
    """the lane-private LDS path (few groups) and the global-atomics path feed the same exact accumulators: identical bits"""
    rng = np.random.default_rng(21)
    n = 300_000
    keys = rng.integers(0, 4, n).astype(np.int64)
    page = pkg.Page(pkg.Block(pkg.BIGINT, keys), rand_block(pkg, rng, pkg.DOUBLE, n, 0.05), rand_block(pkg, rng, pkg.BIGINT, n, 0.05, (-10**12, 10**12)),
                    pkg.Block(pkg.BOOLEAN, rng.integers(0, 2, n).astype(np.uint8)))
    aggs = [(pkg.SUM_DOUBLE, 1), (pkg.AVG_DOUBLE, 1, 3), (pkg.SUM_BIGINT, 2), (pkg.AVG_BIGINT, 2), (pkg.COUNT_COLUMN, 1), (pkg.COUNT_ALL, -1, 3)]
    results = []
    for disable in (False, True):
        if disable:
            monkeypatch.setenv("TGPU_DISABLE_LOWCARD", "1")
        c = pkg.Context(0)
        c.profile_enable(True)
        results.append(run_agg(pkg, c, [page], [pkg.BIGINT], [0], aggs))
        prof = c.profile()
        assert ("agg_accumulate_lowcard" in prof) == (not disable)
        c.close()
    a, b = results
    assert len(a) == 4 and [r[0] for r in a] == [r[0] for r in b]
    for ra, rb in zip(a, b):
        assert ra[3] == rb[3] and ra[5] == rb[5] and ra[6] == rb[6]
        assert ulp_diff([ra[1], ra[2], ra[4]], [rb[1], rb[2], rb[4]]).max() == 0
    # and both equal the exact oracle
    o = oracle.BigintGroupByHash(8)
    gids = o.get_group_ids(oracle.Col(pkg.BIGINT, keys))
    _, exact = oracle.agg_double_sum_exact(gids, page.getBlock(1).values, 4, nulls=page.getBlock(1).nulls)
    assert ulp_diff([r[1] for r in a], exact).max() == 0


# ---------------------------------------------------------------------------------------------------------------------
# operator fusion: FilterAndProject + LookupJoin in one generated kernel == the two reference operators back to back
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("join_type", [0, 1])
@pytest.mark.parametrize("key_type", ["BIGINT", "INTEGER"])
def test_fused_filter_project_join_matches_unfused_and_oracle(pkg, oracle, monkeypatch, join_type, key_type):
    rng = np.random.default_rng(17)
    kt = getattr(pkg, key_type)
    nb, n = 20_000, 150_000
    bkeys = rng.permutation(60_000)[:nb]
    build = pkg.Page(pkg.Block(kt, bkeys.astype(np.int64 if kt == pkg.BIGINT else np.int32)), pkg.Block(pkg.DOUBLE, rng.standard_normal(nb)),
                     rand_block(pkg, rng, pkg.VARCHAR, nb, 0.05, (0, 9)))
    T = [kt, pkg.DOUBLE, pkg.DOUBLE, pkg.DATE]
    probe = pkg.Page(rand_block(pkg, rng, kt, n, 0.03, (0, 60_000)), rand_block(pkg, rng, pkg.DOUBLE, n, 0.05), rand_block(pkg, rng, pkg.DOUBLE, n, 0.0),
                     rand_block(pkg, rng, pkg.DATE, n, 0.02, (9000, 9400)))
    f, c = pkg.field, pkg.constant
    filt = f(3, pkg.DATE) > 9204
    projs = [f(0, kt), f(1, pkg.DOUBLE) * (c(1.0, pkg.DOUBLE) - f(2, pkg.DOUBLE)), f(3, pkg.DATE)]
    results = {}
    for mode in ("fused", "unfused"):
        if mode == "unfused":
            monkeypatch.setenv("TGPU_DISABLE_FUSION", "1")
        ctx = pkg.Context(0)
        ctx.profile_enable(True)
        bf = pkg.HashBuilderOperatorFactory(ctx, 1, [kt, pkg.DOUBLE, pkg.VARCHAR], [1, 2], [0])
        jf = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 2, bf.lookup_source_factory, T, filt, projs, [0], probe_output_channels=[1, 0, 2], join_type=join_type)
        b = bf.createOperator()
        b.addInput(build)
        b.finish()
        op = jf.createOperator()
        out = pkg.to_pages(op, [probe, pkg.Page(*[pkg.Block(t, []) for t in T]), probe])
        results[mode] = [r for p in out for r in p.rows()]
        prof = ctx.profile()
        assert ("fused_filter_probe" in prof) == (mode == "fused")
        op.close(); b.close(); ctx.close()
    assert results["fused"] == results["unfused"]
    # oracle composition: filter positions -> projections -> PagesHash probe
    prog = pkg.expressions.FlatProgram(filt, projs)
    cols = [ocol(oracle, blk) for blk in probe.blocks]
    pos = oracle.filter_positions(prog.nodes, prog.filter_root, b"", cols)
    rev, rev_null = oracle.project(prog.nodes, prog.projection_roots[1], b"", cols, pos)
    keyblk = probe.getBlock(0)
    kcol = oracle.Col(kt, keyblk.values[pos], None if keyblk.nulls is None else keyblk.nulls[pos])
    ph = oracle.PagesHash([ocol(oracle, build.getBlock(0))])
    opx, obx = ph.probe([kcol], probe_outer=bool(join_type))
    bl = build.rows()
    want = []
    for i, j in zip(opx, obx):
        r = pos[i]
        want.append((None if rev_null[i] else float(rev[i]), keyblk.get(int(r)), probe.getBlock(3).get(int(r))) + ((bl[j][1], bl[j][2]) if j >= 0 else (None, None)))
    assert results["fused"] == want + want


def test_fused_join_carry_variant_matches_the_two_pass_gather(pkg, monkeypatch):
    """TGPU_FJ_CARRY=1: the probe pass evaluates the probe-side outputs from its row registers and stores them with the pairs, the emit
    pass moves them (jit.cpp FJ_CARRY; opt-in): same rows as the default two-pass gather, null outputs included, on the
    bitmap and the DIRECT table layouts"""
    rng = np.random.default_rng(89)
    nb, n = 40_000, 300_000
    T = [pkg.BIGINT, pkg.DOUBLE, pkg.DOUBLE, pkg.DATE, pkg.INTEGER]
    probe = pkg.Page(rand_block(pkg, rng, pkg.BIGINT, n, 0.03, (0, 100_000)), rand_block(pkg, rng, pkg.DOUBLE, n, 0.05), rand_block(pkg, rng, pkg.DOUBLE, n, 0.0),
                     rand_block(pkg, rng, pkg.DATE, n, 0.02, (9000, 9400)), rand_block(pkg, rng, pkg.INTEGER, n, 0.1, (-5, 5)))
    f, c = pkg.field, pkg.constant
    filt = f(3, pkg.DATE) > 9150
    projs = [f(0, pkg.BIGINT), f(1, pkg.DOUBLE) * (c(1.0, pkg.DOUBLE) - f(2, pkg.DOUBLE)), f(3, pkg.DATE), f(4, pkg.INTEGER)]
    for dup in (False, True):     # unique ascending keys -> DIRECT layout; a repeated key -> hash table + exact bitmap
        bkeys = np.sort(rng.permutation(100_000)[:nb]).astype(np.int64)
        if dup:
            bkeys[7] = bkeys[6]
        build = pkg.Page(pkg.Block(pkg.BIGINT, bkeys), pkg.Block(pkg.BIGINT, np.arange(nb, dtype=np.int64)))
        got = {}
        for carry in ("0", "1"):
            monkeypatch.setenv("TGPU_FJ_CARRY", carry)
            ctx = pkg.Context(0)
            ctx.profile_enable(True)
            bf = pkg.HashBuilderOperatorFactory(ctx, 1, [pkg.BIGINT, pkg.BIGINT], [1], [0])
            jf = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 2, bf.lookup_source_factory, T, filt, projs, [0], probe_output_channels=[1, 0, 3, 2])
            b = bf.createOperator()
            b.addInput(build)
            b.finish()
            op = jf.createOperator()
            got[carry] = [r for p in pkg.to_pages(op, [probe, probe]) for r in p.rows()]
            assert "fused_filter_probe" in ctx.profile() or dup       # (duplicate build keys: position links -> unfused composition)
            op.close(); b.close(); ctx.close()
        assert len(got["1"]) > 50_000 and got["1"] == got["0"]
        assert any(r[0] is None for r in got["1"]) and any(r[2] is None for r in got["1"])


def test_fused_join_falls_back_when_projection_can_raise(pkg, ctx):
    # a checked BIGINT projection must still raise for selected rows that do NOT match (FilterAndProject semantics)
    build = pkg.Page(pkg.Block(pkg.BIGINT, np.array([1, 2], dtype=np.int64)))
    probe = pkg.Page(pkg.Block(pkg.BIGINT, np.array([1, 5], dtype=np.int64)), pkg.Block(pkg.BIGINT, np.array([3, 2**62], dtype=np.int64)))
    f = pkg.field
    bf = pkg.HashBuilderOperatorFactory(ctx, 1, [pkg.BIGINT], [], [0])
    jf = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 2, bf.lookup_source_factory, [pkg.BIGINT] * 2, None, [f(0, pkg.BIGINT), f(1, pkg.BIGINT) * 4], [0])
    b = bf.createOperator()
    b.addInput(build)
    b.finish()
    with pytest.raises(pkg.TgpuError) as e:
        pkg.to_pages(jf.createOperator(), [probe])
    assert e.value.code == -2


@pytest.mark.parametrize("stride", [1, 10**12])
def test_join_prefilters_dense_bitmap_and_sparse_bloom(pkg, ctx, oracle, stride):
    """dense key domains use the exact bitmap pre-filter, sparse ones the Bloom filter: same pairs as the oracle either way"""
    rng = np.random.default_rng(23)
    bkeys = (rng.permutation(300_000)[:100_000].astype(np.int64) - 150_000) * stride
    pkeys = (rng.integers(-200_000, 200_000, 400_000).astype(np.int64)) * stride
    rows, stats = run_join(pkg, ctx, [pkg.Page(pkg.Block(pkg.BIGINT, bkeys), pkg.Block(pkg.BIGINT, np.arange(len(bkeys), dtype=np.int64)))],
                           [pkg.Page(pkg.Block(pkg.BIGINT, pkeys))], [pkg.BIGINT] * 2, [pkg.BIGINT], [0], [0], out_b=[1], out_p=[0])
    op, ob = oracle.PagesHash([oracle.Col(pkg.BIGINT, bkeys)]).probe([oracle.Col(pkg.BIGINT, pkeys)])
    assert rows == [(int(pkeys[i]), int(j)) for i, j in zip(op, ob)]
    # and through the fused filter+project+probe kernel
    f = pkg.field
    bf = pkg.HashBuilderOperatorFactory(ctx, 1, [pkg.BIGINT], [0], [0])
    jf = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 2, bf.lookup_source_factory, [pkg.BIGINT], None, [f(0, pkg.BIGINT)], [0])
    b = bf.createOperator()
    b.addInput(pkg.Page(pkg.Block(pkg.BIGINT, bkeys)))
    b.finish()
    out = pkg.to_pages(jf.createOperator(), [pkg.Page(pkg.Block(pkg.BIGINT, pkeys))])
    got = np.concatenate([p.getBlock(0).values for p in out])
    assert np.array_equal(got, pkeys[op])


@pytest.mark.parametrize("n", [1, 63, 64, 255, 767, 768, 769, 1023, 1024, 1025, 4097, 65_537, 13_000_003])
def test_fused_filter_probe_tile_and_chunk_boundaries(pkg, ctx, oracle, n):
    """the software-pipelined fused filter+probe at page sizes around its tile (768 rows by default, 1024 with TGPU_FJ_STRIPES=4) and chunk boundaries; the largest
    size makes a workgroup take chunks of several consecutive tiles.  Expected = numpy filter + the oracle's probe."""
    rng = np.random.default_rng(41 + n % 97)
    bkeys = rng.permutation(400_000)[:120_000].astype(np.int64) * 3 + 7
    pkeys = rng.integers(0, 1_300_000, n).astype(np.int64)
    dates = rng.integers(9000, 9400, n).astype(np.int32)
    f, c = pkg.field, pkg.constant
    bf = pkg.HashBuilderOperatorFactory(ctx, 1, [pkg.BIGINT], [0], [0])
    b = bf.createOperator()
    b.addInput(pkg.Page(pkg.Block(pkg.BIGINT, bkeys)))
    b.finish()
    jf = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 2, bf.lookup_source_factory, [pkg.BIGINT, pkg.DATE], f(1, pkg.DATE) > c(9200, pkg.DATE),
                                                     [f(0, pkg.BIGINT), f(1, pkg.DATE)], [0], probe_output_channels=[0, 1])
    out = pkg.to_pages(jf.createOperator(), [pkg.Page(pkg.Block(pkg.BIGINT, pkeys), pkg.Block(pkg.DATE, dates))])
    sel = np.nonzero(dates > 9200)[0]
    op, ob = oracle.PagesHash([oracle.Col(pkg.BIGINT, bkeys)]).probe([oracle.Col(pkg.BIGINT, pkeys[sel])])
    want_rows = sel[op]
    got_k = np.concatenate([p.getBlock(0).values for p in out]) if out else np.zeros(0, dtype=np.int64)
    got_d = np.concatenate([p.getBlock(1).values for p in out]) if out else np.zeros(0, dtype=np.int32)
    got_b = np.concatenate([p.getBlock(2).values for p in out]) if out else np.zeros(0, dtype=np.int64)
    assert np.array_equal(got_k, pkeys[want_rows]) and np.array_equal(got_d, dates[want_rows])
    assert np.array_equal(got_b, bkeys[ob])     # the build side's key column, gathered at the matching build positions
    b.close()


@pytest.mark.parametrize("join_type", [0, 1])
def test_fused_join_keeps_two_small_pages_in_flight(pkg, oracle, join_type, monkeypatch):
    """Probe pages the operator may keep are probed asynchronously (operators.cpp FusedFilterProjectJoinOperator).  Four protocols, same rows in
    the same order, equal to the oracle:
      batched  (default) small pages are collected and probed as ONE sequence of rows per launch (the kernels' multi variant over a page list);
      async    (TGPU_DISABLE_PROBE_BATCHING) one launch per page, two pages deep: pass 1 of the new page shares its launch with pass 2 of
               the page before last (fj_pair); getOutput returns null until a page's successors have been added or finish() was called;
      unpaired (+ TGPU_DISABLE_PROBE_PAIRING) the same with separate launches;
      sync     (TGPU_DISABLE_ASYNC_JOIN) a page's output is there when addInput returns.
    Page sizes around the 768-row tile, an empty page, a page without a match, pages with and without null vectors."""
    rng = np.random.default_rng(211)
    bkeys = rng.permutation(50_000)[:20_000].astype(np.int64)
    f, c = pkg.field, pkg.constant
    sizes = [5_000, 1, 0, 70_000, 300, 4_096, 12_345, 767, 768, 769, 1, 1_536, 100_000, 2]
    pages = []
    for i, n in enumerate(sizes):
        knulls = (rng.random(n) < 0.05).astype(np.uint8) if i % 3 == 0 else None
        dnulls = (rng.random(n) < 0.05).astype(np.uint8) if i % 4 == 1 else None
        pages.append(pkg.Page(pkg.Block(pkg.BIGINT, rng.integers(0, 60_000, n).astype(np.int64), knulls), pkg.Block(pkg.DATE, rng.integers(9000, 9400, n).astype(np.int32), dnulls)))
    pages[4] = pkg.Page(pkg.Block(pkg.BIGINT, np.full(300, 10**9, dtype=np.int64)), pkg.Block(pkg.DATE, np.full(300, 9300, dtype=np.int32)))   # no match at all
    got = {}
    for mode in ("batched", "async", "unpaired", "sync"):
        if mode != "batched":
            monkeypatch.setenv("TGPU_DISABLE_PROBE_BATCHING", "1")
        if mode == "unpaired":
            monkeypatch.setenv("TGPU_DISABLE_PROBE_PAIRING", "1")
        if mode == "sync":
            monkeypatch.setenv("TGPU_DISABLE_ASYNC_JOIN", "1")
        ctx = pkg.Context(0)
        ctx.profile_enable(True)
        bf = pkg.HashBuilderOperatorFactory(ctx, 1, [pkg.BIGINT], [0], [0])
        b = bf.createOperator()
        b.addInput(pkg.Page(pkg.Block(pkg.BIGINT, bkeys)))
        b.finish()
        jf = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 2, bf.lookup_source_factory, [pkg.BIGINT, pkg.DATE], f(1, pkg.DATE) > c(9200, pkg.DATE),
                                                         [f(0, pkg.BIGINT), f(1, pkg.DATE)], [0], probe_output_channels=[0, 1], join_type=join_type)
        op = jf.createOperator()
        outs, nulls_seen = [], 0
        for i, pg in enumerate(pages):
            assert op.needsInput()
            op.addInput(pg)
            o = op.getOutput()
            if o is None:
                nulls_seen += 1
            else:
                outs.append(o.to_host()); o.release()
            if mode != "sync" and i == 0:
                assert o is None and op.needsInput()     # one page collected / in flight: nothing to hand over yet, room for the next page
        op.finish()
        assert not op.needsInput()
        while not op.isFinished():
            o = op.getOutput()
            if o is not None:
                outs.append(o.to_host()); o.release()
        got[mode] = [r for pg in outs for r in pg.rows()]
        prof = ctx.profile()
        if mode == "sync":
            assert 1 <= nulls_seen <= 5       # the empty page; a tiny page the filter drops; inner join: the page without a match
        if mode == "batched":
            assert nulls_seen == len(pages) and len(outs) == 1 and prof["fused_filter_probe"]["count"] == 1    # one launch, one output page
        assert ("fused_probe_pair" in prof) == (mode == "async")
        op.close(); b.close(); ctx.close()
    assert got["batched"] == got["async"] == got["sync"] == got["unpaired"]
    ph = oracle.PagesHash([oracle.Col(pkg.BIGINT, bkeys)])
    want = []
    for pg in pages:
        kb, db = pg.getBlock(0), pg.getBlock(1)
        k, d = kb.values, db.values
        dn = db.nulls if db.nulls is not None else np.zeros(len(d), dtype=np.uint8)
        sel = np.nonzero((d > 9200) & (dn == 0))[0]
        if len(sel) == 0:
            continue
        kn = None if kb.nulls is None else kb.nulls[sel]
        opx, obx = ph.probe([oracle.Col(pkg.BIGINT, k[sel], kn)], probe_outer=bool(join_type))
        want += [(None if (kn is not None and kn[i]) else int(k[sel[i]]), int(d[sel[i]]), int(bkeys[j]) if j >= 0 else None) for i, j in zip(opx, obx)]
    assert got["async"] == want and len(want) > 10_000


def _drain(ops):
    """host rows of what finished operators still hold, page by page"""
    pages = []
    for op in ops:
        while not op.isFinished():
            o = op.getOutput()
            if o is not None:
                pages.append(o.to_host().rows())
                o.release()
    return pages


def test_fused_join_closed_with_a_page_in_flight_and_errors_of_in_flight_pages(pkg, ctx, monkeypatch):
    """close() with a probe page in flight gives its signal slot back (100 operators: more than the context has slots); an expression error
    of an in-flight page is raised by the getOutput that completes it, and the operator stays usable for close()"""
    f = pkg.field
    bf = pkg.HashBuilderOperatorFactory(ctx, 1, [pkg.BIGINT], [], [0])
    b = bf.createOperator()
    b.addInput(pkg.Page(pkg.Block(pkg.BIGINT, np.arange(100, dtype=np.int64))))
    b.finish()
    jf = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 2, bf.lookup_source_factory, [pkg.BIGINT] * 2, None, [f(0, pkg.BIGINT), f(1, pkg.BIGINT)], [0])
    page = pkg.Page(pkg.Block(pkg.BIGINT, np.arange(50, 150, dtype=np.int64)), pkg.Block(pkg.BIGINT, np.arange(100, dtype=np.int64)))
    monkeypatch.setenv("TGPU_DISABLE_PROBE_BATCHING", "1")    # (every page gets its own launch: it is in flight when the operator is closed)
    for _ in range(100):
        op = jf.createOperator()
        op.addInput(page)
        assert op.getOutput() is None
        op.close()
    monkeypatch.delenv("TGPU_DISABLE_PROBE_BATCHING")
    # the filter divides by zero on a selected row of the SECOND page: raised when that page is completed, after the first page's output
    jf2 = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 3, bf.lookup_source_factory, [pkg.BIGINT] * 2, (pkg.constant(100, pkg.BIGINT) / f(1, pkg.BIGINT)) > 0,
                                                      [f(0, pkg.BIGINT), f(1, pkg.BIGINT)], [0])
    good = pkg.Page(pkg.Block(pkg.BIGINT, np.arange(50, 150, dtype=np.int64)), pkg.Block(pkg.BIGINT, np.arange(1, 101, dtype=np.int64)))
    # (a) pages that share a launch fail together: the error comes out of the call that completes the launch
    op = jf2.createOperator()
    op.addInput(good)
    assert op.getOutput() is None
    op.addInput(page)                       # second column holds a zero
    assert op.getOutput() is None
    op.finish()
    with pytest.raises(pkg.TgpuError) as e:
        op.getOutput()
    assert e.value.code == -7               # DIVISION_BY_ZERO
    assert op.isFinished()
    op.close()
    # (b) one launch per page: the good page's output first, then the error
    monkeypatch.setenv("TGPU_DISABLE_PROBE_BATCHING", "1")
    op = jf2.createOperator()
    op.addInput(good)
    assert op.getOutput() is None
    op.addInput(page)
    assert op.getOutput() is None           # two pages in flight: nothing is handed out while the driver keeps bringing pages
    op.finish()
    o = op.getOutput()
    assert o is not None and o.position_count == 50
    o.release()
    with pytest.raises(pkg.TgpuError) as e:
        op.getOutput()
    assert e.value.code == -7
    assert op.isFinished()
    op.close()
    monkeypatch.delenv("TGPU_DISABLE_PROBE_BATCHING")
    # a driver that polls twice without bringing a page gets what is in flight (a slow source does not hold finished work back)
    op = jf.createOperator()
    op.addInput(page)
    assert op.getOutput() is None
    o = op.getOutput()
    assert o is not None and o.position_count == 50
    o.release()
    op.close()
    # every signal slot of the context taken (64 operators with a launch in flight): a further operator's single page falls back to the scan
    # launch + copy, its collected pages are probed one by one -- same rows
    monkeypatch.setenv("TGPU_DISABLE_PROBE_BATCHING", "1")
    holders = [jf.createOperator() for _ in range(64)]
    for h in holders:
        h.addInput(page)
    one = jf.createOperator()
    one.addInput(page)
    monkeypatch.delenv("TGPU_DISABLE_PROBE_BATCHING")
    many = jf.createOperator()
    for _ in range(3):
        many.addInput(page)
    for o_ in (one, many):
        o_.finish()
    rows_one = [r for pg in _drain(o_ for o_ in [one]) for r in pg]
    rows_many = [r for pg in _drain(o_ for o_ in [many]) for r in pg]
    want = [(k, k - 50) for k in range(50, 100)]
    assert rows_one == want and rows_many == want * 3
    for h in holders:
        h.close()
    one.close(); many.close(); b.close()


@pytest.mark.parametrize("layout", ["clustered", "spread", "duplicates"])
def test_join_int_table_key_layouts(pkg, ctx, oracle, layout):
    """int-key table under key sets that stress its slot hash (runs of consecutive keys far apart, the TPCH orderkey pattern,
    duplicate build keys): the pairs equal the oracle's in all cases"""
    rng = np.random.default_rng(31)
    if layout == "clustered":      # two runs of consecutive keys at the ends of a 2e9-wide domain
        bkeys = np.concatenate([np.arange(200_000), 2_000_000_000 + np.arange(200_000)]).astype(np.int64)
    elif layout == "spread":       # TPCH-like: 8 consecutive keys, then a gap
        og = np.sort(rng.permutation(1_000_000)[:200_000])
        bkeys = ((og // 8) * 32 + og % 8 + 1).astype(np.int64)
    else:                          # duplicates on the build side: position links
        bkeys = rng.integers(0, 50_000, 200_000).astype(np.int64)
    pkeys = np.concatenate([rng.choice(bkeys, 150_000), rng.integers(-10, int(bkeys.max()) + 10, 150_000)]).astype(np.int64)
    rows, stats = run_join(pkg, ctx, [pkg.Page(pkg.Block(pkg.BIGINT, bkeys), pkg.Block(pkg.BIGINT, np.arange(len(bkeys), dtype=np.int64)))],
                           [pkg.Page(pkg.Block(pkg.BIGINT, pkeys))], [pkg.BIGINT] * 2, [pkg.BIGINT], [0], [0], out_b=[1], out_p=[0])
    op, ob = oracle.PagesHash([oracle.Col(pkg.BIGINT, bkeys)]).probe([oracle.Col(pkg.BIGINT, pkeys)])
    assert rows == [(int(pkeys[i]), int(j)) for i, j in zip(op, ob)]
    if layout != "duplicates":     # and through the fused filter+project+probe kernel (no duplicate build keys there)
        f = pkg.field
        bf = pkg.HashBuilderOperatorFactory(ctx, 1, [pkg.BIGINT], [0], [0])
        jf = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 2, bf.lookup_source_factory, [pkg.BIGINT], None, [f(0, pkg.BIGINT)], [0])
        b = bf.createOperator()
        b.addInput(pkg.Page(pkg.Block(pkg.BIGINT, bkeys)))
        b.finish()
        out = pkg.to_pages(jf.createOperator(), [pkg.Page(pkg.Block(pkg.BIGINT, pkeys))])
        got = np.concatenate([p.getBlock(0).values for p in out])
        assert np.array_equal(got, pkeys[op])


@pytest.mark.parametrize("step", ["single", "partial_final"])
def test_hash_aggregation_many_groups_sums_in_java_row_order(pkg, oracle, step):
    """many groups: rows are sorted by group id and one lane per group adds them in row order, page after page -- the
    reference's per-position loop order (AccumulatorCompiler.java:487-566), so DOUBLE sums / averages are BIT-IDENTICAL to the
    Java-order oracle (including NaN / infinities, nulls and masks); counts and BIGINT sums are exact"""
    rng = np.random.default_rng(37)
    n, g = 150_000, 20_000
    pages, allk, allv, alln, allm, alli = [], [], [], [], [], []
    for _ in range(3):
        keys = rng.integers(0, g, n).astype(np.int64)
        vals = rng.standard_normal(n) * 10.0 ** rng.integers(-8, 9, n)
        vals[rng.integers(0, n, 5)] = np.inf
        vals[rng.integers(0, n, 3)] = np.nan
        nulls = (rng.random(n) < 0.1).astype(np.uint8)
        mask = rng.integers(0, 2, n).astype(np.uint8)
        ints = rng.integers(-10**15, 10**15, n).astype(np.int64)
        pages.append(pkg.Page(pkg.Block(pkg.BIGINT, keys), pkg.Block(pkg.DOUBLE, vals, nulls), pkg.Block(pkg.BOOLEAN, mask), pkg.Block(pkg.BIGINT, ints)))
        allk.append(keys); allv.append(vals); alln.append(nulls); allm.append(mask); alli.append(ints)
    aggs = [(pkg.SUM_DOUBLE, 1), (pkg.AVG_DOUBLE, 1), (pkg.SUM_DOUBLE, 1, 2), (pkg.COUNT_ALL, -1), (pkg.SUM_BIGINT, 3), (pkg.AVG_BIGINT, 3), (pkg.COUNT_COLUMN, 1)]
    ctx = pkg.Context(0)
    ctx.profile_enable(True)
    if step == "single":
        rows = run_agg(pkg, ctx, pages, [pkg.BIGINT], [0], aggs, expected=g)
        assert "agg_accumulate_ordered" in ctx.profile()
    else:
        # PARTIAL per page (one operator each), FINAL over the partial pages in order
        partials = []
        for pg in pages:
            f = pkg.HashAggregationOperatorFactory(ctx, 0, [pkg.BIGINT], [0], aggs, step=pkg.PARTIAL, expected_groups=g)
            op = f.createOperator()
            partials += pkg.to_pages(op, [pg])
            op.close()
        # intermediate layout: key, then (count, sum) per sum / avg and (count) per count aggregate
        fin, ch = [], 1
        for a in aggs:
            fin.append((a[0], ch))
            ch += 1 if a[0] in (pkg.COUNT_ALL, pkg.COUNT_COLUMN) else 2
        rows = run_agg(pkg, ctx, partials, [pkg.BIGINT], [0], fin, step=pkg.FINAL, expected=g)
        assert "agg_combine_ordered" in ctx.profile()
    ctx.close()
    keys, vals, nulls, mask, ints = (np.concatenate(x) for x in (allk, allv, alln, allm, alli))
    o = oracle.BigintGroupByHash(g)
    if step == "single":
        gids = o.get_group_ids(oracle.Col(pkg.BIGINT, keys))
        ng = o.group_count
        cnt, java = oracle.agg_double_sum(gids, vals, ng, nulls=nulls)
        cnt_m, java_m = oracle.agg_double_sum(gids, vals, ng, nulls=nulls, mask=mask)
    else:
        # the reference's two-level plan: Java-order sum per page and group, then the partial sums added in page order
        gl = [o.get_group_ids(oracle.Col(pkg.BIGINT, k)) for k in allk]
        ng = o.group_count
        cnt, java, cnt_m, java_m = np.zeros(ng, np.int64), np.zeros(ng), np.zeros(ng, np.int64), np.zeros(ng)
        # the FINAL operator assigns group ids in first-seen order of the PARTIAL pages (= first-seen order of the raw pages)
        for gi, v, nl, m in zip(gl, allv, alln, allm):
            c1, s1 = oracle.agg_double_sum(gi, v, ng, nulls=nl)
            c2, s2 = oracle.agg_double_sum(gi, v, ng, nulls=nl, mask=m)
            seen = np.zeros(ng, bool); seen[np.unique(gi)] = True
            with np.errstate(invalid="ignore"):
                java = np.where(seen & (c1 > 0), java + s1, java); cnt += c1
                java_m = np.where(seen & (c2 > 0), java_m + s2, java_m); cnt_m += c2
        gids = np.concatenate(gl)
    assert len(rows) == ng

    def same_bits(got, want):
        got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
        return np.array_equal(got.view(np.int64)[~np.isnan(want)], want.view(np.int64)[~np.isnan(want)]) and np.array_equal(np.isnan(got), np.isnan(want))

    none_nan = lambda xs: [np.nan if x is None else x for x in xs]
    has = cnt > 0
    assert same_bits(np.array(none_nan([r[1] for r in rows]))[has], java[has])
    with np.errstate(invalid="ignore", divide="ignore"):
        assert same_bits(np.array(none_nan([r[2] for r in rows]))[has], (java / cnt)[has])
    hm = cnt_m > 0
    assert same_bits(np.array(none_nan([r[3] for r in rows]))[hm], java_m[hm])
    assert [r[4] for r in rows] == list(np.bincount(gids, minlength=ng))
    isum = np.zeros(ng, dtype=object)
    for gi, v in zip(gids.tolist(), ints.tolist()):
        isum[gi] += v
    assert [r[5] for r in rows] == [int(x) for x in isum]
    assert [r[7] for r in rows] == list(cnt)


@pytest.mark.parametrize("ngroups", [4, 30, 3000])
def test_fused_filter_project_aggregation_matches_unfused_and_oracle(pkg, oracle, monkeypatch, ngroups):
    """FilterAndProject fused into HashAggregation (row mask + in-register projections) == the unfused composition == the
    oracle composition; 4 groups exercise the lane-private LDS accumulators fed by one-byte group ids, 30 groups the one-byte ids
    widened for the row-order (ORDERED) accumulation, 3000 groups int32 ids + ORDERED"""
    rng = np.random.default_rng(29)
    n = 120_000
    T = [pkg.VARCHAR, pkg.BIGINT, pkg.DOUBLE, pkg.DOUBLE, pkg.DOUBLE, pkg.DATE, pkg.BOOLEAN, pkg.BIGINT]
    nf = 0.0 if ngroups == 4 else 0.02
    keys1 = rand_block(pkg, rng, pkg.VARCHAR, n, nf, (0, {4: 2, 30: 6}.get(ngroups, 60)))
    keys2 = rand_block(pkg, rng, pkg.BIGINT, n, nf, (0, {4: 2, 30: 4}.get(ngroups, 50)))
    page = pkg.Page(keys1, keys2, rand_block(pkg, rng, pkg.DOUBLE, n, 0.05), rand_block(pkg, rng, pkg.DOUBLE, n, 0.0, (0, 11)),
                    rand_block(pkg, rng, pkg.DOUBLE, n, 0.03, (0, 9)), rand_block(pkg, rng, pkg.DATE, n, 0.01, (9000, 9400)),
                    rand_block(pkg, rng, pkg.BOOLEAN, n, 0.1), rand_block(pkg, rng, pkg.BIGINT, n, 0.05, (-10**9, 10**9)))
    f, c = pkg.field, pkg.constant
    one = c(1.0, pkg.DOUBLE)
    filt = f(5, pkg.DATE) <= 9300
    projs = [f(0, pkg.VARCHAR), f(1, pkg.BIGINT), f(2, pkg.DOUBLE), f(2, pkg.DOUBLE) * (one - f(3, pkg.DOUBLE) / c(100.0, pkg.DOUBLE)),
             f(2, pkg.DOUBLE) * (one - f(3, pkg.DOUBLE) / c(100.0, pkg.DOUBLE)) * (one + f(4, pkg.DOUBLE) / c(100.0, pkg.DOUBLE)), f(6, pkg.BOOLEAN), f(7, pkg.BIGINT)]
    aggs = [(pkg.SUM_DOUBLE, 2), (pkg.SUM_DOUBLE, 3), (pkg.AVG_DOUBLE, 4), (pkg.COUNT_ALL, -1), (pkg.SUM_DOUBLE, 3, 5), (pkg.SUM_BIGINT, 6), (pkg.AVG_BIGINT, 6),
            (pkg.COUNT_COLUMN, 2), (pkg.COUNT_ALL, -1, 5)]
    results = {}
    for mode in ("fused", "unfused"):
        if mode == "unfused":
            monkeypatch.setenv("TGPU_DISABLE_FUSION", "1")
        ctx = pkg.Context(0)
        ctx.profile_enable(True)
        fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 0, T, filt, projs, [pkg.VARCHAR, pkg.BIGINT], [0, 1], aggs)
        out = pkg.to_pages(fac.createOperator(), [page, pkg.Page(*[pkg.Block(t, []) for t in T]), page])
        results[mode] = [r for p in out for r in p.rows()]
        prof = ctx.profile()
        # (ORDERED with >= 256 rows per group on average: the chained kernel, fa_ordered_chain; below that one lane per group)
        assert ("fused_project_accumulate_lowcard" in prof or "fused_project_accumulate_ordered_chain" in prof or "fused_project_accumulate_ordered" in prof) == (mode == "fused")
        if mode == "fused":
            assert ("fused_project_accumulate_lowcard" in prof) == (ngroups == 4)
        else:
            assert ("agg_accumulate_ordered" in prof or "agg_accumulate_ordered_chain" in prof) == (ngroups != 4)
        ctx.close()
    a, b = results["fused"], results["unfused"]
    assert len(a) == len(b) and [r[:2] for r in a] == [r[:2] for r in b]        # same groups in the same (first-seen) order
    for ra, rb in zip(a, b):
        assert ra[5] == rb[5] and ra[7] == rb[7] and ra[9] == rb[9] and ra[10] == rb[10]
        fa_ = [np.nan if x is None else x for x in (ra[2], ra[3], ra[4], ra[6], ra[8])]
        fb_ = [np.nan if x is None else x for x in (rb[2], rb[3], rb[4], rb[6], rb[8])]
        # few groups: both feed the exact accumulators; many groups: both sum in row order (ORDERED mode) -- identical bits either way
        assert ulp_diff(fa_, fb_).max() == 0
    # oracle composition: filter -> projections -> MultiChannelGroupByHash -> exact sums
    prog = pkg.expressions.FlatProgram(filt, projs)
    cols = [ocol(oracle, blk) for blk in page.blocks]
    pos = oracle.filter_positions(prog.nodes, prog.filter_root, b"", cols)
    kb = [pkg.Block(pkg.VARCHAR, [keys1.get(int(i)) for i in pos]), pkg.Block(pkg.BIGINT, [keys2.get(int(i)) for i in pos])]
    og = oracle.MultiChannelGroupByHash([pkg.VARCHAR, pkg.BIGINT], 100)
    gids = og.get_group_ids([ocol(oracle, x) for x in kb])
    ng = og.group_count
    assert ng == len(a)
    v3, n3 = oracle.project(prog.nodes, prog.projection_roots[3], b"", cols, pos)
    got3 = np.array([np.nan if r[3] is None else r[3] for r in a])
    if ngroups == 4:
        cnt3, sum3 = oracle.agg_double_sum_exact(gids, v3, ng, nulls=n3)
        want3 = np.where(cnt3 > 0, 2 * sum3, np.nan)   # the page was fed twice: exact doubling
    else:
        # many groups: row-order sums, the page's rows twice in a row (Java-order oracle, bit-identical)
        cnt3, sum3 = oracle.agg_double_sum(np.concatenate([gids, gids]), np.concatenate([v3, v3]), ng, nulls=np.concatenate([n3, n3]))
        want3 = np.where(cnt3 > 0, sum3, np.nan)
    assert ulp_diff(got3, want3).max() == 0
    assert [r[5] for r in a] == list(2 * oracle.agg_count(gids, len(pos), ng))


def test_group_by_hash_optimistic_sub_batch_overflow_retry(pkg, oracle, monkeypatch):
    """after a sub-batch without new groups the next one is 64x larger; if that one floods the table with new keys the probe
    kernel flags the overflow, the table is rebuilt larger and the rows are re-run -- ids must still equal the Java order"""
    monkeypatch.setenv("TGPU_GBH_SUBBATCH", "1000")
    c = pkg.Context(0)
    rng = np.random.default_rng(31)
    vals = np.concatenate([np.zeros(2000, dtype=np.int64), rng.permutation(200_000)[:60_000].astype(np.int64) + 1, np.zeros(500, dtype=np.int64)])
    blk = pkg.Block(pkg.BIGINT, vals)
    gbh = pkg.GroupByHash(c, [pkg.BIGINT], [0], expected_size=10)
    o = oracle.BigintGroupByHash(10)
    assert np.array_equal(gbh.getGroupIds(pkg.Page(blk)), o.get_group_ids(ocol(oracle, blk)))
    assert gbh.getGroupCount() == o.group_count == 60_001
    gbh.close()
    c.close()


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("shape", ["few_then_many", "many_from_the_start", "few"])
def test_double_sums_do_not_depend_on_how_the_rows_are_cut_into_pages(pkg, shape, fused):
    """the accumulators' DOUBLE mode (exact limbs vs row-order sums) is decided from the stream's first 65 536 rows, not from its first page:
    one input cut into pages three different ways gives the same bits -- a stream that starts with a handful of groups and
    grows to thousands, one with thousands from the start, one that stays small; fused and unfused operators"""
    rng = np.random.default_rng({"few_then_many": 3, "many_from_the_start": 5, "few": 7}[shape])
    n = 200_000
    if shape == "few_then_many":
        keys = np.concatenate([rng.integers(0, 3, 70_000), rng.integers(0, 5000, n - 70_000)])
    elif shape == "many_from_the_start":
        keys = rng.integers(0, 5000, n)
    else:
        keys = rng.integers(0, 4, n)
    keys = keys.astype(np.int64)
    vals = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 7, n)
    nulls = (rng.random(n) < 0.02).astype(np.uint8)
    B, D = pkg.BIGINT, pkg.DOUBLE
    f = pkg.field
    cuts = {"one_page": [n], "small_first_page": [10, 1000, 50_000, 64_000, n], "ragged": [7_777, 65_536, 65_537, 131_072, n], "tiny_pages": list(range(4_000, n, 4_000)) + [n]}
    results = {}
    for name, ends in cuts.items():
        ctx = pkg.Context(0)
        pages, a = [], 0
        for z in ends:
            pages.append(pkg.Page(pkg.Block(B, keys[a:z]), pkg.Block(D, vals[a:z], nulls[a:z])))
            a = z
        aggs = [(pkg.SUM_DOUBLE, 1), (pkg.AVG_DOUBLE, 1), (pkg.COUNT_COLUMN, 1)]
        if fused:
            fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 0, [B, D], None, [f(0, B), f(1, D)], [B], [0], aggs)
        else:
            fac = pkg.HashAggregationOperatorFactory(ctx, 0, [B], [0], aggs, expected_groups=100)
        out = pkg.to_pages(fac.createOperator(), pages)
        results[name] = [r for p in out for r in p.rows()]
        ctx.close()
    base = results["one_page"]
    assert len(base) == len(np.unique(keys))
    for name, rows in results.items():
        assert [r[0] for r in rows] == [r[0] for r in base], name
        a_ = np.array([[np.nan if x is None else x for x in r[1:3]] for r in rows])
        b_ = np.array([[np.nan if x is None else x for x in r[1:3]] for r in base])
        assert ulp_diff(a_.ravel(), b_.ravel()).max() == 0, name
        assert [r[3] for r in rows] == [r[3] for r in base], name


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("groups,n", [(1, 5_000), (4, 70_000), (300, 90_000), (4096, 1_300_000), (100_000, 400_000)])
def test_aggregation_java_order_chained_few_groups(pkg, oracle, groups, n, fused, monkeypatch):
    """SUM_ORDER_JAVA with few groups: one workgroup per group, the DOUBLE sums as chains fed from LDS (fused operator: the generated
    fa_ordered_chain; plain HashAggregationOperator over the already filtered and projected rows: agg_ordered_chain_kernel) -- bit-identical to the
    Java-order oracle (DoubleSumAggregation.java:34-38, per-position loop AccumulatorCompiler.java:487-566) and to the lane-per-group kernel,
    over several pages, with filtered rows, nulls, masks, NaN / infinities, skewed group sizes (groups without rows in a page, groups of one
    row, tiles that end inside the 960-row stretch); counts and BIGINT sums exact.  The last case has too many groups for a workgroup each:
    one lane per group, and the lane of the one heavy group (7 % of the rows) hands it over to a chain after 4 096 rows"""
    rng = np.random.default_rng(101 + groups)
    B, D, BO = pkg.BIGINT, pkg.DOUBLE, pkg.BOOLEAN
    f, c = pkg.field, pkg.constant
    T = [B, D, BO, B, D]
    filt = f(4, D) < 0.9
    projs = [f(0, B), f(1, D), f(2, BO), f(3, B), f(1, D) * (c(1.0, D) - f(4, D))]
    aggs = [(pkg.SUM_DOUBLE, 1), (pkg.AVG_DOUBLE, 1), (pkg.SUM_DOUBLE, 1, 2), (pkg.COUNT_ALL, -1), (pkg.SUM_BIGINT, 3), (pkg.AVG_BIGINT, 3), (pkg.SUM_DOUBLE, 4), (pkg.COUNT_COLUMN, 1)]
    cols = []
    for page in range(3):
        m = n if page != 1 else n // 3 + 449
        keys = np.minimum((rng.pareto(1.2, m) * max(1, groups // 8)).astype(np.int64), groups - 1)      # skewed: some groups huge, some of one row
        if page == 0:
            keys[:groups] = np.arange(groups)            # (every group known after the first page: the ids of the oracle and the table agree)
        vals = rng.standard_normal(m) * 10.0 ** rng.integers(-8, 9, m)
        vals[rng.integers(0, m, 4)] = np.inf
        vals[rng.integers(0, m, 2)] = np.nan
        vals[rng.integers(0, m, 4)] = -0.0
        cols.append((keys, vals, (rng.random(m) < 0.1).astype(np.uint8), rng.integers(0, 2, m).astype(np.uint8), rng.integers(-10**15, 10**15, m).astype(np.int64), rng.random(m)))

    def run(disable):
        if disable:
            monkeypatch.setenv("TGPU_DISABLE_ORDERED_CHAIN", "1")
        else:
            monkeypatch.delenv("TGPU_DISABLE_ORDERED_CHAIN", raising=False)
        ctx = pkg.Context(0)
        ctx.profile_enable(True)
        ctx.set_double_sum_order(pkg.SUM_ORDER_JAVA)
        if fused:
            pages = [pkg.Page(pkg.Block(B, k), pkg.Block(D, v, nl), pkg.Block(BO, m_), pkg.Block(B, i_), pkg.Block(D, d_)) for k, v, nl, m_, i_, d_ in cols]
            fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 0, T, filt, projs, [B], [0], aggs)
        else:
            pages = []
            for k, v, nl, m_, i_, d_ in cols:
                keep = d_ < 0.9
                with np.errstate(invalid="ignore", over="ignore"):
                    prod = v * (1.0 - d_)
                pages.append(pkg.Page(pkg.Block(B, k[keep]), pkg.Block(D, v[keep], nl[keep]), pkg.Block(BO, m_[keep]), pkg.Block(B, i_[keep]), pkg.Block(D, prod[keep], nl[keep])))
            fac = pkg.HashAggregationOperatorFactory(ctx, 0, [B], [0], aggs, expected_groups=groups)
        out = pkg.to_pages(fac.createOperator(), pages)
        prof = ctx.profile()
        ctx.close()
        return [r for p_ in out for r in p_.rows()], prof

    names = ("fused_project_accumulate_ordered_chain", "fused_project_accumulate_ordered") if fused else ("agg_accumulate_ordered_chain", "agg_accumulate_ordered")
    rows, prof = run(False)
    if groups <= 65536:
        assert names[0] in prof
    else:
        assert names[0] not in prof and names[1] in prof and (not fused or "fused_project_accumulate_ordered_handoff" in prof)
    rows_lane, prof_lane = run(True)
    assert "fused_project_accumulate_ordered_handoff" not in prof_lane
    assert names[0] not in prof_lane and names[1] in prof_lane

    def bits(rs):
        return [[None if x is None else (np.float64(x).view(np.int64).item() if isinstance(x, float) else x) for x in r] for r in rs]
    assert bits(rows) == bits(rows_lane)
    keys, vals, nulls, mask, ints, disc = (np.concatenate([cl[i] for cl in cols]) for i in range(6))
    sel = np.nonzero(disc < 0.9)[0]
    keys, vals, nulls, mask, ints, disc = (a[sel] for a in (keys, vals, nulls, mask, ints, disc))
    o = oracle.BigintGroupByHash(groups)
    gids = o.get_group_ids(oracle.Col(pkg.BIGINT, keys))
    ng = o.group_count
    assert len(rows) == ng
    by_key = {r[0]: r for r in rows}
    first = {}
    for k_, g_ in zip(keys.tolist(), gids.tolist()):
        first.setdefault(g_, k_)
    got = [by_key[first[g_]] for g_ in range(ng)]
    with np.errstate(invalid="ignore", over="ignore"):
        cnt, java = oracle.agg_double_sum(gids, vals, ng, nulls=nulls)
        cnt_m, java_m = oracle.agg_double_sum(gids, vals, ng, nulls=nulls, mask=mask)
        cnt_d, java_d = oracle.agg_double_sum(gids, vals * (1.0 - disc), ng, nulls=nulls)
        cnt_i, java_i = oracle.agg_long_avg(gids, ints, ng)
        cnt_l, exact = oracle.agg_long_sum(gids, ints, ng)

    def same(col, want, have_rows):
        for g_ in range(ng):
            x = got[g_][col]
            if not have_rows[g_]:
                assert x is None, (col, g_)
            else:
                w = want[g_]
                assert (np.isnan(x) and np.isnan(w)) or np.float64(x).view(np.int64) == np.float64(w).view(np.int64), (col, g_, x, w)
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        same(1, java, cnt > 0)
        same(2, java / np.maximum(cnt, 1), cnt > 0)
        same(3, java_m, cnt_m > 0)
        same(6, java_i / np.maximum(cnt_i, 1), cnt_i > 0)
        same(7, java_d, cnt_d > 0)
    rows_of = np.bincount(gids, minlength=ng)
    for g_ in range(ng):
        assert got[g_][4] == rows_of[g_] and got[g_][5] == exact[g_], g_
        assert got[g_][8] == cnt[g_], g_


@pytest.mark.parametrize("plan", ["single", "partial_final", "fused", "fused_java_order", "spilled"])
@pytest.mark.parametrize("groups", [3, 300, 50_000])
def test_min_max_bigint_every_path(pkg, oracle, groups, plan):
    """min(bigint) / max(bigint) (AbstractMinMaxAggregationFunction.java:233-289) next to sums and counts, through every accumulation path:
    few groups (lane-private LDS counts, the extremes straight to the state word; the fused operator's one-pass launches from the fourth
    page on), a few hundred and 50 000 groups (row-order modes: chained kernel / one lane per group), PARTIAL -> FINAL (the (count, value)
    pair as intermediate state), the fused filter/project operator in both DOUBLE orders, and spilled runs merged at the end.  Nulls,
    masks, groups whose rows are all null (-> null), int64 extremes; bit-exact against the oracle's row-at-a-time restatement"""
    rng = np.random.default_rng(500 + groups)
    B, D, BO = pkg.BIGINT, pkg.DOUBLE, pkg.BOOLEAN
    n, npages = 40_000, 6
    cols = []
    for page in range(npages):
        keys = rng.integers(0, groups, n).astype(np.int64)
        vals = rng.integers(-2**62, 2**62, n).astype(np.int64)
        vals[rng.integers(0, n, 3)] = -(2**63)
        vals[rng.integers(0, n, 3)] = 2**63 - 1
        small = rng.integers(-1000, 1000, n).astype(np.int64)
        nulls = (rng.random(n) < 0.2).astype(np.uint8)
        if groups > 3:
            nulls[keys == 1] = 1                     # a group whose values are all null
        mask = rng.integers(0, 2, n).astype(np.uint8)
        dbl = rng.standard_normal(n)
        cols.append((keys, vals, nulls, mask, small, dbl))
    aggs = [(pkg.MIN_BIGINT, 1), (pkg.MAX_BIGINT, 1), (pkg.MIN_BIGINT, 1, 2), (pkg.SUM_BIGINT, 3), (pkg.MAX_BIGINT, 3), (pkg.COUNT_ALL, -1), (pkg.SUM_DOUBLE, 4), (pkg.COUNT_COLUMN, 1)]
    pages = [pkg.Page(pkg.Block(B, k), pkg.Block(B, v, nl), pkg.Block(BO, m), pkg.Block(B, sm), pkg.Block(D, d)) for k, v, nl, m, sm, d in cols]
    ctx = pkg.Context(0)
    ctx.profile_enable(True)
    f = pkg.field
    if plan == "single":
        rows = run_agg(pkg, ctx, pages, [B], [0], aggs, expected=groups)
    elif plan == "partial_final":
        partials = []
        for pg in pages:
            op = pkg.HashAggregationOperatorFactory(ctx, 0, [B], [0], aggs, step=pkg.PARTIAL, expected_groups=groups).createOperator()
            partials += pkg.to_pages(op, [pg])
            op.close()
        fin, ch = [], 1
        for a in aggs:
            fin.append((a[0], ch))
            ch += 1 if a[0] in (pkg.COUNT_ALL, pkg.COUNT_COLUMN) else 2
        rows = run_agg(pkg, ctx, partials, [B], [0], fin, step=pkg.FINAL, expected=groups)
    elif plan in ("fused", "fused_java_order"):
        if plan == "fused_java_order":
            ctx.set_double_sum_order(pkg.SUM_ORDER_JAVA)
        fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 0, [B, B, BO, B, D], None, [f(0, B), f(1, B), f(2, BO), f(3, B), f(4, D)], [B], [0], aggs)
        rows = [r for p_ in pkg.to_pages(fac.createOperator(), pages) for r in p_.rows()]
        if groups == 3 and plan == "fused":
            assert "fused_filter_group_accumulate_onepass" in ctx.profile()
    else:
        op = pkg.HashAggregationOperatorFactory(ctx, 0, [B], [0], aggs, expected_groups=groups, spill_enabled=True).createOperator()
        rows = _drive_with_revokes(op, pages, True)
        assert op.spillStats()[0] >= 1
        op.close()
    ctx.close()
    keys, vals, nulls, mask, small, dbl = (np.concatenate([c[i] for c in cols]) for i in range(6))
    o = oracle.BigintGroupByHash(groups)
    gids = o.get_group_ids(oracle.Col(B, keys))
    ng = o.group_count
    assert len(rows) == ng
    by_key = {r[0]: r for r in rows}
    first = {}
    for k_, g_ in zip(keys.tolist(), gids.tolist()):
        first.setdefault(g_, k_)
    got = [by_key[first[g_]] for g_ in range(ng)]
    c_min, mins = oracle.agg_long_minmax(gids, vals, ng, True, nulls=nulls)
    c_max, maxs = oracle.agg_long_minmax(gids, vals, ng, False, nulls=nulls)
    c_mm, mmin = oracle.agg_long_minmax(gids, vals, ng, True, nulls=nulls, mask=mask)
    c_s, sums = oracle.agg_long_sum(gids, small, ng)
    c_sm, smax = oracle.agg_long_minmax(gids, small, ng, False)
    for g_ in range(ng):
        r = got[g_]
        assert r[1] == (int(mins[g_]) if c_min[g_] else None) and r[2] == (int(maxs[g_]) if c_max[g_] else None), (g_, r)
        assert r[3] == (int(mmin[g_]) if c_mm[g_] else None), (g_, r)
        assert r[4] == int(sums[g_]) and r[5] == int(smax[g_]) and r[6] == int(c_s[g_]) and r[8] == int(c_min[g_]), (g_, r)
    if groups > 3:
        assert by_key[1][1] is None and by_key[1][2] is None and by_key[1][8] == 0


@pytest.mark.parametrize("plan", ["single", "partial_final", "fused", "spilled"])
@pytest.mark.parametrize("groups", [3, 300, 50_000])
def test_min_max_double_every_path(pkg, oracle, groups, plan):
    """min(double) / max(double) (AbstractMinMaxAggregationFunction.java:227-230,291-306; Double.compare for min, MinMaxCompare.maxDouble for
    max) through the accumulation paths, bit for bit the oracle's row-at-a-time restatement: infinities, NaN (above everything for min, below
    everything for max: a group of NaNs only answers NaN), -0.0, nulls, masks, all-null groups.  Inputs hold -0.0 but no +0.0: see the
    deviation pinned in test_max_double_of_zeros_of_both_signs"""
    rng = np.random.default_rng(700 + groups)
    B, D, BO = pkg.BIGINT, pkg.DOUBLE, pkg.BOOLEAN
    n, npages = 40_000, 6
    cols = []
    for page in range(npages):
        keys = rng.integers(0, groups, n).astype(np.int64)
        vals = rng.standard_normal(n) * 10.0 ** rng.integers(-300, 300, n)
        for special in (np.inf, -np.inf, np.nan, -0.0):
            vals[rng.integers(0, n, 40)] = special
        if groups > 3:
            vals[keys == 2] = np.nan                 # a group of NaNs only
        nulls = (rng.random(n) < 0.2).astype(np.uint8)
        if groups > 3:
            nulls[keys == 1] = 1                     # a group whose values are all null
        mask = rng.integers(0, 2, n).astype(np.uint8)
        cols.append((keys, vals, nulls, mask))
    aggs = [(pkg.MIN_DOUBLE, 1), (pkg.MAX_DOUBLE, 1), (pkg.MAX_DOUBLE, 1, 2), (pkg.COUNT_COLUMN, 1), (pkg.SUM_DOUBLE, 1, 2)]
    pages = [pkg.Page(pkg.Block(B, k), pkg.Block(D, v, nl), pkg.Block(BO, m)) for k, v, nl, m in cols]
    ctx = pkg.Context(0)
    f = pkg.field
    if plan == "single":
        rows = run_agg(pkg, ctx, pages, [B], [0], aggs[:4], expected=groups)
    elif plan == "partial_final":
        partials = []
        for pg in pages:
            op = pkg.HashAggregationOperatorFactory(ctx, 0, [B], [0], aggs[:4], step=pkg.PARTIAL, expected_groups=groups).createOperator()
            partials += pkg.to_pages(op, [pg])
            op.close()
        fin, ch = [], 1
        for a in aggs[:4]:
            fin.append((a[0], ch))
            ch += 1 if a[0] in (pkg.COUNT_ALL, pkg.COUNT_COLUMN) else 2
        rows = run_agg(pkg, ctx, partials, [B], [0], fin, step=pkg.FINAL, expected=groups)
    elif plan == "fused":
        fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 0, [B, D, BO], None, [f(0, B), f(1, D), f(2, BO)], [B], [0], aggs[:4])
        rows = [r for p_ in pkg.to_pages(fac.createOperator(), pages) for r in p_.rows()]
    else:
        op = pkg.HashAggregationOperatorFactory(ctx, 0, [B], [0], aggs[:4], expected_groups=groups, spill_enabled=True).createOperator()
        rows = _drive_with_revokes(op, pages, True)
        op.close()
    ctx.close()
    keys, vals, nulls, mask = (np.concatenate([c[i] for c in cols]) for i in range(4))
    o = oracle.BigintGroupByHash(groups)
    gids = o.get_group_ids(oracle.Col(B, keys))
    ng = o.group_count
    assert len(rows) == ng
    by_key = {r[0]: r for r in rows}
    first = {}
    for k_, g_ in zip(keys.tolist(), gids.tolist()):
        first.setdefault(g_, k_)
    c_min, mins = oracle.agg_double_minmax(gids, vals, ng, True, nulls=nulls)
    c_max, maxs = oracle.agg_double_minmax(gids, vals, ng, False, nulls=nulls)
    c_mm, mmax = oracle.agg_double_minmax(gids, vals, ng, False, nulls=nulls, mask=mask)
    canon = lambda x: None if x is None else (0x7ff8000000000000 if x != x else int(np.float64(x).view(np.int64)))
    for g_ in range(ng):
        r = by_key[first[g_]]
        want = [mins[g_] if c_min[g_] else None, maxs[g_] if c_max[g_] else None, mmax[g_] if c_mm[g_] else None]
        assert [canon(x) for x in r[1:4]] == [canon(x) for x in want] and r[4] == c_min[g_], (g_, r, want)
    if groups > 3:
        assert by_key[1][1] is None and by_key[2][1] != by_key[2][1] and by_key[2][2] != by_key[2][2]


def test_max_double_of_zeros_of_both_signs(pkg, ctx, oracle):
    """the one place where max(double) is NOT the reference's bits, pinned knowingly (tgpu.h TGPU_AGG_MAX_DOUBLE): MinMaxCompare.maxDouble
    keeps the zero that came first (-0.0 > +0.0 is false both ways), an order-independent maximum has to choose: +0.0.  Equal under =.
    min is exact: Double.compare puts -0.0 below +0.0"""
    page = pkg.Page(pkg.Block(pkg.BIGINT, np.zeros(4, dtype=np.int64)), pkg.Block(pkg.DOUBLE, np.array([-0.0, 0.0, -0.0, -1.0])))
    ((_, mn, mx),) = run_agg(pkg, ctx, [page], [pkg.BIGINT], [0], [(pkg.MIN_DOUBLE, 1), (pkg.MAX_DOUBLE, 1)])
    _, java_max = oracle.agg_double_minmax(np.zeros(4, dtype=np.int64), page.getBlock(1).values, 1, False)
    assert mn == -1.0 and mx == 0.0 and not np.signbit(mx) and np.signbit(java_max[0]) and java_max[0] == mx


def _onepass_pages(pkg, rng, npages, rows, late_groups, error_page=None):
    """pages of a Q1-like program: 2 varchar(1) keys (3 x 2 values), some pages add a new key value late in the stream"""
    pages = []
    for i in range(npages):
        k1 = rng.integers(0, 3, rows)
        k2 = rng.integers(0, 2, rows)
        a = [("A", "N", "R")[x] for x in k1]
        b = [("F", "O")[x] for x in k2]
        if i in late_groups:               # a few rows of a group nobody has seen, in the middle of the page
            for r in (rows // 2, rows // 2 + 7):
                a[r] = late_groups[i]
        qty = rng.integers(1, 51, rows).astype(np.float64)
        price = qty * rng.integers(90000, 210000, rows) / 100.0
        disc = rng.integers(0, 11, rows) / 100.0
        ship = rng.integers(8036, 10562, rows).astype(np.int32)
        div = np.ones(rows, dtype=np.int64)
        if error_page == i:
            div[rows // 3] = 0
        pages.append(pkg.Page(pkg.Block(pkg.VARCHAR, a), pkg.Block(pkg.VARCHAR, b), pkg.Block(pkg.DOUBLE, qty), pkg.Block(pkg.DOUBLE, price),
                              pkg.Block(pkg.DOUBLE, disc, (rng.random(rows) < 0.02).astype(np.uint8)), pkg.Block(pkg.DATE, ship), pkg.Block(pkg.BIGINT, div)))
    return pages


def _onepass_program(pkg, with_division=False):
    f, c = pkg.field, pkg.constant
    V, D, DT, B = pkg.VARCHAR, pkg.DOUBLE, pkg.DATE, pkg.BIGINT
    T = [V, V, D, D, D, DT, B]
    filt = f(5, DT) <= 10471
    if with_division:                      # a filter that can raise: 10 / div > 0
        filt = pkg.and_(filt, (c(10, B) / f(6, B)) > 0)
    projs = [f(0, V), f(1, V), f(2, D), f(3, D), f(3, D) * (c(1.0, D) - f(4, D))]
    aggs = [(pkg.SUM_DOUBLE, 2), (pkg.SUM_DOUBLE, 3), (pkg.SUM_DOUBLE, 4), (pkg.AVG_DOUBLE, 4), (pkg.COUNT_ALL, -1), (pkg.COUNT_COLUMN, 4)]
    return T, filt, projs, aggs


@pytest.mark.parametrize("batch_rows", ["1", "20000", None])
@pytest.mark.parametrize("late", [{}, {6: "X"}, {5: "X", 6: "Y", 9: "Z"}, {4: "X", 5: "X"}])
def test_fused_aggregation_one_launch_per_page_equals_the_two_launch_path(pkg, monkeypatch, late, batch_rows):
    """once the group set has settled the fused aggregation runs ONE launch per page -- or per batch of small pages: by default pages wait until
    2^24 rows have come together (here: all of them, launched by finish()); TGPU_ONEPASS_BATCH_ROWS=20000 makes launches of three pages, =1 one
    per page -- and reads a launch's counters a call later (FusedAggGpu::onepass): same rows, bit for bit, as the probe + accumulate path -- also
    when pages in the middle of the stream bring new groups (their totals are dropped on the device, the pages re-run through the insert
    protocol, ids in first-seen order)"""
    monkeypatch.setenv("TGPU_MODE_PREFIX_ROWS", "5000")   # (the DOUBLE mode is decided within the first page: the stream settles early)
    if batch_rows:
        monkeypatch.setenv("TGPU_ONEPASS_BATCH_ROWS", batch_rows)
    rng = np.random.default_rng(41)
    pages = _onepass_pages(pkg, rng, 12, 9000, late)
    T, filt, projs, aggs = _onepass_program(pkg)
    results = {}
    for mode in ("onepass", "two_launch"):
        if mode == "two_launch":
            monkeypatch.setenv("TGPU_DISABLE_ONEPASS", "1")
        ctx = pkg.Context(0)
        ctx.profile_enable(True)
        fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 0, T, filt, projs, [pkg.VARCHAR, pkg.VARCHAR], [0, 1], aggs)
        out = pkg.to_pages(fac.createOperator(), pages)
        results[mode] = [r for p in out for r in p.rows()]
        prof = ctx.profile()
        launches = prof.get("fused_filter_group_accumulate_onepass", {"count": 0})["count"]
        if mode == "onepass" and batch_rows == "1":
            assert launches >= 12 - 3 - 2 * len(late)      # host pages are library-owned copies: the one-pass path needs no promise for them
        elif mode == "onepass":
            assert 1 <= launches <= (4 if batch_rows else 2)
        else:
            assert launches == 0
        ctx.close()
    a, b = results["onepass"], results["two_launch"]
    assert [r[:2] for r in a] == [r[:2] for r in b] and len(a) >= 6 + len(set(late.values()))
    for ra, rb in zip(a, b):
        assert ra[6] == rb[6] and ra[7] == rb[7]
        assert ulp_diff([np.nan if x is None else x for x in ra[2:6]], [np.nan if x is None else x for x in rb[2:6]]).max() == 0


@pytest.mark.parametrize("late", [{}, {5: "X", 6: "Y", 9: "Z"}])
def test_fused_one_pass_launches_with_min_and_max(pkg, monkeypatch, late):
    """min(bigint) / max(bigint) in the one-launch-per-page protocol: their rows go straight to the state word (no pending copy), so a page
    that brings a new group -- its totals are dropped, the page is run again -- applies the rows of its known groups twice: idempotent.
    Expected values: min / max per group over the selected, non-null rows; next to them a sum and a count, which must NOT see a row twice"""
    monkeypatch.setenv("TGPU_MODE_PREFIX_ROWS", "5000")
    monkeypatch.setenv("TGPU_ONEPASS_BATCH_ROWS", "1")
    rng = np.random.default_rng(43)
    V, DT, B = pkg.VARCHAR, pkg.DATE, pkg.BIGINT
    rows_per_page, pages, host = 9000, [], []
    for i in range(12):
        a = [("A", "N", "R")[x] for x in rng.integers(0, 3, rows_per_page)]
        b = [("F", "O")[x] for x in rng.integers(0, 2, rows_per_page)]
        if i in late:                       # a few rows of a group nobody has seen, in the middle of the page
            for r in (rows_per_page // 2, rows_per_page // 2 + 7):
                a[r] = late[i]
        val = rng.integers(-10**12, 10**12, rows_per_page).astype(np.int64)
        nulls = (rng.random(rows_per_page) < 0.05).astype(np.uint8)
        small = rng.integers(0, 100, rows_per_page).astype(np.int64)
        ship = rng.integers(8036, 10562, rows_per_page).astype(np.int32)
        pages.append(pkg.Page(pkg.Block(V, a), pkg.Block(V, b), pkg.Block(B, val, nulls), pkg.Block(B, small), pkg.Block(DT, ship)))
        host.append((a, b, val, nulls, small, ship))
    f = pkg.field
    T = [V, V, B, B, DT]
    aggs = [(pkg.MIN_BIGINT, 2), (pkg.MAX_BIGINT, 2), (pkg.MAX_BIGINT, 3), (pkg.SUM_BIGINT, 3), (pkg.COUNT_ALL, -1), (pkg.COUNT_COLUMN, 2)]
    ctx = pkg.Context(0)
    ctx.profile_enable(True)
    fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 0, T, f(4, DT) <= 10471, [f(0, V), f(1, V), f(2, B), f(3, B)], [V, V], [0, 1], aggs)
    rows = [r for p_ in pkg.to_pages(fac.createOperator(), pages) for r in p_.rows()]
    assert ctx.profile().get("fused_filter_group_accumulate_onepass", {"count": 0})["count"] >= 12 - 3 - 2 * len(late)
    ctx.close()
    want = {}
    for a, b, val, nulls, small, ship in host:
        for i in np.nonzero(ship <= 10471)[0]:
            w = want.setdefault((a[i], b[i]), [None, None, int(small[i]), 0, 0, 0])
            if not nulls[i]:
                v = int(val[i])
                w[0], w[1], w[5] = (v if w[0] is None else min(w[0], v)), (v if w[1] is None else max(w[1], v)), w[5] + 1
            w[2], w[3], w[4] = max(w[2], int(small[i])), w[3] + int(small[i]), w[4] + 1
    assert len(rows) == len(want) >= 6 + len(set(late.values()))
    for r in rows:
        assert list(r[2:8]) == want[(r[0], r[1])], r


def test_fused_aggregation_launch_over_ragged_pages_with_and_without_null_vectors(pkg, monkeypatch):
    """one launch over a list of pages (fq_onepass_multi): page sizes around the 2048-row tile (1, 2047, 2048, 2049 rows ...), pages whose
    columns carry null vectors next to pages that carry none, launches of 1 to 64 pages -- bit for bit the two-launch path"""
    monkeypatch.setenv("TGPU_MODE_PREFIX_ROWS", "3000")
    rng = np.random.default_rng(53)
    T, filt, projs, aggs = _onepass_program(pkg)
    sizes = [4000, 4000, 4000, 4000, 1, 2047, 2048, 2049, 5, 4096, 10_000, 3, 6143, 6145, 70_000, 2, 2048] + [17] * 70 + [30_000]
    pages = []
    for i, n in enumerate(sizes):
        pg = _onepass_pages(pkg, rng, 1, n, {})[0]
        if i % 3 == 1:      # no null vector anywhere in this page
            blocks = [pkg.Block(b.type, b.values, None, b.offsets) for b in pg.blocks]
            pg = pkg.Page(*blocks)
        pages.append(pg)
    results = {}
    for mode in ("multi", "two_launch"):
        if mode == "two_launch":
            monkeypatch.setenv("TGPU_DISABLE_ONEPASS", "1")
        else:
            monkeypatch.setenv("TGPU_ONEPASS_BATCH_ROWS", "9000")
        ctx = pkg.Context(0)
        ctx.profile_enable(True)
        fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 0, T, filt, projs, [pkg.VARCHAR, pkg.VARCHAR], [0, 1], aggs)
        out = pkg.to_pages(fac.createOperator(), pages)
        results[mode] = [r for p in out for r in p.rows()]
        launches = ctx.profile().get("fused_filter_group_accumulate_onepass", {"count": 0})["count"]
        assert launches == (6 if mode == "multi" else 0)   # 5 + 3 + 3 + 1 + 64 + 9 pages (9000 rows or 64 pages make a launch)
        ctx.close()
    a, b = results["multi"], results["two_launch"]
    assert [r[:2] for r in a] == [r[:2] for r in b] and len(a) == 6
    for ra, rb in zip(a, b):
        assert ra[6] == rb[6] and ra[7] == rb[7]
        assert ulp_diff([np.nan if x is None else x for x in ra[2:6]], [np.nan if x is None else x for x in rb[2:6]]).max() == 0


@pytest.mark.parametrize("late_group", [False, True])
def test_fused_aggregation_splits_a_table_sized_page(pkg, monkeypatch, late_group):
    """a page of 9 M rows (>= 2^23): three leading slices through the insert protocol, the rest in ONE one-pass launch; with a group that first
    appears deep inside the rest the launch is dirty and the rest is re-run through the two-launch path.  Bit for bit the unsplit path
    (TGPU_DISABLE_PAGE_SPLIT), group order included."""
    rng = np.random.default_rng(61)
    n = 9_000_000
    V, D, DT, B = pkg.VARCHAR, pkg.DOUBLE, pkg.DATE, pkg.BIGINT
    k1 = np.frombuffer(b"ANR", dtype=np.uint8)[rng.integers(0, 3, n)].copy()
    k2 = np.frombuffer(b"FO", dtype=np.uint8)[rng.integers(0, 2, n)].copy()
    if late_group:
        k1[7_000_003] = ord("X")
    off = np.arange(n + 1, dtype=np.int32)
    qty = rng.integers(1, 51, n).astype(np.float64)
    price = qty * rng.integers(90000, 210000, n) / 100.0
    disc = rng.integers(0, 11, n) / 100.0
    ship = rng.integers(8036, 10562, n).astype(np.int32)
    page = pkg.Page(pkg.Block(V, k1, None, off), pkg.Block(V, k2, None, off), pkg.Block(D, qty), pkg.Block(D, price), pkg.Block(D, disc), pkg.Block(DT, ship),
                    pkg.Block(B, np.ones(n, dtype=np.int64)))
    T, filt, projs, aggs = _onepass_program(pkg)
    rows = {}
    for mode in ("split", "unsplit"):
        if mode == "unsplit":
            monkeypatch.setenv("TGPU_DISABLE_PAGE_SPLIT", "1")
        ctx = pkg.Context(0)
        ctx.profile_enable(True)
        fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 0, T, filt, projs, [V, V], [0, 1], aggs)
        out = pkg.to_pages(fac.createOperator(), [page])
        rows[mode] = [r for p_ in out for r in p_.rows()]
        launches = ctx.profile().get("fused_filter_group_accumulate_onepass", {"count": 0})["count"]
        assert launches == (1 if mode == "split" else 0)
        ctx.close()
    a, b = rows["split"], rows["unsplit"]
    assert [r[:2] for r in a] == [r[:2] for r in b] and len(a) == (7 if late_group else 6)
    for ra, rb in zip(a, b):
        assert ra[6] == rb[6] and ra[7] == rb[7]
        assert ulp_diff([np.nan if x is None else x for x in ra[2:6]], [np.nan if x is None else x for x in rb[2:6]]).max() == 0


@pytest.mark.parametrize("batch_rows", ["1", "12000", None])
def test_fused_aggregation_one_launch_per_page_raises_expression_errors_a_call_later(pkg, monkeypatch, batch_rows):
    """a filter that divides by zero on a page of the one-pass stream: the page's error word comes back with its counters, one call later; a
    launch over several pages is re-run page by page and the first failing page raises"""
    if batch_rows:
        monkeypatch.setenv("TGPU_ONEPASS_BATCH_ROWS", batch_rows)
    monkeypatch.setenv("TGPU_MODE_PREFIX_ROWS", "5000")
    rng = np.random.default_rng(43)
    pages = _onepass_pages(pkg, rng, 10, 5000, {}, error_page=7)
    T, filt, projs, aggs = _onepass_program(pkg, with_division=True)
    ctx = pkg.Context(0)
    fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 0, T, filt, projs, [pkg.VARCHAR, pkg.VARCHAR], [0, 1], aggs)
    with pytest.raises(pkg.TgpuError) as e:
        pkg.to_pages(fac.createOperator(), pages)
    assert e.value.code == -7
    ctx.close()


def test_fused_aggregation_one_launch_per_page_needs_the_promise_for_borrowed_device_blocks(pkg, monkeypatch):
    """borrowed device blocks are only kept across calls under tgpu_context_set_device_input_stable; results are the same either way"""
    monkeypatch.setenv("TGPU_MODE_PREFIX_ROWS", "5000")
    rng = np.random.default_rng(47)
    host = _onepass_pages(pkg, rng, 10, 6000, {7: "X"})
    T, filt, projs, aggs = _onepass_program(pkg)
    f = pkg.field
    rows = {}
    for promise in (False, True):
        ctx = pkg.Context(0)
        ctx.profile_enable(True)
        ctx.set_device_input_stable(promise)
        # device-resident pages whose memory the test owns (the output pages of an identity operator, handed on as BORROWED device blocks)
        ident = pkg.FilterAndProjectOperatorFactory(ctx, 9, T, None, [f(i, t) for i, t in enumerate(T)])
        owners = [pkg.to_pages(ident.createOperator(), [p], to_host=False)[0] for p in host]
        fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 0, T, filt, projs, [pkg.VARCHAR, pkg.VARCHAR], [0, 1], aggs)
        out = pkg.to_pages(fac.createOperator(), [o.as_device_page() for o in owners])
        rows[promise] = [r for p in out for r in p.rows()]
        assert (ctx.profile().get("fused_filter_group_accumulate_onepass", {"count": 0})["count"] > 0) == promise
        for o in owners:
            o.release()
        ctx.close()
    assert rows[True] == rows[False]


def test_fused_aggregation_keeps_the_buffers_of_library_pages_that_wait_for_their_launch(pkg, monkeypatch):
    """pages of this library (another operator's output) handed on with addInput(OutputPage) and released right away -- the way a driver
    moves pages: the aggregation collects them for a common launch and re-runs them when the launch meets a new group, so it must hold their
    buffers itself (the context's device_input_stable promise is set here and must not be what keeps them: the pool reuses a released
    page's memory for the next page)"""
    monkeypatch.setenv("TGPU_MODE_PREFIX_ROWS", "3000")
    rng = np.random.default_rng(59)
    T, filt, projs, aggs = _onepass_program(pkg)
    host = _onepass_pages(pkg, rng, 30, 6000, {17: "X", 24: "Y"})
    f = pkg.field
    rows = {}
    for mode in ("owned", "host"):
        ctx = pkg.Context(0)
        ctx.set_device_input_stable(True)
        ident = pkg.FilterAndProjectOperatorFactory(ctx, 9, T, None, [f(i, t) for i, t in enumerate(T)]).createOperator()
        op = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 0, T, filt, projs, [pkg.VARCHAR, pkg.VARCHAR], [0, 1], aggs).createOperator()
        for pg in host:
            if mode == "host":
                op.addInput(pg)
                continue
            ident.addInput(pg)
            o = ident.getOutput()
            op.addInput(o)
            o.release()          # its memory goes back to the pool: the next identity output reuses it
        op.finish()
        out = []
        while not op.isFinished():
            o = op.getOutput()
            if o is not None:
                out += o.to_host().rows()
                o.release()
        rows[mode] = out
        op.close(); ident.close(); ctx.close()
    assert rows["owned"] == rows["host"] and 6 + 2 <= len(rows["host"]) <= 6 + 2 * 2


# (the exchange tests live in tests/test_gpu_exchange.py)


# ---- TopN (M/operator/TopNOperator.java) ---------------------------------------------------------------------------------------
_TOPN_TYPES = {"BIGINT": 1, "DOUBLE": 4, "VARCHAR": 6}
_SORT = {"ASC_NULLS_FIRST": 0, "ASC_NULLS_LAST": 1, "DESC_NULLS_FIRST": 2, "DESC_NULLS_LAST": 3}


@pytest.mark.parametrize("name", ["testSingleFieldKey", "testMultiFieldKey", "testReverseOrder"])
def test_top_n_golden(pkg, ctx, name):
    # T/operator/TestTopNOperator.java:77-165: the literal pages and expectations of the reference's own tests
    case = GOLD["top_n"][name]
    types = [_TOPN_TYPES[t] for t in case["types"]]
    pages = [pkg.Page(*[pkg.Block(t, [r[i] for r in pg]) for i, t in enumerate(types)]) for pg in case["pages"]]
    fac = pkg.TopNOperatorFactory(ctx, 0, types, case["n"], case["sort_channels"], [_SORT[o] for o in case["sort_orders"]])
    out = pkg.to_pages(fac.createOperator(), pages)
    assert [list(r) for p in out for r in p.rows()] == case["expect_rows"]


def test_top_n_limit_zero(pkg, ctx):
    # T/operator/TestTopNOperator.java:167-183
    fac = pkg.TopNOperatorFactory(ctx, 0, [pkg.BIGINT], 0, [0], [pkg.DESC_NULLS_LAST])
    op = fac.createOperator()
    assert op.getOutput() is None and op.isFinished() and not op.needsInput() and op.getOutput() is None
    op.close()


@pytest.mark.parametrize("n_rows,n", [(1, 5), (1000, 1), (5000, 100), (300_000, 10), (300_000, 5000), (6_000, 100_000)])
def test_top_n_matches_oracle(pkg, ctx, oracle, n_rows, n):
    """random pages (nulls, NaN / +-0.0 / infinities, few distinct values = many ties, varchar longer than the 8-byte order code)
    over every sort order and multi-channel sort keys: the rows equal the oracle's, including the input order of equal rows"""
    rng = np.random.default_rng(53 + n_rows % 89 + n)
    dbl = rng.integers(-5, 6, n_rows).astype(np.float64) / 4.0
    special = rng.integers(0, n_rows, max(n_rows // 50, 1))
    dbl[special] = rng.choice([np.nan, np.inf, -np.inf, -0.0, 0.0], len(special))
    strs = [None if k % 17 == 0 else "key-prefix-%03d%s" % (k % 40, "x" * (k % 3)) for k in rng.integers(0, 1000, n_rows)]
    blocks = [pkg.Block(pkg.DOUBLE, dbl, (rng.random(n_rows) < 0.05).astype(np.uint8)), pkg.Block(pkg.VARCHAR, strs),
              rand_block(pkg, rng, pkg.BIGINT, n_rows, 0.05, (-50, 50)), rand_block(pkg, rng, pkg.DATE, n_rows, 0.0, (9000, 9020)),
              pkg.Block(pkg.BIGINT, np.arange(n_rows, dtype=np.int64))]
    types = [pkg.DOUBLE, pkg.VARCHAR, pkg.BIGINT, pkg.DATE, pkg.BIGINT]
    ocols = [ocol(oracle, b) for b in blocks]
    # three pages, so that the streaming path (per-page winners + final selection) is exercised
    cuts = [0, n_rows // 3, n_rows // 2, n_rows]
    pages = [pkg.Page(*[pkg.Block(t, b.to_list()[a:z]) if t == pkg.VARCHAR else
                        pkg.Block(t, b.values[a:z], None if b.nulls is None else b.nulls[a:z]) for t, b in zip(types, blocks)]) for a, z in zip(cuts[:-1], cuts[1:]) if z > a]
    for sort_channels, sort_orders in ([[0], [pkg.DESC_NULLS_LAST]], [[0], [pkg.ASC_NULLS_FIRST]], [[1, 2], [pkg.ASC_NULLS_LAST, pkg.DESC_NULLS_FIRST]],
                                      [[3, 0, 1], [pkg.DESC_NULLS_FIRST, pkg.ASC_NULLS_LAST, pkg.DESC_NULLS_LAST]]):
        fac = pkg.TopNOperatorFactory(ctx, 0, types, n, sort_channels, sort_orders)
        out = pkg.to_pages(fac.createOperator(), pages)
        got = np.concatenate([p.getBlock(4).values for p in out]) if out else np.zeros(0, dtype=np.int64)
        want = oracle.top_n(ocols, n, sort_channels, sort_orders)
        assert np.array_equal(got, want), (sort_channels, sort_orders)


# ---- OrderBy (M/operator/OrderByOperator.java) ------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["testSingleFieldKey", "testMultiFieldKey", "testReverseOrder"])
def test_order_by_golden(pkg, ctx, name):
    # T/operator/TestOrderByOperator.java:128-228
    case = GOLD["order_by"][name]
    types = [_TOPN_TYPES[t] for t in case["types"]]
    pages = [pkg.Page(*[pkg.Block(t, [r[i] for r in pg]) for i, t in enumerate(types)]) for pg in case["pages"]]
    fac = pkg.OrderByOperatorFactory(ctx, 0, types, case["output_channels"], 10, case["sort_channels"], [_SORT[o] for o in case["sort_orders"]])
    out = pkg.to_pages(fac.createOperator(), pages)
    assert [list(r) for p in out for r in p.rows()] == case["expect_rows"]


def test_order_by_sequence_and_oracle(pkg, ctx, oracle):
    # T/operator/TestOrderByOperator.java:90-126: 80 000-row sequence page sorted descending
    n = 80_000
    page = pkg.Page(pkg.Block(pkg.BIGINT, np.arange(n, dtype=np.int64)), pkg.Block(pkg.DOUBLE, np.arange(n, dtype=np.float64)))
    fac = pkg.OrderByOperatorFactory(ctx, 0, [pkg.BIGINT, pkg.DOUBLE], [1], 10, [0], [pkg.DESC_NULLS_LAST])
    out = pkg.to_pages(fac.createOperator(), [page])
    got = np.concatenate([p.getBlock(0).values for p in out])
    assert np.array_equal(got, np.arange(n, dtype=np.float64)[::-1])
    # random multi-channel sort with nulls and ties over several pages == the oracle's order (equal rows in input order)
    rng = np.random.default_rng(61)
    m = 120_000
    blocks = [rand_block(pkg, rng, pkg.BIGINT, m, 0.1, (-20, 20)), rand_block(pkg, rng, pkg.DOUBLE, m, 0.1), pkg.Block(pkg.BIGINT, np.arange(m, dtype=np.int64))]
    types = [pkg.BIGINT, pkg.DOUBLE, pkg.BIGINT]
    pages = [pkg.Page(*[pkg.Block(t, b.values[a:z], None if b.nulls is None else b.nulls[a:z]) for t, b in zip(types, blocks)]) for a, z in ((0, 50_000), (50_000, m))]
    fac = pkg.OrderByOperatorFactory(ctx, 0, types, [2], 10, [0, 1], [pkg.ASC_NULLS_FIRST, pkg.DESC_NULLS_LAST])
    out = pkg.to_pages(fac.createOperator(), pages)
    got = np.concatenate([p.getBlock(0).values for p in out])
    want = oracle.top_n([ocol(oracle, b) for b in blocks], m, [0, 1], [0, 3])
    assert np.array_equal(got, want)


# ---------------------------------------------------------------------------------------------------------------------
# SerializedPage <-> HBM: byte-exact against the numpy restatement of the reference's serde
# ---------------------------------------------------------------------------------------------------------------------
SERDE_TYPES = lambda pkg: [pkg.BIGINT, pkg.INTEGER, pkg.DATE, pkg.DOUBLE, pkg.BOOLEAN, pkg.VARCHAR]


def _serde_cols(oracle, blocks):
    """oracle columns of the blocks as the product sees them: the ingest drops a null vector that holds no null, so such a block is
    written with mayHaveNull = 0 where Java keeps the flag of an all-false valueIsNull array (tgpu.h; both decode to equal blocks)"""
    cols = [ocol(oracle, b) for b in blocks]
    for c in cols:
        if c.nulls is not None and not c.nulls.any():
            c.nulls = None
    return cols


def _assert_same_page(pkg, got, blocks, n):
    assert got.position_count == n and len(got.blocks) == len(blocks)
    for g, b in zip(got.blocks, blocks):
        b = b.flatten()
        nl = np.zeros(n, dtype=np.uint8) if b.nulls is None else np.asarray(b.nulls[:n], dtype=np.uint8)
        gn = np.zeros(n, dtype=np.uint8) if g.nulls is None else np.asarray(g.nulls[:n], dtype=np.uint8)
        assert np.array_equal(gn, nl)
        if b.type == pkg.VARCHAR:
            assert g.to_list() == b.to_list()
        else:
            keep = nl == 0
            assert np.array_equal(np.asarray(g.values[:n])[keep].view(np.uint8), np.asarray(b.values[:n])[keep].view(np.uint8))


@pytest.mark.parametrize("n,null_frac", [(0, 0.0), (1, 0.0), (1, 1.0), (7, 0.4), (8, 0.4), (9, 0.4), (1000, 0.0), (1000, 0.3), (100003, 0.2), (100003, 1.0)])
def test_serialize_page_matches_reference_bytes(pkg, ctx, oracle, n, null_frac):
    rng = np.random.default_rng(n + int(null_frac * 10))
    blocks = [rand_block(pkg, rng, t, n, null_frac=null_frac) for t in SERDE_TYPES(pkg)]
    page = pkg.Page(*blocks, position_count=n)
    want = oracle.serialize_page(_serde_cols(oracle, blocks))
    got = ctx.serialize_page(page)
    assert got == want
    # and back: the reference's bytes decode into the same page on the GPU
    back = ctx.deserialize_page(want, SERDE_TYPES(pkg))
    _assert_same_page(pkg, back.to_host(), blocks, n)
    back.release()


def test_serde_known_answer_sizes_on_gpu(pkg, ctx, oracle):
    # TestPagesSerde.java:64-110 (tests/golden): sizes the reference asserts for its own writer
    g = GOLD["pages_serde"]
    one = ctx.serialize_page(pkg.Page(pkg.Block(pkg.BIGINT, np.array([123], dtype=np.int64))))
    two = ctx.serialize_page(pkg.Page(pkg.Block(pkg.BIGINT, np.array([123, 456], dtype=np.int64))))
    assert len(one) - g["bigint_page_overhead"] == g["bigint_first_value"] and len(two) - len(one) == g["bigint_second_value"]
    a = ctx.serialize_page(pkg.Page(pkg.Block(pkg.VARCHAR, ["alice"])))
    b = ctx.serialize_page(pkg.Page(pkg.Block(pkg.VARCHAR, ["alice", "bob"])))
    assert len(a) - g["varchar_empty_page_bytes"] == g["varchar_alice"] and len(b) - len(a) == g["varchar_bob"]
    # testRoundTrip: three identical VARCHAR channels
    blk = pkg.Block(pkg.VARCHAR, g["roundtrip_strings"])
    data = ctx.serialize_page(pkg.Page(blk, blk, blk))
    out = ctx.deserialize_page(data, [pkg.VARCHAR] * 3)
    assert [b.to_list() for b in out.to_host().blocks] == [g["roundtrip_strings"]] * 3
    out.release()


def test_deserialize_rle_and_dictionary_blocks(pkg, ctx, oracle):
    O = oracle
    # the empty BIGINT page exactly as the Java writer emits it (52 bytes): RLE around a one-position null block
    null_block = O.serialize_block(O.Col(O.BIGINT, [0], nulls=[1]))
    data = O.serialized_page(0, [O.rle_block(null_block, 0)])
    assert len(data) == GOLD["pages_serde"]["bigint_empty_page_bytes"]
    out = ctx.deserialize_page(data, [pkg.BIGINT])
    assert out.position_count == 0
    out.release()
    # RLE of a value and of a null; DICTIONARY over varchar and double dictionaries with nulls
    rng = np.random.default_rng(5)
    n = 777
    ids = rng.integers(0, 5, n)
    words = ["", "a", None, "dictionary", "zz"]
    dvals = np.array([1.5, -0.0, np.nan, 0.0, 7e300])
    dnull = np.array([0, 0, 0, 1, 0], dtype=np.uint8)
    blocks = [O.rle_block(O.serialize_block(O.Col(O.BIGINT, [42])), n), O.rle_block(O.serialize_block(O.Col(O.VARCHAR, [None])), n),
              O.dictionary_block(O.serialize_block(O.Col(O.VARCHAR, words)), ids), O.dictionary_block(O.serialize_block(O.Col(O.DOUBLE, dvals, dnull)), ids)]
    data = O.serialized_page(n, blocks)
    types = [pkg.BIGINT, pkg.VARCHAR, pkg.VARCHAR, pkg.DOUBLE]
    out = ctx.deserialize_page(data, types)
    host = out.to_host()
    _, want = O.deserialize_page(data, types)
    assert host.blocks[0].to_list() == [42] * n
    assert host.blocks[1].to_list() == [None] * n
    assert host.blocks[2].to_list() == [words[i] for i in ids]
    keep = dnull[ids] == 0
    assert np.array_equal(np.asarray(host.blocks[3].nulls[:n]) != 0, ~keep)
    assert np.array_equal(np.asarray(host.blocks[3].values[:n])[keep].view(np.int64), dvals[ids][keep].view(np.int64))
    assert np.array_equal(want[3].values[keep].view(np.int64), dvals[ids][keep].view(np.int64))
    out.release()


def test_deserialize_lz4_compressed_pages(pkg, ctx, oracle):
    """exchange.compression-enabled: the payload of a SerializedPage is one LZ4 block (M/execution/buffer/PagesSerde.java:73-93,153-165),
    inflated on the device.  Blocks come from the oracle's greedy compressor (valid LZ4, checked against its own decompressor): long
    literal runs, long and overlapping matches (runs of equal bytes: offset 1), matches at the 64 KB offset limit, incompressible data."""
    rng = np.random.default_rng(23)
    n = 40_000
    cases = [
        [pkg.Block(pkg.BIGINT, np.zeros(n, dtype=np.int64))],                                                   # one long overlapping match
        [pkg.Block(pkg.BIGINT, np.arange(n, dtype=np.int64)), pkg.Block(pkg.VARCHAR, ["row%d" % (i % 13) for i in range(n)])],
        [rand_block(pkg, rng, pkg.BIGINT, n, 0.1, (0, 7)), rand_block(pkg, rng, pkg.DOUBLE, n, 0.0), rand_block(pkg, rng, pkg.VARCHAR, n, 0.2, (0, 30))],
        [pkg.Block(pkg.BIGINT, np.tile(rng.integers(-2**62, 2**62, 8192), 5)[:n].astype(np.int64))],              # matches 65 536 bytes back
        [pkg.Block(pkg.BIGINT, np.array([], dtype=np.int64))],
    ]
    for blocks in cases:
        page = pkg.Page(*blocks)
        plain = oracle.serialize_page([ocol(oracle, b) for b in blocks])
        packed = oracle.compress_serialized_page(plain)
        assert packed[4] == 1 and oracle.lz4_block_decompress(packed[13:], int.from_bytes(packed[5:9], "little")) == plain[13:]
        got = ctx.deserialize_page(packed, [b.type for b in blocks]).to_host()
        assert got.rows() == page.rows()
    # a corrupt block (an offset reaching before the start of the output) fails cleanly
    bad = bytearray(oracle.compress_serialized_page(oracle.serialize_page([oracle.Col(oracle.BIGINT, np.zeros(1000, dtype=np.int64))])))
    at, lit = 14, bad[13] >> 4                       # the first record: token, [length bytes], literals, then the 2-byte offset
    if lit == 15:
        while True:
            lit += bad[at]
            at += 1
            if bad[at - 1] != 255:
                break
    at += lit
    bad[at:at + 2] = (60000).to_bytes(2, "little")
    with pytest.raises(pkg.TgpuError) as e:
        ctx.deserialize_page(bytes(bad), [pkg.BIGINT])
    assert e.value.code == -1


def test_deserialize_rejects_what_it_cannot_read(pkg, ctx, oracle):
    data = bytearray(oracle.serialize_page([oracle.Col(oracle.BIGINT, [1, 2, 3])]))
    for marker in (2, 3):   # PageCodecMarker.java:24-25 ENCRYPTED (alone, or with COMPRESSED)
        bad = bytearray(data)
        bad[4] = marker
        with pytest.raises(pkg.TgpuError) as e:
            ctx.deserialize_page(bytes(bad), [pkg.BIGINT])
        assert e.value.code == -8
    with pytest.raises(pkg.TgpuError) as e:
        ctx.deserialize_page(bytes(data[:-5]), [pkg.BIGINT])
    assert e.value.code == -1
    with pytest.raises(pkg.TgpuError) as e:   # a LONG_ARRAY block where the consumer expects a 4-byte type
        ctx.deserialize_page(bytes(data), [pkg.INTEGER])
    assert e.value.code == -1
    with pytest.raises(pkg.TgpuError):
        ctx.deserialize_page(bytes(data), [pkg.BIGINT, pkg.BIGINT])


def test_deserialize_rejects_inconsistent_offsets_and_null_counts(pkg, ctx, oracle):
    """corrupt exchange / spill pages must fail on the host (the JVM would throw IndexOutOfBounds), never reach a kernel:
    VARIABLE_WIDTH end offsets that are negative, descending or beyond the block size; a nonNullCount that disagrees with the null bits"""
    import struct
    good = oracle.serialize_page([oracle.Col(oracle.VARCHAR, ["ab", "cde", "", "f"])])
    ends = struct.pack("<4i", 2, 5, 5, 6)
    at = good.index(ends)
    assert ctx.deserialize_page(good, [pkg.VARCHAR]).to_host().getBlock(0).to_list() == ["ab", "cde", "", "f"]
    for tampered in [(2, 1, 5, 6), (-1, 5, 5, 6), (2, 5, 7, 6), (2, 5, 5, 9)]:
        bad = bytearray(good)
        bad[at:at + 16] = struct.pack("<4i", *tampered)
        with pytest.raises(pkg.TgpuError) as e:
            ctx.deserialize_page(bytes(bad), [pkg.VARCHAR])
        assert e.value.code == -1, tampered
    col = oracle.Col(oracle.BIGINT, np.array([7, 0, 9, 0, 11], dtype=np.int64), np.array([0, 1, 0, 1, 0], dtype=np.uint8))
    good = oracle.serialize_page([col])
    assert ctx.deserialize_page(good, [pkg.BIGINT]).to_host().getBlock(0).to_list() == [7, None, 9, None, 11]
    # LongArrayBlockEncoding with nulls: ... mayHaveNull(1) | null bits (1 byte for 5 positions) | nonNullCount(int32) | values
    bits = bytes([0b01010000])
    at = good.index(b"\x01" + bits + struct.pack("<i", 3)) + 2
    for count in (2, 4, 5, 0):
        bad = bytearray(good)
        bad[at:at + 4] = struct.pack("<i", count)
        with pytest.raises(pkg.TgpuError) as e:
            ctx.deserialize_page(bytes(bad), [pkg.BIGINT])
        assert e.value.code == -1, count


def test_serde_round_trip_device_page_full_size(pkg, ctx):
    # size-independent property at a full page: serialize(deserialize(bytes)) == bytes, through a device-resident page (the
    # decoded OutputPage is fed back without leaving HBM)
    rng = np.random.default_rng(11)
    n = 4_000_000
    blocks = [rand_block(pkg, rng, pkg.BIGINT, n, null_frac=0.1), rand_block(pkg, rng, pkg.DOUBLE, n), rand_block(pkg, rng, pkg.DATE, n, null_frac=0.5),
              pkg.Block(pkg.VARCHAR, np.frombuffer(rng.integers(65, 91, n).astype(np.uint8).tobytes(), dtype=np.uint8), None, np.arange(n + 1, dtype=np.int32))]
    types = [pkg.BIGINT, pkg.DOUBLE, pkg.DATE, pkg.VARCHAR]
    data = ctx.serialize_page(pkg.Page(*blocks, position_count=n))
    out = ctx.deserialize_page(data, types)
    again = ctx.serialize_page(out.as_device_page())
    assert again == data
    out.release()


# ---------------------------------------------------------------------------------------------------------------------
# PartitionedOutputOperator against the row-at-a-time PagePartitioner restatement
# ---------------------------------------------------------------------------------------------------------------------
def _drain_partitions(op, parts):
    got = [[] for _ in range(parts)]
    order = []
    while True:
        e = op.poll()
        if e is None:
            break
        p, out = e
        order.append(p)
        got[p] += out.to_host().rows()
        out.release()
    return got, order


@pytest.mark.parametrize("parts,replicates_any_row,null_channel,hash_channel,local",
                         [(1, False, -1, -1, False), (8, False, -1, -1, False), (7, False, 0, -1, False), (8, True, -1, -1, False), (5, True, 1, -1, False),
                          (300, False, 0, -1, False), (8, False, -1, 3, False), (1024, False, -1, -1, False), (16, False, 1, -1, True), (64, True, -1, 3, True)])
def test_partitioned_output_operator_vs_page_partitioner(pkg, ctx, oracle, parts, replicates_any_row, null_channel, hash_channel, local):
    rng = np.random.default_rng(parts + 17 * int(replicates_any_row) + null_channel)
    types = [pkg.BIGINT, pkg.VARCHAR, pkg.DOUBLE, pkg.BIGINT]
    fac = pkg.PartitionedOutputOperatorFactory(ctx, 31, types, [0, 1], parts, hash_channel=hash_channel, replicates_any_row=replicates_any_row, null_channel=null_channel,
                                               local=local)
    op = fac.createOperator()
    ref = oracle.PagePartitioner(parts, replicates_any_row, null_channel, local)
    assert op.needsInput() and op.getOutput() is None
    total_rows = 0
    for n in (0, 1, 5000, 33333):   # several pages: the "replicate any row" state carries over (:411-418)
        blocks = [rand_block(pkg, rng, pkg.BIGINT, n, 0.03, (0, 10**6)), rand_block(pkg, rng, pkg.VARCHAR, n, 0.03, (0, 50)), rand_block(pkg, rng, pkg.DOUBLE, n, 0.1),
                  pkg.Block(pkg.BIGINT, rng.integers(-2**63, 2**63 - 1, n))]
        page = pkg.Page(*blocks, position_count=n)
        raw = blocks[3].values if hash_channel >= 0 else oracle.hash_rows([ocol(oracle, blocks[0]), ocol(oracle, blocks[1])])
        want = ref.partition_page([ocol(oracle, b) for b in blocks], raw) if n else [[] for _ in range(parts)]
        op.addInput(page)
        got, order = _drain_partitions(op, parts)
        assert order == sorted(order) and len(set(order)) == len(order)   # flush order: ascending partition, one page each (:451-470)
        rows = page.rows()
        for p in range(parts):
            assert got[p] == [rows[i] for i in want[p]], (n, p)
        total_rows += sum(len(w) for w in want)
    assert op.info()["rowsAdded"] == total_rows
    op.finish()
    assert op.isFinished() and op.poll() is None
    op.close()
    fac.close()


def test_partitioned_output_feeds_the_serde(pkg, ctx, oracle):
    # the reference's flush: partition page -> PagesSerde.serialize -> OutputBuffer (:451-486); bytes must decode to the same rows
    rng = np.random.default_rng(8)
    n, parts = 20000, 4
    types = [pkg.BIGINT, pkg.VARCHAR]
    blocks = [rand_block(pkg, rng, pkg.BIGINT, n, 0.05, (0, 1000)), rand_block(pkg, rng, pkg.VARCHAR, n, 0.05, (0, 30))]
    page = pkg.Page(*blocks)
    fac = pkg.PartitionedOutputOperatorFactory(ctx, 32, types, [0], parts)
    op = fac.createOperator()
    op.addInput(page)
    pid = oracle.partition_remote(oracle.hash_rows([ocol(oracle, blocks[0])]), parts)
    rows = page.rows()
    while True:
        e = op.poll()
        if e is None:
            break
        p, out = e
        data = ctx.serialize_page(out.as_device_page())
        _, cols = oracle.deserialize_page(data, types)
        back = ctx.deserialize_page(data, types)
        assert back.to_host().rows() == [rows[i] for i in np.nonzero(pid == p)[0]]
        assert cols[0].n == int((pid == p).sum())
        back.release()
        out.release()
    op.close()
    fac.close()


def test_partitioned_output_rejects_constant_arguments(pkg, ctx):
    with pytest.raises(pkg.TgpuError) as e:
        pkg.PartitionedOutputOperatorFactory(ctx, 33, [pkg.BIGINT], [-1], 4)
    assert e.value.code == -8
    with pytest.raises(pkg.TgpuError) as e:   # LocalPartitionGenerator masks with count - 1
        pkg.PartitionedOutputOperatorFactory(ctx, 33, [pkg.BIGINT], [0], 6, local=True)
    assert e.value.code == -1


# ---------------------------------------------------------------------------------------------------------------------
# MergePages as an operator: the cases of T/operator/project/TestMergePages.java + random streams vs the
# page-at-a-time restatement (same page boundaries, same rows)
# ---------------------------------------------------------------------------------------------------------------------
MERGE_TYPES = lambda pkg: [pkg.BIGINT, pkg.INTEGER, pkg.DOUBLE]   # TestMergePages.java:42 uses BIGINT, REAL, DOUBLE: REAL has INTEGER's 4-byte layout


def _seq_page(pkg, types, n, start=0):
    return pkg.Page(*[pkg.Block(t, v) for t, v in zip(types, sequence_page(types, n, *([start] * len(types))))], position_count=n)


def _run_merge(pkg, ctx, oracle, types, pages, min_size, min_rows, max_size):
    fac = pkg.MergePagesOperatorFactory(ctx, 41, types, min_size, min_rows, max_size)
    op = fac.createOperator()
    ref = oracle.MergePages(min_size, min_rows, max_size)
    got, want = [], []
    for page in pages:
        assert op.needsInput()
        op.addInput(page)
        want += ref.add([ocol(oracle, b) for b in page.blocks])
        while True:
            o = op.getOutput()
            if o is None:
                break
            got.append(o.to_host().rows())
            o.release()
    op.finish()
    want += ref.finish()
    while not op.isFinished():
        o = op.getOutput()
        assert o is not None
        got.append(o.to_host().rows())
        o.release()
    op.close()
    fac.close()
    all_rows = [p.rows() for p in pages]
    assert got == [[all_rows[i][r] for i, r in out] for out in want]
    return got


def test_merge_pages_reference_cases(pkg, ctx, oracle):
    types = MERGE_TYPES(pkg)
    size = lambda page: oracle.page_size_in_bytes([ocol(oracle, b) for b in page.blocks])
    big = 2**31 - 1
    page = _seq_page(pkg, types, 10)
    assert size(page) == 10 * (9 + 5 + 9)
    # testMinPageSizeThreshold (:45-59), testMinRowCountThreshold (:61-76): the page passes through as it is
    assert _run_merge(pkg, ctx, oracle, types, [page], size(page), big, big) == [page.rows()]
    assert _run_merge(pkg, ctx, oracle, types, [page], 1024 * 1024, 10, big) == [page.rows()]
    # testBufferSmallPages (:107-124): two halves come out as one page
    whole = _seq_page(pkg, types, 20)
    halves = [_seq_page(pkg, types, 10, 0), _seq_page(pkg, types, 10, 10)]
    assert _run_merge(pkg, ctx, oracle, types, halves, size(whole) + 1, 21, big) == [whole.rows()]
    # testFlushOnBigPage (:126-143): the buffered small page first, then the big page
    small, large = _seq_page(pkg, types, 10), _seq_page(pkg, types, 100)
    assert _run_merge(pkg, ctx, oracle, types, [small, large], size(large), 100, big) == [small.rows(), large.rows()]
    # testFlushOnFullPage (:145-164): BIGINT only, max page size = the whole page: two full pages out of four halves
    t1 = [pkg.BIGINT]
    whole = _seq_page(pkg, t1, 20)
    halves = [_seq_page(pkg, t1, 10, 0), _seq_page(pkg, t1, 10, 10)]
    assert _run_merge(pkg, ctx, oracle, t1, halves + halves, size(whole) // 2 + 1, 11, size(whole)) == [whole.rows(), whole.rows()]


def test_merge_pages_random_stream_and_device_input(pkg, ctx, oracle):
    rng = np.random.default_rng(12)
    types = [pkg.BIGINT, pkg.VARCHAR, pkg.DOUBLE, pkg.BOOLEAN]
    pages = []
    for n in rng.choice([0, 1, 3, 50, 700, 5000, 40000], 40):
        pages.append(pkg.Page(*[rand_block(pkg, rng, t, int(n), null_frac=0.1 if i != 2 else 0.0) for i, t in enumerate(types)], position_count=int(n)))
    _run_merge(pkg, ctx, oracle, types, pages, 64 * 1024, 4096, 256 * 1024)
    # device-resident (borrowed) input pages: a passing-through page is copied once, results unchanged
    fp = pkg.FilterAndProjectOperatorFactory(ctx, 42, [pkg.BIGINT], None, [pkg.field(0, pkg.BIGINT)])
    op = fp.createOperator()
    op.addInput(pkg.Page(pkg.Block(pkg.BIGINT, np.arange(100000, dtype=np.int64))))
    dev = op.getOutput()
    fac = pkg.MergePagesOperatorFactory(ctx, 43, [pkg.BIGINT], 1024, 10, 1 << 20)
    m = fac.createOperator()
    m.addInput(dev.as_device_page())
    out = m.getOutput()
    dev.release()   # the input may go away: the operator owns what it hands out
    assert out.to_host().blocks[0].to_list() == list(range(100000))
    out.release()
    # the same page handed over as an output-page handle (tgpu_operator_add_input_output_page): buffers shared, no copy; small pages
    # given that way are appended like any other, and every operator accepts the handle form
    op.addInput(pkg.Page(pkg.Block(pkg.BIGINT, np.arange(5, dtype=np.int64))))   # below both thresholds of `fac`: buffered
    small = op.getOutput()
    op.addInput(pkg.Page(pkg.Block(pkg.BIGINT, np.arange(100000, 200000, dtype=np.int64))))
    big = op.getOutput()
    m.addInput(small)
    assert m.getOutput() is None
    m.addInput(big)
    small.release(); big.release()
    outs = []
    while True:
        o = m.getOutput()
        if o is None:
            break
        outs.append(o.to_host().blocks[0].to_list())
        o.release()
    assert outs == [list(range(5)), list(range(100000, 200000))]
    agg = pkg.HashAggregationOperatorFactory(ctx, 45, [pkg.BIGINT], [0], [(pkg.COUNT_ALL, -1)], expected_groups=16)
    a = agg.createOperator()
    op.addInput(pkg.Page(pkg.Block(pkg.BIGINT, np.arange(1000, dtype=np.int64) % 7)))
    pg = op.getOutput()
    a.addInput(pg)
    pg.release()
    a.finish()
    res = a.getOutput()
    assert sorted(res.to_host().rows()) == [(k, len(range(k, 1000, 7))) for k in range(7)]
    res.release()
    a.close(); agg.close(); m.close(); fac.close(); op.close(); fp.close()


def test_merge_pages_argument_checks(pkg, ctx):
    for args in ((-1, 1, 10), (1, -1, 10), (1, 1, 0), (11, 1, 10)):   # MergePages.java:102-106
        with pytest.raises(pkg.TgpuError) as e:
            pkg.MergePagesOperatorFactory(ctx, 44, [pkg.BIGINT], *args)
        assert e.value.code == -1


# ---------------------------------------------------------------------------------------------------------------------
# LOOKUP_OUTER / FULL_OUTER joins + LookupOuterOperator against the oracle's pair list + visited set
# ---------------------------------------------------------------------------------------------------------------------
def _outer_join(pkg, ctx, build_pages, probe_page_lists, types_b, types_p, key_b, key_p, join_type, out_b, out_p, fused=None):
    """several probe operators over one lookup source, then the outer operator; returns (probe rows per operator, outer rows)"""
    bf = pkg.HashBuilderOperatorFactory(ctx, 1, types_b, out_b, key_b)
    if fused is None:
        jf = pkg.LookupJoinOperatorFactory(ctx, 2, bf.lookup_source_factory, types_p, key_p, probe_output_channels=out_p, join_type=join_type)
    else:
        jf = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 2, bf.lookup_source_factory, types_p, fused[0], fused[1], key_p, probe_output_channels=out_p, join_type=join_type)
    of = pkg.LookupOuterOperatorFactory(ctx, 3, bf.lookup_source_factory, [types_p[c] for c in out_p] if fused is None else fused[2])
    build = bf.createOperator()
    probes = [jf.createOperator() for _ in probe_page_lists]
    outer = of.createOperator()
    assert outer.isBlocked() and not outer.needsInput() and outer.getOutput() is None
    for p in build_pages:
        build.addInput(p)
    build.finish()
    assert outer.isBlocked()   # the probes are not done yet (PartitionedLookupSourceFactory.java:259-297)
    probe_rows = []
    for op, pages in zip(probes, probe_page_lists):
        rows = []
        for pg in pkg.to_pages(op, pages):
            rows.extend(pg.rows())
        probe_rows.append(rows)
        op.close()
        assert outer.isBlocked()   # ... and the factory may still create more probe operators
    jf.noMoreOperators()
    assert not outer.isBlocked() and not build.isFinished()   # the table stays alive for the outer operator
    o = outer.getOutput()
    outer_rows = o.to_host().rows() if o is not None else []
    if o is not None:
        o.release()
    assert outer.isFinished() and outer.getOutput() is None
    assert build.isFinished()
    build.close(); outer.close(); jf.close(); of.close(); bf.close()
    return probe_rows, outer_rows


@pytest.mark.parametrize("join_type", ["LOOKUP_OUTER", "FULL_OUTER"])
@pytest.mark.parametrize("key_type,domain", [("BIGINT", (0, 3000)), ("VARCHAR", (0, 400)), ("BIGINT", (10**12, 10**12 + 50))])
def test_outer_joins_vs_oracle(pkg, ctx, oracle, join_type, key_type, domain):
    rng = np.random.default_rng(len(join_type) + domain[1] % 97)
    kt = getattr(pkg, key_type)
    jt = getattr(pkg, join_type)
    types = [kt, pkg.BIGINT]
    build_pages = []
    for n in (900, 1, 1400):
        build_pages.append(pkg.Page(rand_block(pkg, rng, kt, n, 0.04, domain), pkg.Block(pkg.BIGINT, rng.integers(0, 10**9, n).astype(np.int64))))
    probe_lists = [[pkg.Page(rand_block(pkg, rng, kt, n, 0.04, domain), pkg.Block(pkg.BIGINT, np.arange(n, dtype=np.int64)))] for n in (1200, 800)]
    probe_rows, outer_rows = _outer_join(pkg, ctx, build_pages, probe_lists, types, types, [0], [0], jt, out_b=[0, 1], out_p=[0, 1])
    cat = [pkg.Block(types[c], [v for pg in build_pages for v in pg.getBlock(c).to_list()]) for c in range(2)]
    brows = list(zip(cat[0].to_list(), cat[1].to_list()))
    ph = oracle.PagesHash([ocol(oracle, cat[0])])
    visited = set()
    for rows, pages in zip(probe_rows, probe_lists):
        blk = pages[0].getBlock(0)
        op, ob = ph.probe([ocol(oracle, blk)], probe_outer=(join_type == "FULL_OUTER"))
        prow = pages[0].rows()
        assert rows == [prow[i] + (brows[j] if j >= 0 else (None, None)) for i, j in zip(op, ob)]
        visited |= {int(j) for j in ob if j >= 0}
    # OuterPositionIterator: every build position nobody matched (null keys included), ascending; probe channels null
    assert outer_rows == [(None, None) + brows[j] for j in range(len(brows)) if j not in visited]


def test_outer_join_fused_probe_and_empty_sides(pkg, ctx, oracle):
    # the JIT-fused filter + probe records its matches too (DIRECT table layout: dense unique keys)
    B = pkg.BIGINT
    n_b, n_p = 5000, 20000
    build = pkg.Page(pkg.Block(B, np.arange(n_b, dtype=np.int64) * 2), pkg.Block(B, np.arange(n_b, dtype=np.int64) + 7))
    rng = np.random.default_rng(3)
    pk = rng.integers(0, 3 * n_b, n_p).astype(np.int64)
    probe = pkg.Page(pkg.Block(B, pk), pkg.Block(B, np.arange(n_p, dtype=np.int64)))
    f = pkg.field
    fused = (f(1, B) < 15000, [f(0, B), f(1, B)], [B, B])
    probe_rows, outer_rows = _outer_join(pkg, ctx, [build], [[probe]], [B, B], [B, B], [0], [0], pkg.FULL_OUTER, out_b=[1], out_p=[0, 1], fused=fused)
    sel = np.arange(n_p) < 15000
    hit = (pk % 2 == 0) & (pk < 2 * n_b)
    assert probe_rows[0] == [(int(k), int(i), int(k // 2 + 7) if h else None) for k, i, h in zip(pk[sel], np.arange(n_p)[sel], hit[sel])]
    matched = set((pk[sel & hit] // 2).tolist())
    assert outer_rows == [(None, None, j + 7) for j in range(n_b) if j not in matched]
    # T/operator/TestHashJoinOperator.java:1071-1108 testLookupOuterJoinWithEmptyLookupSource: no output at all
    probe_rows, outer_rows = _outer_join(pkg, ctx, [], [[pkg.Page(pkg.Block(pkg.VARCHAR, ["test"]))]], [pkg.VARCHAR], [pkg.VARCHAR], [0], [0], pkg.LOOKUP_OUTER,
                                         out_b=[0], out_p=[0])
    assert probe_rows == [[]] and outer_rows == []
    # no probe operator ever matched anything: the whole build side comes out
    probe_rows, outer_rows = _outer_join(pkg, ctx, [build], [[]], [B, B], [B, B], [0], [0], pkg.LOOKUP_OUTER, out_b=[0, 1], out_p=[0])
    assert outer_rows == [(None, 2 * j, j + 7) for j in range(n_b)]


def test_serde_many_small_random_pages(pkg, ctx, oracle):
    """ragged shapes: every position count from 0 to 70 (all byte-tail lengths of the packed null bits), channels in random order and
    number, all-null and no-null vectors, empty strings"""
    rng = np.random.default_rng(99)
    all_types = SERDE_TYPES(pkg)
    for n in range(0, 71):
        k = int(rng.integers(1, 7))
        types = [all_types[int(i)] for i in rng.integers(0, len(all_types), k)]
        blocks = []
        for t in types:
            frac = float(rng.choice([0.0, 0.0, 0.2, 1.0]))
            if t == pkg.VARCHAR:
                vals = [None if rng.random() < frac else ("" if rng.random() < 0.3 else "s" * int(rng.integers(1, 9))) for _ in range(n)]
                blocks.append(pkg.Block(pkg.VARCHAR, vals))
            else:
                blocks.append(rand_block(pkg, rng, t, n, null_frac=frac))
        # (the ingest drops a null vector that holds no null, so such a block is written with mayHaveNull = 0 where Java keeps the
        # flag of an all-false valueIsNull array; both decode to the same block -- the expectation is normalised the same way)
        want = oracle.serialize_page(_serde_cols(oracle, blocks))
        page = pkg.Page(*blocks, position_count=n)
        assert ctx.serialize_page(page) == want, (n, types)
        back = ctx.deserialize_page(want, types)
        _assert_same_page(pkg, back.to_host(), blocks, n)
        back.release()


def test_two_threads_share_a_context(pkg, ctx, oracle):
    """tgpu.h threading rule: distinct handles of one context may be driven from different threads (ctypes drops the GIL during
    the calls, so the two drivers really overlap): results stay those of the single-threaded run"""
    import threading

    rng = np.random.default_rng(77)
    n = 200_000
    f = pkg.field
    errors, results = [], {}

    def drive(tag, seed):
        try:
            r = np.random.default_rng(seed)
            keys = r.integers(0, 50, n).astype(np.int64)
            vals = r.integers(-1000, 1000, n).astype(np.int64)
            page = pkg.Page(pkg.Block(pkg.BIGINT, keys), pkg.Block(pkg.BIGINT, vals))
            out = []
            for it in range(15):
                fp = pkg.FilterAndProjectOperatorFactory(ctx, 50, [pkg.BIGINT, pkg.BIGINT], f(1, pkg.BIGINT) < 500, [f(0, pkg.BIGINT), f(1, pkg.BIGINT)])
                ag = pkg.HashAggregationOperatorFactory(ctx, 51, [pkg.BIGINT], [0], [(pkg.SUM_BIGINT, 1), (pkg.COUNT_ALL, -1)], expected_groups=64)
                a, b = fp.createOperator(), ag.createOperator()
                a.addInput(page)
                mid = a.getOutput()
                b.addInput(mid)
                mid.release()
                b.finish()
                res = b.getOutput()
                out.append(sorted(res.to_host().rows()))
                res.release()
                a.close(); b.close(); fp.close(); ag.close()
            sel = vals < 500
            want = sorted((int(k), int(vals[sel & (keys == k)].sum()), int((sel & (keys == k)).sum())) for k in np.unique(keys[sel]))
            assert all(o == want for o in out)
            results[tag] = True
        except Exception as e:   # surfaced in the main thread below
            errors.append((tag, repr(e)))

    threads = [threading.Thread(target=drive, args=(i, 1000 + i)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(results) == 3


def test_fused_aggregation_filter_errors_surface(pkg, ctx):
    """an arithmetic error inside the fused filter of FilterProject -> HashAggregation is raised by addInput, like the unfused
    FilterAndProjectOperator would (the error word travels with the group-probe counters: one read-back)"""
    n = 50_000
    a = np.arange(n, dtype=np.int64) + 1
    b = np.ones(n, dtype=np.int64)
    f = pkg.field
    T = [pkg.BIGINT, pkg.BIGINT, pkg.BIGINT]
    keys = np.arange(n, dtype=np.int64) % 3

    def run(divisors):
        fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 60, T, (f(0, pkg.BIGINT) / f(1, pkg.BIGINT)) > 0, [f(2, pkg.BIGINT), f(0, pkg.BIGINT)],
                                                              [pkg.BIGINT], [0], [(pkg.COUNT_ALL, -1)])
        op = fac.createOperator()
        try:
            op.addInput(pkg.Page(pkg.Block(pkg.BIGINT, a), pkg.Block(pkg.BIGINT, divisors), pkg.Block(pkg.BIGINT, keys)))
            op.finish()
            out = op.getOutput()
            rows = sorted(out.to_host().rows())
            out.release()
            return rows
        finally:
            op.close()
            fac.close()

    assert run(b) == [(k, len(range(k, n, 3))) for k in range(3)]
    bad = b.copy()
    bad[31_337] = 0
    with pytest.raises(pkg.TgpuError) as e:
        run(bad)
    assert e.value.code == -7 and "31337" in e.value.message


# ---------------------------------------------------------------------------------------------------------------------
# DynamicFilterSourceOperator against the position-at-a-time restatement
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("max_distinct,max_size,limit", [(10_000, 1 << 30, 1 << 30), (50, 1 << 30, 1 << 30), (10_000, 2_000, 1 << 30), (50, 1 << 30, 3_000),
                                                          (50, 1 << 30, 0), (10_000, 1 << 30, 100)])
def test_dynamic_filter_source_vs_restatement(pkg, ctx, oracle, max_distinct, max_size, limit):
    rng = np.random.default_rng(max_distinct + limit % 1000)
    types = [pkg.BIGINT, pkg.VARCHAR, pkg.DOUBLE, pkg.DATE, pkg.BIGINT]
    channels = [0, 1, 2, 3]
    fac = pkg.DynamicFilterSourceOperatorFactory(ctx, 70, types, channels, max_distinct, max_size, limit)
    op = fac.createOperator()
    ref = oracle.DynamicFilterSource(types, channels, max_distinct, max_size, limit)
    for n in (300, 0, 1000, 2500):
        blocks = [rand_block(pkg, rng, pkg.BIGINT, n, 0.05, (-500, 500)), rand_block(pkg, rng, pkg.VARCHAR, n, 0.05, (0, 40)), rand_block(pkg, rng, pkg.DOUBLE, n, 0.05, (0, 30)),
                  rand_block(pkg, rng, pkg.DATE, n, 1.0 if n == 300 else 0.05, (9000, 9100)), pkg.Block(pkg.BIGINT, np.arange(n, dtype=np.int64))]
        if n >= 1000:
            blocks[2].values[:3] = [np.nan, -0.0, 0.0]
        page = pkg.Page(*blocks, position_count=n)
        assert op.needsInput()
        op.addInput(page)
        out = op.getOutput()   # the page passes through unchanged (:375-381)
        norm = lambda rows: [tuple("nan" if isinstance(v, float) and v != v else v for v in r) for r in rows]
        assert out is not None and norm(out.to_host().rows()) == norm(page.rows())
        out.release()
        ref.add([ocol(oracle, b) for b in blocks])
    op.finish()
    assert op.isFinished()
    for k in range(len(channels)):
        got, want = op.domain(k), ref.domain(k)
        if want[0] == "values" and types[channels[k]] == pkg.DOUBLE:
            assert got[0] == "values" and [float(v) for v in got[1]] == [float(v) for v in want[1]]
        else:
            assert got == want, (k, got[:1], want[:1])
    op.close()
    fac.close()


def test_dynamic_filter_source_small_cases(pkg, ctx):
    # only nulls in an orderable channel -> NONE once the sets were dropped (:366-369); duplicate channels rejected (:105-106)
    fac = pkg.DynamicFilterSourceOperatorFactory(ctx, 71, [pkg.BIGINT, pkg.BIGINT], [0, 1], 2, 1 << 30, 1000)
    op = fac.createOperator()
    op.addInput(pkg.Page(pkg.Block(pkg.BIGINT, [None, None, None, None]), pkg.Block(pkg.BIGINT, np.array([5, 7, 9, 11], dtype=np.int64))))
    op.getOutput().release()
    op.finish()
    assert op.domain(0) == ("none",) and op.domain(1) == ("range", 5, 11)
    op.close()
    fac.close()
    with pytest.raises(pkg.TgpuError):
        pkg.DynamicFilterSourceOperatorFactory(ctx, 72, [pkg.BIGINT], [0, 0], 10, 10, 10)


# ---------------------------------------------------------------------------------------------------------------------
# ownership / lifetime / threading regressions (round-1 review)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["borrowed", "owned"])
def test_identity_projection_output_survives_release_of_its_input(pkg, ctx, mode):
    """tgpu.h ownership rule: an output page is valid until ITS release, whatever happens to the page it was computed from.  A
    FilterAndProject whose filter selects every row passes identity projections through: the passed-through blocks must keep
    (owned input) or get (borrowed device input) buffers of their own -- the upstream page is released and its memory reused
    before the output is read."""
    n = 300_000
    B, V = pkg.BIGINT, pkg.VARCHAR
    f = pkg.field
    vals = np.arange(n, dtype=np.int64)
    strs = [f"s{i % 97}" for i in range(n)]
    nulls = (vals % 11 == 0).astype(np.uint8)
    up = pkg.FilterAndProjectOperatorFactory(ctx, 0, [B, V], None, [f(0, B) + 1, f(1, V)])
    upstream = pkg.to_pages(up.createOperator(), [pkg.Page(pkg.Block(B, vals, nulls), pkg.Block(V, strs))], to_host=False)[0]
    fp = pkg.FilterAndProjectOperatorFactory(ctx, 1, [B, V], pkg.or_(f(0, B) >= 0, pkg.is_null(f(0, B))), [f(0, B), f(1, V)])
    op = fp.createOperator()
    op.addInput(upstream if mode == "owned" else upstream.as_device_page())
    out = op.getOutput()
    upstream.release()
    # the caching allocator hands the freed buffers to the next allocations of the same size: overwrite them
    junk = pkg.to_pages(up.createOperator(), [pkg.Page(pkg.Block(B, vals * 0 - 7), pkg.Block(V, ["zzzzzz"] * n))], to_host=False)
    ctx.synchronize()
    host = out.to_host()
    want = [None if nulls[i] else int(vals[i]) + 1 for i in range(n)]
    assert host.getBlock(0).to_list() == want
    assert host.getBlock(1).to_list() == strs
    out.release()
    for j in junk:
        j.release()
    op.close()


def test_context_destroyed_before_its_handles(pkg):
    """tgpu_context_destroy defers until the last factory / operator / output page created from the context is gone, whatever the
    order in which the caller drops its handles (an operator finalised after its context used to crash the host)"""
    c = pkg.Context(0)
    f = pkg.field
    fac = pkg.FilterAndProjectOperatorFactory(c, 0, [pkg.BIGINT], f(0, pkg.BIGINT) > 2, [f(0, pkg.BIGINT) * 2])
    op = fac.createOperator()
    op.addInput(pkg.Page(pkg.Block(pkg.BIGINT, np.arange(10, dtype=np.int64))))
    out = op.getOutput()
    g = pkg.GroupByHash(c, [pkg.BIGINT], [0])
    c.close()                                            # requested; handles are still alive
    assert out.to_host().getBlock(0).to_list() == [6, 8, 10, 12, 14, 16, 18]
    assert list(g.getGroupIds(pkg.Page(pkg.Block(pkg.BIGINT, np.array([5, 5, 9], dtype=np.int64))))) == [0, 0, 1]
    out.release()
    op.close()
    g.close()
    fac.close()                                          # the last handle destroys the context
    c2 = pkg.Context(0)                                  # and the device is still usable
    assert c2.hash_page(pkg.Page(pkg.Block(pkg.BIGINT, np.array([1], dtype=np.int64))), [0]).shape == (1,)
    c2.close()


def test_entry_points_bind_the_calling_thread_to_the_context_device(pkg, ctx):
    """HIP's current device is per thread; every entry point that takes a handle binds the caller to the context's device first
    (c_api.cpp bind_thread) -- also on a thread that never touched HIP before"""
    import threading
    L = pkg._lib.lib()
    before = L.tgpu_debug_bind_count()
    got = {}

    def run():
        got["h"] = ctx.hash_page(pkg.Page(pkg.Block(pkg.BIGINT, np.array([1, 2, 3], dtype=np.int64))), [0])

    t = threading.Thread(target=run)
    t.start()
    t.join()
    assert len(got["h"]) == 3 and L.tgpu_debug_bind_count() > before


def test_join_bridge_shared_by_driver_threads(pkg, ctx, oracle):
    """the lookup-source bridge is shared by the build driver and several probe drivers on their own threads (the reference's
    model: PartitionedLookupSourceFactory.java:146-205): probes are created / blocked / probing / closed while the build side lends
    the table; every probe sees the oracle's pairs and the build operator unblocks once the last probe is gone"""
    import threading
    import time
    rng = np.random.default_rng(5)
    bk = rng.permutation(40_000)[:15_000].astype(np.int64)
    bf = pkg.HashBuilderOperatorFactory(ctx, 2, [pkg.BIGINT], [0], [0])
    jf = pkg.LookupJoinOperatorFactory(ctx, 3, bf.lookup_source_factory, [pkg.BIGINT], [0])
    src = oracle.PagesHash([oracle.Col(oracle.BIGINT, bk)])
    errors = []

    def probe_driver(seed):
        try:
            r = np.random.default_rng(seed)
            for _ in range(5):
                pk = r.integers(0, 40_000, 30_000).astype(np.int64)
                op = jf.createOperator()
                while op.isBlocked():
                    time.sleep(0.0005)
                op.addInput(pkg.Page(pkg.Block(pkg.BIGINT, pk)))
                out = op.getOutput()
                wp, wb = src.probe([oracle.Col(oracle.BIGINT, pk)])
                host = out.to_host()
                assert np.array_equal(host.getBlock(0).values, pk[wp]) and np.array_equal(host.getBlock(1).values, bk[wb])
                out.release()
                op.finish()
                assert op.isFinished()
                op.close()
        except Exception as e:
            errors.append(repr(e))

    threads = [threading.Thread(target=probe_driver, args=(100 + i,)) for i in range(4)]
    for t in threads:
        t.start()
    b = bf.createOperator()
    time.sleep(0.01)
    b.addInput(pkg.Page(pkg.Block(pkg.BIGINT, bk)))
    b.finish()
    assert b.isBlocked()                   # probes may still be running
    for t in threads:
        t.join()
    assert not errors, errors
    jf.noMoreOperators()
    assert not b.isBlocked() and b.isFinished()
    b.close()


# ---------------------------------------------------------------------------------------------------------------------
# J7: join filter functions (M/operator/JoinHash.java:44-47,82-130) + the outer-join fixtures of T/operator/TestHashJoinOperator.java:714-1200
# ---------------------------------------------------------------------------------------------------------------------
def _golden_filter(pkg, case, n_build_channels):
    f = pkg.field
    if case["source"].startswith("T/operator/TestHashJoinOperator.java:714"):
        return f(n_build_channels + 1, pkg.BIGINT) >= 1025
    vals = case["filter_probe_values"]
    e = f(n_build_channels, pkg.VARCHAR).eq(vals[0])
    for v in vals[1:]:
        e = pkg.or_(e, f(n_build_channels, pkg.VARCHAR).eq(v))
    return e


@pytest.mark.parametrize("name", ["testProbeOuterJoinWithFilterFunction", "testOuterJoinWithNullProbe", "testOuterJoinWithNullProbeAndFilterFunction",
                                  "testOuterJoinWithNullBuild", "testOuterJoinWithNullBuildAndFilterFunction", "testOuterJoinWithNullOnBothSides",
                                  "testOuterJoinWithNullOnBothSidesAndFilterFunction"])
def test_outer_joins_and_join_filter_function_golden(pkg, ctx, name):
    case = GOLD["hash_join"][name]
    if isinstance(case["build"], str):
        T = [pkg.VARCHAR, pkg.BIGINT, pkg.BIGINT]
        bpage, ppage = pkg.Page(*blocks_of(pkg, T, sequence_page(T, 10, 20, 30, 40))), pkg.Page(*blocks_of(pkg, T, sequence_page(T, 15, 20, 1020, 2020)))
    else:
        T = [pkg.VARCHAR]
        bpage, ppage = pkg.Page(pkg.Block(pkg.VARCHAR, case["build"])), pkg.Page(pkg.Block(pkg.VARCHAR, case["probe"]))
    bf = pkg.HashBuilderOperatorFactory(ctx, 0, T, list(range(len(T))), [0])
    if case["filter"]:
        bf.lookup_source_factory.setJoinFilter(T, _golden_filter(pkg, case, len(T)))
    b = bf.createOperator()
    b.addInput(bpage)
    b.finish()
    jf = pkg.LookupJoinOperatorFactory(ctx, 1, bf.lookup_source_factory, T, [0], join_type=pkg.PROBE_OUTER)
    rows = [list(r) for pg in pkg.to_pages(jf.createOperator(), [ppage]) for r in pg.rows()]
    assert rows == case["expect_rows"]


@pytest.mark.parametrize("join_type", ["INNER", "PROBE_OUTER", "LOOKUP_OUTER", "FULL_OUTER"])
def test_joins_with_empty_lookup_source_golden(pkg, ctx, join_type):
    want = GOLD["hash_join"]["emptyLookupSource"]["expect_rows_by_join_type"][join_type]
    T = [pkg.VARCHAR]
    bf = pkg.HashBuilderOperatorFactory(ctx, 0, T, [0], [0])
    b = bf.createOperator()
    b.finish()
    jf = pkg.LookupJoinOperatorFactory(ctx, 1, bf.lookup_source_factory, T, [0], join_type=getattr(pkg, join_type))
    rows = [list(r) for pg in pkg.to_pages(jf.createOperator(), [pkg.Page(pkg.Block(pkg.VARCHAR, ["test"]))]) for r in pg.rows()]
    assert rows == want


def test_lookup_join_page_builder_scenarios(pkg, ctx):
    """T/operator/TestLookupJoinPageBuilder.java:86-156: the (probe position, build position) pair patterns of testDifferentPositions as joins"""
    G = GOLD["lookup_join_page_builder"]
    vals = np.arange(100, dtype=np.int64)
    bf = pkg.HashBuilderOperatorFactory(ctx, 0, [pkg.BIGINT], [0], [0])
    b = bf.createOperator()
    b.addInput(pkg.Page(pkg.Block(pkg.BIGINT, vals)))
    b.finish()
    jf = pkg.LookupJoinOperatorFactory(ctx, 1, bf.lookup_source_factory, [pkg.BIGINT], [0])
    probes = {"non_sequential_positions": [int(v) if v % 2 == 0 else None for v in vals], "everything": [int(v) for v in vals],
              "some_sequential_positions": [int(v) if 10 <= v < 50 else None for v in vals], "empty": [None] * 100}
    for name, probe in probes.items():
        rows = [list(r) for pg in pkg.to_pages(jf.createOperator(), [pkg.Page(pkg.Block(pkg.BIGINT, probe))]) for r in pg.rows()]
        assert rows == G[name]["expect_rows"], name


@pytest.mark.parametrize("join_type", ["INNER", "PROBE_OUTER", "LOOKUP_OUTER", "FULL_OUTER"])
def test_join_filter_function_random_vs_oracle(pkg, ctx, oracle, join_type):
    """duplicate build keys (chains), nulls on both sides, a predicate over build AND probe channels; LOOKUP_OUTER / FULL_OUTER: only
    positions that passed the filter count as visited"""
    rng = np.random.default_rng(91)
    nb, npr = 4000, 9000
    B, D = pkg.BIGINT, pkg.DOUBLE
    bkey = rand_block(pkg, rng, B, nb, 0.05, (0, 600))
    bval = rand_block(pkg, rng, B, nb, 0.1, (-50, 50))
    pkey = rand_block(pkg, rng, B, npr, 0.05, (0, 700))
    pval = rand_block(pkg, rng, B, npr, 0.1, (-50, 50))
    f = pkg.field
    # build channels 0, 1; probe channels 2, 3: build.val < probe.val AND probe.val <> 7
    filt = pkg.and_(f(1, B) < f(3, B), f(3, B).ne(7))
    bf = pkg.HashBuilderOperatorFactory(ctx, 0, [B, B], [0, 1], [0])
    bf.lookup_source_factory.setJoinFilter([B, B], filt)
    b = bf.createOperator()
    b.addInput(pkg.Page(bkey, bval))
    b.finish()
    jt = getattr(pkg, join_type)
    jf = pkg.LookupJoinOperatorFactory(ctx, 1, bf.lookup_source_factory, [B, B], [0], join_type=jt)
    got = [r for pg in pkg.to_pages(jf.createOperator(), [pkg.Page(pkey, pval)]) for r in pg.rows()]
    prog = pkg.expressions.FlatProgram(filt, [])
    ph = oracle.PagesHash([ocol(oracle, bkey)])
    outer = join_type in ("PROBE_OUTER", "FULL_OUTER")
    op, ob = oracle.probe_with_filter(ph, [ocol(oracle, pkey)], [ocol(oracle, bkey), ocol(oracle, bval)], [ocol(oracle, pkey), ocol(oracle, pval)],
                                      prog.nodes, prog.filter_root, b"", probe_outer=outer)
    pk, pv, bk, bv = pkey.to_list(), pval.to_list(), bkey.to_list(), bval.to_list()
    want = [(pk[p], pv[p]) + ((None, None) if q < 0 else (bk[q], bv[q])) for p, q in zip(op, ob)]
    assert got == want
    if join_type in ("LOOKUP_OUTER", "FULL_OUTER"):
        jf.noMoreOperators()
        of = pkg.LookupOuterOperatorFactory(ctx, 2, bf.lookup_source_factory, [B, B])
        outer_op = of.createOperator()
        assert not outer_op.isBlocked()          # every probe operator is finished and no more will be created
        o = outer_op.getOutput()                 # a source operator: its page comes without a finish() call (LookupOuterOperator.java:175-221)
        outer_rows = o.to_host().rows() if o is not None else []
        assert outer_op.isFinished()
        visited = set(int(q) for q in ob if q >= 0)
        assert outer_rows == [(None, None, bk[q], bv[q]) for q in range(nb) if q not in visited]


# ---------------------------------------------------------------------------------------------------------------------
# F7: dictionary-aware filter / projection (M/operator/project/DictionaryAwarePageFilter.java:56-110, DictionaryAwarePageProjection.java)
# ---------------------------------------------------------------------------------------------------------------------
def _run_fp_counting(pkg, ctx, pages, types, filt, projs):
    fac = pkg.FilterAndProjectOperatorFactory(ctx, 0, types, filt, projs)
    op = fac.createOperator()
    outs = []
    for pg in pages:
        op.addInput(pg)
        o = op.getOutput()
        if o is not None:
            outs.append(o.to_host())
            o.release()
    dict_pages = pkg._lib.lib().tgpu_debug_dictionary_pages(op.handle)
    op.finish()
    op.close()
    return [r for pg in outs for r in pg.rows()], dict_pages


def test_columnar_page_processor_golden(pkg, ctx):
    """T/operator/TestColumnarPageProcessor.java:46-86 testProcess / testProcessWithDictionary: the identity projection of a (BIGINT, VARCHAR)
    sequence page, flat and as dictionary blocks (20 entries, ids i % 20): one output page equal to the input"""
    n = GOLD["columnar_page_processor"]["positions"]
    f = pkg.field
    fac = pkg.FilterAndProjectOperatorFactory(ctx, 0, [pkg.BIGINT, pkg.VARCHAR], None, [f(0, pkg.BIGINT), f(1, pkg.VARCHAR)])
    flat = pkg.Page(pkg.Block(pkg.BIGINT, np.arange(n, dtype=np.int64)), pkg.Block(pkg.VARCHAR, [str(i) for i in range(n)]))
    ids = (np.arange(n) % (n // 5)).astype(np.int32)
    dic = pkg.Page(pkg.DictionaryBlock(pkg.Block(pkg.BIGINT, np.arange(n // 5, dtype=np.int64)), ids), pkg.DictionaryBlock(pkg.Block(pkg.VARCHAR, [str(i) for i in range(n // 5)]), ids))
    for page, want in ((flat, [(i, str(i)) for i in range(n)]), (dic, [(i % (n // 5), str(i % (n // 5))) for i in range(n)])):
        out = pkg.to_pages(fac.createOperator(), [page])
        assert len(out) == 1 and out[0].rows() == want


def test_dictionary_aware_filter_project_equals_flat_path(pkg, ctx, monkeypatch):
    rng = np.random.default_rng(12)
    n = 50_000
    B, V, D = pkg.BIGINT, pkg.VARCHAR, pkg.DOUBLE
    f, c = pkg.field, pkg.constant
    # dictionaries with a null entry and with DUPLICATE entries (a DictionaryBlock's dictionary need not be distinct)
    sdict = pkg.Block(V, ["AUTOMOBILE", "BUILDING", None, "FURNITURE", "BUILDING", "HOUSEHOLD", "MACHINERY"])
    sids = rng.integers(0, 7, n).astype(np.int32)
    bdict = pkg.Block(B, [5, -3, None, 2**40, 7, 5])
    bids = rng.integers(0, 6, n).astype(np.int32)
    other = pkg.Block(D, rng.standard_normal(n), (rng.random(n) < 0.1).astype(np.uint8))
    cases = [
        ([V, D], [pkg.DictionaryBlock(sdict, sids), other], f(0, V).eq("BUILDING"), [f(0, V), f(1, D), f(0, V) < c("C", V)]),
        ([V, D], [pkg.DictionaryBlock(sdict, sids), other], pkg.or_(pkg.is_null(f(0, V)), f(0, V) >= c("H", V)), [f(1, D), pkg.is_null(f(0, V))]),
        ([B, D], [pkg.DictionaryBlock(bdict, bids), other], f(0, B) > 0, [f(0, B) * 3 + 1, f(0, B), f(1, D), pkg.cast(f(0, B), D)]),
        ([B, D], [pkg.DictionaryBlock(bdict, bids), other], None, [f(0, B) % 4, f(1, D)]),
        ([B, D], [pkg.RunLengthEncodedBlock(pkg.Block(B, [42]), n), other], f(0, B).eq(42), [f(0, B) + 1, f(1, D)]),
        ([B, D], [pkg.RunLengthEncodedBlock(pkg.Block(B, [42]), n), other], f(0, B).eq(41), [f(0, B) + 1]),
        ([B, D], [pkg.RunLengthEncodedBlock(pkg.Block(B, [None]), n), other], pkg.is_null(f(0, B)), [pkg.coalesce(f(0, B), c(9, B)), f(1, D)]),
    ]
    for types, blocks, filt, projs in cases:
        page = pkg.Page(*blocks)
        got, dict_pages = _run_fp_counting(pkg, ctx, [page, page], types, filt, projs)
        assert dict_pages == 2                      # both pages took the per-dictionary-entry path
        monkeypatch.setenv("TGPU_DISABLE_DICTIONARY_AWARE", "1")
        want, flat_pages = _run_fp_counting(pkg, ctx, [page, page], types, filt, projs)
        monkeypatch.delenv("TGPU_DISABLE_DICTIONARY_AWARE")
        assert flat_pages == 0
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert all((x == y) or (isinstance(x, float) and isinstance(y, float) and np.isnan(x) and np.isnan(y)) for x, y in zip(g, w)), (g, w)


def test_dictionary_aware_path_falls_back_when_an_entry_raises(pkg, ctx):
    """DictionaryAwarePageProjection.java:139-152: a dictionary entry whose projection raises may belong to no selected row; the block
    is then processed the normal way, and the error surfaces only if a SELECTED row raises"""
    B = pkg.BIGINT
    f = pkg.field
    n = 10_000
    d = pkg.Block(B, [1, 2, 2**62, 3])
    ids = (np.arange(n) % 2).astype(np.int32)                 # only entries 0 and 1 are referenced
    sel = pkg.Block(B, np.ones(n, dtype=np.int64))
    rows, dict_pages = _run_fp_counting(pkg, ctx, [pkg.Page(pkg.DictionaryBlock(d, ids), sel)], [B, B], None, [f(0, B) * 4])
    assert dict_pages == 0 and [r[0] for r in rows[:4]] == [4, 8, 4, 8]
    ids2 = (np.arange(n) % 3).astype(np.int32)                # entry 2 (overflows) is referenced: the reference's error
    with pytest.raises(pkg.TgpuError) as e:
        _run_fp_counting(pkg, ctx, [pkg.Page(pkg.DictionaryBlock(d, ids2), sel)], [B, B], None, [f(0, B) * 4])
    assert e.value.code == -2


# ---------------------------------------------------------------------------------------------------------------------
# boundary protocol: would-block, duplicate(), memory revoke
# ---------------------------------------------------------------------------------------------------------------------
def test_would_block_duplicate_and_revoke_protocol(pkg, ctx, oracle):
    B = pkg.BIGINT
    bk = np.arange(0, 2000, 2, dtype=np.int64)
    bf = pkg.HashBuilderOperatorFactory(ctx, 0, [B], [0], [0])
    jf = pkg.LookupJoinOperatorFactory(ctx, 1, bf.lookup_source_factory, [B], [0])
    with pytest.raises(pkg.TgpuError) as e:          # HashBuilderOperator.java:150-152: "Parallel hash build cannot be duplicated"
        bf.duplicate()
    assert e.value.code == -8
    jf2 = jf.duplicate()                             # a second probe pipeline over the same join bridge
    p1, p2 = jf.createOperator(), jf2.createOperator()
    # the build side has not lent its table: no page AND blocked -> TGPU_WOULD_BLOCK (Operator.java:32-35)
    assert p1.isBlocked() and not p1.needsInput()
    assert p1.getOutput() is None and p1.last_get_output_status == 1
    b = bf.createOperator()
    b.addInput(pkg.Page(pkg.Block(B, bk)))
    b.startMemoryRevoke()                            # nothing revocable: done at once (Operator.java:53-79)
    b.finishMemoryRevoke()
    b.finish()
    assert not p1.isBlocked() and p1.needsInput()
    assert p1.getOutput() is None and p1.last_get_output_status == 0     # nothing to hand out, but not blocked either
    probe = np.arange(0, 3000, 3, dtype=np.int64)
    want_p, want_b = oracle.PagesHash([oracle.Col(B, bk)]).probe([oracle.Col(B, probe)])
    want = [(int(probe[i]), int(bk[j])) for i, j in zip(want_p, want_b)]
    for op in (p1, p2):
        assert [r for pg in pkg.to_pages(op, [pkg.Page(pkg.Block(B, probe))]) for r in pg.rows()] == want
    # the build operator stays blocked until EVERY probe factory (the duplicate too) has seen noMoreOperators
    jf.noMoreOperators()
    assert b.isBlocked()
    jf2.noMoreOperators()
    assert not b.isBlocked() and b.isFinished()
    # other factories duplicate into independent ones
    f = pkg.field
    fp = pkg.FilterAndProjectOperatorFactory(ctx, 2, [B], f(0, B) > 5, [f(0, B) * 2])
    fp2 = fp.duplicate()
    fp.noMoreOperators()
    with pytest.raises(pkg.TgpuError):
        fp.createOperator()
    out = pkg.to_pages(fp2.createOperator(), [pkg.Page(pkg.Block(B, np.arange(10, dtype=np.int64)))])
    assert out[0].getBlock(0).to_list() == [12, 14, 16, 18]


# ---------------------------------------------------------------------------------------------------------------------
# J12: the build side as P HashBuilderOperators behind a local exchange (PartitionedLookupSourceFactory.java:110-124,
# PartitionedLookupSource.java:87-153,212-262)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("join_type", ["INNER", "FULL_OUTER"])
def test_partitioned_lookup_source_factory(pkg, ctx, oracle, join_type):
    rng = np.random.default_rng(55)
    P = 4
    B, V = pkg.BIGINT, pkg.VARCHAR
    nb, npr = 6000, 20_000
    build_pages = []
    for k in range(3):                                  # several build pages, duplicate keys, nulls
        build_pages.append(pkg.Page(rand_block(pkg, rng, B, nb // 3, 0.03, (0, 900)), rand_block(pkg, rng, V, nb // 3, 0.1, (0, 50))))
    probe = pkg.Page(rand_block(pkg, rng, B, npr, 0.03, (0, 1100)))
    bf = pkg.HashBuilderOperatorFactory(ctx, 0, [B, V], [0, 1], [0], partition_count=P)
    jt = getattr(pkg, join_type)
    jf = pkg.LookupJoinOperatorFactory(ctx, 1, bf.lookup_source_factory, [B], [0], join_type=jt)
    builders = [bf.createOperator() for _ in range(P)]
    with pytest.raises(pkg.TgpuError):
        bf.createOperator()                             # one build operator per partition
    probe_op = jf.createOperator()
    # the local exchange in front of the build: LocalPartitionGenerator on the key's raw hash (TGPU_PARTITION_LOCAL)
    ex = pkg.PartitionedOutputOperatorFactory(ctx, 2, [B, V], [0], P, local=True).createOperator()
    parts = [[] for _ in range(P)]
    for pg in build_pages:
        ex.addInput(pg)
        while True:
            polled = ex.poll()
            if polled is None:
                break
            part, out = polled
            parts[part].append(out.to_host())
            builders[part].addInput(out)
            out.release()
    for k, b in enumerate(builders):
        assert probe_op.isBlocked()                     # until EVERY partition has been lent (PartitionedLookupSourceFactory.java:146-205)
        b.finish()
    assert not probe_op.isBlocked()
    # what the reference's PartitionedLookupSource yields = one PagesHash per partition; all rows of a key sit in one partition, so the pairs
    # equal those of a single table over the partition-major concatenation of the build rows
    cat_keys = [v for part in parts for pg in part for v in pg.getBlock(0).to_list()]
    cat_vals = [v for part in parts for pg in part for v in pg.getBlock(1).to_list()]
    kblock = pkg.Block(B, cat_keys)
    for part_no, part in enumerate(parts):              # the exchange routed every non-null key by the oracle's local partition function
        for pg in part:
            blk = pg.getBlock(0)
            nn = [i for i in range(blk.position_count) if not blk.isNull(i)]
            if nn:
                sub = pkg.Block(B, [blk.get(i) for i in nn])
                assert set(oracle.partition_local(oracle.hash_rows([ocol(oracle, sub)]), P).tolist()) == {part_no}
    ph = oracle.PagesHash([ocol(oracle, kblock)])
    outer = join_type == "FULL_OUTER"
    op, ob = ph.probe([ocol(oracle, probe.getBlock(0))], probe_outer=outer)
    pk = probe.getBlock(0).to_list()
    want = [(pk[p],) + ((None, None) if q < 0 else (cat_keys[q], cat_vals[q])) for p, q in zip(op, ob)]
    got = [r for pg in pkg.to_pages(probe_op, [probe]) for r in pg.rows()]
    assert got == want
    if outer:
        jf.noMoreOperators()
        oo = pkg.LookupOuterOperatorFactory(ctx, 3, bf.lookup_source_factory, [B]).createOperator()
        o = oo.getOutput()
        visited = set(int(q) for q in ob if q >= 0)
        assert o.to_host().rows() == [(None, cat_keys[q], cat_vals[q]) for q in range(len(cat_keys)) if q not in visited]   # partition by partition (:233-262)
    L = pkg._lib.lib()
    import ctypes as C
    for partition, pos in [(0, 0), (3, 12345), (1, 2**30)]:
        enc = L.tgpu_partitioned_join_position_encode(partition, pos, P)
        assert enc == (pos << 3) | partition            # shiftSize = numberOfTrailingZeros(4) + 1 = 3 (PartitionedLookupSource.java:101-102,222-226)
        a, b_ = C.c_int32(), C.c_int32()
        assert L.tgpu_partitioned_join_position_decode(enc, P, C.byref(a), C.byref(b_)) == 0 and (a.value, b_.value) == (partition, pos)


# ---------------------------------------------------------------------------------------------------------------------
# D5: output pages cut at the reference's page-full granularity (S/PageBuilder.java:126-129, PageBuilderStatus.java:49-60)
# ---------------------------------------------------------------------------------------------------------------------
def test_output_page_cuts_multi_slice_aggregation_golden(pkg):
    """T/operator/TestHashAggregationOperator.java:478-511 testMultiSliceAggregationOutput: 1.5 pages worth of groups come out as 2 pages"""
    case = GOLD["hash_aggregation"]["testMultiSliceAggregationOutput"]
    c = pkg.Context(0)
    c.set_max_output_page(max_bytes=case["max_page_size_bytes"])
    n = case["rows"]
    B = pkg.BIGINT
    page = pkg.Page(*blocks_of(pkg, [B, B], sequence_page([B, B], n, 0, 0)))
    fac = pkg.HashAggregationOperatorFactory(c, 0, [B], [1], [(pkg.COUNT_COLUMN, 0), (pkg.AVG_BIGINT, 1)], expected_groups=100_000)
    op = fac.createOperator()
    op.addInput(page)
    op.finish()
    pages = []
    while not op.isFinished():
        assert not op.needsInput()
        o = op.getOutput()
        if o is not None:
            pages.append(o.to_host())
            o.release()
    assert len(pages) == case["expect_output_pages"]
    rows = [r for pg in pages for r in pg.rows()]
    assert rows == [(i, 1, float(i)) for i in range(n)]
    # row limit: a join's output in pages of at most 8192 rows, order kept across the cuts
    c.set_max_output_page(max_rows=8192)
    bf = pkg.HashBuilderOperatorFactory(c, 1, [B], [0], [0])
    b = bf.createOperator()
    b.addInput(pkg.Page(pkg.Block(B, np.arange(30_000, dtype=np.int64))))
    b.finish()
    jf = pkg.LookupJoinOperatorFactory(c, 2, bf.lookup_source_factory, [B], [0])
    outs = pkg.to_pages(jf.createOperator(), [pkg.Page(pkg.Block(B, np.arange(0, 60_000, 2, dtype=np.int64)))])
    assert [o.position_count for o in outs] == [8192, 6808] and [r[0] for o in outs for r in o.rows()] == list(range(0, 30_000, 2))
    op.close(); b.close(); fac.close(); bf.close(); jf.close()
    c.close()


# ---------------------------------------------------------------------------------------------------------------------
# F2: ScanFilterAndProjectOperator over a page source, lazy blocks (M/operator/ScanFilterAndProjectOperator.java:232-287,354-397)
# ---------------------------------------------------------------------------------------------------------------------
def _drain_source_operator(op):
    """T/operator/OperatorAssertion.toPages for a source operator: getOutput until finished (blocked polls are just retried)"""
    pages, spins = [], 0
    while not op.isFinished() and spins < 100_000:
        spins += 1
        o = op.getOutput()
        if o is not None:
            pages.append(o.to_host())
            o.release()
    assert op.isFinished()
    return pages


def test_scan_filter_project_page_source_golden(pkg, ctx):
    # T/operator/TestScanFilterAndProjectOperator.java:98-130 testPageSource: one VARCHAR sequence page of 10 000 rows, projection field(0)
    V, B = pkg.VARCHAR, pkg.BIGINT
    f = pkg.field
    vals = sequence_values(V, 0, 10_000)
    fac = pkg.ScanFilterAndProjectOperatorFactory(ctx, 0, [V], None, [f(0, V)])
    op = fac.createOperator()
    assert not op.needsInput() and op.isBlocked()                 # waiting for its split
    assert op.getOutput() is None and op.last_get_output_status == 1
    src = pkg.PageSource([pkg.Page(pkg.Block(V, vals))], blocked_polls=2)
    op.addSplit(src)
    op.noMoreSplits()
    pages = _drain_source_operator(op)
    assert [r[0] for pg in pages for r in pg.rows()] == vals
    assert src.closed and op.stats()["processedPositions"] == 10_000
    with pytest.raises(pkg.TgpuError):
        op.addInput(pkg.Page(pkg.Block(V, ["x"])))
    # :133-179 testPageSourceMergeOutput: 4 pages of 100 rows, filter field(0) = 10, projection field(0) -> the four matching rows
    pgs = [pkg.Page(pkg.Block(B, np.arange(100, dtype=np.int64))) for _ in range(4)]
    fac2 = pkg.ScanFilterAndProjectOperatorFactory(ctx, 1, [B], f(0, B).eq(10), [f(0, B)])
    op2 = fac2.createOperator()
    op2.addSplit(pkg.PageSource(pgs))
    op2.noMoreSplits()
    assert [r[0] for pg in _drain_source_operator(op2) for r in pg.rows()] == [10, 10, 10, 10]


def test_scan_filter_project_record_cursor_golden_and_mixed_types(pkg, ctx, oracle):
    # T/operator/TestScanFilterAndProjectOperator.java:221-253 testRecordCursorSource: the split's source is a RecordCursor over the VARCHAR
    # sequence page of 10 000 rows, projection field(0): every row comes out, in order (tgpu_scan_operator_add_record_cursor)
    V, B, D, BO, DT = pkg.VARCHAR, pkg.BIGINT, pkg.DOUBLE, pkg.BOOLEAN, pkg.DATE
    f, c = pkg.field, pkg.constant
    vals = sequence_values(V, 0, 10_000)
    fac = pkg.ScanFilterAndProjectOperatorFactory(ctx, 0, [V], None, [f(0, V)])
    op = fac.createOperator()
    cur = pkg.RecordCursor([V], [(v,) for v in vals])
    op.addSplit(cur)
    op.noMoreSplits()
    pages = _drain_source_operator(op)
    assert [r[0] for pg in pages for r in pg.rows()] == vals
    assert cur.closed and op.stats()["processedPositions"] == 10_000
    # every getter, nulls, more rows than one batch of the adapter (65 536), a filter and computed projections: equal to the oracle's
    # filter / project over the same rows as one page
    rng = np.random.default_rng(223)
    n = 150_000
    rows = []
    words = ["", "a", "BUILDING", "héllo", "x" * 50]
    for i in range(n):
        rows.append((None if i % 97 == 0 else int(rng.integers(-10**12, 10**12)), None if i % 89 == 0 else float(rng.standard_normal()),
                     None if i % 83 == 0 else bool(rng.integers(0, 2)), None if i % 79 == 0 else words[int(rng.integers(0, len(words)))],
                     None if i % 73 == 0 else int(rng.integers(9000, 9400))))
    T = [B, D, BO, V, DT]
    filt = pkg.and_(f(4, DT) > c(9100, DT), pkg.not_(pkg.is_null(f(0, B))))
    projs = [f(0, B) + c(1, B), f(1, D) * c(2.0, D), f(2, BO), f(3, V), f(4, DT)]
    op = pkg.ScanFilterAndProjectOperatorFactory(ctx, 1, T, filt, projs).createOperator()
    cur = pkg.RecordCursor(T, rows)
    op.addSplit(cur)
    op.noMoreSplits()
    got = [r for pg in _drain_source_operator(op) for r in pg.rows()]
    page = pkg.Page(*[pkg.Block(t, [r[i] for r in rows]) for i, t in enumerate(T)])
    prog = pkg.expressions.FlatProgram(filt, projs)
    cols = [ocol(oracle, b) for b in page.blocks]
    pos = oracle.filter_positions(prog.nodes, prog.filter_root, bytes(prog.pool), cols)
    want_cols = []
    for root in prog.projection_roots[:3] + prog.projection_roots[4:]:
        v, nl = oracle.project(prog.nodes, root, bytes(prog.pool), cols, pos)
        want_cols.append([None if nl[i] else v[i] for i in range(len(pos))])
    assert len(got) == len(pos) and cur.closed
    assert [r[0] for r in got] == [None if x is None else int(x) for x in want_cols[0]]
    assert [None if r[1] is None else np.float64(r[1]).tobytes() for r in got] == [None if x is None else np.float64(x).tobytes() for x in want_cols[1]]
    assert [r[2] for r in got] == [None if x is None else bool(x) for x in want_cols[2]]
    assert [r[3] for r in got] == [rows[p][3] for p in pos]
    assert [r[4] for r in got] == [None if x is None else int(x) for x in want_cols[3]]


def test_scan_filter_project_lazy_blocks(pkg, ctx):
    """T/operator/project/TestPageProcessor.java:156-184 (SelectAll: filter block loaded, projection block loaded for the output),
    :219-234 testSelectNoneFilterLazyLoad (a projection-only lazy channel is NOT loaded when the filter selects nothing),
    :236-253 testProjectLazyLoad (a channel nobody reads is never loaded); T/operator/TestScanFilterAndProjectOperator.java:182-218"""
    B = pkg.BIGINT
    f = pkg.field
    n = 100
    loads = []

    def lazy(name, values):
        def load():
            loads.append(name)
            return pkg.Block(B, values)
        return pkg.LazyBlock(B, n, load)

    def run(filt, projs, blocks):
        fac = pkg.ScanFilterAndProjectOperatorFactory(ctx, 0, [B, B], filt, projs)
        op = fac.createOperator()
        op.addSplit(pkg.PageSource([pkg.Page(*blocks)]))
        op.noMoreSplits()
        pages = _drain_source_operator(op)
        return [r for pg in pages for r in pg.rows()], op.stats()

    a, b = np.arange(0, 100, dtype=np.int64), np.arange(100, 200, dtype=np.int64)
    # select all: both blocks end up loaded, the filter's first
    loads.clear()
    rows, st = run(f(0, B) >= 0, [f(0, B), f(1, B)], [lazy("filter", a), lazy("projection", b)])
    assert rows == list(zip(a.tolist(), b.tolist())) and loads == ["filter", "projection"] and st["lazyBlocksLoaded"] == 2
    # select none: the projection-only channel is never loaded ("Lazy block should not be loaded")
    loads.clear()

    def must_not_load():
        raise AssertionError("Lazy block should not be loaded")
    rows, st = run(f(0, B) < 0, [f(1, B)], [pkg.Block(B, a), pkg.LazyBlock(B, n, must_not_load)])
    assert rows == [] and st["lazyBlocksSkipped"] == 1 and st["lazyBlocksLoaded"] == 0
    # a projection that does not read channel 1: never loaded
    rows, st = run(f(0, B) >= 0, [f(0, B) + 1], [pkg.Block(B, a), pkg.LazyBlock(B, n, must_not_load)])
    assert [r[0] for r in rows] == (a + 1).tolist() and st["lazyBlocksSkipped"] == 1
    # a partial filter with a lazy projection channel: loaded because rows survive
    loads.clear()
    rows, st = run(pkg.between(f(0, B), 25, 74), [f(1, B) * 2], [pkg.Block(B, a), lazy("projection", b)])
    assert [r[0] for r in rows] == (b[25:75] * 2).tolist() and loads == ["projection"]


@pytest.mark.parametrize("speculation", [True, False])
def test_fused_aggregation_speculative_accumulate_falls_back_on_new_groups_and_errors(pkg, oracle, monkeypatch, speculation):
    """steady-state pages of the fused aggregation enqueue their accumulate launch behind the group probe, gated on the probe's counters
    (groupby.h GbhSpeculateFn).  A page that brings a NEW group closes the gate: the speculative launch does nothing and the page is
    accumulated after the insert protocol -- counts and exact sums equal the oracle's, with and without speculation; an expression
    error of the fused filter on a speculated page is raised as on any other."""
    if not speculation:
        monkeypatch.setenv("TGPU_DISABLE_SPECULATION", "1")
    rng = np.random.default_rng(61)
    n = 40_000
    T = [pkg.BIGINT, pkg.DOUBLE, pkg.BIGINT]
    f, c = pkg.field, pkg.constant

    def make(keys):
        return pkg.Page(pkg.Block(pkg.BIGINT, rng.choice(np.array(keys, dtype=np.int64), n)), pkg.Block(pkg.DOUBLE, rng.uniform(-1e6, 1e6, n)),
                        pkg.Block(pkg.BIGINT, rng.integers(1, 100, n)))
    pages = [make([7, 9]), make([7, 9]), make([9, 7]), make([7, 9, 11]), make([11, 7, 9]), make([7])]
    filt = (c(1000, pkg.BIGINT) / f(2, pkg.BIGINT)) >= 11          # selects divisor <= 90; divisor 0 raises DIVISION_BY_ZERO
    projs = [f(0, pkg.BIGINT), f(1, pkg.DOUBLE) * c(0.5, pkg.DOUBLE)]
    aggs = [(pkg.SUM_DOUBLE, 1), (pkg.COUNT_ALL, -1), (pkg.AVG_DOUBLE, 1)]
    ctx = pkg.Context(0)
    ctx.profile_enable(True)
    fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 0, T, filt, projs, [pkg.BIGINT], [0], aggs)
    rows = [r for p in pkg.to_pages(fac.createOperator(), pages) for r in p.rows()]
    prof = ctx.profile()
    # one accumulate launch per page, plus the gated one of the page that brought group 11 (page 0 meets an empty table: no speculation)
    assert prof["fused_project_accumulate_lowcard"]["count"] == len(pages) + (1 if speculation else 0)
    keys = np.concatenate([np.asarray(p.getBlock(0).to_list(), dtype=np.int64) for p in pages])
    vals = np.concatenate([np.asarray(p.getBlock(1).to_list()) for p in pages]) * 0.5
    div = np.concatenate([np.asarray(p.getBlock(2).to_list(), dtype=np.int64) for p in pages])
    sel = (1000 // div) >= 11
    og = oracle.BigintGroupByHash(100)
    gids = og.get_group_ids(ocol(oracle, pkg.Block(pkg.BIGINT, keys[sel])))
    assert [r[0] for r in rows] == [int(k) for k in og.values()[0]]      # first-seen order
    assert sorted(r[0] for r in rows) == [7, 9, 11]
    cnt, total = oracle.agg_double_sum_exact(gids, vals[sel], og.group_count)
    assert [r[2] for r in rows] == list(cnt)
    assert ulp_diff(np.array([r[1] for r in rows]), total).max() == 0
    assert ulp_diff(np.array([r[3] for r in rows]), total / cnt).max() == 0
    # the error path: a steady-state (speculated) page whose filter divides by zero on one row
    op = fac.createOperator()
    op.addInput(pages[0])
    op.addInput(pages[1])
    bad = make([7, 9])
    bad_div = np.asarray(bad.getBlock(2).to_list(), dtype=np.int64)
    bad_div[n // 2] = 0
    bad = pkg.Page(bad.getBlock(0), bad.getBlock(1), pkg.Block(pkg.BIGINT, bad_div))
    with pytest.raises(pkg.TgpuError) as e:
        op.addInput(bad)
    assert e.value.code == -7
    op.close()
    ctx.close()


@pytest.mark.parametrize("join_type", [0, 1])
def test_fused_join_gathers_fixed_width_build_channels_in_the_emit_pass(pkg, monkeypatch, join_type):
    """the fused probe's emit pass gathers the build side's fixed-width output channels itself (8 / 4 / 1 byte wide, with and without
    null vectors; unmatched PROBE_OUTER rows come out null): same rows as the unfused composition with its per-channel gathers"""
    rng = np.random.default_rng(83)
    nb, n = 30_000, 200_000
    bkeys = rng.permutation(90_000)[:nb].astype(np.int64)
    build = pkg.Page(pkg.Block(pkg.BIGINT, bkeys), rand_block(pkg, rng, pkg.BIGINT, nb, 0.1), rand_block(pkg, rng, pkg.INTEGER, nb, 0.0, (-5, 5)),
                     rand_block(pkg, rng, pkg.BOOLEAN, nb, 0.2), rand_block(pkg, rng, pkg.DOUBLE, nb, 0.0), rand_block(pkg, rng, pkg.DATE, nb, 0.3, (0, 20000)))
    BT = [pkg.BIGINT, pkg.BIGINT, pkg.INTEGER, pkg.BOOLEAN, pkg.DOUBLE, pkg.DATE]
    T = [pkg.BIGINT, pkg.DATE]
    probe = pkg.Page(rand_block(pkg, rng, pkg.BIGINT, n, 0.02, (0, 90_000)), rand_block(pkg, rng, pkg.DATE, n, 0.0, (9000, 9400)))
    f = pkg.field
    results = {}
    for mode in ("fused", "unfused"):
        if mode == "unfused":
            monkeypatch.setenv("TGPU_DISABLE_FUSION", "1")
        ctx = pkg.Context(0)
        ctx.profile_enable(True)
        for out_b in ([1, 2, 3, 4], [5, 3]):
            bf = pkg.HashBuilderOperatorFactory(ctx, 1, BT, out_b, [0])
            jf = pkg.FilterProjectLookupJoinOperatorFactory(ctx, 2, bf.lookup_source_factory, T, f(1, pkg.DATE) > 9100, [f(0, pkg.BIGINT), f(1, pkg.DATE)], [0],
                                                            probe_output_channels=[1, 0], join_type=join_type)
            b = bf.createOperator()
            b.addInput(build)
            b.finish()
            op = jf.createOperator()
            out = pkg.to_pages(op, [probe, probe])
            results[(mode, tuple(out_b))] = [r for p in out for r in p.rows()]
            op.close(); b.close()
        prof = ctx.profile()
        assert ("fused_filter_probe" in prof) == (mode == "fused")
        if mode == "fused":
            assert "join_gather" not in prof and "gather" not in prof     # no gather launches of their own
        ctx.close()
    for out_b in ((1, 2, 3, 4), (5, 3)):
        a, b = results[("fused", out_b)], results[("unfused", out_b)]
        assert len(a) > 50_000 and a == b
        if join_type == 1:
            assert any(r[2] is None and r[3] is None for r in a)       # unmatched probe rows: build channels null


def test_fused_aggregation_low_cardinality_fold_many_items_growing_groups(pkg, oracle):
    """the low-cardinality accumulate keeps per-workgroup folded partials across pages and flushes them at the end (device_agg.h
    tg_lc_fold / tg_fold_flush): 20 groups x 3 aggregates = 60 (group, aggregate) items (two fold passes), groups appearing page by
    page, 128-bit bigint sums with carries in both directions, double sums exact"""
    rng = np.random.default_rng(97)
    n = 70_000
    T = [pkg.BIGINT, pkg.BIGINT, pkg.DOUBLE]
    f = pkg.field
    pages, allk, allv, alld = [], [], [], []
    for p_i in range(5):
        k = np.repeat(rng.integers(0, 4 * (p_i + 1), n // 2), 2).astype(np.int64)   # 4, 8, ... 20 distinct keys; rows 2j, 2j + 1 share a key
        big = rng.choice(np.array([2**62, 2**63 - 1000, 2**61], dtype=np.int64), n // 2)
        v = np.empty(n, dtype=np.int64)
        v[0::2] = big + rng.integers(-100, 100, n // 2)                             # +B and -B of a pair cancel: a group's total is small, but a
        v[1::2] = -big + rng.integers(-100, 100, n // 2)                            # lane sees one sign only and runs far beyond 64 bits
        d = rng.uniform(-1e9, 1e9, n) * 10.0 ** rng.integers(-6, 7, n)
        pages.append(pkg.Page(pkg.Block(pkg.BIGINT, k), pkg.Block(pkg.BIGINT, v), pkg.Block(pkg.DOUBLE, d)))
        allk.append(k); allv.append(v); alld.append(d)
    ctx = pkg.Context(0)
    ctx.profile_enable(True)
    fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 0, T, None, [f(0, pkg.BIGINT), f(1, pkg.BIGINT), f(2, pkg.DOUBLE)], [pkg.BIGINT], [0],
                                                          [(pkg.SUM_BIGINT, 1), (pkg.COUNT_ALL, -1), (pkg.SUM_DOUBLE, 2)])
    rows = [r for p in pkg.to_pages(fac.createOperator(), pages) for r in p.rows()]
    prof = ctx.profile()
    assert prof["fused_project_accumulate_lowcard"]["count"] >= len(pages) and "agg_fold_flush" in prof
    ctx.close()
    k, v, d = np.concatenate(allk), np.concatenate(allv), np.concatenate(alld)
    og = oracle.BigintGroupByHash(100)
    gids = og.get_group_ids(ocol(oracle, pkg.Block(pkg.BIGINT, k)))
    ng = og.group_count
    assert ng == 20 and [r[0] for r in rows] == [int(x) for x in og.values()[0]]
    isum = [0] * ng
    for g, x in zip(gids.tolist(), v.tolist()):
        isum[g] += x
    cnt, dsum = oracle.agg_double_sum_exact(gids, d, ng)
    assert [r[2] for r in rows] == list(cnt)
    assert ulp_diff(np.array([r[3] for r in rows]), dsum).max() == 0
    assert all(-(2**63) <= s < 2**63 for s in isum)
    assert [r[1] for r in rows] == isum


def _drive_with_revokes(op, pages, revoke):
    """T/operator/OperatorAssertion.java:84-156 (toPagesPartial + finishOperator with revokeMemory): before every addInput and after
    every getOutput the driver revokes whatever the operator holds as revocable memory"""
    out = []

    def revoke_all():
        if revoke and op.revocableMemoryBytes() > 0:
            op.startMemoryRevoke()
            op.finishMemoryRevoke()
    for pg in pages:
        revoke_all()
        assert op.needsInput()
        op.addInput(pg)
        o = op.getOutput()
        assert o is None
    op.finish()
    for _ in range(1000):
        if op.isFinished():
            break
        o = op.getOutput()
        if o is not None:
            out.append(o.to_host())
            o.release()
        revoke_all()
    assert op.isFinished() and not op.needsInput()
    return [r for p in out for r in p.rows()]


@pytest.mark.parametrize("hash_enabled,spill_enabled,revoke", [(True, True, True), (True, True, False), (False, False, False), (False, True, True), (False, True, False)])
def test_hash_aggregation_golden_with_spill(pkg, oracle, hash_enabled, spill_enabled, revoke):
    """T/operator/TestHashAggregationOperator.java:138-220 testHashAggregation over its data provider (hashEnabled, spillEnabled,
    revokeMemoryWhenAddingPages): the same rows whatever was spilled, compared ignoring order like assertOperatorEqualsIgnoreOrder; the
    spill state is what :219 asserts (spills happen iff spilling is enabled and the driver revokes -- or, without revokes, never here:
    this library has no internal memory limit that would force one)"""
    n = GOLD["hash_aggregation"]["testHashAggregation"]["rows"]
    types = [pkg.VARCHAR, pkg.VARCHAR, pkg.VARCHAR, pkg.BIGINT, pkg.BOOLEAN]
    pages = []
    for base in (100_000, 200_000, 300_000):
        blocks = blocks_of(pkg, types, sequence_page(types, n, 100, 0, base, 0, 500))
        if hash_enabled:
            blocks = blocks + [pkg.Block(pkg.BIGINT, oracle.hash_rows([ocol(oracle, blocks[1])]))]
        pages.append(pkg.Page(*blocks))
    aggs = [(pkg.COUNT_ALL, -1), (pkg.SUM_BIGINT, 3), (pkg.AVG_BIGINT, 3), (pkg.COUNT_COLUMN, 0), (pkg.COUNT_COLUMN, 4)]
    ctx = pkg.Context(0)
    fac = pkg.HashAggregationOperatorFactory(ctx, 0, [pkg.VARCHAR], [1], aggs, hash_channel=5 if hash_enabled else -1, expected_groups=100_000,
                                             spill_enabled=spill_enabled)
    op = fac.createOperator()
    rows = _drive_with_revokes(op, pages, revoke)
    spills, spilled_bytes = op.spillStats()
    assert (spills > 0) == (spill_enabled and revoke)
    if spill_enabled and revoke:
        assert spills == 3 and spilled_bytes > n * 3 * 8     # every page's groups left HBM once (the last run when the output was built)
    op.close()
    ctx.close()
    assert len(rows) == n
    off = 2 if hash_enabled else 1
    want = sorted((str(i), 3, 3 * i, float(i), 3, 3) for i in range(n))
    assert sorted((r[0],) + tuple(r[off:]) for r in rows) == want
    if spill_enabled and revoke:
        # merged runs come out in raw-hash order (MergeHashSort)
        hk = oracle.hash_rows([oracle.Col(pkg.VARCHAR, [r[0] for r in rows])])
        assert np.all(np.diff(hk.astype(np.int64)) >= 0)


@pytest.mark.parametrize("hash_enabled,spill_enabled,revoke", [(True, True, True), (True, True, False), (False, False, False), (False, True, True), (False, True, False)])
def test_hash_builder_resize_golden(pkg, oracle, hash_enabled, spill_enabled, revoke):
    """T/operator/TestHashAggregationOperator.java:360-400 testHashBuilderResize over its data provider: a group key of 200 000 bytes (larger than
    a block builder's limit) between two pages of short keys.  The reference asserts that the operator gets through; here also the rows"""
    case = GOLD["hash_aggregation"]["testHashBuilderResize"]
    big = "\0" * case["big_value_bytes"]
    pages = []
    for keys in ([str(i) for i in range(100, 110)], [big], [str(i) for i in range(100, 110)]):
        blocks = [pkg.Block(pkg.VARCHAR, keys)]
        if hash_enabled:
            blocks.append(pkg.Block(pkg.BIGINT, oracle.hash_rows([ocol(oracle, blocks[0])])))
        pages.append(pkg.Page(*blocks))
    ctx = pkg.Context(0)
    fac = pkg.HashAggregationOperatorFactory(ctx, 0, [pkg.VARCHAR], [0], [(pkg.COUNT_COLUMN, 0)], hash_channel=1 if hash_enabled else -1, expected_groups=100_000,
                                             spill_enabled=spill_enabled)
    op = fac.createOperator()
    rows = _drive_with_revokes(op, pages, revoke)
    op.close()
    ctx.close()
    assert sorted((r[0], r[-1]) for r in rows) == sorted([(str(i), 2) for i in range(100, 110)] + [(big, 1)])


def test_hash_builder_resize_limit_and_memory_tracking_golden(pkg, ctx):
    """T/operator/TestHashAggregationOperator.java:438-476 testHashBuilderResizeLimit and :697-746 testMemoryTracking.  The limit itself is the
    engine's (a memory pool of 3 MB throws ExceededMemoryLimitException from the bytes the operator reports through its memory context,
    INTEGRATION.md); what this side owes is the report: with the 5 000 000-byte key in, the operator accounts for more than the limit.
    Memory tracking with LONG_MIN as written: > 0 after addInput, 0 once the output is drained and the operator closed"""
    case = GOLD["hash_aggregation"]["testHashBuilderResizeLimit"]
    fac = pkg.HashAggregationOperatorFactory(ctx, 0, [pkg.VARCHAR], [0], [(pkg.COUNT_COLUMN, 0)], expected_groups=100_000)
    op = fac.createOperator()
    op.addInput(pkg.Page(pkg.Block(pkg.VARCHAR, [str(i) for i in range(100, 110)])))
    assert op.memoryBytes() < case["limit_bytes"]
    op.addInput(pkg.Page(pkg.Block(pkg.VARCHAR, ["\0" * case["big_value_bytes"]])))
    assert op.memoryBytes() > case["limit_bytes"]                       # the pool's reserve() would throw here
    op.close()
    track = GOLD["hash_aggregation"]["testMemoryTracking"]
    op = pkg.HashAggregationOperatorFactory(ctx, 0, [pkg.BIGINT], [0], [(pkg.MIN_BIGINT, 0)], expected_groups=100_000).createOperator()
    assert op.needsInput()
    op.addInput(pkg.Page(pkg.Block(pkg.BIGINT, np.arange(track["rows"], dtype=np.int64))))
    assert op.memoryBytes() > 0
    op.finish()
    rows = []
    while not op.isFinished():
        o = op.getOutput()
        if o is not None:
            rows += o.to_host().rows()
            o.release()
    assert sorted(rows) == [(i, i) for i in range(track["rows"])]
    assert op.memoryBytes() == 0
    op.close()


@pytest.mark.parametrize("groups", [5, 40_000])
def test_hash_aggregation_spill_merges_exact_states(pkg, oracle, groups):
    """spilled runs carry the exact accumulator state: after any number of revokes double sums are still the correctly rounded exact sums
    (few groups); many groups: each run holds row-order sums, combined sequentially in run order like the reference's merge; bigint sums and
    counts exact, varchar + bigint keys with nulls"""
    rng = np.random.default_rng(101)
    n = 60_000
    pages, K1, K2, V, D = [], [], [], [], []
    for _ in range(4):
        k1 = rand_block(pkg, rng, pkg.VARCHAR, n, 0.02, (0, 2 if groups == 5 else 400))                  # few groups: {null, "0", "1"} x {0, 1}: the exact
        k2 = rand_block(pkg, rng, pkg.BIGINT, n, 0.0 if groups == 5 else 0.01, (0, 2 if groups == 5 else 100))   # (low-cardinality) accumulators
        v = rand_block(pkg, rng, pkg.BIGINT, n, 0.1, (-10**12, 10**12))
        d = pkg.Block(pkg.DOUBLE, rng.standard_normal(n) * 10.0 ** rng.integers(-8, 9, n), (rng.random(n) < 0.05).astype(np.uint8))
        pages.append(pkg.Page(k1, k2, v, d))
    aggs = [(pkg.COUNT_ALL, -1), (pkg.SUM_BIGINT, 2), (pkg.SUM_DOUBLE, 3), (pkg.AVG_DOUBLE, 3), (pkg.COUNT_COLUMN, 3)]
    results = {}
    for spill in (False, True):
        ctx = pkg.Context(0)
        fac = pkg.HashAggregationOperatorFactory(ctx, 0, [pkg.VARCHAR, pkg.BIGINT], [0, 1], aggs, expected_groups=1000, spill_enabled=spill)
        op = fac.createOperator()
        rows = _drive_with_revokes(op, pages, spill)
        assert (op.spillStats()[0] == 4) == spill
        results[spill] = {(r[0], r[1]): r[2:] for r in rows}
        assert len(results[spill]) == len(rows)
        op.close()
        ctx.close()
    a, b = results[False], results[True]
    assert a.keys() == b.keys() and (len(a) == 6 if groups == 5 else len(a) > 20_000)
    for key in a:
        ra, rb = a[key], b[key]
        assert ra[0] == rb[0] and ra[1] == rb[1] and ra[4] == rb[4]
    sa = np.array([[np.nan if x is None else x for x in a[k][2:4]] for k in a])
    sb = np.array([[np.nan if x is None else x for x in b[k][2:4]] for k in a])
    if groups == 5:
        assert ulp_diff(sa[:, 0], sb[:, 0]).max() == 0 and ulp_diff(sa[:, 1], sb[:, 1]).max() == 0       # exact either way
    else:
        # many groups: every run (= one page here) holds row-order (Java-order) sums, and the runs are merged the reference's way --
        # SpillableHashAggregationBuilder.mergeFromDisk (:229-240) feeds their intermediate states to addIntermediate = combine
        # (DoubleSumAggregation.java:48-52: state = state + other), run after run: total = ((run0 + run1) + run2) + run3 in doubles.
        og = oracle.MultiChannelGroupByHash([pkg.VARCHAR, pkg.BIGINT], 1000)
        per_page, key_of = [], {}
        for pg in pages:
            gids = og.get_group_ids([ocol(oracle, pg.getBlock(0)), ocol(oracle, pg.getBlock(1))])
            per_page.append((gids, pg.getBlock(3)))
            first = np.unique(gids, return_index=True)
            for g, row in zip(first[0].tolist(), first[1].tolist()):
                key_of.setdefault(g, (pg.getBlock(0).get(row), pg.getBlock(1).get(row)))
        ng = og.group_count
        want, seen = np.zeros(ng), np.zeros(ng, dtype=bool)
        for gids, blk in per_page:
            cnt, js = oracle.agg_double_sum(gids, blk.values, ng, nulls=blk.nulls)      # the run's Java-order sums
            in_run = np.zeros(ng, dtype=bool)
            in_run[np.unique(gids)] = True                                             # a group the run holds combines even with count 0 (sum 0.0)
            want = np.where(in_run, want + np.where(cnt > 0, js, 0.0), want)
            seen |= cnt > 0
        got = np.array([np.nan if b[key_of[i]][2] is None else b[key_of[i]][2] for i in range(ng)])
        has = ~np.isnan(got)
        assert has.sum() > 20_000 and ulp_diff(got[has], want[has]).max() == 0


def test_spill_fused_aggregation_final_step_and_unsupported_factories(pkg, ctx, oracle):
    """the fused filter / project + aggregation operator spills like the plain one (same result as without revokes); a FINAL aggregation
    spills its merged intermediate states; PARTIAL steps and other operators hold nothing revocable"""
    rng = np.random.default_rng(107)
    n = 50_000
    T = [pkg.BIGINT, pkg.DOUBLE, pkg.DATE]
    f, c = pkg.field, pkg.constant
    pages = [pkg.Page(rand_block(pkg, rng, pkg.BIGINT, n, 0.0, (0, 5)), rand_block(pkg, rng, pkg.DOUBLE, n, 0.02), rand_block(pkg, rng, pkg.DATE, n, 0.0, (9000, 9400))) for _ in range(3)]
    aggs = [(pkg.SUM_DOUBLE, 1), (pkg.COUNT_ALL, -1), (pkg.AVG_DOUBLE, 1)]
    got = {}
    for spill in (False, True):
        fac = pkg.FilterProjectHashAggregationOperatorFactory(ctx, 0, T, f(2, pkg.DATE) > 9100, [f(0, pkg.BIGINT), f(1, pkg.DOUBLE) * c(3.0, pkg.DOUBLE)], [pkg.BIGINT], [0], aggs)
        if spill:
            fac.setSpillEnabled(True)
        op = fac.createOperator()
        got[spill] = sorted(_drive_with_revokes(op, pages, spill))
        assert (op.spillStats()[0] > 0) == spill
        op.close()
    assert len(got[True]) == 5 and got[True] == got[False]          # exact accumulators: bit-identical with and without spills
    # PARTIAL -> FINAL with the FINAL step spilling
    part = pkg.HashAggregationOperatorFactory(ctx, 1, [pkg.BIGINT], [0], [(pkg.SUM_DOUBLE, 1), (pkg.COUNT_ALL, -1)], step=pkg.PARTIAL, spill_enabled=True)
    p_op = part.createOperator()
    assert p_op.revocableMemoryBytes() == 0
    inter = []
    for pg in pages:
        p_op.addInput(pg)
        assert p_op.revocableMemoryBytes() == 0                      # output-partial steps get the in-memory builder (HashAggregationOperator.java:390)
        p_op.startMemoryRevoke()
    p_op.finish()
    while not p_op.isFinished():
        o = p_op.getOutput()
        if o is not None:
            inter.append(o.to_host())
    assert p_op.spillStats() == (0, 0)
    fin = {}
    for spill in (False, True):
        ff = pkg.HashAggregationOperatorFactory(ctx, 2, [pkg.BIGINT], [0], [(pkg.SUM_DOUBLE, 1), (pkg.COUNT_ALL, 3)], step=pkg.FINAL, spill_enabled=spill)
        op = ff.createOperator()
        fin[spill] = sorted(_drive_with_revokes(op, inter + inter, spill))
        assert (op.spillStats()[0] > 0) == spill
        op.close()
    assert len(fin[True]) == 5 and [r[0] for r in fin[True]] == [r[0] for r in fin[False]] and [r[2] for r in fin[True]] == [r[2] for r in fin[False]]
    assert ulp_diff(np.array([r[1] for r in fin[True]]), np.array([r[1] for r in fin[False]])).max() == 0
    with pytest.raises(pkg.TgpuError) as e:
        pkg.FilterAndProjectOperatorFactory(ctx, 3, T, None, [f(0, pkg.BIGINT)]).setSpillEnabled(True)
    assert e.value.code == -8


@pytest.mark.parametrize("hash_enabled", [True, False])
def test_multiple_partial_flushes_golden(pkg, ctx, oracle, hash_enabled):
    """T/operator/TestHashAggregationOperator.java:512-591 testMultiplePartialFlushes (LONG_MIN as written):
    a PARTIAL aggregation with a 1 kB limit fills up, stops taking input, drains, and takes input again"""
    case = GOLD["hash_aggregation"]["testMultiplePartialFlushes"]
    pages = []
    for start in (0, 500, 1000, 1500):
        keys = np.arange(start, start + 500, dtype=np.int64)
        blocks = [pkg.Block(pkg.BIGINT, keys)]
        if hash_enabled:
            blocks.append(pkg.Block(pkg.BIGINT, oracle.hash_rows([oracle.Col(pkg.BIGINT, keys)])))
        pages.append(pkg.Page(*blocks))
    fac = pkg.HashAggregationOperatorFactory(ctx, 0, [pkg.BIGINT], [0], [(pkg.MIN_BIGINT, 0)], step=pkg.PARTIAL, hash_channel=1 if hash_enabled else -1, expected_groups=100_000)
    fac.setMaxPartialMemory(case["max_partial_memory_bytes"])
    op = fac.createOperator()
    it = iter(pages)
    fed = 0
    while op.needsInput():          # fill up the aggregation (:547-550)
        pg = next(it, None)
        if pg is None:
            break
        op.addInput(pg)
        fed += 1
    assert 0 < fed < len(pages)
    out = []
    while True:                     # drain the output (partial flush) :560-568
        o = op.getOutput()
        if o is None:
            break
        out.append(o.to_host())
    assert out and op.needsInput()  # :570-574
    for pg in it:
        while not op.needsInput():
            o = op.getOutput()
            assert o is not None
            out.append(o.to_host())
        op.addInput(pg)
    op.finish()
    while not op.isFinished():
        o = op.getOutput()
        if o is not None:
            out.append(o.to_host())
    rows = [r for p in out for r in p.rows()]
    # PARTIAL output of min(bigint): (key, [hash], count, value) -- the reference's NullableLongState is the value itself; compare keys and values
    assert sorted((r[0], r[-1]) for r in rows) == [(i, i) for i in range(case["rows"])]
    assert op.memoryBytes() >= 0
    op.close()


def test_merge_with_memory_spill_golden(pkg, oracle):
    """T/operator/TestHashAggregationOperator.java:594-634 testMergeWithMemorySpill (LONG_MIN as written): 150 000 groups
    spilled by the driver's revokes, 10 more in memory, merged when the output is built"""
    case = GOLD["hash_aggregation"]["testMergeWithMemorySpill"]
    ctx = pkg.Context(0)
    pages = [pkg.Page(pkg.Block(pkg.BIGINT, np.arange(0, 150_000, dtype=np.int64))), pkg.Page(pkg.Block(pkg.BIGINT, np.arange(150_000, 150_010, dtype=np.int64)))]
    fac = pkg.HashAggregationOperatorFactory(ctx, 0, [pkg.BIGINT], [0], [(pkg.MIN_BIGINT, 0)], expected_groups=1, spill_enabled=True)
    op = fac.createOperator()
    rows = _drive_with_revokes(op, pages, True)
    assert op.spillStats()[0] >= 1
    assert sorted(rows) == [(i, i) for i in range(case["rows"])]
    op.close()
    ctx.close()


def _dfs_cases():
    import dfs_fixtures
    return dfs_fixtures.cases(GOLD)


@pytest.mark.parametrize("name", [c[0] for c in _dfs_cases()])
def test_dynamic_filter_source_golden(pkg, ctx, name):
    """T/operator/TestDynamicFilterSourceOperator.java through the GPU operator: pages pass through unchanged (verifyPassthrough) and every
    filter channel's domain is the reference's (value sets as sets; the two VARCHAR size-limit cases answer ALL, the documented superset)"""
    import dfs_fixtures as F
    _, types, channels, params, ops = next(c for c in _dfs_cases() if c[0] == name)
    fac = pkg.DynamicFilterSourceOperatorFactory(ctx, 90, types, channels, *params)
    for pages, expect in ops:
        op = fac.createOperator()
        for pg in pages:
            page = pkg.Page(*[pkg.Block(t, F.column_values(t, desc)) for t, desc in zip(types, pg)])
            assert op.needsInput()
            op.addInput(page)
            out = op.getOutput()
            norm = lambda rows: [tuple("nan" if isinstance(v, float) and v != v else v for v in r) for r in rows]
            assert out is not None and norm(out.to_host().rows()) == norm(page.rows())
            out.release()
        op.finish()
        assert op.isFinished()
        got = [F.normalise(op.domain(k), types[channels[k]]) for k in range(len(channels))]
        got = [("none",) if g == ("values", []) else g for g in got]     # an empty value set IS Domain.none() (empty build side, only nulls)
        assert got == [F.expected(e) for e in expect]
        op.close()
    fac.close()


@pytest.mark.parametrize("name", list(GOLD["partitioned_output"]["cases"]))
def test_partitioned_output_operator_golden(pkg, ctx, oracle, name):
    """T/operator/TestPartitionedOutputOperator.java:98-181: flat, dictionary and run-length input pages through the LocalPartitionGenerator
    partitioning into 512 partitions, without and with replication (an all-null channel 0 sends every row to every partition):
    OutputPositions as the reference asserts them; which rows land where is checked against the oracle's partition function"""
    g = GOLD["partitioned_output"]
    case = g["cases"][name]
    n, parts, page_count = 1000, 512, 10
    seq = np.arange(n, dtype=np.int64)
    if case["block"] == "TESTING_BLOCK":
        blk, vals = pkg.Block(pkg.BIGINT, seq), seq
    elif case["block"] == "TESTING_DICTIONARY_BLOCK":
        ids = (np.arange(n) % 200).astype(np.int32)
        blk, vals = pkg.DictionaryBlock(pkg.Block(pkg.BIGINT, np.arange(200, dtype=np.int64)), ids), ids.astype(np.int64)
    else:
        blk, vals = pkg.RunLengthEncodedBlock(pkg.Block(pkg.BIGINT, [g["rle_value"]]), n), np.full(n, g["rle_value"], dtype=np.int64)
    if case["replicate"]:
        types = [pkg.BIGINT, pkg.BIGINT]
        page = pkg.Page(pkg.RunLengthEncodedBlock(pkg.Block(pkg.BIGINT, [None]), n), blk)
        fac = pkg.PartitionedOutputOperatorFactory(ctx, 95, types, [0], parts, null_channel=0, local=True)
    else:
        types = [pkg.BIGINT]
        page = pkg.Page(blk)
        fac = pkg.PartitionedOutputOperatorFactory(ctx, 95, types, [0], parts, local=True)
    op = fac.createOperator()
    want_pid = oracle.partition_local(oracle.hash_rows([oracle.Col(pkg.BIGINT, vals)]), parts)
    per_partition = {}
    for _ in range(page_count):
        assert op.needsInput()
        op.addInput(page)
        while True:
            e = op.poll()
            if e is None:
                break
            part, out = e
            h = out.to_host()
            per_partition.setdefault(part, []).append(np.asarray(h.blocks[-1].values[:h.position_count]))
            if case["replicate"]:
                assert h.blocks[0].nulls is not None and h.blocks[0].nulls[:h.position_count].all()
            out.release()
    op.finish()
    info = op.info()
    assert info["rowsAdded"] == case["output_positions"]
    if "pages_added" in case:
        assert info["pagesAdded"] == case["pages_added"]
    for part, chunks in per_partition.items():
        got = np.concatenate(chunks)
        want = np.tile(vals if case["replicate"] else vals[want_pid == part], page_count)
        assert np.array_equal(got, want), part
    assert sum(len(np.concatenate(c)) for c in per_partition.values()) == case["output_positions"]
    op.close()
    fac.close()
