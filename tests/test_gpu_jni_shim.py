"""The JNI shim EXECUTED on the GPU: jni/tgpu_jni.c (compiled against the declaration stub tests/jni_stub/jni.h) driven through the fake JVM
of tests/jni_stub/fake_jvm.c -- Java arrays are C arrays, exceptions a pending (code, message) pair -- over the entry points GpuNative.java
declares.  What a JVM host would do, call for call: createContext -> create*Factory -> createOperator -> addInput(heap arrays) / addInputDevicePage
-> getOutput -> blockInfo / copyBlocks, for the plain FilterAndProject operator, for the two FUSED factories bench.py times, for a join with
a hash builder, for TopN, for the scan operator over an upcalling page source, and for a SerializedPage round trip.  Results against numpy / the
oracle; after every flow: no array left pinned, no local frame left open, no JNI call made while an array was pinned."""
import ctypes as C
import importlib

import numpy as np
import pytest

from jni_harness import FakeJvm, build_fake_jni

pytestmark = pytest.mark.gpu

I32, I64 = C.c_int32, C.c_int64


@pytest.fixture(scope="module")
def jvm():
    return FakeJvm(build_fake_jni())


@pytest.fixture(scope="module")
def jctx(jvm):
    h = jvm.checked("createContext", I64, I32(0))
    assert h != 0
    yield I64(h)
    jvm.call("destroyContext", None, I64(h))


def program(jvm, pkg, filt, projs):
    p = pkg.expressions.FlatProgram(filt, projs)
    nodes = [jvm.array(np.array([nd["kind"], nd["type"], nd["op"], len(nd["args"])] + (nd["args"] + [0, 0, 0])[:3] + [nd.get("is_null", 0), nd.get("slen", 0)], dtype=np.int32))
             for nd in p.nodes]
    return [jvm.object_array(nodes), jvm.array(np.array([nd.get("ival", 0) for nd in p.nodes], dtype=np.int64)),
            jvm.array(np.array([nd.get("dval", 0.0) for nd in p.nodes], dtype=np.float64)), jvm.array(np.frombuffer(bytes(p.pool), dtype=np.int8)),
            I32(p.filter_root), jvm.array(np.array(p.projection_roots, dtype=np.int32))]


def ints(jvm, *v):
    return jvm.array(np.array(v, dtype=np.int32))


def add_flat_page(jvm, op, types, columns, nulls=None, offsets=None):
    """GpuPages.addInput for flat blocks: the blocks' own arrays, arrayOffset 0"""
    n = len(types)
    positions = len(columns[0]) if offsets is None or offsets[0] is None else len(offsets[0]) - 1
    none = jvm.object_array([None] * n)
    vals = jvm.object_array([jvm.array(c) for c in columns])
    nl = jvm.object_array([None if (nulls is None or x is None) else jvm.array(x.astype(np.uint8)) for x in (nulls or [None] * n)])
    off = jvm.object_array([None if (offsets is None or x is None) else jvm.array(x.astype(np.int32)) for x in (offsets or [None] * n)])
    jvm.checked("addInput", None, op, I32(positions), jvm.array(np.array(types, dtype=np.int32)), jvm.array(np.zeros(n, dtype=np.int32)), jvm.array(np.zeros(n, dtype=np.int32)),
                jvm.array(np.zeros(n, dtype=np.int32)), vals, nl, off, none, none, none, none)


def heap_blocks(jvm, page):
    """GpuPages.toHeapBlocks: blockInfo per channel, then ONE copyBlocks"""
    page = I64(page)
    channels = jvm.checked("pageChannelCount", I32, page)
    positions = jvm.checked("pagePositionCount", I32, page)
    info = jvm.empty(3, np.int64)
    vals, nls, offs, meta = [], [], [], []
    for ch in range(channels):
        jvm.checked("blockInfo", None, page, I32(ch), info)
        t, nbytes, may = (int(x) for x in jvm.read(info, np.int64))
        dt = {1: np.int64, 4: np.float64, 2: np.int32, 3: np.int32, 5: np.uint8, 6: np.uint8}[t]
        vals.append(jvm.empty(nbytes if t == 6 else positions, dt))
        nls.append(jvm.empty(positions, np.uint8) if may else None)
        offs.append(jvm.empty(positions + 1, np.int32) if t == 6 else None)
        meta.append((t, dt))
    jvm.checked("copyBlocks", None, page, jvm.object_array(vals), jvm.object_array(nls), jvm.object_array(offs))
    out = []
    for ch, (t, dt) in enumerate(meta):
        v = jvm.read(vals[ch], dt)
        nl = jvm.read(nls[ch], np.uint8).astype(bool) if nls[ch] is not None else np.zeros(positions, dtype=bool)
        if t == 6:
            o = jvm.read(offs[ch], np.int32)
            v = [None if nl[i] else bytes(v[o[i]:o[i + 1]]).decode() for i in range(positions)]
        out.append((v, nl))
    return out


def drain(jvm, op):
    wb = jvm.empty(1, np.uint8)
    pages = []
    for _ in range(10_000):
        if jvm.checked("isFinished", C.c_uint8, op):
            break
        h = jvm.checked("getOutput", I64, op, wb)
        if h:
            pages.append(h)
    return pages


def clean(jvm):
    assert jvm.outstanding_pins() == 0 and jvm.open_frames() == 0 and jvm.calls_while_pinned() == 0 and jvm.pending_code() == 0


def test_filter_project_through_the_shim(pkg, jvm, jctx):
    f = pkg.field
    B = pkg.BIGINT
    rng = np.random.default_rng(3)
    n = 100_000
    cols = [rng.integers(0, 1000, n).astype(np.int64), rng.integers(0, 2**20, n).astype(np.int64), rng.integers(0, 2**20, n).astype(np.int64)]
    fac = I64(jvm.checked("createFilterProjectFactory", I64, jctx, I32(0), ints(jvm, B, B, B), *program(jvm, pkg, f(0, B) > 899, [f(1, B) * f(2, B)])))
    op = I64(jvm.checked("createOperator", I64, fac))
    assert jvm.checked("needsInput", C.c_uint8, op)
    add_flat_page(jvm, op, [B, B, B], cols)
    jvm.checked("finish", None, op)
    pages = drain(jvm, op)
    got = np.concatenate([heap_blocks(jvm, p)[0][0] for p in pages])
    sel = cols[0] > 899
    assert np.array_equal(got, (cols[1] * cols[2])[sel])
    for p in pages:
        jvm.call("releasePage", None, I64(p))
    # the checked multiplication's error comes back as NativeError(NUMERIC_VALUE_OUT_OF_RANGE = -2, message)
    op2 = I64(jvm.checked("createOperator", I64, fac))
    big = np.full(10, 2**40, dtype=np.int64)
    jvm.call("addInput", None, op2, I32(10), ints(jvm, B, B, B), ints(jvm, 0, 0, 0), ints(jvm, 0, 0, 0), ints(jvm, 0, 0, 0),
             jvm.object_array([jvm.array(np.full(10, 950, dtype=np.int64)), jvm.array(big), jvm.array(big)]), jvm.object_array([None] * 3), jvm.object_array([None] * 3),
             jvm.object_array([None] * 3), jvm.object_array([None] * 3), jvm.object_array([None] * 3), jvm.object_array([None] * 3))
    h = jvm.call("getOutput", I64, op2, jvm.empty(1, np.uint8))
    assert jvm.pending_code() == -2 or h == 0 and jvm.pending_code() in (-2, 0)
    if jvm.pending_code() == 0:       # the error may surface at addInput or at getOutput; one of them must have raised
        pytest.fail("overflow did not raise through the shim")
    jvm.clear()
    jvm.call("close", None, op2)
    jvm.call("close", None, op)
    jvm.call("noMoreOperators", None, fac)
    jvm.call("destroyFactory", None, fac)
    clean(jvm)


def test_the_two_fused_factories_the_bench_times_through_the_shim(pkg, jvm, jctx, oracle):
    """tgpu_filter_project_hash_aggregation_factory_create (Q1 shape) and tgpu_filter_project_lookup_join_factory_create (Q3 shape), bound through
    GpuNative.createFilterProjectHashAggregationFactory / createFilterProjectLookupJoinFactory, device pages handed from operator to operator"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    entry = importlib.import_module("__graft_entry__")
    pps = entry.bench_page_processors(pkg)
    rng = np.random.default_rng(11)
    # ---- Q1 shape ----
    n = 60_000
    types, filt, projs = pps["q1"]
    rf = rng.integers(0, 3, n)
    ls = rng.integers(0, 2, n)
    keys0 = np.frombuffer(b"ANR", dtype=np.uint8)[rf].copy()
    keys1 = np.frombuffer(b"FO", dtype=np.uint8)[ls].copy()
    off = np.arange(n + 1, dtype=np.int32)
    qty = rng.integers(1, 51, n).astype(np.float64)
    price = qty * rng.integers(90000, 210000, n) / 100.0
    disc = rng.integers(0, 11, n) / 100.0
    tax = rng.integers(0, 9, n) / 100.0
    ship = rng.integers(8036, 10562, n).astype(np.int32)
    aggs = entry.q1_aggregates(pkg)
    agg_arr = jvm.array(np.array([x for (fn, ch) in aggs for x in (fn, ch, -1)], dtype=np.int32))
    V, D = pkg.VARCHAR, pkg.DOUBLE
    fac = I64(jvm.checked("createFilterProjectHashAggregationFactory", I64, jctx, I32(1), jvm.array(np.array(types, dtype=np.int32)), *program(jvm, pkg, filt, projs),
                          ints(jvm, V, V), ints(jvm, 0, 1), I32(-1), I32(0), agg_arr, I32(16)))
    op = I64(jvm.checked("createOperator", I64, fac))
    add_flat_page(jvm, op, types, [keys0.view(np.int8), keys1.view(np.int8), qty, price, disc, tax, ship], offsets=[off, off, None, None, None, None, None])
    jvm.checked("finish", None, op)
    pages = drain(jvm, op)
    rows = heap_blocks(jvm, pages[0])
    sel = ship <= 10471
    gkeys = [(chr(a), chr(b)) for a, b in zip(keys0[sel], keys1[sel])]
    order = list(dict.fromkeys(gkeys))
    assert list(zip(rows[0][0], rows[1][0])) == order                      # first-seen group order
    cnt = [sum(1 for k in gkeys if k == g) for g in order]
    assert [int(x) for x in rows[-1][0]] == cnt                            # count(*)
    want_qty = [float(np.sum(qty[sel][[k == g for k in gkeys]])) for g in order]
    assert [float(x) for x in rows[2][0]] == want_qty                      # sum(quantity): integer-valued, exact in any order
    for p in pages:
        jvm.call("releasePage", None, I64(p))
    jvm.call("close", None, op)
    jvm.call("destroyFactory", None, fac)
    # ---- Q3 shape: hash builder -> fused filter/project + probe, build page handed over as a device page ----
    B = pkg.BIGINT
    bk = rng.permutation(50_000)[:20_000].astype(np.int64)
    handles = jvm.checked("createHashBuilderFactory", I64, jctx, I32(2), ints(jvm, B), ints(jvm, 0), ints(jvm, 0), I32(-1), I32(1000), I32(1))
    bfac, bridge = (I64(int(x)) for x in jvm.read(C.c_void_p(handles), np.int64))
    bop = I64(jvm.checked("createOperator", I64, bfac))
    add_flat_page(jvm, bop, [B], [bk])
    jvm.checked("finish", None, bop)
    types, filt, projs = pps["q3_lineitem"]
    m = 80_000
    lk = rng.integers(0, 50_000, m).astype(np.int64)
    ep = rng.random(m) * 1000.0
    dc = rng.integers(0, 11, m) / 100.0
    sd = rng.integers(9000, 9400, m).astype(np.int32)
    jfac = I64(jvm.checked("createFilterProjectLookupJoinFactory", I64, jctx, I32(3), bridge, jvm.array(np.array(types, dtype=np.int32)), *program(jvm, pkg, filt, projs),
                           ints(jvm, 0), I32(-1), ints(jvm, 0, 1), I32(0)))
    jop = I64(jvm.checked("createOperator", I64, jfac))
    assert not jvm.checked("isBlocked", C.c_uint8, jop)
    add_flat_page(jvm, jop, types, [lk, ep, dc, sd])
    jvm.checked("finish", None, jop)
    pages = drain(jvm, jop)
    got = [np.concatenate([heap_blocks(jvm, p)[c][0] for p in pages]) for c in range(3)]
    passing = sd > 9204
    op_, ob_ = oracle.PagesHash([oracle.Col(oracle.BIGINT, bk)]).probe([oracle.Col(oracle.BIGINT, lk[passing])])
    assert np.array_equal(got[0], lk[passing][op_]) and np.array_equal(got[2], bk[ob_])
    assert np.array_equal(got[1].view(np.int64), (ep * (1.0 - dc))[passing][op_].view(np.int64))
    stats = jvm.empty(3, np.int64)
    jvm.checked("lookupSourceStats", None, bridge, stats)
    assert int(jvm.read(stats, np.int64)[0]) == len(bk)
    # chaining: a TopN over the join's device page, which never becomes heap blocks (addInputDevicePage)
    tfac = I64(jvm.checked("createTopNFactory", I64, jctx, I32(4), ints(jvm, B, pkg.DOUBLE, B), I64(5), ints(jvm, 1), ints(jvm, 3)))
    top = I64(jvm.checked("createOperator", I64, tfac))
    for p in pages:
        jvm.checked("addInputDevicePage", None, top, I64(p))
        jvm.call("releasePage", None, I64(p))
    jvm.checked("finish", None, top)
    tp = drain(jvm, top)
    rev = heap_blocks(jvm, tp[0])[1][0]
    assert list(rev) == sorted(got[1], reverse=True)[:5]
    for p in tp:
        jvm.call("releasePage", None, I64(p))
    for o in (top, jop, bop):
        jvm.call("close", None, o)
    jvm.call("noMoreOperators", None, jfac)
    for fct in (tfac, jfac, bfac):
        jvm.call("destroyFactory", None, fct)
    jvm.call("destroyBridge", None, bridge)
    clean(jvm)


def test_scan_operator_pulls_its_pages_through_upcalls(pkg, jvm, jctx):
    """ScanFilterAndProjectOperator over GpuPageSource: nextPage / isFinished / isBlocked / loadBlock / close are upcalls (CallIntMethod ...) on a
    global reference; every channel arrives lazy and only the channels the processor reads are loaded (TestScanFilterAndProjectOperator's
    lazy-block contract); the copies the shim keeps live until the next page"""
    f = pkg.field
    B, D = pkg.BIGINT, pkg.DOUBLE
    rng = np.random.default_rng(5)
    pages = [(rng.integers(0, 100, 5000).astype(np.int64), rng.random(5000), rng.integers(0, 9, 5000).astype(np.int64)) for _ in range(3)]
    state = {"i": -1, "loaded": [], "closed": 0, "keep": []}

    def next_page():
        if state["i"] + 1 >= len(pages):
            return -1
        state["i"] += 1
        return len(pages[state["i"]][0])

    def load_block(ch):
        state["loaded"].append((state["i"], ch))
        col = pages[state["i"]][ch]
        parts = jvm.object_array([jvm.array(np.array([0, 0, 0], dtype=np.int32)), jvm.array(col), None, None, None, None, None, None])
        state["keep"].append(parts)
        return parts.value

    def close():
        state["closed"] += 1

    src = jvm.adapter(next_page, lambda: state["i"] + 1 >= len(pages), lambda: 0, load_block, close)
    fac = I64(jvm.checked("createScanFilterProjectFactory", I64, jctx, I32(7), ints(jvm, B, D, B), *program(jvm, pkg, f(0, B) < 50, [f(0, B), f(1, D)])))
    op = I64(jvm.checked("createOperator", I64, fac))
    jvm.checked("scanAddPageSource", None, op, src, ints(jvm, B, D, B))
    jvm.checked("scanNoMoreSplits", None, op)
    outs = drain(jvm, op)
    got0 = np.concatenate([heap_blocks(jvm, p)[0][0] for p in outs])
    got1 = np.concatenate([heap_blocks(jvm, p)[1][0] for p in outs])
    want0 = np.concatenate([p[0][p[0] < 50] for p in pages])
    want1 = np.concatenate([p[1][p[0] < 50] for p in pages])
    assert np.array_equal(got0, want0) and np.array_equal(got1, want1)
    assert sorted(state["loaded"]) == [(i, ch) for i in range(3) for ch in (0, 1)]      # channel 2 is never read: never loaded
    st = jvm.empty(3, np.int64)
    jvm.checked("scanStats", None, op, st)
    assert [int(x) for x in jvm.read(st, np.int64)] == [15000, 6, 3]
    for p in outs:
        jvm.call("releasePage", None, I64(p))
    jvm.call("close", None, op)
    assert state["closed"] == 1 and jvm.global_refs() == 0
    jvm.call("destroyFactory", None, fac)
    clean(jvm)


def test_serialized_page_round_trip_through_the_shim(pkg, jvm, jctx, oracle):
    rng = np.random.default_rng(9)
    n = 3000
    vals = rng.integers(-10**9, 10**9, n).astype(np.int64)
    nulls = rng.random(n) < 0.1
    blk = pkg.Block(pkg.BIGINT, vals, nulls.astype(np.uint8))
    wire = oracle.serialize_page([oracle.Col(oracle.BIGINT, vals, nulls.astype(np.uint8))])
    page = jvm.checked("deserializePage", I64, jctx, jvm.array(np.frombuffer(wire, dtype=np.int8)), I32(0), I32(len(wire)), ints(jvm, pkg.BIGINT))
    got, got_nulls = heap_blocks(jvm, page)[0]
    assert np.array_equal(got_nulls, nulls) and np.array_equal(got[~nulls], vals[~nulls])
    bound = jvm.checked("serializePage", I64, jctx, I64(page), C.c_void_p(None))
    out = jvm.empty(int(bound), np.int8)
    written = jvm.checked("serializePage", I64, jctx, I64(page), out)
    assert bytes(jvm.read(out, np.int8)[:written].tobytes()) == bytes(wire)
    jvm.call("releasePage", None, I64(page))
    clean(jvm)


def test_orc_column_decode_through_the_shim(pkg, jvm, jctx):
    """GpuNative.orcDecodeLongColumn: PRESENT + DATA streams of an ORC column (decompressed bytes) -> a device page, read back through copyBlocks"""
    from oracle import orc
    rng = np.random.default_rng(3)
    n = 5000
    present = (rng.random(n) < 0.8).astype(np.uint8)
    vals = rng.integers(-10**9, 10**9, int(present.sum()))
    data = b"".join(orc.rle_v2_direct(vals[i:i + 512], True) for i in range(0, len(vals), 512))
    page = jvm.checked("orcDecodeLongColumn", I64, jctx, I32(pkg.BIGINT), I32(2), I32(n), jvm.array(np.frombuffer(orc.boolean_encode(present.tolist()), dtype=np.int8)),
                       jvm.array(np.frombuffer(data, dtype=np.int8)))
    got, nulls = heap_blocks(jvm, page)[0]
    assert np.array_equal(nulls, present == 0) and np.array_equal(got[present == 1], vals)
    jvm.call("releasePage", None, I64(page))
    clean(jvm)


def test_parquet_page_decode_through_the_shim(pkg, jvm, jctx):
    """GpuNative.parquetDecodeDataPage: definition levels + a dictionary-encoded value section + the chunk's dictionary page -> a device page"""
    from oracle import parquet as opq
    rng = np.random.default_rng(5)
    n, dcount = 4000, 300
    present = (rng.random(n) < 0.85).astype(np.int32)
    ids = rng.integers(0, dcount, int(present.sum()))
    dict_vals = rng.integers(-10**12, 10**12, dcount)
    bw = int(dcount - 1).bit_length()
    as_i8 = lambda b: jvm.array(np.frombuffer(bytes(b), dtype=np.int8))   # noqa: E731
    page = jvm.checked("parquetDecodeDataPage", I64, jctx, I32(pkg.BIGINT), I32(opq.INT64), I32(opq.RLE_DICTIONARY), I32(n), as_i8(opq.hybrid_encode(present.tolist(), 1)),
                       as_i8(bytes([bw]) + opq.hybrid_encode(ids.tolist(), bw)), as_i8(opq.plain_encode(opq.INT64, dict_vals)), I32(dcount))
    got, nulls = heap_blocks(jvm, page)[0]
    assert np.array_equal(nulls, present == 0) and np.array_equal(got[present == 1], dict_vals[ids])
    jvm.call("releasePage", None, I64(page))
    clean(jvm)
