"""ORC stream decode on the device (csrc/orc.hip behind tgpu_orc_decode_*) against the oracle (oracle/orc_oracle.c, pinned on the writer's
statistics of the reference's ORC test resources in tests/test_orc_oracle_cpu.py): the streams of those files, every RLEv2 run kind at its edge
sizes, PRESENT streams, 32-bit range errors, boolean and dictionary-string columns.  Bit-exact."""
import base64
import importlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def gorc():
    return importlib.import_module("presto-1_amd.orc")


@pytest.fixture(scope="module")
def orc():
    from oracle import orc as m
    return m


def test_the_reference_files_streams_decode_like_the_oracle_and_the_writers_statistics(pkg, ctx, gorc, orc):
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "orc_streams.json")))
    seen = 0
    for f in fx["files"]:
        for st in f["stripes"]:
            for s in st["streams"]:
                ctype = f["column_types"][s["column"]]
                if s["kind"] != "DATA" or ctype not in ("INT", "LONG"):
                    continue
                data = base64.b64decode(s["bytes"])
                want = orc.rle_v2(data, True)
                t = pkg.INTEGER if ctype == "INT" else pkg.BIGINT
                got = gorc.decode_long_column(ctx, t, st["rows"], data).to_host().getBlock(0)
                assert got.nulls is None or not got.nulls.any()
                assert np.array_equal(got.values.astype(np.int64), want[:st["rows"]])
                stats = (st["statistics"] or [None] * 99)[s["column"]]
                if stats and stats.get("int") and stats["number_of_values"]:
                    assert int(got.values.min()) == stats["int"]["min"] and int(got.values.max()) == stats["int"]["max"]
                    if stats["int"]["sum"] is not None:
                        assert sum(int(v) for v in got.values) == stats["int"]["sum"]
                seen += 1
    assert seen >= 11


def test_every_rle_v2_run_kind_and_width(pkg, ctx, gorc, orc):
    rng = np.random.default_rng(7)
    streams = []
    for width in list(range(1, 25)) + [26, 28, 30, 32, 40, 48, 56, 64]:
        for n in (1, 2, 63, 64, 65, 511, 512):
            hi = (1 << (width - 1)) - 1 if width < 64 else (1 << 62)
            vals = rng.integers(-hi - 1 if width > 1 else 0, hi + 1 if width > 1 else 1, n)
            streams.append(orc.rle_v2_direct([orc.zigzag(int(v)) for v in vals], False, width=width))
    streams += [orc.rle_v2_short_repeat(v, c, True) for v in (0, -1, 2**40, -2**62) for c in (3, 10)]
    streams += [orc.rle_v2_delta(10, [0] * 5, True), orc.rle_v2_delta(-3, [2] * 511, True), orc.rle_v2_delta(2**40, [-7] * 300, True)]
    for n in (2, 3, 64, 65, 500, 511):
        inc = [int(rng.integers(0, 5))] + rng.integers(0, 2**20, n - 1).tolist()
        streams.append(orc.rle_v2_delta(int(rng.integers(-1000, 1000)), inc, True))
        dec = [-int(rng.integers(1, 5))] + rng.integers(0, 2**9, n - 1).tolist()
        streams.append(orc.rle_v2_delta(2**50, dec, True))
    for n, gaps in ((40, [3, 9]), (512, [255, 255, 1]), (300, [17, 200, 60]), (512, [0] + [16] * 30)):
        base, fb, pw = int(rng.integers(-5000, 5000)), int(rng.choice([3, 8, 13])), int(rng.choice([4, 12, 24]))
        patches = [(g, int(rng.integers(1, 1 << pw)) if not (g == 255 and i < len(gaps) - 1 and False) else 0) for i, g in enumerate(gaps)]
        low = rng.integers(0, 1 << fb, n).astype(np.int64)
        vals = base + low
        at = 0
        for g, p_ in patches:
            at += g
            if at < n:
                vals[at] += p_ << fb
        streams.append(orc.rle_v2_patched_base(vals, base, fb, pw, [(g, p_) for g, p_ in patches if sum(x for x, _ in patches[:patches.index((g, p_)) + 1]) < n] or patches[:1]))
    # a patch list with a (255, 0) continuation entry: the gap to the next patch is 255 + 45
    vals = 100 + rng.integers(0, 256, 512).astype(np.int64)
    vals[300] += 0x7 << 8
    streams.append(orc.rle_v2_patched_base(vals, 100, 8, 4, [(255, 0), (45, 7)]))
    whole = b"".join(streams)
    want = orc.rle_v2(whole, True)
    got = gorc.decode_long_column(ctx, pkg.BIGINT, len(want), whole).to_host().getBlock(0).values
    assert np.array_equal(got, want)
    for s in streams[::7]:      # and stream by stream
        w = orc.rle_v2(s, True)
        assert np.array_equal(gorc.decode_long_column(ctx, pkg.BIGINT, len(w), s).to_host().getBlock(0).values, w)


def test_present_stream_expands_values_to_their_rows(pkg, ctx, gorc, orc):
    rng = np.random.default_rng(11)
    for n, null_frac in ((1, 1.0), (9, 0.5), (10_000, 0.3), (70_001, 0.01), (513, 0.0)):
        present = (rng.random(n) >= null_frac).astype(np.uint8)
        vals = rng.integers(-2**31, 2**31 - 1, int(present.sum()))
        data = b"".join(orc.rle_v2_direct(vals[i:i + 512], True) for i in range(0, len(vals), 512))
        for t, dt in ((pkg.BIGINT, np.int64), (pkg.DATE, np.int32)):
            blk = gorc.decode_long_column(ctx, t, n, data, present=orc.boolean_encode(present.tolist())).to_host().getBlock(0)
            nulls = blk.nulls if blk.nulls is not None else np.zeros(n, dtype=np.uint8)
            assert np.array_equal(nulls.astype(bool), present == 0)
            assert np.array_equal(blk.values[present == 1], vals.astype(dt))
    with pytest.raises(pkg.TgpuError) as e:        # LongInputStreamV2.next(int[]): "Decoded value out of range for a 32bit number"
        gorc.decode_long_column(ctx, pkg.INTEGER, 3, orc.rle_v2_direct([1, 2**31, 3], True))
    assert "32bit" in str(e.value)
    with pytest.raises(pkg.TgpuError):             # a truncated run
        gorc.decode_long_column(ctx, pkg.BIGINT, 512, orc.rle_v2_direct(list(range(512)), True)[:-5])


def test_boolean_and_dictionary_string_columns(pkg, ctx, gorc, orc):
    rng = np.random.default_rng(13)
    n = 30_000
    bits = (rng.random(n) < 0.4).astype(np.uint8)
    got = gorc.decode_boolean_column(ctx, n, orc.boolean_encode(bits.tolist())).to_host().getBlock(0)
    assert np.array_equal(got.values, bits)
    present = (rng.random(n) < 0.9).astype(np.uint8)
    vb = (rng.random(int(present.sum())) < 0.5).astype(np.uint8)
    got = gorc.decode_boolean_column(ctx, n, orc.boolean_encode(vb.tolist()), present=orc.boolean_encode(present.tolist())).to_host().getBlock(0)
    assert np.array_equal(got.nulls.astype(bool), present == 0) and np.array_equal(got.values[present == 1], vb)
    # SliceDictionaryColumnReader: ids (unsigned RLEv2) into a dictionary given as LENGTH stream + DICTIONARY_DATA
    words = ["", "a", "BUILDING", "MACHINERY", "x" * 40, "AUTOMOBILE", "héllo"]
    ids = rng.integers(0, len(words), int(present.sum()))
    id_stream = b"".join(orc.rle_v2_direct(ids[i:i + 512], False) for i in range(0, len(ids), 512))
    enc = [w.encode("utf-8") for w in words]
    length_stream = orc.rle_v2_direct([len(b) for b in enc], False)
    got = gorc.decode_dictionary_string_column(ctx, n, id_stream, len(words), length_stream, b"".join(enc), present=orc.boolean_encode(present.tolist())).to_host().getBlock(0)
    want = []
    it = iter(ids.tolist())
    for p_ in present:
        want.append(words[next(it)] if p_ else None)
    assert got.to_list() == want
    with pytest.raises(pkg.TgpuError):             # an id outside the dictionary
        gorc.decode_dictionary_string_column(ctx, 3, orc.rle_v2_direct([0, 9, 1], False), 2, orc.rle_v2_direct([1, 1], False), b"ab")


def test_direct_string_column(pkg, ctx, gorc, orc):
    """SliceDirectColumnReader.java:100-232: LENGTH = one unsigned RLEv2 length per NON-NULL row, DATA = those rows' bytes; with and without a
    PRESENT stream, empty strings, an all-null column, a stream whose lengths do not add up to the data"""
    rng = np.random.default_rng(17)
    n = 20_000
    words = ["", "a", "BUILDING", "MACHINERY", "x" * 300, "AUTOMOBILE", "héllo wörld"]
    for with_present in (False, True):
        present = (rng.random(n) < 0.85).astype(np.uint8) if with_present else np.ones(n, dtype=np.uint8)
        pick = rng.integers(0, len(words), int(present.sum()))
        enc = [words[i].encode("utf-8") for i in pick]
        lens = [len(b) for b in enc]
        length_stream = b"".join(orc.rle_v2_direct(lens[i:i + 512], False) for i in range(0, len(lens), 512))
        got = gorc.decode_direct_string_column(ctx, n, b"".join(enc), length_stream, present=orc.boolean_encode(present.tolist()) if with_present else None).to_host().getBlock(0)
        it = iter(pick.tolist())
        want = [words[next(it)] if p_ else None for p_ in present]
        assert got.to_list() == want
    none = gorc.decode_direct_string_column(ctx, 100, b"", b"", present=orc.boolean_encode([0] * 100)).to_host().getBlock(0)
    assert none.to_list() == [None] * 100
    assert gorc.decode_direct_string_column(ctx, 0, b"", b"").position_count == 0
    with pytest.raises(pkg.TgpuError):
        gorc.decode_direct_string_column(ctx, 3, b"abcdef", orc.rle_v2_direct([1, 1, 1], False))


def test_double_column(pkg, ctx, gorc, orc):
    """DoubleColumnReader.java:92-175: DATA = the non-null rows' doubles as 8 little-endian bytes; bit patterns (NaN payloads, -0.0, infinities,
    denormals) arrive unchanged, null rows hold 0"""
    rng = np.random.default_rng(19)
    n = 50_000
    vals = rng.standard_normal(n)
    vals[:6] = [np.nan, -0.0, np.inf, -np.inf, 5e-324, np.float64(np.frombuffer(np.uint64(0x7ff8000000000123).tobytes(), dtype=np.float64)[0])]
    got = gorc.decode_double_column(ctx, n, vals.astype("<f8").tobytes()).to_host().getBlock(0)
    assert got.nulls is None and np.array_equal(got.values.view(np.uint64), vals.view(np.uint64))
    present = (rng.random(n) < 0.8).astype(np.uint8)
    present[:8] = 1
    nn = vals[: int(present.sum())]
    got = gorc.decode_double_column(ctx, n, nn.astype("<f8").tobytes(), present=orc.boolean_encode(present.tolist())).to_host().getBlock(0)
    assert np.array_equal(got.nulls.astype(bool), present == 0)
    assert np.array_equal(got.values[present == 1].view(np.uint64), nn.view(np.uint64)) and not got.values[present == 0].any()
    with pytest.raises(pkg.TgpuError):             # a truncated DATA stream
        gorc.decode_double_column(ctx, 10, b"\x00" * 72)
    assert gorc.decode_double_column(ctx, 0, b"").position_count == 0


def test_rle_v1_columns(pkg, ctx, gorc, orc):
    """the DIRECT / DICTIONARY column encodings of files written before Hive 0.12: integer streams are RLEv1 (LongInputStreamV1.java:47-103: runs of
    3..130 values with a byte delta, literal groups of up to 128 varints).  Streams written by the test encoder (oracle/orc.py, checked against
    the oracle's decoder) through the long, dictionary-string and direct-string readers; a stream cut inside a varint is refused."""
    rng = np.random.default_rng(23)
    vals = np.concatenate([np.arange(0, 3000, 3), rng.integers(-10**15, 10**15, 5000), np.full(1000, -7), np.arange(5000, 0, -5), rng.integers(-3, 3, 777),
                           np.array([np.iinfo(np.int64).max, np.iinfo(np.int64).min, 0, -1, 1])]).astype(np.int64)
    stream = orc.rle_v1_encode(vals.tolist(), True)
    assert np.array_equal(orc.rle_v1(stream, True, cap=len(vals) + 10), vals)
    got = gorc.decode_long_column(ctx, pkg.BIGINT, len(vals), stream, encoding=gorc.DIRECT).to_host().getBlock(0)
    assert got.nulls is None and np.array_equal(got.values, vals)
    n = len(vals) + 4000
    present = np.ones(n, dtype=np.uint8)
    present[rng.choice(n, 4000, replace=False)] = 0
    got = gorc.decode_long_column(ctx, pkg.BIGINT, n, stream, present=orc.boolean_encode(present.tolist()), encoding=gorc.DIRECT).to_host().getBlock(0)
    assert np.array_equal(got.nulls.astype(bool), present == 0) and np.array_equal(got.values[present == 1], vals)
    small = rng.integers(-(2**31), 2**31, 3000).astype(np.int64)
    got = gorc.decode_long_column(ctx, pkg.INTEGER, len(small), orc.rle_v1_encode(small.tolist(), True), encoding=gorc.DIRECT).to_host().getBlock(0)
    assert np.array_equal(got.values, small.astype(np.int32))
    with pytest.raises(pkg.TgpuError):             # cut inside a literal's varint
        gorc.decode_long_column(ctx, pkg.BIGINT, 2, bytes([0xfe, 0x80, 0x80]), encoding=gorc.DIRECT)
    # strings: DICTIONARY (ids + lengths RLEv1) and DIRECT (lengths RLEv1)
    words = ["", "a", "BUILDING", "MACHINERY", "x" * 200, "héllo"]
    m = 20_000
    ids = rng.integers(0, len(words), m)
    enc = [w.encode("utf-8") for w in words]
    got = gorc.decode_dictionary_string_column(ctx, m, orc.rle_v1_encode(ids.tolist(), False), len(words), orc.rle_v1_encode([len(b) for b in enc], False), b"".join(enc),
                                               encoding=gorc.DICTIONARY).to_host().getBlock(0)
    assert got.to_list() == [words[i] for i in ids]
    rows = [enc[i] for i in ids]
    got = gorc.decode_direct_string_column(ctx, m, b"".join(rows), orc.rle_v1_encode([len(b) for b in rows], False), encoding=gorc.DIRECT).to_host().getBlock(0)
    assert got.to_list() == [words[i] for i in ids]
