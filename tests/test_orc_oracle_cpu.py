"""The oracle's ORC stream decoders (oracle/orc_oracle.c: RLEv2 / RLEv1 integers, bit unpacking, byte RLE, booleans) pinned on what the reference
holds: the streams of its own ORC test resources (tests/golden/orc_streams.json, made by tools/extract_orc_fixtures.py: decompressed stream bytes)
against the statistics the files' WRITER recorded per column (count / min / max / sum -- known answers computed by Apache ORC, not here) and the
row counts the reference's tests assert; the value lists of TestLongDecode.java:36-56; TestLongBitPacker.java:41-60's widths x lengths over
java.util.Random(0) bytes (the oracle's unpackGeneric against an independent big-integer bit reader)."""
import base64
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def orc():
    from oracle import orc as m
    return m


@pytest.fixture(scope="module")
def fixture():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "orc_streams.json")))


def streams_of(fixture):
    for f in fixture["files"]:
        for si, st in enumerate(f["stripes"]):
            for s in st["streams"]:
                col = s["column"]
                yield f, si, st, s, f["column_types"][col], st["encodings"][col]["kind"], base64.b64decode(s["bytes"])


def test_what_the_reference_tests_assert_about_the_files(fixture):
    lz4, hive = fixture["files"]
    assert lz4["compression_kind"] == 4 and lz4["number_of_rows"] == 10_000          # TestOrcLz4.java:45-46
    assert sum(s["rows"] for s in lz4["stripes"]) == 10_000
    assert hive["number_of_rows"] == 2                                                # TestOrcWithoutRowGroupInfo.java:62


def test_integer_streams_decode_to_the_writers_statistics(orc, fixture):
    checked = 0
    per_file_column = {}
    for f, si, st, s, ctype, enc, data in streams_of(fixture):
        if s["kind"] != "DATA" or ctype not in ("INT", "LONG", "SHORT"):
            continue
        assert enc in ("DIRECT_V2", "DIRECT")
        vals = orc.rle_v2(data, True) if enc == "DIRECT_V2" else orc.rle_v1(data, True)
        assert len(vals) == st["rows"], (f["file"], si, s["column"])             # no PRESENT stream in these files: one value per row
        per_file_column.setdefault((f["file"], s["column"]), []).append(vals)
        stats = (st["statistics"] or [None] * 99)[s["column"]]
        if stats and stats.get("int") and stats["number_of_values"]:
            assert stats["number_of_values"] == len(vals)
            assert int(vals.min()) == stats["int"]["min"] and int(vals.max()) == stats["int"]["max"]
            if stats["int"]["sum"] is not None:
                assert sum(int(v) for v in vals) == stats["int"]["sum"]
            checked += 1
    assert checked >= 6                     # 3 integer columns x 2 stripes of apache-lz4.orc carry stripe statistics
    # the file-level statistics over the concatenated stripes
    f = fixture["files"][0]
    for col in (1, 2, 3):
        vals = np.concatenate(per_file_column[(f["file"], col)])
        st = f["file_statistics"][col]
        assert len(vals) == st["number_of_values"] == 10_000
        assert int(vals.min()) == st["int"]["min"] and int(vals.max()) == st["int"]["max"]
        if st["int"]["sum"] is not None:
            assert sum(int(v) for v in vals) == st["int"]["sum"]
    y = np.concatenate(per_file_column[(f["file"], 2)])
    assert np.array_equal(np.sort(y), np.arange(10_000))          # the INT column: every value 0..9999 once (sum 49 995 000, min 0, max 9999)


def test_vint_values_of_test_long_decode(orc):
    """TestLongDecode.java:36-56: the listed values + [-100000, 100000), signed and unsigned: write -> read is the identity, and the bytes are the
    LEB128 groups of the (zigzag) value"""
    imax, imin, lmax, lmin = 2**31 - 1, -2**31, 2**63 - 1, -2**63
    listed = [0, 1, -1, imax, imax + 1, imax - 1, imin, imin + 1, imin - 1, lmax, lmax - 1, lmin + 1]
    for v in listed + list(range(-100_000, 100_000, 37)):
        for signed in (True, False):
            b = orc.write_vlong(v, signed)
            u = (((v << 1) ^ (v >> 63)) if signed else v) & (2**64 - 1)
            want = bytearray()
            while True:
                if u & ~0x7f == 0:
                    want.append(u)
                    break
                want.append(0x80 | (u & 0x7f))
                u >>= 7
            assert b == bytes(want)
            got, used = orc.read_vint(b, signed)
            assert used == len(b) and got == (v if signed else (v if v >= 0 else v + 2**64 - 2**64 * (1 if v + 2**64 >= 2**63 else 0)))


def java_random_bytes(n, seed=0):
    """java.util.Random(seed).nextInt(256), n times (TestLongBitPacker's RandomByteInputStream)"""
    s = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)
    out = bytearray()
    for _ in range(n):
        s = (s * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        r = s >> (48 - 31)                 # next(31)
        out.append((256 * r) >> 31)        # nextInt(bound) for a power-of-two bound
    return bytes(out)


def test_bit_unpacking_over_the_reference_tests_widths_and_lengths(orc):
    data = java_random_bytes(64 * 128 // 8 + 8)
    big = int.from_bytes(data, "big")
    total = len(data) * 8
    for length in range(0, 128, 3):
        for width in range(1, 65):
            vals, read = orc.unpack(data, length, width)
            assert read == (length * width + 7) // 8
            want = [(big >> (total - (i + 1) * width)) & ((1 << width) - 1) for i in range(length)]
            assert [int(v) & (2**64 - 1) for v in vals] == want, (length, width)


def test_bit_width_tables(orc):
    assert [orc.decode_bit_width(n) for n in range(32)] == list(range(1, 25)) + [26, 28, 30, 32, 40, 48, 56, 64]      # LongDecode.java:47-76
    assert [orc.closest_fixed_bits(w) for w in (0, 1, 24, 25, 26, 27, 29, 31, 33, 41, 49, 57, 64)] == [1, 1, 24, 26, 26, 28, 30, 32, 40, 48, 56, 64, 64]


def test_run_kinds_round_trip_through_the_test_encoders(orc):
    """SHORT_REPEAT / DIRECT / DELTA / PATCHED_BASE runs and byte RLE / boolean streams written by the test encoders (oracle/orc.py, after the ORC
    specification) decode to what went in -- the kinds the reference's two files do not all contain"""
    rng = np.random.default_rng(1)
    vals = rng.integers(-10**12, 10**12, 300)
    assert np.array_equal(orc.rle_v2(orc.rle_v2_direct(vals, True), True), vals)
    assert np.array_equal(orc.rle_v2(orc.rle_v2_direct(np.abs(vals), False), False), np.abs(vals))
    assert list(orc.rle_v2(orc.rle_v2_short_repeat(-77, 9, True), True)) == [-77] * 9
    assert list(orc.rle_v2(orc.rle_v2_delta(5, [3] * 100, True), True)) == [5 + 3 * i for i in range(101)]
    deltas = [7] + rng.integers(1, 1000, 200).tolist()
    assert list(orc.rle_v2(orc.rle_v2_delta(-50, deltas, True), True)) == np.cumsum([-50] + deltas).tolist()
    neg = [-4] + rng.integers(1, 50, 99).tolist()
    assert list(orc.rle_v2(orc.rle_v2_delta(10_000, neg, True), True)) == np.cumsum([10_000, -4] + [-d for d in neg[1:]]).tolist()
    base, fb, pw = 1000, 8, 12
    low = rng.integers(0, 256, 400)
    patches = [(17, 0x5a5), (200, 0x001), (150, 0xfff)]
    want = base + low.astype(np.int64)
    at = 0
    for gap, p in patches:
        at += gap
        want[at] += p << fb
    assert np.array_equal(orc.rle_v2(orc.rle_v2_patched_base(want, base, fb, pw, patches), False), want)
    raw = bytes(rng.integers(0, 4, 5000).astype(np.uint8))
    assert bytes(orc.byte_rle(orc.byte_rle_encode(raw))) == raw
    bits = (rng.random(12_345) < 0.3).astype(np.uint8)
    assert np.array_equal(orc.boolean(orc.boolean_encode(bits.tolist()), len(bits)), bits)
