"""The Parquet page decoders' CPU restatement (oracle/parquet_oracle.c + oracle/parquet.py) against (a) the one Parquet file of the reference these
types cover, with the row its own test asserts, and (b) files written and read back by Apache Arrow in every page layout the decoders cover."""
import sys
import os

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parquet_cases as cases   # noqa: E402
import parquet_pages as pp      # noqa: E402


@pytest.fixture(scope="module")
def opq():
    from oracle import parquet as m
    return m


def test_the_references_own_parquet_file_decodes_to_the_row_its_test_asserts(opq):
    fx = cases.reference_fixture()
    rows = []
    for c, page, (dl, vals) in cases.fixture_pages(fx):
        assert page["kind"] == "DATA_V1" and c["physical"] == pp.INT32
        rows += opq.decode_data_page(c["physical"], page["encoding"], page["num_values"], vals, dl)
    assert [[v] for v in rows] == fx["asserted_rows"]        # TestParquetSymlinkInputFormat.java:63: row(42)


def decode_chunk(opq, chunk):
    dictionary, dcount, out = None, 0, []
    for page in chunk["pages"]:
        if page["kind"] == "DICTIONARY":
            dictionary, dcount = page["bytes"], page["num_values"]
            continue
        dl, vals = pp.split_data_page(chunk, page)
        out += opq.decode_data_page(chunk["physical"], page["encoding"], page["num_values"], vals, dl, dictionary, dcount)
    return out


def same(a, b):
    return len(a) == len(b) and all((x is None and y is None) or (x is not None and y is not None and (x == y or (isinstance(x, float) and np.float64(x).tobytes() == np.float64(y).tobytes())))
                                    for x, y in zip(a, b))


def test_pages_written_by_arrow_decode_to_what_arrow_reads(opq, tmp_path):
    seen = set()
    for label, path, table in cases.write_cases(tmp_path):
        for chunk in pp.column_chunks(path):
            got = decode_chunk(opq, chunk)
            want = cases.expected_column(table, chunk["name"], chunk["physical"])
            assert same(got, want), (label, chunk["name"])
            seen |= {(p["kind"], p["encoding"]) for p in chunk["pages"]}
    # every layout occurred: PLAIN and dictionary data pages of both versions, PLAIN dictionary pages
    assert {("DATA_V1", pp.PLAIN), ("DATA_V2", pp.PLAIN), ("DICTIONARY", pp.PLAIN)} <= seen
    assert any(k == "DATA_V1" and e in (pp.PLAIN_DICTIONARY, pp.RLE_DICTIONARY) for k, e in seen) and any(k == "DATA_V2" and e in (pp.PLAIN_DICTIONARY, pp.RLE_DICTIONARY) for k, e in seen)


def test_delta_binary_packed_pages_written_by_arrow_and_of_chosen_shapes(opq, tmp_path):
    """DELTA_BINARY_PACKED (INT32 / INT64): Arrow's writer against Arrow's reader through the restatement, then sections of chosen shapes through
    the test encoder (the specification read backwards): every miniblock width 0..64, wrapping deltas, counts around the block and miniblock
    edges, a different block shape, and the failure modes"""
    pages = 0
    for label, path, table in cases.write_delta_cases(tmp_path):
        for chunk in pp.column_chunks(path):
            assert same(decode_chunk(opq, chunk), cases.expected_column(table, chunk["name"], chunk["physical"])), (label, chunk["name"])
            want_enc = opq.DELTA_LENGTH_BYTE_ARRAY if chunk["name"].startswith("s") else opq.DELTA_BYTE_ARRAY if chunk["name"].startswith("p") else opq.DELTA_BINARY_PACKED
            assert all(p_["encoding"] == want_enc for p_ in chunk["pages"])
            pages += len(chunk["pages"])
    assert pages >= 20
    rng = np.random.default_rng(11)
    for n in (0, 1, 2, 32, 33, 34, 128, 129, 130, 257, 1000):
        for physical, bits in ((opq.INT64, 64), (opq.INT32, 32)):
            lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
            vals = rng.integers(lo, hi, n, endpoint=True).tolist()                      # deltas wrap around the type's range
            assert opq.delta_binary_packed(physical, opq.delta_encode(vals, physical), n) == vals
            assert opq.delta_binary_packed(physical, opq.delta_encode(vals, physical, 256, 8), n) == vals
    for w in range(0, 64):       # a miniblock whose packed deltas need exactly w bits
        vals = np.cumsum([0] + [int(x) for x in rng.integers(0, 1 << w, 200, dtype=np.uint64 if w == 63 else np.int64)]).tolist() if w < 62 else None
        if vals is not None:
            assert opq.delta_binary_packed(opq.INT64, opq.delta_encode(vals, opq.INT64), len(vals)) == vals, w
    sec = opq.delta_encode(list(range(300)), opq.INT64)
    with pytest.raises(ValueError):
        opq.delta_binary_packed(opq.INT64, sec[:len(sec) // 2], 300)                     # cut short
    with pytest.raises(ValueError):
        opq.delta_binary_packed(opq.INT64, sec, 301)                                     # fewer values than wanted
    with pytest.raises(ValueError):
        opq.delta_binary_packed(opq.DOUBLE, sec, 300)                                    # ParquetEncoding.java:151
    # DELTA_LENGTH_BYTE_ARRAY: lengths, then the bytes; lengths that overrun the bytes are refused
    words = [b"", b"a", b"x" * 700, "héllo".encode("utf-8")]
    for n in (0, 1, 33, 1000):
        vals = [words[int(x)] for x in rng.integers(0, len(words), n)]
        assert opq.delta_length_byte_array(opq.delta_length_encode(vals), n) == vals
    with pytest.raises(ValueError):
        opq.delta_length_byte_array(opq.delta_length_encode([b"abc", b"defg"])[:-2], 2)
    # DELTA_BYTE_ARRAY: prefixes of the value before + suffixes; a prefix longer than the value before is refused
    pw = [w.encode("utf-8") for w in cases.PREFIX_WORDS]
    for n in (0, 1, 2, 33, 1000):
        vals = sorted(pw[int(x)] for x in rng.integers(0, len(pw), n))
        assert opq.delta_byte_array(opq.delta_byte_array_encode(vals), n) == vals
        vals = [pw[int(x)] for x in rng.integers(0, len(pw), n)]
        assert opq.delta_byte_array(opq.delta_byte_array_encode(vals), n) == vals
    bad = opq.delta_encode([0, 5], opq.INT32) + opq.delta_length_encode([b"abc", b"d"])      # the second value claims 5 bytes of a 3-byte one
    with pytest.raises(ValueError):
        opq.delta_byte_array(bad, 2)


def test_hybrid_streams_of_every_width_and_their_failure_modes(opq):
    rng = np.random.default_rng(5)
    for bw in list(range(1, 33)):
        hi = 1 << min(bw, 31)
        v = np.concatenate([rng.integers(0, hi, 200), np.full(100, hi - 1), rng.integers(0, hi, 13), np.zeros(64, dtype=np.int64), rng.integers(0, hi, 1)])
        s = opq.hybrid_encode(v.tolist(), bw)
        assert np.array_equal(opq.hybrid(s, bw, len(v)), v.astype(np.int64).astype(np.uint32).view(np.int32) if bw == 32 else v)
        assert np.array_equal(opq.hybrid(s, bw, 150), opq.hybrid(s, bw, len(v))[:150])      # fewer values wanted than the stream holds
        with pytest.raises(ValueError):
            opq.hybrid(s[: len(s) // 2], bw, len(v))                                         # the stream ends before the values do
    assert np.array_equal(opq.hybrid(b"", 0, 7), np.zeros(7))                                # a one-entry dictionary: bit width 0
    with pytest.raises(ValueError):
        opq.hybrid(bytes([0x00]), 3, 1)                                                      # an RLE run of length 0
    assert opq.plain_values(pp.BYTE_ARRAY, opq.plain_encode(pp.BYTE_ARRAY, [b"", b"ab", b"x" * 70]), 3) == [b"", b"ab", b"x" * 70]
    with pytest.raises(ValueError):
        opq.plain_values(pp.BYTE_ARRAY, b"\x05\x00\x00\x00ab", 1)                            # a value longer than its section
    bits = [bool(x) for x in rng.integers(0, 2, 77)]
    assert opq.plain_values(pp.BOOLEAN, opq.plain_encode(pp.BOOLEAN, bits), 77) == bits
