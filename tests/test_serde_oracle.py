"""CPU tests of the SerializedPage restatement (oracle/oracle.py): the known-answer sizes of the reference's own test
(TestPagesSerde.java:64-110, tests/golden) and round trips of every block encoding."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))["pages_serde"]


def test_known_answer_sizes(oracle):
    O = oracle
    # TestPagesSerde.testBigintSerializedSize: an empty BIGINT page is written as an RLE block around a one-position null block
    null_block = O.serialize_block(O.Col(O.BIGINT, [0], nulls=[1]))
    assert len(O.serialized_page(0, [O.rle_block(null_block, 0)])) == GOLD["bigint_empty_page_bytes"]
    one = len(O.serialize_page([O.Col(O.BIGINT, [123])]))
    two = len(O.serialize_page([O.Col(O.BIGINT, [123, 456])]))
    assert one - GOLD["bigint_page_overhead"] == GOLD["bigint_first_value"]
    assert two - one == GOLD["bigint_second_value"]
    # testVarcharSerializedSize
    e = len(O.serialize_page([O.Col(O.VARCHAR, [])]))
    a = len(O.serialize_page([O.Col(O.VARCHAR, ["alice"])]))
    b = len(O.serialize_page([O.Col(O.VARCHAR, ["alice", "bob"])]))
    assert (e, a - e, b - a) == (GOLD["varchar_empty_page_bytes"], GOLD["varchar_alice"], GOLD["varchar_bob"])


def test_round_trip_reference_page(oracle):
    # TestPagesSerde.testRoundTrip: three identical VARCHAR channels
    O = oracle
    c = O.Col(O.VARCHAR, GOLD["roundtrip_strings"])
    data = O.serialize_page([c, c, c])
    n, cols = O.deserialize_page(data, [O.VARCHAR] * 3)
    assert n == 4 and len(cols) == 3
    for got in cols:
        assert np.array_equal(got.offsets, c.offsets) and bytes(got.values[: got.offsets[-1]]) == b"alicebobcharliedave" and got.nulls is None


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 64, 1000])
def test_round_trip_all_types_with_nulls(oracle, n):
    O = oracle
    rng = np.random.default_rng(n)
    nulls = lambda: (rng.random(n) < 0.3).astype(np.uint8)
    cols = [O.Col(O.BIGINT, rng.integers(-2**62, 2**62, n), nulls()), O.Col(O.INTEGER, rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)),
            O.Col(O.DATE, rng.integers(0, 20000, n).astype(np.int32), nulls()), O.Col(O.DOUBLE, rng.standard_normal(n), nulls()),
            O.Col(O.BOOLEAN, rng.integers(0, 2, n).astype(np.uint8), nulls()),
            O.Col(O.VARCHAR, [None if rng.random() < 0.3 else "v%d" % k * (k % 4) for k in rng.integers(0, 50, n)])]
    types = [c.type for c in cols]
    data = O.serialize_page(cols)
    m, back = O.deserialize_page(data, types)
    assert m == n
    assert O.serialize_page(back) == data   # byte-stable
    for a, b in zip(cols, back):
        keep = np.ones(n, dtype=bool) if a.nulls is None else a.nulls == 0
        assert (a.nulls is None) == (b.nulls is None) or n == 0
        if a.type == O.VARCHAR:
            assert np.array_equal(a.offsets, b.offsets)
        else:
            assert np.array_equal(np.asarray(a.values)[keep].view(np.uint8), np.asarray(b.values)[keep].view(np.uint8))


def test_null_bit_order(oracle):
    # EncoderUtil.java:45-57: position 0 is the most significant bit; the tail byte keeps its positions in the high bits
    O = oracle
    blk = O.serialize_block(O.Col(O.BOOLEAN, np.zeros(11, dtype=np.uint8), nulls=[1, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0]))
    body = blk[4 + len(b"BYTE_ARRAY") + 4:]
    assert body[0] == 1 and body[1] == 0b10000001 and body[2] == 0b01000000
    assert int.from_bytes(body[3:7], "little") == 8   # non-null positions


def test_lz4_block_round_trip(oracle):
    """the oracle's LZ4 block compressor / decompressor (test infrastructure for the compressed-page ingest): valid blocks, exact round trip"""
    import numpy as np
    rng = np.random.default_rng(3)
    for data in [b"", b"x", b"abc" * 700, bytes(70_000), rng.integers(0, 3, 20_000, dtype=np.uint8).tobytes(), rng.integers(0, 256, 5_000, dtype=np.uint8).tobytes()]:
        packed = oracle.lz4_block_compress(data)
        assert oracle.lz4_block_decompress(packed, len(data)) == data
    page = oracle.serialize_page([oracle.Col(oracle.BIGINT, np.zeros(5000, dtype=np.int64))])
    packed = oracle.compress_serialized_page(page)
    assert packed[4] == 1 and len(packed) < len(page) // 10 and int.from_bytes(packed[5:9], "little") == len(page) - 13
