"""ctypes wrapper over oracle/libtrino_oracle.so -- the CPU restatement of the reference algorithms -- plus the restatements that
are byte / index bookkeeping rather than arithmetic and therefore live here in numpy or small Python loops: the SerializedPage wire
format (serialize_page / deserialize_page), PagePartitioner, MergePages and DynamicFilterSource (page- or position-at-a-time state
machines, used on small test inputs only).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package.  See trino_oracle.h for the reference citations of the C functions; the Python restatements cite
theirs next to each class.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtrino_oracle.so")

BIGINT, INTEGER, DATE, DOUBLE, BOOLEAN, VARCHAR = 1, 2, 3, 4, 5, 6
_NP = {BIGINT: np.int64, INTEGER: np.int32, DATE: np.int32, DOUBLE: np.float64, BOOLEAN: np.uint8}

OK, ERR_INVALID, ERR_NUMERIC_VALUE_OUT_OF_RANGE, ERR_INSUFFICIENT_RESOURCES, ERR_DIVISION_BY_ZERO = 0, -1, -2, -3, -7


def build(force=False):
    src = os.path.join(_HERE, "trino_oracle.c")
    hdr = os.path.join(_HERE, "trino_oracle.h")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtrino_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class OColumn(C.Structure):
    _fields_ = [("type", C.c_int32), ("n", C.c_int32), ("values", C.c_void_p), ("nulls", C.c_void_p), ("offsets", C.c_void_p)]


class OExprNode(C.Structure):
    _fields_ = [("kind", C.c_int32), ("type", C.c_int32), ("op", C.c_int32), ("n_args", C.c_int32),
                ("args", C.c_int32 * 3), ("is_null", C.c_int32), ("ival", C.c_int64), ("dval", C.c_double),
                ("slen", C.c_int32), ("pad", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        i32, i64, f64, vp = C.c_int32, C.c_int64, C.c_double, C.c_void_p
        sig = {
            "o_hash_long": (i64, [i64]), "o_hash_int": (i64, [i32]), "o_hash_double": (i64, [f64]),
            "o_hash_boolean": (i64, [C.c_uint8]),
            "o_xxh64": (C.c_uint64, [vp, C.c_size_t, C.c_uint64]), "o_xxh64_long": (i64, [i64]),
            "o_combine_hash": (i64, [i64, i64]), "o_murmur3_fmix": (C.c_uint64, [C.c_uint64]),
            "o_array_size": (i32, [i32, C.c_float]), "o_calculate_max_fill": (i32, [i32]),
            "o_hash_rows": (None, [vp, i32, i32, vp]),
            "o_partition_remote": (i32, [i64, i32]), "o_partition_local": (i32, [i64, i32]),
            "o_bigint_gbh_new": (vp, [i32]), "o_bigint_gbh_free": (None, [vp]),
            "o_bigint_gbh_get_group_ids": (i32, [vp, vp, vp]), "o_bigint_gbh_contains": (i32, [vp, vp, i32]),
            "o_bigint_gbh_group_count": (i32, [vp]), "o_bigint_gbh_capacity": (i32, [vp]),
            "o_bigint_gbh_hash_collisions": (i64, [vp]), "o_bigint_gbh_rehash_count": (i32, [vp]),
            "o_bigint_gbh_values": (None, [vp, vp, vp, vp]),
            "o_multi_gbh_new": (vp, [i32, vp, i32]), "o_multi_gbh_free": (None, [vp]),
            "o_multi_gbh_get_group_ids": (i32, [vp, vp, vp, i32, vp]),
            "o_multi_gbh_contains": (i32, [vp, vp, i32, i64]),
            "o_multi_gbh_group_count": (i32, [vp]), "o_multi_gbh_capacity": (i32, [vp]),
            "o_multi_gbh_rehash_count": (i32, [vp]), "o_multi_gbh_group_rows": (None, [vp, vp, vp]),
            "o_agg_double_sum": (None, [vp, vp, vp, vp, i32, vp, vp]),
            "o_agg_long_avg": (None, [vp, vp, vp, vp, i32, vp, vp]),
            "o_agg_long_sum": (i32, [vp, vp, vp, vp, i32, vp, vp]),
            "o_agg_long_minmax": (None, [vp, vp, vp, vp, i32, i32, vp, vp]),
            "o_agg_double_minmax": (None, [vp, vp, vp, vp, i32, i32, vp, vp]),
            "o_agg_count": (None, [vp, vp, vp, i32, vp]),
            "o_exact_sum": (f64, [vp, i64]),
            "o_agg_double_sum_exact": (None, [vp, vp, vp, vp, i64, i32, vp, vp]),
            "o_pages_hash_new": (vp, [vp, i32, i32, vp]), "o_pages_hash_free": (None, [vp]),
            "o_pages_hash_size": (i32, [vp]), "o_pages_hash_link_count": (i32, [vp]),
            "o_pages_hash_links": (vp, [vp]), "o_pages_hash_keys": (vp, [vp]), "o_pages_hash_collisions": (i64, [vp]),
            "o_pages_hash_get_address_index": (i32, [vp, vp, i32, i64]),
            "o_join_probe": (i64, [vp, vp, i32, vp, i32, vp, vp, i64]),
            "o_filter": (i32, [vp, i32, C.c_char_p, vp, i32, vp, vp]),
            "o_project": (i32, [vp, i32, C.c_char_p, vp, vp, i32, vp, vp, vp]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Col:
    """Flat column: the oracle-side mirror of LongArrayBlock / IntArrayBlock / ByteArrayBlock / VariableWidthBlock."""

    def __init__(self, type_id, values, nulls=None, offsets=None):
        self.type = type_id
        if type_id == VARCHAR:
            if offsets is None:  # list of python str/bytes/None
                items = list(values)
                nl = np.array([v is None for v in items], dtype=np.uint8)
                bs = [b"" if v is None else (v.encode("utf-8") if isinstance(v, str) else bytes(v)) for v in items]
                offsets = np.zeros(len(bs) + 1, dtype=np.int32)
                if bs:
                    offsets[1:] = np.cumsum([len(b) for b in bs])
                values = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if sum(len(b) for b in bs) else np.zeros(1, dtype=np.uint8)
                if nulls is None and nl.any():
                    nulls = nl
            self.values = np.ascontiguousarray(values, dtype=np.uint8)
            self.offsets = np.ascontiguousarray(offsets, dtype=np.int32)
            self.n = len(self.offsets) - 1
        else:
            self.values = np.ascontiguousarray(values, dtype=_NP[type_id])
            self.offsets = None
            self.n = len(self.values)
        self.nulls = None if nulls is None else np.ascontiguousarray(nulls, dtype=np.uint8)

    def struct(self):
        return OColumn(self.type, self.n, _ptr(self.values), _ptr(self.nulls), _ptr(self.offsets))


def col_array(cols):
    arr = (OColumn * max(1, len(cols)))()
    for i, c in enumerate(cols):
        arr[i] = c.struct()
    return arr


# ---- hashes -------------------------------------------------------------------------------------
def hash_long(v):
    return lib().o_hash_long(int(v))


def hash_double(v):
    return lib().o_hash_double(float(v))


def xxh64(data: bytes, seed=0):
    buf = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(1, dtype=np.uint8)
    return lib().o_xxh64(_ptr(buf), len(data), seed)


def xxh64_long(v):
    return lib().o_xxh64_long(int(v))


def hash_rows(cols):
    n = cols[0].n
    out = np.zeros(n, dtype=np.int64)
    lib().o_hash_rows(col_array(cols), len(cols), n, _ptr(out))
    return out


def partition_remote(raw_hashes, count):
    L = lib()
    return np.array([L.o_partition_remote(int(h), count) for h in raw_hashes], dtype=np.int32)


def partition_local(raw_hashes, count):
    L = lib()
    return np.array([L.o_partition_local(int(h), count) for h in raw_hashes], dtype=np.int32)


# ---- group by -----------------------------------------------------------------------------------
class BigintGroupByHash:
    def __init__(self, expected_size):
        self.h = lib().o_bigint_gbh_new(expected_size)

    def __del__(self):
        if getattr(self, "h", None):
            lib().o_bigint_gbh_free(self.h)
            self.h = None

    def get_group_ids(self, col):
        out = np.zeros(col.n, dtype=np.int64)
        s = col.struct()
        rc = lib().o_bigint_gbh_get_group_ids(self.h, C.byref(s), _ptr(out))
        if rc != 0:
            raise OracleError(rc)
        return out

    def contains(self, col, pos):
        s = col.struct()
        return bool(lib().o_bigint_gbh_contains(self.h, C.byref(s), pos))

    @property
    def group_count(self):
        return lib().o_bigint_gbh_group_count(self.h)

    @property
    def capacity(self):
        return lib().o_bigint_gbh_capacity(self.h)

    @property
    def rehash_count(self):
        return lib().o_bigint_gbh_rehash_count(self.h)

    def values(self):
        n = self.group_count
        v = np.zeros(n, dtype=np.int64)
        nl = np.zeros(n, dtype=np.uint8)
        rh = np.zeros(n, dtype=np.int64)
        lib().o_bigint_gbh_values(self.h, _ptr(v), _ptr(nl), _ptr(rh))
        return v, nl, rh


class MultiChannelGroupByHash:
    def __init__(self, types, expected_size):
        t = np.array(types, dtype=np.int32)
        self.h = lib().o_multi_gbh_new(len(types), _ptr(t), expected_size)

    def __del__(self):
        if getattr(self, "h", None):
            lib().o_multi_gbh_free(self.h)
            self.h = None

    def get_group_ids(self, cols, hashes=None):
        n = cols[0].n
        out = np.zeros(n, dtype=np.int64)
        hs = None if hashes is None else np.ascontiguousarray(hashes, dtype=np.int64)
        rc = lib().o_multi_gbh_get_group_ids(self.h, col_array(cols), _ptr(hs), n, _ptr(out))
        if rc != 0:
            raise OracleError(rc)
        return out

    def contains(self, cols, pos, raw_hash):
        return bool(lib().o_multi_gbh_contains(self.h, col_array(cols), pos, int(raw_hash)))

    @property
    def group_count(self):
        return lib().o_multi_gbh_group_count(self.h)

    @property
    def capacity(self):
        return lib().o_multi_gbh_capacity(self.h)

    @property
    def rehash_count(self):
        return lib().o_multi_gbh_rehash_count(self.h)

    def group_rows(self):
        n = self.group_count
        fr = np.zeros(n, dtype=np.int64)
        rh = np.zeros(n, dtype=np.int64)
        lib().o_multi_gbh_group_rows(self.h, _ptr(fr), _ptr(rh))
        return fr, rh


class OracleError(Exception):
    def __init__(self, code, row=None):
        super().__init__(f"oracle error {code} at row {row}")
        self.code = code
        self.row = row


# ---- accumulators -------------------------------------------------------------------------------
def _gid(gids):
    return None if gids is None else np.ascontiguousarray(gids, dtype=np.int64)


def agg_double_sum(gids, values, ngroups, nulls=None, mask=None):
    counts = np.zeros(ngroups, dtype=np.int64)
    sums = np.zeros(ngroups, dtype=np.float64)
    v = np.ascontiguousarray(values, dtype=np.float64)
    g = _gid(gids)
    lib().o_agg_double_sum(_ptr(g), _ptr(v), _ptr(nulls), _ptr(mask), len(v), _ptr(counts), _ptr(sums))
    return counts, sums


def agg_double_sum_exact(gids, values, ngroups, nulls=None, mask=None):
    counts = np.zeros(ngroups, dtype=np.int64)
    sums = np.zeros(ngroups, dtype=np.float64)
    v = np.ascontiguousarray(values, dtype=np.float64)
    g = _gid(gids)
    lib().o_agg_double_sum_exact(_ptr(g), _ptr(v), _ptr(nulls), _ptr(mask), len(v), ngroups, _ptr(counts), _ptr(sums))
    return counts, sums


def agg_long_avg(gids, values, ngroups, nulls=None, mask=None):
    counts = np.zeros(ngroups, dtype=np.int64)
    sums = np.zeros(ngroups, dtype=np.float64)
    v = np.ascontiguousarray(values, dtype=np.int64)
    g = _gid(gids)
    lib().o_agg_long_avg(_ptr(g), _ptr(v), _ptr(nulls), _ptr(mask), len(v), _ptr(counts), _ptr(sums))
    return counts, sums


def agg_long_sum(gids, values, ngroups, nulls=None, mask=None):
    counts = np.zeros(ngroups, dtype=np.int64)
    sums = np.zeros(ngroups, dtype=np.int64)
    v = np.ascontiguousarray(values, dtype=np.int64)
    g = _gid(gids)
    rc = lib().o_agg_long_sum(_ptr(g), _ptr(v), _ptr(nulls), _ptr(mask), len(v), _ptr(counts), _ptr(sums))
    if rc != 0:
        raise OracleError(rc)
    return counts, sums


def agg_long_minmax(gids, values, ngroups, is_min, nulls=None, mask=None):
    """(counts, extremes): extremes[g] is meaningful where counts[g] > 0 (the state is null otherwise)"""
    counts = np.zeros(ngroups, dtype=np.int64)
    out = np.zeros(ngroups, dtype=np.int64)
    v = np.ascontiguousarray(values, dtype=np.int64)
    g = _gid(gids)
    lib().o_agg_long_minmax(_ptr(g), _ptr(v), _ptr(nulls), _ptr(mask), len(v), 1 if is_min else 0, _ptr(counts), _ptr(out))
    return counts, out


def agg_double_minmax(gids, values, ngroups, is_min, nulls=None, mask=None):
    """(counts, extremes) in the reference's row order semantics (first of equal values, NaN rules): meaningful where counts[g] > 0"""
    counts = np.zeros(ngroups, dtype=np.int64)
    out = np.zeros(ngroups, dtype=np.float64)
    v = np.ascontiguousarray(values, dtype=np.float64)
    g = _gid(gids)
    lib().o_agg_double_minmax(_ptr(g), _ptr(v), _ptr(nulls), _ptr(mask), len(v), 1 if is_min else 0, _ptr(counts), _ptr(out))
    return counts, out


def agg_count(gids, n, ngroups, nulls=None, mask=None):
    counts = np.zeros(ngroups, dtype=np.int64)
    g = _gid(gids)
    lib().o_agg_count(_ptr(g), _ptr(nulls), _ptr(mask), n, _ptr(counts))
    return counts


def exact_sum(values):
    v = np.ascontiguousarray(values, dtype=np.float64)
    return lib().o_exact_sum(_ptr(v), len(v))


# ---- join ---------------------------------------------------------------------------------------
class PagesHash:
    def __init__(self, key_cols, hashes=None):
        self.cols = list(key_cols)  # keep buffers alive
        self.n = key_cols[0].n
        self._arr = col_array(self.cols)
        self.hashes = None if hashes is None else np.ascontiguousarray(hashes, dtype=np.int64)
        self.h = lib().o_pages_hash_new(self._arr, len(self.cols), self.n, _ptr(self.hashes))

    def __del__(self):
        if getattr(self, "h", None):
            lib().o_pages_hash_free(self.h)
            self.h = None

    @property
    def hash_size(self):
        return lib().o_pages_hash_size(self.h)

    @property
    def link_count(self):
        return lib().o_pages_hash_link_count(self.h)

    def links(self):
        p = lib().o_pages_hash_links(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(max(self.n, 1),))[: self.n].copy()

    def get_address_index(self, probe_cols, pos, raw_hash):
        return lib().o_pages_hash_get_address_index(self.h, col_array(probe_cols), pos, int(raw_hash))

    def probe(self, probe_cols, hashes=None, probe_outer=False):
        n = probe_cols[0].n
        hs = None if hashes is None else np.ascontiguousarray(hashes, dtype=np.int64)
        cap = max(16, 2 * n)
        while True:
            op = np.zeros(cap, dtype=np.int32)
            ob = np.zeros(cap, dtype=np.int32)
            cnt = lib().o_join_probe(self.h, col_array(probe_cols), n, _ptr(hs), int(probe_outer), _ptr(op), _ptr(ob), cap)
            if cnt >= 0:
                return op[:cnt].copy(), ob[:cnt].copy()
            cap = -cnt


# ---- expressions --------------------------------------------------------------------------------
EX_INPUT, EX_CONST, EX_CALL, EX_SPECIAL = 0, 1, 2, 3
OPS = {"ADD": 1, "SUBTRACT": 2, "MULTIPLY": 3, "DIVIDE": 4, "MODULUS": 5, "NEGATE": 6, "EQUAL": 7, "NOT_EQUAL": 8,
       "LESS_THAN": 9, "LESS_THAN_OR_EQUAL": 10, "GREATER_THAN": 11, "GREATER_THAN_OR_EQUAL": 12, "NOT": 13, "CAST": 14}
FORMS = {"AND": 1, "OR": 2, "IF": 3, "IS_NULL": 4, "COALESCE": 5, "BETWEEN": 6}


def encode_nodes(flat_nodes, pool: bytes):
    """flat_nodes: list of dicts {kind,type,op,args,is_null,ival,dval,slen} (the product's serialisation)."""
    arr = (OExprNode * len(flat_nodes))()
    for i, nd in enumerate(flat_nodes):
        e = arr[i]
        e.kind, e.type, e.op = nd["kind"], nd["type"], nd["op"]
        e.n_args = len(nd.get("args", []))
        for k, a in enumerate(nd.get("args", [])):
            e.args[k] = a
        e.is_null = int(nd.get("is_null", 0))
        e.ival = int(nd.get("ival", 0))
        e.dval = float(nd.get("dval", 0.0))
        e.slen = int(nd.get("slen", 0))
    return arr


def filter_positions(flat_nodes, root, pool, cols):
    n = cols[0].n if cols else 0
    arr = encode_nodes(flat_nodes, pool)
    pos = np.zeros(max(n, 1), dtype=np.int32)
    err_row = C.c_int32(-1)
    rc = lib().o_filter(arr, root, pool, col_array(cols), n, _ptr(pos), C.byref(err_row))
    if rc < 0:
        raise OracleError(rc, err_row.value)
    return pos[:rc].copy()


def project(flat_nodes, root, pool, cols, positions):
    arr = encode_nodes(flat_nodes, pool)
    t = flat_nodes[root]["type"]
    n_sel = len(positions)
    out = np.zeros(max(n_sel, 1), dtype=_NP[t])
    nulls = np.zeros(max(n_sel, 1), dtype=np.uint8)
    pos = np.ascontiguousarray(positions, dtype=np.int32)
    err_row = C.c_int32(-1)
    rc = lib().o_project(arr, root, pool, col_array(cols), _ptr(pos), n_sel, _ptr(out), _ptr(nulls), C.byref(err_row))
    if rc < 0:
        raise OracleError(rc, err_row.value)
    return out[:n_sel], nulls[:n_sel]


# ---- TopN ---------------------------------------------------------------------------------------
ASC_NULLS_FIRST, ASC_NULLS_LAST, DESC_NULLS_FIRST, DESC_NULLS_LAST = 0, 1, 2, 3


def top_n(cols, n, sort_channels, sort_orders):
    """row numbers of the n first rows in the order of the sort channels (TopNOperator); equal rows keep their input order"""
    rows = cols[0].n
    out = np.zeros(max(min(n, rows), 1), dtype=np.int32)
    sc = np.ascontiguousarray(sort_channels, dtype=np.int32)
    so = np.ascontiguousarray(sort_orders, dtype=np.int32)
    L = lib()
    L.o_top_n.restype = C.c_int32
    k = L.o_top_n(col_array(cols), rows, int(n), _ptr(sc), _ptr(so), len(sc), _ptr(out))
    return out[:k]


# ---- SerializedPage (exchange / spill wire format) ---------------------------------------------------------------------
# Restated in numpy from the reference's writers and readers (byte arithmetic only):
#   framing   M/execution/buffer/PagesSerdeUtil.java:45-71 (writeRawPage, writeSerializedPage), PagesSerde.java:64-115
#   names     M/metadata/InternalBlockEncodingSerde.java:56-80, 103-116 (length-prefixed encoding name)
#   blocks    S/block/LongArrayBlockEncoding.java:37-105, IntArrayBlockEncoding, ByteArrayBlockEncoding,
#             VariableWidthBlockEncoding.java:37-76, RunLengthBlockEncoding.java:31-53, DictionaryBlockEncoding.java:33-80
#   null bits S/block/EncoderUtil.java:33-71, 84-118
# Pinned on the known-answer sizes of M/.../TestPagesSerde.java:64-110 (tests/golden) -- the reference holds no golden BYTES for
# this format, so the byte content is pinned by those sizes plus the format restatement only.
_ENCODING = {BIGINT: b"LONG_ARRAY", DOUBLE: b"LONG_ARRAY", INTEGER: b"INT_ARRAY", DATE: b"INT_ARRAY", BOOLEAN: b"BYTE_ARRAY", VARCHAR: b"VARIABLE_WIDTH"}
_WIDTH = {b"LONG_ARRAY": 8, b"INT_ARRAY": 4, b"BYTE_ARRAY": 1}


def _i32le(v):
    return int(v).to_bytes(4, "little", signed=True)


def _null_bits(nulls, n):
    """EncoderUtil.encodeNullsAsBits: mayHaveNull byte, then position p = bit (7 - p % 8) of byte p // 8"""
    if nulls is None:
        return b"\x00"
    return b"\x01" + np.packbits(np.asarray(nulls[:n], dtype=np.uint8) != 0).tobytes()


def serialize_block(col: "Col") -> bytes:
    name = _ENCODING[col.type]
    out = [_i32le(len(name)), name, _i32le(col.n)]
    if col.type == VARCHAR:
        base = int(col.offsets[0])
        out.append((col.offsets[1:].astype(np.int64) - base).astype("<i4").tobytes())
        out.append(_null_bits(col.nulls, col.n))
        total = int(col.offsets[col.n]) - base
        out += [_i32le(total), col.values[base:base + total].tobytes()]
    else:
        out.append(_null_bits(col.nulls, col.n))
        raw = col.values[: col.n].view(np.uint8).reshape(col.n, -1) if col.n else np.zeros((0, 1), dtype=np.uint8)
        if col.nulls is None:
            out.append(raw.tobytes())
        else:
            keep = np.asarray(col.nulls[: col.n]) == 0
            out += [_i32le(int(keep.sum())), raw[keep].tobytes()]
    return b"".join(out)


def rle_block(value_block: bytes, count: int) -> bytes:
    """RunLengthBlockEncoding.writeBlock around an already serialized one-position block"""
    return _i32le(3) + b"RLE" + _i32le(count) + value_block


def dictionary_block(dictionary_block_bytes: bytes, ids) -> bytes:
    """DictionaryBlockEncoding.writeBlock around an already serialized dictionary block (the 24 id bytes are arbitrary)"""
    ids = np.asarray(ids, dtype="<i4")
    return _i32le(10) + b"DICTIONARY" + _i32le(len(ids)) + dictionary_block_bytes + ids.tobytes() + bytes(24)


def serialized_page(position_count: int, blocks) -> bytes:
    """writeRawPage + writeSerializedPage (no compression, no encryption) around serialized blocks"""
    payload = _i32le(len(blocks)) + b"".join(blocks)
    return _i32le(position_count) + b"\x00" + _i32le(len(payload)) + _i32le(len(payload)) + payload


def serialize_page(cols) -> bytes:
    n = cols[0].n if cols else 0
    return serialized_page(n, [serialize_block(c) for c in cols])


def _read_block(data, at, type_id):
    ln = int.from_bytes(data[at:at + 4], "little", signed=True)
    name = bytes(data[at + 4:at + 4 + ln])
    at += 4 + ln
    n = int.from_bytes(data[at:at + 4], "little", signed=True)
    at += 4

    def null_bits(at):
        if data[at] == 0:
            return None, at + 1
        nb = (n + 7) // 8
        bits = np.unpackbits(np.frombuffer(data[at + 1:at + 1 + nb], dtype=np.uint8))[:n]
        return bits.astype(np.uint8), at + 1 + nb

    if name in _WIDTH:
        w = _WIDTH[name]
        nulls, at = null_bits(at)
        dt = {8: "<i8", 4: "<i4", 1: "u1"}[w]
        if nulls is None:
            vals = np.frombuffer(data[at:at + n * w], dtype=dt).copy()
            at += n * w
        else:
            nn = int.from_bytes(data[at:at + 4], "little", signed=True)
            at += 4
            compact = np.frombuffer(data[at:at + nn * w], dtype=dt)
            at += nn * w
            vals = np.zeros(n, dtype=dt)
            vals[nulls == 0] = compact
        if type_id == DOUBLE:
            vals = vals.view(np.float64)
        return Col(type_id, vals.astype(_NP[type_id], copy=False), nulls), at
    if name == b"VARIABLE_WIDTH":
        ends = np.frombuffer(data[at:at + 4 * n], dtype="<i4")
        at += 4 * n
        nulls, at = null_bits(at)
        total = int.from_bytes(data[at:at + 4], "little", signed=True)
        at += 4
        pool = np.frombuffer(data[at:at + total], dtype=np.uint8).copy() if total else np.zeros(1, dtype=np.uint8)
        at += total
        return Col(VARCHAR, pool, nulls, np.concatenate([[0], ends]).astype(np.int32)), at
    if name in (b"RLE", b"DICTIONARY"):
        inner, at = _read_block(data, at, type_id)
        if name == b"RLE":
            ids = np.zeros(n, dtype=np.int64)
        else:
            ids = np.frombuffer(data[at:at + 4 * n], dtype="<i4").astype(np.int64)
            at += 4 * n + 24
        nulls = None if inner.nulls is None else inner.nulls[ids]
        if type_id == VARCHAR:
            items = [None if (inner.nulls is not None and inner.nulls[i]) else bytes(inner.values[inner.offsets[i]:inner.offsets[i + 1]]) for i in ids]
            c = Col(VARCHAR, items)
            if nulls is not None and c.nulls is None:
                c.nulls = np.ascontiguousarray(nulls, dtype=np.uint8)
            return c, at
        return Col(type_id, inner.values[ids], nulls), at
    raise ValueError(f"unknown block encoding {name!r}")


def deserialize_page(data: bytes, types):
    """readSerializedPage + readRawPage: (position_count, [Col])"""
    data = memoryview(data)
    n = int.from_bytes(data[0:4], "little", signed=True)
    assert data[4] == 0, "compressed / encrypted pages are not restated"
    size = int.from_bytes(data[9:13], "little", signed=True)
    assert int.from_bytes(data[5:9], "little", signed=True) == size and 13 + size == len(data)
    at = 13
    channels = int.from_bytes(data[at:at + 4], "little", signed=True)
    at += 4
    assert channels == len(types)
    cols = []
    for t in types:
        c, at = _read_block(data, at, t)
        cols.append(c)
    assert at == len(data)
    return n, cols


# ---- PartitionedOutputOperator ---------------------------------------------------------------------------------------------
class PagePartitioner:
    """PagePartitioner.partitionPage (M/operator/PartitionedOutputOperator.java:406-426), row at a time: the positions every
    partition receives from each page, in append order.  partition = (rawHash & 0x7fff...) % count (HashGenerator.java:24-35)."""

    def __init__(self, partition_count, replicates_any_row=False, null_channel=-1, local=False):
        self.local = local   # LocalPartitionGenerator.java:45-65 instead of HashGenerator.java:24-35
        self.count = partition_count
        self.replicates_any_row = replicates_any_row
        self.null_channel = null_channel
        self.has_any_row_been_replicated = False

    def partition_page(self, cols, raw_hashes):
        n = cols[0].n if cols else 0
        out = [[] for _ in range(self.count)]
        nulls = cols[self.null_channel].nulls if self.null_channel >= 0 else None
        parts = (partition_local if self.local else partition_remote)(np.asarray(raw_hashes, dtype=np.int64), self.count) if n else []
        for position in range(n):
            replicate = (self.replicates_any_row and not self.has_any_row_been_replicated) or (nulls is not None and nulls[position] != 0)
            if replicate:
                for p in range(self.count):
                    out[p].append(position)
                self.has_any_row_been_replicated = True
            else:
                out[int(parts[position])].append(position)
        return out


# ---- MergePages -------------------------------------------------------------------------------------------------------------
def page_size_in_bytes(cols):
    """Page.getSizeInBytes of flat blocks: (width + 1) per fixed-width cell (S/block/LongArrayBlock.java:69 and siblings),
    length + 5 per VARCHAR cell (VariableWidthBlock.java:121-124)"""
    s = 0
    for c in cols:
        if c.type == VARCHAR:
            s += int(c.offsets[c.n]) - int(c.offsets[0]) + 5 * c.n
        else:
            s += (np.dtype(_NP[c.type]).itemsize + 1) * c.n
    return s


class MergePages:
    """MergePagesTransformation.process (M/operator/project/MergePages.java:122-190), page at a time: returns, for the pages fed so
    far, the list of output pages as lists of (input page index, row) pairs"""

    def __init__(self, min_page_size_in_bytes, min_row_count, max_page_size_in_bytes):
        self.min_size, self.min_rows, self.max_size = min_page_size_in_bytes, min_row_count, max_page_size_in_bytes
        self.buffer, self.buffered_size, self.index = [], 0, 0

    def add(self, cols):
        n = cols[0].n
        rows = [(self.index, r) for r in range(n)]
        self.index += 1
        out = []
        if n >= self.min_rows or page_size_in_bytes(cols) >= self.min_size:   # :145-157 (an unloaded LazyBlock cannot occur here)
            if self.buffer:
                out.append(self.flush())
            out.append(rows)
            return out
        self.buffer += rows   # :159
        self.buffered_size += page_size_in_bytes(cols)
        if self.buffered_size >= self.max_size:   # PageBuilder.isFull, S/PageBuilder.java:126-129
            out.append(self.flush())
        return out

    def flush(self):
        rows, self.buffer, self.buffered_size = self.buffer, [], 0
        return rows

    def finish(self):
        return [self.flush()] if self.buffer else []   # :134-142


# ---- DynamicFilterSourceOperator --------------------------------------------------------------------------------------------
class DynamicFilterSource:
    """DynamicFilterSourceOperator.addInput / finish (M/operator/DynamicFilterSourceOperator.java:213-424) position at a time, with the
    one documented deviation of the product (block-accounting size model instead of JVM retained sizes); min / max for every orderable
    type but DOUBLE (:187-190; BOOLEAN as 0 / 1, VARCHAR by unsigned bytes).  domain(k) -> ("all",) | ("values", [..first-seen order..]) |
    ("range", lo, hi) | ("none",)"""

    def __init__(self, types, channels, max_distinct_values, max_filter_size_in_bytes, min_max_collection_limit):
        self.types, self.channels = list(types), list(channels)
        self.max_distinct, self.max_size, self.limit = max_distinct_values, max_filter_size_in_bytes, min_max_collection_limit
        self.sets = [dict() for _ in channels]          # insertion-ordered: value (None = null, "nan" marker for NaN) -> True
        self.min_max_channels = [k for k, ch in enumerate(channels) if min_max_collection_limit > 0 and self.types[ch] != DOUBLE]
        self.mins = {} if self.min_max_channels else None  # None = not collecting min / max
        self.maxs = {}

    @staticmethod
    def _values(col):
        out = []
        for i in range(col.n):
            if col.nulls is not None and col.nulls[i]:
                out.append(None)
            elif col.type == VARCHAR:
                out.append(bytes(col.values[col.offsets[i]:col.offsets[i + 1]]).decode())
            elif col.type == DOUBLE:
                v = float(col.values[i])
                out.append("nan" if v != v else (0.0 if v == 0.0 else v))   # one NaN group, -0.0 == +0.0 (the product's group-by equality)
            elif col.type == BOOLEAN:
                out.append(bool(col.values[i]))
            else:
                out.append(int(col.values[i]))
        return out

    def _size(self, k):
        t = self.types[self.channels[k]]
        d = len(self.sets[k])
        if t == VARCHAR:
            return sum(len(v.encode()) for v in self.sets[k] if v is not None) + 5 * d
        return (np.dtype(_NP[t]).itemsize + 1) * d

    def _update_min_max(self, k, values):
        vals = [v for v in values if v is not None]
        if not vals:
            return
        key = (lambda v: v.encode()) if isinstance(vals[0], str) else (lambda v: int(v))   # Slice.compareTo: unsigned bytes; BOOLEAN false < true
        lo, hi = min(vals, key=key), max(vals, key=key)
        self.mins[k] = lo if k not in self.mins else min(self.mins[k], lo, key=key)
        self.maxs[k] = hi if k not in self.maxs else max(self.maxs[k], hi, key=key)

    def add(self, cols):
        n = cols[0].n if cols else 0
        if self.sets is None:
            if self.mins is None:
                return
            self.limit -= n
            if self.limit < 0:
                self.mins = None                                    # handleMinMaxCollectionLimitExceeded
                return
            for k in self.min_max_channels:
                self._update_min_max(k, self._values(cols[self.channels[k]]))
            return
        self.limit -= n
        for k, ch in enumerate(self.channels):
            for v in self._values(cols[ch]):
                self.sets[k].setdefault(v, True)
        if max((len(s) for s in self.sets), default=0) > self.max_distinct or sum(self._size(k) for k in range(len(self.channels))) > self.max_size:
            if not self.min_max_channels or self.limit < 0:         # handleTooLargePredicate
                self.mins = None
            else:
                for k in self.min_max_channels:
                    self._update_min_max(k, list(self.sets[k]))
            self.sets = None

    def domain(self, k):
        if self.sets is not None:
            return ("values", [v for v in self.sets[k] if v is not None and v != "nan"])
        if self.mins is None or k not in self.min_max_channels:
            return ("all",)
        if k not in self.mins:
            return ("none",)
        lo, hi = self.mins[k], self.maxs[k]
        return ("range", lo, hi) if isinstance(lo, str) else ("range", int(lo), int(hi))


# ---- join filter function -------------------------------------------------------------------------
def probe_with_filter(pages_hash, probe_key_cols, build_cols, probe_cols, flat_nodes, root, pool, probe_outer=False):
    """JoinHash.getJoinPosition / getNextJoinPosition with a JoinFilterFunction (M/operator/JoinHash.java:82-130): the positions of a
    key's chain are visited newest -> oldest and a position is eligible only if filter(build position, probe position, probe page) holds;
    a PROBE_OUTER row without an eligible position comes out once with build position -1 (LookupJoinOperator.java:354-361).
    The expression's channels [0, len(build_cols)) are the build side's, the following ones the probe page's.  Returns (probe idx, build idx)."""
    op, ob = pages_hash.probe(probe_key_cols)                    # key-equal candidates in chain order
    keep = np.zeros(len(op), dtype=bool)
    if len(op):
        def take(c, idx):
            if c.type == VARCHAR:
                items = []
                for i in idx:
                    if c.nulls is not None and c.nulls[i]:
                        items.append(None)
                    else:
                        items.append(bytes(c.values[c.offsets[i]:c.offsets[i + 1]]))
                return Col(VARCHAR, items)
            return Col(c.type, c.values[idx], None if c.nulls is None else c.nulls[idx])
        pair_cols = [take(c, ob) for c in build_cols] + [take(c, op) for c in probe_cols]
        pos = filter_positions(flat_nodes, root, pool, pair_cols)   # row-at-a-time: selected = !wasNull && value
        keep[pos] = True
    op, ob = op[keep], ob[keep]
    if not probe_outer:
        return op, ob
    n = probe_key_cols[0].n
    out_p, out_b, at = [], [], 0
    for r in range(n):
        any_match = False
        while at < len(op) and op[at] == r:
            out_p.append(r)
            out_b.append(int(ob[at]))
            at += 1
            any_match = True
        if not any_match:
            out_p.append(r)
            out_b.append(-1)
    return np.array(out_p, dtype=np.int32), np.array(out_b, dtype=np.int32)


# ---- LZ4 block format (what PagesSerde's Lz4Compressor / Lz4Decompressor exchange; io.airlift:aircompressor, un-vendored) -----------
def lz4_block_compress(src: bytes) -> bytes:
    """a plain greedy LZ4 block compressor (test infrastructure: produces valid blocks, not the reference's exact bytes): 4-byte
    hash table, matches extended forwards, the format's end-of-block rules kept (last 5 bytes literals, no match within 12 of the end)"""
    n = len(src)
    out = bytearray()
    anchor = 0

    def emit(lit_end, match_len, offset):
        lit = lit_end - anchor
        token_l = min(lit, 15)
        token_m = 0 if match_len is None else min(match_len - 4, 15)
        out.append((token_l << 4) | token_m)
        if lit >= 15:
            rest = lit - 15
            while rest >= 255:
                out.append(255)
                rest -= 255
            out.append(rest)
        out.extend(src[anchor:lit_end])
        if match_len is not None:
            out.append(offset & 0xFF)
            out.append(offset >> 8)
            if match_len - 4 >= 15:
                rest = match_len - 4 - 15
                while rest >= 255:
                    out.append(255)
                    rest -= 255
                out.append(rest)

    table = {}
    i = 0
    limit = n - 12
    while i < limit:
        key = src[i:i + 4]
        cand = table.get(key)
        table[key] = i
        if cand is not None and i - cand <= 65535:
            m = 4
            while i + m < n - 5 and src[cand + m] == src[i + m]:
                m += 1
            emit(i, m, i - cand)
            i += m
            anchor = i
        else:
            i += 1
    emit(n, None, 0)
    return bytes(out)


def lz4_block_decompress(src: bytes, n: int) -> bytes:
    out = bytearray()
    ip = 0
    while ip < len(src):
        token = src[ip]; ip += 1
        lit = token >> 4
        if lit == 15:
            while True:
                b = src[ip]; ip += 1
                lit += b
                if b != 255:
                    break
        out += src[ip:ip + lit]
        ip += lit
        if ip >= len(src):
            break
        offset = src[ip] | (src[ip + 1] << 8); ip += 2
        ml = token & 15
        if ml == 15:
            while True:
                b = src[ip]; ip += 1
                ml += b
                if b != 255:
                    break
        ml += 4
        for _ in range(ml):
            out.append(out[-offset])
    assert len(out) == n
    return bytes(out)


def compress_serialized_page(page: bytes) -> bytes:
    """PagesSerde.serialize with a compressor (PagesSerde.java:73-93): the payload as one LZ4 block, COMPRESSED marker set, sizeInBytes =
    compressed size, uncompressedSize kept -- whatever the ratio (the reference keeps the page uncompressed above 0.8)"""
    positions, markers = page[:4], page[4]
    uncompressed = int.from_bytes(page[5:9], "little")
    payload = page[13:13 + uncompressed]
    comp = lz4_block_compress(payload)
    return positions + bytes([markers | 1]) + _i32le(uncompressed) + _i32le(len(comp)) + comp
