"""ctypes wrappers of oracle/parquet_oracle.c + the page-level assembly (levels -> nulls, values / dictionary ids -> rows) and the test ENCODERS
(hybrid runs, PLAIN sections) that tests use to write pages of chosen shapes.  Test infrastructure, like the rest of oracle/."""
import ctypes as C
import struct

import numpy as np

from . import oracle as _o

BOOLEAN, INT32, INT64, DOUBLE, BYTE_ARRAY = 0, 1, 2, 5, 6
PLAIN, PLAIN_DICTIONARY, RLE, DELTA_BINARY_PACKED, DELTA_LENGTH_BYTE_ARRAY, DELTA_BYTE_ARRAY, RLE_DICTIONARY = 0, 2, 3, 5, 6, 7, 8


def _lib():
    L = _o.lib()
    if not getattr(L, "_pq_ready", False):
        i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p
        for name, res, args in (("o_pq_hybrid", i64, [vp, i64, i32, vp, i64]), ("o_pq_plain_byte_array", i64, [vp, i64, i64, vp, vp, i64]), ("o_pq_plain_boolean", i64, [vp, i64, i64, vp]),
                                ("o_pq_delta_binary_packed", i64, [vp, i64, i64, i32, vp, vp]), ("o_pq_delta_byte_array", i64, [vp, i64, i64, vp, vp, vp, i64])):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        L._pq_ready = True
    return L


def _bytes(b):
    a = np.frombuffer(bytes(b), dtype=np.uint8).copy() if len(b) else np.zeros(1, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def hybrid(data, bit_width, want):
    a, p = _bytes(data)
    out = np.zeros(max(want, 1), dtype=np.int32)
    n = _lib().o_pq_hybrid(p, len(data), bit_width, out.ctypes.data_as(C.c_void_p), want)
    if n < 0:
        raise ValueError("corrupt hybrid stream")
    return out[:n].copy()


def delta_binary_packed(physical, data, count, with_end=False):
    """`count` DELTA_BINARY_PACKED values (INT32 / INT64) as a python list (with_end: and the bytes the section takes)"""
    if physical not in (INT32, INT64):
        raise ValueError("DELTA_BINARY_PACKED is for INT32 and INT64 columns")     # ParquetEncoding.java:151
    a, p = _bytes(data)
    out = np.zeros(max(count, 1), dtype=np.int64)
    end = C.c_int64(0)
    n = _lib().o_pq_delta_binary_packed(p, len(data), count, 32 if physical == INT32 else 64, out.ctypes.data_as(C.c_void_p), C.byref(end))
    if n < 0:
        raise ValueError("corrupt DELTA_BINARY_PACKED section")
    return (out[:n].tolist(), end.value) if with_end else out[:n].tolist()


def delta_length_byte_array(data, count):
    """DELTA_LENGTH_BYTE_ARRAY (Encodings.md; ParquetEncoding.java:156-163): the lengths as a DELTA_BINARY_PACKED section, then the bytes back to back"""
    lengths, at = delta_binary_packed(INT32, data, count, with_end=True)
    out = []
    for l in lengths:
        if l < 0 or at + l > len(data):
            raise ValueError("DELTA_LENGTH_BYTE_ARRAY lengths do not fit the bytes that follow them")
        out.append(bytes(data[at:at + l]))
        at += l
    return out


def delta_byte_array(data, count, pool_cap=None):
    """DELTA_BYTE_ARRAY (Encodings.md "Delta Strings"; ParquetEncoding.java:165-173): prefix lengths, then the suffixes as DELTA_LENGTH_BYTE_ARRAY"""
    a, p = _bytes(data)
    if pool_cap is None:       # the decoded size: the sum of the prefix and suffix lengths (shared prefixes make it far larger than the section)
        prefixes, used = delta_binary_packed(INT32, data, count, with_end=True)
        pool_cap = 64 + sum(max(x, 0) for x in prefixes) + sum(max(x, 0) for x in delta_binary_packed(INT32, data[used:], count))
    cap = pool_cap
    scratch = np.zeros(max(2 * count, 1), dtype=np.int64)
    off = np.zeros(count + 1, dtype=np.int32)
    pool = np.zeros(cap, dtype=np.uint8)
    used = _lib().o_pq_delta_byte_array(p, len(data), count, scratch.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), pool.ctypes.data_as(C.c_void_p), cap)
    if used < 0:
        raise ValueError("corrupt DELTA_BYTE_ARRAY section")
    raw = pool.tobytes()
    return [raw[off[i]:off[i + 1]] for i in range(count)]


def plain_values(physical, data, count):
    """`count` PLAIN values as a python list"""
    if physical == BYTE_ARRAY:
        a, p = _bytes(data)
        off = np.zeros(count + 1, dtype=np.int32)
        pool = np.zeros(max(len(data), 1), dtype=np.uint8)
        used = _lib().o_pq_plain_byte_array(p, len(data), count, off.ctypes.data_as(C.c_void_p), pool.ctypes.data_as(C.c_void_p), len(pool))
        if used < 0:
            raise ValueError("corrupt PLAIN BYTE_ARRAY section")
        raw = pool.tobytes()
        return [raw[off[i]:off[i + 1]] for i in range(count)]
    if physical == BOOLEAN:
        a, p = _bytes(data)
        out = np.zeros(max(count, 1), dtype=np.uint8)
        if _lib().o_pq_plain_boolean(p, len(data), count, out.ctypes.data_as(C.c_void_p)) < 0:
            raise ValueError("corrupt PLAIN BOOLEAN section")
        return [bool(x) for x in out[:count]]
    dt = {INT32: "<i4", INT64: "<i8", DOUBLE: "<f8"}[physical]
    if len(data) < count * np.dtype(dt).itemsize:
        raise ValueError("PLAIN section too short")
    return np.frombuffer(bytes(data), dtype=dt, count=count).tolist()


def decode_data_page(physical, encoding, n, values, definition_levels=None, dictionary=None, dictionary_count=0):
    """one data page of a flat column as a python list (None = null): PrimitiveColumnReader.readPageV1 / initDataReader over the decoders above"""
    present = [True] * n if definition_levels is None else [bool(x) for x in hybrid(definition_levels, 1, n)]
    nn = sum(present)
    if encoding == PLAIN:
        vals = plain_values(physical, values, nn)
    elif encoding == DELTA_BINARY_PACKED:
        vals = delta_binary_packed(physical, values, nn)
    elif encoding == DELTA_LENGTH_BYTE_ARRAY:
        if physical != BYTE_ARRAY:
            raise ValueError("DELTA_LENGTH_BYTE_ARRAY is for BYTE_ARRAY columns")     # ParquetEncoding.java:160
        vals = delta_length_byte_array(values, nn)
    elif encoding == DELTA_BYTE_ARRAY:
        if physical != BYTE_ARRAY:
            raise ValueError("DELTA_BYTE_ARRAY is for BYTE_ARRAY columns here")     # (ParquetEncoding.java:170 also allows FIXED_LEN_BYTE_ARRAY: not a type of this library)
        vals = delta_byte_array(values, nn)
    elif encoding == RLE:   # ParquetEncoding.RLE for VALUES: BOOLEAN only (bit width 1), a 4-byte length in front of the hybrid stream (ParquetEncoding.java:105-115,198-212)
        if physical != BOOLEAN:
            raise ValueError("RLE value encoding is for BOOLEAN columns")
        (length,) = struct.unpack_from("<I", values, 0)
        vals = [bool(x) for x in hybrid(values[4:4 + length], 1, nn)]
    else:
        d = plain_values(physical, dictionary, dictionary_count) if dictionary_count else []   # (an all-null chunk may come without a dictionary page)
        ids = hybrid(values[1:], values[0], nn) if nn else []
        vals = [d[i] for i in ids]
    it = iter(vals)
    return [next(it) if p else None for p in present]


# ---- test encoders (the specification read backwards) -----------------------------------------------------------------------------------------
def uleb(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def hybrid_rle_run(value, count, bit_width):
    return uleb(count << 1) + int(value).to_bytes((bit_width + 7) // 8, "little")


def hybrid_packed_run(values, bit_width):
    """a bit-packed run: the values padded with zeros to a multiple of 8"""
    vals = list(values) + [0] * (-len(values) % 8)
    acc, bits, out = 0, 0, bytearray()
    for v in vals:
        acc |= (int(v) & ((1 << bit_width) - 1)) << bits
        bits += bit_width
        while bits >= 8:
            out.append(acc & 0xff)
            acc >>= 8
            bits -= 8
    return uleb(((len(vals) // 8) << 1) | 1) + bytes(out)


def hybrid_encode(values, bit_width, rng=None):
    """a hybrid stream over `values`: runs of >= 8 equal values as RLE, the rest bit-packed (in groups of 8 between RLE runs, like the writers)"""
    vals = [int(v) for v in values]
    out, i, pend = bytearray(), 0, []

    def flush(final):
        nonlocal pend
        keep = len(pend) if final else len(pend) - len(pend) % 8
        if keep:
            out.extend(hybrid_packed_run(pend[:keep], bit_width))
        pend = pend[keep:]
    while i < len(vals):
        j = i
        while j < len(vals) and vals[j] == vals[i]:
            j += 1
        if j - i >= 8 and len(pend) % 8 == 0:
            flush(False)
            out.extend(hybrid_rle_run(vals[i], j - i, bit_width))
            i = j
        else:
            pend.append(vals[i])
            i += 1
    flush(True)
    return bytes(out)


def zigzag_uleb(v):
    return uleb(((v << 1) ^ (v >> 63)) & ((1 << 64) - 1))


def delta_encode(values, physical, block_size=128, miniblocks=4):
    """a DELTA_BINARY_PACKED section over `values` (wrapping arithmetic of the physical type's width, like the writers)"""
    bits = 32 if physical == INT32 else 64
    mask = (1 << bits) - 1

    def signed(u, b=64):
        u &= (1 << b) - 1
        return u - (1 << b) if u >> (b - 1) else u
    vals = [int(v) for v in values]
    out = bytearray(uleb(block_size) + uleb(miniblocks) + uleb(len(vals)) + zigzag_uleb(vals[0] if vals else 0))
    mini = block_size // miniblocks
    deltas = [signed((vals[i] - vals[i - 1]) & mask, bits) for i in range(1, len(vals))]
    for b0 in range(0, len(deltas), block_size):
        block = deltas[b0:b0 + block_size]
        md = min(block)
        out += zigzag_uleb(md)
        rel = [(d - md) & ((1 << 64) - 1) for d in block]
        widths, chunks = [], []
        for m in range(miniblocks):
            part = rel[m * mini:(m + 1) * mini]
            if not part:
                widths.append(0)
                continue
            w = max(x.bit_length() for x in part)
            widths.append(w)
            part = part + [0] * (mini - len(part))
            acc, nb, by = 0, 0, bytearray()
            for x in part:
                acc |= x << nb
                nb += w
                while nb >= 8:
                    by.append(acc & 0xff)
                    acc >>= 8
                    nb -= 8
            chunks.append(bytes(by))
        out += bytes(widths) + b"".join(chunks)
    return bytes(out)


def delta_length_encode(values):
    return delta_encode([len(b) for b in values], INT32) + b"".join(bytes(b) for b in values)


def delta_byte_array_encode(values):
    """prefix = the longest common prefix with the value before (what the writers do); any prefix <= that would be valid"""
    prefixes, suffixes, prev = [], [], b""
    for v in values:
        v = bytes(v)
        k = 0
        while k < min(len(prev), len(v)) and prev[k] == v[k]:
            k += 1
        prefixes.append(k)
        suffixes.append(v[k:])
        prev = v
    return delta_encode(prefixes, INT32) + delta_length_encode(suffixes)


def plain_encode(physical, values):
    if physical == BYTE_ARRAY:
        return b"".join(struct.pack("<I", len(b)) + bytes(b) for b in values)
    if physical == BOOLEAN:
        bits = list(values) + [0] * (-len(values) % 8)
        return bytes(sum((1 if bits[i + k] else 0) << k for k in range(8)) for i in range(0, len(bits), 8))
    dt = {INT32: "<i4", INT64: "<i8", DOUBLE: "<f8"}[physical]
    return np.asarray(values, dtype=dt).tobytes()
