"""ctypes wrappers of oracle/orc_oracle.c (the ORC stream decoders' CPU restatement).  Test infrastructure, like the rest of oracle/."""
import ctypes as C

import numpy as np

from . import oracle as _o


def _lib():
    L = _o.lib()
    if not getattr(L, "_orc_ready", False):
        i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p
        for name, res, args in (("o_orc_decode_bit_width", i32, [i32]), ("o_orc_closest_fixed_bits", i32, [i32]), ("o_orc_read_vint", i64, [vp, i64, i32, vp]),
                                ("o_orc_write_vlong", i32, [i64, i32, vp]), ("o_orc_unpack", i64, [vp, i64, i64, i32, vp]), ("o_orc_rle_v2", i64, [vp, i64, i32, vp, i64]),
                                ("o_orc_rle_v1", i64, [vp, i64, i32, vp, i64]), ("o_orc_byte_rle", i64, [vp, i64, vp, i64]), ("o_orc_boolean", i64, [vp, i64, i64, vp])):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        L._orc_ready = True
    return L


def _bytes(b):
    a = np.frombuffer(bytes(b), dtype=np.uint8).copy() if len(b) else np.zeros(1, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def rle_v2(data, signed, cap=1 << 22):
    a, p = _bytes(data)
    out = np.zeros(cap, dtype=np.int64)
    n = _lib().o_orc_rle_v2(p, len(data), int(signed), out.ctypes.data_as(C.c_void_p), cap)
    if n < 0:
        raise ValueError(f"corrupt RLEv2 stream ({n})")
    return out[:n].copy()


def rle_v1(data, signed, cap=1 << 22):
    a, p = _bytes(data)
    out = np.zeros(cap, dtype=np.int64)
    n = _lib().o_orc_rle_v1(p, len(data), int(signed), out.ctypes.data_as(C.c_void_p), cap)
    if n < 0:
        raise ValueError(f"corrupt RLEv1 stream ({n})")
    return out[:n].copy()


def byte_rle(data, cap=1 << 22):
    a, p = _bytes(data)
    out = np.zeros(cap, dtype=np.uint8)
    n = _lib().o_orc_byte_rle(p, len(data), out.ctypes.data_as(C.c_void_p), cap)
    if n < 0:
        raise ValueError(f"corrupt byte RLE stream ({n})")
    return out[:n].copy()


def boolean(data, count):
    a, p = _bytes(data)
    out = np.zeros(max(count, 1), dtype=np.uint8)
    n = _lib().o_orc_boolean(p, len(data), count, out.ctypes.data_as(C.c_void_p))
    if n < 0:
        raise ValueError(f"corrupt boolean stream ({n})")
    return out[:count].copy()


def unpack(data, count, bit_size):
    """(values, bytes read) of LongBitPacker.unpackGeneric"""
    a, p = _bytes(data)
    out = np.zeros(max(count, 1), dtype=np.int64)
    read = _lib().o_orc_unpack(p, len(data), count, bit_size, out.ctypes.data_as(C.c_void_p))
    return out[:count].copy(), read


def write_vlong(value, signed):
    buf = (C.c_uint8 * 10)()
    n = _lib().o_orc_write_vlong(C.c_int64(value), int(signed), buf)
    return bytes(buf[:n])


def read_vint(data, signed):
    a, p = _bytes(data)
    used = C.c_int64()
    v = _lib().o_orc_read_vint(p, len(data), int(signed), C.byref(used))
    return int(v), used.value


def decode_bit_width(n):
    return _lib().o_orc_decode_bit_width(n)


def closest_fixed_bits(w):
    return _lib().o_orc_closest_fixed_bits(w)


# ---- writers (test data generators; follow the public ORC specification -- the reference's writer is not restated) --------------------
def pack_bits(values, bit_size):
    """big-endian bit packing, the inverse of unpackGeneric"""
    acc, nbits, out = 0, 0, bytearray()
    for v in values:
        acc = (acc << bit_size) | (int(v) & ((1 << bit_size) - 1))
        nbits += bit_size
        while nbits >= 8:
            nbits -= 8
            out.append((acc >> nbits) & 0xff)
            acc &= (1 << nbits) - 1
    if nbits:
        out.append((acc << (8 - nbits)) & 0xff)
    return bytes(out)


_WIDTH_CODE = {**{w: w - 1 for w in range(1, 25)}, 26: 24, 28: 25, 30: 26, 32: 27, 40: 28, 48: 29, 56: 30, 64: 31}


def zigzag(v):
    return ((v << 1) ^ (v >> 63)) & ((1 << 64) - 1)


def rle_v2_direct(values, signed, width=None):
    """one DIRECT run (<= 512 values)"""
    vals = [zigzag(int(v)) if signed else int(v) & ((1 << 64) - 1) for v in values]
    need = max(1, max(vals).bit_length()) if vals else 1
    width = width or closest_fixed_bits(need)
    n = len(vals) - 1
    return bytes([0x40 | (_WIDTH_CODE[width] << 1) | (n >> 8), n & 0xff]) + pack_bits(vals, width)


def rle_v2_short_repeat(value, count, signed):
    v = zigzag(int(value)) if signed else int(value)
    size = max(1, (v.bit_length() + 7) // 8)
    return bytes([((size - 1) << 3) | (count - 3)]) + v.to_bytes(size, "big")


def rle_v2_delta(first, deltas, signed):
    """one DELTA run: first value + deltas of one sign (or all equal: fixed delta)"""
    out = bytearray()
    n = len(deltas)
    if len(set(deltas)) == 1:
        out += bytes([0xc0 | (n >> 8), n & 0xff]) + write_vlong(first, signed) + write_vlong(deltas[0], True)
        return bytes(out)
    mags = [abs(d) for d in deltas[1:]]
    width = closest_fixed_bits(max(1, max(mags).bit_length())) if mags else 1
    if width == 1:
        width = 2      # (width code 0 means "fixed delta")
    out += bytes([0xc0 | (_WIDTH_CODE[width] << 1) | (n >> 8), n & 0xff]) + write_vlong(first, signed) + write_vlong(deltas[0], True) + pack_bits(mags, width)
    return bytes(out)


def rle_v2_patched_base(values, base, fb, patch_width, patches):
    """one PATCHED_BASE run: values = base + (low fb bits | patch << fb at the patched positions); patches = [(gap, patch value)] with gaps <= 255"""
    n = len(values) - 1
    bw = max(1, (abs(base).bit_length() + 8) // 8)
    bb = abs(base) | ((1 << (bw * 8 - 1)) if base < 0 else 0)
    pgw = max(1, max(g for g, _ in patches).bit_length())
    out = bytearray([0x80 | (_WIDTH_CODE[fb] << 1) | (n >> 8), n & 0xff, ((bw - 1) << 5) | _WIDTH_CODE[patch_width], ((pgw - 1) << 5) | len(patches)])
    out += bb.to_bytes(bw, "big")
    out += pack_bits([(int(v) - base) & ((1 << fb) - 1) for v in values], fb)
    out += pack_bits([(g << patch_width) | p for g, p in patches], closest_fixed_bits(patch_width + pgw))
    return bytes(out)


def rle_v1_encode(values, signed):
    """an RLEv1 stream (LongInputStreamV1.java:47-103 read backwards): runs of 3..130 values with a constant delta in [-128, 127], else literal
    groups of up to 128 varints"""
    vals = [int(v) for v in values]
    out, lit, i = bytearray(), [], 0

    def flush():
        nonlocal lit
        while lit:
            chunk, lit = lit[:128], lit[128:]
            out.append(0x100 - len(chunk))
            for v in chunk:
                out.extend(write_vlong(v, signed))
    while i < len(vals):
        j = i + 1
        if j < len(vals) and -128 <= vals[j] - vals[i] <= 127:
            d = vals[j] - vals[i]
            while j + 1 < len(vals) and j + 1 - i < 130 and vals[j + 1] - vals[j] == d:
                j += 1
            if j + 1 - i >= 3:
                flush()
                out.append(j + 1 - i - 3)
                out.append(d & 0xff)
                out.extend(write_vlong(vals[i], signed))
                i = j + 1
                continue
        lit.append(vals[i])
        i += 1
    flush()
    return bytes(out)


def byte_rle_encode(data):
    out, i = bytearray(), 0
    data = bytes(data)
    while i < len(data):
        j = i
        while j < len(data) and data[j] == data[i] and j - i < 130:
            j += 1
        if j - i >= 3:
            out += bytes([j - i - 3, data[i]])
            i = j
            continue
        j = i
        while j < len(data) and j - i < 128 and not (j + 2 < len(data) and data[j] == data[j + 1] == data[j + 2]):
            j += 1
        out += bytes([0x100 - (j - i)]) + data[i:j]
        i = j
    return bytes(out)


def boolean_encode(bits):
    bits = list(bits)
    by = bytearray()
    for i in range(0, len(bits), 8):
        chunk = bits[i:i + 8] + [0] * (8 - len(bits[i:i + 8]))
        by.append(sum(b << (7 - k) for k, b in enumerate(chunk)))
    return byte_rle_encode(by)
