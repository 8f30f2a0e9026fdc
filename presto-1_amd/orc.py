"""Scan-side decode, first slice: ctypes mirror of tgpu_orc_decode_* (include/tgpu.h) -- the streams of one ORC column of one
stripe / row group (decompressed bytes) -> a device-resident block.  Mirrors what io.trino.orc.reader.LongColumnReader / BooleanColumnReader /
SliceDictionaryColumnReader produce (lib/trino-orc/src/main/java/io/trino/orc/reader/)."""
import ctypes as C

from . import _lib
from .spi import OutputPage

DIRECT, DICTIONARY, DIRECT_V2, DICTIONARY_V2 = 0, 1, 2, 3


def _buf(b):
    if b is None:
        return None, 0
    b = bytes(b)
    return C.create_string_buffer(b, len(b)) if len(b) else C.create_string_buffer(1), len(b)


def decode_long_column(ctx, type_id, position_count, data, present=None, encoding=DIRECT_V2) -> OutputPage:
    pb, pl = _buf(present)
    db, dl = _buf(data)
    out = C.c_void_p()
    _lib.check(_lib.lib().tgpu_orc_decode_long_column(ctx.handle, type_id, encoding, position_count, pb, pl, db, dl, C.byref(out)))
    return OutputPage(out)


def decode_boolean_column(ctx, position_count, data, present=None) -> OutputPage:
    pb, pl = _buf(present)
    db, dl = _buf(data)
    out = C.c_void_p()
    _lib.check(_lib.lib().tgpu_orc_decode_boolean_column(ctx.handle, position_count, pb, pl, db, dl, C.byref(out)))
    return OutputPage(out)


def decode_dictionary_string_column(ctx, position_count, data, dictionary_size, length_stream, dictionary_data, present=None, encoding=DICTIONARY_V2) -> OutputPage:
    pb, pl = _buf(present)
    db, dl = _buf(data)
    lb, ll = _buf(length_stream)
    xb, xl = _buf(dictionary_data)
    out = C.c_void_p()
    _lib.check(_lib.lib().tgpu_orc_decode_dictionary_string_column(ctx.handle, encoding, position_count, pb, pl, db, dl, dictionary_size, lb, ll, xb, xl, C.byref(out)))
    return OutputPage(out)


def decode_direct_string_column(ctx, position_count, data, length_stream, present=None, encoding=DIRECT_V2) -> OutputPage:
    """SliceDirectColumnReader: LENGTH (one length per non-null row) + DATA (their bytes) -> a flat VARCHAR block"""
    pb, pl = _buf(present)
    db, dl = _buf(data)
    lb, ll = _buf(length_stream)
    out = C.c_void_p()
    _lib.check(_lib.lib().tgpu_orc_decode_direct_string_column(ctx.handle, encoding, position_count, pb, pl, db, dl, lb, ll, C.byref(out)))
    return OutputPage(out)


def decode_double_column(ctx, position_count, data, present=None) -> OutputPage:
    """DoubleColumnReader: DATA = the non-null rows' doubles (8 little-endian bytes each)"""
    pb, pl = _buf(present)
    db, dl = _buf(data)
    out = C.c_void_p()
    _lib.check(_lib.lib().tgpu_orc_decode_double_column(ctx.handle, position_count, pb, pl, db, dl, C.byref(out)))
    return OutputPage(out)
