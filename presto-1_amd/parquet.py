"""ctypes mirror of tgpu_parquet_decode_data_page (include/tgpu.h): one data page of a flat Parquet column -> a device-resident one-channel page."""
import ctypes as C

import numpy as np

from . import _lib
from .spi import OutputPage

BOOLEAN, INT32, INT64, DOUBLE, BYTE_ARRAY = 0, 1, 2, 5, 6          # parquet.thrift Type
PLAIN, PLAIN_DICTIONARY, RLE, DELTA_BINARY_PACKED, DELTA_LENGTH_BYTE_ARRAY, DELTA_BYTE_ARRAY, RLE_DICTIONARY = 0, 2, 3, 5, 6, 7, 8   # parquet.thrift Encoding (RLE as a value encoding: BOOLEAN; DELTA_BINARY_PACKED: INT32 / INT64; DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY: BYTE_ARRAY)


def _buf(b):
    if b is None:
        return None, 0
    a = np.frombuffer(bytes(b), dtype=np.uint8) if len(b) else np.zeros(1, dtype=np.uint8)
    return a.ctypes.data_as(C.c_void_p), len(b)


def decode_data_page(ctx, type_id, physical, encoding, position_count, values, definition_levels=None, dictionary=None, dictionary_count=0) -> OutputPage:
    """PrimitiveColumnReader.readPageV1 / readPageV2 for a flat column: definition levels (hybrid of bit width 1, no length prefix; None = required
    column) + the value section (+ the chunk's PLAIN dictionary page for the dictionary encodings)"""
    keep = [np.frombuffer(bytes(b), dtype=np.uint8) if b is not None and len(b) else None for b in (definition_levels, values, dictionary)]
    ptr = [k.ctypes.data_as(C.c_void_p) if k is not None else None for k in keep]
    ln = [0 if b is None else len(b) for b in (definition_levels, values, dictionary)]
    out = C.c_void_p()
    _lib.check(_lib.lib().tgpu_parquet_decode_data_page(ctx.handle, type_id, physical, encoding, position_count, ptr[0], ln[0], ptr[1], ln[1], ptr[2], ln[2], dictionary_count, C.byref(out)))
    return OutputPage(out)
