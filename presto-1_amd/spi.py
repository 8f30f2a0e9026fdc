"""Host-side mirror of the reference's columnar data model (io.trino.spi.Page / io.trino.spi.block.*).

Blocks are numpy-backed, exactly the arrays the Java blocks hold: `values`, one null byte per position
(`valueIsNull`, S/block/LongArrayBlock.java:38-41), `offsets` for VARCHAR (S/block/VariableWidthBlock.java:38-43),
ids + dictionary (S/block/DictionaryBlock.java:40-100), single value x n (S/block/RunLengthEncodedBlock.java:30-70).
A DeviceBlock wraps memory that is already in HBM (e.g. a torch tensor's data_ptr) for the device-resident benches.
"""
import ctypes as C

import numpy as np

from . import _lib

BIGINT, INTEGER, DATE, DOUBLE, BOOLEAN, VARCHAR = 1, 2, 3, 4, 5, 6
TYPE_NAMES = {BIGINT: "bigint", INTEGER: "integer", DATE: "date", DOUBLE: "double", BOOLEAN: "boolean", VARCHAR: "varchar"}
NP_DTYPE = {BIGINT: np.int64, INTEGER: np.int32, DATE: np.int32, DOUBLE: np.float64, BOOLEAN: np.uint8}
FLAT, DICTIONARY, RLE = 0, 1, 2
HOST, DEVICE = 0, 1


def _ptr(a):
    return None if a is None else a.ctypes.data


class Block:
    """Flat block of any supported type (LongArrayBlock / IntArrayBlock / ByteArrayBlock / VariableWidthBlock)."""

    encoding = FLAT

    def __init__(self, type_id, values, nulls=None, offsets=None):
        self.type = type_id
        if type_id == VARCHAR:
            if offsets is None:
                items = list(values)
                if nulls is None and any(v is None for v in items):
                    nulls = np.array([v is None for v in items], dtype=np.uint8)
                bs = [b"" if v is None else (v.encode("utf-8") if isinstance(v, str) else bytes(v)) for v in items]
                offsets = np.zeros(len(bs) + 1, dtype=np.int32)
                if bs:
                    offsets[1:] = np.cumsum([len(b) for b in bs])
                joined = b"".join(bs)
                values = np.frombuffer(joined, dtype=np.uint8).copy() if joined else np.zeros(1, dtype=np.uint8)
            self.values = np.ascontiguousarray(values, dtype=np.uint8)
            self.offsets = np.ascontiguousarray(offsets, dtype=np.int32)
            self.position_count = len(self.offsets) - 1
        else:
            if nulls is None and not isinstance(values, np.ndarray):
                items = list(values)
                if any(v is None for v in items):
                    nulls = np.array([v is None for v in items], dtype=np.uint8)
                    values = [0 if v is None else v for v in items]
                else:
                    values = items
            self.values = np.ascontiguousarray(values, dtype=NP_DTYPE[type_id])
            self.offsets = None
            self.position_count = len(self.values)
        self.nulls = None if nulls is None else np.ascontiguousarray(nulls, dtype=np.uint8)
        if self.nulls is not None:
            assert len(self.nulls) == self.position_count

    # -- io.trino.spi.block.Block accessors used by tests --
    def getPositionCount(self):
        return self.position_count

    def isNull(self, position):
        return bool(self.nulls is not None and self.nulls[position])

    def get(self, position):
        if self.isNull(position):
            return None
        if self.type == VARCHAR:
            return bytes(self.values[self.offsets[position]:self.offsets[position + 1]]).decode("utf-8", "replace")
        v = self.values[position]
        if self.type == BOOLEAN:
            return bool(v)
        if self.type == DOUBLE:
            return float(v)
        return int(v)

    def to_list(self):
        return [self.get(i) for i in range(self.position_count)]

    def flatten(self):
        return self

    def _fill(self, s: _lib.Block, keep):
        s.type, s.encoding, s.memory, s.position_count = self.type, FLAT, HOST, self.position_count
        s.values, s.nulls, s.offsets = _ptr(self.values), _ptr(self.nulls), _ptr(self.offsets)
        s.ids = None
        keep.append(self)


class DictionaryBlock:
    encoding = DICTIONARY

    def __init__(self, dictionary: Block, ids):
        self.dictionary = dictionary
        self.ids = np.ascontiguousarray(ids, dtype=np.int32)
        self.type = dictionary.type
        self.position_count = len(self.ids)

    def getPositionCount(self):
        return self.position_count

    def flatten(self):
        d = self.dictionary.flatten()
        vals = [d.get(int(i)) for i in self.ids]
        return Block(self.type, vals)

    def to_list(self):
        return self.flatten().to_list()

    def _fill(self, s, keep):
        ds = _lib.Block()
        self.dictionary._fill(ds, keep)
        keep.append(ds)
        s.type, s.encoding, s.memory, s.position_count = self.type, DICTIONARY, HOST, self.position_count
        s.values = s.nulls = s.offsets = None
        s.ids = _ptr(self.ids)
        s.dictionary = C.pointer(ds)
        keep.append(self)


class RunLengthEncodedBlock:
    encoding = RLE

    def __init__(self, value: Block, position_count):
        assert value.position_count == 1
        self.value = value
        self.type = value.type
        self.position_count = position_count

    def getPositionCount(self):
        return self.position_count

    def flatten(self):
        return Block(self.type, [self.value.get(0)] * self.position_count)

    def to_list(self):
        return self.flatten().to_list()

    def _fill(self, s, keep):
        ds = _lib.Block()
        self.value._fill(ds, keep)
        keep.append(ds)
        s.type, s.encoding, s.memory, s.position_count = self.type, RLE, HOST, self.position_count
        s.values = s.nulls = s.offsets = s.ids = None
        s.dictionary = C.pointer(ds)
        keep.append(self)


class LazyBlock:
    """io.trino.spi.block.LazyBlock (S/block/LazyBlock.java): `loader()` returns the loaded block; only meaningful in pages a page source hands to a
    ScanFilterAndProjectOperator (tgpu_page_source.load_block)"""

    encoding = 3   # TGPU_LAZY

    def __init__(self, type_id, position_count, loader):
        self.type, self.position_count, self.loader = type_id, position_count, loader
        self.loaded = None

    def getPositionCount(self):
        return self.position_count

    def isLoaded(self):
        return self.loaded is not None

    def load(self):
        if self.loaded is None:
            self.loaded = self.loader()
        return self.loaded

    def _fill(self, s, keep):
        s.type, s.encoding, s.memory, s.position_count = self.type, 3, HOST, self.position_count
        s.values = s.nulls = s.offsets = s.ids = None
        keep.append(self)


class DeviceBlock:
    """A flat block whose arrays already live in HBM.  `values`/`nulls`/`offsets` are objects exposing data_ptr()
    (torch tensors) or plain integer device addresses; the owner keeps them alive."""

    encoding = FLAT

    def __init__(self, type_id, position_count, values, nulls=None, offsets=None):
        self.type = type_id
        self.position_count = position_count
        self.values, self.nulls, self.offsets = values, nulls, offsets

    @staticmethod
    def _addr(x):
        if x is None:
            return None
        return x.data_ptr() if hasattr(x, "data_ptr") else int(x)

    def getPositionCount(self):
        return self.position_count

    def _fill(self, s, keep):
        s.type, s.encoding, s.memory, s.position_count = self.type, FLAT, DEVICE, self.position_count
        s.values, s.nulls, s.offsets = self._addr(self.values), self._addr(self.nulls), self._addr(self.offsets)
        s.ids = None
        keep.append(self)


class Page:
    """io.trino.spi.Page (S/Page.java:33-73): blocks + positionCount."""

    def __init__(self, *blocks, position_count=None):
        if len(blocks) == 1 and isinstance(blocks[0], (list, tuple)):
            blocks = tuple(blocks[0])
        self.blocks = list(blocks)
        if position_count is None:
            position_count = self.blocks[0].position_count if self.blocks else 0
        self.position_count = position_count
        for b in self.blocks:
            assert b.position_count == self.position_count, "block position counts differ"

    def getPositionCount(self):
        return self.position_count

    def getChannelCount(self):
        return len(self.blocks)

    def getBlock(self, channel):
        return self.blocks[channel]

    def appendColumn(self, block):
        return Page(*(self.blocks + [block]))

    def rows(self):
        cols = [b.to_list() for b in self.blocks]
        return [tuple(c[i] for c in cols) for i in range(self.position_count)]

    def to_c(self):
        """-> (ctypes Page, keep-alive list)."""
        keep = []
        arr = (_lib.Block * max(1, len(self.blocks)))()
        for i, b in enumerate(self.blocks):
            b._fill(arr[i], keep)
        keep.append(arr)
        p = _lib.Page(self.position_count, len(self.blocks), arr)
        return p, keep


class OutputPage:
    """A device-resident page returned by an operator (tgpu_output_page)."""

    def __init__(self, handle):
        self.handle = handle
        self._as_page_keep = None

    @property
    def position_count(self):
        return _lib.lib().tgpu_output_page_position_count(self.handle)

    @property
    def channel_count(self):
        return _lib.lib().tgpu_output_page_channel_count(self.handle)

    def to_host(self) -> Page:
        L = _lib.lib()
        n = self.position_count
        nch = self.channel_count
        info, bufs = [], []
        vptr, nptr, optr = (C.c_void_p * max(nch, 1))(), (C.c_void_p * max(nch, 1))(), (C.c_void_p * max(nch, 1))()
        for ch in range(nch):
            t, vb, mn = C.c_int32(), C.c_int64(), C.c_int32()
            _lib.check(L.tgpu_output_page_block_info(self.handle, ch, C.byref(t), C.byref(vb), C.byref(mn)))
            nulls = np.zeros(max(n, 1), dtype=np.uint8)
            if t.value == VARCHAR:
                values = np.zeros(max(vb.value, 1), dtype=np.uint8)
                offsets = np.zeros(n + 1, dtype=np.int32)
                optr[ch] = offsets.ctypes.data
            else:
                values = np.zeros(max(n, 1), dtype=NP_DTYPE[t.value])
                offsets = None
            vptr[ch], nptr[ch] = values.ctypes.data, nulls.ctypes.data
            info.append((t.value, mn.value))
            bufs.append((values, nulls, offsets))
        _lib.check(L.tgpu_output_page_copy_blocks(self.handle, nch, vptr, nptr, optr))   # one stream synchronisation for the page
        blocks = []
        for (t, mn), (values, nulls, offsets) in zip(info, bufs):
            if t == VARCHAR:
                blocks.append(Block(VARCHAR, values, nulls[:n] if mn else None, offsets))
            else:
                blocks.append(Block(t, values[:n], nulls[:n] if mn else None))
        return Page(*blocks, position_count=n)

    def as_device_page(self) -> Page:
        """Zero-copy view for feeding the next GPU operator (valid until release())."""
        cp = _lib.Page()
        _lib.check(_lib.lib().tgpu_output_page_as_page(self.handle, C.byref(cp)))
        blocks = []
        for i in range(cp.channel_count):
            b = cp.blocks[i]
            blocks.append(DeviceBlock(b.type, b.position_count, b.values, b.nulls, b.offsets))
        pg = Page(*blocks, position_count=cp.position_count)
        pg._owner = self
        return pg

    def release(self):
        if self.handle:
            _lib.lib().tgpu_output_page_release(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:   # interpreter shutdown: module globals may already be gone
            pass
