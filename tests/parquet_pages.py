"""Test helper: walks a Parquet file down to the page payloads with a thrift-compact reader of its own (public Parquet format specification; nothing of
the reference or of parquet-mr is used).  Only what the parity tests need: flat schemas, UNCOMPRESSED column chunks.

    for chunk in column_chunks(path): chunk["physical"], chunk["optional"], chunk["pages"] = [{"kind": "DICTIONARY" | "DATA_V1" | "DATA_V2", ...}]"""
import struct

BOOLEAN, INT32, INT64, INT96, FLOAT, DOUBLE, BYTE_ARRAY, FIXED_LEN_BYTE_ARRAY = range(8)          # parquet.thrift Type
PLAIN, PLAIN_DICTIONARY, RLE, BIT_PACKED, RLE_DICTIONARY = 0, 2, 3, 4, 8                           # parquet.thrift Encoding


class Thrift:
    """thrift compact protocol: a struct comes back as {field id: value}; nested structs / lists recursively"""

    def __init__(self, data, at=0):
        self.b, self.at = data, at

    def varint(self):
        v, s = 0, 0
        while True:
            x = self.b[self.at]
            self.at += 1
            v |= (x & 0x7f) << s
            s += 7
            if not x & 0x80:
                return v

    def zigzag(self):
        v = self.varint()
        return (v >> 1) ^ -(v & 1)

    def value(self, t):
        if t in (1, 2):
            return t == 1
        if t == 3:
            self.at += 1
            return self.b[self.at - 1]
        if t in (4, 5, 6):
            return self.zigzag()
        if t == 7:
            self.at += 8
            return struct.unpack_from("<d", self.b, self.at - 8)[0]
        if t == 8:
            n = self.varint()
            self.at += n
            return bytes(self.b[self.at - n:self.at])
        if t in (9, 10):
            h = self.b[self.at]
            self.at += 1
            n, et = h >> 4, h & 15
            if n == 15:
                n = self.varint()
            if et in (1, 2):   # a list of bools: one byte each
                out = [self.b[self.at + i] == 1 for i in range(n)]
                self.at += n
                return out
            return [self.value(et) for _ in range(n)]
        if t == 12:
            return self.struct()
        raise ValueError(f"thrift type {t}")

    def struct(self):
        out, fid = {}, 0
        while True:
            h = self.b[self.at]
            self.at += 1
            if h == 0:
                return out
            delta, t = h >> 4, h & 15
            fid = fid + delta if delta else self.zigzag()
            out[fid] = self.value(t)


def column_chunks(path):
    data = open(path, "rb").read()
    assert data[:4] == b"PAR1" and data[-4:] == b"PAR1", "not a Parquet file"
    flen = struct.unpack_from("<I", data, len(data) - 8)[0]
    meta = Thrift(data, len(data) - 8 - flen).struct()
    schema = meta[2]
    leaves = [e for e in schema[1:] if 5 not in e or not e[5]]            # flat schema: every element after the root is a leaf
    assert len(leaves) == len(schema) - 1, "nested schemas are not walked by this helper"
    out = []
    for rg in meta[4]:
        for ci, cc in enumerate(rg[1]):
            md = cc[3]
            assert md[4] == 0, "only UNCOMPRESSED column chunks"
            at = md.get(11) or md[9]
            if md.get(11) is not None and md[11] > 0:
                at = min(md[11], md[9])
            end, pages, seen = at + md[7], [], 0
            while at < end and seen < md[5]:
                t = Thrift(data, at)
                ph = t.struct()
                body = data[t.at:t.at + ph[3]]
                at = t.at + ph[3]
                if ph[1] == 2:
                    pages.append({"kind": "DICTIONARY", "num_values": ph[7][1], "encoding": ph[7][2], "bytes": body})
                elif ph[1] == 0:
                    h = ph[5]
                    pages.append({"kind": "DATA_V1", "num_values": h[1], "encoding": h[2], "definition_level_encoding": h[3], "bytes": body})
                    seen += h[1]
                elif ph[1] == 3:
                    h = ph[8]
                    pages.append({"kind": "DATA_V2", "num_values": h[1], "num_nulls": h[2], "encoding": h[4], "definition_levels_byte_length": h[5],
                                  "repetition_levels_byte_length": h[6], "bytes": body})
                    seen += h[1]
            out.append({"name": leaves[ci][4].decode(), "physical": md[1], "optional": leaves[ci].get(3, 0) == 1, "row_group_rows": rg[3], "num_values": md[5], "pages": pages})
    return out


def split_data_page(chunk, page):
    """(definition-level bytes or None, value bytes) of a data page of a FLAT column: V1 pages carry the levels as a 4-byte length + RLE hybrid
    in front of the values (only when the column is optional), V2 pages next to them with the length in the header"""
    b = page["bytes"]
    if page["kind"] == "DATA_V1":
        if not chunk["optional"]:
            return None, b
        n = struct.unpack_from("<I", b, 0)[0]
        return b[4:4 + n], b[4 + n:]
    rl, dl = page["repetition_levels_byte_length"], page["definition_levels_byte_length"]
    return (b[rl:rl + dl] if chunk["optional"] else None), b[rl + dl:]
