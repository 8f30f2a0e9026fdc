"""Build-container script: turns the ORC files the reference holds as test resources (lib/trino-orc/src/test/resources/*.orc) into
tests/golden/orc_streams.json -- DATA, not source: for every column of every stripe the DECOMPRESSED bytes of its streams (what
io.trino.orc.stream.LongInputStreamV2 / LongInputStreamV1 / BooleanInputStream read) together with what the file's WRITER recorded about the
column (ORC column statistics of the stripe: number of values, has-null, integer min / max / sum) and what the reference's tests assert
about the file (row counts, compression kind: TestOrcLz4.java:45-46, TestOrcWithoutRowGroupInfo.java:62-63).  The statistics are the
known answers the oracle's stream decoders are pinned on: they were computed by the writer (Apache ORC / Hive), not by anything in this
repository.

Nothing of the reference's code is used or copied: the container format is walked with a minimal protobuf varint reader following the
public ORC specification (postscript -> footer -> metadata -> stripe footers; compression chunk headers); LZ4 blocks are inflated with the
oracle's block coder.   python tools/extract_orc_fixtures.py   (needs /root/reference; writes tests/golden/orc_streams.json)"""
import base64
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RES = "/root/reference/lib/trino-orc/src/test/resources"


def varint(b, at):
    v, s = 0, 0
    while True:
        x = b[at]
        at += 1
        v |= (x & 0x7f) << s
        s += 7
        if not x & 0x80:
            return v, at


def fields(b):
    """(field number, wire type, value) of one protobuf message; length-delimited values as bytes"""
    at, out = 0, []
    while at < len(b):
        key, at = varint(b, at)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, at = varint(b, at)
        elif wt == 2:
            n, at = varint(b, at)
            v = b[at:at + n]
            at += n
        elif wt == 1:
            v = b[at:at + 8]
            at += 8
        elif wt == 5:
            v = b[at:at + 4]
            at += 4
        else:
            raise ValueError(wt)
        out.append((f, wt, v))
    return out


def zigzag(v):
    return (v >> 1) ^ -(v & 1)


def first(fs, f, default=None):
    for k, _, v in fs:
        if k == f:
            return v
    return default


def inflate(buf, kind, block_size):
    """ORC compression framing: 3-byte little-endian header per chunk = (length << 1) | isOriginal"""
    if kind == 0:
        return bytes(buf)
    from oracle import oracle
    out, at = bytearray(), 0
    while at < len(buf):
        h = buf[at] | (buf[at + 1] << 8) | (buf[at + 2] << 16)
        at += 3
        n, original = h >> 1, h & 1
        chunk = bytes(buf[at:at + n])
        at += n
        if original:
            out += chunk
        elif kind == 1:
            out += zlib.decompress(chunk, -15)
        elif kind == 4:
            out += lz4_raw(chunk, block_size)
        else:
            raise ValueError(f"compression kind {kind} not handled")
    return bytes(out)


def lz4_raw(src, limit):
    """an LZ4 block whose uncompressed size is not stored: decode until the input is used up (ORC: at most compressionBlockSize bytes)"""
    out, i = bytearray(), 0
    while i < len(src):
        tok = src[i]
        i += 1
        ll = tok >> 4
        if ll == 15:
            while True:
                x = src[i]
                i += 1
                ll += x
                if x != 255:
                    break
        out += src[i:i + ll]
        i += ll
        if i >= len(src):
            break
        off = src[i] | (src[i + 1] << 8)
        i += 2
        ml = tok & 15
        if ml == 15:
            while True:
                x = src[i]
                i += 1
                ml += x
                if x != 255:
                    break
        ml += 4
        start = len(out) - off
        for k in range(ml):
            out.append(out[start + k])
    assert len(out) <= limit
    return bytes(out)


KINDS = ["PRESENT", "DATA", "LENGTH", "DICTIONARY_DATA", "DICTIONARY_COUNT", "SECONDARY", "ROW_INDEX", "BLOOM_FILTER", "BLOOM_FILTER_UTF8"]
ENCODINGS = ["DIRECT", "DICTIONARY", "DIRECT_V2", "DICTIONARY_V2"]
TYPES = ["BOOLEAN", "BYTE", "SHORT", "INT", "LONG", "FLOAT", "DOUBLE", "STRING", "BINARY", "TIMESTAMP", "LIST", "MAP", "STRUCT", "UNION", "DECIMAL", "DATE", "VARCHAR", "CHAR"]


def column_stats(msg):
    fs = fields(msg)
    st = {"number_of_values": first(fs, 1, 0), "has_null": bool(first(fs, 10, 0))}
    ints = first(fs, 2)
    if ints is not None:
        i = fields(ints)
        st["int"] = {"min": zigzag(first(i, 1, 0)), "max": zigzag(first(i, 2, 0)), "sum": zigzag(first(i, 3)) if first(i, 3) is not None else None}
    return st


def extract(path, asserted):
    d = open(path, "rb").read()
    ps_len = d[-1]
    ps = fields(d[-1 - ps_len:-1])
    footer_len, kind, block = first(ps, 1), first(ps, 2, 0), first(ps, 3, 256 * 1024)
    meta_len = first(ps, 5, 0)
    footer = fields(inflate(d[-1 - ps_len - footer_len:-1 - ps_len], kind, block))
    meta = fields(inflate(d[-1 - ps_len - footer_len - meta_len:-1 - ps_len - footer_len], kind, block)) if meta_len else []
    types = [TYPES[first(fields(v), 1, 0)] for f, _, v in footer if f == 4]
    stripe_stats = [[column_stats(c) for f2, _, c in fields(v) if f2 == 1] for f, _, v in meta if f == 1]
    out = {"file": os.path.basename(path), "compression_kind": kind, "number_of_rows": first(footer, 6, 0), "asserted_by": asserted, "column_types": types,
           "file_statistics": [column_stats(v) for f, _, v in footer if f == 7], "stripes": []}
    for si, (f, _, v) in enumerate([x for x in footer if x[0] == 3]):
        s = fields(v)
        off, ilen, dlen, flen, rows = first(s, 1, 0), first(s, 2, 0), first(s, 3, 0), first(s, 4, 0), first(s, 5, 0)
        sf = fields(inflate(d[off + ilen + dlen:off + ilen + dlen + flen], kind, block))
        encs = []
        for f2, _, c in sf:
            if f2 == 2:
                e = fields(c)
                encs.append({"kind": ENCODINGS[first(e, 1, 0)], "dictionary_size": first(e, 2, 0)})
        at, streams = off, []
        for f2, _, c in sf:
            if f2 != 1:
                continue
            m = fields(c)
            k, col, ln = first(m, 1, 0), first(m, 2, 0), first(m, 3, 0)
            raw = d[at:at + ln]
            at += ln
            if KINDS[k] in ("ROW_INDEX", "BLOOM_FILTER", "BLOOM_FILTER_UTF8"):
                continue
            streams.append({"column": col, "kind": KINDS[k], "bytes": base64.b64encode(inflate(raw, kind, block)).decode()})
        out["stripes"].append({"rows": rows, "encodings": encs, "streams": streams, "statistics": stripe_stats[si] if si < len(stripe_stats) else None})
    return out


if __name__ == "__main__":
    files = [("apache-lz4.orc", "lib/trino-orc/src/test/java/io/trino/orc/TestOrcLz4.java:45-46 (compression LZ4, 10_000 rows)"),
             ("orcFileWithoutRowGroupInfo.orc", "lib/trino-orc/src/test/java/io/trino/orc/TestOrcWithoutRowGroupInfo.java:62-63 (2 rows, no row-group stride)")]
    res = {"_comment": "made by tools/extract_orc_fixtures.py from the reference's ORC test resources: decompressed stream bytes + the writer's own column statistics",
           "files": [extract(os.path.join(RES, n), a) for n, a in files]}
    p = os.path.join(ROOT, "tests", "golden", "orc_streams.json")
    json.dump(res, open(p, "w"), indent=1)
    for f in res["files"]:
        print(f["file"], f["compression_kind"], f["number_of_rows"], f["column_types"], [(len(s["streams"]), s["rows"], [e["kind"] for e in s["encodings"]]) for s in f["stripes"]])
        print("  file stats", f["file_statistics"])
    print("wrote", p, os.path.getsize(p))
