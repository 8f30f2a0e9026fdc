"""Shared by the CPU (oracle) and GPU (device) Parquet tests: Parquet files written by Apache Arrow (an independent implementation of the format the
reference reads through parquet-mr) in the page layouts the decoders cover, with the values Arrow's own reader returns as the expectation."""
import base64
import json
import os

import numpy as np

import parquet_pages as pp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_fixture():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "parquet_pages.json")))


def fixture_pages(fx):
    for c in fx["columns"]:
        chunk = {"optional": c["optional"]}
        for p in c["pages"]:
            page = dict(p, bytes=base64.b64decode(p["bytes"]))
            yield c, page, pp.split_data_page(chunk, page)


def arrow_table(n, seed):
    import pyarrow as pa
    rng = np.random.default_rng(seed)
    words = ["", "a", "BUILDING", "MACHINERY", "x" * 300, "héllo wörld", "AUTOMOBILE"]

    def nullable(vals, every):
        return [None if every and i % every == 0 else v for i, v in enumerate(vals)]
    return pa.table({
        "i64": pa.array(nullable(rng.integers(-10**15, 10**15, n).tolist(), 7), type=pa.int64()),
        "i64_runs": pa.array(nullable(np.repeat(rng.integers(0, 50, n // 50 + 1), 50)[:n].tolist(), 0), type=pa.int64()),
        "i32": pa.array(nullable(rng.integers(-2**31, 2**31, n).tolist(), 5), type=pa.int32()),
        "date": pa.array(nullable(rng.integers(8000, 11000, n).tolist(), 11), type=pa.date32()),
        "f64": pa.array(nullable(rng.standard_normal(n).tolist(), 3), type=pa.float64()),
        "flag": pa.array(nullable([bool(x) for x in rng.integers(0, 2, n)], 13), type=pa.bool_()),
        "s": pa.array(nullable([words[int(x)] for x in rng.integers(0, len(words), n)], 4), type=pa.string()),
        "s_required": pa.array([words[int(x)] for x in rng.integers(0, len(words), n)], type=pa.string()),
        "all_null": pa.array([None] * n, type=pa.int32()),
    }, schema=None)


def write_cases(tmp_dir, n=20_000):
    """(label, path, table) for every layout: dictionary on / off x data page V1 / V2, small pages so that a chunk has several"""
    import pyarrow as pa
    import pyarrow.parquet as pq
    out = []
    for seed, (dic, ver) in enumerate([(False, "1.0"), (True, "1.0"), (False, "2.0"), (True, "2.0")]):
        t = arrow_table(n, 100 + seed)
        # a required column: Arrow marks a field required only through the schema
        fields = [pa.field(f.name, f.type, nullable=f.name != "s_required") for f in t.schema]
        t = t.cast(pa.schema(fields))
        path = os.path.join(str(tmp_dir), f"case_{int(dic)}_{ver}.parquet")
        pq.write_table(t, path, compression="NONE", use_dictionary=dic, data_page_version=ver, write_statistics=False, data_page_size=16384)
        out.append((f"dictionary={dic} pages=V{ver[0]}", path, pq.read_table(path)))
    return out


DELTA_WORDS = ["", "a", "BUILDING", "MACHINERY", "x" * 300, "héllo wörld", "AUTOMOBILE"]
PREFIX_WORDS = ["", "a", "BUILD", "BUILDING", "MACHINE", "MACHINERY", "x" * 300, "x" * 299 + "y", "héllo wörld", "héllo", "AUTO", "AUTOMOBILE"]


def write_delta_cases(tmp_dir, n=20_000):
    """INT32 / INT64 columns written DELTA_BINARY_PACKED (ParquetEncoding.java:146-154), both data page versions: random values (wide miniblocks), a
    sorted key column (narrow ones), dates, nullable and required, an all-null column"""
    import pyarrow as pa
    import pyarrow.parquet as pq
    out = []
    for seed, ver in enumerate(("1.0", "2.0")):
        rng = np.random.default_rng(300 + seed)

        def nullable(vals, every):
            return [None if every and i % every == 0 else v for i, v in enumerate(vals)]
        t = pa.table({
            "i64": pa.array(nullable(rng.integers(-2**63, 2**63 - 1, n).tolist(), 7), type=pa.int64()),
            "i64_runs": pa.array((np.cumsum(rng.integers(0, 9, n)) + 10**12).tolist(), type=pa.int64()),
            "i32": pa.array(nullable(rng.integers(-2**31, 2**31, n).tolist(), 5), type=pa.int32()),
            "date": pa.array(nullable(rng.integers(8000, 11000, n).tolist(), 11), type=pa.date32()),
            "all_null": pa.array([None] * n, type=pa.int32()),
            # strings as DELTA_LENGTH_BYTE_ARRAY (ParquetEncoding.java:156-163)
            "s": pa.array(nullable([DELTA_WORDS[int(x)] for x in rng.integers(0, len(DELTA_WORDS), n)], 4), type=pa.string()),
            "s_required": pa.array([DELTA_WORDS[int(x)] for x in rng.integers(0, len(DELTA_WORDS), n)], type=pa.string()),
            # strings as DELTA_BYTE_ARRAY (ParquetEncoding.java:165-173): a sorted key column (long shared prefixes) and words with nulls
            "p_sorted": pa.array(sorted("key%07d" % int(x) for x in rng.integers(0, 10**6, n)), type=pa.string()),
            "p": pa.array(nullable([PREFIX_WORDS[int(x)] for x in rng.integers(0, len(PREFIX_WORDS), n)], 6), type=pa.string()),
        })
        t = t.cast(pa.schema([pa.field(f.name, f.type, nullable=f.name != "s_required") for f in t.schema]))
        path = os.path.join(str(tmp_dir), f"delta_{ver}.parquet")
        pq.write_table(t, path, compression="NONE", use_dictionary=False, data_page_version=ver, write_statistics=False, data_page_size=16384,
                       column_encoding={c: "DELTA_LENGTH_BYTE_ARRAY" if c.startswith("s") else "DELTA_BYTE_ARRAY" if c.startswith("p") else "DELTA_BINARY_PACKED" for c in t.column_names})
        out.append((f"delta pages=V{ver[0]}", path, pq.read_table(path)))
    return out


def expected_column(table, name, physical):
    col = table.column(name).to_pylist()
    if physical == pp.BYTE_ARRAY:
        return [None if v is None else v.encode("utf-8") for v in col]
    if table.schema.field(name).type == __import__("pyarrow").date32():
        import datetime
        return [None if v is None else (v - datetime.date(1970, 1, 1)).days for v in col]
    return col


TYPE_OF = {"p": "VARCHAR", "p_sorted": "VARCHAR", "i64": "BIGINT", "i64_runs": "BIGINT", "i32": "INTEGER", "date": "DATE", "f64": "DOUBLE", "flag": "BOOLEAN", "s": "VARCHAR", "s_required": "VARCHAR", "all_null": "INTEGER"}
