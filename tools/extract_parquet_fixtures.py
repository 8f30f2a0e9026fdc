"""Build-container script: turns the one Parquet data file of the reference whose column type the device decoders cover
(testing/trino-product-tests/src/main/resources/io/trino/tests/hive/data/single_int_column/data.parquet) into tests/golden/parquet_pages.json -- DATA,
not source: the column chunk's page payloads (base64) with what the page headers say about them, and the known answer the reference's own test
asserts for the file (TestParquetSymlinkInputFormat.java:63: SELECT * returns exactly row(42)).  The file is walked with tests/parquet_pages.py
(a thrift-compact reader following the public Parquet specification); nothing of the reference's code is used.
    python tools/extract_parquet_fixtures.py      (needs /root/reference)"""
import base64
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parquet_pages as pp   # noqa: E402

SRC = "/root/reference/testing/trino-product-tests/src/main/resources/io/trino/tests/hive/data/single_int_column/data.parquet"
out = {"_comment": "made by tools/extract_parquet_fixtures.py from the reference's single_int_column/data.parquet: page payloads + header fields; `asserted_rows` is what "
                   "testing/trino-product-tests/src/main/java/io/trino/tests/hive/TestParquetSymlinkInputFormat.java:63 asserts SELECT * returns",
       "file": "single_int_column/data.parquet", "asserted_rows": [[42]], "columns": []}
for c in pp.column_chunks(SRC):
    out["columns"].append({"name": c["name"], "physical": c["physical"], "optional": c["optional"], "num_values": c["num_values"],
                           "pages": [{**{k: v for k, v in p.items() if k != "bytes"}, "bytes": base64.b64encode(p["bytes"]).decode()} for p in c["pages"]]})
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "parquet_pages.json"), "w"), indent=1)
print("wrote tests/golden/parquet_pages.json", [(c["name"], len(c["pages"])) for c in out["columns"]])
