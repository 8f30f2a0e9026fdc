"""Shared by the CPU (oracle) and GPU tests of the DynamicFilterSourceOperator fixtures (tests/golden/reference_vectors.json
"dynamic_filter_source", transcribed from T/operator/TestDynamicFilterSourceOperator.java)."""
import numpy as np

BIGINT, INTEGER, DATE, DOUBLE, BOOLEAN, VARCHAR = 1, 2, 3, 4, 5, 6
_NP = {BIGINT: np.int64, INTEGER: np.int32, DATE: np.int32, DOUBLE: np.float64, BOOLEAN: np.uint8}


def column_values(type_id, desc):
    """python values (None = null) of one column descriptor"""
    if "values" in desc:
        vals = desc["values"]
    elif "seq" in desc:
        vals = list(range(desc["seq"][0], desc["seq"][1]))
    elif "repeat" in desc:
        vals = [desc["repeat"][0]] * desc["repeat"][1]
    elif "nulls" in desc:
        vals = [None] * desc["nulls"]
    elif "text_repeat" in desc:
        vals = [desc["text_repeat"][0] * desc["text_repeat"][1]]
    else:
        raise ValueError(desc)
    if type_id == DOUBLE:
        vals = [None if v is None else (float("nan") if v == "nan" else float(v)) for v in vals]
    return vals


def cases(gold):
    """(name, types, channels, params, [(pages, expect) per operator])"""
    out = []
    for name, c in gold["dynamic_filter_source"].items():
        if name.startswith("_"):
            continue
        ops = c.get("operators") or [{"pages": c["pages"], "expect": c["expect"]}]
        out.append((name, c["types"], c["channels"], (c["max_distinct_values"], c["max_filter_size_in_bytes"], c["min_max_collection_limit"]), [(o["pages"], o["expect"]) for o in ops]))
    return out


def normalise(domain, type_id):
    """what a domain() answer is compared as: value sets sorted (Domain.multipleValues is a sorted set)"""
    if domain[0] == "values":
        vals = list(domain[1])
        if type_id == DOUBLE:
            vals = [float(v) for v in vals]
        if type_id == BOOLEAN:
            vals = [bool(v) for v in vals]
        return ("values", sorted(vals))
    return tuple(domain)


def expected(expect):
    if expect[0] == "values":
        return ("values", sorted(expect[1]))
    if expect[0] == "range_text":
        return ("range", expect[1][0] * expect[1][1], expect[2][0] * expect[2][1])
    if expect[0] == "values_seq":
        return ("values", list(range(expect[1], expect[2])))
    return tuple(expect)
