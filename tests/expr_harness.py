"""Shared driver of the expression known-answer tests: loads tests/golden/expression_vectors.json, turns each SQL text into the IR
(sqlmini), packs the cases into page processors (<= 12 computed projections, <= 20 input channels each) and evaluates them on an
"engine":

    engine(input_types, columns, filter_expr, projections) -> list of per-projection value lists (one entry per selected row)

The CPU test binds the engine to the C oracle (o_filter / o_project), the GPU test to FilterAndProjectOperator through the C ABI,
so both are pinned on the same literals of the reference's tests.  Also the Java-semantics helpers the parametrised
TestExpressionCompiler loops need (the expectation there is a Java expression over the value tables, not a literal)."""
import json
import math
import os

import sqlmini
from sqlmini import BIGINT, BOOLEAN, DOUBLE, INTEGER, VARCHAR, Unsupported

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "expression_vectors.json")))
TYPE_IDS = {"bigint": BIGINT, "integer": INTEGER, "double": DOUBLE, "boolean": BOOLEAN, "varchar": VARCHAR}
ERROR_CODES = {"NUMERIC_VALUE_OUT_OF_RANGE": -2, "DIVISION_BY_ZERO": -7, "INVALID_CAST_ARGUMENT": -9}
MAX_PROJ, MAX_COLS = 12, 20


def all_cases():
    out = []
    for name, grp in GOLD["operators"].items():
        out += grp["cases"]
    out += GOLD["expression_compiler"]["literal_cases"]["cases"]
    return out


def parsed_cases():
    """-> (value cases [(case, ast)], error cases [(case, ast)], number skipped as outside the IR)"""
    values, errors, skipped = [], [], 0
    for c in all_cases():
        try:
            ast = sqlmini.parse(c["sql"])
            if "error" in c:
                errors.append((c, ast))
                continue
            t = TYPE_IDS.get(c["type"])
            if t is None:
                raise Unsupported("result type " + c["type"])
            if ast.type == sqlmini.UNKNOWN:
                raise Unsupported("untyped result")
            if ast.type != t:
                raise Unsupported(f"analysis gives type {ast.type}, the test declares {t}")
            values.append((c, ast))
        except Unsupported:
            skipped += 1
    return values, errors, skipped


def one_row_page(columns):
    """[(type, value)] -> per-column python lists (one row)"""
    return [(t, [v]) for t, v in columns]


def run_value_cases(E, engine, hoist):
    """evaluates every supported value case; returns (checked, mismatches [(case, got)])"""
    values, _, _ = parsed_cases()
    checked, bad = 0, []
    batch, low = [], sqlmini.Lowering(E, hoist)

    def flush():
        nonlocal batch, low, checked
        if not batch:
            return
        cols = low.columns if (hoist and low.columns) else [(BIGINT, 0)]
        types = [t for t, _ in cols]
        got = engine(types, one_row_page(cols), None, [e for _, e in batch])
        for (c, _), g in zip(batch, got):
            want = sqlmini.expected_value(c["type"], c["expected"])
            checked += 1
            if len(g) != 1 or not sqlmini.same_value(TYPE_IDS[c["type"]], g[0], want):
                bad.append((c, g))
        batch, low = [], sqlmini.Lowering(E, hoist)

    for c, ast in values:
        if ast.type == VARCHAR and not (hoist and ast.kind == "lit"):
            continue   # computed VARCHAR projections are outside the page processor (jit.cpp); a hoisted literal is an identity projection
        if hoist and len(low.columns) + sqlmini.literal_count(ast) > MAX_COLS:
            flush()
        batch.append((c, low.lower(ast)))
        if len(batch) >= MAX_PROJ:
            flush()
    flush()
    return checked, bad


def run_error_cases(E, engine, hoist, error_type):
    """every error case must raise the reference's error code; returns (checked, wrong [(case, what happened)])"""
    _, errors, _ = parsed_cases()
    checked, bad = 0, []
    for c, ast in errors:
        low = sqlmini.Lowering(E, hoist)
        try:
            e = low.lower(ast)
        except Unsupported:
            continue
        cols = low.columns if (hoist and low.columns) else [(BIGINT, 0)]
        checked += 1
        try:
            got = engine([t for t, _ in cols], one_row_page(cols), None, [e])
            bad.append((c, f"no error, result {got}"))
        except error_type as ex:
            if ex.code != ERROR_CODES[c["error"]]:
                bad.append((c, f"code {ex.code}"))
    return checked, bad


# ---- Java semantics for the parametrised loops (the reference computes its expectation with these Java operators) ----------
def jwrap(v, bits):
    m = 1 << bits
    v &= m - 1
    return v - m if v >> (bits - 1) else v


def jdiv(a, b):
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def jrem(a, b):
    return a - jdiv(a, b) * b


def jarith(op, a, b, bits):
    """Java int / long arithmetic (bits = 32 / 64); None when an operand is null"""
    if a is None or b is None:
        return None
    if op == "+":
        return jwrap(a + b, bits)
    if op == "-":
        return jwrap(a - b, bits)
    if op == "*":
        return jwrap(a * b, bits)
    if op == "/":
        return jwrap(jdiv(a, b), bits)
    return jwrap(jrem(a, b), bits)


def jdarith(op, a, b):
    """Java double arithmetic"""
    if a is None or b is None:
        return None
    a, b = float(a), float(b)
    if op == "+":
        return a + b
    if op == "-":
        return a - b
    if op == "*":
        try:
            return a * b
        except OverflowError:
            return math.copysign(float("inf"), a) * math.copysign(1.0, b)
    if op == "/":
        if b == 0:
            return float("nan") if a == 0 or a != a else math.copysign(float("inf"), a) * math.copysign(1.0, b)
        try:
            return a / b
        except OverflowError:
            return math.copysign(float("inf"), a) * math.copysign(1.0, b)
    if b == 0 or abs(a) == float("inf"):
        return float("nan")
    return math.fmod(a, b)


def jcmp(op, a, b):
    if a is None or b is None:
        return None
    return {"=": a == b, "<>": a != b, ">": a > b, "<": a < b, ">=": a >= b, "<=": a <= b}[op]


def jbetween(v, lo, hi):
    """TestExpressionCompiler.between (T/sql/gen/TestExpressionCompiler.java:797-813): three-valued `lo <= v AND v <= hi`"""
    if v is None:
        return None
    a = None if lo is None else lo <= v
    b = None if hi is None else v <= hi
    if a is False or b is False:
        return False
    if a is None or b is None:
        return None
    return True


def cross(*lists):
    """row-major cross product as per-column lists"""
    rows = [[]]
    for lst in lists:
        rows = [r + [x] for r in rows for x in lst]
    return [[r[k] for r in rows] for k in range(len(lists))]


# ---- engines -----------------------------------------------------------------------------------------------------------------
class Engine:
    """pages(types, pages, filter, projections) -> one (position_count, [per-projection value lists]) per OUTPUT page;
    calling the engine itself evaluates a single page and returns the per-projection lists"""

    def __call__(self, types, columns, filt, projs):
        out = self.pages(types, [columns], filt, projs)
        res = [[] for _ in projs]
        for _, cols in out:
            for k in range(len(projs)):
                res[k] += cols[k]
        return res


def oracle_engine(oracle, E):
    """page processor semantics on the C oracle: o_filter, then o_project per projection on the selected positions; an input page
    that selects no row produces no output page (PageProcessor.java:111-137)"""
    import numpy as np

    def col(t, vals):
        if t == VARCHAR:
            return oracle.Col(VARCHAR, vals)
        nulls = np.array([v is None for v in vals], dtype=np.uint8)
        data = [0 if v is None else v for v in vals]
        return oracle.Col(t, np.array(data, dtype=oracle._NP[t]), nulls if nulls.any() else None)

    class OracleEngine(Engine):
        def pages(self, types, pages, filt, projs):
            prog = E.FlatProgram(filt, projs)
            pool = bytes(prog.pool)
            result = []
            for columns in pages:
                n = len(columns[0][1]) if columns else 0
                if n == 0:
                    continue
                cols = [col(t, vals) for t, vals in columns]
                pos = oracle.filter_positions(prog.nodes, prog.filter_root, pool, cols) if prog.filter_root >= 0 else np.arange(n, dtype=np.int32)
                if len(pos) == 0:
                    continue
                out = []
                for r in prog.projection_roots:
                    nd = prog.nodes[r]
                    if nd["kind"] == 0:   # InputPageProjection
                        vals = columns[nd["op"]][1]
                        out.append([vals[i] for i in pos])
                        continue
                    v, nl = oracle.project(prog.nodes, r, pool, cols, pos)
                    conv = float if nd["type"] == DOUBLE else (bool if nd["type"] == BOOLEAN else int)
                    out.append([None if nl[i] else conv(v[i]) for i in range(len(pos))])
                result.append((len(pos), out))
            return result

    return OracleEngine()


def product_engine(pkg, ctx):
    """FilterAndProjectOperator through the C ABI (tgpu_filter_project_factory_create + the Operator protocol)"""
    class ProductEngine(Engine):
        def pages(self, types, pages, filt, projs):
            fac = pkg.FilterAndProjectOperatorFactory(ctx, 0, types, filt, projs)
            op = fac.createOperator()
            outs = pkg.to_pages(op, [pkg.Page(*[pkg.Block(t, vals) for t, vals in columns]) for columns in pages])
            op.close()
            fac.close()
            return [(o.position_count, [o.getBlock(k).to_list() for k in range(len(projs))]) for o in outs]

    return ProductEngine()


# ---- page-level fixtures: T/operator/TestFilterAndProjectOperator.java, T/operator/project/TestPageProcessor.java -------------
def run_page_fixtures(E, engine, ref_gold):
    """ref_gold = tests/golden/reference_vectors.json; asserts inside"""
    f, c = E.field, E.constant
    fp = ref_gold["filter_and_project"]
    seq = lambda n: ([str(i) for i in range(n)], list(range(n)))   # rowPagesBuilder(VARCHAR, BIGINT).addSequencePage(n, 0, 0)
    page = [(VARCHAR, seq(100)[0]), (BIGINT, seq(100)[1])]
    # test(): filter field(1) <= 9, projections field(0), field(1) + 5
    out = engine.pages([VARCHAR, BIGINT], [page], f(1, BIGINT) <= 9, [f(0, VARCHAR), f(1, BIGINT) + 5])
    rows = [list(r) for _, cols in out for r in zip(*cols)]
    assert rows == fp["test"]["expect_rows"]
    # testMergeOutput(): four pages, filter field(1) = 10, projection field(1); rows compared (assertOperatorEquals)
    out = engine.pages([VARCHAR, BIGINT], [page] * 4, f(1, BIGINT).eq(10), [f(1, BIGINT)])
    rows = [list(r) for _, cols in out for r in zip(*cols)]
    assert rows == fp["testMergeOutput"]["expect_rows"]
    # TestPageProcessor: position-range filters over createLongSequenceBlock(0, n)
    for name, case in ref_gold["page_processor"].items():
        if name.startswith("_"):
            continue
        n = case["input_rows"]
        filt = None
        if case.get("filter_range") is not None:
            a, cnt = case["filter_range"]
            filt = E.between(f(0, BIGINT), a, a + cnt - 1)
        projs = [f(ch, BIGINT) for ch in case["projection_channels"]]
        out = engine.pages([BIGINT], [[(BIGINT, list(range(n)))]], filt, projs)
        assert len(out) == case["expect_pages"], name
        if "expect_position_count" in case:
            assert out[0][0] == case["expect_position_count"] and len(out[0][1]) == case["expect_channel_count"], name
        if "expect_sequence" in case:
            lo, hi = case["expect_sequence"]
            assert out[0][1][0] == list(range(lo, hi)), name


# ---- the parametrised loops of T/sql/gen/TestExpressionCompiler.java over its value tables ------------------------------------
def run_loops(E, engine):
    """One page per loop: a row per value combination, a projection per operator; expectations from the Java operators the test
    itself applies.  Returns (checked cells, mismatches)."""
    T = GOLD["expression_compiler"]["value_tables"]
    small, extreme, int_rights = T["smallInts"], T["extremeInts"], T["intRights"]
    int_lefts = small + extreme
    dlefts, drights = T["doubleLefts"], T["doubleRights"]
    slefts, srights = T["stringLefts"], T["stringRights"]
    bools = T["booleanValues"]
    f, c = E.field, E.constant
    checked, bad = 0, []
    CMPS = [("=", "EQUAL"), ("<>", "NOT_EQUAL"), (">", "GREATER_THAN"), ("<", "LESS_THAN"), (">=", "GREATER_THAN_OR_EQUAL"), ("<=", "LESS_THAN_OR_EQUAL")]
    ARI = [("+", "ADD"), ("-", "SUBTRACT"), ("*", "MULTIPLY"), ("/", "DIVIDE"), ("%", "MODULUS")]

    def check(name, types, columns, items):
        nonlocal checked
        for at in range(0, len(items), MAX_PROJ):
            part = items[at:at + MAX_PROJ]
            got = engine(types, list(zip(types, columns)), None, [e for e, _, _ in part])
            for (e, t, want), g in zip(part, got):
                assert len(g) == len(want), (name, len(g), len(want))
                for row, (x, y) in enumerate(zip(g, want)):
                    checked += 1
                    if not sqlmini.same_value(t, x, y):
                        bad.append((name, row, [col[row] for col in columns], x, y))

    def dbl(e):
        return E.cast(e, DOUBLE)

    def big(e):
        return E.cast(e, BIGINT)

    # testBinaryOperatorsIntegralIntegral :336-367
    L, R = cross(small, int_rights)
    l, r = f(0, INTEGER), f(1, INTEGER)
    items = [(E.call(n, BOOLEAN, l, r), BOOLEAN, [jcmp(s, a, b) for a, b in zip(L, R)]) for s, n in CMPS]
    items += [(E.call(n, INTEGER, l, r), INTEGER, [jarith(s, a, b, 32) for a, b in zip(L, R)]) for s, n in ARI]
    items.append((E.if_(l.eq(r), c(None, INTEGER), l), INTEGER, [None if (a is None or (b is not None and a == b)) else a for a, b in zip(L, R)]))   # nullif :348
    LL = [None if a is None else a * 1000000000 for a in L]
    ll = f(2, BIGINT)
    items += [(E.call(n, BIGINT, ll, big(r)), BIGINT, [jarith(s, a, b, 64) for a, b in zip(LL, R)]) for s, n in ARI]
    check("IntegralIntegral", [INTEGER, INTEGER, BIGINT], [L, R, LL], items)

    # testBinaryOperatorsIntegralDouble :370-398
    L, R = cross(int_lefts, drights)
    l, r = f(0, INTEGER), f(1, DOUBLE)
    items = [(E.call(n, BOOLEAN, dbl(l), r), BOOLEAN, [jcmp(s, None if a is None else float(a), b) for a, b in zip(L, R)]) for s, n in CMPS]
    items += [(E.call(n, DOUBLE, dbl(l), r), DOUBLE, [jdarith(s, a, b) for a, b in zip(L, R)]) for s, n in ARI]
    check("IntegralDouble", [INTEGER, DOUBLE], [L, R], items)

    # testBinaryOperatorsDoubleIntegral :401-425
    L, R = cross(dlefts, int_rights)
    l, r = f(0, DOUBLE), f(1, INTEGER)
    items = [(E.call(n, BOOLEAN, l, dbl(r)), BOOLEAN, [jcmp(s, a, None if b is None else float(b)) for a, b in zip(L, R)]) for s, n in CMPS]
    items += [(E.call(n, DOUBLE, l, dbl(r)), DOUBLE, [jdarith(s, a, b) for a, b in zip(L, R)]) for s, n in ARI]
    check("DoubleIntegral", [DOUBLE, INTEGER], [L, R], items)

    # testBinaryOperatorsDoubleDouble :428-452
    L, R = cross(dlefts, drights)
    l, r = f(0, DOUBLE), f(1, DOUBLE)
    items = [(E.call(n, BOOLEAN, l, r), BOOLEAN, [jcmp(s, a, b) for a, b in zip(L, R)]) for s, n in CMPS]
    items += [(E.call(n, DOUBLE, l, r), DOUBLE, [jdarith(s, a, b) for a, b in zip(L, R)]) for s, n in ARI]
    items.append((E.if_(l.eq(r), c(None, DOUBLE), l), DOUBLE, [None if (a is None or (b is not None and a == b)) else a for a, b in zip(L, R)]))
    check("DoubleDouble", [DOUBLE, DOUBLE], [L, R], items)

    # testBinaryOperatorsString :610-631 (String.compareTo == unsigned byte order for these ASCII values)
    L, R = cross(slefts, srights)
    l, r = f(0, VARCHAR), f(1, VARCHAR)
    items = [(E.call(n, BOOLEAN, l, r), BOOLEAN, [jcmp(s, a, b) for a, b in zip(L, R)]) for s, n in CMPS]
    check("String", [VARCHAR, VARCHAR], [L, R], items)

    # testBinaryOperatorsBoolean :318-333
    L, R = cross(bools, bools)
    l, r = f(0, BOOLEAN), f(1, BOOLEAN)
    items = [(E.call(n, BOOLEAN, l, r), BOOLEAN, [jcmp(s, a, b) for a, b in zip(L, R)]) for s, n in CMPS[:2]]
    items.append((E.if_(l.eq(r), c(None, BOOLEAN), l), BOOLEAN, [None if (a is None or (b is not None and a == b)) else a for a, b in zip(L, R)]))
    check("Boolean", [BOOLEAN, BOOLEAN], [L, R], items)

    # testTernaryOperators* :696-761: first BETWEEN second AND third
    A, B, C = cross(int_lefts, int_lefts, int_rights)
    check("TernaryLongLong", [INTEGER] * 3, [A, B, C], [(E.between(f(0, INTEGER), f(1, INTEGER), f(2, INTEGER)), BOOLEAN, [jbetween(a, b, x) for a, b, x in zip(A, B, C)])])
    A, B, C = cross(int_lefts, dlefts, int_rights)
    check("TernaryLongDouble", [INTEGER, DOUBLE, INTEGER], [A, B, C],
          [(E.between(dbl(f(0, INTEGER)), f(1, DOUBLE), dbl(f(2, INTEGER))), BOOLEAN, [jbetween(a, b, x) for a, b, x in zip(A, B, C)])])
    A, B, C = cross(dlefts, dlefts, int_rights)
    check("TernaryDoubleDouble", [DOUBLE, DOUBLE, INTEGER], [A, B, C],
          [(E.between(f(0, DOUBLE), f(1, DOUBLE), dbl(f(2, INTEGER))), BOOLEAN, [jbetween(a, b, x) for a, b, x in zip(A, B, C)])])
    A, B, C = cross(slefts, slefts, srights)
    check("TernaryString", [VARCHAR] * 3, [A, B, C], [(E.between(f(0, VARCHAR), f(1, VARCHAR), f(2, VARCHAR)), BOOLEAN, [jbetween(a, b, x) for a, b, x in zip(A, B, C)])])

    # testCast :816-872 (casts to varchar are outside the IR)
    v = f(0, BOOLEAN)
    check("CastBoolean", [BOOLEAN], [bools], [
        (E.cast(v, BOOLEAN), BOOLEAN, bools), (E.cast(v, INTEGER), INTEGER, [None if x is None else int(x) for x in bools]),
        (E.cast(v, BIGINT), BIGINT, [None if x is None else int(x) for x in bools]), (E.cast(v, DOUBLE), DOUBLE, [None if x is None else float(x) for x in bools])])
    v = f(0, INTEGER)
    check("CastInteger", [INTEGER], [int_lefts], [
        (E.cast(v, BOOLEAN), BOOLEAN, [None if x is None else x != 0 for x in int_lefts]), (E.cast(v, INTEGER), INTEGER, int_lefts),
        (E.cast(v, BIGINT), BIGINT, int_lefts), (E.cast(v, DOUBLE), DOUBLE, [None if x is None else float(x) for x in int_lefts])])
    v = f(0, DOUBLE)
    in_range = [x for x in dlefts if x is None or -2.0**63 <= x < 2.0**63]   # :841 the bigint cast is only asserted for values inside the long range
    check("CastDouble", [DOUBLE], [dlefts], [(E.cast(v, BOOLEAN), BOOLEAN, [None if x is None else x != 0.0 for x in dlefts]), (E.cast(v, DOUBLE), DOUBLE, dlefts)])
    check("CastDoubleBigint", [DOUBLE], [in_range], [(E.cast(v, BIGINT), BIGINT, [None if x is None else int(x) for x in in_range])])   # Double.longValue() truncates (:842)

    # testUnaryOperators :264-306
    v = f(0, BOOLEAN)
    check("UnaryBoolean", [BOOLEAN], [bools], [(E.is_null(v), BOOLEAN, [x is None for x in bools]), (E.not_(E.is_null(v)), BOOLEAN, [x is not None for x in bools])])
    v = f(0, INTEGER)
    longs = [None if x is None else jwrap(x * 10000000000, 64) for x in int_lefts]   # Java long arithmetic (:276)
    check("UnaryInteger", [INTEGER, BIGINT], [int_lefts, longs], [
        (-v, INTEGER, [None if x is None else -x for x in int_lefts]), (-f(1, BIGINT), BIGINT, [None if x is None else -x for x in longs]),
        (E.is_null(v), BOOLEAN, [x is None for x in int_lefts]), (E.not_(E.is_null(v)), BOOLEAN, [x is not None for x in int_lefts])])
    v = f(0, DOUBLE)
    check("UnaryDouble", [DOUBLE], [dlefts], [(-v, DOUBLE, [None if x is None else -x for x in dlefts]), (E.is_null(v), BOOLEAN, [x is None for x in dlefts])])
    v = f(0, VARCHAR)
    check("UnaryString", [VARCHAR], [slefts], [(v, VARCHAR, slefts), (E.is_null(v), BOOLEAN, [x is None for x in slefts])])
    return checked, bad
