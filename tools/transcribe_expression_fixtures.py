#!/usr/bin/env python3
"""Transcribes the known answers of the reference's scalar-operator and expression-compiler tests into
tests/golden/expression_vectors.json (data only: the SQL text of each assertion, its declared type and the literal result).

Run in the build container (reads /root/reference as text; the reference is Java and is not executed):

    python tools/transcribe_expression_fixtures.py

Sources (T/ = core/trino-main/src/test/java/io/trino/):
    T/type/TestBigintOperators.java, TestIntegerOperators.java, TestDoubleOperators.java, TestBooleanOperators.java
        assertFunction(sql, type, expected) / assertNumericOverflow(sql, message) / assertInvalidFunction(sql, code)
    T/sql/gen/TestExpressionCompiler.java
        the literal assertExecute(sql, type, expected) lines (AND / OR / NOT / IF / COALESCE truth tables ...) and the value tables
        its parametrised loops iterate over (booleanValues, smallInts, intRights, doubleLefts, ... :117-132)

The expected values in those files are Java constant expressions (`37 + 100000000037L`, `37.7 % 17.1`, `(double) 37`): they are
evaluated here with Java's arithmetic (32-/64-bit wrap-around, truncating division, remainder with the dividend's sign, IEEE
doubles) by a small constant-expression evaluator -- no part of the system under test is involved.  Assertions whose SQL or
expected value is not a plain constant (string concatenation with Math.nextUp(...), helper calls) are skipped and counted.
"""
import json
import math
import os
import re
import struct
import sys

REF = "/root/reference/core/trino-main/src/test/java/io/trino"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "expression_vectors.json")

I32 = (1 << 32)
I64 = (1 << 64)


def wrap(v, bits):
    m = 1 << bits
    v &= m - 1
    return v - m if v >> (bits - 1) else v


class JVal:
    def __init__(self, kind, v):
        self.kind, self.v = kind, v   # kind: int | long | double | boolean | string | null

    def __repr__(self):
        return f"{self.kind}:{self.v!r}"


class Unsupported(Exception):
    pass


CONSTS = {
    "Long.MAX_VALUE": JVal("long", 2**63 - 1), "Long.MIN_VALUE": JVal("long", -2**63),
    "Integer.MAX_VALUE": JVal("int", 2**31 - 1), "Integer.MIN_VALUE": JVal("int", -2**31),
    "Double.NaN": JVal("double", float("nan")), "Double.POSITIVE_INFINITY": JVal("double", float("inf")),
    "Double.NEGATIVE_INFINITY": JVal("double", float("-inf")), "Double.MAX_VALUE": JVal("double", sys.float_info.max),
    "Double.MIN_VALUE": JVal("double", 5e-324), "Double.MIN_NORMAL": JVal("double", 2.2250738585072014e-308),
    "true": JVal("boolean", True), "false": JVal("boolean", False), "null": JVal("null", None),
}

TOKEN = re.compile(r"""\s*(?:
    (?P<num>(?:0[xX][0-9a-fA-F_]+|\d[\d_]*\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?)[LlFfDd]?)
  | (?P<str>"(?:[^"\\]|\\.)*")
  | (?P<id>[A-Za-z_][A-Za-z_0-9]*(?:\.[A-Za-z_][A-Za-z_0-9]*)*)
  | (?P<op>[-+*/%()])
)""", re.X)


def tokenize(text):
    pos, out = 0, []
    text = text.strip()
    while pos < len(text):
        m = TOKEN.match(text, pos)
        if not m or m.end() == pos:
            raise Unsupported(f"cannot tokenize {text[pos:]!r}")
        pos = m.end()
        for k in ("num", "str", "id", "op"):
            if m.group(k) is not None:
                out.append((k, m.group(k)))
    return out


class JavaConst:
    """recursive-descent evaluator for the constant expressions the tests use as expected values"""

    def __init__(self, text):
        self.t = tokenize(text)
        self.i = 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else (None, None)

    def take(self):
        tok = self.peek()
        self.i += 1
        return tok

    def parse(self):
        v = self.additive()
        if self.i != len(self.t):
            raise Unsupported("trailing tokens")
        return v

    def additive(self):
        a = self.multiplicative()
        while self.peek() in (("op", "+"), ("op", "-")):
            op = self.take()[1]
            b = self.multiplicative()
            a = binop(op, a, b)
        return a

    def multiplicative(self):
        a = self.unary()
        while self.peek() in (("op", "*"), ("op", "/"), ("op", "%")):
            op = self.take()[1]
            b = self.unary()
            a = binop(op, a, b)
        return a

    def unary(self):
        k, v = self.peek()
        if (k, v) == ("op", "-"):
            self.take()
            a = self.unary()
            if a.kind == "double":
                return JVal("double", -a.v)
            if a.kind in ("int", "long"):
                return JVal(a.kind, wrap(-a.v, 32 if a.kind == "int" else 64))
            raise Unsupported("negation of " + a.kind)
        if (k, v) == ("op", "+"):
            self.take()
            return self.unary()
        if (k, v) == ("op", "("):
            # a cast: "(" type ")" unary
            if self.i + 2 < len(self.t) and self.t[self.i + 1][0] == "id" and self.t[self.i + 1][1] in ("double", "long", "int") and self.t[self.i + 2] == ("op", ")"):
                ty = self.t[self.i + 1][1]
                self.i += 3
                return cast(ty, self.unary())
            self.take()
            a = self.additive()
            if self.take() != ("op", ")"):
                raise Unsupported("unbalanced parenthesis")
            return a
        return self.primary()

    def primary(self):
        k, v = self.take()
        if k == "num":
            s = v.replace("_", "")
            if s[-1] in "fF" and not s.lower().startswith("0x"):
                raise Unsupported("float literal")
            if s[-1] in "lL":
                return JVal("long", wrap(int(s[:-1], 0), 64))
            if s[-1] in "dD" and not s.lower().startswith("0x"):
                return JVal("double", float(s[:-1]))
            if re.fullmatch(r"\d+", s) or s.lower().startswith("0x"):
                return JVal("int", wrap(int(s, 0), 32))
            return JVal("double", float(s))
        if k == "str":
            return JVal("string", bytes(v[1:-1], "utf-8").decode("unicode_escape"))
        if k == "id" and v in CONSTS:
            return CONSTS[v]
        raise Unsupported(f"token {v!r}")


def cast(ty, a):
    if a.kind not in ("int", "long", "double"):
        raise Unsupported("cast of " + a.kind)
    if ty == "double":
        return JVal("double", float(a.v))
    bits = 32 if ty == "int" else 64
    if a.kind == "double":
        if a.v != a.v:
            return JVal(ty, 0)
        lim = 2 ** (bits - 1)
        return JVal(ty, max(-lim, min(lim - 1, int(a.v))) if abs(a.v) != float("inf") else (lim - 1 if a.v > 0 else -lim))
    return JVal(ty, wrap(a.v, bits))


def binop(op, a, b):
    if a.kind == "string" or b.kind == "string":
        raise Unsupported("string arithmetic")
    if a.kind not in ("int", "long", "double") or b.kind not in ("int", "long", "double"):
        raise Unsupported("arithmetic on " + a.kind + ", " + b.kind)
    if a.kind == "double" or b.kind == "double":
        x, y = float(a.v), float(b.v)
        if op == "+":
            r = x + y
        elif op == "-":
            r = x - y
        elif op == "*":
            r = x * y
        elif op == "/":
            r = x / y if y != 0 else (float("nan") if x == 0 or x != x else math.copysign(float("inf"), x) * math.copysign(1.0, y))
        else:
            r = math.fmod(x, y) if y != 0 and abs(x) != float("inf") else float("nan")
        return JVal("double", r)
    kind = "long" if "long" in (a.kind, b.kind) else "int"
    bits = 64 if kind == "long" else 32
    x, y = a.v, b.v
    if op == "+":
        r = x + y
    elif op == "-":
        r = x - y
    elif op == "*":
        r = x * y
    else:
        if y == 0:
            raise Unsupported("integer division by zero")
        q = abs(x) // abs(y)
        q = q if (x >= 0) == (y >= 0) else -q
        r = q if op == "/" else x - q * y
    return JVal(kind, wrap(r, bits))


def jfmt(v):
    """JSON form of an expected value"""
    if v.kind == "null":
        return None
    if v.kind == "double":
        if v.v != v.v:
            return "NaN"
        if abs(v.v) == float("inf"):
            return "Infinity" if v.v > 0 else "-Infinity"
        return v.v
    return v.v


TYPE_NAMES = {"BIGINT": "bigint", "INTEGER": "integer", "DOUBLE": "double", "BOOLEAN": "boolean", "VARCHAR": "varchar", "REAL": "real",
              "SMALLINT": "smallint", "TINYINT": "tinyint", "UNKNOWN": "unknown", "VARBINARY": "varbinary", "DATE": "date"}


def split_args(text):
    """top-level comma split of a Java argument list"""
    out, depth, cur, in_str, esc = [], 0, "", False, False
    for ch in text:
        if in_str:
            cur += ch
            if esc:
                esc = False
            elif ch == "\\":
                esc = True
            elif ch == '"':
                in_str = False
            continue
        if ch == '"':
            in_str = True
            cur += ch
        elif ch in "([{":
            depth += 1
            cur += ch
        elif ch in ")]}":
            depth -= 1
            cur += ch
        elif ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def sql_of(arg):
    """the SQL text of the first argument: a plain string literal, or format("...%s...", CONST)"""
    m = re.fullmatch(r'"((?:[^"\\]|\\.)*)"', arg)
    if m:
        return bytes(m.group(1), "utf-8").decode("unicode_escape")
    m = re.fullmatch(r'format\("((?:[^"\\]|\\.)*)"\s*,\s*([A-Za-z_.]+)\)', arg)
    if m and m.group(2) in CONSTS:
        v = CONSTS[m.group(2)]
        return bytes(m.group(1), "utf-8").decode("unicode_escape").replace("%s", str(v.v), 1)
    raise Unsupported("sql is not a constant: " + arg[:60])


ERROR_BY_MESSAGE = [("overflow", "NUMERIC_VALUE_OUT_OF_RANGE"), ("Out of range", "NUMERIC_VALUE_OUT_OF_RANGE"), ("Division by zero", "DIVISION_BY_ZERO")]


def scan(rel, call_names):
    path = os.path.join(REF, rel)
    cases, skipped = [], 0
    lines = open(path).read().split("\n")
    for ln, line in enumerate(lines, 1):
        s = line.strip()
        m = re.match(r"(" + "|".join(call_names) + r")\((.*)\);\s*$", s)
        if not m:
            continue
        fn, args = m.group(1), split_args(m.group(2))
        src = f"T/{rel}:{ln}"
        try:
            sql = sql_of(args[0])
            if fn in ("assertFunction", "assertExecute"):
                if len(args) != 3:
                    raise Unsupported("argument count")
                ty = args[1]
                mt = re.fullmatch(r"createVarcharType\((\d+)\)", ty)
                tname = "varchar" if mt else TYPE_NAMES.get(ty)
                if tname is None:
                    raise Unsupported("type " + ty)
                val = JavaConst(args[2]).parse()
                cases.append({"src": src, "sql": sql, "type": tname, "expected": jfmt(val)})
            elif fn == "assertNumericOverflow":
                cases.append({"src": src, "sql": sql, "error": "NUMERIC_VALUE_OUT_OF_RANGE", "message": sql_of(args[1])})
            elif fn == "assertInvalidFunction":
                code = args[1]
                if code in ("INVALID_CAST_ARGUMENT", "DIVISION_BY_ZERO", "NUMERIC_VALUE_OUT_OF_RANGE"):
                    cases.append({"src": src, "sql": sql, "error": code})
                else:
                    msg = sql_of(code)
                    for frag, name in ERROR_BY_MESSAGE:
                        if frag in msg:
                            cases.append({"src": src, "sql": sql, "error": name, "message": msg})
                            break
                    else:
                        raise Unsupported("error " + msg)
        except Unsupported:
            skipped += 1
    return cases, skipped


def value_tables():
    """the arrays TestExpressionCompiler's parametrised loops iterate over (:117-132), parsed from their initialisers"""
    text = open(os.path.join(REF, "sql/gen/TestExpressionCompiler.java")).read()
    want = {"booleanValues": "Boolean", "smallInts": "Integer", "extremeInts": "Integer", "intRights": "Integer", "intMiddle": "Integer",
            "doubleLefts": "Double", "doubleRights": "Double", "doubleMiddle": "Double", "stringLefts": "String", "stringRights": "String",
            "longLefts": "Long", "longRights": "Long"}
    out = {}
    for name, jt in want.items():
        m = re.search(r"private static final " + jt + r"\[\] " + name + r" = \{(.*?)\};", text, re.S)
        assert m, name
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        vals = []
        for item in split_args(body):
            vals.append(jfmt(JavaConst(item).parse()))
        out[name] = vals
    return out


def main():
    doc = {"_comment": "Known answers transcribed from the reference's own tests by tools/transcribe_expression_fixtures.py (data: SQL text, declared type, "
                       "literal result; Java constant expressions evaluated with Java arithmetic by that script).  T/ = core/trino-main/src/test/java/io/trino/.",
           "operators": {}, "expression_compiler": {}}
    total = 0
    for rel in ("type/TestBigintOperators.java", "type/TestIntegerOperators.java", "type/TestDoubleOperators.java", "type/TestBooleanOperators.java"):
        cases, skipped = scan(rel, ["assertFunction", "assertNumericOverflow", "assertInvalidFunction"])
        doc["operators"][os.path.basename(rel)[:-5]] = {"cases": cases, "skipped_not_constant": skipped}
        total += len(cases)
    cases, skipped = scan("sql/gen/TestExpressionCompiler.java", ["assertExecute"])
    doc["expression_compiler"]["literal_cases"] = {"cases": cases, "skipped_not_constant": skipped}
    doc["expression_compiler"]["value_tables"] = {"source": "T/sql/gen/TestExpressionCompiler.java:117-132", **value_tables()}
    doc["expression_compiler"]["loops"] = {
        "_comment": "the parametrised loops, as (SQL template, result rule) pairs; the rule is the Java expression the test computes its expectation with",
        "testBinaryOperatorsIntegralIntegral": {"source": "T/sql/gen/TestExpressionCompiler.java:336-367", "lefts": "smallInts", "rights": "intRights",
                                                "long_left": "left * 1000000000L"},
        "testBinaryOperatorsIntegralDouble": {"source": "T/sql/gen/TestExpressionCompiler.java:370-398", "lefts": "intLefts = smallInts + extremeInts", "rights": "doubleRights"},
        "testBinaryOperatorsDoubleIntegral": {"source": "T/sql/gen/TestExpressionCompiler.java:401-425", "lefts": "doubleLefts", "rights": "intRights"},
        "testBinaryOperatorsDoubleDouble": {"source": "T/sql/gen/TestExpressionCompiler.java:428-452", "lefts": "doubleLefts", "rights": "doubleRights"},
        "testBinaryOperatorsString": {"source": "T/sql/gen/TestExpressionCompiler.java:610-631", "lefts": "stringLefts", "rights": "stringRights"},
        "testBinaryOperatorsBoolean": {"source": "T/sql/gen/TestExpressionCompiler.java:318-333", "values": "booleanValues"},
        "testTernaryOperatorsLongLong": {"source": "T/sql/gen/TestExpressionCompiler.java:696-710", "first": "intLefts", "second": "intLefts", "third": "intRights"},
        "testTernaryOperatorsLongDouble": {"source": "T/sql/gen/TestExpressionCompiler.java:713-727", "first": "intLefts", "second": "doubleLefts", "third": "intRights"},
        "testTernaryOperatorsDoubleDouble": {"source": "T/sql/gen/TestExpressionCompiler.java:730-744", "first": "doubleLefts", "second": "doubleLefts", "third": "intRights"},
        "testTernaryOperatorsString": {"source": "T/sql/gen/TestExpressionCompiler.java:747-761", "first": "stringLefts", "second": "stringLefts", "third": "stringRights"},
        "testCast": {"source": "T/sql/gen/TestExpressionCompiler.java:816-872", "booleans": "booleanValues", "ints": "intLefts", "doubles": "doubleLefts"},
        "testUnaryOperators": {"source": "T/sql/gen/TestExpressionCompiler.java:264-306", "booleans": "booleanValues", "ints": "intLefts", "doubles": "doubleLefts", "strings": "stringLefts"},
        "testIf": {"source": "T/sql/gen/TestExpressionCompiler.java:961-978", "conditions": "booleanValues", "true_values": "stringLefts", "false_values": "stringRights"},
    }
    total += len(cases)
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1, allow_nan=False)
        f.write("\n")
    for k, v in doc["operators"].items():
        print(k, len(v["cases"]), "cases,", v["skipped_not_constant"], "skipped")
    print("TestExpressionCompiler literal cases", len(cases), "skipped", skipped)
    print("total", total, "->", OUT)


if __name__ == "__main__":
    main()
