"""A small SQL scalar-expression front end for the parity tests: the SQL text of the reference's own operator tests
(tests/golden/expression_vectors.json) -> the RowExpression IR the page processor takes (presto-1_amd/expressions.py).

It does what the reference's parser + analyzer + SqlToRowExpressionTranslator do for this subset
(M/sql/relational/SqlToRowExpressionTranslator.java): literal typing (an integer literal that fits 32 bits is INTEGER, else BIGINT;
`1E0` is DOUBLE; `1.5` would be DECIMAL -> unsupported), implicit coercion to the common super type (INTEGER < BIGINT < DOUBLE), and
lowering of NULLIF / IS NOT NULL / NOT BETWEEN to the special forms the IR has.  Anything outside the IR (decimals, varchar casts,
functions, IS DISTINCT FROM, ...) raises Unsupported and the caller skips the case.

Two lowering modes: literals as ConstantExpressions, or every literal hoisted into an input channel of a one-row page (so the same
known answers also exercise the column-load paths of the generated kernels)."""
import math
import re
import struct

BIGINT, INTEGER, DATE, DOUBLE, BOOLEAN, VARCHAR = 1, 2, 3, 4, 5, 6
UNKNOWN = 0
TYPE_BY_NAME = {"bigint": BIGINT, "integer": INTEGER, "int": INTEGER, "double": DOUBLE, "boolean": BOOLEAN, "varchar": VARCHAR, "date": DATE}


class Unsupported(Exception):
    pass


TOKEN = re.compile(r"""\s*(?:
    (?P<num>\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?)
  | (?P<str>'(?:[^']|'')*')
  | (?P<id>[A-Za-z_][A-Za-z_0-9]*)
  | (?P<op><>|!=|<=|>=|\|\||[-+*/%(),=<>])
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


class Node:
    """typed AST node: kind in {lit, call, special}"""

    def __init__(self, kind, type_id, name=None, args=(), value=None):
        self.kind, self.type, self.name, self.args, self.value = kind, type_id, name, list(args), value


def lit(value, type_id):
    return Node("lit", type_id, value=value)


def common_type(a, b):
    if a == b:
        return a
    if a == UNKNOWN:
        return b
    if b == UNKNOWN:
        return a
    num = {INTEGER: 0, BIGINT: 1, DOUBLE: 2}
    if a in num and b in num:
        return a if num[a] > num[b] else b
    raise Unsupported("no common type")


def coerce(node, type_id):
    if node.type == type_id:
        return node
    if node.type == UNKNOWN:
        if node.kind != "lit":
            raise Unsupported("untyped non-literal")
        return lit(None, type_id)
    num = {INTEGER: 0, BIGINT: 1, DOUBLE: 2}
    if node.type in num and type_id in num and num[type_id] > num[node.type]:
        return Node("call", type_id, "CAST", [node])
    raise Unsupported(f"cannot coerce {node.type} to {type_id}")


CMP = {"=": "EQUAL", "<>": "NOT_EQUAL", "!=": "NOT_EQUAL", "<": "LESS_THAN", "<=": "LESS_THAN_OR_EQUAL", ">": "GREATER_THAN", ">=": "GREATER_THAN_OR_EQUAL"}
ARITH = {"+": "ADD", "-": "SUBTRACT", "*": "MULTIPLY", "/": "DIVIDE", "%": "MODULUS"}


class Parser:
    def __init__(self, text):
        self.t = tokenize(text)
        self.i = 0

    def peek(self, k=0):
        j = self.i + k
        return self.t[j] if j < len(self.t) else (None, None)

    def take(self):
        tok = self.peek()
        self.i += 1
        return tok

    def kw(self, word, k=0):
        t = self.peek(k)
        return t[0] == "id" and t[1].upper() == word

    def expect_kw(self, word):
        if not self.kw(word):
            raise Unsupported(f"expected {word}")
        self.take()

    def expect_op(self, op):
        if self.peek() != ("op", op):
            raise Unsupported(f"expected {op}")
        self.take()

    def parse(self):
        e = self.or_expr()
        if self.i != len(self.t):
            raise Unsupported("trailing tokens: " + repr(self.t[self.i:]))
        return e

    def boolean(self, n):
        if n.type == UNKNOWN:
            return coerce(n, BOOLEAN)
        if n.type != BOOLEAN:
            raise Unsupported("boolean expected")
        return n

    def or_expr(self):
        a = self.and_expr()
        while self.kw("OR"):
            self.take()
            b = self.and_expr()
            a = Node("special", BOOLEAN, "OR", [self.boolean(a), self.boolean(b)])
        return a

    def and_expr(self):
        a = self.not_expr()
        while self.kw("AND"):
            self.take()
            b = self.not_expr()
            a = Node("special", BOOLEAN, "AND", [self.boolean(a), self.boolean(b)])
        return a

    def not_expr(self):
        if self.kw("NOT"):
            self.take()
            a = self.not_expr()
            return Node("call", BOOLEAN, "NOT", [self.boolean(a)])
        return self.predicate()

    def predicate(self):
        a = self.additive()
        k, v = self.peek()
        if k == "op" and v in CMP:
            self.take()
            b = self.additive()
            t = common_type(a.type, b.type)
            if t == UNKNOWN:
                raise Unsupported("comparison of two untyped nulls")
            return Node("call", BOOLEAN, CMP[v], [coerce(a, t), coerce(b, t)])
        negate = False
        if self.kw("NOT") and self.kw("BETWEEN", 1):
            self.take()
            negate = True
        if self.kw("BETWEEN"):
            self.take()
            lo = self.additive()
            self.expect_kw("AND")
            hi = self.additive()
            t = common_type(common_type(a.type, lo.type), hi.type)
            if t == UNKNOWN:
                raise Unsupported("BETWEEN of untyped nulls")
            n = Node("special", BOOLEAN, "BETWEEN", [coerce(a, t), coerce(lo, t), coerce(hi, t)])
            return Node("call", BOOLEAN, "NOT", [n]) if negate else n
        if self.kw("IS"):
            self.take()
            neg = False
            if self.kw("NOT"):
                self.take()
                neg = True
            if self.kw("NULL"):
                self.take()
                if a.type == UNKNOWN:
                    a = coerce(a, BOOLEAN)
                n = Node("special", BOOLEAN, "IS_NULL", [a])
                return Node("call", BOOLEAN, "NOT", [n]) if neg else n
            raise Unsupported("IS DISTINCT FROM")
        return a

    def additive(self):
        a = self.multiplicative()
        while self.peek() in (("op", "+"), ("op", "-")):
            op = self.take()[1]
            b = self.multiplicative()
            a = self.arith(op, a, b)
        if self.peek() == ("op", "||"):
            raise Unsupported("concatenation")
        return a

    def multiplicative(self):
        a = self.unary()
        while self.peek() in (("op", "*"), ("op", "/"), ("op", "%")):
            op = self.take()[1]
            b = self.unary()
            a = self.arith(op, a, b)
        return a

    def arith(self, op, a, b):
        t = common_type(a.type, b.type)
        if t not in (INTEGER, BIGINT, DOUBLE):
            raise Unsupported("arithmetic on non-numeric type")
        return Node("call", t, ARITH[op], [coerce(a, t), coerce(b, t)])

    def unary(self):
        if self.peek() == ("op", "-"):
            self.take()
            if self.peek() == ("num", "9223372036854775808"):   # the parser folds the sign into this one literal (Long.MIN_VALUE)
                self.take()
                return lit(-2**63, BIGINT)
            a = self.unary()
            if a.type not in (INTEGER, BIGINT, DOUBLE):
                raise Unsupported("negation of non-numeric type")
            return Node("call", a.type, "NEGATE", [a])
        if self.peek() == ("op", "+"):
            self.take()
            return self.unary()
        return self.primary()

    def type_name(self):
        k, v = self.take()
        if k != "id":
            raise Unsupported("type expected")
        name = v.lower()
        if name == "double" and self.kw("PRECISION"):
            self.take()
        if self.peek() == ("op", "("):   # varchar(n), decimal(p, s)
            raise Unsupported("parametrised type")
        if name not in TYPE_BY_NAME or name in ("varchar", "date"):
            raise Unsupported("type " + name)
        return TYPE_BY_NAME[name]

    def primary(self):
        k, v = self.take()
        if k == "num":
            if re.fullmatch(r"\d+", v):
                x = int(v)
                if x <= 2**31 - 1:
                    return lit(x, INTEGER)
                if x <= 2**63 - 1:
                    return lit(x, BIGINT)
                raise Unsupported("integer literal out of range")
            if "e" in v.lower():
                return lit(float(v), DOUBLE)
            raise Unsupported("decimal literal")
        if k == "str":
            return lit(v[1:-1].replace("''", "'"), VARCHAR)
        if k == "op" and v == "(":
            e = self.or_expr()
            self.expect_op(")")
            return e
        if k != "id":
            raise Unsupported(f"unexpected token {v!r}")
        u = v.upper()
        if u == "TRUE":
            return lit(True, BOOLEAN)
        if u == "FALSE":
            return lit(False, BOOLEAN)
        if u == "NULL":
            return lit(None, UNKNOWN)
        if u in ("BIGINT", "INTEGER", "DOUBLE") and (self.peek()[0] == "str" or (u == "DOUBLE" and self.kw("PRECISION"))):
            if self.kw("PRECISION"):
                self.take()
            s = self.take()[1][1:-1].strip()
            if u == "DOUBLE":
                low = s.lower().lstrip("+-")
                if low == "nan":
                    return lit(float("nan"), DOUBLE)
                if low in ("infinity", "inf"):
                    return lit(float("-inf") if s.startswith("-") else float("inf"), DOUBLE)
                return lit(float(s), DOUBLE)
            x = int(s)
            if u == "INTEGER" and not -2**31 <= x <= 2**31 - 1:
                raise Unsupported("integer literal out of range")
            return lit(x, BIGINT if u == "BIGINT" else INTEGER)
        if u in ("INFINITY", "NAN") and self.peek() == ("op", "(") and self.peek(1) == ("op", ")"):   # MathFunctions.infinity() / nan()
            self.take()
            self.take()
            return lit(float("inf") if u == "INFINITY" else float("nan"), DOUBLE)
        if u == "CAST":
            self.expect_op("(")
            a = self.or_expr()
            self.expect_kw("AS")
            t = self.type_name()
            self.expect_op(")")
            if a.type == UNKNOWN:
                return lit(None, t)
            if a.type == VARCHAR:
                raise Unsupported("cast from varchar")
            if a.type == t:
                return Node("call", t, "CAST", [a])
            return Node("call", t, "CAST", [a])
        if u == "IF":
            self.expect_op("(")
            c = self.or_expr()
            self.expect_op(",")
            a = self.or_expr()
            self.expect_op(",")
            b = self.or_expr()
            self.expect_op(")")
            t = common_type(a.type, b.type)
            if t == UNKNOWN:
                raise Unsupported("IF of untyped nulls")
            return Node("special", t, "IF", [self.boolean(c), coerce(a, t), coerce(b, t)])
        if u == "COALESCE":
            self.expect_op("(")
            args = [self.or_expr()]
            while self.peek() == ("op", ","):
                self.take()
                args.append(self.or_expr())
            self.expect_op(")")
            t = UNKNOWN
            for a in args:
                t = common_type(t, a.type)
            if t == UNKNOWN:
                raise Unsupported("COALESCE of untyped nulls")
            if len(args) > 3:
                raise Unsupported("COALESCE with more than 3 arguments")
            return Node("special", t, "COALESCE", [coerce(a, t) for a in args])
        if u == "NULLIF":
            # NullIfCodeGenerator: the first argument unless first = second is true
            self.expect_op("(")
            a = self.or_expr()
            self.expect_op(",")
            b = self.or_expr()
            self.expect_op(")")
            if a.type == UNKNOWN:
                raise Unsupported("NULLIF of an untyped null")
            t = common_type(a.type, b.type)
            eq = Node("call", BOOLEAN, "EQUAL", [coerce(a, t), coerce(b, t)])
            return Node("special", a.type, "IF", [eq, lit(None, a.type), a])
        raise Unsupported("identifier " + v)


def parse(sql):
    return Parser(sql).parse()


class Lowering:
    """AST -> RowExpression.  hoist=False: literals become constants.  hoist=True: every distinct (type, value) literal becomes an
    input channel of a one-row page (columns / types collected in .columns)."""

    def __init__(self, E, hoist):
        self.E, self.hoist = E, hoist
        self.columns = []     # (type, value)
        self._index = {}

    def literal(self, n):
        if n.type == UNKNOWN:
            raise Unsupported("untyped null")
        if not self.hoist:
            return self.E.constant(n.value, n.type)
        v = n.value
        key = (n.type, repr(v)) if not (isinstance(v, float) and v != v) else (n.type, "nan")
        if key not in self._index:
            self._index[key] = len(self.columns)
            self.columns.append((n.type, v))
        return self.E.field(self._index[key], n.type)

    def lower(self, n):
        E = self.E
        if n.kind == "lit":
            return self.literal(n)
        args = [self.lower(a) for a in n.args]
        if n.kind == "call":
            return E.call(n.name, n.type, *args)
        return E.SpecialForm(n.name, n.type, args)


def literal_count(n):
    if n.kind == "lit":
        return 1
    return sum(literal_count(a) for a in n.args)


def expected_value(case_type, expected):
    """JSON form -> python value (doubles: 'NaN' / 'Infinity' / '-Infinity' strings)"""
    if expected is None:
        return None
    if case_type == "double":
        if isinstance(expected, str):
            return {"NaN": float("nan"), "Infinity": float("inf"), "-Infinity": float("-inf")}[expected]
        return float(expected)
    return expected


def same_value(type_id, got, want):
    if got is None or want is None:
        return got is None and want is None
    if type_id == DOUBLE:
        if math.isnan(want):
            return math.isnan(got)
        return struct.pack("<d", float(got)) == struct.pack("<d", float(want))   # bit-exact, -0.0 != 0.0
    if type_id == BOOLEAN:
        return bool(got) == bool(want)
    return got == want
