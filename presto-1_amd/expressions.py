"""RowExpression IR: the input language of the GPU page-processor compiler.

Mirrors io.trino.sql.relational.{InputReferenceExpression, ConstantExpression, CallExpression, SpecialForm}
(M/sql/relational/*.java; forms: SpecialForm.java:137-152).  `a > b` is kept as GREATER_THAN here; the reference's
translator rewrites it to `b < a` (SqlToRowExpressionTranslator.java:317-320), which is equivalent for the types supported.
"""
import ctypes as C

from . import _lib
from .spi import BIGINT, BOOLEAN, DATE, DOUBLE, INTEGER, VARCHAR

EX_INPUT, EX_CONST, EX_CALL, EX_SPECIAL = 0, 1, 2, 3
OPS = {"ADD": 1, "SUBTRACT": 2, "MULTIPLY": 3, "DIVIDE": 4, "MODULUS": 5, "NEGATE": 6, "EQUAL": 7, "NOT_EQUAL": 8,
       "LESS_THAN": 9, "LESS_THAN_OR_EQUAL": 10, "GREATER_THAN": 11, "GREATER_THAN_OR_EQUAL": 12, "NOT": 13, "CAST": 14}
FORMS = {"AND": 1, "OR": 2, "IF": 3, "IS_NULL": 4, "COALESCE": 5, "BETWEEN": 6}
_CMP = {"EQUAL", "NOT_EQUAL", "LESS_THAN", "LESS_THAN_OR_EQUAL", "GREATER_THAN", "GREATER_THAN_OR_EQUAL"}


class RowExpression:
    type = None

    # small builder sugar so tests read like SQL
    def __add__(self, o): return call("ADD", self.type, self, _wrap(o, self.type))
    def __sub__(self, o): return call("SUBTRACT", self.type, self, _wrap(o, self.type))
    def __mul__(self, o): return call("MULTIPLY", self.type, self, _wrap(o, self.type))
    def __truediv__(self, o): return call("DIVIDE", self.type, self, _wrap(o, self.type))
    def __mod__(self, o): return call("MODULUS", self.type, self, _wrap(o, self.type))
    def __neg__(self): return call("NEGATE", self.type, self)
    def __lt__(self, o): return call("LESS_THAN", BOOLEAN, self, _wrap(o, self.type))
    def __le__(self, o): return call("LESS_THAN_OR_EQUAL", BOOLEAN, self, _wrap(o, self.type))
    def __gt__(self, o): return call("GREATER_THAN", BOOLEAN, self, _wrap(o, self.type))
    def __ge__(self, o): return call("GREATER_THAN_OR_EQUAL", BOOLEAN, self, _wrap(o, self.type))
    def eq(self, o): return call("EQUAL", BOOLEAN, self, _wrap(o, self.type))
    def ne(self, o): return call("NOT_EQUAL", BOOLEAN, self, _wrap(o, self.type))


class InputReferenceExpression(RowExpression):
    def __init__(self, field, type_id):
        self.field, self.type = field, type_id


class ConstantExpression(RowExpression):
    def __init__(self, value, type_id):
        self.value, self.type = value, type_id


class CallExpression(RowExpression):
    def __init__(self, name, type_id, arguments):
        assert name in OPS, name
        self.name, self.type, self.arguments = name, type_id, list(arguments)


class SpecialForm(RowExpression):
    def __init__(self, form, type_id, arguments):
        assert form in FORMS, form
        self.form, self.type, self.arguments = form, type_id, list(arguments)


def _wrap(v, type_id):
    return v if isinstance(v, RowExpression) else ConstantExpression(v, type_id)


def field(channel, type_id):
    return InputReferenceExpression(channel, type_id)


def constant(value, type_id):
    return ConstantExpression(value, type_id)


def call(name, type_id, *args):
    return CallExpression(name, type_id, args)


def and_(a, b): return SpecialForm("AND", BOOLEAN, [a, b])
def or_(a, b): return SpecialForm("OR", BOOLEAN, [a, b])
def not_(a): return call("NOT", BOOLEAN, a)
def is_null(a): return SpecialForm("IS_NULL", BOOLEAN, [a])
def if_(c, t, f): return SpecialForm("IF", t.type, [c, t, f])
def coalesce(*args): return SpecialForm("COALESCE", args[0].type, list(args))
def between(v, lo, hi): return SpecialForm("BETWEEN", BOOLEAN, [v, _wrap(lo, v.type), _wrap(hi, v.type)])
def cast(a, type_id): return call("CAST", type_id, a)


class FlatProgram:
    """A filter + projections flattened into one node array (the serialisation the C ABI takes)."""

    def __init__(self, filter_expr, projections):
        self.nodes = []  # plain dicts: the serialisable form of the program
        self.pool = bytearray()
        self.filter_root = -1 if filter_expr is None else self._add(filter_expr)
        self.projection_roots = [self._add(p) for p in projections]

    def _add(self, e):
        if isinstance(e, InputReferenceExpression):
            nd = dict(kind=EX_INPUT, type=e.type, op=e.field, args=[])
        elif isinstance(e, ConstantExpression):
            nd = dict(kind=EX_CONST, type=e.type, op=0, args=[], is_null=int(e.value is None))
            if e.value is not None:
                if e.type == DOUBLE:
                    nd["dval"] = float(e.value)
                elif e.type == VARCHAR:
                    b = e.value.encode("utf-8") if isinstance(e.value, str) else bytes(e.value)
                    nd["ival"], nd["slen"] = len(self.pool), len(b)
                    self.pool += b
                else:
                    nd["ival"] = int(e.value)
        elif isinstance(e, CallExpression):
            args = [self._add(a) for a in e.arguments]
            nd = dict(kind=EX_CALL, type=e.type, op=OPS[e.name], args=args)
        elif isinstance(e, SpecialForm):
            args = [self._add(a) for a in e.arguments]
            assert len(args) <= 3, "special forms take at most 3 arguments here"
            nd = dict(kind=EX_SPECIAL, type=e.type, op=FORMS[e.form], args=args)
        else:
            raise TypeError(e)
        self.nodes.append(nd)
        return len(self.nodes) - 1

    def to_c(self):
        arr = (_lib.ExprNode * max(1, len(self.nodes)))()
        for i, nd in enumerate(self.nodes):
            x = arr[i]
            x.kind, x.type, x.op = nd["kind"], nd["type"], nd["op"]
            x.n_args = len(nd["args"])
            for k, a in enumerate(nd["args"]):
                x.args[k] = a
            x.is_null = nd.get("is_null", 0)
            x.ival = nd.get("ival", 0)
            x.dval = nd.get("dval", 0.0)
            x.slen = nd.get("slen", 0)
        roots = (C.c_int32 * max(1, len(self.projection_roots)))(*self.projection_roots)
        pool = bytes(self.pool)
        spec = _lib.PageProcessorSpec(arr, len(self.nodes), pool, len(pool), self.filter_root, len(self.projection_roots), roots)
        return spec, [arr, roots, pool]
