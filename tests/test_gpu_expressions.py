"""GPU side of the expression known answers: the same vectors tests/test_expression_golden_cpu.py pins the oracle on
(tests/golden/expression_vectors.json, tests/golden/reference_vectors.json: T/type/Test*Operators.java,
T/sql/gen/TestExpressionCompiler.java, T/operator/TestFilterAndProjectOperator.java, T/operator/project/TestPageProcessor.java)
run through FilterAndProjectOperator over the C ABI, i.e. through the hiprtc-generated gfx950 kernels.  Bit-exact (doubles by bit
pattern, NaN == NaN)."""
import json
import os

import pytest

import expr_harness as H

pytestmark = pytest.mark.gpu
REF_GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("hoist", [False, True], ids=["constants", "columns"])
def test_page_processor_matches_operator_known_answers(pkg, ctx, hoist):
    checked, bad = H.run_value_cases(pkg.expressions, H.product_engine(pkg, ctx), hoist)
    assert not bad, bad[:5]
    assert checked >= 360


@pytest.mark.parametrize("hoist", [False, True], ids=["constants", "columns"])
def test_page_processor_raises_the_reference_errors(pkg, ctx, hoist):
    checked, bad = H.run_error_cases(pkg.expressions, H.product_engine(pkg, ctx), hoist, pkg.TgpuError)
    assert not bad, bad[:5]
    assert checked >= 18


def test_page_processor_matches_expression_compiler_loops(pkg, ctx):
    checked, bad = H.run_loops(pkg.expressions, H.product_engine(pkg, ctx))
    assert not bad, bad[:5]
    assert checked > 5000


def test_filter_project_and_page_processor_fixtures(pkg, ctx):
    H.run_page_fixtures(pkg.expressions, H.product_engine(pkg, ctx), REF_GOLD)


def test_loops_match_the_oracle_cell_for_cell(pkg, ctx, oracle):
    """belt and braces: on the loop pages the HIP path and the oracle are also compared with each other"""
    got, want = [], []

    class Tee(H.Engine):
        def pages(self, types, pages, filt, projs):
            a = H.product_engine(pkg, ctx).pages(types, pages, filt, projs)
            b = H.oracle_engine(oracle, pkg.expressions).pages(types, pages, filt, projs)
            got.append(repr(a))
            want.append(repr(b))
            return a

    H.run_loops(pkg.expressions, Tee())
    assert got == want
