"""CPU-only checks of the boundary: libtgpu.so loads, exports every symbol include/tgpu.h declares, fails loudly
without a GPU (no CPU fallback), and the host-side mirrors (Page / Block / RowExpression) serialise correctly."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "tgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg._lib.lib()
    names = header_symbols()
    assert len(names) >= 40
    for name in names:
        assert hasattr(L, name), f"libtgpu.so does not export {name}"
    # and the ctypes table covers the whole header
    assert set(names) <= set(pkg._lib.SYMBOLS), set(names) - set(pkg._lib.SYMBOLS)


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.TgpuError) as e:
        pkg.Context(0)
    assert e.value.code == -6


def test_product_never_imports_oracle():
    pkg_dir = os.path.join(ROOT, "presto-1_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.replace("// oracle", ""), f"{f} mentions the oracle"


def test_block_round_trip(pkg):
    b = pkg.Block(pkg.VARCHAR, ["a", None, "ccc"])
    assert b.to_list() == ["a", None, "ccc"]
    assert list(b.offsets) == [0, 1, 1, 4]
    d = pkg.DictionaryBlock(pkg.Block(pkg.BIGINT, [10, 20, 30]), [2, 0, 2])
    assert d.to_list() == [30, 10, 30]
    r = pkg.RunLengthEncodedBlock(pkg.Block(pkg.DOUBLE, [1.5]), 4)
    assert r.to_list() == [1.5] * 4
    pg = pkg.Page(b, pkg.Block(pkg.BIGINT, [1, 2, 3]))
    cp, keep = pg.to_c()
    assert cp.position_count == 3 and cp.channel_count == 2
    assert cp.blocks[0].type == pkg.VARCHAR and cp.blocks[1].type == pkg.BIGINT


def test_flat_program_serialisation(pkg):
    f = pkg.field
    prog = pkg.expressions.FlatProgram(pkg.and_(f(0, pkg.BIGINT) > 5, f(3, pkg.VARCHAR).eq("BUILDING")), [f(1, pkg.BIGINT) * f(2, pkg.BIGINT)])
    assert prog.filter_root == len(prog.nodes) - 4 or prog.filter_root >= 0
    assert bytes(prog.pool) == b"BUILDING"
    spec, keep = prog.to_c()
    assert spec.node_count == len(prog.nodes) and spec.projection_count == 1


def test_jit_compiles_without_gpu(pkg, tmp_path):
    """The expression compiler (hiprtc, gfx950) runs on the CPU-only build box: this is build()'s pre-warm path."""
    f = pkg.field
    src = pkg.page_processor_source([pkg.BIGINT, pkg.DOUBLE], f(0, pkg.BIGINT) < 7, [f(1, pkg.DOUBLE) * (pkg.constant(1.0, pkg.DOUBLE) - f(1, pkg.DOUBLE))])
    assert "fp_count" in src and "fp_emit" in src and "__ballot" in src
    pkg.precompile_page_processor([pkg.BIGINT, pkg.DOUBLE], f(0, pkg.BIGINT) < 7, [f(1, pkg.DOUBLE) * (pkg.constant(1.0, pkg.DOUBLE) - f(1, pkg.DOUBLE))])
    with pytest.raises(pkg.TgpuError) as e:  # type errors surface as COMPILER_ERROR like PageFunctionCompiler.java:199-205
        pkg.precompile_page_processor([pkg.BIGINT], f(0, pkg.DOUBLE) < 7.0, [])
    assert e.value.code == -4


def test_every_handle_entry_point_binds_the_device():
    """source-level check (no GPU here): every extern "C" function of c_api.cpp that takes a handle and reaches device code goes
    through guard_on(ctx_of(handle), ...), which binds the calling thread to the context's device (hipSetDevice is per thread)"""
    text = open(os.path.join(ROOT, "presto-1_amd", "csrc", "c_api.cpp")).read()
    body = text[text.index('extern "C" {'):]
    fns = re.split(r"\n(?=(?:int32_t|int64_t|void|const char \*)\s*\*?tgpu_[a-z0-9_]+\()", body)
    handle = re.compile(r"(?:const\s+)?tgpu_(?:context|operator_factory|operator|lookup_source_factory|group_by_hash|output_page)\s*\*\s*[a-z_]+")
    checked = 0
    for fn in fns:
        sig = fn.split("{", 1)[0]
        if not handle.search(sig) or not re.search(r"\bguard(_on)?\(", fn):
            continue
        checked += 1
        assert "guard_on(ctx_of(" in fn, sig.strip()[:80]
    assert checked >= 40


# (the JNI shim and the Java glue: tests/test_jni_shim_cpu.py, tests/test_java_glue_cpu.py)
