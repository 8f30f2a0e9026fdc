"""The JNI shim (jni/tgpu_jni.c) without a JDK: compiled -- every line -- against tests/jni_stub/jni.h (a self-written declaration stub, labelled
as such) with -Wall -Wextra -Werror, linked against libtgpu.so (--no-undefined: a call of a function the library does not export fails the link),
compared signature by signature with java/io/trino/operator/gpu/GpuNative.java, and EXECUTED on tests/jni_stub/fake_jvm.c for the paths that need
no GPU: error -> NativeError(code, message), argument validation in front of the library, pin / local-frame discipline.
The GPU flows through the shim are in tests/test_gpu_jni_shim.py."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from jni_harness import ROOT, FakeJvm, build_fake_jni, header_symbols

JAVA_TO_JNI = {"int": "jint", "long": "jlong", "boolean": "jboolean", "int[]": "jintArray", "long[]": "jlongArray", "double[]": "jdoubleArray", "byte[]": "jbyteArray",
               "boolean[]": "jbooleanArray", "Object[]": "jobjectArray", "int[][]": "jobjectArray", "String": "jstring", "GpuPageSource": "jobject", "void": "void"}
MACROS = {"PROGRAM_PARAMS": "jobjectArray nodes, jlongArray ivals, jdoubleArray dvals, jbyteArray pool, jint filterRoot, jintArray projectionRoots",
          "PAGE_PARAMS": "jint positions, jintArray types, jintArray encodings, jintArray arrayOffsets, jintArray dictPositions, jobjectArray values, jobjectArray nulls, "
                         "jobjectArray offsets, jobjectArray ids, jobjectArray dvalues, jobjectArray dnulls, jobjectArray doffsets"}


def shim_signatures():
    c = open(os.path.join(ROOT, "jni", "tgpu_jni.c")).read()
    c = re.sub(r"/\*.*?\*/", "", c, flags=re.S)
    sigs = {}
    for ret, name, params in re.findall(r"\bJFN\((\w+), (\w+)\)\(JNIEnv \*env, jclass c([^)]*)\)\s*\{", c):
        for m, text in MACROS.items():
            params = params.replace(m, text)
        types = [p.strip().rsplit(" ", 1)[0].strip() for p in params.split(",") if p.strip()]
        sigs[name] = (ret, types)
    for name in re.findall(r"^BOOL_CALL\((\w+),", c, flags=re.M):
        sigs[name] = ("jboolean", ["jlong"])
    for name in re.findall(r"^VOID_OP_CALL\((\w+),", c, flags=re.M):
        sigs[name] = ("void", ["jlong"])
    return sigs


def java_signatures():
    java = open(os.path.join(ROOT, "java", "io", "trino", "operator", "gpu", "GpuNative.java")).read()
    java = re.sub(r"/\*.*?\*/", "", java, flags=re.S)
    java = re.sub(r"//[^\n]*", "", java)
    sigs = {}
    for ret, name, params in re.findall(r"public static native ([\w\[\]]+) (\w+)\((.*?)\);", java, flags=re.S):
        types = [p.strip().rsplit(" ", 1)[0].strip() for p in params.split(",") if p.strip()]
        sigs[name] = (JAVA_TO_JNI[ret] if ret != "long[]" and ret != "byte[]" else JAVA_TO_JNI[ret], [JAVA_TO_JNI[t] for t in types])
    return sigs


def test_shim_and_gpu_native_declare_the_same_methods_with_the_same_signatures():
    c, j = shim_signatures(), java_signatures()
    assert set(c) == set(j), set(c) ^ set(j)
    assert len(c) >= 65
    for name in sorted(c):
        assert c[name] == j[name], (name, c[name], j[name])


def test_shim_only_calls_functions_the_header_declares_and_binds_every_operator_factory():
    c = open(os.path.join(ROOT, "jni", "tgpu_jni.c")).read()
    called = set(re.findall(r"\b(tgpu_[a-z0-9_]+)\(", c)) | set(re.findall(r"_CALL\(\w+, (tgpu_[a-z0-9_]+)\)", c))
    declared = set(header_symbols())
    assert called <= declared, called - declared
    # every factory of the header (what LocalExecutionPlanner would construct) is reachable from Java
    factories = {n for n in declared if n.endswith("_factory_create")}
    assert factories <= called, factories - called
    # what stays unbound is the test / tooling surface, by name
    unbound = declared - called
    allowed = {"tgpu_group_by_hash_create", "tgpu_group_by_hash_destroy", "tgpu_group_by_hash_add_page", "tgpu_group_by_hash_get_group_ids", "tgpu_group_by_hash_contains",
               "tgpu_group_by_hash_group_count", "tgpu_group_by_hash_capacity", "tgpu_group_by_hash_estimated_size", "tgpu_group_by_hash_rehash_count",
               "tgpu_group_by_hash_append_values", "tgpu_hash_page", "tgpu_partition_page", "tgpu_profile_reset", "tgpu_version", "tgpu_set_resource_dir",
               "tgpu_pinned_alloc", "tgpu_pinned_free", "tgpu_output_page_copy_block",
               "tgpu_exchange_create_with_transport", "tgpu_partitioned_join_position_encode",
               "tgpu_scan_operator_add_record_cursor",   # (Java keeps a RecordPageSource as the page source it is: INTEGRATION.md)
               "tgpu_partitioned_join_position_decode", "tgpu_lookup_source_factory_destroy"}
    assert unbound <= allowed | {"tgpu_lookup_source_factory_destroy"}, unbound - allowed


@pytest.fixture(scope="module")
def jvm():
    return FakeJvm(build_fake_jni())


def test_shim_compiles_against_the_stub_with_all_warnings_as_errors_and_links_against_the_library():
    path = build_fake_jni(force=True)   # gcc -Wall -Wextra -Werror ... -Wl,--no-undefined
    assert os.path.exists(path)


def test_a_failed_native_call_becomes_native_error_with_code_and_message(jvm):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: createContext succeeds")
    assert jvm.call("createContext", C.c_int64, C.c_int32(0)) == 0
    assert jvm.pending_code() == -6 and "HIP" in jvm.pending_message().upper() or "device" in jvm.pending_message().lower()
    jvm.clear()


def test_argument_validation_happens_in_front_of_the_library(jvm):
    # deserializePage: offset / length outside the array (unchecked p + offset)
    data = jvm.array(np.zeros(16, dtype=np.int8))
    types = jvm.array(np.array([1], dtype=np.int32))
    assert jvm.call("deserializePage", C.c_int64, C.c_int64(0), data, C.c_int32(8), C.c_int32(16), types) == 0
    assert jvm.pending_code() == -1 and "offset" in jvm.pending_message()
    jvm.clear()
    # addInput: a values array shorter than arrayOffset + positions never reaches the library (operator handle 0 would crash it)
    n = 1
    vals = jvm.object_array([jvm.array(np.arange(5, dtype=np.int64))])
    empty = jvm.object_array([None])
    jvm.call("addInput", None, C.c_int64(0), C.c_int32(10), jvm.array(np.array([1], dtype=np.int32)), jvm.array(np.zeros(n, dtype=np.int32)),
             jvm.array(np.zeros(n, dtype=np.int32)), jvm.array(np.zeros(n, dtype=np.int32)), vals, empty, empty, empty, empty, empty, empty)
    assert jvm.pending_code() == -1
    jvm.clear()
    # a malformed aggregate array (not triples)
    ints = lambda *v: jvm.array(np.array(v, dtype=np.int32))
    assert jvm.call("createHashAggregationFactory", C.c_int64, C.c_int64(0), C.c_int32(0), ints(1), ints(0), C.c_int32(-1), C.c_int32(0), ints(1, 0), C.c_int32(10),
                    C.c_uint8(1)) == 0
    assert jvm.pending_code() == -1
    jvm.clear()
    assert jvm.outstanding_pins() == 0 and jvm.open_frames() == 0 and jvm.calls_while_pinned() == 0


def test_position_encoding_validation(pkg):
    L = pkg._lib.lib()
    L.tgpu_partitioned_join_position_encode.restype = C.c_int64
    assert L.tgpu_partitioned_join_position_encode(3, 5, 8) == (5 << 4) | 3          # PartitionedLookupSource.java:222-226
    for bad in ((0, 0, 0), (0, 0, 6), (8, 0, 8), (-1, 0, 8), (0, -1, 8)):
        assert L.tgpu_partitioned_join_position_encode(*bad) < 0
