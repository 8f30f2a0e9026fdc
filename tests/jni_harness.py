"""Builds jni/tgpu_jni.c + tests/jni_stub/fake_jvm.c into one shared object (against tests/jni_stub/jni.h, the self-written JNI declaration
stub) and wraps the fake JVM for the tests: arrays, object arrays, the pending NativeError, pin / frame counters, page-source adapters."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "jni_stub")
OUT = os.path.join(STUB, "build", "libtgpu_jni_fake.so")


def header_symbols():
    text = open(os.path.join(ROOT, "include", "tgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tgpu_[a-z0-9_]+)\s*\(", text)))


def build_fake_jni(force=False):
    srcs = [os.path.join(ROOT, "jni", "tgpu_jni.c"), os.path.join(STUB, "fake_jvm.c")]
    deps = srcs + [os.path.join(STUB, "jni.h"), os.path.join(ROOT, "include", "tgpu.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-I" + STUB, "-I" + os.path.join(ROOT, "include")] + srcs +
                          ["-L" + os.path.join(ROOT, "presto-1_amd"), "-ltgpu", "-Wl,-rpath," + os.path.join(ROOT, "presto-1_amd"), "-Wl,--no-undefined", "-o", OUT])
    return OUT


NEXT_PAGE = C.CFUNCTYPE(C.c_int32)
BOOL_FN = C.CFUNCTYPE(C.c_uint8)
LOAD_BLOCK = C.CFUNCTYPE(C.c_void_p, C.c_int32)
VOID_FN = C.CFUNCTYPE(None)


class FakeJvm:
    PREFIX = "Java_io_trino_operator_gpu_GpuNative_"

    def __init__(self, path):
        self.lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        L = self.lib
        L.fj_env.restype = C.c_void_p
        L.fj_new_array.restype = C.c_void_p
        L.fj_new_array.argtypes = [C.c_int32, C.c_size_t, C.c_void_p]
        L.fj_new_object_array.restype = C.c_void_p
        L.fj_new_object_array.argtypes = [C.c_int32]
        L.fj_set_object.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.fj_array_data.restype = C.c_void_p
        L.fj_array_data.argtypes = [C.c_void_p]
        L.fj_array_length.argtypes = [C.c_void_p]
        L.fj_string_chars.restype = C.c_char_p
        L.fj_string_chars.argtypes = [C.c_void_p]
        L.fj_pending_message.restype = C.c_char_p
        L.fj_new_adapter.restype = C.c_void_p
        L.fj_new_adapter.argtypes = [NEXT_PAGE, BOOL_FN, BOOL_FN, LOAD_BLOCK, VOID_FN]
        self.env = C.c_void_p(L.fj_env())
        self.keep = []

    # ---- objects ----
    def array(self, a):
        if a is None:
            return C.c_void_p(None)
        a = np.ascontiguousarray(a)
        return C.c_void_p(self.lib.fj_new_array(len(a), a.dtype.itemsize, a.ctypes.data_as(C.c_void_p)))

    def empty(self, n, dtype):
        return self.array(np.zeros(n, dtype=dtype))

    def object_array(self, items):
        o = C.c_void_p(self.lib.fj_new_object_array(len(items)))
        for i, it in enumerate(items):
            if it is not None:
                self.lib.fj_set_object(o, i, it)
        return o

    def read(self, arr, dtype):
        n = self.lib.fj_array_length(arr)
        p = self.lib.fj_array_data(arr)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(np.dtype(dtype)))), shape=(max(n, 1),))[:n].copy()

    def string(self, s):
        return self.lib.fj_string_chars(s).decode()

    def adapter(self, next_page, is_finished, is_blocked, load_block, close):
        cbs = [NEXT_PAGE(next_page), BOOL_FN(is_finished), BOOL_FN(is_blocked), LOAD_BLOCK(load_block), VOID_FN(close)]
        self.keep.append(cbs)
        return C.c_void_p(self.lib.fj_new_adapter(*cbs))

    # ---- calls ----
    def call(self, name, restype, *args):
        fn = getattr(self.lib, self.PREFIX + name)
        fn.restype = restype
        return fn(self.env, None, *args)

    def checked(self, name, restype, *args):
        r = self.call(name, restype, *args)
        if self.lib.fj_pending_code() != 0:
            code, msg = self.pending_code(), self.pending_message()
            self.clear()
            raise RuntimeError(f"NativeError({code}): {msg}")
        return r

    def pending_code(self):
        return self.lib.fj_pending_code()

    def pending_message(self):
        return self.lib.fj_pending_message().decode()

    def clear(self):
        self.lib.fj_clear_pending()

    def outstanding_pins(self):
        return self.lib.fj_outstanding_pins()

    def open_frames(self):
        return self.lib.fj_open_frames()

    def calls_while_pinned(self):
        return self.lib.fj_calls_while_pinned()

    def global_refs(self):
        return self.lib.fj_global_refs()
