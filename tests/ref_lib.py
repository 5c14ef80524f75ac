"""ctypes face of oracle/libcruse_ref.so -- the plain-C CPU twins of the C ABI's core entry points (oracle/cruse_ref.c).
Test infrastructure: loaded by tests only.  The library is built on first use when it is not there (gcc; the GPU box receives the
prebuilt file with the snapshot and has gcc too)."""
import ctypes
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "oracle", "cruse_ref.c")
LIB = os.path.join(ROOT, "oracle", "libcruse_ref.so")
HEADER = os.path.join(ROOT, "include", "cruse_hip.h")


def build():
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"])


def load():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        build()
    return ctypes.CDLL(LIB)


def twin_names():
    """the *_ref functions the C source defines"""
    src = open(SRC).read()
    return sorted(set(re.findall(r"^int (cruse_[a-z0-9_]+)_ref\(", src, flags=re.M)))


def header_params(name):
    """parameter list (type strings) of an entry point in include/cruse_hip.h"""
    h = open(HEADER).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    m = re.search(r"\bint " + name + r"\(([^;]*?)\);", h, flags=re.S)
    assert m, name
    return [" ".join(p.split()) for p in m.group(1).split(",")]


def source_params(name):
    src = open(SRC).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    m = re.search(r"\bint " + name + r"_ref\(([^{]*?)\)\s*\{", src, flags=re.S)
    assert m, name
    return [" ".join(p.split()) for p in m.group(1).split(",")]


def _arg(a):
    if a is None:
        return ctypes.c_void_p(0)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return ctypes.c_void_p(a.ctypes.data)
    if isinstance(a, float):
        return ctypes.c_float(a)
    if isinstance(a, (int, np.integer)):
        return ctypes.c_longlong(int(a)) if abs(int(a)) >= 2 ** 31 else ctypes.c_int(int(a))
    return a


def call(lib, name, *args):
    """call twin `name` (without the _ref suffix); numpy arrays are passed by address, Python floats as C float, ints as C int
    (wrap long long parameters in ctypes.c_longlong)"""
    fn = getattr(lib, name + "_ref")
    fn.restype = ctypes.c_int
    rc = fn(*[_arg(a) for a in args])
    assert rc == 0, (name, rc)


LL = ctypes.c_longlong


def ptr_array(arrs):
    """HOST array of pointers (w_hh / b_hh of the GRU entry points)"""
    return (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
