"""ctypes binding of oracle/_build/librsqc_oracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from rnaseqc_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librsqc_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "rsqc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_create.argtypes = [C.POINTER(abi.Params), C.POINTER(C.c_void_p)]
        _lib.oracle_set_annotation.argtypes = [C.c_void_p, C.POINTER(abi.AnnotationStruct), C.c_void_p]
        _lib.oracle_set_bed.argtypes = [C.c_void_p, C.POINTER(abi.BedStruct)]
        _lib.oracle_submit.argtypes = [C.c_void_p, C.POINTER(abi.BatchStruct)]
        _lib.oracle_finalize.argtypes = [C.c_void_p, C.POINTER(abi.ResultsStruct)]
        _lib.oracle_destroy.argtypes = [C.c_void_p]
        _lib.oracle_destroy.restype = None
        _lib.oracle_last_error.argtypes = [C.c_void_p]
        _lib.oracle_last_error.restype = C.c_char_p
        _lib.oracle_exit_order.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
        _lib.oracle_median_f64.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]
        _lib.oracle_statistics.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]
        _lib.oracle_statistics.restype = None
        _lib.oracle_library_complexity.argtypes = [C.c_double, C.c_double, C.c_double]
        _lib.oracle_library_complexity.restype = C.c_uint
    return _lib


class OracleError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("oracle error %d %s" % (code, msg))
        self.code = code


class Oracle:
    """Sequential CPU oracle with the same call sequence as the product engine."""

    def __init__(self, params: abi.Params):
        self._l = lib()
        self._h = C.c_void_p()
        self._keep = []
        self._check(self._l.oracle_create(C.byref(params), C.byref(self._h)))

    def _check(self, rc):
        if rc != 0:
            raise OracleError(rc, self._l.oracle_last_error(self._h).decode() if self._h else "")

    def set_annotation(self, ann, owned=None):
        s = ann.to_struct()
        o = None if owned is None else np.ascontiguousarray(owned, dtype=np.uint8)
        self._keep += [ann, s, o]
        self._check(self._l.oracle_set_annotation(self._h, C.byref(s), abi.ptr(o)))

    def set_bed(self, bed):
        s = bed.to_struct()
        self._keep += [bed, s]
        self._check(self._l.oracle_set_bed(self._h, C.byref(s)))

    def submit(self, batch):
        s = batch.to_struct()
        self._check(self._l.oracle_submit(self._h, C.byref(s)))

    def finalize(self) -> abi.Results:
        rs = abi.ResultsStruct()
        self._check(self._l.oracle_finalize(self._h, C.byref(rs)))
        return abi.Results(rs)

    def exit_order(self) -> np.ndarray:
        p, n = C.c_void_p(), C.c_uint32()
        self._l.oracle_exit_order(self._h, C.byref(p), C.byref(n))
        return abi._view(p.value, n.value, np.uint32)

    def close(self):
        if self._h:
            self._l.oracle_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_oracle(params, ann, batches, bed=None, owned=None) -> abi.Results:
    o = Oracle(params)
    o.set_annotation(ann, owned)
    if bed is not None:
        o.set_bed(bed)
    for b in batches:
        o.submit(b)
    r = o.finalize()
    o.close()
    return r


def median(values) -> float:
    a = np.ascontiguousarray(values, dtype=np.float64)
    out = C.c_double()
    rc = lib().oracle_median_f64(abi.ptr(a), len(a), C.byref(out))
    if rc:
        raise OracleError(rc, "median of empty list")
    return out.value


def statistics(values):
    a = np.array(values, dtype=np.float64)
    out = (C.c_double * 4)()
    lib().oracle_statistics(abi.ptr(a), len(a), out)
    return tuple(out)


def library_complexity(dup, unique, limit=1e9) -> int:
    return int(lib().oracle_library_complexity(float(dup), float(unique), float(limit)))
