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
        _lib.oracle_set_reference.argtypes = [C.c_void_p, C.POINTER(abi.ReferenceStruct)]
        _lib.oracle_submit.argtypes = [C.c_void_p, C.POINTER(abi.BatchStruct)]
        _lib.oracle_finalize.argtypes = [C.c_void_p, C.POINTER(abi.ResultsStruct)]
        _lib.oracle_destroy.argtypes = [C.c_void_p]
        _lib.oracle_destroy.restype = None
        _lib.oracle_last_error.argtypes = [C.c_void_p]
        _lib.oracle_last_error.restype = C.c_char_p
        _lib.oracle_enable_trace.argtypes = [C.c_void_p]
        _lib.oracle_get_trace.argtypes = [C.c_void_p, C.POINTER(OracleTrace)]
        _lib.oracle_exit_order.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
        _lib.oracle_median_f64.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]
        _lib.oracle_statistics.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]
        _lib.oracle_statistics.restype = None
        _lib.oracle_library_complexity.argtypes = [C.c_double, C.c_double, C.c_double]
        _lib.oracle_library_complexity.restype = C.c_uint
    return _lib


class OracleTrace(C.Structure):
    _fields_ = [("n_eligible", C.c_uint64), ("span", C.c_void_p), ("l_qseq", C.c_void_p),
                ("n_batches", C.c_uint64), ("batch_end", C.c_void_p), ("batch_file_index", C.c_void_p),
                ("n_samples", C.c_uint64), ("sample_file_index", C.c_void_p), ("sample_size", C.c_void_p)]


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

    def set_reference(self, ref):
        s = ref.to_struct()
        self._keep += [ref, s]
        self._check(self._l.oracle_set_reference(self._h, C.byref(s)))

    def submit(self, batch):
        s = batch.to_struct()
        self._check(self._l.oracle_submit(self._h, C.byref(s)))

    def finalize(self) -> abi.Results:
        rs = abi.ResultsStruct()
        self._check(self._l.oracle_finalize(self._h, C.byref(rs)))
        return abi.Results(rs).materialise()

    def enable_trace(self):
        self._check(self._l.oracle_enable_trace(self._h))

    def shard_info(self):
        """distributed.ShardInfo of what this oracle instance processed (needs enable_trace() before the first submit;
        run with fragment_samples = 2^32 - 1 so that the shard's sampler is not cut off before the merge)."""
        from rnaseqc_amd import distributed
        t = OracleTrace()
        self._check(self._l.oracle_get_trace(self._h, C.byref(t)))
        span = abi._view(t.span, t.n_eligible, np.uint32); lq = abi._view(t.l_qseq, t.n_eligible, np.int32)
        ends = abi._view(t.batch_end, t.n_batches, np.uint64).astype(np.int64)
        off, keys, vals = [0], [], []
        lo = 0
        for hi in ends:
            k, v = distributed.read_length_transfer(span[lo:hi], lq[lo:hi])
            keys.append(k); vals.append(v); off.append(off[-1] + len(k)); lo = int(hi)
        return distributed.ShardInfo(
            batch_file_index=abi._view(t.batch_file_index, t.n_batches, np.uint64), batch_records=np.zeros(t.n_batches, np.uint64),
            rl_offset=np.array(off, np.uint32), rl_span=np.concatenate(keys) if keys else np.zeros(0, np.uint32),
            rl_state=np.concatenate(vals) if vals else np.zeros(0, np.int32),
            sample_file_index=abi._view(t.sample_file_index, t.n_samples, np.uint64),
            sample_size=abi._view(t.sample_size, t.n_samples, np.uint32))

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


def run_oracle(params, ann, batches, bed=None, owned=None, reference=None) -> abi.Results:
    o = Oracle(params)
    o.set_annotation(ann, owned)
    if bed is not None:
        o.set_bed(bed)
    if reference is not None:
        o.set_reference(reference)
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


# ---------------------------------------------------------------------------
# oracle/_ref/libref_metrics.so: the reference's own src/Metrics.cpp (see
# ref_metrics_harness.cpp).  Present only where /root/reference was available
# at build time (this container); it travels to the GPU box as a built file.
_REF_SO = os.path.join(_HERE, "_ref", "libref_metrics.so")
_ref = None


def build_ref() -> bool:
    subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return os.path.exists(_REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        if not os.path.exists(_REF_SO):
            return None
        _ref = C.CDLL(_REF_SO)
        _ref.ref_median.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]
        _ref.ref_statistics.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]
        _ref.ref_statistics.restype = None
        if hasattr(_ref, "ref_advanced_statistics"):
            _ref.ref_advanced_statistics.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]
            _ref.ref_advanced_statistics.restype = None
        _ref.ref_metrics_print.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_char_p]
        _ref.ref_frac.argtypes = [C.c_uint64, C.c_uint64]
        _ref.ref_frac.restype = C.c_double
        _ref.ref_coverage_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int,
                                          C.c_ulong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p]
    return _ref


def ref_median(values):
    a = np.ascontiguousarray(values, dtype=np.float64)
    out = C.c_double()
    rc = ref_lib().ref_median(abi.ptr(a), len(a), C.byref(out))
    if rc:
        raise OracleError(rc, "reference computeMedian threw range_error")
    return out.value


def ref_statistics(values):
    a = np.ascontiguousarray(values, dtype=np.float64)
    out = (C.c_double * 4)()
    ref_lib().ref_statistics(abi.ptr(a), len(a), out)
    return tuple(out)


def ref_advanced_statistics(values):
    a = np.ascontiguousarray(values, dtype=np.uint32)
    out = (C.c_double * 4)()
    ref_lib().ref_advanced_statistics(abi.ptr(a), len(a), out)
    return list(out)


def ref_metrics_print(counter_dict, path):
    names = list(counter_dict.keys())
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    vals = np.array([counter_dict[n] for n in names], dtype=np.uint64)
    rc = ref_lib().ref_metrics_print(len(names), arr, abi.ptr(vals), path.encode())
    if rc:
        raise OracleError(rc, "ref_metrics_print")


def ref_coverage_run(gene_exon_off, exon_len, gene_strand, commit_exon, commit_off, commit_len, commit_read,
                     mask=500, bias_offset=0, bias_window=100, bias_gene_length=200, coverage_tsv=None):
    G = len(gene_exon_off) - 1
    E = int(gene_exon_off[-1])
    a = [np.ascontiguousarray(gene_exon_off, np.uint32), np.ascontiguousarray(exon_len, np.int64),
         np.ascontiguousarray(gene_strand, np.int32), np.ascontiguousarray(commit_exon, np.uint32),
         np.ascontiguousarray(commit_off, np.int64), np.ascontiguousarray(commit_len, np.uint32),
         np.ascontiguousarray(commit_read, np.uint32)]
    out = dict(gene_valid=np.zeros(G, np.uint8), gene_mean=np.zeros(G), gene_std=np.zeros(G), gene_cv=np.zeros(G),
               exon_cv_valid=np.zeros(E, np.uint8), exon_cv=np.zeros(E), bias_ratio=np.zeros(G))
    rc = ref_lib().ref_coverage_run(G, abi.ptr(a[0]), abi.ptr(a[1]), abi.ptr(a[2]), len(a[3]), abi.ptr(a[3]),
                                    abi.ptr(a[4]), abi.ptr(a[5]), abi.ptr(a[6]), mask, bias_offset, bias_window,
                                    bias_gene_length, abi.ptr(out["gene_valid"]), abi.ptr(out["gene_mean"]),
                                    abi.ptr(out["gene_std"]), abi.ptr(out["gene_cv"]), abi.ptr(out["exon_cv_valid"]),
                                    abi.ptr(out["exon_cv"]), abi.ptr(out["bias_ratio"]),
                                    coverage_tsv.encode() if coverage_tsv else None)
    if rc < 0:
        raise OracleError(rc, "reference coverage path threw")
    out["counted_genes"] = rc
    return out
