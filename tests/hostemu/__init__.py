"""TEST HARNESS: the product's per-record core compiled for the host (see hostemu.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

from rnaseqc_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhostemu.so")
_ROOT = os.path.dirname(os.path.dirname(_HERE))


def build():
    srcs = [os.path.join(_HERE, "hostemu.cpp"), os.path.join(_ROOT, "rnaseqc_amd", "csrc", "rsqc_read.h"),
            os.path.join(_ROOT, "rnaseqc_amd", "csrc", "rsqc_index.h"), os.path.join(_ROOT, "include", "rnaseqc_amd.h")]
    if not os.path.exists(_SO) or any(os.path.getmtime(_SO) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", srcs[0], "-o", _SO])
    return _SO


class Out:
    pass


def run(params, ann, batch):
    lib = C.CDLL(build())
    a, b = ann.to_struct(), batch.to_struct()
    o = Out()
    G, E = ann.n_genes_listed, ann.n_exons
    o.counters = np.zeros(abi.N_COUNTERS, np.uint64)
    o.gene_reads = np.zeros(G, np.uint64); o.gene_unique = np.zeros(G, np.uint64); o.gene_fragments = np.zeros(G, np.uint64)
    o.exon_reads = np.zeros(E, np.float64)
    rl, nov = C.c_int32(), C.c_uint64()
    rc = lib.hostemu_run(C.byref(params), C.byref(a), C.byref(b), abi.ptr(o.counters), abi.ptr(o.gene_reads),
                         abi.ptr(o.gene_unique), abi.ptr(o.gene_fragments), abi.ptr(o.exon_reads), C.byref(rl), None,
                         C.byref(nov))
    if rc:
        raise RuntimeError("hostemu rc=%d" % rc)
    o.read_length = rl.value
    o.n_overflow = nov.value
    return o
