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


def run(params, ann, batch, mode=1, want_cov=False):
    """mode 1: the elementary-interval feature stage (what the per-record kernel runs), 0: the row-table one."""
    lib = C.CDLL(build())
    lib.hostemu_set_mode(int(mode))
    a, b = ann.to_struct(), batch.to_struct()
    o = Out()
    G, E = ann.n_genes_listed, ann.n_exons
    o.counters = np.zeros(abi.N_COUNTERS, np.uint64)
    o.gene_reads = np.zeros(G, np.uint64); o.gene_unique = np.zeros(G, np.uint64); o.gene_fragments = np.zeros(G, np.uint64)
    o.exon_reads = np.zeros(E, np.float64)
    rl, nov = C.c_int32(), C.c_uint64()
    o.cov = None
    if want_cov:
        total = int(sum(int(ann.exon_row_end[i]) - int(ann.exon_row_start[i]) + 1 for i in range(E))) + ann.n_genes + 8
        o.cov = np.zeros(total, np.uint32)
    rc = lib.hostemu_run(C.byref(params), C.byref(a), C.byref(b), abi.ptr(o.counters), abi.ptr(o.gene_reads),
                         abi.ptr(o.gene_unique), abi.ptr(o.gene_fragments), abi.ptr(o.exon_reads), C.byref(rl),
                         abi.ptr(o.cov) if want_cov else None, C.byref(nov))
    if rc:
        raise RuntimeError("hostemu rc=%d" % rc)
    o.read_length = rl.value
    o.n_overflow = nov.value
    return o


_K1SO = os.path.join(_HERE, "libk1emu.so")
_K1SO_DEFAULT = os.path.join(_HERE, "libk1emu_default.so")


def build_k1(coarse=True):
    """coarse=True: the kernel with its opt-in paths compiled in (-DK1E_COARSE, the coarse table; -DK1E_UNIFORM2=1, the wave-uniform
    two-block path: a superset of the default code, so the suite exercises all look-up paths); coarse=False: the product's default configuration."""
    csrc = os.path.join(_ROOT, "rnaseqc_amd", "csrc")
    so = _K1SO if coarse else _K1SO_DEFAULT
    srcs = [os.path.join(_HERE, "k1_emu.cpp"), os.path.join(_HERE, "wavemu.h")] + \
           [os.path.join(csrc, f) for f in ("rsqc_read.h", "rsqc_index.h", "rsqc_k1.h", "rsqc_k1s.h", "rsqc_kr.h", "rsqc_k4.h", "rsqc_wave.h", "rsqc_device.h")] + \
           [os.path.join(_ROOT, "include", "rnaseqc_amd.h")]
    # (RSQC_EMU_DEFS="-DK1E_..." : the emulation of an A/B build of the kernel; remove the .so files before and after)
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-unused-function",
                               "-Wno-unused-variable"] + (["-DK1E_COARSE", "-DK1E_UNIFORM2=1"] if coarse else []) + os.environ.get("RSQC_EMU_DEFS", "").split() + [srcs[0], "-o", so])
    return so


def run_k1(params, ann, batch, grid=2, want_cov=False, slow_kernel=True, coarse=True, bed=None):
    """The per-record KERNELS (rsqc_k1.h) on the 64-lane fiber emulation of wavemu.h, `grid` workgroups of 256 lanes."""
    lib = C.CDLL(build_k1(coarse))
    a, b = ann.to_struct(), batch.to_struct()
    o = Out()
    G, E = ann.n_genes_listed, ann.n_exons
    o.counters = np.zeros(abi.N_COUNTERS, np.uint64)
    o.gene_reads = np.zeros(G, np.uint64); o.gene_unique = np.zeros(G, np.uint64); o.gene_fragments = np.zeros(G, np.uint64)
    o.exon_reads = np.zeros(E, np.float64)
    rl = C.c_int32(); stats = np.zeros(4, np.uint64)
    o.cov = None
    if want_cov:
        total = int(sum(int(ann.exon_row_end[i]) - int(ann.exon_row_start[i]) + 1 for i in range(E))) + ann.n_genes + 8
        o.cov = np.zeros(total, np.uint32)
    bs = bed.to_struct() if bed is not None else None           # (with a BED: the --bed instance classify_ei_kernel<true>, candidates checked)
    rc = lib.k1emu_run_bed(C.byref(params), C.byref(a), C.byref(b), C.byref(bs) if bs is not None else None, C.c_int(grid), C.c_int(1 if slow_kernel else 0),
                           abi.ptr(o.counters), abi.ptr(o.gene_reads), abi.ptr(o.gene_unique), abi.ptr(o.gene_fragments), abi.ptr(o.exon_reads), C.byref(rl),
                           abi.ptr(o.cov) if want_cov else None, abi.ptr(stats))
    if rc:
        raise RuntimeError("k1emu rc=%d" % rc)
    o.read_length = rl.value
    o.n_overflow = int(stats[0]); o.n_listed = int(stats[1]) & 0xFFFFFFFF; o.n_deferred = int(stats[1]) >> 32; o.n_pairs = int(stats[2]); o.n_coarse = int(stats[3]) if bed is None else 0; o.n_candidates = int(stats[3]) if bed is not None else 0
    lib.k1emu_uniform_calls.restype = C.c_ulonglong
    o.n_uniform = int(lib.k1emu_uniform_calls())
    lib.k1emu_ucache_hits.restype = C.c_ulonglong
    o.n_ucache_hits = int(lib.k1emu_ucache_hits())
    lib.k1emu_uniform2_calls.restype = C.c_ulonglong
    o.n_uniform2 = int(lib.k1emu_uniform2_calls())
    return o


_K4SO = os.path.join(_HERE, "libk4emu.so")


def build_k4():
    csrc = os.path.join(_ROOT, "rnaseqc_amd", "csrc")
    srcs = [os.path.join(_HERE, "k4_emu.cpp"), os.path.join(_HERE, "wavemu.h")] + \
           [os.path.join(csrc, f) for f in ("rsqc_k4.h", "rsqc_wave.h", "rsqc_device.h", "rsqc_read.h")]
    if not os.path.exists(_K4SO) or any(os.path.getmtime(_K4SO) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-unused-function",
                               "-Wno-unused-variable", srcs[0], "-o", _K4SO])
    return _K4SO


def run_k4(seed, n_genes, n_chunks, n_names, hot_reads, arena=False, wide=False):
    """The fragment-counting KERNELS (rsqc_k4.h) on the 64-lane fiber emulation against a std::set per gene.
    Returns (rc, stats): rc 0 = every gene's count equals its name set; stats = pairs, keys kept by frag_local, partitions,
    partitions left to the second counting instance, distinct (gene, name) pairs, chunk capacity."""
    lib = C.CDLL(build_k4())
    lib.k4emu_run.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    st = np.zeros(6, np.uint64)
    rc = lib.k4emu_run(seed, n_genes, n_chunks, n_names, hot_reads, (1 if arena else 0) | (2 if wide else 0), st.ctypes.data)
    return rc, dict(zip(("pairs", "kept", "partitions", "fuller", "distinct", "chunk_cap"), (int(x) for x in st)))


_K3SO = os.path.join(_HERE, "libk3emu.so")


def build_k3():
    csrc = os.path.join(_ROOT, "rnaseqc_amd", "csrc")
    srcs = [os.path.join(_HERE, "k3_emu.cpp"), os.path.join(_HERE, "wavemu.h")] + \
           [os.path.join(csrc, f) for f in ("rsqc_k3.h", "rsqc_wave.h", "rsqc_device.h", "rsqc_read.h", "rsqc_index.h")]
    if not os.path.exists(_K3SO) or any(os.path.getmtime(_K3SO) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-unused-function",
                               "-Wno-unused-variable", srcs[0], "-o", _K3SO])
    return _K3SO


def run_k3(params, ann, cov_diff, gene_reads, force=0):
    """The end-of-file coverage KERNEL (rsqc_k3.h) on the 64-lane fiber emulation: from the difference array and the gene counts
    of a pass (hostemu.run(..., want_cov=True)) to per-gene mean / std / CV, per-exon CV and the bias accumulators."""
    lib = C.CDLL(build_k3())
    a = ann.to_struct()
    o = Out()
    G, E = ann.n_genes_listed, ann.n_exons
    o.gene_cov_mean = np.zeros(G, np.float64); o.gene_cov_std = np.zeros(G, np.float64); o.gene_cov_cv = np.zeros(G, np.float64)
    o.gene_cov_valid = np.zeros(G, np.uint8); o.exon_cv = np.zeros(E, np.float64); o.exon_cv_valid = np.zeros(E, np.uint8)
    o.bias_three = np.zeros(G, np.uint64); o.bias_five = np.zeros(G, np.uint64)
    stats = np.zeros(4, np.uint64)
    cov = np.ascontiguousarray(cov_diff, np.uint32); gr = np.ascontiguousarray(gene_reads, np.uint64)
    rc = lib.k3emu_run(C.byref(params), C.byref(a), abi.ptr(cov), abi.ptr(gr), C.c_int(force), abi.ptr(o.gene_cov_mean), abi.ptr(o.gene_cov_std),
                       abi.ptr(o.gene_cov_cv), abi.ptr(o.gene_cov_valid), abi.ptr(o.exon_cv), abi.ptr(o.exon_cv_valid),
                       abi.ptr(o.bias_three), abi.ptr(o.bias_five), abi.ptr(stats))
    o.rc = rc
    o.classes = [int(x) for x in stats]
    return o


_K5SO = os.path.join(_HERE, "libk5emu.so")


def build_k5():
    csrc = os.path.join(_ROOT, "rnaseqc_amd", "csrc")
    srcs = [os.path.join(_HERE, "k5_emu.cpp"), os.path.join(_HERE, "wavemu.h")] + [os.path.join(csrc, f) for f in ("rsqc_k5.h", "rsqc_device.h", "rsqc_read.h")]
    if not os.path.exists(_K5SO) or any(os.path.getmtime(_K5SO) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-unused-function",
                               "-Wno-unused-variable", srcs[0], "-o", _K5SO])
    return _K5SO


def run_k5(seed, n_names, max_samples, hot=0):
    """The fragment-size KERNELS (rsqc_k5.h) on the 64-lane fiber emulation against a literal std::map walk in file order.
    hot: records of one extra name (a bucket beyond the LDS sort); -1: one or two candidates per name + crafted collisions of the set's mix.  Returns (rc, candidates, samples, kept, distinct sizes, listed buckets)."""
    lib = C.CDLL(build_k5())
    lib.k5emu_run.argtypes = [C.c_uint64, C.c_int, C.c_uint32, C.c_int, C.c_void_p]
    stats = np.zeros(7, np.uint64)
    rc = lib.k5emu_run(seed, n_names, max_samples, hot, stats.ctypes.data)
    run_k5.last_paths = (int(stats[5]), int(stats[6]))          # buckets paired through the LDS set / handed to the sort (pair_bucket_hashed)
    return (rc,) + tuple(int(x) for x in stats[:5])
