"""TEST HARNESS: the product's per-record core (rnaseqc_amd/csrc/rsqc_read.h, what the per-read kernel runs per lane) on
HOSTILE records, with hostemu.cpp built under the address / undefined-behaviour sanitizers.

With the device decode the records of a file reach the per-read kernel without a host parser in between; a BAM can hold
positions next to 2^31, operations of 2^28 - 1 bases, thousands of N operations, mates anywhere.  Whatever the fields are
(inside the format's ranges), the core must stay inside the annotation's tables and the coverage array: on the GPU an
out-of-bounds atomic is a dead device.  Run as a script under LD_PRELOAD=libasan (tests/test_core_semantics_host.py does):

    python -m tests.hostemu.core_fuzz <rounds> <seed>
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

from rnaseqc_amd import abi, synth
from rnaseqc_amd.model import Batch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhostemu_san.so")


def build():
    src = os.path.join(_HERE, "hostemu.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fvisibility=hidden", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-fno-sanitize=signed-integer-overflow,shift", src, "-o", _SO])
    return _SO


def hostile_batch(rng, n, contig_lengths):
    n_ref = len(contig_lengths)
    recs = []
    tid = 0
    for i in range(n):
        if rng.random() < 0.002 and tid + 1 < n_ref:
            tid += 1
        L = int(contig_lengths[tid])
        kind = rng.integers(0, 8)
        if kind == 0:
            pos = int(rng.integers(0, L))
        elif kind == 1:
            pos = int(rng.choice([0, 1, L - 1, L, L + 1, 2**31 - 1, 2**31 - 200, 2**30, -1]))
        elif kind == 2:
            pos = int(rng.integers(max(0, L - 400), L + 400))
        else:
            pos = int(rng.integers(0, L))
        pos = min(pos, 2**31 - 1)
        n_ops = int(rng.choice([0, 1, 2, 3, 5, 8, 40, 300, 3000], p=[.03, .25, .2, .2, .15, .1, .04, .02, .01]))
        cigar = []
        for _ in range(n_ops):
            op = int(rng.choice([abi.CIG_M, abi.CIG_I, abi.CIG_D, abi.CIG_N, abi.CIG_S, abi.CIG_H, abi.CIG_P, abi.CIG_EQ, abi.CIG_X]))
            r = rng.random()
            ln = int(rng.integers(0, 200)) if r < 0.7 else int(rng.integers(0, 100000)) if r < 0.9 else int(rng.choice([0, 2**28 - 1, 2**27, 2**24]))
            cigar.append((op, ln))
        flag = int(rng.integers(0, 4096))
        if rng.random() < 0.6:
            flag &= ~(abi.FSECONDARY | abi.FQCFAIL | abi.FSUPP | abi.FUNMAP)
            flag |= abi.FPAIRED | abi.FPROPER
        recs.append(dict(qname="q%d" % int(rng.integers(0, n)), tid=tid, pos=pos, mtid=int(rng.integers(-1, n_ref)),
                         mpos=int(rng.choice([0, pos, 2**31 - 1, int(rng.integers(0, L))])), isize=int(rng.integers(-2**31, 2**31 - 1)),
                         flag=flag, mapq=int(rng.integers(0, 256)), cigar=cigar, l_qseq=int(rng.choice([0, 76, 150, 65534, 65535, 2**20])),
                         nm=int(rng.choice([-1, 0, 0, 0, 3, 6, 7, 254, 255, 70000])) if rng.random() < 0.9 else None))
    return recs


def main(rounds, seed):
    lib = C.CDLL(build())
    rng = np.random.default_rng(seed)
    ran = 0
    for it in range(rounds):
        lengths = [int(rng.choice([3_000_000, 2**31 - 1])), int(rng.integers(100_000, 2_000_000)), 400]
        ann = synth.make_annotation(seed=int(rng.integers(1, 1 << 30)),
                                    contigs=[("chrA", min(lengths[0], 3_000_000), int(rng.integers(1, 300))), ("chrB", lengths[1], int(rng.integers(0, 1 + lengths[1] // 50_000))), ("chrC", 400, 0)])
        recs = hostile_batch(rng, 3000, lengths)
        batch = Batch.from_records(recs)
        for kw in (dict(), dict(legacy=1), dict(stranded=abi.STRAND_REVERSE, unpaired=1)):
            p = abi.default_params(**kw)
            a, b = ann.to_struct(), batch.to_struct()
            G, E = ann.n_genes_listed, ann.n_exons
            counters = np.zeros(abi.N_COUNTERS, np.uint64)
            g0 = np.zeros(G, np.uint64); g1 = np.zeros(G, np.uint64); g2 = np.zeros(G, np.uint64); ex = np.zeros(E, np.float64)
            rl, nov = C.c_int32(), C.c_uint64()
            rc = lib.hostemu_run(C.byref(p), C.byref(a), C.byref(b), abi.ptr(counters), abi.ptr(g0), abi.ptr(g1), abi.ptr(g2), abi.ptr(ex),
                                 C.byref(rl), None, C.byref(nov))
            assert rc in (0, abi.ERR_CAPACITY, abi.ERR_BAD_CIGAR), rc
            ran += 1
    print("core_fuzz: %d runs of 3000 hostile records" % ran)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
