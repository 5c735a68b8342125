#!/usr/bin/env python
"""Section timers of the end-of-file stage (diagnostic build `make -C rnaseqc_amd/csrc prof`): the longest gene of K3 step by
step, frag_local_kernel and frag_count_kernel per section.  Usage: python tools/fin_prof.py [--pairs N]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RSQC_LIB", os.path.join(ROOT, "rnaseqc_amd", "lib", "librnaseqc_amd_prof.so"))
import numpy as np
from rnaseqc_amd import abi, engine, synth
ap = argparse.ArgumentParser(); ap.add_argument("--pairs", type=int, default=50_000_000)
args = ap.parse_args()
ann = synth.make_annotation(seed=1, contigs=synth.human_contigs())
batch = synth.make_reads_sharded(ann, args.pairs, seed=2, workers=24)[0]
e = engine.Engine(abi.default_params()); e.set_annotation(ann); h = e.upload(batch)
lib = engine.load_library()
for rep in range(2):
    e.reset(); lib.rsqc_debug_fin_prof(None, 1); e.submit_resident(h); e.wait(); e.finalize(lazy=True)
out = (C.c_ulonglong * 64)(); lib.rsqc_debug_fin_prof(out, 0)
o = np.array(out[:], dtype=np.float64)
tm = e.timing(); print("records", batch.n, {k: round(v, 3) for k, v in tm.items() if isinstance(v, float)})
print("K3, slowest workgroup of the longest-gene launch (gene %d of the order): coding %d bases, %d exons, vector in LDS: %d" % (o[27], o[28], o[29], o[30]))
names = ["scan (16-bit attempt)", "scan (in memory)", "per-exon CV", "argmax + gate", "radix select", "trim", "window medians", "gene mean/std"]
prev = o[0]
for k in range(1, 9):
    if o[k]:
        print("  %-24s %9.0f ticks" % (names[k - 1] if k != 2 or o[1] else "scan", o[k] - prev)); prev = o[k]
print("  total %.0f ticks (shader clock; about %.3f ms at 2.0 GHz)" % (prev - o[0], (prev - o[0]) / 2.0e6))
for base, nm, secs in ((32, "frag_local_kernel", ["loop top", "clear + sync + loads", "LDS de-dup", "partition rank", "sync", "cursor reservation + sync", "scatter"]),
                       (48, "frag_count_kernel", ["loop top (skips)", "clear + sync", "load + CAS", "sum + sync"])):
    tot = o[base:base + 14].sum()
    print("%s: %d pieces; ticks per piece (thread 0), share" % (nm, o[base + 15]))
    for k, s in enumerate(secs):
        print("  %-28s %8.1f   %5.1f %%" % (s, o[base + k] / max(o[base + 15], 1), 100 * o[base + k] / max(tot, 1)))
    if base == 48: print("  keys per partition %.1f" % (o[62] / max(o[63], 1)))
if hasattr(lib, "rsqc_debug_pairs"):
    hh = (C.c_ulonglong * 32768)(); gg = (C.c_uint32 * 32768)(); cc = C.c_uint32()
    if lib.rsqc_debug_pairs(hh, gg, C.byref(cc)) == 0 and cc.value:
        n = min(cc.value, 32768); h = np.array(hh[:n], dtype=np.uint64); g = np.array(gg[:n], dtype=np.uint64)
        k = h ^ (g << np.uint64(32))
        order = np.argsort(k, kind="stable"); ks = k[order]; same = ks[1:] == ks[:-1]
        d = np.abs(order[1:][same].astype(np.int64) - order[:-1][same].astype(np.int64))
        print("chunk 1000: %d pairs, %d distinct, %d repeated inside the chunk; distance between the two in chunk order: p10 %d p25 %d p50 %d p75 %d p90 %d p99 %d"
              % (n, len(np.unique(k)), same.sum(), *[np.percentile(d, q) for q in (10, 25, 50, 75, 90, 99)]))
        np.save(os.path.join(ROOT, "gpurun_out", "dbg_chunk_hash.npy"), h); np.save(os.path.join(ROOT, "gpurun_out", "dbg_chunk_gene.npy"), g)
e.close()
