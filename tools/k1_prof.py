#!/usr/bin/env python
"""Section timers of K1 (diagnostic build `make -C rnaseqc_amd/csrc prof`): where the waves' cycles go, stalls included.
Usage: RSQC_LIB=rnaseqc_amd/lib/librnaseqc_amd_prof.so python tools/k1_prof.py [--pairs N] [--chr1]"""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RSQC_LIB", os.path.join(ROOT, "rnaseqc_amd", "lib", "librnaseqc_amd_prof.so"))
import numpy as np
from rnaseqc_amd import abi, engine, synth
ap = argparse.ArgumentParser(); ap.add_argument("--pairs", type=int, default=10_000_000); ap.add_argument("--chr1", action="store_true")
args = ap.parse_args()
contigs = [synth.HUMAN_CONTIGS[0]] if args.chr1 else synth.human_contigs()
ann = synth.make_annotation(seed=1, contigs=contigs)
batch = synth.make_reads(ann, args.pairs, seed=2) if args.chr1 else synth.make_reads_sharded(ann, args.pairs, seed=2, workers=24)[0]
e = engine.Engine(abi.default_params()); e.set_annotation(ann); h = e.upload(batch)
lib = engine.load_library()
names = ["loop tail (prev. tile)", "segments + issue next loads", "unpack", "CIGAR walk", "gate cascade", "sums + Read-Length (+BED)",
         "sort by shape + queues", "landing wait", "(before a feature stage)", "one-block tiles", "two-block tiles", "long-CIGAR tiles",
         "tail flush", "wg epilogue", "", ""]
if os.environ.get("K1_STAGE_MARKS"):             # a -DK1E_STAGE_MARKS build: the marks sit inside the feature stage
    names = ["loop tail + phase A of tiles without a call", "queue entries read (+ name-hash gathers issued)", "rank words issued", "rank words landed, entries issued",
             "entries landed, blocks resolved", "gene sets, class flags, counters", "commit: LDS tables, coverage atomics, pairs", "one-block call on the wave-uniform path (whole)",
             "phase A (+ tail of the call before)", "one-block call: counters to LDS, exit", "two-block call: exit", "three-block call: park / unpark, exit", "tail flush", "wg epilogue", "", ""]
for rep in range(2):
    e.reset(); lib.rsqc_debug_k1_prof(None, 1); e.submit_resident(h); e.wait()
out = (C.c_ulonglong * 48)(); lib.rsqc_debug_k1_prof(out, 0)
cyc = np.array(out[:16], dtype=np.float64); cnt = np.array(out[16:32], dtype=np.float64)
tiles = (batch.n + 63) // 64
tm = e.timing()
print("records %d  tiles %d  K1 %.3f ms" % (batch.n, tiles, tm["classify_ms"] / max(tm["classify_launches"], 1)))
for k in range(15):
    if cnt[k]: print("  [%2d] %-30s %5.1f %%   %8.0f cycles/tile   (%.2f marks/tile)" % (k, names[k], 100 * cyc[k] / cyc.sum(), cyc[k] / tiles, cnt[k] / tiles))
print("  total %.0f cycles/tile/wave (s_memtime ticks)" % (cyc[:16].sum() / tiles))
e.close()
