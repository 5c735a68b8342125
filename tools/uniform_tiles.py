#!/usr/bin/env python3
"""How many one-block feature-stage calls of the per-record kernel have EVERY block inside one elementary interval (the
wave-uniform path, rsqc_k1.h: k1e_uniform1), and how far apart the two mates of a fragment sit in the file -- on one contig of
the bench workload, generated at the workload's own density (per-contig seeds: the contig is the one bench.py generates).
No GPU needed.   usage: tools/uniform_tiles.py [contig index, default 20] [pairs, default 50000000]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rnaseqc_amd import synth

c = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
ann = synth.make_annotation(seed=1, contigs=synth.human_contigs())
b, _ = synth.make_reads_sharded(ann, pairs, seed=2, contigs=[c], workers=1, with_unmapped=False)
gm, em = ann.gene_row_contig == c, ann.exon_row_contig == c
bp = np.unique(np.concatenate([ann.gene_row_start[gm], ann.gene_row_end[gm] + 1, ann.exon_row_start[em], ann.exon_row_end[em] + 1]).astype(np.int64))
nc, off, cig = b.n_cigar.astype(np.int64), b.cigar_off.astype(np.int64), b.cigar
op, ln = cig & 15, cig >> 4
isblk = (op == 0) | (op == 7) | (op == 8)
isref = isblk | (op == 2) | (op == 3)
rec_of = np.repeat(np.arange(b.n), nc)
nb = np.bincount(rec_of, weights=isblk, minlength=b.n).astype(int)
idx = np.flatnonzero(isblk)
fb = np.full(b.n, -1, np.int64); fb[rec_of[idx][::-1]] = idx[::-1]
csum = np.concatenate([[0], np.cumsum(np.where(isref, ln, 0))])
bs = b.pos.astype(np.int64) + 1 + (csum[fb] - csum[off]); be = bs + ln[fb]
r1 = np.flatnonzero((nb == 1) & (nc <= 4))
n = len(r1) // 64 * 64
js = np.searchsorted(bp, bs[r1][:n], side="right").reshape(-1, 64); je = np.searchsorted(bp, be[r1][:n], side="right").reshape(-1, 64)
print("contig %d: %d records, %d breakpoints; blocks per record: %s" % (c, b.n, len(bp), np.round(np.bincount(nb)[:5] / b.n, 3)))
print("one-block calls of 64 records: %d, all blocks in ONE interval: %.1f %%" % (n // 64, 100 * (js.min(1) == je.max(1)).mean()))
order = np.argsort(b.qhash, kind="stable"); hs = b.qhash[order]
same = hs[1:] == hs[:-1]
d = np.abs(order[1:][same] - order[:-1][same])
print("records between the mates of a fragment: median %d, 90 %% below %d, 99 %% below %d" % (np.median(d), np.percentile(d, 90), np.percentile(d, 99)))
# ... and of those, how many lie in the SAME interval as the uniform call before them (what a one-entry cache of the last interval
# answers without touching the index), in file order and in the order of a wave that takes every fourth piece of four tiles
uni = js.min(1) == je.max(1)
iv = js.min(1)
prev_same = np.zeros(len(uni), bool); prev_same[1:] = uni[1:] & uni[:-1] & (iv[1:] == iv[:-1])
print("uniform calls that repeat the interval of the call before them: %.1f %% of all one-block calls (%.1f %% of the uniform ones)" %
      (100 * prev_same.mean(), 100 * prev_same.sum() / max(uni.sum(), 1)))
last = {}
hits = 0
for k in range(len(uni)):
    if uni[k]:
        # the last uniform interval seen, whatever lay between (the cache is only rewritten by a uniform call that missed)
        if last.get(0) == iv[k]: hits += 1
        last[0] = iv[k]
print("... with non-uniform calls in between not clearing the cache: %.1f %% of all one-block calls" % (100.0 * hits / len(uni)))
