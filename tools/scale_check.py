#!/usr/bin/env python
"""Scale check of the whole boundary against the oracle: chr1-sized annotation + BED intervals + N pairs in three
host-fed batches (fragment-size sampler K5, multi-batch Read-Length, K4 partitions of hot genes at full size)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rnaseqc_amd import abi, engine, synth
from oracle import binding
from tests.compare import assert_results_match

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=5_000_000)
ap.add_argument("--genome", action="store_true")
args = ap.parse_args()
ann = synth.make_annotation(seed=1, contigs=synth.human_contigs() if args.genome else None)
bed = synth.make_bed(ann)
batch = synth.make_reads(ann, args.pairs, seed=2)
cuts = [0, batch.n // 3 + 5, 2 * batch.n // 3 + 11, batch.n]
parts = [batch.slice(cuts[k], cuts[k + 1]) for k in range(3)]
p = abi.default_params()
t = time.time(); got = engine.run_engine(p, ann, parts, bed=bed); t_gpu = time.time() - t
t = time.time(); want = binding.run_oracle(p, ann, parts, bed=bed); t_cpu = time.time() - t
assert_results_match(got, want)
print("OK: %d records, %d BED intervals, %d fragment sizes (%d samples left); GPU path %.2f s, oracle %.2f s" %
      (batch.n, len(bed.contig), len(got.fragment_size), got.fragment_samples_remaining, t_gpu, t_cpu))
