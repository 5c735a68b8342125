#!/usr/bin/env python
"""The CLI's BAM loop on ONE box: the tree's build against a variant's (gpurun_variants/<name>/{bin,lib}), same BAM, -vv output kept.
Usage: python tools/r6_cli_ab.py [--pairs N] [--variants r5base ...]"""
import argparse, json, os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rnaseqc_amd import bamio, synth
ap = argparse.ArgumentParser(); ap.add_argument("--pairs", type=int, default=25_000_000); ap.add_argument("--variants", nargs="*", default=["r5base"])
ap.add_argument("--seq-mode", type=int, default=0)
args = ap.parse_args()
contigs = synth.human_contigs()
ann = synth.make_annotation(seed=1, contigs=contigs)
batch = synth.make_reads_sharded(ann, args.pairs, seed=2, workers=16)[0]
d = tempfile.mkdtemp(prefix="rsqc_cliab_")
bam, gtf = os.path.join(d, "s.bam"), os.path.join(d, "s.gtf")
bamio.write_bam_fast(bam, [(c[0], c[1]) for c in contigs], batch, threads=32, seq_mode=args.seq_mode)
bamio.write_gtf(gtf, ann)
print("records", batch.n, "bam bytes", os.path.getsize(bam), flush=True)
exes = [("tree", os.path.join(ROOT, "rnaseqc_amd", "bin", "rnaseqc"))] + [(v, os.path.join(ROOT, "gpurun_variants", v, "bin", "rnaseqc")) for v in args.variants]
for rep in range(3):
    for name, exe in exes:
        out = os.path.join(d, "out_%s_%d" % (name, rep))
        p = subprocess.run([exe, gtf, bam, out, "-vv"], env=dict(os.environ, RSQC_DECODE="device", RSQC_DECODE_PROFILE="1"), capture_output=True, text=True)
        m = re.search(r"Average Reads/Sec: ([0-9.e+]+)", p.stdout); e = re.search(r"Time Elapsed: ([0-9.e+-]+)", p.stdout)
        dec = [l for l in p.stderr.splitlines() if "[decode]" in l or "CPU share" in l]
        print(name, rep, "rc", p.returncode, "loop_s", e and e.group(1), "reads/s", m and m.group(1), "|", " ".join(dec)[:400], flush=True)
        if rep == 2:
            print("   ", " | ".join(l for l in p.stdout.splitlines() if "Wall time" in l or "decode" in l.lower())[:600], flush=True)
