#!/usr/bin/env python
"""Host decode sweep for the end-to-end tier: one 100 M-record BAM, the CLI under several thread counts / group sizes.
Usage: python tools/decode_sweep.py [--pairs N] "T:G[:I]" ...   (T = RSQC_HOST_THREADS, G = group MiB, I = inflate threads)"""
import argparse, os, re, subprocess, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rnaseqc_amd import bamio, synth
ap = argparse.ArgumentParser(); ap.add_argument("--pairs", type=int, default=50_000_000); ap.add_argument("configs", nargs="*")
ap.add_argument("--seq-mode", type=int, default=0); ap.add_argument("--batch", type=int, default=0)
args = ap.parse_args()
contigs = synth.human_contigs()
ann = synth.make_annotation(seed=1, contigs=contigs)
batch, _ = synth.make_reads_sharded(ann, args.pairs, seed=2, workers=16)
d = tempfile.mkdtemp(prefix="rsqc_sweep_")
bam, gtf, out = os.path.join(d, "s.bam"), os.path.join(d, "s.gtf"), os.path.join(d, "out")
t = time.time(); bamio.write_bam_fast(bam, [(c[0], c[1]) for c in contigs], batch, threads=16, seq_mode=args.seq_mode); print("bam written %.1f s, %.1f B/rec" % (time.time() - t, os.path.getsize(bam) / batch.n), flush=True)
bamio.write_gtf(gtf, ann)
exe = os.path.join(ROOT, "rnaseqc_amd", "bin", "rnaseqc")
for cfg in (args.configs or ["16:64"]):
    parts = cfg.split(":")
    env = dict(os.environ, RSQC_HOST_THREADS=parts[0], RSQC_HOST_GROUP_BYTES=str(int(parts[1]) << 20))
    if len(parts) > 2: env["RSQC_HOST_INFLATE_THREADS"] = parts[2]
    if args.batch: env["RSQC_BATCH"] = str(args.batch)
    env["RSQC_HOST_PROFILE"] = "1"
    best = 0; prof = ""
    for rep in range(2):
        p = subprocess.run([exe, gtf, bam, out, "-vv"], env=env, capture_output=True, text=True)
        m = re.search(r"Average Reads/Sec: ([0-9.e+]+)", p.stdout)
        if m and float(m.group(1)) > best: best = float(m.group(1)); prof = [l for l in p.stderr.split("\n") if l.startswith("[bam]")]
    print("%-12s %.1f M reads/s   %s" % (cfg, best / 1e6, prof[0] if prof else ""), flush=True)
shutil.rmtree(d, ignore_errors=True)
