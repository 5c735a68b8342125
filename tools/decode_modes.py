#!/usr/bin/env python
"""End-to-end tier, both decode modes: one BAM per SEQ/QUAL flavour, the CLI with RSQC_DECODE=host and =device.
Usage: python tools/decode_modes.py [--pairs N] [--seq-modes 0,1] [--prof DIR]"""
import argparse, os, re, subprocess, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rnaseqc_amd import bamio, synth, hostinfo
ap = argparse.ArgumentParser(); ap.add_argument("--pairs", type=int, default=25_000_000); ap.add_argument("--seq-modes", default="0,1")
ap.add_argument("--prof", default=""); ap.add_argument("--modes", default="host,device"); ap.add_argument("--reps", type=int, default=2)
args = ap.parse_args()
contigs = synth.human_contigs()
ann = synth.make_annotation(seed=1, contigs=contigs)
batch, _ = synth.make_reads_sharded(ann, args.pairs, seed=2, workers=min(16, hostinfo.effective_cpus()))
d = tempfile.mkdtemp(prefix="rsqc_modes_")
gtf, out = os.path.join(d, "s.gtf"), os.path.join(d, "out")
bamio.write_gtf(gtf, ann)
exe = os.path.join(ROOT, "rnaseqc_amd", "bin", "rnaseqc")
for sm in [int(x) for x in args.seq_modes.split(",")]:
    bam = os.path.join(d, "s%d.bam" % sm)
    t = time.time(); bamio.write_bam_fast(bam, [(c[0], c[1]) for c in contigs], batch, threads=16, seq_mode=sm)
    print("seq_mode %d: %d records, bam written in %.1f s, %.1f B/rec" % (sm, batch.n, time.time() - t, os.path.getsize(bam) / batch.n), flush=True)
    ref = None
    for mode in args.modes.split(","):
        best = 0
        for rep in range(args.reps):
            p = subprocess.run([exe, gtf, bam, out + mode, "-vv"], env=dict(os.environ, RSQC_DECODE=mode), capture_output=True, text=True)
            for l in p.stderr.split("\n"):
                if l.startswith("[decode]"): print("   ", l, flush=True)
            m = re.search(r"Average Reads/Sec: ([0-9.e+]+)", p.stdout)
            if p.returncode or not m:
                print(mode, "FAILED rc", p.returncode, p.stderr[-500:], flush=True); break
            best = max(best, float(m.group(1)))
        print("  %-7s %.1f M reads/s" % (mode, best / 1e6), flush=True)
        if best:
            g = open(os.path.join(out + mode, "s%d.bam.gene_reads.gct" % sm)).read()
            if ref is None: ref = g
            else: print("  outputs identical:", g == ref, flush=True)
    if args.prof and "device" in args.modes:
        os.makedirs(args.prof, exist_ok=True)
        subprocess.run("cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d %s/sm%d -o k -- %s %s %s %s -vv > /dev/null 2>&1" % (args.prof, sm, exe, gtf, bam, out + "prof"),
                       shell=True, env=dict(os.environ, RSQC_DECODE="device"))
shutil.rmtree(d, ignore_errors=True)
