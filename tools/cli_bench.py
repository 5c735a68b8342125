#!/usr/bin/env python
"""End-to-end timing of the `rnaseqc` CLI (row (f) of the hot-path scope): synthetic coordinate-sorted BAM + GTF on
local disk -> BGZF inflate + BAM parse (host threads) -> GPU hot path -> report files.  Prints one JSON line.
Usage: python tools/cli_bench.py [--pairs N] [--threads T] [--genome]"""
import argparse, ctypes as C, json, os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rnaseqc_amd import bamio, synth

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=5_000_000)
ap.add_argument("--threads", type=int, default=0, help="decode threads (0 = the CLI default)")
ap.add_argument("--genome", action="store_true")
ap.add_argument("--keep", default="")
args = ap.parse_args()

contigs = synth.human_contigs() if args.genome else [synth.HUMAN_CONTIGS[0]]
ann = synth.make_annotation(seed=1, contigs=contigs)
t = time.time(); batch = synth.make_reads(ann, args.pairs, seed=2); t_gen = time.time() - t
d = args.keep or tempfile.mkdtemp(prefix="rsqc_cli_")
os.makedirs(d, exist_ok=True)
bam, gtf, out = os.path.join(d, "s.bam"), os.path.join(d, "s.gtf"), os.path.join(d, "out")
t = time.time(); bamio.write_bam_fast(bam, [(c[0], c[1]) for c in contigs], batch, threads=32); t_bam = time.time() - t
bamio.write_gtf(gtf, ann)
env = dict(os.environ)
if args.threads:
    env["RSQC_HOST_THREADS"] = str(args.threads)
# decode only (host library, no GPU)
lib = C.CDLL(os.path.join(ROOT, "rnaseqc_amd", "lib", "librsqc_host.so"))
lib.host_bam_read_all_ex.restype = C.c_void_p
t = time.time()
h = lib.host_bam_read_all_ex(bam.encode(), b"ch", None, 0, args.threads or min(os.cpu_count(), 64), C.c_ulonglong(1 << 21))
t_dec = time.time() - t
lib.host_bam_free(C.c_void_p(h))
runs = []
for rep in range(2):                       # second run: page cache warm, GPU driver warm
    t = time.time()
    p = subprocess.run([os.path.join(ROOT, "rnaseqc_amd", "bin", "rnaseqc"), gtf, bam, out, "-vv"], env=env, capture_output=True, text=True)
    wall = time.time() - t
    m = re.search(r"Average Reads/Sec: ([0-9.e+]+)", p.stdout)
    e = re.search(r"Time Elapsed: ([0-9.e+-]+)", p.stdout)
    runs.append({"rc": p.returncode, "wall_s": round(wall, 3), "bam_loop_s": float(e.group(1)) if e else None,
                 "bam_loop_reads_per_s": float(m.group(1)) if m else None})
# the same file with zlib forced for the BGZF inflate (default: libdeflate when the shared library is present)
t = time.time()
p = subprocess.run([os.path.join(ROOT, "rnaseqc_amd", "bin", "rnaseqc"), gtf, bam, out, "-vv"], env=dict(env, RSQC_HOST_ZLIB="1"), capture_output=True, text=True)
m = re.search(r"Average Reads/Sec: ([0-9.e+]+)", p.stdout)
zlib_run = {"rc": p.returncode, "wall_s": round(time.time() - t, 3), "bam_loop_reads_per_s": float(m.group(1)) if m else None}
print(json.dumps({"records": int(batch.n), "bam_bytes": os.path.getsize(bam), "genes": int(ann.n_genes),
                  "decode_only_s": round(t_dec, 3), "decode_only_reads_per_s": batch.n / t_dec,
                  "threads": args.threads or "default", "cli_runs": runs, "cli_run_zlib": zlib_run, "gen_s": round(t_gen, 1), "bam_write_s": round(t_bam, 1)}))
