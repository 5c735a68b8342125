cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/t61_tests.log
python - > gpurun_out/t61_ab.log 2>&1 <<'PY'
import os, subprocess, sys, re
R = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, R)
from rnaseqc_amd import bamio, synth
contigs = synth.human_contigs(); ann = synth.make_annotation(seed=1, contigs=contigs)
batch, _ = synth.make_reads_sharded(ann, 50_000_000, seed=2, workers=16)
bamio.write_gtf("/tmp/s.gtf", ann)
exe = os.path.join(R, "rnaseqc_amd", "bin", "rnaseqc")
for sm in (0, 1):
    bam = "/tmp/s%d.bam" % sm
    bamio.write_bam_fast(bam, [(c[0], c[1]) for c in contigs], batch, threads=16, seq_mode=sm)
    for rep in range(2):
        for thr in ("0", "12", "8"):
            env = dict(os.environ, RSQC_DECODE="device", RSQC_DECODE_PROFILE="1", RSQC_DECODE_CPU_THREADS=thr)
            p = subprocess.run([exe, "/tmp/s.gtf", bam, "/tmp/out_%s" % thr, "-vv"], env=env, capture_output=True, text=True)
            m = re.search(r"Average Reads/Sec: ([0-9.e+]+)", p.stdout)
            d = [l[9:] for l in p.stderr.split("\n") if ("calls" in l and not l.startswith("[decode] 0 calls")) or "host:" in l]
            same = open("/tmp/out_%s/s%d.bam.gene_reads.gct" % (thr, sm)).read() == open("/tmp/out_0/s%d.bam.gene_reads.gct" % sm).read() if p.returncode == 0 else None
            print("seq_mode %d cpu threads %2s: %.1f M reads/s same=%s | %s" % (sm, thr, float(m.group(1)) / 1e6 if m else -1, same, " | ".join(d) if d else p.stderr[-300:]), flush=True)
PY
