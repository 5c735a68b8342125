cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - > gpurun_out/t37_ring.log 2>&1 <<'PY'
import os, subprocess, sys, re
R = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, R)
from rnaseqc_amd import bamio, synth
contigs = synth.human_contigs(); ann = synth.make_annotation(seed=1, contigs=contigs)
batch, _ = synth.make_reads_sharded(ann, 15_000_000, seed=2, workers=16)
bamio.write_gtf("/tmp/s.gtf", ann)
exe = os.path.join(R, "rnaseqc_amd", "bin", "rnaseqc")
for sm in (1, 0):
    bam = "/tmp/s%d.bam" % sm
    bamio.write_bam_fast(bam, [(c[0], c[1]) for c in contigs], batch, threads=16, seq_mode=sm)
    for lib in ("", "_r13", "_r14", "_r15"):
        env = dict(os.environ, RSQC_DECODE="device", RSQC_DECODE_PROFILE="1", RSQC_INFLATE_WPW="1")
        if lib: env["LD_PRELOAD"] = os.path.join(R, "rnaseqc_amd", "lib", "librnaseqc_amd%s.so" % lib)
        p = subprocess.run([exe, "/tmp/s.gtf", bam, "/tmp/out", "-vv"], env=env, capture_output=True, text=True)
        m = re.search(r"Average Reads/Sec: ([0-9.e+]+)", p.stdout)
        d = [l for l in p.stderr.split("\n") if "calls" in l or "workgroups" in l]
        print("seq_mode %d ring %s: %.1f M reads/s\n   %s" % (sm, lib or "_r12", float(m.group(1)) / 1e6 if m else -1, "\n   ".join(d) if d else p.stderr[-300:]), flush=True)
PY
