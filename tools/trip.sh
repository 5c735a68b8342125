cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - > gpurun_out/t47_ab.log 2>&1 <<'PY'
import os, subprocess, sys, re, shutil
R = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, R)
from rnaseqc_amd import bamio, synth
contigs = synth.human_contigs(); ann = synth.make_annotation(seed=1, contigs=contigs)
batch, _ = synth.make_reads_sharded(ann, 25_000_000, seed=2, workers=16)
bamio.write_gtf("/tmp/s.gtf", ann)
# one private copy of the CLI + library per variant: exactly one copy of the kernels in each process
variants = ["v1", "v7", "v8", "v9", "v10", "v1"]
for v in variants:
    d = "/tmp/var_%s" % v
    os.makedirs(d + "/bin", exist_ok=True); os.makedirs(d + "/lib", exist_ok=True)
    shutil.copy(os.path.join(R, "rnaseqc_amd", "bin", "rnaseqc"), d + "/bin/rnaseqc")
    shutil.copy(os.path.join(R, "gpurun_variants", v + ".so"), d + "/lib/librnaseqc_amd.so")
for sm in (0, 1):
    bam = "/tmp/s%d.bam" % sm
    bamio.write_bam_fast(bam, [(c[0], c[1]) for c in contigs], batch, threads=16, seq_mode=sm)
    for rep in range(1):
        for v in variants:
            env = dict(os.environ, RSQC_DECODE="device", RSQC_DECODE_PROFILE="1")
            p = subprocess.run(["/tmp/var_%s/bin/rnaseqc" % v, "/tmp/s.gtf", bam, "/tmp/out", "-vv"], env=env, capture_output=True, text=True)
            m = re.search(r"Average Reads/Sec: ([0-9.e+]+)", p.stdout)
            d = [l for l in p.stderr.split("\n") if "calls" in l]
            o = [l for l in p.stderr.split("\n") if "workgroups" in l]
            print("seq_mode %d %-8s: %.1f M reads/s  %s | %s" % (sm, v, float(m.group(1)) / 1e6 if m else -1, d[0][9:] if d else p.stderr[-300:], o[0][25:60] if o else ""), flush=True)
PY
