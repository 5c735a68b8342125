cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_cli.py tests/test_gpu_decode.py::test_cli_device_decode_equals_host_decode -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/t63_tests.log
python - > gpurun_out/t63_wall.log 2>&1 <<'PY'
import os, subprocess, sys, re, time
R = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, R)
from rnaseqc_amd import bamio, synth
contigs = synth.human_contigs(); ann = synth.make_annotation(seed=1, contigs=contigs)
batch, _ = synth.make_reads_sharded(ann, 50_000_000, seed=2, workers=16)
bamio.write_gtf("/tmp/s.gtf", ann)
exe = os.path.join(R, "rnaseqc_amd", "bin", "rnaseqc")
bam = "/tmp/s0.bam"
bamio.write_bam_fast(bam, [(c[0], c[1]) for c in contigs], batch, threads=16, seq_mode=0)
for rep in range(3):
    for mode in ("device", "host"):
        t = time.time()
        p = subprocess.run([exe, "/tmp/s.gtf", bam, "/tmp/out", "-vv"], env=dict(os.environ, RSQC_DECODE=mode), capture_output=True, text=True)
        w = time.time() - t
        m = re.search(r"Average Reads/Sec: ([0-9.e+]+)", p.stdout)
        print("%s: %.1f M reads/s, wall %.2f s, rc %d" % (mode, float(m.group(1)) / 1e6 if m else -1, w, p.returncode), flush=True)
PY
