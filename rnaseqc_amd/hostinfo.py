"""CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU boxes expose 256 hardware
threads under a 16-CPU CFS quota; threads beyond the quota are throttled, not run)."""
import os


def effective_cpus() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    q = 0.0
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            q = float(a) / float(b)
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                q = quota / period
        except Exception:
            pass
    if q > 0:
        n = min(n, max(1, int(q + 0.5)))
    return max(1, n)


K1_SOURCES = ("rsqc_k1.h", "rsqc_read.h", "rsqc_wave.h", "rsqc_device.h", "rsqc_index.h", "rsqc_kernels.hip")


def k1_code_hash() -> str:
    """Digest of the per-record kernel's sources (the files classify_ei_kernel is compiled from).  bench.py stamps its
    roofline.traffic with it: the HBM-traffic figure is a cache of separate rocprofv3 --pmc passes (profiles/k1_traffic.json,
    written by tools/pmc.sh with the hash of the code it measured) and is dropped when the kernel has changed since."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for f in K1_SOURCES:
        with open(os.path.join(root, f), "rb") as fh:
            h.update(f.encode()); h.update(fh.read())
    return h.hexdigest()[:16]
