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
