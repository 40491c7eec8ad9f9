"""Host CPU share helpers (BLAS thread pools sized to the cores this process may actually use)."""
from __future__ import annotations

import math
import os


def available_cpus() -> int:
    """min(CPU affinity, cgroup CPU quota, os.cpu_count())."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, math.ceil(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, math.ceil(q / per)))
        except Exception:
            continue
    return max(1, n)


def limit_blas_threads(n: int | None = None):
    """Cap the BLAS/OpenMP pools of this process; returns the threadpoolctl limiter (or None)."""
    n = n or available_cpus()
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=n)
    except Exception:
        return None
