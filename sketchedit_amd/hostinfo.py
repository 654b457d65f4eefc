"""How many CPUs this process may really use: the affinity mask AND the cgroup CPU quota (a container that shows 256 CPUs
may be limited to 16 CPUs' worth of time -- the GPU boxes of this project are: /sys/fs/cgroup/cpu.max = "1600000 100000").
Sizing thread / worker pools by os.cpu_count() there only buys throttling."""
import math
import os


def cgroup_cpu_quota():
    """CPUs' worth of time the cgroup grants (float), or None when unlimited / unknown (cgroup v2, then v1)"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max" and float(p) > 0:
            return float(q) / float(p)
        return None
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return q / p if q > 0 and p > 0 else None
    except (OSError, ValueError):
        return None


def effective_cpus():
    """min(affinity mask, cgroup quota), at least 1"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    q = cgroup_cpu_quota()
    if q is not None:
        n = min(n, max(1, int(math.floor(q + 1e-9))))
    return max(1, n)


def cgroup_throttle_stats():
    """(nr_periods, nr_throttled, throttled_usec) of this process's cgroup (v2 cpu.stat), or None: how often the CPU quota
    stopped EVERY thread of the cgroup -- the launch thread included -- until the end of a scheduler period"""
    try:
        d = {}
        with open("/sys/fs/cgroup/cpu.stat") as f:
            for ln in f:
                k, _, v = ln.partition(" ")
                d[k] = int(v)
        return d.get("nr_periods", 0), d.get("nr_throttled", 0), d.get("throttled_usec", 0)
    except (OSError, ValueError):
        return None
