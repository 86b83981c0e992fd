"""Host facts for the CPU baseline / tests: how many cores this process may really use."""
import os


def usable_cpus(cap=16):
    """min(cgroup quota, affinity, cap).  The GPU boxes expose 256 logical CPUs but give a 1-GPU job a
    16-core share; running torch-CPU with 256 threads there is orders of magnitude slower than with 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))
