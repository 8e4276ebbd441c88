#!/usr/bin/env python
"""tools/cpu_probe.py -- what CPU resources does this box really give us?  Prints the cgroup
quota / affinity and times the compiled reference (oracle/_ref) IndexFlatL2 at several OpenMP
thread counts so that bench.py's cpu_baseline can pick a sane thread count."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
          "/sys/fs/cgroup/cpuset.cpus.effective", "/proc/loadavg"):
    try:
        print(f, "=", open(f).read().strip())
    except OSError as e:
        print(f, "-", e)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|^CPU\\(s\\)' ; free -g | head -2")
from oracle.pyoracle import Ref
from faiss_amd.datasets import synthetic_dataset
_, xb, xq = synthetic_dataset(128, 1000, 100000, 2048, seed=1)
idx = Ref.index_factory(128, "Flat")
idx.add(xb)
for nthr in (8, 16, 32, 64, 128, 256):
    Ref.set_threads(nthr)
    t0 = time.time()
    idx.search(xq[:256], 100)
    t1 = time.time()
    idx.search(xq, 100)
    t2 = time.time()
    print("threads %3d: warm %.3fs, 2048 queries x 100k rows %.3fs -> %.0f QPS" % (nthr, t1 - t0, t2 - t1, 2048 / (t2 - t1)),
          flush=True)
    if t2 - t1 > 20:
        break
