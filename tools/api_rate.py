#!/usr/bin/env python3
"""End-to-end rate of the host-buffer entry points (rt_trace_closest / rt_trace_any: host arrays in, host arrays out, PCIe included).
usage: tools/api_rate.py [n_rays]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracing_amd as rta  # noqa: E402
import scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
sc = scenes.cornellbox(rta)
sc.build(max_per_leaf=2)
sc.upload(0)
rays = scenes.random_rays(n, 5, lo=(-1.7, 0, -1.7), hi=(1.7, 3.4, 1.7))
tmax = (np.random.default_rng(6).random(n) * 3).astype(np.float32)
sc.trace_closest(rays[:1000])
for name, call in (("rt_trace_closest", lambda: sc.trace_closest(rays)), ("rt_trace_any", lambda: sc.trace_any(rays, tmax))):
    call()
    t0 = time.perf_counter()
    for _ in range(3):
        out = call()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {n} rays in {dt * 1e3:.1f} ms = {n / dt / 1e6:.1f} Mrays/s end to end (checksum {int(np.asarray(out).view(np.uint8).sum() % 100000)})", flush=True)
for m in (1, 1000):
    sc.trace_closest(rays[:m])
    t0 = time.perf_counter()
    for _ in range(200):
        sc.trace_closest(rays[:m])
    print(f"rt_trace_closest, {m} ray(s) per call: {(time.perf_counter() - t0) / 200 * 1e6:.0f} us per call", flush=True)
