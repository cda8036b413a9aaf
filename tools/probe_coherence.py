#!/usr/bin/env python3
"""Probe: what would ordering the rays of a bounce buy the closest walk?  Traces the same incoherent secondary rays of
the cornellbox through rt_trace_closest three ways — as shading leaves them (pixel order, random directions), binned by
direction octant, and sorted by octant + origin cell — and prints nothing: run it under
  rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/probe_coherence.py
and read the three k_closest durations from the kernel trace (GPU box only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracing_amd as rta
import scenes

sc = scenes.cornellbox(rta)
sc.build()
sc.upload(0)
rng = np.random.default_rng(1)
# first hits of random rays inside the room, then uniformly random bounce directions from the hit points
n = 4_000_000
o = rng.uniform((-1.5, 0.1, -1.5), (1.5, 3.0, 1.5), (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros(n, rta.RAY)
rays["o"], rays["d"] = o, d
first = sc.trace_closest(rays)
ok = first["tri"] >= 0
p = o[ok] + d[ok] * first["t"][ok, None]
d2 = rng.normal(size=p.shape).astype(np.float32)
d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
sec = np.zeros(len(p), rta.RAY)
sec["o"], sec["d"] = p - d[ok] * 1e-3, d2
sec = sec[: 3_000_000]
octant = (sec["d"][:, 0] > 0).astype(np.int64) | ((sec["d"][:, 1] > 0).astype(np.int64) << 1) | ((sec["d"][:, 2] > 0).astype(np.int64) << 2)
cell = np.clip(((sec["o"] + 1.6) / 3.2 * 16).astype(np.int64), 0, 15)
key = (octant << 12) | (cell[:, 0] << 8) | (cell[:, 1] << 4) | cell[:, 2]
a = sc.trace_closest(sec)                                        # dispatch 2: as produced
b = sc.trace_closest(sec[np.argsort(octant, kind="stable")])     # dispatch 3: octant bins
c = sc.trace_closest(sec[np.argsort(key, kind="stable")])        # dispatch 4: octant + origin cell
print(len(sec), (a["tri"] >= 0).mean())
