#!/usr/bin/env python3
"""How much do the walks depend on the ORDER in which rays sit in their queue?  cornellbox: the rays of bounce 0, 1, 2 of a
1080p frame (diffuse bounces, made here with numpy from rt_trace_closest's hits) and the shadow segments towards the ceiling light,
each traced in (a) path order (the order the renderer's queues have: children where their parents were), (b) a random permutation,
(c) sorted by direction octant + 30-bit Morton code of the origin, (d) sorted by a coarse key (octant + 4x4x4 origin cell: what a
one-pass binning by the shading kernels could produce), (e) 256-ray blocks shuffled, (f) shuffled within windows of 4096.  Run under `rocprofv3 --kernel-trace --stats`: the script prints the order of
its launches, tools/coherence_report.py pairs them with the trace's durations.
usage: tools/coherence_probe.py [width height [cornellbox|spheres10m]]  (spheres10m: the 9.6 M-triangle field, shadow segments
towards the sky — cosine-distributed directions, no end)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracing_amd as rta  # noqa: E402
import scenes  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
rng = np.random.default_rng(11)

WORKLOAD = sys.argv[3] if len(sys.argv) > 3 else "cornellbox"
if WORKLOAD == "spheres10m":
    sc = scenes.sphere_field(rta)
    sc.build(max_per_leaf=2)
    sc.upload(0)
    info = {"camera_view": rta.look_at((0.5, 0.5, 2.6), (0.5, 0.5, 0.5), (0, 1, 0)), "camera_fov": 39.6}
else:
    sc = scenes.cornellbox(rta)
    sc.build(max_per_leaf=2)
    sc.upload(0)
    info = sc.info()
verts, tris = sc.array(rta.ARR_VERTS), sc.array(rta.ARR_TRIS)
p0, p1, p2 = (verts["pos"][tris["idx"][:, k]].astype(np.float64) for k in range(3))
gn = np.cross(p1 - p0, p2 - p0)
gn /= np.maximum(np.linalg.norm(gn, axis=1, keepdims=True), 1e-30)
lo, hi = verts["pos"].min(0).astype(np.float64), verts["pos"].max(0).astype(np.float64)


def primary():
    c = [rta.camera_ray(info["camera_view"], info["camera_fov"], W, H, x, y) for x, y in ((W / 2, H / 2), (0.5, 0.5), (W - 0.5, 0.5), (0.5, H - 0.5))]
    fwd = c[0]["d"].astype(np.float64)
    plane = [r["d"].astype(np.float64) / np.dot(r["d"].astype(np.float64), fwd) for r in c]
    x, y = np.meshgrid((np.arange(W) + 0.5 - 0.5) / (W - 1), (np.arange(H) + 0.5 - 0.5) / (H - 1))
    d = plane[1][None, None] + x[..., None] * (plane[2] - plane[1])[None, None] + y[..., None] * (plane[3] - plane[1])[None, None]
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    rays = np.zeros(W * H, rta.RAY)
    rays["o"] = c[0]["o"]
    rays["d"] = d.reshape(-1, 3).astype(np.float32)
    return rays


def bounce(rays, hits):
    """Cosine-distributed continuation of every ray that hit (Utils.h:70-83 in spirit; the distribution is what matters here)."""
    ok = hits["tri"] >= 0
    r, h = rays[ok], hits[ok]
    n = gn[h["tri"]]
    d_in = r["d"].astype(np.float64)
    n = np.where((np.sum(n * d_in, 1) > 0)[:, None], -n, n)
    p = r["o"].astype(np.float64) + d_in * h["t"].astype(np.float64)[:, None]
    u1, u2 = rng.random(len(r)), rng.random(len(r))
    rr, phi = np.sqrt(u1), 2 * np.pi * u2
    a = np.where(np.abs(n[:, :1]) > 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    t = np.cross(a, n)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(n, t)
    d = t * (rr * np.cos(phi))[:, None] + b * (rr * np.sin(phi))[:, None] + n * np.sqrt(np.maximum(0.0, 1 - u1))[:, None]
    out = np.zeros(len(r), rta.RAY)
    out["o"] = (p + n * 1e-3).astype(np.float32)
    out["d"] = d.astype(np.float32)
    return out


def spread(v):  # 10 bits -> every third bit
    v = v.astype(np.uint64) & 0x3FF
    v = (v | (v << 16)) & 0x30000FF
    v = (v | (v << 8)) & 0x300F00F
    v = (v | (v << 4)) & 0x30C30C3
    v = (v | (v << 2)) & 0x9249249
    return v


def key_fine(rays):
    c = np.clip(((rays["o"].astype(np.float64) - lo) / (hi - lo) * 1024).astype(np.int64), 0, 1023)
    m = spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2)
    octant = ((rays["d"][:, 0] < 0) * 1 + (rays["d"][:, 1] < 0) * 2 + (rays["d"][:, 2] < 0) * 4).astype(np.uint64)
    return (octant << 30) | m


def key_coarse(rays):
    c = np.clip(((rays["o"].astype(np.float64) - lo) / (hi - lo) * 4).astype(np.int64), 0, 3)
    octant = ((rays["d"][:, 0] < 0) * 1 + (rays["d"][:, 1] < 0) * 2 + (rays["d"][:, 2] < 0) * 4).astype(np.int64)
    return octant * 64 + c[:, 0] * 16 + c[:, 1] * 4 + c[:, 2]


def orders(rays):
    n = len(rays)
    # blocks: 256-ray blocks of the path order in random order (every wave still gets neighbours, the 64 contiguous shards of the
    # API's queue each get a sample of the whole frame); windows: path order shuffled within windows of 4096 (a wave's rays are no
    # longer neighbours, every shard keeps its part of the frame)
    nb = (n + 255) // 256
    blocks = (rng.permutation(nb)[:, None] * 256 + np.arange(256)[None, :]).reshape(-1)
    blocks = blocks[blocks < n]
    windows = np.arange(n)
    for a in range(0, n, 4096):
        windows[a:a + 4096] = a + rng.permutation(min(4096, n - a))
    return {"path": np.arange(n), "random": rng.permutation(n), "sorted": np.argsort(key_fine(rays), kind="stable"),
            "binned": np.argsort(key_coarse(rays), kind="stable"), "blocks": blocks, "windows": windows}


launches = []
rays = primary()
for b in range(3):
    ref = None
    for name, perm in orders(rays).items():
        hits = sc.trace_closest(rays[perm])
        launches.append((f"closest bounce{b} {name}", len(rays)))
        back = np.empty_like(hits)
        back[perm] = hits
        if ref is None:
            ref = back
        assert back.tobytes() == ref.tobytes(), "the order of the queue changed a result"
    # shadow segments from this bounce's hit points to points on a 1 x 1 patch under the ceiling (the cornell light's place)
    nxt = bounce(rays, ref)
    target = np.stack([rng.uniform(-0.5, 0.5, len(nxt)) * (hi[0] - lo[0]) * 0.3 + (lo[0] + hi[0]) / 2,
                       np.full(len(nxt), lo[1] + (hi[1] - lo[1]) * 0.98),
                       rng.uniform(-0.5, 0.5, len(nxt)) * (hi[2] - lo[2]) * 0.3 + (lo[2] + hi[2]) / 2], 1)
    seg = np.zeros(len(nxt), rta.RAY)
    seg["o"] = nxt["o"]
    if WORKLOAD == "spheres10m":
        seg["d"] = bounce(rays, ref)["d"]  # another cosine-distributed direction from the same points
        tmax = np.full(len(nxt), np.inf, np.float32)
    else:
        dv = target - nxt["o"].astype(np.float64)
        dist = np.linalg.norm(dv, axis=1)
        seg["d"] = (dv / dist[:, None]).astype(np.float32)
        tmax = (dist - 1e-3).astype(np.float32)
    ref_any = None
    for name, perm in orders(seg).items():
        occ = sc.trace_any(seg[perm], tmax[perm])
        launches.append((f"shadow bounce{b} {name}", len(seg)))
        back = np.empty_like(occ)
        back[perm] = occ
        if ref_any is None:
            ref_any = back
        assert np.array_equal(back, ref_any)
    print(f"bounce {b}: {len(rays)} rays, {np.count_nonzero(ref['tri'] >= 0)} hit, {len(seg)} shadow segments, {ref_any.mean():.3f} occluded", flush=True)
    rays = nxt
for name, n in launches:
    print(f"LAUNCH {name} {n}")
