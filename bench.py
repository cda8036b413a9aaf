#!/usr/bin/env python3
"""bench.py — headline benchmark of the path-tracing hot path (BASELINE.json configs[1]).

Workload: cornellbox.glb (13 170 triangles) at 1920x1080, 8 bounces, GGX + Beckmann material mapping, camera from the
file's Camera node chain, uniform 16x8 environment.  One "step" = SPP_STEP (32) samples per pixel over the whole frame =
one rt_render call (cut by the library into four 8-spp wavefront passes in flight); the default 32 steps are the
configuration's 1024 spp.

Multi-GPU (one process per GPU; torch.distributed over gloo for the launch, the barriers and the statistics, RCCL only for the
frame's reduce, through the library's own communicator):
  --scaling strong (default)  the frame is FIXED: steps x SPP_STEP samples per pixel of ONE Hammersley set.  With
                              --sharding samples each rank renders its contiguous window of the set, with --sharding tiles
                              its interleaved 64x64 tiles (rt_render_tiles) at all samples.  T_1 / (G T_G) is the north
                              star's scaling efficiency; the frame is the 1-GPU frame up to fp32 summation order.
  --scaling weak              every rank renders steps x SPP_STEP samples of a (world x that)-sample set.
Either way the ranks share nothing while rendering (no data-path collective) and ONE reduce of the fp32 radiance buffer
to rank 0 — rt_reduce of the C-ABI: RCCL over xGMI — closes the frame inside the timed region.

The default one-GPU run appends a short leg on the HBM-bound workload (`"side": {"spheres10m": ...}`, --no-side skips it) and the
CPU baseline.  Prints ONE JSON line on rank 0.  `value` = (closest-hit + shadow rays actually traced, all ranks) / wall seconds, rays
counted by device counters, radiance accumulated in HBM (device output, nothing crosses PCIe in the timed region)."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WIDTH, HEIGHT, DEPTH, SPP_STEP = 1920, 1080, 8, 32  # (main() raises the raster for --workload spheres10m)
# MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec; L2 (4 MiB per XCD, 32 MiB aggregate) ~34.5 TB/s; LDS ~150 TB/s for ds_read_b64/b128 with every CU streaming
HBM_PEAK_GBPS, L2_PEAK_GBPS, LDS_PEAK_GBPS = 8000.0, 34500.0, 150000.0
PROFILE_ROUND = "r03"


def kernel_source_sha16():
    """Identifies the kernels a counter summary was measured on: sha256 over the device sources."""
    h = hashlib.sha256()
    for name in ("rt_device.h", "rt_hip.hip"):
        with open(os.path.join(ROOT, "raytracing_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def cpu_baseline(scenes, rta, sc, seconds_target=15.0):
    """The oracle (CPU restatement, kind "port") timed on this host on a bounded sample of the same workload."""
    import oracle_bindings as orc

    osc = orc.scene_from_host(sc)
    cam = scenes.cornell_camera(rta, sc, WIDTH, HEIGHT)
    cores = min(16, os.cpu_count() or 1)
    tile = (720, 405, 1200, 675)  # central 480x270 window of the 1080p raster
    spp = 1
    total = {"rays_closest": 0, "rays_shadow": 0, "paths": 0, "seconds": 0.0}
    done = 0
    # double the sample count until ~seconds_target of CPU time has been spent
    while total["seconds"] < seconds_target and done < 1024:
        _, c = orc.render(osc, cam, WIDTH, HEIGHT, spp, DEPTH, seed=1, spp_begin=done, spp_total=1024, tile=tile, threads=cores)
        for k in total:
            total[k] += c[k]
        done += spp
        spp = min(spp * 2, 1024 - done) or 1
    rays = total["rays_closest"] + total["rays_shadow"]

    # two short side legs (SURVEY §8 d): one thread, and the reference's threading model — every thread drawing from ONE
    # locked generator, as glibc rand() makes the reference do (Utils.h:7-10) — on the same tile at a few samples
    def leg(threads, shared, seconds):
        n, t, r = 1, 0.0, 0
        while t < seconds and n <= 64:
            _, c = orc.render(osc, cam, WIDTH, HEIGHT, n, DEPTH, seed=2, spp_total=1024, tile=tile, threads=threads, shared_rng=shared)
            t += c["seconds"]
            r += c["rays_closest"] + c["rays_shadow"]
            n *= 2
        return r / t / 1e6

    return {
        "value": rays / total["seconds"] / 1e6,
        "unit": "Mrays/s",
        "cores": cores,
        "kind": "port",
        "samples_per_s": total["paths"] / total["seconds"],
        "sample": f"oracle/ CPU restatement, {cores} threads, thread-local counter RNG, 480x270 central tile of the 1080p "
                  f"raster, {done} spp, depth {DEPTH}: {total['paths']} paths, {rays} rays in {total['seconds']:.1f} s",
        "one_thread_mrays_per_s": leg(1, 0, 4.0),
        "shared_locked_rng_mrays_per_s": leg(cores, 1, 4.0),
    }


def roofline(args, stats, alone, visits, two_level):
    """The dominant kernel against its roofs.

    bound "hbm": `achieved` = HBM bytes the kernel moves per launch (rocprofv3 PMC passes over THIS kernel source,
    FETCH_SIZE corrected as MI355X_MICROARCH.md prescribes, + WRITE_SIZE) / its stand-alone launch duration, measured live in
    this run (HIP events on the launching stream, one pass at a time).  null when no counter summary matches the loaded
    kernels: an algorithmic figure is never printed as an HBM fraction.
    The algorithmic bytes (SURVEY 8 d: 64 B per pair record visited, 52 B per triangle tested, 48 B per ray; visit counts from
    the oracle's instrumented walk) are reported beside it, split by the level that serves them — the LDS copy of the tree
    top, and everything behind the vector L1 — each against that level's peak."""
    kern = "k_closest" if alone["ms_closest"] >= alone["ms_shadow"] else "k_shadow"
    which = "closest" if kern == "k_closest" else "shadow"
    v = visits[which]
    a_rays, a_ms, a_n = alone[f"rays_{which}"], alone[f"ms_{which}"], max(alone[f"launches_{which}"], 1)
    f_rays, f_ms, f_n = stats[f"rays_{which}"], stats[f"ms_{which}"], max(stats[f"launches_{which}"], 1)
    kernel_name = kern + ("2" if two_level else "")
    launch_s = a_ms * 1e-3 / a_n
    alg_bytes = a_rays * v["bytes_per_ray"] / a_n
    out = {"bound": "hbm", "kernel": kernel_name, "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None,
           "avg_launch_ms": a_ms / a_n, "launches": a_n, "rays_per_launch": a_rays / a_n,
           "timing": "stand-alone: one untimed extra step with one pass at a time (RT_RENDER_SERIAL), HIP events around each launch",
           "in_flight": {"avg_launch_ms": f_ms / f_n, "launches": f_n,
                         "note": "event brackets of the timed steps: up to four passes share the GPU, a bracket includes its neighbours' work"}}
    pmc_path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_summary_{args.workload}.json")
    counters = {"file": os.path.relpath(pmc_path, ROOT), "matches_loaded_kernels": False}
    if os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))
        meta = pmc.get("_meta", {})
        counters["measured_on_source_sha16"] = meta.get("source_sha16")
        counters["matches_loaded_kernels"] = meta.get("source_sha16") == kernel_source_sha16() and meta.get("max_per_leaf") == args.max_per_leaf
        k = pmc.get(kernel_name, {})
        if counters["matches_loaded_kernels"] and k.get("dispatches"):
            for name in ("valu_active_lanes_of_64", "l2_hit_rate", "mean_waves_per_busy_cycle", "valu_busy_frac", "lds_bank_conflict_cycles_per_lds_inst",
                         "SQ_WAIT_ANY_frac_of_wave_cycles"):
                if name in k:
                    counters[name] = k[name]
            if "hbm_read_bytes_corrected" in k and "hbm_write_bytes" in k:
                out["traffic"] = (k["hbm_read_bytes_corrected"] + k["hbm_write_bytes"]) / k["dispatches"]
                out["achieved"] = out["traffic"] / launch_s / 1e9
                out["frac"] = out["achieved"] / HBM_PEAK_GBPS
    out["counters"] = counters
    if "valu_busy_frac" in counters:
        # what the counters say binds the kernel when its HBM fraction is small: the share of SIMD cycles that issue vector
        # arithmetic, and how many of a wave's 64 lanes those instructions carry
        out["limiter"] = {"resource": "vector ALU issue" if counters["valu_busy_frac"] > 0.7 else "no single unit: LDS / vector-L1 transactions per visit against vector issue at ~37 of 64 lanes, 8 waves per SIMD (profiles/r02_walk_probes.txt, profiles/r03_quad_walk.txt)",
                          "valu_busy_frac": counters["valu_busy_frac"], "valu_active_lanes_of_64": counters.get("valu_active_lanes_of_64")}
    alg = {"bytes_per_launch": alg_bytes, "bytes_per_ray": v["bytes_per_ray"], "gbps": alg_bytes / launch_s / 1e9,
           "inner_per_ray": v["inner_per_ray"], "triangles_per_ray": v["leaf_per_ray"]}
    # share of the inner visits that read the LDS copy of the tree top (records [0, 256)): counted on the GPU by the -DRT_STATS
    # build of the walks (tools/lane_stats.py)
    lane_path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_lane_stats_{args.workload}.json")
    top = None
    if os.path.exists(lane_path):
        lane = json.load(open(lane_path))
        if lane.get("max_per_leaf") == args.max_per_leaf:
            top = lane.get(which, {}).get("inner_from_lds_top_frac")
            alg["lanes_per_inner_step"] = lane.get(which, {}).get("lanes_per_inner_step")
    if top is not None:
        lds_bytes = a_rays * 64.0 * v["inner_per_ray"] * top / a_n
        rest = alg_bytes - lds_bytes
        alg["served_by"] = {
            "lds": {"gbps": lds_bytes / launch_s / 1e9, "peak": LDS_PEAK_GBPS, "frac": lds_bytes / launch_s / 1e9 / LDS_PEAK_GBPS},
            "l1_l2_hbm": {"gbps": rest / launch_s / 1e9, "peak": L2_PEAK_GBPS, "frac": rest / launch_s / 1e9 / L2_PEAK_GBPS,
                          "note": "everything behind the vector L1, priced against the L2 peak; what of it reaches HBM is `traffic`"},
        }
    out["algorithmic"] = alg
    return out


def make_scene(rta, scenes, np, workload, max_per_leaf):
    """-> (scene, camera factory(width, height), label, depth, visits file stem, two_level, width, height)"""
    width, height, depth, two_level = 1920, 1080, DEPTH, False
    visits_name = "visits_cornellbox_1080p_d8"
    if workload == "cornellbox":
        sc = scenes.cornellbox(rta)
        cam_of = lambda w, h: scenes.cornell_camera(rta, sc, w, h)
        label = "cornellbox.glb"
    elif workload == "random1m":  # BASELINE configs[2]: traversal-bound, incoherent; the tree (64 MB) and triangles (48 MB) exceed the L2
        sc = scenes.random_triangles(rta, 1_000_000)
        cam_of = lambda w, h: scenes.random_triangles_camera(rta, w, h)
        depth = 1
        visits_name = "visits_random1m_1080p_d1"
        label = "1M random triangles (seed 1234), one rectangle light"
    elif workload == "instanced5m":  # BASELINE configs[4], kept two-level: TLAS + ONE mesh BVH + per-instance ray transforms
        sc = scenes.legacy_instanced(rta, max_per_leaf=max_per_leaf)  # TriangleMesh(mesh, max_per_node); the reference's constant is 1
        cam_of = lambda w, h: scenes.legacy_instanced_camera(rta, w, h)
        visits_name = "visits_instanced5m_1080p_d8"
        two_level = True
        label = "room of Rectangle instances + one 99 856-triangle mesh x 50 Instance transforms (two-level), rectangle light + uniform environment"
    elif workload == "instanced5m_flat":  # the same surfaces flattened into one 5 M-triangle world-space mesh (Scene::add_instance)
        sc = scenes.legacy_instanced_flat(rta)
        cam_of = lambda w, h: scenes.legacy_instanced_camera(rta, w, h)
        visits_name = "visits_instanced5m_flat_1080p_d8"
        label = "room + one 99 856-triangle mesh x 50, FLATTENED into one 5 M-triangle mesh, rectangle light + uniform environment"
    else:  # BASELINE configs[3] geometry, one GPU's share: 614 MB of pair records + 922 MB of triangles and normals, beyond every cache
        width, height = 3840, 2160
        sc = scenes.sphere_field(rta)
        sc.set_env(np.full((8, 16, 3), 0.3, np.float32))
        cam_of = lambda w, h: scenes.random_triangles_camera(rta, w, h)
        visits_name = "visits_spheres10m_4k_d8"
        label = "10 000 tessellated spheres (9.6M triangles, seed 1234), rectangle light + uniform environment"
    return sc, cam_of, label, depth, visits_name, two_level, width, height


def side_leg(rta, scenes, np, torch, workload, device, steps=4, spp_step=8, max_per_leaf=2):
    """A short leg on the ONE workload whose roof is HBM (BASELINE configs[3] geometry, 9.6 M triangles at 4K: the records do not
    fit any cache), so that a driver-run line carries a number for it: whole-step throughput with passes in flight, and the
    dominant kernel against the HBM peak exactly as roofline() prices the headline's.  BVH build time is reported, not timed in."""
    import types

    t0 = time.time()
    sc, cam_of, label, depth, visits_name, two_level, width, height = make_scene(rta, scenes, np, workload, max_per_leaf)
    if two_level:
        sc.build()  # (the mesh BVH was built by new_triangle_mesh)
    else:
        sc.build(max_per_leaf=max_per_leaf)
    t_build = time.time() - t0
    sc.upload(device)
    cam = cam_of(width, height)
    frame = torch.zeros((height, width, 3), dtype=torch.float32, device=f"cuda:{device}")
    torch.cuda.synchronize()

    def render(i, **kw):
        _, st = sc.render(cam, width, height, spp=spp_step, max_depth=depth, seed=1, spp_begin=i * spp_step, spp_total=(steps + 2) * spp_step,
                          device_out=frame.data_ptr(), **kw)
        return st

    for i in range(2):
        render(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = None
    for i in range(steps):
        st = render(2 + i, timing=True)
        stats = dict(st) if stats is None else {k: stats[k] + st[k] for k in st}
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rays = stats["rays_closest"] + stats["rays_shadow"]
    out = {"workload": f"{label}, BVH::build(bounds, {max_per_leaf}, BVH_SAH), {width}x{height}, {depth} bounces, {spp_step} spp per step", "value": rays / elapsed / 1e6,
           "unit": "Mrays/s", "samples_per_s": stats["paths"] / elapsed, "steps": steps, "warmup": 2, "ms_per_step": elapsed * 1000.0 / steps,
           "triangles": int(len(sc.array(rta.ARR_FLAT_TRIS if two_level else rta.ARR_TRIS))), "instances": int(len(sc.array(rta.ARR_INSTANCES))),
           "bvh_build_s": t_build, "rays_per_sample": rays / stats["paths"]}
    visits_json = os.path.join(ROOT, "profiles", visits_name + "_leaf2.json")
    if os.path.exists(visits_json):
        render(0, timing=True, serial=True)
        alone = render(1, timing=True, serial=True)
        out["roofline"] = roofline(types.SimpleNamespace(workload=workload, max_per_leaf=max_per_leaf), stats, alone, json.load(open(visits_json)), two_level)
        out["kernel_ms_standalone_step"] = {k: alone[k] for k in ("ms_total", "ms_generate", "ms_closest", "ms_shade", "ms_shadow", "ms_accumulate")}
    del sc, frame
    return out


def main():
    global WIDTH, HEIGHT
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-pass", type=int, default=0)
    ap.add_argument("--spp-step", type=int, default=SPP_STEP, help="samples per pixel of one step (one rt_render call over the whole frame)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the spheres10m side leg of the default one-GPU run")
    ap.add_argument("--max-per-leaf", type=int, default=None, help="max_per_node of BVH::build (1 = what the reference's TriangleMesh passes); "
                    "default: 2, random1m 1 (instanced5m: TriangleMesh(mesh, max_per_node))")
    ap.add_argument("--workload", choices=["cornellbox", "random1m", "spheres10m", "instanced5m", "instanced5m_flat"], default="cornellbox",
                    help="cornellbox = BASELINE configs[1] (the headline); random1m = configs[2], 1M random triangles, primary + shadow "
                         "rays only; spheres10m = configs[3] geometry on ONE GPU: 9.6M-triangle sphere field, 3840x2160, 8 bounces; "
                         "instanced5m = configs[4] as a TWO-LEVEL scene: one 100k-triangle mesh placed 50 times by Instance transforms")
    ap.add_argument("--sharding", choices=["samples", "tiles"], default=None,
                    help="multi-GPU partition: sample windows of the one sample set (default), or interleaved 64x64 image tiles "
                         "(default for instanced5m*: BASELINE configs[4])")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: the frame (steps x spp-step samples per pixel) is fixed and split over the GPUs; weak: every GPU renders that much")
    ap.add_argument("--tile", type=int, default=64, help="tile edge of --sharding tiles")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import raytracing_amd as rta
    import scenes
    from raytracing_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if rta.device_count() < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # RT_BENCH_REHEARSE=1: every rank on GPU 0 with the gloo backend — exercises the N > 1 control path (partitions, barriers,
    # max-over-ranks timing, the frame reduce through a host copy) on a one-GPU box; its numbers mean nothing
    rehearse = bool(os.environ.get("RT_BENCH_REHEARSE"))
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        # Launch, barriers and statistics go over gloo (CPU tensors); RCCL is used for exactly one thing, the frame's reduce, through
        # the library's own communicator (rt_comm_create).  Torch's NCCL process group is NOT created beside it: two RCCL
        # communicators over the same GPUs in one process is a combination no hardware run has vouched for, and the
        # control traffic is a few bytes.  RT_BENCH_TORCH_RCCL=1 brings torch's group back (the frame then goes through dist.reduce).
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: gloo must not depend on the host name resolving
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch_rccl = None
    if world > 1 and not rehearse and os.environ.get("RT_BENCH_TORCH_RCCL"):
        torch_rccl = dist.new_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    if args.max_per_leaf is None:  # measured per workload: random1m walks 3 % slower with two triangles per leaf, the others 5-8 % faster
        args.max_per_leaf = 1 if args.workload == "random1m" else 2
    if args.sharding is None:
        args.sharding = "tiles" if args.workload.startswith("instanced5m") else "samples"
    if args.sharding == "tiles" and args.scaling == "weak":
        raise SystemExit("--sharding tiles splits a fixed frame: use --scaling strong")

    # ---- scene: load, build the BVH on the host ONCE (rank 0; the other ranks take its nodes), flatten, upload
    t0 = time.time()
    sc, cam_of_wh, label, depth, visits_name, two_level, WIDTH, HEIGHT = make_scene(rta, scenes, np, args.workload, args.max_per_leaf)
    cam_of = lambda: cam_of_wh(WIDTH, HEIGHT)
    bvh_shared = False
    if two_level:
        sc.build()  # (the mesh BVH was built by new_triangle_mesh: 0.2 s)
    elif world > 1:
        # the SAH build is the expensive part of set-up (9 s at 9.6 M triangles on 16 threads): rank 0 builds, the others load its tree
        cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rt_bench_bvh_{args.workload}_{args.max_per_leaf}_{os.environ.get('MASTER_PORT', '0')}.npz")
        if rank == 0:
            sc.build(max_per_leaf=args.max_per_leaf)
            np.savez(cache, nodes=sc.array(rta.ARR_NODES), indices=sc.array(rta.ARR_INDICES))
        dist.barrier()
        if rank != 0:
            with np.load(cache) as z:
                sc.set_bvh(z["nodes"], z["indices"])
        dist.barrier()
        if rank == 0:
            os.remove(cache)
        bvh_shared = True
    else:
        # BVH::build(bounds, max_per_node, BVH_SAH) (BVH.h:29): the reference's own builder.  Its legacy TriangleMesh passes
        # 1 (Object.cpp:21); Mesh::build of the target design is declared and never written (Mesh.h:24).  Two triangles per
        # leaf walk 5 % faster with identical hits (3: 6.65, 4: 6.42, 8: 5.25 Grays/s against 7.07).
        sc.build(max_per_leaf=args.max_per_leaf)
    cam = cam_of()
    label = f"{label}, BVH::build(bounds, {args.max_per_leaf}, BVH_SAH),"
    visits_json = os.path.join(ROOT, "profiles", visits_name + ("_leaf2" if args.max_per_leaf == 2 else "") + ".json")
    if args.max_per_leaf not in (1, 2):
        visits_json = ""  # no visit statistics for this tree: no algorithmic figures
    t_build = time.time() - t0
    sc.upload(local_rank)

    # ---- this rank's share of the frame
    K, W = args.steps, args.warmup
    spp_step = args.spp_step
    tiles = None
    if args.scaling == "weak":
        first, count, spp_total = sharding.sample_window(rank, world, K * spp_step)
    elif args.sharding == "samples":
        first, count, spp_total = sharding.strong_sample_window(rank, world, K * spp_step)
    else:
        first, count, spp_total = 0, K * spp_step, K * spp_step
        tiles = rta.shard_tiles(rank, world, WIDTH, HEIGHT, args.tile) if world > 1 else None
    # this rank's rt_render calls of the timed region: as few as spp_step allows, of EQUAL size (80 spp are 27 + 27 + 26, not
    # 32 + 32 + 16: a short last call would run 4-spp passes, which the walks do measurably worse)
    n_calls = (count + spp_step - 1) // spp_step
    sizes = [count // n_calls + (1 if i < count % n_calls else 0) for i in range(n_calls)]
    calls = [(first + sum(sizes[:i]), sizes[i]) for i in range(n_calls)]
    frame = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.float32, device=f"cuda:{local_rank}")
    scratch = torch.zeros_like(frame)
    torch.cuda.synchronize()  # the library works on its own streams: the fills must have finished (include/rt_hip.h, RT_RENDER_DEVICE_OUT)

    def render(begin, n, timing, serial=False, target=None):
        _, st = sc.render(cam, WIDTH, HEIGHT, spp=n, max_depth=depth, seed=1, spp_begin=begin, spp_total=spp_total, spp_per_pass=args.spp_per_pass,
                          device_out=(target if target is not None else frame).data_ptr(), timing=timing, serial=serial, tiles=tiles)
        return st

    # the frame's reduce: rt_reduce (RCCL through the C-ABI); the one-GPU rehearsal goes through a host copy and gloo
    comm = None
    reduce_path = "none (one GPU)"
    if world > 1 and not rehearse and torch_rccl is None:
        uid = torch.from_numpy(rta.Comm.unique_id() if rank == 0 else np.zeros(128, np.uint8))
        dist.broadcast(uid, src=0)
        try:
            comm = rta.Comm.create(uid.numpy(), rank, world, local_rank)
            reduce_path = "rt_reduce (C-ABI, RCCL ncclReduce over xGMI)"
        except rta.RtError as e:  # loud, and the run still measures: the same sum through a host copy
            print(f"bench.py: rank {rank}: rt_comm_create failed ({e}); the frame reduce falls back to gloo through a host copy", file=sys.stderr, flush=True)
            reduce_path = "gloo through a host copy — rt_comm_create failed, see stderr"
        ok = torch.tensor([1 if comm is not None else 0])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and comm is not None:  # all ranks must take the same path
            comm.close()
            comm = None
            reduce_path = "gloo through a host copy — rt_comm_create failed on another rank"
    elif world > 1 and torch_rccl is not None:
        reduce_path = "torch.distributed.reduce over torch's own RCCL group (RT_BENCH_TORCH_RCCL)"
    elif world > 1:
        reduce_path = "gloo through a host copy (RT_BENCH_REHEARSE)"

    def reduce_frame(buf):
        if world == 1:
            return
        if comm is not None:
            comm.reduce(buf.data_ptr(), buf.numel(), 0)
        elif torch_rccl is not None:
            dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM, group=torch_rccl)
        else:
            host = buf.cpu()
            sharding.reduce_frame(host, dst=0)
            buf.copy_(host)

    for i in range(W):  # warm-up: W full-size calls + the collective's first use (channel set-up is not the frame's cost)
        render(first, calls[0][1], False, target=scratch)
    if W > 0 or world > 1:
        reduce_frame(scratch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = None
    for begin, n in calls:
        st = render(begin, n, True)
        stats = dict(st) if stats is None else {k: stats[k] + st[k] for k in st}
    torch.cuda.synchronize()
    t_rendered = time.perf_counter() - t0  # this rank's own rendering: what a load imbalance shows up in
    reduce_frame(frame)  # the frame's single reduce (no-op at world 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    mine = torch.tensor([elapsed, float(stats["rays_closest"]), float(stats["rays_shadow"]), float(stats["paths"]), t_rendered, float(count),
                         float(len(calls))], dtype=torch.float64)
    per_rank = [mine]
    if world > 1:
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
    table = torch.stack(per_rank)  # [rank, field]
    elapsed = float(table[:, 0].max())  # the job takes as long as its slowest rank
    rays_c, rays_s, paths = float(table[:, 1].sum()), float(table[:, 2].sum()), float(table[:, 3].sum())

    if rank == 0:
        rays = rays_c + rays_s
        total_spp = K * spp_step * (world if args.scaling == "weak" else 1)
        share = (f"sample-pass sharding: {count} spp per GPU" if args.sharding == "samples" else f"image-tile sharding: interleaved {args.tile}x{args.tile} tiles, "
                 f"{len(tiles) if tiles is not None else 'all'} per GPU") if world > 1 else "one GPU"
        out = {
            "metric": "Mrays/s and samples/s per GPU (1/2/4/8 MI355X), 1080p cornellbox, 8 bounces",
            "value": rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "samples_per_s": paths / elapsed,
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed * 1000.0 / K,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{label} {WIDTH}x{HEIGHT}, {depth} bounces, " + ("GGX+Beckmann materials, " if args.workload == "cornellbox" else "") +
                            f"{spp_step} spp per step, frame of {total_spp} spp ({spp_total}-sample Hammersley set), {share} + one reduce",
                "triangles": int(len(sc.array(rta.ARR_TRIS))) if args.workload != "instanced5m" else int(len(sc.array(rta.ARR_FLAT_TRIS))),
                "instances": int(len(sc.array(rta.ARR_INSTANCES))),
                "sharding": args.sharding if world > 1 else None,
                "reduce": reduce_path,
                "bvh_built_once_and_shared": bvh_shared,
                "control_plane": "gloo (launch, barriers, statistics)" if world > 1 else None,
                "calls_per_rank": [int(x) for x in table[:, 6].tolist()],
                "spp_per_rank": [int(x) for x in table[:, 5].tolist()],
                "rays_closest": rays_c,
                "rays_shadow": rays_s,
                "rays_per_sample": rays / paths,
                "bvh_build_s": t_build,
                "frame_mean": float(frame.mean().item()) / total_spp,
            },
        }
        # per-rank view of the timed region, so that an imbalance is visible in the line: a rank's own rendering time (before the
        # reduce), its wall time through the closing barrier, and the rays it traced
        render_s = table[:, 4].tolist()
        out["per_rank"] = {"render_s": render_s, "elapsed_s": table[:, 0].tolist(), "rays": (table[:, 1] + table[:, 2]).tolist(),
                           "render_s_min": min(render_s), "render_s_max": max(render_s),
                           "imbalance": max(render_s) / (sum(render_s) / len(render_s)) - 1.0,
                           "reduce_and_barrier_s_on_slowest": elapsed - max(render_s)}
        # ---- roofline of the dominant kernel: one untimed extra step with one pass at a time, so that a launch's duration is its own
        visits = json.load(open(visits_json)) if visits_json and os.path.exists(visits_json) else None
        if visits:
            render(first, min(spp_step, count), True, serial=True, target=scratch)
            alone = render(first, min(spp_step, count), True, serial=True, target=scratch)
            out["roofline"] = roofline(args, stats, alone, visits, two_level)
            out["kernel_ms_standalone_step"] = {k: alone[k] for k in ("ms_total", "ms_generate", "ms_closest", "ms_shade", "ms_shadow", "ms_accumulate")}
        out["kernel_ms"] = {k: stats[k] for k in ("ms_total", "ms_generate", "ms_closest", "ms_shade", "ms_shadow", "ms_accumulate")}
        if not args.no_cpu_baseline and world == 1 and args.workload == "cornellbox":
            out["cpu_baseline"] = cpu_baseline(scenes, rta, sc)
        if not args.no_side and world == 1 and args.workload == "cornellbox":
            try:
                # the workload whose roof IS HBM (configs[3] geometry), and configs[4] as the legacy two-level scene
                out["side"] = {"spheres10m": side_leg(rta, scenes, np, torch, "spheres10m", local_rank),
                               "instanced5m": side_leg(rta, scenes, np, torch, "instanced5m", local_rank, steps=2, spp_step=32)}
            except Exception as e:  # the headline line must not depend on the side legs
                out["side"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
