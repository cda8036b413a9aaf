#!/usr/bin/env python3
"""bench.py — headline benchmark of the path-tracing hot path (BASELINE.json configs[1]).

Workload: cornellbox.glb (13 170 triangles) at 1920x1080, 8 bounces, GGX + Beckmann material mapping, camera from the
file's Camera node chain, uniform 16x8 environment.  One "step" = one rt_render call tracing SPP_STEP samples per pixel
of a (steps x SPP_STEP x n_gpus)-sample Hammersley set; the default 32 steps x 32 spp = the configuration's 1024 spp (rt_render cuts a step into four 8-spp wavefront passes in flight).
Multi-GPU: sample-pass sharding — rank r renders its own window of sample indices (no data-path collective) and ONE
RCCL reduce of the fp32 radiance buffer to rank 0 closes the frame (inside the timed region).

Prints ONE JSON line on rank 0.  `value` = (closest-hit + shadow rays actually traced, all ranks) / wall seconds, rays
counted by device counters, radiance accumulated in HBM (device output, nothing crosses PCIe in the timed region)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WIDTH, HEIGHT, DEPTH, SPP_STEP = 1920, 1080, 8, 32  # (main() raises the raster for --workload spheres10m)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
VISITS_JSON = os.path.join(ROOT, "profiles", "visits_cornellbox_1080p_d8.json")
# rocprofv3 --pmc passes over `bench.py --steps 1` (tools/profile_pmc.sh + tools/summarize_pmc.py), FETCH_SIZE doubled as
# MI355X_MICROARCH.md prescribes for gfx950; roofline.traffic = HBM bytes per launch of the dominant kernel from it
PMC_JSON = os.path.join(ROOT, "profiles", "r01_pmc_summary_final.json")


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


def main():
    global WIDTH, HEIGHT
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-pass", type=int, default=0)
    ap.add_argument("--spp-step", type=int, default=SPP_STEP, help="samples per pixel traced by one step (one rt_render call)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-per-leaf", type=int, default=None, help="max_per_node of BVH::build (1 = what the reference's TriangleMesh passes); "
                    "default: 2, random1m 1 (instanced5m: TriangleMesh(mesh, max_per_node))")
    ap.add_argument("--workload", choices=["cornellbox", "random1m", "spheres10m", "instanced5m", "instanced5m_flat"], default="cornellbox",
                    help="cornellbox = BASELINE configs[1] (the headline); random1m = configs[2], 1M random triangles, primary + shadow "
                         "rays only; spheres10m = configs[3] geometry on ONE GPU: 9.6M-triangle sphere field, 3840x2160, 8 bounces; "
                         "instanced5m = configs[4] as a TWO-LEVEL scene: one 100k-triangle mesh placed 50 times by Instance transforms")
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
    # RT_BENCH_REHEARSE=1: every rank on GPU 0 with the gloo backend — exercises the N > 1 control path (windows, barriers,
    # max-over-ranks timing, the frame reduce through a host copy) on a one-GPU box; its numbers mean nothing
    rehearse = bool(os.environ.get("RT_BENCH_REHEARSE"))
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if args.max_per_leaf is None:  # measured per workload: random1m walks 3 % slower with two triangles per leaf, the others 5-8 % faster
        args.max_per_leaf = 1 if args.workload == "random1m" else 2
    # ---- scene: load, build the BVH on the host, flatten, upload once
    t0 = time.time()
    depth = DEPTH
    visits_json = VISITS_JSON
    if args.workload == "cornellbox":
        sc = scenes.cornellbox(rta)
        # BVH::build(bounds, max_per_node, BVH_SAH) (BVH.h:29): the reference's own builder.  Its legacy TriangleMesh passes
        # 1 (Object.cpp:21); Mesh::build of the target design is declared and never written (Mesh.h:24).  Two triangles per
        # leaf walk 5 % faster with identical hits (6.75 -> 7.07 Grays/s; 3: 6.65, 4: 6.42, 8: 5.25).
        sc.build(max_per_leaf=args.max_per_leaf)
        cam = scenes.cornell_camera(rta, sc, WIDTH, HEIGHT)
        label = "cornellbox.glb"
    elif args.workload == "random1m":  # BASELINE configs[2]: traversal-bound, incoherent; the tree (64 MB) and triangles (48 MB) exceed the L2
        sc = scenes.random_triangles(rta, 1_000_000)
        sc.build(max_per_leaf=args.max_per_leaf)  # the reference's SAH builder (multi-threaded, same output)
        cam = scenes.random_triangles_camera(rta, WIDTH, HEIGHT)
        depth = 1
        visits_json = os.path.join(ROOT, "profiles", "visits_random1m_1080p_d1.json")
        label = "1M random triangles (seed 1234), one rectangle light"
    elif args.workload == "instanced5m":  # BASELINE configs[4], kept two-level: TLAS + ONE mesh BVH + per-instance ray transforms
        sc = scenes.legacy_instanced(rta, max_per_leaf=args.max_per_leaf)  # TriangleMesh(mesh, max_per_node); the reference's constant is 1
        sc.build()
        cam = scenes.legacy_instanced_camera(rta, WIDTH, HEIGHT)
        visits_json = os.path.join(ROOT, "profiles", "visits_instanced5m_1080p_d8.json")
        label = "room of Rectangle instances + one 99 856-triangle mesh x 50 Instance transforms (two-level), rectangle light + uniform environment"
    elif args.workload == "instanced5m_flat":  # the same surfaces flattened into one 5 M-triangle world-space mesh (Scene::add_instance)
        sc = scenes.legacy_instanced_flat(rta)
        sc.build(max_per_leaf=args.max_per_leaf)
        cam = scenes.legacy_instanced_camera(rta, WIDTH, HEIGHT)
        visits_json = os.path.join(ROOT, "profiles", "visits_instanced5m_flat_1080p_d8.json")
        label = "room + one 99 856-triangle mesh x 50, FLATTENED into one 5 M-triangle mesh, rectangle light + uniform environment"
    else:  # BASELINE configs[3] geometry, one GPU's share: 614 MB of pair records + 922 MB of triangles and normals, beyond every cache
        WIDTH, HEIGHT = 3840, 2160
        sc = scenes.sphere_field(rta)
        sc.build(max_per_leaf=args.max_per_leaf)
        cam = scenes.random_triangles_camera(rta, WIDTH, HEIGHT)
        sc.set_env(np.full((8, 16, 3), 0.3, np.float32))
        visits_json = os.path.join(ROOT, "profiles", "visits_spheres10m_4k_d8.json")
        label = "10 000 tessellated spheres (9.6M triangles, seed 1234), rectangle light + uniform environment"
    label = f"{label}, BVH::build(bounds, {args.max_per_leaf}, BVH_SAH),"
    if args.max_per_leaf == 2:  # visit statistics are per tree (tools/measure_visits.py <workload>_leaf2)
        visits_json = visits_json[:-5] + "_leaf2.json"
    elif args.max_per_leaf != 1:
        visits_json = ""  # no visit statistics for this tree: no roofline
    t_build = time.time() - t0
    sc.upload(local_rank)

    K, W = args.steps, args.warmup
    spp_step = args.spp_step
    first, _, spp_total = sharding.sample_window(rank, world, K * spp_step)  # this rank's window of the Hammersley set
    frame = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.float32, device=f"cuda:{local_rank}")

    def step(i, timing, serial=False, target=None):
        _, st = sc.render(cam, WIDTH, HEIGHT, spp=spp_step, max_depth=depth, seed=1, spp_begin=first + (i % K) * spp_step,
                          spp_total=spp_total, spp_per_pass=args.spp_per_pass, device_out=(target if target is not None else frame).data_ptr(),
                          timing=timing, serial=serial)
        return st

    for i in range(W):
        step(i, False)
    frame.zero_()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = None
    for i in range(K):
        st = step(i, True)
        if stats is None:
            stats = dict(st)
        else:
            for k in st:
                stats[k] += st[k]
    if rehearse and world > 1:
        host = frame.cpu()
        sharding.reduce_frame(host, dst=0)
        frame.copy_(host)
    else:
        sharding.reduce_frame(frame, dst=0)  # the frame's single RCCL reduce over xGMI (no-op at world 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    totals = torch.tensor([elapsed, float(stats["rays_closest"]), float(stats["rays_shadow"]), float(stats["paths"])],
                          dtype=torch.float64, device="cpu" if rehearse else f"cuda:{local_rank}")
    if world > 1:
        tmax = totals[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
        totals[0] = tmax[0]
    elapsed, rays_c, rays_s, paths = [float(x) for x in totals.tolist()]

    if rank == 0:
        rays = rays_c + rays_s
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
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{label} {WIDTH}x{HEIGHT}, {depth} bounces, " + ("GGX+Beckmann materials, " if args.workload == "cornellbox" else "") + f"{spp_step} spp per step, "
                            f"{K * spp_step} spp per GPU ({spp_total}-sample Hammersley set), sample-pass sharding + one reduce",
                "triangles": int(len(sc.array(rta.ARR_TRIS))) if args.workload != "instanced5m" else int(len(sc.array(rta.ARR_FLAT_TRIS))),
                "instances": int(len(sc.array(rta.ARR_INSTANCES))),
                "rays_closest": rays_c,
                "rays_shadow": rays_s,
                "rays_per_sample": rays / paths,
                "bvh_build_s": t_build,
                "frame_mean": float(frame.mean().item()) / (K * spp_step * world),
            },
        }
        # ---- roofline of the dominant kernel (rank 0's own launches, HIP events on the library's stream)
        visits = json.load(open(visits_json)) if os.path.exists(visits_json) else None
        kern = "k_closest" if stats["ms_closest"] >= stats["ms_shadow"] else "k_shadow"  # (k_closest2 / k_shadow2 on a two-level scene)
        if visits and max(stats["ms_closest"], stats["ms_shadow"]) > 0:
            v = visits["closest" if kern == "k_closest" else "shadow"]
            n_rays = stats["rays_closest"] if kern == "k_closest" else stats["rays_shadow"]
            ms = stats["ms_closest"] if kern == "k_closest" else stats["ms_shadow"]
            launches = stats["launches_closest"] if kern == "k_closest" else stats["launches_shadow"]
            gbps = n_rays * v["bytes_per_ray"] / (ms * 1e-3) / 1e9
            traffic = None
            # HBM bytes per launch from the PMC passes of tools/profile_pmc.sh over this workload and tree (two-level kernels
            # are listed as k_closest2 / k_shadow2)
            pmc_json = PMC_JSON if args.workload == "cornellbox" else os.path.join(ROOT, "profiles", f"r01_pmc_summary_{args.workload}.json")
            if os.path.exists(pmc_json) and args.max_per_leaf == (1 if args.workload == "random1m" else 2):
                pmc = json.load(open(pmc_json))
                pmc = pmc.get(kern + "2", {}) if args.workload == "instanced5m" else pmc.get(kern, {})
                if pmc.get("dispatches") and "hbm_read_bytes_corrected" in pmc and "hbm_write_bytes" in pmc:
                    traffic = (pmc["hbm_read_bytes_corrected"] + pmc["hbm_write_bytes"]) / pmc["dispatches"]
            out["roofline"] = {
                "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                "traffic": traffic, "algorithmic_bytes_per_launch": n_rays * v["bytes_per_ray"] / max(launches, 1), "kernel": kern, "bytes_per_ray": v["bytes_per_ray"], "rays_per_launch": n_rays / max(launches, 1),
                "avg_launch_ms": ms / max(launches, 1), "launches": launches,
            }
            # the same kernel alone on the GPU: one untimed step with one sample pass at a time (RT_RENDER_SERIAL).  In the
            # timed region up to four passes share the GPU, so a launch's event-bracketed duration includes its neighbours' work.
            scratch = torch.zeros_like(frame)
            step(0, True, serial=True, target=scratch)
            alone = step(1 % K, True, serial=True, target=scratch)
            a_rays = alone["rays_closest"] if kern == "k_closest" else alone["rays_shadow"]
            a_ms = alone["ms_closest"] if kern == "k_closest" else alone["ms_shadow"]
            a_n = alone["launches_closest"] if kern == "k_closest" else alone["launches_shadow"]
            a_gbps = a_rays * v["bytes_per_ray"] / (a_ms * 1e-3) / 1e9
            out["roofline"]["standalone"] = {"achieved": a_gbps, "frac": a_gbps / HBM_PEAK_GBPS, "avg_launch_ms": a_ms / max(a_n, 1),
                                             "launches": a_n, "ms_per_step": alone["ms_total"],
                                             "note": "untimed extra step, one pass at a time" + ("; the tree is L2/LDS-resident on cornellbox, so "
                                                     "algorithmic bytes are not HBM bytes (see traffic)" if args.workload == "cornellbox" else "")}
        out["kernel_ms"] = {k: stats[k] for k in ("ms_total", "ms_generate", "ms_closest", "ms_shade", "ms_shadow", "ms_accumulate")}
        if not args.no_cpu_baseline and world == 1 and args.workload == "cornellbox":
            out["cpu_baseline"] = cpu_baseline(scenes, rta, sc)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
