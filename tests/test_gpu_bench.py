"""bench.py's output contract, on the GPU: one JSON line with the fields the driver reads, the roofline object, and the
N > 1 control path (windows of the sample set, barriers, max-over-ranks timing, the frame reduce) rehearsed with two
ranks sharing the one GPU over gloo (RT_BENCH_REHEARSE; its numbers mean nothing)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _last_json(stdout):
    lines = [ln for ln in stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def _check_common(j, n_gpus, steps, warmup):
    assert j["metric"] and j["unit"] == "Mrays/s" and j["value"] > 0 and j["samples_per_s"] > 0
    assert (j["n_gpus"], j["steps"], j["warmup"]) == (n_gpus, steps, warmup)
    assert j["ms_per_step"] > 0 and j["higher_is_better"] is True and j["scaling"] == "strong"
    assert j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert "cornellbox" in j["config"]["workload"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] in ("k_closest", "k_shadow")
    # an HBM fraction is printed only from measured HBM bytes of the loaded kernels, and is a fraction
    if r["counters"]["matches_loaded_kernels"] and r["traffic"] is not None:
        assert r["traffic"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] <= 1
        assert abs(r["achieved"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    else:
        assert r["achieved"] is None and r["frac"] is None and r["traffic"] is None
    assert r["launches"] > 0 and r["avg_launch_ms"] > 0 and r["in_flight"]["avg_launch_ms"] >= 0.5 * r["avg_launch_ms"]
    a = r["algorithmic"]
    assert a["bytes_per_launch"] > 0 and a["gbps"] > 0
    for level in a.get("served_by", {}).values():
        assert 0 < level["frac"] <= 1  # each level's share against that level's own peak
    # the kernels of a stand-alone step, one after the other, fit the step
    k = j["kernel_ms_standalone_step"]
    assert k["ms_closest"] + k["ms_shade"] + k["ms_shadow"] + k["ms_generate"] + k["ms_accumulate"] <= 1.02 * k["ms_total"]


def test_bench_line(gpu):
    p = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-side"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    j = _last_json(p.stdout)
    _check_common(j, 1, 2, 1)
    # value = rays of the timed steps / their time
    rays = j["config"]["rays_closest"] + j["config"]["rays_shadow"]
    assert abs(j["value"] - rays / (j["ms_per_step"] * 2 * 1e-3) / 1e6) < 1e-3 * j["value"]


def test_bench_side_leg_on_the_hbm_bound_workload(gpu):
    """The default one-GPU run carries a short leg on the 9.6 M-triangle field (`side.spheres10m`), priced like the headline."""
    p = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=500)
    assert p.returncode == 0, p.stderr[-2000:]
    j = _last_json(p.stdout)
    assert "error" not in j["side"], j["side"]
    two = j["side"]["instanced5m"]  # configs[4] as the legacy two-level scene
    assert two["value"] > 0 and two["instances"] > 50 and two["roofline"]["kernel"] in ("k_closest2", "k_shadow2")
    side = j["side"]["spheres10m"]
    assert side["value"] > 0 and side["unit"] == "Mrays/s" and side["triangles"] > 9_000_000 and "3840x2160" in side["workload"]
    r = side["roofline"]
    assert r["bound"] == "hbm" and r["kernel"] in ("k_closest", "k_shadow") and r["algorithmic"]["bytes_per_ray"] > 1000
    if r["counters"]["matches_loaded_kernels"] and r["traffic"] is not None:
        assert 0 < r["frac"] <= 1
    assert j["per_rank"]["imbalance"] == 0.0 and len(j["per_rank"]["render_s"]) == 1


def test_bench_two_ranks_share_the_sample_set(gpu):
    env = dict(os.environ, RT_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-side"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    j = _last_json(p.stdout)
    _check_common(j, 2, 2, 1)
    assert "frame of 64 spp (64-sample Hammersley set), sample-pass sharding: 32 spp per GPU" in j["config"]["workload"]  # two windows of ONE frame
    one = _run_one_gpu(2)
    assert abs(j["config"]["frame_mean"] - one["config"]["frame_mean"]) < 1e-5 * one["config"]["frame_mean"]  # the same frame as one GPU renders
    assert j["config"]["rays_closest"] == one["config"]["rays_closest"] and j["config"]["rays_shadow"] == one["config"]["rays_shadow"]
    # the line shows every rank: its own rendering time, its rays, and how uneven the ranks were
    pr = j["per_rank"]
    assert len(pr["render_s"]) == len(pr["rays"]) == 2 and sum(pr["rays"]) == j["config"]["rays_closest"] + j["config"]["rays_shadow"]
    assert 0 < pr["render_s_min"] <= pr["render_s_max"] <= max(pr["elapsed_s"]) and pr["imbalance"] >= 0
    assert j["config"]["control_plane"].startswith("gloo") and j["config"]["spp_per_rank"] == [32, 32]


def test_bench_strong_scaling_cuts_a_rank_window_into_equal_calls(gpu):
    """5 steps of 32 spp over 2 ranks = 80 spp per rank: three calls of 27 + 27 + 26, not 32 + 32 + 16."""
    env = dict(os.environ, RT_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-side"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    j = _last_json(p.stdout)
    assert j["config"]["spp_per_rank"] == [80, 80] and j["config"]["calls_per_rank"] == [3, 3]
    assert j["config"]["rays_closest"] > 0 and abs(j["samples_per_s"] * j["ms_per_step"] * 5e-3 - 1920 * 1080 * 160) < 1.0


def _run_one_gpu(steps):
    p = subprocess.run([sys.executable, "bench.py", "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline", "--no-side"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return _last_json(p.stdout)


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("extra,expect", [(["--sharding", "tiles"], "image-tile sharding: interleaved 64x64 tiles"),
                                          (["--scaling", "weak"], "frame of 128 spp (128-sample Hammersley set)")])
def test_bench_two_ranks_tiles_and_weak(gpu, extra, expect):
    env = dict(os.environ, RT_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-side"] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    j = _last_json(p.stdout)
    assert expect in j["config"]["workload"] and j["n_gpus"] == 2
    if "tiles" in extra:  # disjoint tiles of the one-GPU frame: the same frame exactly, the same rays
        one = _run_one_gpu(2)
        assert j["scaling"] == "strong" and j["config"]["frame_mean"] == one["config"]["frame_mean"]
        assert j["config"]["rays_closest"] == one["config"]["rays_closest"]
    else:
        assert j["scaling"] == "weak"
