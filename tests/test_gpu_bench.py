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
    p = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    j = _last_json(p.stdout)
    _check_common(j, 1, 2, 1)
    # value = rays of the timed steps / their time
    rays = j["config"]["rays_closest"] + j["config"]["rays_shadow"]
    assert abs(j["value"] - rays / (j["ms_per_step"] * 2 * 1e-3) / 1e6) < 1e-3 * j["value"]


def test_bench_two_ranks_share_the_sample_set(gpu):
    env = dict(os.environ, RT_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    j = _last_json(p.stdout)
    _check_common(j, 2, 2, 1)
    assert "frame of 64 spp (64-sample Hammersley set), sample-pass sharding: 32 spp per GPU" in j["config"]["workload"]  # two windows of ONE frame
    one = _run_one_gpu(2)
    assert abs(j["config"]["frame_mean"] - one["config"]["frame_mean"]) < 1e-5 * one["config"]["frame_mean"]  # the same frame as one GPU renders
    assert j["config"]["rays_closest"] == one["config"]["rays_closest"] and j["config"]["rays_shadow"] == one["config"]["rays_shadow"]


def _run_one_gpu(steps):
    p = subprocess.run([sys.executable, "bench.py", "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT,
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
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra
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
