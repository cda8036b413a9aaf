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
    assert j["ms_per_step"] > 0 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert "cornellbox" in j["config"]["workload"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] in ("k_closest", "k_shadow")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 4
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["launches"] > 0 and r["avg_launch_ms"] > 0


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
           "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    j = _last_json(p.stdout)
    _check_common(j, 2, 2, 1)
    assert "64 spp per GPU (128-sample Hammersley set)" in j["config"]["workload"]  # two windows of one sample set
