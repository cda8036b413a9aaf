"""The N>1 path on CPU: world_size-2 gloo.  Each rank renders its sample window (the oracle stands in for the GPU
renderer here — same window arithmetic as rt_render), one reduce assembles the frame, and the result equals the
one-process render up to fp32 summation order."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist
import raytracing_amd as rta, oracle_bindings as orc, scenes
from raytracing_amd import sharding
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
sc = scenes.quad_room(rta); sc.build(); osc = orc.scene_from_host(sc)
w, h, spp = 24, 20, 4
cam = rta.camera(rta.look_at((0.0, 1.0, 3.2), (0.0, 1.0, 0.0), (0, 1, 0)), 45.0, w, h)
mode = sys.argv[1]
frame = np.zeros((h, w, 3), np.float32)
if mode == "samples":
    b, n, tot = sharding.sample_window(rank, world, spp // world)
    orc.render(osc, cam, w, h, n, 4, seed=9, spp_begin=b, spp_total=tot, out=frame, threads=2)
else:
    for tile in sharding.tiles_for_rank(w, h, rank, world, extent=8):
        orc.render(osc, cam, w, h, spp, 4, seed=9, tile=tile, out=frame, threads=2)
t = torch.from_numpy(frame)
sharding.reduce_frame(t, dst=0)
if rank == 0:
    one, _ = orc.render(osc, cam, w, h, spp, 4, seed=9, threads=2)
    if mode == "samples":
        assert np.allclose(t.numpy(), one, rtol=1e-5, atol=1e-6), np.abs(t.numpy() - one).max()
    else:
        assert np.array_equal(t.numpy(), one)
    print("OK", mode)
dist.destroy_process_group()
"""


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(mode, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {ROOT!r}\n" + WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script), mode]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert f"OK {mode}" in out.stdout


def test_sample_pass_sharding_world2(tmp_path):
    _launch("samples", tmp_path)


def test_tile_sharding_world2(tmp_path):
    _launch("tiles", tmp_path)
