"""-m gpu: the multi-GPU pieces below Python, as far as ONE GPU can exercise them: rt_render_tiles (a rank's whole tile list
in one call), rt_comm_* / rt_reduce (RCCL with one rank; the shared-device rehearsal communicator with several), and the
C++ Renderer with Options::devices — every reduced frame compared with the ONE-RANK frame of the HIP path itself:
bit for bit for image tiles, to 1e-5 for sample windows (fp32 summation order)."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def room(rta, gpu):
    sc = scenes.quad_room(rta)
    sc.build()
    sc.upload(gpu)
    return sc


VIEW = ((0.0, 1.0, 3.2), (0.0, 1.0, 0.0), (0, 1, 0))


def test_render_tiles_is_render(rta, room):
    w, h, spp = 100, 70, 6  # not multiples of the tile edge: clipped tiles on the right and at the bottom
    cam = rta.camera(rta.look_at(*VIEW), 45.0, w, h)
    one, st1 = room.render(cam, w, h, spp=spp, max_depth=4, seed=5)
    for world in (1, 2, 3):
        frame = np.zeros((h, w, 3), np.float32)
        paths = 0
        for rank in range(world):
            tiles = rta.shard_tiles(rank, world, w, h, 16)
            part = np.zeros((h, w, 3), np.float32)
            _, st = room.render(cam, w, h, spp=spp, max_depth=4, seed=5, tiles=tiles, out=part)
            assert not (part != 0).any(axis=2)[_mask_outside(tiles, w, h)].any()  # nothing outside the rank's tiles
            frame += part  # disjoint tiles: adding zeros is exact
            paths += st["paths"]
        assert np.array_equal(frame, one) and paths == st1["paths"]
    # a list is one call's pixels whatever its order, and the same list again takes the cached-table path
    tiles = rta.shard_tiles(0, 1, w, h, 16)[::-1].copy()
    again, _ = room.render(cam, w, h, spp=spp, max_depth=4, seed=5, tiles=tiles)
    again2, _ = room.render(cam, w, h, spp=spp, max_depth=4, seed=5, tiles=tiles)
    assert np.array_equal(again, one) and np.array_equal(again2, one)
    with pytest.raises(rta.RtError, match="overlap"):
        room.render(cam, w, h, spp=1, max_depth=2, tiles=[(0, 0, 20, 20), (10, 10, 30, 30)])
    with pytest.raises(rta.RtError, match="bad tile"):
        room.render(cam, w, h, spp=1, max_depth=2, tiles=[(0, 0, 20, 20), (90, 60, 101, 70)])
    with pytest.raises(rta.RtError, match="empty"):
        room.render(cam, w, h, spp=1, max_depth=2, tiles=np.zeros((0, 4), np.int32))
    # the cached list is only good for the raster it was checked against: the same list on a smaller raster is refused
    # (it used to pass as "the list of the previous call" and write beyond the rows of the smaller frame)
    room.render(cam, w, h, spp=1, max_depth=2, tiles=tiles)
    small = rta.camera(rta.look_at(*VIEW), 45.0, w - 20, h - 10)
    with pytest.raises(rta.RtError, match="bad tile"):
        room.render(small, w - 20, h - 10, spp=1, max_depth=2, tiles=tiles)


def _mask_outside(tiles, w, h):
    m = np.ones((h, w), bool)
    for x0, y0, x1, y1 in tiles:
        m[y0:y1, x0:x1] = False
    return m


def test_rccl_communicator_of_one_rank(rta, gpu):
    """RCCL through the C-ABI (dlopen'd librccl): unique id, ncclCommInitRank / ncclCommInitAll with one rank, ncclReduce in
    place — a one-rank sum is the buffer itself.  (More ranks need more GPUs: RCCL refuses two ranks on one device.)"""
    import torch

    buf = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.5
    want = buf.cpu().numpy().copy()
    torch.cuda.synchronize()
    uid = rta.Comm.unique_id()
    assert uid.any()
    c = rta.Comm.create(uid, 0, 1, gpu)
    assert (c.rank, c.world) == (0, 1)
    c.reduce(buf.data_ptr(), buf.numel(), 0)
    assert np.array_equal(buf.cpu().numpy(), want)
    c.close()
    (c,) = rta.Comm.create_local([gpu])
    c.reduce(buf.data_ptr(), buf.numel(), 0)
    assert np.array_equal(buf.cpu().numpy(), want)
    with pytest.raises(rta.RtError):
        c.reduce(buf.data_ptr(), buf.numel(), 3)  # root outside the communicator
    c.close()
    with pytest.raises(rta.RtError):
        rta.Comm.create(uid, 2, 2, gpu)


def test_rehearsal_communicator_sums_in_rank_order(rta, gpu):
    import threading

    import torch

    comms = rta.Comm.create_local([gpu, gpu, gpu])
    bufs = [torch.full((4097,), float(k + 1), device="cuda") * torch.arange(4097, device="cuda") for k in range(3)]
    keep = [b.cpu().numpy().copy() for b in bufs]
    torch.cuda.synchronize()
    errs = []

    def run(k):
        try:
            comms[k].reduce(bufs[k].data_ptr(), bufs[k].numel(), 1)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    for _ in range(2):  # the communicator is reusable
        ts = [threading.Thread(target=run, args=(k,)) for k in range(3)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs
        keep[1] = (keep[1] + keep[0]) + keep[2]
        assert np.array_equal(bufs[1].cpu().numpy(), keep[1])
        assert np.array_equal(bufs[0].cpu().numpy(), keep[0]) and np.array_equal(bufs[2].cpu().numpy(), keep[2])
    [c.close() for c in comms]


@pytest.mark.parametrize("ranks", [2, 3])
def test_cpp_renderer_over_several_ranks(rta, gpu, ranks):
    """Renderer::render_image with Options::devices = [gpu] * ranks (host threads, scene replicas, rt_render /
    rt_render_tiles into device frames, rt_reduce, resolve on rank 0) against the one-device Renderer."""
    sc = scenes.quad_room(rta)
    sc.build()
    sc.upload(gpu)
    view = rta.look_at(*VIEW)
    w, h, spp = 96, 72, 8
    rgb1, rad1, st1 = sc.render_image(view, 45.0, w, h, spp, 4, seed=6)
    rgb_t, rad_t, st_t = sc.render_image_sharded(view, 45.0, w, h, spp, 4, [gpu] * ranks, sharding=1, tile_size=16, seed=6)
    assert np.array_equal(rad_t, rad1) and np.array_equal(rgb_t, rgb1)
    assert st_t["paths"] == st1["paths"] and st_t["rays_closest"] == st1["rays_closest"] and st_t["rays_shadow"] == st1["rays_shadow"]
    rgb_s, rad_s, st_s = sc.render_image_sharded(view, 45.0, w, h, spp, 4, [gpu] * ranks, sharding=0, seed=6)
    assert np.allclose(rad_s, rad1, rtol=1e-5, atol=1e-6)
    assert np.abs(rgb_s.astype(int) - rgb1.astype(int)).max() <= 1
    assert st_s["paths"] == st1["paths"] and st_s["rays_closest"] == st1["rays_closest"]
    # more ranks than samples: refused, not a hang
    with pytest.raises(rta.RtError):
        sc.render_image_sharded(view, 45.0, 32, 32, 1, 4, [gpu] * ranks, sharding=0)
    # more ranks than tiles: the spare ranks contribute zero frames
    rgb_b, rad_b, _ = sc.render_image_sharded(view, 45.0, 32, 32, 2, 4, [gpu] * ranks, sharding=1, tile_size=64, seed=6)
    _, rad_1, _ = sc.render_image(view, 45.0, 32, 32, 2, 4, seed=6)
    assert np.array_equal(rad_b, rad_1)
