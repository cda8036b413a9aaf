"""-m gpu: rt_trace_closest_noalloc (the entry point that allocates nothing: device pointers in and out, scratch in the
library's static device array) against rt_trace_closest, bit for bit, through the C-ABI — two calls in a row, the second
with a smaller input (the array keeps the first call's contents), then a two-level scene and the error paths."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def _device_call(rta, sc, rays):
    import torch

    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 6).copy()).cuda()
    d_hits = torch.full((max(len(rays), 1), 4), 7.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    rc = sc.trace_closest_noalloc(d_rays.data_ptr(), len(rays), d_hits.data_ptr())
    assert rc == 0, rta.last_error()
    return d_hits.cpu().numpy()[: len(rays)].copy().view(rta.HIT).reshape(-1)


def _same(got, want):
    assert got.tobytes() == want.tobytes(), f"{np.count_nonzero(got['tri'] != want['tri'])} triangle ids differ"


def test_noalloc_matches_trace_closest_twice(rta, gpu):
    sc = scenes.cornellbox(rta)
    sc.build(max_per_leaf=2)
    sc.upload(gpu)
    big = scenes.random_rays(300_000, 41, lo=(-1.7, 0, -1.7), hi=(1.7, 3.4, 1.7))
    want_big = sc.trace_closest(big)
    _same(_device_call(rta, sc, big), want_big)
    assert (want_big["tri"] >= 0).mean() > 0.5 and (want_big["tri"] < 0).any()
    small = scenes.random_rays(1_001, 42, lo=(-1.7, 0, -1.7), hi=(1.7, 3.4, 1.7))  # fewer rays than shards x 64: most shards empty
    _same(_device_call(rta, sc, small), sc.trace_closest(small))
    _same(_device_call(rta, sc, big[:7]), want_big[:7])
    _same(_device_call(rta, sc, big), want_big)  # and up again


def test_noalloc_two_level_scene(rta, gpu):
    sc = scenes.legacy_two_level(rta)  # analytic primitives + mesh instances under a TLAS
    sc.build()
    sc.upload(gpu)
    rays = scenes.random_rays(50_000, 43, lo=(-2, 0, -2), hi=(2, 3, 2))
    _same(_device_call(rta, sc, rays), sc.trace_closest(rays))
    _same(_device_call(rta, sc, rays[:333]), sc.trace_closest(rays[:333]))


def test_noalloc_limits_and_errors(rta, gpu):
    import torch

    sc = scenes.quad_room(rta)
    sc.build()
    sc.upload(gpu)
    d = torch.zeros(64, 6, device="cuda")
    h = torch.zeros(64, 4, device="cuda")
    torch.cuda.synchronize()
    assert sc.trace_closest_noalloc(d.data_ptr(), rta.NOALLOC_MAX_RAYS + 1, h.data_ptr()) == 4  # RT_ERR_LIMIT, before anything is read
    assert "524288" in rta.last_error()
    assert sc.trace_closest_noalloc(d.data_ptr(), 0, h.data_ptr()) == 0
    assert sc.trace_closest_noalloc(0, 8, h.data_ptr()) == 1
    assert sc.trace_closest_noalloc(d.data_ptr(), 8, h.data_ptr() + 4) == 1  # misaligned hits
    host = np.zeros((8, 6), np.float32)
    assert sc.trace_closest_noalloc(host.ctypes.data, 8, h.data_ptr()) == 1  # a host pointer is refused, not dereferenced
    assert "device memory" in rta.last_error()
    # at the limit exactly
    n = rta.NOALLOC_MAX_RAYS
    rays = scenes.random_rays(n, 44, lo=(-1, 0, -1), hi=(1, 2, 1))
    _same(_device_call(rta, sc, rays), sc.trace_closest(rays))
