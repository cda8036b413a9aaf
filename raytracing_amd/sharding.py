"""Multi-GPU sharding of the render call (one process per GPU, torch.distributed over RCCL/xGMI).

Every (pixel, sample) of the reference's hot loop (main.cpp:972-988) is independent and the scene is read-only, so
the scene is replicated and the loop is split with NO data-path collective:
  * sample-pass sharding: rank r renders sample indices [r*n, (r+1)*n) of ONE (world*n)-sample Hammersley set
    (hammersley_2d takes (i, N): Utils.h:148-151), then a single reduce(SUM) of the fp32 radiance buffer to rank 0;
  * image-tile sharding: interleaved tiles (the reference's TileQueue order, main.cpp:904-935, dealt round-robin);
    tiles are disjoint, so the same reduce(SUM) of zero-initialised buffers assembles the frame.
"""


def sample_window(rank, world, spp_per_rank):
    """-> (spp_begin, spp_count, spp_total) for rt_render on this rank."""
    if not (0 <= rank < world) or spp_per_rank <= 0:
        raise ValueError("bad rank/world/spp")
    return rank * spp_per_rank, spp_per_rank, world * spp_per_rank


def tiles_for_rank(width, height, rank, world, extent=64):
    """Row-major extent x extent tiles (TileQueue::next_tile order), tile t owned by rank t % world.
    -> list of (x0, y0, x1, y1)."""
    if not (0 <= rank < world) or extent <= 0:
        raise ValueError("bad rank/world/extent")
    tw = (width + extent - 1) // extent
    th = (height + extent - 1) // extent
    out = []
    for t in range(rank, tw * th, world):
        x0, y0 = (t % tw) * extent, (t // tw) * extent
        out.append((x0, y0, min(x0 + extent, width), min(y0 + extent, height)))
    return out


def reduce_frame(frame, dst=0):
    """The frame's single collective: SUM-reduce the fp32 radiance buffer to rank `dst` (RCCL on GPUs, gloo on CPU)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM)
    return frame
