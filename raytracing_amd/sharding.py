"""Multi-GPU sharding of the render call (one process per GPU, torch.distributed over RCCL/xGMI).

Every (pixel, sample) of the reference's hot loop (main.cpp:972-988) is independent and the scene is read-only, so
the scene is replicated and the loop is split with NO data-path collective:
  * sample-pass sharding: rank r renders sample indices [r*n, (r+1)*n) of ONE (world*n)-sample Hammersley set
    (hammersley_2d takes (i, N): Utils.h:148-151), then a single reduce(SUM) of the fp32 radiance buffer to rank 0;
  * image-tile sharding: interleaved tiles (the reference's TileQueue order, main.cpp:904-935, dealt round-robin);
    tiles are disjoint, so the same reduce(SUM) of zero-initialised buffers assembles the frame.
"""


def sample_window(rank, world, spp_per_rank):
    """Weak scaling (every rank renders spp_per_rank samples of a world * spp_per_rank set):
    -> (spp_begin, spp_count, spp_total) for rt_render on this rank."""
    if not (0 <= rank < world) or spp_per_rank <= 0:
        raise ValueError("bad rank/world/spp")
    return rank * spp_per_rank, spp_per_rank, world * spp_per_rank


def strong_sample_window(rank, world, spp_total):
    """Strong scaling (the frame's spp_total samples are FIXED and cut into `world` windows; the partition is the C++ host
    mirror's, rt::shard_sample_window): -> (spp_begin, spp_count, spp_total)."""
    import raytracing_amd as rta

    if not (0 <= rank < world) or spp_total < world:
        raise ValueError("bad rank/world/spp")
    begin, count = rta.shard_sample_window(rank, world, spp_total)
    return begin, count, spp_total


def tiles_for_rank(width, height, rank, world, extent=64):
    """Row-major extent x extent tiles (TileQueue::next_tile order), tile t owned by rank t % world (the C++ host mirror's
    rt::shard_tiles).  -> list of (x0, y0, x1, y1)."""
    import raytracing_amd as rta

    if not (0 <= rank < world) or extent <= 0:
        raise ValueError("bad rank/world/extent")
    return [tuple(int(v) for v in t) for t in rta.shard_tiles(rank, world, width, height, extent)]


def reduce_frame(frame, dst=0):
    """The frame's single collective: SUM-reduce the fp32 radiance buffer to rank `dst` (RCCL on GPUs, gloo on CPU)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM)
    return frame
