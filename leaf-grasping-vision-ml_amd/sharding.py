"""Frame sharding across the GPUs of one node: frames are independent, frame i goes to rank
i mod world, every rank owns its handle / stream / workspace / CNN weight copy, and there is NO data-path
collective.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only for the timing barrier,
the max-over-ranks reduction and the optional host-side gather of the tiny per-frame results."""
import time

import torch


def frame_partition(n_frames, rank, world):
    """Indices of the frames rank `rank` scores: i mod world == rank."""
    return list(range(rank, n_frames, world))


def barrier_max_time(fn, dist=None, device=None):
    """Run fn() between two barriers (+ device synchronisation when on a GPU) and return the MAX elapsed
    seconds over all ranks (bench.py's timing contract)."""
    def sync():
        if device is not None and device.type == "cuda":
            torch.cuda.synchronize(device)
        if dist is not None and dist.is_initialized():
            dist.barrier()
        if device is not None and device.type == "cuda":
            torch.cuda.synchronize(device)

    sync()
    t0 = time.perf_counter()
    out = fn()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None and dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def gather_results(local_results, n_frames, rank, world, dist=None):
    """Host-side gather of per-frame results (<= a few hundred bytes per frame) into frame order on every
    rank.  Not a data-path collective: the score planes never leave their GPU."""
    if dist is None or not dist.is_initialized() or world == 1:
        return list(local_results)
    parts = [None] * world
    dist.all_gather_object(parts, list(local_results))
    out = [None] * n_frames
    for r in range(world):
        for j, i in enumerate(frame_partition(n_frames, r, world)):
            out[i] = parts[r][j]
    return out
