"""Tile-level sharding across the GPUs of one node.

Reference tiles are independent (reference predict_whu.py:100-104 walks them
one by one), so rank r of N takes tiles r, r+N, r+2N, ...; weights are
replicated; the only communication is one gather of the finished depth /
confidence maps to rank 0 (RCCL over xGMI when the backend is "nccl", gloo in
the CPU tests).  No collective sits on the data path.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("ADAMVS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def tiles_of_rank(n_tiles, rank, world):
    """Round-robin tile ownership: rank r owns tiles r, r+world, ..."""
    return list(range(rank, n_tiles, world))


def gather_maps(depth, conf, n_tiles, dst=0):
    """Gather per-rank [T_r,H,W] depth / confidence maps to `dst` in global tile order.

    Ranks may own different numbers of tiles (n_tiles not divisible by world): shorter
    ranks are padded to the maximum for the collective.  Returns (depth, conf) of shape
    [n_tiles,H,W] on `dst`, (None, None) elsewhere.  With world == 1 it is the identity.
    """
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return depth, conf
    rank, world = dist.get_rank(), dist.get_world_size()
    per_rank = (n_tiles + world - 1) // world
    H, W = depth.shape[-2:]
    dev = depth.device
    if dist.get_backend() == "gloo" and dev.type != "cpu":
        dev = torch.device("cpu")          # gloo gathers host tensors (CPU tests, single-GPU dry runs of the N > 1 path)
    own = depth.shape[0]
    pack = torch.empty(2, per_rank, H, W, device=dev, dtype=depth.dtype)
    pack[0, :own] = depth
    pack[1, :own] = conf
    if own < per_rank:
        pack[:, own:] = 0
    bufs = [torch.empty_like(pack) for _ in range(world)] if rank == dst else None
    dist.gather(pack, bufs, dst=dst)
    if rank != dst:
        return None, None
    out_d = torch.empty(n_tiles, H, W, device=dev, dtype=depth.dtype)
    out_c = torch.empty_like(out_d)
    for r in range(world):                   # round-robin ownership = a strided slice (one copy kernel per map)
        n_r = len(range(r, n_tiles, world))
        out_d[r::world] = bufs[r][0, :n_r]
        out_c[r::world] = bufs[r][1, :n_r]
    return out_d, out_c


def run_sharded(infer_tiles, n_tiles, dst=0):
    """infer_tiles(list_of_tile_indices) -> (depth [T,H,W], conf [T,H,W]) for this rank's tiles;
    returns the gathered maps on `dst`."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    depth, conf = infer_tiles(tiles_of_rank(n_tiles, rank, world))
    return gather_maps(depth, conf, n_tiles, dst)
