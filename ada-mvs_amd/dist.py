"""Tile-level sharding across the GPUs of one node.

Reference tiles are independent (reference predict_whu.py:100-104 walks them
one by one), so rank r of N takes tiles r, r+N, r+2N, ...; weights are
replicated; the only communication is one gather of the finished depth /
confidence maps to rank 0 (RCCL over xGMI when the backend is "nccl", gloo in
the CPU tests).  No collective sits on the data path.
"""
import os

# the host driver of this pool only supports dmabuf IPC; must be in the environment before the HIP runtime starts,
# i.e. before anything in the process touches the GPU (importing this module first is enough)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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


class MapGatherer:
    """The per-step gather of a sharded run with everything allocated once and the collective off the critical path.

        g = MapGatherer(n_tiles, tiles_here, H, W, device)        # once
        for step in ...:
            replay the hot path -> depth, conf [tiles_here, H, W]
            g.start(depth, conf)       # copies the maps out of the (reused) output buffers, issues the gather, returns
        depth_all, conf_all = g.finish()                          # rank dst: [n_tiles, H, W] of the LAST started step

    start() of step k+1 first completes the gather of step k (its result is overwritten: a throughput loop only needs
    the last one).  With the RCCL backend the collective runs on RCCL's own stream, after the copy and concurrently
    with the next replay; two staging buffers alternate so that a gather in flight is never overwritten.  With gloo
    (CPU tests, single-GPU dry runs) the staging buffers live in host memory and the gather is synchronous."""

    def __init__(self, n_tiles, tiles_here, H, W, device, dst=0, dtype=torch.float32):
        self.n_tiles, self.own, self.dst = n_tiles, tiles_here, dst
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.work, self.slot, self.last = None, 0, None
        if self.world == 1:
            return
        self.per_rank = (n_tiles + self.world - 1) // self.world
        self.host = dist.get_backend() == "gloo"
        dev = torch.device("cpu") if self.host else device
        self.pack = [torch.zeros(2, self.per_rank, H, W, device=dev, dtype=dtype) for _ in range(2)]
        if self.host and device.type != "cpu":
            self.pack = [p.pin_memory() for p in self.pack]
        self.bufs = [torch.empty(2, self.per_rank, H, W, device=dev, dtype=dtype) for _ in range(self.world)] if self.rank == dst else None
        self.out = torch.empty(2, n_tiles, H, W, device=dev, dtype=dtype) if self.rank == dst else None

    def start(self, depth, conf):
        if self.world == 1:
            self.last = (depth, conf)
            return
        self._complete()
        pack = self.pack[self.slot]
        self.slot ^= 1
        pack[0, :self.own].copy_(depth, non_blocking=True)
        pack[1, :self.own].copy_(conf, non_blocking=True)
        if self.host:
            if depth.is_cuda:
                torch.cuda.current_stream().synchronize()
            dist.gather(pack, self.bufs, dst=self.dst)
        else:
            self.work = dist.gather(pack, self.bufs, dst=self.dst, async_op=True)

    def _complete(self):
        if self.work is not None:
            self.work.wait()               # orders the current stream after the collective; does not block the host
            self.work = None

    def finish(self):
        """-> (depth, conf) [n_tiles, H, W] in global tile order on `dst`, (None, None) elsewhere."""
        if self.world == 1:
            return self.last
        self._complete()
        if self.rank != self.dst:
            return None, None
        for r in range(self.world):              # round-robin ownership = a strided slice (one copy per rank)
            n_r = len(range(r, self.n_tiles, self.world))
            self.out[:, r::self.world] = self.bufs[r][:, :n_r]
        return self.out[0], self.out[1]


def views_of_rank(n_views, rank, world):
    """Source views s = rank, rank + world, ... of the S source views."""
    return list(range(rank, n_views, world))


def all_gather_maps(t):
    """all_gather of equally shaped maps -> [world, *t.shape] on every rank (RCCL on device tensors; gloo through the host)."""
    world = dist.get_world_size()
    if dist.get_backend() == "gloo":
        host = t.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host)
        return torch.stack(parts, 0).to(t.device)
    out = torch.empty((world,) + tuple(t.shape), device=t.device, dtype=t.dtype)
    dist.all_gather_into_tensor(out, t.contiguous())
    return out


def stage_with_sharded_views(net, feat_cl, B, C, h, w, rt, planes, num_depth, group, twin, workspaces, rank, world):
    """Stage 1 of one tile (or batch) computed by `world` ranks together -- the latency mode of SURVEY.md section 8e for
    BASELINE cfg5 (1 reference + 8 source views on 8 GPUs).  Pass A (pair similarity, CostRegNet2D, softmax / max /
    regression: reference models/adamvs.py:462-490) is independent per source view and carries 62 % of cfg5's flops:
    rank r scores source views r, r + world, ...; ONE all_gather of the view weights and pair depths ([S][B][h*w] each,
    295 KB per view at cfg5) follows; every rank then runs pass B (aggregation, recurrence, soft-argmin) on all
    views.  The result equals the single-process stage bit for bit (a view's weights do not depend on the other
    views).  feat_cl [V*B, h*w, C] view-major, rt [B,S,12]: every rank holds all views (the images are replicated).
    -> (view_weight [S,B,h,w], pair_depth [S,B,h,w], depth, confidence)"""
    from . import _lib
    S = feat_cl.shape[0] // B - 1
    dev = feat_cl.device
    mine = views_of_rank(S, rank, world)
    per_rank = (S + world - 1) // world
    part = torch.zeros(2, per_rank, B, h, w, device=dev, dtype=torch.float32)          # [vw | pd][local view]
    if mine:
        idx = torch.tensor([0] + [1 + v for v in mine], device=dev)
        feat_sub = feat_cl.reshape(S + 1, B, h * w, C).index_select(0, idx).reshape(-1, h * w, C)
        rt_sub = rt.index_select(1, torch.tensor(mine, device=dev)).contiguous()
        n = len(mine)
        unused = torch.empty(1, device=dev, dtype=torch.float32)           # depth / confidence are not written by pass A
        net.run(feat_sub, B, C, h, w, rt_sub, None, None, group, twin, planes=planes, num_depth=num_depth, workspaces=workspaces,
                phases=_lib.PHASE_VIEW_WEIGHTS, outputs=(part[0, :n], part[1, :n], unused, unused))     # leading slices: contiguous [n,B,h,w]
    gathered = all_gather_maps(part) if world > 1 else part[None]                       # [world, 2, per_rank, B, h, w]
    vw = torch.empty(S, B, h, w, device=dev, dtype=torch.float32)
    pd = torch.empty(S, B, h, w, device=dev, dtype=torch.float32)
    for r in range(world):
        for i, v in enumerate(views_of_rank(S, r, world)):
            vw[v] = gathered[r, 0, i]
            pd[v] = gathered[r, 1, i]
    Ho, Wo = (2 * h, 2 * w) if net.in_up else (h, w)
    depth = torch.empty(B, Ho, Wo, device=dev, dtype=torch.float32)
    conf = torch.empty(B, Ho, Wo, device=dev, dtype=torch.float32)
    net.run(feat_cl, B, C, h, w, rt, None, None, group, twin, planes=planes, num_depth=num_depth, workspaces=workspaces,
            phases=_lib.PHASE_AGGREGATE | _lib.PHASE_RECURRENCE | _lib.PHASE_SOFT_ARGMIN, outputs=(vw, pd, depth, conf))
    return vw, pd, depth, conf


def run_sharded(infer_tiles, n_tiles, dst=0):
    """infer_tiles(list_of_tile_indices) -> (depth [T,H,W], conf [T,H,W]) for this rank's tiles;
    returns the gathered maps on `dst`."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    depth, conf = infer_tiles(tiles_of_rank(n_tiles, rank, world))
    return gather_maps(depth, conf, n_tiles, dst)
