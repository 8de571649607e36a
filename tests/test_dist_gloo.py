"""N > 1 path on CPU: world_size 2, gloo.  Tile sharding + the final gather;
the per-tile compute is replaced by a stand-in because the HIP path needs a GPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ada_mvs_amd  # noqa: F401
from ada_mvs_amd import dist as adist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_tiles, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = adist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)

    def infer(tiles):          # stand-in for the HIP hot path: map value = global tile index
        d = torch.stack([torch.full((4, 6), float(t)) for t in tiles]) if tiles else torch.zeros(0, 4, 6)
        return d, d + 0.5

    depth, conf = adist.run_sharded(infer, n_tiles, dst=0)
    if rank == 0:
        q.put((depth.clone(), conf.clone()))
    else:
        assert depth is None and conf is None
    dist.barrier()
    dist.destroy_process_group()


def _worker_steps(rank, world, port, n_tiles, q):
    """bench.py's loop: one MapGatherer, several steps, the result of the last started step survives."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    adist.init_from_env(backend="gloo")
    tiles = adist.tiles_of_rank(n_tiles, rank, world)
    g = adist.MapGatherer(n_tiles, len(tiles), 4, 6, torch.device("cpu"))
    out = torch.zeros(max(len(tiles), 1), 4, 6)[:len(tiles)]          # reused "graph output" buffers
    for step in range(3):
        for i, t in enumerate(tiles):
            out[i] = 100.0 * step + t
        g.start(out, out + 0.25)
    depth, conf = g.finish()
    if rank == 0:
        q.put((depth.clone(), conf.clone()))
    else:
        assert depth is None and conf is None
    dist.barrier()
    dist.destroy_process_group()


def _worker_every_step(rank, world, port, n_tiles, q):
    """A predict-style caller: finish() after EVERY start(), the result consumed (cloned) per step -- next to the throughput loop
    above, where only the last step's maps are read.  The staging buffers alternate and the receive set is reused: a step's
    result must never carry the previous or the next step's values."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    adist.init_from_env(backend="gloo")
    tiles = adist.tiles_of_rank(n_tiles, rank, world)
    g = adist.MapGatherer(n_tiles, len(tiles), 4, 6, torch.device("cpu"))
    out = torch.zeros(max(len(tiles), 1), 4, 6)[:len(tiles)]          # reused "graph output" buffers
    per_step = []
    for step in range(4):
        for i, t in enumerate(tiles):
            out[i] = 100.0 * step + t
        g.start(out, out + 0.25)
        out.fill_(-1.0)                                               # the caller reuses its buffers at once: start() has copied
        depth, conf = g.finish()
        if rank == 0:
            per_step.append((depth.clone(), conf.clone()))
        else:
            assert depth is None and conf is None
    if rank == 0:
        q.put((torch.stack([d for d, _ in per_step]), torch.stack([c for _, c in per_step])))
    dist.barrier()
    dist.destroy_process_group()


def _worker_views(rank, world, port, n_views, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    adist.init_from_env(backend="gloo")
    mine = adist.views_of_rank(n_views, rank, world)
    per_rank = (n_views + world - 1) // world
    part = torch.zeros(per_rank, 2, 3)
    for i, v in enumerate(mine):
        part[i] = 10.0 * v                          # the "view weight" of source view v
    g = adist.all_gather_maps(part)                 # [world, per_rank, 2, 3] on every rank
    full = torch.empty(n_views, 2, 3)
    for r in range(world):
        for i, v in enumerate(adist.views_of_rank(n_views, r, world)):
            full[v] = g[r, i]
    if rank == 0:
        q.put((full.clone(), full.clone()))
    dist.barrier()
    dist.destroy_process_group()


def _run(n_tiles, worker=None):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=worker or _worker, args=(r, 2, port, n_tiles, q)) for r in range(2)]
    for p in procs:
        p.start()
    depth, conf = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return depth, conf


def test_tile_ownership_is_a_partition():
    for n, world in ((32, 8), (5, 2), (3, 4), (1, 8)):
        owned = sum((adist.tiles_of_rank(n, r, world) for r in range(world)), [])
        assert sorted(owned) == list(range(n))


def test_gather_orders_tiles_globally_even_and_uneven():
    for n_tiles in (4, 5):
        depth, conf = _run(n_tiles)
        assert depth.shape == (n_tiles, 4, 6)
        for t in range(n_tiles):
            assert float(depth[t].min()) == float(depth[t].max()) == float(t)
            assert float(conf[t, 0, 0]) == t + 0.5


def test_single_process_is_identity():
    d, c = torch.ones(2, 3, 3), torch.zeros(2, 3, 3)
    gd, gc = adist.gather_maps(d, c, 2)
    assert gd is d and gc is c


def test_map_gatherer_steps_even_and_uneven():
    """Preallocated, double-buffered gather of bench.py's step loop: after three steps rank 0 holds step 2's maps."""
    for n_tiles in (4, 5):
        depth, conf = _run(n_tiles, _worker_steps)
        assert depth.shape == (n_tiles, 4, 6)
        for t in range(n_tiles):
            assert float(depth[t].min()) == float(depth[t].max()) == 200.0 + t
            assert float(conf[t, 0, 0]) == 200.25 + t


def test_map_gatherer_consumed_after_every_step():
    """finish() after every start(): each step's gathered maps are that step's, for even and uneven tile counts."""
    for n_tiles in (4, 5):
        depth, conf = _run(n_tiles, _worker_every_step)
        assert depth.shape == (4, n_tiles, 4, 6)
        for step in range(4):
            for t in range(n_tiles):
                assert float(depth[step, t].min()) == float(depth[step, t].max()) == 100.0 * step + t
                assert float(conf[step, t].min()) == float(conf[step, t].max()) == 100.0 * step + t + 0.25


def test_map_gatherer_single_process_is_identity():
    g = adist.MapGatherer(2, 2, 3, 3, torch.device("cpu"))
    d, c = torch.ones(2, 3, 3), torch.zeros(2, 3, 3)
    g.start(d, c)
    gd, gc = g.finish()
    assert gd is d and gc is c


def test_source_views_are_dealt_and_gathered_back():
    """The exchange step of the latency mode (cfg5: source views over ranks): every rank reassembles all views."""
    for n_views in (8, 5, 1):
        owned = sum((adist.views_of_rank(n_views, r, 2) for r in range(2)), [])
        assert sorted(owned) == list(range(n_views))
        full, _ = _run(n_views, _worker_views)
        assert [float(full[v, 0, 0]) for v in range(n_views)] == [10.0 * v for v in range(n_views)]
