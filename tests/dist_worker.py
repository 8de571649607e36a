"""One rank of a sharded run of the real HIP path (helper of tests/test_dist_gpu.py; started as a fresh child process).

    RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT in the environment, as torchrun sets them;
    python tests/dist_worker.py <cfg> <n_tiles> <out.npz>

Every rank runs Infer_AdaMVSNet on its tiles (tiles r, r + world, ...; images seeded by the global tile index, rig b of
an n_tiles-tile batch), three steps through one MapGatherer like bench.py; rank 0 writes the gathered maps.
ADAMVS_BENCH_ONE_DEVICE=1 puts every rank on device 0 (with ADAMVS_DIST_BACKEND=gloo: the 1-GPU dry run of the N > 1 path).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ada_mvs_amd  # noqa: E402,F401
from ada_mvs_amd import dist as adist, synth  # noqa: E402  (sets HSA_ENABLE_IPC_MODE_LEGACY before the GPU is touched)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def latency_mode(cfg, out, rank, world, dev):
    """Every rank holds the same tile; source views are dealt to the ranks for pass A of stage 1 (one all_gather)."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    c = synth.CONFIGS[cfg] if cfg in synth.CONFIGS else dict(views=9, H=64, W=96, ndepths=[16, 8, 4], num_depth=16)
    m = Infer_AdaMVSNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m = m.to(dev).eval()
    m.view_shard = (rank, world)
    imgs, proj, dv = synth.tile_inputs(c, batch=2, seed=3)
    with torch.no_grad():
        o = m(imgs.to(dev), {k: v.to(dev) for k, v in proj.items()}, dv.to(dev))
    torch.cuda.synchronize()
    np.savez(out + ".rank%d.npz" % rank, depth=o["depth"].cpu().numpy(), conf=o["photometric_confidence"].cpu().numpy(),
             vw=torch.stack([t[:, 0] for t in o["stage1"]["pair_confidence"][:c["views"] - 1]]).cpu().numpy(),
             pd=torch.stack(o["stage1"]["pair_result"]).cpu().numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def main():
    cfg, n_tiles, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, world, local = adist.init_from_env()
    if n_tiles == 0:                             # latency mode
        if os.environ.get("ADAMVS_BENCH_ONE_DEVICE"):
            local = 0
        torch.cuda.set_device(local)
        return latency_mode(cfg, out, rank, world, torch.device("cuda", local))
    if os.environ.get("ADAMVS_BENCH_ONE_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    c = synth.CONFIGS[cfg]
    m = Infer_AdaMVSNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO[:len(c["ndepths"])], False, [8, 8, 8])
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m = m.to(dev).eval()
    tiles = adist.tiles_of_rank(n_tiles, rank, world)
    imgs = torch.cat([synth.tile_inputs(cfg, 1, seed=t)[0] for t in tiles], 0).to(dev)
    _, proj, dv = synth.tile_inputs(cfg, batch=n_tiles, seed=0)
    proj = {k: v[tiles].to(dev) for k, v in proj.items()}
    dv = dv[tiles].to(dev)
    g = adist.MapGatherer(n_tiles, len(tiles), c["H"], c["W"], dev)
    with torch.no_grad():
        for _ in range(3):
            o = m(imgs, proj, dv)
            g.start(o["depth"], o["photometric_confidence"])
    depth, conf = g.finish()
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out, depth=depth.cpu().numpy(), conf=conf.cpu().numpy(), backend=torch.distributed.get_backend())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
