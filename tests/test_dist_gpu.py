"""The N > 1 path with the REAL compute: tile sharding + the per-step gather around Infer_AdaMVSNet on the GPU.

Two fresh child processes per test (never a re-exec of a process that has touched the GPU):
  * both ranks on device 0 over gloo -- runs on the 1-GPU box of the round-end check;
  * one rank per device over RCCL (backend "nccl") -- skipped unless two devices are visible.
The gathered maps must equal, bit for bit, the maps of one process running all tiles as one batch.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
import ada_mvs_amd  # noqa: F401
from ada_mvs_amd import synth

pytestmark = pytest.mark.gpu
CFG, N_TILES = "tiny", 5                      # uneven: rank 0 owns three tiles, rank 1 two


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _single_process_maps():
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    c = synth.CONFIGS[CFG]
    m = Infer_AdaMVSNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m = m.cuda().eval()
    imgs = torch.cat([synth.tile_inputs(CFG, 1, seed=t)[0] for t in range(N_TILES)], 0).cuda()
    _, proj, dv = synth.tile_inputs(CFG, batch=N_TILES, seed=0)
    with torch.no_grad():
        o = m(imgs, {k: v.cuda() for k, v in proj.items()}, dv.cuda())
    return o["depth"].cpu().numpy(), o["photometric_confidence"].cpu().numpy()


def _run_ranks(tmp_path, extra_env, cfg=CFG, n_tiles=N_TILES, world=2):
    port = _free_port()
    out = str(tmp_path / "gathered.npz")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), cfg, str(n_tiles), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-2000:] for l in logs)
    return np.load(out) if n_tiles else [np.load(out + ".rank%d.npz" % r) for r in range(world)]


def _check(z, backend):
    depth, conf = _single_process_maps()
    assert str(z["backend"]) == backend
    assert z["depth"].shape == depth.shape == (N_TILES, synth.CONFIGS[CFG]["H"], synth.CONFIGS[CFG]["W"])
    assert np.array_equal(z["depth"], depth) and np.array_equal(z["conf"], conf)


def test_two_ranks_on_one_device_over_gloo_match_single_process(tmp_path):
    _check(_run_ranks(tmp_path, {"ADAMVS_BENCH_ONE_DEVICE": "1", "ADAMVS_DIST_BACKEND": "gloo"}), "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible GPUs (RCCL over xGMI)")
def test_two_ranks_over_rccl_match_single_process(tmp_path):
    _check(_run_ranks(tmp_path, {}), "nccl")


def test_latency_mode_source_views_sharded_over_three_ranks(tmp_path):
    """SURVEY.md section 8e, cfg5's latency mode: 8 source views dealt to 3 ranks (3 + 3 + 2: uneven) for pass A of stage
    1, one all_gather of the view weights and pair depths, pass B on every rank.  Every rank must end with the maps of a
    single process, bit for bit (a view's weight does not depend on the other views)."""
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    cfg = dict(views=9, H=64, W=96, ndepths=[16, 8, 4], num_depth=16)
    m = Infer_AdaMVSNet(cfg["num_depth"], cfg["ndepths"], synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    m.load_state_dict(synth.seeded_state_dict(m, seed=0))
    m = m.cuda().eval()
    imgs, proj, dv = synth.tile_inputs(cfg, batch=2, seed=3)
    with torch.no_grad():
        o = m(imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv.cuda())
    ranks = _run_ranks(tmp_path, {"ADAMVS_BENCH_ONE_DEVICE": "1", "ADAMVS_DIST_BACKEND": "gloo"}, cfg="nine_views", n_tiles=0, world=3)
    vw = torch.stack([t[:, 0] for t in o["stage1"]["pair_confidence"][:8]]).cpu().numpy()
    for z in ranks:
        assert np.array_equal(z["vw"], vw) and np.array_equal(z["pd"], torch.stack(o["stage1"]["pair_result"]).cpu().numpy())
        assert np.array_equal(z["depth"], o["depth"].cpu().numpy())
        assert np.array_equal(z["conf"], o["photometric_confidence"].cpu().numpy())


def test_bench_two_ranks_strong_scaling_dry_run(tmp_path):
    """bench.py under torch.distributed.run with two ranks, `--tiles-total 5` (strong scaling: tile t on rank t mod 2, an uneven deal), both
    ranks on device 0 over gloo -- the N > 1 code path of the bench (tile ownership, the asynchronous gather, the max-over-ranks
    clock, one JSON line from rank 0) on a 1-GPU box.  Not a scaling point."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, ADAMVS_BENCH_ONE_DEVICE="1", ADAMVS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "cfg1", "--tiles-total", "5"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_tiles_per_step"] == 5
    assert d["value"] > 0 and d["steps"] == 2


def test_bench_self_launches_its_ranks(tmp_path):
    """Plain `python bench.py --gpus 2 ...` with no launcher around it (the way the driver starts the 1-GPU line): the process
    must start two fresh ranks itself and report n_gpus == 2 -- never measure one GPU under a --gpus 2 label."""
    import json
    env = dict(os.environ, ADAMVS_BENCH_ONE_DEVICE="1", ADAMVS_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cfg1", "--tiles-total", "5"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo" and d["launcher"] == "self"
    assert [x["rank"] for x in d["devices"]] == [0, 1] and len({x["pid"] for x in d["devices"]}) == 2
    assert d["scaling"] == "strong" and d["config"]["global_tiles_per_step"] == 5 and d["value"] > 0
    assert d["roofline"] is None and d["cpu_baseline"] is None


def test_bench_cfg4_as_stated_eight_ranks_dry_run(tmp_path):
    """BASELINE config 4 as it is written -- `python bench.py --gpus 8 --workload cfg3 --tiles-total 32 --precision bf16x3`: 32 cascade
    tiles dealt to 8 ranks, 4 per rank, one gather per step -- with all eight ranks on device 0 over gloo: the exact command a node with
    eight GPUs will run has executed end to end (self-launch, tile ownership, per-rank graphs, the gather, the max-over-ranks clock,
    one JSON line).  One device shared by eight processes: NOT a scaling point, only the code path."""
    import json
    env = dict(os.environ, ADAMVS_BENCH_ONE_DEVICE="1", ADAMVS_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--workload", "cfg3",
           "--tiles-total", "32", "--precision", "bf16x3", "--launch-timeout=840"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["backend"] == "gloo" and d["launcher"] == "self"
    assert [x["rank"] for x in d["devices"]] == list(range(8)) and len({x["pid"] for x in d["devices"]}) == 8
    assert d["scaling"] == "strong" and d["config"]["global_tiles_per_step"] == 32 and d["config"]["tiles_per_gpu_per_step"] == "4..4"
    assert d["value"] > 0 and d["roofline"] is None and d["cpu_baseline"] is None
