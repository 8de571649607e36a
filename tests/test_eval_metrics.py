"""Evaluation side of the reference's `--mode test`: metrics, running mean and the loss
value against numbers the reference's own utils.py / models/adamvs.py produced (tests/golden/eval_metrics.npz,
tools/gen_golden_eval.py); the test loop end to end on the GPU twin model."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import ada_mvs_amd  # noqa: F401
from ada_mvs_amd import evaluate, synth, utils


def test_metrics_match_reference_run():
    g = load_golden("eval_metrics")
    est, gt, mask, interval = g["est"], g["gt"], g["mask"], g["interval"]
    assert float(utils.AbsDepthError_metrics(est, gt, mask, float(interval * 100.0))) == g["abs_depth_error"]
    assert float(utils.AbsDepthError_metrics(est, gt, mask, 1.0)) == g["abs_depth_error_tight"]
    assert float(utils.Thres_metrics(est, gt, mask, float(interval * 1.0))) == g["thres1"]
    assert float(utils.Thres_metrics(est, gt, mask, float(interval * 6.0))) == g["thres6"]
    assert float(utils.Inter_metrics(est, gt, interval, mask, 3)) == g["inter3"]
    m = utils.DictAverageMeter()
    m.update({"a": 1.0, "b": 4.0})
    m.update({"a": 2.0, "b": 0.5})
    assert [m.mean()["a"], m.mean()["b"]] == g["meter_mean"].tolist()
    with pytest.raises(NotImplementedError):
        m.update({"a": torch.tensor(1.0)})
    # an image without a pixel under the threshold: nan, as torch.mean of an empty tensor
    assert torch.isnan(utils.AbsDepthError_metrics(est[2:], gt[2:], mask[2:], 1e-9))


def test_converters_walk_nested_containers():
    t = torch.tensor([1.5])
    assert utils.tensor2float({"a": [t, (t, 2.0)]}) == {"a": [1.5, (1.5, 2.0)]}
    assert isinstance(utils.tensor2numpy({"a": t})["a"], np.ndarray)
    with pytest.raises(NotImplementedError):
        utils.tensor2numpy({"a": "text"})                   # reference utils.py:54-61: tensors and arrays only


def test_loss_value_matches_reference_run():
    g = load_golden("eval_metrics")
    inputs = {}
    for k in ("stage1", "stage2", "stage3"):
        inputs[k] = {"depth": g["loss_%s_depth" % k],
                     "pair_result": [g["loss_stage1_pair%d" % i] for i in range(2)] if k == "stage1" else []}
    inputs["depth"] = inputs["stage3"]["depth"]
    gts = {k: g["loss_gt_%s" % k] for k in ("stage1", "stage2", "stage3")}
    masks = {k: g["loss_mask_%s" % k] for k in ("stage1", "stage2", "stage3")}
    total, last = evaluate.cas_mvs_vis_loss(inputs, gts, masks, dlossw=[0.5, 1.0, 2.0])
    assert float(total) == pytest.approx(float(g["loss_total"]), rel=1e-6) and float(last) == float(g["loss_last"])
    total1, _ = evaluate.cas_mvs_vis_loss(inputs, gts, masks)
    assert float(total1) == pytest.approx(float(g["loss_total_unweighted"]), rel=1e-6)


@pytest.mark.gpu
def test_test_mode_loop_on_the_twin_model(tmp_path):
    """reference train_whu.py:213-262 on AdaMVSNet (eval) over two synthetic samples whose ground truth is the model's own
    depth plus a known offset: the metrics are then known in closed form."""
    from ada_mvs_amd.models.adamvs import AdaMVSNet
    from ada_mvs_amd.datasets.data_io import read_pfm
    c = synth.CONFIGS["tiny"]
    model = AdaMVSNet(c["ndepths"], synth.DEPTH_INTERVALS_RATIO)
    model.load_state_dict(synth.seeded_state_dict(model, seed=0))
    model = model.cuda().eval()
    interval = (synth.DEPTH_RANGE[1] - synth.DEPTH_RANGE[0]) / c["num_depth"]
    samples = []
    for seed in (0, 1):
        imgs, proj, dv = synth.tile_inputs("tiny", batch=1, seed=seed)
        dv3 = torch.cat([dv, torch.full((1, 1), interval)], 1)
        with torch.no_grad():
            out = model(imgs.cuda(), {k: v.cuda() for k, v in proj.items()}, dv3.cuda())
        gt3 = out["depth"].cpu() + 0.5 * interval                       # every pixel off by half an interval
        depth = {"stage1": gt3[:, ::4, ::4].contiguous(), "stage2": gt3[:, ::2, ::2].contiguous(), "stage3": gt3}
        mask = {k: torch.ones_like(v) for k, v in depth.items()}
        mask["stage3"][:, :8] = 0
        cam = np.zeros((1, 2, 4, 4), dtype=np.float32)
        cam[0, 1, 3] = [400, interval, c["num_depth"], 600]
        samples.append({"imgs": imgs, "proj_matrices": proj, "depth_values": dv3, "depth": depth, "mask": mask,
                        "depth_interval": torch.tensor([interval]), "outimage": torch.zeros(1, c["H"], c["W"], 3),
                        "outcam": torch.from_numpy(cam), "out_view": ["view%d" % seed], "out_name": ["%03d" % seed]})
    lines = []
    mean = evaluate.test(model, samples, output_folder=str(tmp_path), log=lines.append)
    assert len(lines) == 3 and lines[-1].startswith("final")
    assert mean["abs_depth_error"] == pytest.approx(0.5 * interval, rel=1e-4)
    assert mean["thres1interval_error"] == 1.0 and mean["thres6interval_error"] == 1.0 and mean["thres3interval_error"] == 1.0
    assert mean["loss"] > 0 and mean["depth_loss"] == pytest.approx(0.5 * interval - 0.5, rel=1e-3)     # smooth-L1 beyond 1: |x| - 0.5
    depth0, _ = read_pfm(str(tmp_path / "view0" / "000_init.pfm"))
    assert depth0.shape == (c["H"], c["W"])
    assert (tmp_path / "view1" / "001_prob.pfm").exists() and (tmp_path / "view1" / "001.txt").exists()
