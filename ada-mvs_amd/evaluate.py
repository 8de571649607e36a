"""`train_whu.py --mode test` of the reference on the MI355X path.

    reference train_whu.py:213-262 (test), 303-342 (test_sample); models/adamvs.py:8-46 (cas_mvs_vis_loss)

The model is the train/test twin `AdaMVSNet` in eval mode (ada_mvs_amd/models/adamvs.py) -- forward only.  A sample is
what the reference's test dataset yields: "imgs" [B,V,3,H,W], "proj_matrices" {"stage1..3": [B,V,4,4]}, "depth_values"
[B,3] = (min, max, interval), "depth" / "mask" {"stage1..3": [B,h,w]}, "depth_interval" [B], and for the output folder
"outimage", "outcam", "out_view", "out_name".  The reference's own test dataset (datasets/cas_total_rscv.py) reads
OpenEXR depth maps through cv2 and is not rebuilt here: any torch Dataset with these keys plugs in.
"""
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from .datasets.data_io import save_pfm, write_red_cam
from .utils import AbsDepthError_metrics, DictAverageMeter, Inter_metrics, Thres_metrics, tensor2float, tensor2numpy, tocuda


def cas_mvs_vis_loss(inputs, depth_gt_ms, mask_ms, **kwargs):
    """Value of the training loss the test mode prints next to the metrics (reference models/adamvs.py:8-46): per stage,
    smooth-L1 of the depth map (the reference slices `[0:1]` and indexes with the batch's mask: batch size 1, as its test
    loader uses) and the mean smooth-L1 of the per-view depths, both resampled to the ground truth's size, weighted by dlossw.  -> (total_loss, depth_loss of the last stage)"""
    weights = kwargs.get("dlossw", None)
    total = torch.tensor(0.0, dtype=torch.float32, device=mask_ms["stage1"].device)
    depth_loss = None
    for key in [k for k in inputs.keys() if "stage" in k]:
        stage = inputs[key]
        gt, mask = depth_gt_ms[key], mask_ms[key] > 0.5
        size = [gt.shape[1], gt.shape[2]]
        est = F.interpolate(stage["depth"][0:1].unsqueeze(1), size, mode="bilinear", align_corners=False).squeeze(1)
        depth_loss = F.smooth_l1_loss(est[mask], gt[mask], reduction="mean")
        pair_loss = 0
        if len(stage["pair_result"]) > 0:
            for pair in stage["pair_result"]:
                pair = F.interpolate(pair.unsqueeze(1), size, mode="bilinear", align_corners=False).squeeze(1)
                pair_loss = pair_loss + F.smooth_l1_loss(pair[mask], gt[mask], reduction="mean")
            pair_loss = pair_loss / len(stage["pair_result"])
        w = weights[int(key.replace("stage", "")) - 1] if weights is not None else 1.0
        total = total + w * pair_loss + w * depth_loss
    return total, depth_loss


@torch.no_grad()
def test_sample(model, sample, num_stage=3, dlossw=(0.5, 1.0, 2.0), detailed_summary=True):
    """One batch through the model + the scalars of reference train_whu.py:303-342."""
    model.eval()
    s = tocuda({k: v for k, v in sample.items() if k in ("imgs", "proj_matrices", "depth_values", "depth", "mask", "depth_interval")})
    depth_gt_ms, mask_ms, depth_interval = s["depth"], s["mask"], s["depth_interval"]
    depth_gt, mask = depth_gt_ms["stage%d" % num_stage], mask_ms["stage%d" % num_stage]
    outputs = model(s["imgs"], s["proj_matrices"], s["depth_values"])
    depth_est = outputs["depth"]
    loss, depth_loss = cas_mvs_vis_loss(outputs, depth_gt_ms, mask_ms, dlossw=list(dlossw))
    scalars = {"loss": loss, "depth_loss": depth_loss}
    images = {"depth_est": depth_est, "photometric_confidence": outputs["photometric_confidence"], "depth_gt": depth_gt,
              "ref_img": sample["imgs"][:, 0], "mask": mask}
    saved = {k: sample[k] for k in ("outimage", "outcam", "out_view", "out_name") if k in sample}
    if detailed_summary:
        images["errormap"] = (depth_est - depth_gt).abs() * mask
    valid = mask > 0.5
    scalars["abs_depth_error"] = AbsDepthError_metrics(depth_est, depth_gt, valid, float(depth_interval * 100.0))
    scalars["thres1interval_error"] = Thres_metrics(depth_est, depth_gt, valid, float(depth_interval * 1.0))
    scalars["thres6interval_error"] = Thres_metrics(depth_est, depth_gt, valid, float(depth_interval * 6.0))
    scalars["thres3interval_error"] = Inter_metrics(depth_est, depth_gt, depth_interval, valid, 3)
    return tensor2float(loss), tensor2float(scalars), images, saved


def _imsave(path, arr, fmt):
    """plt.imsave as the reference calls it (train_whu.py:252-258): default colormap for 2-D maps, RGB(A) arrays as they are."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    plt.imsave(path, arr, format=fmt)


def test(model, loader, output_folder=None, num_stage=3, dlossw=(0.5, 1.0, 2.0), log=print):
    """The reference's test() (train_whu.py:213-262): every sample through test_sample, the running mean of the scalars,
    and -- when `output_folder` is given and the samples carry "out_view" / "out_name" -- the reference's files per sample:
    <view>/<name>_init.pfm, _prob.pfm, <name>.txt (with "outcam"), <name>.jpg (with "outimage"; an input of the downstream
    fusion step) and the renderings <view>/color/<name>_init.png (36000 - depth) and _prob.png."""
    meter = DictAverageMeter()
    t0 = time.time()
    for i, sample in enumerate(loader):
        t0 = time.time()
        _, scalars, images, saved = test_sample(model, sample, num_stage, dlossw)
        meter.update(scalars)
        log("Iter {}/{}, time = {:3f}, test results = {}".format(i, len(loader), time.time() - t0,
                                                                 {k: float("{0:.6f}".format(v)) for k, v in scalars.items()}))
        if output_folder is not None and "out_name" in saved:
            depth = np.float32(np.squeeze(tensor2numpy(images["depth_est"])))
            prob = np.float32(np.squeeze(tensor2numpy(images["photometric_confidence"])))
            folder = os.path.join(output_folder, str(saved["out_view"][0]))
            os.makedirs(os.path.join(folder, "color"), exist_ok=True)
            name = saved["out_name"][0]
            save_pfm(os.path.join(folder, "%s_init.pfm" % name), depth)
            save_pfm(os.path.join(folder, "%s_prob.pfm" % name), prob)
            if "outimage" in saved:
                _imsave(os.path.join(folder, "%s.jpg" % name), np.squeeze(tensor2numpy(saved["outimage"])), "jpg")
            if "outcam" in saved:
                write_red_cam(os.path.join(folder, "%s.txt" % name), np.squeeze(tensor2numpy(saved["outcam"])), str(name))
            _imsave(os.path.join(folder, "color", "%s_init.png" % name), np.float32(36000.0) - depth, "png")
            _imsave(os.path.join(folder, "color", "%s_prob.png" % name), prob, "png")
    log("final, time = {:3f}, test results = {}".format(time.time() - t0, meter.mean()))     # the last iteration's time, as the reference prints it
    return meter.mean()
