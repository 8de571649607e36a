"""Golden fixtures for the evaluation metrics and the loss value of the reference's test mode.

    python tools/gen_golden_eval.py          (in the build container; needs /root/reference)

Runs the reference's own utils.py (Thres_metrics, Inter_metrics, AbsDepthError_metrics, DictAverageMeter) and
models/adamvs.py::cas_mvs_vis_loss on seeded tensors.  utils.py imports torchvision.utils at module level for its
TensorBoard helpers; torchvision is not installed, so a stub module whose every attribute access raises lets the
module import -- none of the functions run here touches it.  Writes tests/golden/eval_metrics.npz.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Absent(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        raise RuntimeError("%s is not installed" % self.__name__)


tv = _Absent("torchvision")
tv.utils = _Absent("torchvision.utils")
sys.modules["torchvision"], sys.modules["torchvision.utils"] = tv, tv.utils
sys.path.insert(0, "/root/reference")
import utils as ref_utils  # noqa: E402
from models.adamvs import cas_mvs_vis_loss as ref_loss  # noqa: E402


def main():
    g = torch.Generator().manual_seed(3)
    B, H, W = 3, 20, 28
    gt = 400 + 200 * torch.rand(B, H, W, generator=g)
    est = gt + torch.randn(B, H, W, generator=g) * torch.tensor([0.5, 3.0, 40.0]).reshape(B, 1, 1)
    mask = (torch.rand(B, H, W, generator=g) > 0.3)
    interval = torch.tensor([1.25])
    out = {"gt": gt.numpy(), "est": est.numpy(), "mask": mask.numpy(), "interval": interval.numpy()}
    out["abs_depth_error"] = ref_utils.AbsDepthError_metrics(est, gt, mask, float(interval * 100.0)).numpy()
    out["abs_depth_error_tight"] = ref_utils.AbsDepthError_metrics(est, gt, mask, 1.0).numpy()
    out["thres1"] = ref_utils.Thres_metrics(est, gt, mask, float(interval * 1.0)).numpy()
    out["thres6"] = ref_utils.Thres_metrics(est, gt, mask, float(interval * 6.0)).numpy()
    out["inter3"] = ref_utils.Inter_metrics(est, gt, interval, mask, 3).numpy()
    m = ref_utils.DictAverageMeter()
    m.update({"a": 1.0, "b": 4.0})
    m.update({"a": 2.0, "b": 0.5})
    out["meter_mean"] = np.array([m.mean()["a"], m.mean()["b"]])
    # loss: three stages, two source views at stage 1 only (as the model outputs carry them); batch size 1 -- the
    # reference slices `depth[0:1]` and indexes it with the whole batch's mask, which only works for one sample
    B = 1
    sizes = {"stage1": (10, 14), "stage2": (20, 28), "stage3": (20, 28)}
    gts = {"stage1": gt[:1, ::2, ::2].contiguous(), "stage2": gt[:1], "stage3": gt[:1]}
    masks = {"stage1": mask[:1, ::2, ::2].float().contiguous(), "stage2": mask[:1].float(), "stage3": mask[:1].float()}
    inputs = {}
    for k, (h, w) in sizes.items():
        st = {"depth": 400 + 200 * torch.rand(B, h, w, generator=g),
              "pair_result": [400 + 200 * torch.rand(B, 5, 7, generator=g) for _ in range(2)] if k == "stage1" else []}
        inputs[k] = st
        out["loss_%s_depth" % k] = st["depth"].numpy()
        for i, p in enumerate(st["pair_result"]):
            out["loss_%s_pair%d" % (k, i)] = p.numpy()
    inputs["depth"] = inputs["stage3"]["depth"]               # non-stage keys are skipped by the loss
    for k in sizes:
        out["loss_gt_%s" % k], out["loss_mask_%s" % k] = gts[k].numpy(), masks[k].numpy()
    total, last = ref_loss(inputs, gts, masks, dlossw=[0.5, 1.0, 2.0])
    out["loss_total"], out["loss_last"] = total.numpy(), last.numpy()
    total1, _ = ref_loss(inputs, gts, masks)
    out["loss_total_unweighted"] = total1.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "eval_metrics.npz"), **out)
    print("wrote eval_metrics.npz", {k: float(v) for k, v in out.items() if np.ndim(v) == 0})


if __name__ == "__main__":
    main()
