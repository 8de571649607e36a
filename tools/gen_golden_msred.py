#!/usr/bin/env python3
"""Golden fixtures of the reference's MS-REDNet inference model, build container only.

Runs /root/reference/models/{msrednet,module}.py themselves (never copied; `.cuda()` no-op'ed for the CPU run, as in
tools/gen_golden.py) on the seeded recipes of ada-mvs_amd/synth.py:
  msred_gru_cell.npz    ConvGRUCell2 (x 16 ch, h 16 ch)
  msred_slice_step.npz  two consecutive slice_RED_Regularization steps, C = 32
  msred_e2e_tiny.npz    Infer_CascadeREDNet end to end, 3 views, 64x96, ndepths 16/8/4
  msred_featnet_fpn.npz FeatureNet(arch_mode="fpn") on 2 images of 32x64 (the variant no model class of the reference selects)

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_msred.py [--only-fpn]
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
torch.Tensor.cuda = lambda self, *a, **k: self

import ada_mvs_amd  # noqa: E402,F401
from ada_mvs_amd import synth  # noqa: E402
from models import module as ref_module  # noqa: E402
from models import msrednet as ref_red  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.manual_seed(0)
torch.set_num_threads(8)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print("wrote %-24s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


@torch.no_grad()
def fpn():
    net = ref_red.FeatureNet(base_channels=8, stride=4, num_stage=3, arch_mode="fpn")
    net.load_state_dict(synth.seeded_state_dict(net, seed=4))
    net.eval()
    x = torch.randn(2, 3, 32, 64, generator=torch.Generator().manual_seed(12))
    out = net(x)
    save("msred_featnet_fpn", x=x, stage1=out["stage1"], stage2=out["stage2"], stage3=out["stage3"])


@torch.no_grad()
def main():
    fpn()
    if "--only-fpn" in sys.argv:
        return
    g = torch.Generator().manual_seed(11)
    cell = ref_module.ConvGRUCell2(16, 16, 3)
    cell.load_state_dict(synth.seeded_state_dict(cell, seed=2))
    x, h = torch.randn(2, 16, 12, 20, generator=g), 0.5 * torch.randn(2, 16, 12, 20, generator=g)
    out, _ = cell(x, h)
    save("msred_gru_cell", x=x, h=h, out=out)

    net = ref_red.slice_RED_Regularization(32, 8)
    net.load_state_dict(synth.seeded_state_dict(net, seed=3))
    B, hh, ww = 2, 16, 24
    states = [torch.zeros(B, 8 << k, hh >> k, ww >> k) for k in range(4)]
    rec = {}
    for step in range(2):
        cost = torch.rand(B, 32, hh, ww, generator=g) * 0.5
        reg, *states = net(cost, *states)
        rec["cost%d" % step] = cost
        rec["reg%d" % step] = reg
        for k, s in enumerate(states):
            rec["state%d_%d" % (k + 1, step)] = s
    save("msred_slice_step", **rec)

    c = synth.CONFIGS["tiny"]
    model = ref_red.Infer_CascadeREDNet(num_depth=c["num_depth"], ndepths=c["ndepths"],
                                        depth_interals_ratio=synth.DEPTH_INTERVALS_RATIO, share_cr=False, cr_base_chs=[8, 8, 8])
    model.load_state_dict(synth.seeded_state_dict(model, seed=0))
    model.eval()
    imgs, proj, dv = synth.tile_inputs("tiny", batch=1, seed=0)
    out = model(imgs, proj, dv)
    save("msred_e2e_tiny", depth=out["depth"], photometric_confidence=out["photometric_confidence"],
         depth_stage1=out["stage1"]["depth"], depth_stage2=out["stage2"]["depth"],
         conf_stage1=out["stage1"]["photometric_confidence"])


if __name__ == "__main__":
    main()
