"""Which narrow operand formats hold the 1e-3 bar?  A study on the CPU oracle.  TEST INFRASTRUCTURE (imports oracle/).

    python tests/precision_study.py [--cfg cfg1] [--recipes default,sharp] [--formats ...] [--out profiles/r06_precision_study.jsonl]

Method (the one of DESIGN.md "bf16: what passes"): the oracle's convolutions of the named network parts are re-run with their
OPERANDS rounded the way a matrix instruction with fp32 accumulation would see them -- products of two 11-bit (fp16) or 8-bit
(bf16) significands are exact in fp32, so `conv2d(round(x), round(w))` in fp32 is the emulation -- and the final maps are
compared with the plain fp32 oracle (relative L1, the north star's measure).  Everything that is not a convolution operand
stays fp32: states, blends, sigmoid / tanh, the softmax, the soft-argmin sums, the warp and the aggregation.

Operand formats (x = activations, w = weights; "MFMAs" = matrix instructions per product on gfx950):

    f16        x fp16, w fp16                                   1 MFMA   v_mfma_f32_16x16x32_f16
    f16_ws     x fp16, w = fp16 hi + fp16 lo (22 bits)          2 MFMAs  one cvt per activation, no split of x
    f16_xs     x = fp16 hi + lo, w fp16                         2 MFMAs
    f16x3      both as pairs (hi.hi + hi.lo + lo.hi)            3 MFMAs
    bf16       x bf16, w bf16                                   1 MFMA
    bf16_ws    x bf16, w fp32-exact (split)                     2 MFMAs
    bf16x3     both as bf16 pairs (the shipped mode)            3 MFMAs

Parts (reference lines):  gru = the five ConvGRU convolutions + conv2 (models/module.py:24-52, models/adamvs.py:418-420);
conv1 = SliceCostRegNetRED.conv1 on the aggregated similarity (models/adamvs.py:417); dec = upconv1 / upconv2d (:421-424);
reg = CostRegNet2D (models/adamvs.py:229-238).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ada_mvs_amd  # noqa: E402,F401
from ada_mvs_amd import synth  # noqa: E402
from oracle import adamvs_oracle as O  # noqa: E402


def _pair(t, dt):
    hi = t.to(dt).float()
    return hi + (t - hi).to(dt).float()


def _fmt(name):
    """-> (round_x, round_w)."""
    one = {"f16": torch.float16, "bf16": torch.bfloat16}
    if name == "fp32":
        return (lambda t: t), (lambda t: t)
    if name in one:
        dt = one[name]
        return (lambda t: t.to(dt).float()), (lambda t: t.to(dt).float())
    base, kind = name.rsplit("_", 1) if "_" in name else (name[:-2], "x3")
    dt = one[base]
    single, pair = (lambda t: t.to(dt).float()), (lambda t: _pair(t, dt))
    if kind == "ws":
        return single, pair
    if kind == "xs":
        return pair, single
    if kind == "x3":
        return pair, pair
    raise ValueError(name)


class Rounded:
    """Patches the oracle's network parts with operand-rounded twins for the duration of a `with` block."""

    def __init__(self, parts):
        self.parts = parts                      # {"gru": fmt, "conv1": fmt, "dec": fmt, "reg": fmt}; "gru@0": stage 1 only
        self.stats = {}
        self.stage = 0

    def conv(self, part, x, w, b=None, stride=1, transposed=False):
        # "gru@0": the part at stage 1 only (the stage index comes from the weight prefix of the step in flight)
        fmt = self.parts.get("%s@%d" % (part, self.stage), self.parts.get(part, "fp32"))
        qx, qw = _fmt(fmt)
        if fmt.startswith("f16"):       # range check: fp16 normal range 6.1e-5 .. 65504
            a = x.abs()
            st = self.stats.setdefault(part, [0.0, 0, 0])
            st[0] = max(st[0], float(a.max()))
            st[1] += int(((a > 0) & (a < 6.1e-5)).sum())
            st[2] += a.numel()
        if transposed:
            return F.conv_transpose2d(qx(x), qw(w), b, stride=2, padding=1, output_padding=1)
        return F.conv2d(qx(x), qw(w), b, stride, 1)

    def gru(self, x, h, sd, pre):
        hc = h.shape[1]
        g = self.conv("gru", torch.cat((x, h), 1), sd[pre + "conv_gates.0.weight"], sd[pre + "conv_gates.0.bias"])
        r, u = torch.sigmoid(g[:, :hc]), torch.sigmoid(g[:, hc:])
        c = torch.tanh(self.conv("gru", torch.cat((x, r * h), 1), sd[pre + "convc.0.weight"], sd[pre + "convc.0.bias"]))
        return u * h + (1 - u) * c

    def step(self, cost, s1, s2, sd, pre, in_up):
        self.stage = int(pre.split(".")[1])
        c1 = F.relu(self.conv("conv1", cost, sd[pre + "conv1.conv.weight"]))
        s1 = self.gru(c1, s1, sd, pre + "conv_gru1.")
        c2 = F.relu(self.conv("gru", s1, sd[pre + "conv2.conv.weight"], None, 2))
        s2 = self.gru(c2, s2, sd, pre + "conv_gru2.")
        up1 = self.conv("dec", s2, sd[pre + "upconv1.weight"], sd[pre + "upconv1.bias"], transposed=True)
        s = F.relu(up1 + s1)
        if in_up:
            reg = self.conv("dec", s, sd[pre + "upconv2d.weight"], sd[pre + "upconv2d.bias"], transposed=True)
        else:
            reg = self.conv("dec", s, sd[pre + "upconv2d.weight"], sd[pre + "upconv2d.bias"])
        return reg, s1, s2

    def reg2d(self, x, sd, pre):
        self.stage = int(pre.split(".")[1])

        def cbr(x, p, stride=1):
            return F.relu(O._bn(self.conv("reg", x, sd[p + "conv.weight"], None, stride), sd, p + "bn."))

        def ctbr(x, p):
            return F.relu(O._bn(self.conv("reg", x, sd[p + "0.weight"], transposed=True), sd, p + "1."))
        conv0 = cbr(x, pre + "conv0.")
        conv2 = cbr(cbr(conv0, pre + "conv1.", 2), pre + "conv2.")
        conv4 = cbr(cbr(conv2, pre + "conv3.", 2), pre + "conv4.")
        y = cbr(cbr(conv4, pre + "conv5.", 2), pre + "conv6.")
        y = conv4 + ctbr(y, pre + "conv7.")
        y = conv2 + ctbr(y, pre + "conv9.")
        y = conv0 + ctbr(y, pre + "conv11.")
        return self.conv("reg", y, sd[pre + "prob.weight"], sd[pre + "prob.bias"])

    def __enter__(self):
        self.saved = (O.slice_reg_step, O.cost_reg_net_2d)
        O.slice_reg_step = self.step
        if "reg" in self.parts:
            O.cost_reg_net_2d = self.reg2d
        return self

    def __exit__(self, *a):
        O.slice_reg_step, O.cost_reg_net_2d = self.saved


def rel_l1(x, ref):
    m = torch.isfinite(ref) & torch.isfinite(x)
    return float((x[m] - ref[m]).abs().mean() / ref[m].abs().mean())


def run(cfg, recipe, parts, features, inputs, sd):
    c = synth.CONFIGS[cfg]
    imgs, proj, dv = inputs
    with torch.no_grad(), O.use_grid_sample(), Rounded(parts) as R:
        out = O.infer_adamvs_forward(imgs, proj, dv, sd, c["num_depth"], c["ndepths"],
                                     synth.DEPTH_INTERVALS_RATIO[:len(c["ndepths"])], features=features)
    return out, R.stats


# BN of CostRegNet2D folds into the weights on the device (packing.py); the rounding of the FOLDED weights is what the kernel
# would see.  The study rounds the unfolded weights: the fold is a per-output-channel scale of 0.4 - 1.7, which moves a
# weight's rounding error by less than one bit -- the table's conclusions are about factors of 8 (fp16 against bf16).

CASES = [
    # (label, parts)
    ("gru:f16", {"gru": "f16"}),
    ("gru:f16_ws", {"gru": "f16_ws"}),
    ("gru:f16_xs", {"gru": "f16_xs"}),
    ("gru:f16x3", {"gru": "f16x3"}),
    ("gru:bf16", {"gru": "bf16"}),
    ("gru:bf16_ws", {"gru": "bf16_ws"}),
    ("gru:bf16x3", {"gru": "bf16x3"}),
    ("conv1:f16", {"conv1": "f16"}),
    ("conv1:f16_ws", {"conv1": "f16_ws"}),
    ("conv1:bf16_ws", {"conv1": "bf16_ws"}),
    ("dec:f16", {"dec": "f16"}),
    ("dec:f16_ws", {"dec": "f16_ws"}),
    ("reg:f16", {"reg": "f16"}),
    ("reg:f16_ws", {"reg": "f16_ws"}),
    ("reg:f16_xs", {"reg": "f16_xs"}),
    ("reg:bf16_ws", {"reg": "bf16_ws"}),
    ("gru+conv1:f16", {"gru": "f16", "conv1": "f16"}),
    ("gru+conv1:f16_ws", {"gru": "f16_ws", "conv1": "f16_ws"}),
    ("gru+conv1+dec:f16_ws", {"gru": "f16_ws", "conv1": "f16_ws", "dec": "f16_ws"}),
    ("gru+conv1+reg:f16_ws", {"gru": "f16_ws", "conv1": "f16_ws", "reg": "f16_ws"}),
    ("all:f16_ws", {"gru": "f16_ws", "conv1": "f16_ws", "dec": "f16_ws", "reg": "f16_ws"}),
    ("all:f16", {"gru": "f16", "conv1": "f16", "dec": "f16", "reg": "f16"}),
    # one stage only: where does the last confidence map's error come from?
    ("gru@s1:f16", {"gru@0": "f16"}),
    ("gru@s1:f16_ws", {"gru@0": "f16_ws"}),
    ("gru@s2:f16_ws", {"gru@1": "f16_ws"}),
    ("gru@s3:f16_ws", {"gru@2": "f16_ws"}),
    ("gru+conv1+dec@s1:f16_ws", {"gru@0": "f16_ws", "conv1@0": "f16_ws", "dec@0": "f16_ws"}),
    ("gru+conv1+dec@s1:f16", {"gru@0": "f16", "conv1@0": "f16", "dec@0": "f16"}),
    ("gru+conv1+dec@s1:bf16", {"gru@0": "bf16", "conv1@0": "bf16", "dec@0": "bf16"}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="cfg1")
    ap.add_argument("--recipes", default="default,sharp")
    ap.add_argument("--cases", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet
    c = synth.CONFIGS[a.cfg]
    nst = len(c["ndepths"])
    cases = [x for x in CASES if not a.cases or x[0] in a.cases.split(",")]
    for recipe in a.recipes.split(","):
        m = Infer_AdaMVSNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO[:nst], False, [8, 8, 8])
        sd = synth.seeded_state_dict(m, seed=0, recipe=recipe)
        inputs = synth.tile_inputs(a.cfg, 1, seed=0)
        with torch.no_grad():
            feats = [O.feature_net(inputs[0][:, v], sd) for v in range(inputs[0].shape[1])]
        t0 = time.time()
        ref, _ = run(a.cfg, recipe, {}, feats, inputs, sd)
        print("# %s / %s: fp32 oracle %.1f s" % (a.cfg, recipe, time.time() - t0), flush=True)
        for label, parts in cases:
            out, stats = run(a.cfg, recipe, parts, feats, inputs, sd)
            row = {"cfg": a.cfg, "recipe": recipe, "case": label,
                   "depth": rel_l1(out["depth"], ref["depth"]),
                   "confidence": rel_l1(out["photometric_confidence"], ref["photometric_confidence"])}
            for s in range(nst):
                st, rs = out["stage%d" % (s + 1)], ref["stage%d" % (s + 1)]
                row["s%d" % (s + 1)] = [rel_l1(st["depth"], rs["depth"]),
                                        rel_l1(st["photometric_confidence"], rs["photometric_confidence"])]
            row["pair_conf"] = max(rel_l1(x, y) for x, y in zip(out["stage1"]["pair_confidence"], ref["stage1"]["pair_confidence"]))
            row["f16_range"] = {k: {"max_abs": v[0], "below_normal_frac": v[1] / max(v[2], 1)} for k, v in stats.items()}
            print(json.dumps(row), flush=True)
            if a.out:
                with open(os.path.join(ROOT, a.out), "a") as f:
                    f.write(json.dumps(row) + "\n")


if __name__ == "__main__":
    main()
