#!/usr/bin/env python3
"""Throughput of the Ada-MVS depth-inference hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg1|tiny] [--batch B]

A "step" = one pass of the hot path (reference models/adamvs.py:586-618 minus
FeatureNet0: hypothesis sampling, plane sweep, per-view weighting, recurrent
regularisation, soft-argmin) over a batch of B synthetic reference tiles per
GPU, feature maps already resident in HBM.  N > 1: one process per GPU
(torchrun), tiles sharded over ranks, one RCCL gather of the finished maps per
step.  Rank 0 prints ONE JSON line (contract: see the task description).

Also measured by rank 0 at N = 1 (after the timed region, same shapes):
  roofline      per-kernel HIP-event timing of the stage run phase by phase through the C ABI
  cpu_baseline  the CPU oracle ("port") on one tile of the same workload
"""
import argparse
import json
import os
import sys
import time


def _argv_value(name, default):
    """The value of `--name V` / `--name=V` on the command line, before argparse (and torch) exist."""
    v, argv = default, sys.argv[1:]
    for i, a in enumerate(argv):
        if a == name and i + 1 < len(argv):
            v = argv[i + 1]
        elif a.startswith(name + "="):
            v = a.split("=", 1)[1]
    return v


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): this process becomes the launcher.
    It has imported nothing that can touch the GPU (the check runs before `import torch`), starts N FRESH child processes
    of this file -- one rank per GPU, torchrun's environment contract, never an exec -- relays rank 0's JSON line and
    exits non-zero if any rank does.  A rank that hangs (a collective that never completes) trips `--launch-timeout`
    (default 900 s): every rank is ended by PID, the last stderr lines of every rank are relayed and the exit code is 124
    (ada-mvs_amd/launch.py).  Under torchrun (WORLD_SIZE set) this is a no-op."""
    if os.environ.get("WORLD_SIZE"):
        return
    try:
        n = int(_argv_value("--gpus", 1))
        timeout = float(_argv_value("--launch-timeout", 900.0))
    except ValueError:
        return                                   # argparse reports it
    if n <= 1:
        return
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "adamvs_launch", os.path.join(os.path.dirname(os.path.abspath(__file__)), "ada-mvs_amd", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)              # imports no torch
    code, _ = launch.run_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], n, timeout=timeout,
                               extra_env={"ADAMVS_BENCH_LAUNCHER": "self"})
    sys.exit(code)


if __name__ == "__main__":
    _self_launch()

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import ada_mvs_amd  # noqa: E402
from ada_mvs_amd import _lib, dist as adist, hip_ops, synth  # noqa: E402
from ada_mvs_amd.models.adamvs import Infer_AdaMVSNet  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # v_mfma_f32_16x16x32_bf16, dense (MI355X_MICROARCH.md)


def build_model(cfg, device, precision="fp32"):
    c = synth.CONFIGS[cfg]
    m = Infer_AdaMVSNet(c["num_depth"], c["ndepths"], synth.DEPTH_INTERVALS_RATIO[:len(c["ndepths"])], False, [8, 8, 8],
                        precision=precision)
    sd = synth.seeded_state_dict(m, seed=0)
    m.load_state_dict(sd)
    return m.to(device).eval(), sd


def algorithmic_work(cfg, B):
    """bytes and conv flops per tile, fp32 (e = 4), per phase of every stage."""
    c = synth.CONFIGS[cfg]
    S, e = c["views"] - 1, 4
    out = []
    for s, D in enumerate(c["ndepths"]):
        C = (32, 16, 8)[s]
        h, w = c["H"] // (4, 2, 1)[s], c["W"] // (4, 2, 1)[s]
        hw = h * w
        HoWo = hw * (4 if s < 2 else 1)
        st = {"stage": s + 1, "C": C, "h": h, "w": w, "D": D}
        st["aggregate_bytes"] = B * D * ((S + 1) * C * hw * e + S * hw * 4 + hw * 4)
        st["recurrence_bytes"] = B * D * (96 * hw)
        st["softargmin_bytes"] = B * D * (24 * HoWo)
        st["recurrence_flops"] = B * D * hw * 2 * 7560
        # the GRU convolutions that may run in the F(2x2, 3x3) form (csrc/slice_roles_wino.h), MACs per level-1 pixel of their
        # direct form: gates1 16->16, gates2 32->32 and cand2 32->16 at quarter resolution, cand1 16->8
        st["gru_conv_macs"] = {1: 2304, 2: 2304, 4: 1152, 8: 1152}
        st["conv1_flops"] = B * D * hw * 2 * 72 * C
        # round 5: conv1 runs in the two-row form with F(2, 3) along x: 16 products per two outputs of a row and channel pair where the
        # direct form has 18 (csrc/slice_red.hip::k_conv1_f23; option conv1_f23 bit 1: C = 32, bit 2: C = 16 / 8)
        st["conv1_f23_bit"] = 1 if C == 32 else 2
        if s == 0:
            st["pair_similarity_bytes"] = B * (S * D * C * hw * e + S * C * hw * e + S * D * hw * 4)
            st["costreg_bytes"] = B * 2 * S * D * hw * 4
            st["softmax_bytes"] = B * (S * D * hw * 4 + 2 * S * hw * 4)
            macs = 0
            for res, n in ((1, 3), (4, 3), (16, 3), (64, 2)):         # layers per resolution (conv0,11,prob | 1,2,9 | 3,4,7 | 5,6)
                macs += n * (hw // res) * D * D * 9
            # transposed layers touch 2.25 taps on average instead of 9
            macs -= (hw + hw // 4 + hw // 16) * D * D * 9 * (1 - 2.25 / 9)
            st["costreg_flops"] = B * S * 2 * macs
            # the stride-1 layers conv0, conv2, conv4, conv6, prob (what the F(2x2, 3x3) kernel executes 16/36 of)
            st["costreg_stride1_flops"] = B * S * 2 * (2 * hw + hw // 4 + hw // 16 + hw // 64) * D * D * 9
        out.append(st)
    return out


def timed_phases(model, feats_cl, shapes, proj, dv, interval, steps):
    """Run the cascade phase by phase through the C ABI with HIP events (torch.cuda.Event records on the
    stream the kernels are launched on) around every phase.  -> {phase name: [ms per call]}"""
    from collections import defaultdict
    times = defaultdict(list)
    evs = []

    def mark(name, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        evs.append((name, a, b))
        return r

    for _ in range(steps):
        depth, conf = None, None
        for s in range(model.num_stage):
            name = "stage%d" % (s + 1)
            B, C, h, w = shapes[s]
            net = model.DepthNet[s]
            cur = dv if depth is None else depth
            mode, half_span, planes = hip_ops.plane_source(cur, model.ndepths[s], model.depth_intervals_ratio[s] * interval, [B, h, w])
            rt = hip_ops.relative_transforms(proj[name])
            S = feats_cl[s].shape[0] // B - 1
            D = model.ndepths[s]
            first = conf is None
            desc = hip_ops.stage_desc(B, S, C, h, w, D, net.in_up, first, (0, 0) if first else tuple(conf.shape[-2:]),
                                      _lib.PRECISIONS[net.reg.effective_precision()], _lib.PRECISIONS[net.reg_fuse.precision],
                                      plane_mode=mode, half_span=half_span)
            dev = feats_cl[s].device
            Ho, Wo = (2 * h, 2 * w) if net.in_up else (h, w)
            outs = (torch.empty(S, B, h, w, device=dev), torch.empty(S, B, h, w, device=dev) if first else None,
                    torch.empty(B, Ho, Wo, device=dev), torch.empty(B, Ho, Wo, device=dev))
            w_reg = net.reg.packed(dev) if first else None
            fuse = net.reg_fuse.packed(dev)
            ws = model._stage_workspace[(dev, 0)]

            def phase(mask, timing_only=False):
                return hip_ops.depth_stage_forward(desc, feats_cl[s], rt, planes, conf, w_reg, fuse, ws, phases=mask, outputs=outs,
                                                   timing_only=timing_only)

            if first:
                # pass A split further, op by op; CostRegNet2D layer by layer so that every launch of its
                # convolution kernel is timed by itself (the roofline object is per launch of one kernel)
                # the op-level entry points of pass A take a planes tensor; inside the stage call they are generated
                planes_t = hip_ops.depth_range_samples(cur, D, 0.0, [B, h, w])
                sim = mark("s%d.pair_similarity" % (s + 1), lambda: hip_ops.pair_similarity(feats_cl[s], rt, planes_t, B, S, C, D, h, w))
                prec = _lib.PRECISIONS[net.reg.effective_precision()]
                # softmax / max / regression run in the epilogue of the last layer (csrc/costreg_softmax.h), as in the stage
                fused_sm = _lib.get_option("fuse_softmax") != 0
                score = timed_cost_reg_layers(mark, s + 1, sim, w_reg, S * B, D, h, w, prec,
                                              softmax=(planes_t, S, B, cur) if fused_sm else None)
                vw_pd = score if fused_sm else mark("s%d.softmax_max_regress" % (s + 1),
                                                    lambda: hip_ops.softmax_max_regress(score, planes_t, S, B, D, h, w))
                outs[0].copy_(vw_pd[0])
                del sim, score, planes_t
            else:
                mark("s%d.view_weight_resample" % (s + 1), lambda: phase(_lib.PHASE_VIEW_WEIGHTS))
            # the three phases below are interleaved chunk by chunk in a real run; called one by one each runs alone over
            # all chunks (its own duration, no maps), so the maps that feed the next stage come from one untimed full call
            mark("s%d.aggregate_conv1" % (s + 1), lambda: phase(_lib.PHASE_AGGREGATE, True))
            mark("s%d.recurrence" % (s + 1), lambda: phase(_lib.PHASE_RECURRENCE, True))
            mark("s%d.soft_argmin" % (s + 1), lambda: phase(_lib.PHASE_SOFT_ARGMIN, True))
            phase(_lib.PHASE_AGGREGATE | _lib.PHASE_RECURRENCE | _lib.PHASE_SOFT_ARGMIN)
            depth, conf = outs[2], outs[0]
    torch.cuda.synchronize()
    for name, a, b in evs:
        times[name].append(a.elapsed_time(b))
    return times


COSTREG_PLAN = (  # (layer, mode 0 s1 / 1 s2 / 2 transposed, relu, input, skip), CostRegNet2D.forward (adamvs.py:229-238)
    ("conv0", 0, 1, "x", None), ("conv1", 1, 1, "conv0", None), ("conv2", 0, 1, "conv1", None), ("conv3", 1, 1, "conv2", None),
    ("conv4", 0, 1, "conv3", None), ("conv5", 1, 1, "conv4", None), ("conv6", 0, 1, "conv5", None),
    ("conv7", 2, 1, "conv6", "conv4"), ("conv9", 2, 1, "conv7", "conv2"), ("conv11", 2, 1, "conv9", "conv0"),
    ("prob", 0, 0, "conv11", None))


WINO_SLOT = {"conv0": 0, "conv2": 1, "conv4": 2, "conv6": 3, "prob": 4}


def winograd_active(D, precision):
    """csrc/costreg2d.hip::cost_reg_winograd: fp32, a supported width, not switched off."""
    from ada_mvs_amd import packing
    return precision in (0, "fp32") and D in packing.WINO_WIDTHS and _lib.get_option("winograd") != 0


def timed_cost_reg_layers(mark, stage, x, wpk, N, D, h, w, precision=0, softmax=None):
    """CostRegNet2D through the one-layer C-ABI op, one timing mark per launch.  softmax = (planes, S, B): the last layer runs
    with the softmax / max / regression epilogue and (view_weight, pair_depth) is returned instead of the scores."""
    LW = 9 * D * D + D
    acts = {"x": (x, h, w)}
    # fp32: conv7's and conv9's skip additions run in the CONSUMING transposed layer (in2), as adamvs_cost_reg_net_2d issues
    # them (csrc/costreg2d.hip); conv11's, and all of them in bf16x3, in the producing layer's epilogue (skip)
    defer = precision == 0 and _lib.get_option("costreg_defer_skips") != 0
    wino = winograd_active(D, precision)
    pending = None                       # the addend the next layer has to add to its input
    for i, (name, mode, relu, src, skip) in enumerate(COSTREG_PLAN):
        xin, hi, wi = acts[src]
        wl = wpk[i * LW:(i + 1) * LW]
        sk = acts[skip][0] if skip else None
        give = defer and i + 1 < len(COSTREG_PLAN) and COSTREG_PLAN[i + 1][1] == 2      # only a transposed consumer takes the addend
        in2, pending = pending, (sk if give else None)
        if mode == 0 and wino:          # the stride-1 layers in the F(2x2, 3x3) form, as adamvs_cost_reg_net_2d issues them
            ww = wpk[len(COSTREG_PLAN) * LW:][WINO_SLOT[name] * 16 * D * D:(WINO_SLOT[name] + 1) * 16 * D * D]
            if name == "prob" and softmax is not None:
                planes_t, S, B, dv = softmax
                if _lib.get_option("wino_softmax") != 0 and dv.dim() == 2:
                    # what the step runs: the SM instantiation (per-lane softmax partials in the epilogue, no score volume)
                    # and k_softmax_merge behind it, timed together
                    return mark("s%d.costreg.prob+softmax.mode0" % stage,
                                lambda: hip_ops.prob_softmax_regress_wino(xin, ww, wl[9 * D * D:], dv, S, B, D, hi, wi))
                score = mark("s%d.costreg.prob.mode0" % stage, lambda: hip_ops.conv3x3_dd_wino(xin, ww, wl[9 * D * D:], None, N, D, hi, wi, relu))
                return mark("s%d.costreg.softmax.launch" % stage, lambda: hip_ops.softmax_max_regress(score, planes_t, S, B, D, hi, wi))
            acts[name] = (mark("s%d.costreg.%s.mode0" % (stage, name),
                               lambda: hip_ops.conv3x3_dd_wino(xin, ww, wl[9 * D * D:], None, N, D, hi, wi, relu)), hi, wi)
            continue
        if name == "prob" and softmax is not None:
            planes_t, S, B, _ = softmax
            return mark("s%d.costreg.prob+softmax.mode0" % stage,
                        lambda: hip_ops.prob_softmax_regress(xin, wl, wl[9 * D * D:], planes_t, S, B, D, hi, wi, precision=precision))
        out = mark("s%d.costreg.%s.mode%d" % (stage, name, mode),
                   lambda: hip_ops.conv3x3_dd(xin, wl, wl[9 * D * D:], None if give else sk, N, D, hi, wi, mode, relu,
                                              precision=precision, in2=in2))
        ho, wo = (hi // 2, wi // 2) if mode == 1 else ((2 * hi, 2 * wi) if mode == 2 else (hi, wi))
        acts[name] = (out, ho, wo)
    return acts["prob"][0]


def cpu_baseline(cfg, sd, baseline=8.0):
    """The CPU oracle (a port of the reference's unfused PyTorch-CPU path) on ONE tile of the workload."""
    from oracle import adamvs_oracle as O          # checker / baseline only
    c = synth.CONFIGS[cfg]
    imgs, proj, dv = synth.tile_inputs(cfg, batch=1, seed=0, baseline=baseline)
    sd_cpu = {k: v.detach().cpu() for k, v in sd.items()}
    threads = min(os.cpu_count() or 1, 32)          # more threads than that only add scheduling overhead to the small ops
    torch.set_num_threads(threads)
    with torch.no_grad(), O.use_grid_sample():
        feats = [O.feature_net(imgs[:, v], sd_cpu) for v in range(c["views"])]
        t0 = time.time()
        ref = O.infer_adamvs_forward(imgs, proj, dv, sd_cpu, c["num_depth"], c["ndepths"],
                                     synth.DEPTH_INTERVALS_RATIO[:len(c["ndepths"])], features=feats)
        dt = time.time() - t0
    return {"value": 1.0 / dt, "unit": "depth maps/s", "cores": threads, "kind": "port",
            "sample": "1 tile of %s, hot path only (features precomputed), %.1f s, oracle/adamvs_oracle.py "
                      "(grid_sample form of the warp, as the reference issues it)" % (cfg, dt)}, ref


def rel_l1(x, ref):
    """mean|x - ref| / mean|ref|."""
    x, ref = x.detach().double().cpu(), ref.detach().double().cpu()
    return float((x - ref).abs().mean() / ref.abs().mean().clamp_min(1e-30))


def source_stamp():
    """sha256 over the kernel sources and the C ABI header: profiles/*_traffic.json carry the stamp of the build they were
    measured on, and a stale file is refused instead of being quoted."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "ada-mvs_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "adamvs_hip.h"), "rb").read())
    return h.hexdigest()[:16]


# ---- the sibling model: not the headline, selected with --model msrednet ---------------
def msrednet_flops(views, H, W, ndepths):
    """Convolution flops of one map through slice_RED_Regularization (reference models/msrednet.py:330-366), real channels."""
    total = 0.0
    for s, (scale, C) in enumerate(((4, 32), (2, 16), (1, 8))):
        hw = (H // scale) * (W // scale)
        x, hc = (C, 16, 32, 64), (8, 16, 32, 64)
        per_plane = 0.0
        for k in range(4):
            px = hw / 4 ** k
            per_plane += 2 * 9 * (x[k] + hc[k]) * 3 * hc[k] * px                 # gate_conv (2 hc rows) + output_conv
            if k < 3:
                per_plane += 2 * 9 * x[k] * x[k + 1] * px / 4                     # conv_{k+1}, stride 2
                per_plane += 2 * 9 * hc[k + 1] * hc[k] * px / 4                   # upconv_{k+1}, per input position
        per_plane += 2 * 9 * 8 * 1 * hw                                           # upconv2d
        total += ndepths[s] * per_plane
    return total


def bench_msrednet(args):
    """One JSON line for Infer_CascadeREDNet on --red-batch synthetic tiles per step (same line format, single GPU)."""
    from ada_mvs_amd.models.msrednet import Infer_CascadeREDNet
    c = synth.CONFIGS[args.workload]
    nd = c["ndepths"] if len(c["ndepths"]) == 3 else [48, 32, 8]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    m = Infer_CascadeREDNet(c["num_depth"], nd, synth.DEPTH_INTERVALS_RATIO, False, [8, 8, 8])
    sd = synth.seeded_state_dict(m, seed=0)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    B = args.red_batch
    imgs, proj, dv = synth.tile_inputs(dict(c, ndepths=nd), batch=B, seed=0)
    imgs, dv = imgs.to(dev), dv.to(dev)
    proj = {k: v.to(dev) for k, v in proj.items()}
    interval = (synth.DEPTH_RANGE[1] - synth.DEPTH_RANGE[0]) / c["num_depth"]
    with torch.no_grad():
        maps, shapes = m.extract_features(imgs)
        m.infer_from_features(maps, shapes, proj, dv, interval)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m.infer_from_features(maps, shapes, proj, dv, interval)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g):
            m.infer_from_features(maps, shapes, proj, dv, interval)
        for _ in range(args.warmup):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
    flops = B * msrednet_flops(c["views"], c["H"], c["W"], nd)
    ach = flops / dt / 1e12
    result = {"metric": "depth maps/s", "value": B / dt, "unit": "depth maps/s", "n_gpus": 1, "steps": args.steps,
              "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "dtype": "f32", "data": "synthetic",
              "config": {"workload": "msrednet (Infer_CascadeREDNet), %s shape: %d views, %dx%d, hypotheses %s, fp32" % (
                  args.workload, c["views"], c["W"], c["H"], "/".join(map(str, nd))), "tiles_per_gpu_per_step": B,
                  "launch": "hipGraph replay, four level recurrences as parallel branches"},
              "roofline": {"kernel": "all convolutions of slice_RED_Regularization (k_conv_dd / k_conv_dd_resident)",
                           "bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                           "note": "%d planes x 4 levels x 7 dependent launches on maps down to %dx%d: latency-bound, see DESIGN.md"
                                   % (sum(nd), c["H"] // 32, c["W"] // 32)}}
    if not args.no_cpu_baseline:
        from oracle import msrednet_oracle as MO       # checker / baseline only
        sd_cpu = {k: v.detach().cpu() for k, v in sd.items()}
        threads = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(threads)
        ci, cp, cd = synth.tile_inputs(dict(c, ndepths=nd), batch=1, seed=0)
        with torch.no_grad(), MO.ao.use_grid_sample():
            t0 = time.time()
            MO.infer_cascade_rednet_forward(ci, cp, cd, sd_cpu, c["num_depth"], nd, synth.DEPTH_INTERVALS_RATIO)
            cdt = time.time() - t0
        result["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "depth maps/s", "cores": threads, "kind": "port",
                                  "sample": "1 tile, images to maps, %.1f s, oracle/msrednet_oracle.py" % cdt}
    print(json.dumps(result))


class Workload:
    """One configuration of the hot path set up for timing on this rank: the model, this rank's tiles (images seeded by
    their GLOBAL tile index; rig b of the batch for slot b), FeatureNet0 outputs resident in HBM (upstream
    of the timed region, timed separately) and the captured hipGraph of one step."""

    def __init__(self, cfg, tiles, precision, dev, groups=1, use_graph=True, baseline=8.0, warm=1):
        self.cfg, self.tiles, self.precision, self.dev, self.G, self.baseline = cfg, list(tiles), precision, dev, groups, baseline
        self.c = synth.CONFIGS[cfg]
        self.B = B = len(self.tiles)
        assert B > 0 and B % groups == 0, "tiles per GPU must be a positive multiple of --groups"
        self.Bg = Bg = B // groups
        self.model, self.sd = build_model(cfg, dev, precision)
        imgs = torch.cat([synth.tile_inputs(cfg, 1, seed=t)[0] for t in self.tiles], 0).to(dev)
        _, proj, dv = synth.tile_inputs(cfg, batch=B, seed=0, baseline=baseline)
        proj = {k: v.to(dev) for k, v in proj.items()}
        dv = dv.to(dev)
        self.interval = (synth.DEPTH_RANGE[1] - synth.DEPTH_RANGE[0]) / self.c["num_depth"]
        G = groups
        with torch.no_grad():
            self.groups = [self.model.extract_features(imgs[g * Bg:(g + 1) * Bg]) for g in range(G)]
            torch.cuda.synchronize()
            self.t_feat = float("inf")
            for _ in range(3):               # steady state: the caching allocator re-uses the workspace of the previous call
                self.groups = None
                t0 = time.time()
                self.groups = [self.model.extract_features(imgs[g * Bg:(g + 1) * Bg]) for g in range(G)]
                torch.cuda.synchronize()
                self.t_feat = min(self.t_feat, time.time() - t0)
            del imgs
            self.projs = [{k: v[g * Bg:(g + 1) * Bg].contiguous() for k, v in proj.items()} for g in range(G)]
            self.dvs = [dv[g * Bg:(g + 1) * Bg].contiguous() for g in range(G)]
            self.side = [torch.cuda.Stream() for _ in range(G - 1)]
            for _ in range(max(warm, 1)):
                self.depth, self.conf = self.hot_path()
            torch.cuda.synchronize()
            self.graph = None
            if use_graph:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self.depth, self.conf = self.hot_path()
                self.graph.replay()
                torch.cuda.synchronize()

    def hot_path(self):
        """All groups, each on its own stream, forked from / joined to the current stream."""
        cur = torch.cuda.current_stream()
        G = self.G
        outs = [None] * G
        for g in range(G):
            st = cur if g == 0 else self.side[g - 1]
            if g:
                st.wait_stream(cur)
            with torch.cuda.stream(st):
                o = self.model.infer_from_features(self.groups[g][0], self.groups[g][1], self.projs[g], self.dvs[g], self.interval, group=g)
                outs[g] = (o["depth"], o["photometric_confidence"])
        for g in range(1, G):
            cur.wait_stream(self.side[g - 1])
        if G == 1:
            return outs[0]
        return torch.cat([o[0] for o in outs], 0), torch.cat([o[1] for o in outs], 0)

    def step(self):
        """One pass of the hot path over this rank's tiles -> (depth, confidence) [B,Ho,Wo] (the graph's output buffers)."""
        if self.graph is not None:
            self.graph.replay()
            return self.depth, self.conf
        return self.hot_path()

    def step_fracs(self, ms_per_step):
        """algorithmic conv flops and bytes of ALL phases / the measured step, against the bounding roofline
        (fp32: the fp32 matrix pipe; split-bf16: HBM -- it executes three bf16 products per fp32 product on the bf16 pipe)."""
        work = algorithmic_work(self.cfg, self.B)
        split = self.precision == "bf16x3"
        step_s = 1e-3 * ms_per_step
        tot_flops = sum(w_.get("costreg_flops", 0) + w_["recurrence_flops"] + w_["conv1_flops"] for w_ in work)
        tot_bytes = sum(sum(v for k_, v in w_.items() if k_.endswith("_bytes")) for w_ in work)
        # EXECUTED flops: the stride-1 layers of CostRegNet2D in the F(2x2, 3x3) form issue 16 of the 36 products SURVEY 8d's
        # direct form counts; the split-bf16 mode issues three bf16 products per fp32 product
        wino = winograd_active(work[0]["D"], self.precision)
        executed = tot_flops - (work[0].get("costreg_stride1_flops", 0) * (20.0 / 36.0) if wino else 0.0)
        gru_wino = 0.0
        if not split:           # fp32: stages whose recurrence runs one role per launch take their GRU convolutions in the F(2x2, 3x3) form
            lib = _lib.load()
            mask = lib.adamvs_gru_wino_mask()
            for w_ in work:
                sched = lib.adamvs_recurrence_schedule(0, self.B * w_["h"] * w_["w"])
                if sched == 0 or (sched == 1 and (mask & 7) == 7):
                    eff = mask if sched == 0 else 7          # the three-launch schedule never takes cand1 (bit 8) in that form
                    gru_wino += self.B * w_["D"] * w_["h"] * w_["w"] * 2.0 * sum(m for b_, m in w_["gru_conv_macs"].items() if eff & b_)
        executed -= gru_wino * (20.0 / 36.0)
        if not split:
            f23 = _lib.get_option("conv1_f23")
            executed -= sum(w_["conv1_flops"] for w_ in work if f23 & w_["conv1_f23_bit"]) * (2.0 / 18.0)
            if _lib.get_option("s2_pairs") != 0 and _lib.get_option("conv_rows2") != 1:
                # round 5: the stride-2 layers of CostRegNet2D whose output rows divide into 32-column blocks run in the pair form (15 of
                # 18 products; csrc/costreg2d.hip::k_conv_dd_s2p, D = 192 / 384): conv1, conv3, conv5 have w/2, w/4, w/8 output columns
                # (the launcher sends small grids -- at most 2048 blocks of 8 x 16 outputs -- to the 2-row direct kernel instead)
                pairs = 0.0
                w0 = work[0]
                if "costreg_flops" in w0:
                    S_ = synth.CONFIGS[self.cfg]["views"] - 1
                    for res in (4, 16, 64):
                        wo, ho = w0["w"] // int(res ** 0.5), w0["h"] // int(res ** 0.5)
                        blocks8 = -(-wo // 16) * -(-ho // 8) * S_ * self.B
                        if (wo % 32 == 0 or wo >= 256) and blocks8 > 2048 and w0["D"] in (192, 384):
                            pairs += self.B * S_ * 2.0 * (w0["h"] * w0["w"] // res) * w0["D"] ** 2 * 9
                executed -= pairs * (3.0 / 18.0)
        f_mfma = (tot_flops * 3 / step_s / 1e12 / BF16_MFMA_PEAK_TFLOPS if split else executed / step_s / 1e12 / FP32_MFMA_PEAK_TFLOPS)
        f_hbm = tot_bytes / step_s / 1e9 / HBM_PEAK_GBS
        out = {"step_frac_mfma": f_mfma, "step_frac_hbm": f_hbm, "step_frac": f_hbm if split else f_mfma,
               "step_bound": "hbm" if split else "mfma (fp32)",
               "step_note": "step_frac = utilisation of the bounding roof by the whole step: algorithmic bytes / step / HBM peak "
                            "(bf16x3), EXECUTED conv flops / step / fp32 MFMA peak (fp32)",
               "step_algorithmic": {"conv_gflop_per_tile": tot_flops / self.B / 1e9, "executed_conv_gflop_per_tile": executed / self.B / 1e9,
                                    "gbytes_per_tile": tot_bytes / self.B / 1e9}}
        if wino or gru_wino or not split:
            # the same step priced in SURVEY 8d's direct-form flops: an algorithmic saving, NOT a hardware fraction (can pass 1)
            out["step_frac_direct_equivalent"] = tot_flops / step_s / 1e12 / FP32_MFMA_PEAK_TFLOPS
        return out

    def close(self):
        """Release the graph, the stage workspaces and the features before the next workload is set up."""
        self.graph = None
        self.depth = self.conf = None
        self.groups = self.projs = self.dvs = None
        self.model._stage_workspace.clear()
        self.model = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


def time_plain(wl, steps, warmup):
    """ms per step of a single-rank workload (no gather: with one rank the maps are already where they are wanted)."""
    with torch.no_grad():
        for _ in range(warmup):
            wl.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            wl.step()
        torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


# BASELINE.json configs 3 and 4, measured after the headline's timed region (N = 1 only) and reported under "cascade":
# (key, config, global tile indices of this GPU's batch, precision)
CASCADE_CASES = (
    ("cfg3_b32_bf16x3", "cfg3", list(range(32)), "bf16x3"),
    ("cfg3_b32_fp32", "cfg3", list(range(32)), "fp32"),
    ("cfg4_share_b4_fp32", "cfg3", adist.tiles_of_rank(32, 0, 8), "fp32"),        # 32 tiles over 8 ranks: rank 0 owns 0, 8, 16, 24
    ("cfg4_share_b4_bf16x3", "cfg3", adist.tiles_of_rank(32, 0, 8), "bf16x3"),
    # BASELINE.json configs[4] (9 views, 1536x768, 256/96/16): timing only -- its parity against the oracle (2.5 min of CPU per
    # tile) is held by tests/test_full_size_parity.py on the GPU box
    ("cfg5_b8_fp32", "cfg5", list(range(8)), "fp32"),
    ("cfg5_b8_bf16x3", "cfg5", list(range(8)), "bf16x3"),
)


def bench_cascade(dev, steps, warmup, baseline, use_graph=True):
    """The cascade configurations (BASELINE.json configs[2], configs[3]'s per-GPU share) on one GPU: hipGraph replay, same
    timing brackets as the headline.  Every case starts with global tile 0 on rig 0, so ONE oracle pass of that tile
    (cfg3, full size) checks all of them: parity_rel_l1 per case, 1e-3 enforced by the caller."""
    out, tile0, sd3 = {}, {}, None
    for key, cfg, tiles, precision in CASCADE_CASES:
        c = synth.CONFIGS[cfg]
        wl = Workload(cfg, tiles, precision, dev, use_graph=use_graph, baseline=baseline)
        ms = time_plain(wl, steps if cfg == "cfg3" else min(steps, 5), warmup)
        d, p = wl.step()
        torch.cuda.synchronize()
        if cfg == "cfg3":
            tile0[key] = (d[0].clone(), p[0].clone())
            sd3 = wl.sd
        fr = wl.step_fracs(ms)
        out[key] = {"workload": "%s: %d views, %dx%d, hypotheses %s, %s, %d tiles per step" % (
                        cfg if len(tiles) != 4 else "cfg4 (32 cfg3 tiles over 8 GPUs): one GPU's share", c["views"], c["W"], c["H"],
                        "/".join(map(str, c["ndepths"])), precision, len(tiles)),
                    "maps_per_s": len(tiles) / (ms * 1e-3), "ms_per_step": ms, "ms_per_tile": ms / len(tiles),
                    "step_frac": fr["step_frac"], "step_bound": fr["step_bound"], "steps": steps, "warmup": warmup}
        if "step_frac_direct_equivalent" in fr:
            out[key]["step_frac_direct_equivalent"] = fr["step_frac_direct_equivalent"]
        if cfg != "cfg3":
            out[key]["parity_rel_l1"] = "not checked in this run (the oracle needs minutes per tile): tests/test_full_size_parity.py"
        wl.close()
        del wl
    return out, tile0, sd3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=list(synth.CONFIGS))
    ap.add_argument("--batch", type=int, default=256,
                    help="reference tiles per GPU per step (weak scaling: fixed per GPU).  256 since round 5: ~104 GB of the 288 GB "
                         "(measured 607.8 / 612.4 / 615.5 / 621.6 maps/s at 96 / 128 / 192 / 256 on one box)")
    ap.add_argument("--tiles-total", type=int, default=0,
                    help="strong scaling: this many tiles per step over ALL ranks (tile t on rank t mod N), e.g. 32 with "
                         "--workload cfg3 = BASELINE.json configs[3] as stated; overrides --batch")
    ap.add_argument("--baseline", type=float, default=8.0,
                    help="camera baseline of the synthetic rig per view index (SURVEY.md 8c recipe: 8; larger = more disparity "
                         "per hypothesis plane and out-of-bounds warps)")
    ap.add_argument("--groups", type=int, default=1, help="independent tile groups run concurrently on separate HIP streams")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="fp32: fp32 MFMA, stride-1 CostRegNet2D layers in the F(2x2,3x3) form (the cfg2 headline); bf16x3: split-bf16 MFMA for the convolutions (~1e-5 of fp32)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-cascade", action="store_true",
                    help="skip the cfg3 / cfg4-share measurements that follow the headline (N = 1, default workload only)")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="self-launched N > 1 runs: seconds after which every rank is ended by PID and the launcher exits 124 "
                         "(a rank hanging in a collective); read by the launcher before argparse, listed here for --help")
    ap.add_argument("--red-batch", type=int, default=1, help="--model msrednet: tiles per step")
    ap.add_argument("--model", default="adamvs", choices=["adamvs", "msrednet"],
                    help="adamvs: the headline path; msrednet: the sibling model (SURVEY.md 8f row f3), --red-batch tiles per step, 1 GPU")
    args = ap.parse_args()
    if args.model == "msrednet":
        _lib.load()
        return bench_msrednet(args)

    one_device = bool(os.environ.get("ADAMVS_BENCH_ONE_DEVICE"))   # dry run of the N > 1 path on a 1-GPU box (with ADAMVS_DIST_BACKEND=gloo)
    if one_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = adist.init_from_env()
    if world != args.gpus:
        # a line that says n_gpus = 1 under --gpus 8 would be a wrong scaling point, not a slow one: refuse
        print("bench.py: --gpus %d but %d rank(s) were started (WORLD_SIZE); launch with `python bench.py --gpus N` "
              "(self-launching) or torch.distributed.run --nproc-per-node N" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.load()
    backend = torch.distributed.get_backend() if world > 1 else None
    props = torch.cuda.get_device_properties(local)
    me = {"rank": rank, "device": "cuda:%d" % local, "name": props.name, "uuid": str(getattr(props, "uuid", "")), "pid": os.getpid()}
    devices = [me]
    if world > 1:
        devices = [None] * world
        torch.distributed.all_gather_object(devices, me)
        if not one_device and len({d["uuid"] or d["device"] for d in devices}) != world:
            print("bench.py: %d ranks share devices %s" % (world, devices), file=sys.stderr)
            sys.exit(2)
    cfg = args.workload
    c = synth.CONFIGS[cfg]
    strong = args.tiles_total > 0
    n_tiles = args.tiles_total if strong else world * args.batch
    my_tiles = adist.tiles_of_rank(n_tiles, rank, world)
    if not my_tiles:
        raise SystemExit("bench.py: --tiles-total %d leaves rank %d of %d without a tile" % (n_tiles, rank, world))
    G = args.groups

    with torch.no_grad():
        wl = Workload(cfg, my_tiles, args.precision, dev, groups=G, use_graph=not args.no_graph, baseline=args.baseline,
                      warm=max(args.warmup, 1))
        B, Bg = wl.B, wl.Bg
        depth, conf = wl.depth, wl.conf
        # one gather of the finished maps per step, staged through preallocated buffers and issued asynchronously:
        # the collective of step k overlaps the replay of step k+1 (ada_mvs_amd/dist.py)
        gatherer = adist.MapGatherer(n_tiles, B, depth.shape[-2], depth.shape[-1], dev)

        def step():
            d, p = wl.step()
            gatherer.start(d, p)

        for _ in range(args.warmup):
            step()
        gatherer.finish()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        gathered = gatherer.finish()             # the last gather is inside the timed region
        torch.cuda.synchronize()
        t_own = time.perf_counter()
        if world > 1:
            torch.distributed.barrier()
        elapsed = time.perf_counter() - t0
        rank_ms = [1e3 * (t_own - t0) / args.steps]                  # this rank's own replays + gathers, before the closing barrier
        if world > 1:
            cdev = dev if torch.distributed.get_backend() == "nccl" else "cpu"
            t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t.item())
            mine = torch.tensor([rank_ms[0]], device=cdev, dtype=torch.float64)
            every = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(every, mine)
            rank_ms = [float(x.item()) for x in every]

        result = None
        if rank == 0:
            assert gathered[0].shape[0] == n_tiles and bool(torch.isfinite(gathered[0]).all())
            tile0 = (gathered[0][0].clone(), gathered[1][0].clone())      # the graph's output buffers are reused below
            cascade_txt = "x".join(map(str, c["ndepths"]))
            result = {
                "metric": "depth maps/sec at %dx%dx%d-view x%s-hyp (hot path, features resident in HBM)" % (c["W"], c["H"], c["views"], cascade_txt),
                "value": n_tiles * args.steps / elapsed, "unit": "depth maps/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
                "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
                "rccl_ranks": world, "backend": {"nccl": "nccl (RCCL)"}.get(backend, backend), "devices": devices,
                "launcher": os.environ.get("ADAMVS_BENCH_LAUNCHER", "torch.distributed.run" if world > 1 else "none"),
                # every rank's own time per step before the closing barrier: a straggler GPU shows as max >> min
                "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "per_rank": [round(x, 4) for x in rank_ms]},
                "dtype": "f32" if args.precision == "fp32" else "bf16x3 (split-bf16 MFMA, fp32 accumulate) for the convolutions, f32 elsewhere",
                "data": "synthetic",
                "config": {"workload": "%s: %d views, %dx%d, hypotheses %s, %s" % (
                    cfg, c["views"], c["W"], c["H"], "/".join(map(str, c["ndepths"])), args.precision),
                    "tiles_per_gpu_per_step": B if not strong else "%d..%d" % (n_tiles // world, -(-n_tiles // world)),
                    "concurrent_tile_groups": G, "global_tiles_per_step": n_tiles,
                    "parallelism": "tile-sharded x%d, 1 RCCL gather per step" % world,
                    "launch": "eager" if wl.graph is None else "hipGraph replay",
                    "rig_baseline": args.baseline},
                # upstream of the timed region: FeatureNet0 on all views of a tile (csrc/featnet.hip)
                "feature_net_ms_per_tile": 1e3 * wl.t_feat / B,
                "end_to_end_maps_per_s_per_gpu": B / (wl.t_feat + elapsed / args.steps),
            }

        if rank == 0 and world == 1 and not args.no_roofline:
            result.update(roofline_of(wl, args, result["ms_per_step"]))
        if rank == 0 and world > 1:
            result["roofline"] = result["cpu_baseline"] = None
            result["note"] = "roofline, cpu_baseline, parity_rel_l1 and cascade are measured by the N = 1 line only (per-kernel timing and the oracle run on rank 0's GPU / host cores)"
        parity_ok = True
        sd = wl.sd
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            # the oracle runs tile 0 of this very workload: its maps are the parity check of the timed run
            result["cpu_baseline"], ref = cpu_baseline(cfg, sd, args.baseline)
            result["parity_rel_l1"] = parity_of(tile0, ref, c, wl.interval)
            parity_ok = max(result["parity_rel_l1"]["depth"], result["parity_rel_l1"]["photometric_confidence"]) <= 1e-3
        wl.close()
        del wl, gatherer, gathered, depth, conf

        # BASELINE.json's cascade configurations beside the headline: cfg3 (configs[2]) and one GPU's share of cfg4
        # (configs[3]), each with parity against ONE oracle pass of cfg3's tile 0
        if rank == 0 and world == 1 and cfg == "cfg2" and not strong and not args.no_cascade:
            cas, tiles0, sd3 = bench_cascade(dev, max(3, min(args.steps, 10)), max(1, min(args.warmup, 3)), args.baseline,
                                             use_graph=not args.no_graph)
            if not args.no_cpu_baseline:
                base3, ref3 = cpu_baseline("cfg3", sd3, args.baseline)
                c3 = synth.CONFIGS["cfg3"]
                for key in tiles0:
                    cas[key]["parity_rel_l1"] = parity_of(tiles0[key], ref3, c3, (synth.DEPTH_RANGE[1] - synth.DEPTH_RANGE[0]) / c3["num_depth"])
                    parity_ok = parity_ok and max(cas[key]["parity_rel_l1"]["depth"], cas[key]["parity_rel_l1"]["photometric_confidence"]) <= 1e-3
                cas["cpu_baseline"] = base3
            result["cascade"] = cas
        if rank == 0:
            print(json.dumps(result))
        if not parity_ok:
            print("bench.py: parity_rel_l1 above 1e-3 -- the measured path is wrong", file=sys.stderr)
            sys.exit(3)
    if world > 1:
        torch.distributed.destroy_process_group()


def parity_of(tile0, ref, c, depth_interval):
    """Tile 0 of the measured run against the oracle's maps: relative L1 (the north-star bar, 1e-3) and the mean depth
    error in units of the FINEST hypothesis interval (the last stage's ratio x depth_interval) -- relative L1 of a depth map
    around 500 hides a quarter of an interval behind 5e-4."""
    d, p = tile0[0].detach().double().cpu(), tile0[1].detach().double().cpu()
    rd, rp = ref["depth"][0].double(), ref["photometric_confidence"][0].double()
    finest = synth.DEPTH_INTERVALS_RATIO[len(c["ndepths"]) - 1] * depth_interval
    return {"depth": rel_l1(d, rd), "photometric_confidence": rel_l1(p, rp),
            "depth_abs_err_in_finest_intervals": float((d - rd).abs().mean() / finest),
            "tolerance": 1e-3, "tile": 0, "against": "oracle/adamvs_oracle.py (cpu_baseline run)"}


def wino_note(work, args):
    return winograd_active(work[0]["D"], args.precision)


def roofline_of(wl, args, ms_per_step):
    """roofline + phase tables of the headline workload: HIP-event timing of the stage run phase by phase through the C ABI."""
    result = {}
    cfg, Bg, B = wl.cfg, wl.Bg, wl.B
    feats_cl, shapes = wl.groups[0]
    times = timed_phases(wl.model, feats_cl, shapes, wl.projs[0], wl.dvs[0], wl.interval, max(2, min(args.steps, 5)))
    work = algorithmic_work(cfg, Bg)
    avg = {k: sum(v[1:]) / max(len(v) - 1, 1) for k, v in times.items()}        # drop the first call
    phases = {}
    for k, v in avg.items():                                                   # fold the per-layer marks
        key = k.split(".costreg.")[0] + ".cost_reg_net_2d" if ".costreg." in k else k
        phases[key] = phases.get(key, 0.0) + v
    result["phase_ms_per_step"] = {k: round(v, 4) for k, v in sorted(phases.items())}
    layers = {k.split(".costreg.")[1]: round(v, 4) for k, v in avg.items() if ".costreg." in k}
    if layers:
        result["cost_reg_layers_ms"] = layers          # one launch each: <layer>.mode<0 s1 | 1 s2 | 2 transposed>
        if "prob+softmax.mode0" in layers and wino_note(work, args):
            result["cost_reg_layers_note"] = ("prob+softmax.mode0 = adamvs_prob_softmax_regress_wino, as the step runs it: k_conv_wino<..., SM> "
                                              "(softmax partials in the layer's epilogue, no score volume) + k_softmax_merge, timed together")
    dom = max(phases, key=phases.get)
    st = work[int(dom[1]) - 1]
    kind = dom.split(".", 1)[1]
    c0 = synth.CONFIGS[cfg]
    split = args.precision == "bf16x3"
    if kind == "cost_reg_net_2d":
        # dominant kernel = the stride-1 instantiation of k_conv_dd (conv0, conv2, conv4, prob; conv6 too unless its grid is small).  The split-bf16
        # mode EXECUTES three bf16 products per fp32 product on the bf16 matrix pipe: priced against that pipe's peak.
        hw0, Dd, N = st["h"] * st["w"], st["D"], (c0["views"] - 1) * Bg
        res = {"conv0": 1, "conv2": 2, "conv4": 4, "conv6": 8, "prob": 1, "prob+softmax": 1}       # linear down-scale of the layer's maps

        wino = winograd_active(Dd, args.precision)

        def on_dominant_kernel(layer):
            # the fp32 path sends stride-1 layers of at most 2048 blocks of 8 x 16 pixels to the 2-row kernel
            # k_conv_dd_rows2 (csrc/costreg2d.hip: small_grid_rows2); those launches are not the dominant kernel's
            if layer == "prob+softmax":
                # F(2x2, 3x3): `prob` is the SM instantiation of the same kernel family (softmax partials in its epilogue) with
                # k_softmax_merge (0.4 ms) behind it -- timed together and counted: the fraction is what the step runs, slightly
                # pessimistic.  Direct kernels: the last layer's own instantiation, not part of this kernel's launches.
                return wino
            if split or wino:
                return True
            e = _lib.get_option("conv_rows2")
            if e >= 0:
                return e == 0
            r = res[layer]
            return -(-(st["w"] // r) // 16) * -(-(st["h"] // r) // 8) * N > 2048

        lay = {k: v for k, v in avg.items() if ".costreg." in k and k.endswith("mode0") and on_dominant_kernel(k.split(".")[2])}
        direct = sum(2.0 * N * (hw0 // res[k.split(".")[2]] ** 2) * Dd * Dd * 9 for k in lay)
        # executed flops: three bf16 products per fp32 product (split-bf16); 16 products per 2x2 outputs instead of 36 (F(2x2, 3x3))
        flops = direct * (3 if split else (16.0 / 36.0 if wino else 1))
        ms = sum(lay.values())
        ach = flops / (ms * 1e-3) / 1e12
        peak = BF16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
        kname = ("k_conv_dd_bx3<MT,WM,CONV_S1>, executed bf16 flops = 3 x fp32 products" if split else
                 "k_conv_wino<4,NT,WPS,SM>, fp32 F(2x2,3x3): executed flops = 16/36 of the direct convolution's" if wino else "k_conv_dd<MT,WM,CONV_S1>")
        roof = {"kernel": "%s (CostRegNet2D 3x3 stride-1 layers %s, %d launches per step)" % (
                    kname, "+".join(k.split(".")[2] for k in lay), len(lay)),
                "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "launch_ms": ms / len(lay),
                "flops_per_launch": flops / len(lay), "traffic": None}
        if wino:        # the same launches priced as the direct convolution they replace (can exceed the matrix peak)
            roof["direct_equivalent_tflops"] = direct / (ms * 1e-3) / 1e12
            roof["direct_flops_per_launch"] = direct / len(lay)
        stamp = source_stamp()
        for tpath in sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")), reverse=True):
            tj = json.load(open(tpath))       # HBM bytes per launch from committed PMC passes (tools/profile_round.sh)
            if tj.get("config") != {"workload": cfg, "tiles_per_launch": Bg} or tj.get("precision", "fp32") != args.precision:
                continue
            if tj.get("source_stamp") != stamp:
                roof["traffic_note"] = "%s was measured on another build (stamp %s, this build %s): not quoted" % (
                    os.path.basename(tpath), tj.get("source_stamp"), stamp)
                continue
            if tj.get("hot_path_bytes_per_pass"):       # all kernels of one step, 2*FETCH + WRITE, against SURVEY 8d's algorithmic bytes
                alg = sum(sum(v for k_, v in w_.items() if k_.endswith("_bytes")) for w_ in work)
                roof["traffic_bytes_per_step"] = tj["hot_path_bytes_per_pass"]
                roof["traffic_over_algorithmic_bytes"] = tj["hot_path_bytes_per_pass"] / alg
            # every instantiation of the kernel family the layers run on (F(2x2, 3x3): the one- and two-workgroups-per-CU tilings and
            # the softmax form of `prob`), averaged over their launches
            k0 = [v for k, v in tj["kernels"].items() if ("k_conv_dd_bx3<3, 4, 0" if split else "k_conv_wino<" if wino else "k_conv_dd<3, 4, 0, 4, false, false>") in k]
            if k0:
                roof["traffic"] = sum((2 * v["fetch_size_kib"] + v["write_size_kib"]) * 1024 * v["launches"] for v in k0) / sum(v["launches"] for v in k0)
                roof["traffic_note"] = ("bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (gfx950 float4 correction), %s, "
                                        "same source stamp %s" % (os.path.basename(tpath), stamp))
                break
    elif kind == "recurrence":
        # executed flops: the split-bf16 mode issues three bf16 products per fp32 product; fp32 stages that run one role per
        # launch (or the three-launch schedule) take their gate / level-2 candidate convolutions in the F(2x2, 3x3) form: 16 of 36
        rec_flops = st["recurrence_flops"] * (3 if split else 1)
        if not split:
            lib = _lib.load()
            mask = lib.adamvs_gru_wino_mask()
            sched = lib.adamvs_recurrence_schedule(0, Bg * st["h"] * st["w"])
            if sched == 0 or (sched == 1 and (mask & 7) == 7):
                eff = mask if sched == 0 else 7
                rec_flops -= Bg * st["D"] * st["h"] * st["w"] * 2.0 * sum(m for b_, m in st["gru_conv_macs"].items() if eff & b_) * (20.0 / 36.0)
        ach = rec_flops / (avg[dom] * 1e-3) / 1e12
        peak = BF16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
        # launches per hypothesis depend on the stage size (csrc/recurrence.hip: 6 / 3 / 2 in fp32, 4 / 2 in bf16x3):
        # the figure priced here is one hypothesis = one recurrent step of all tiles
        roof = {"kernel": "the ConvGRU / decoder tile loops of one hypothesis (%d hypotheses per step)" % st["D"], "bound": "mfma",
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "launch_ms": avg[dom] / st["D"], "traffic": None}
        if split:
            roof["note"] = ("short K (72-288) convolutions on 8-16 channel maps: bound by launch latency, LDS and the "
                            "VALU work of the hi/lo split, not by the bf16 matrix pipe")
    else:
        key = {"pair_similarity": "pair_similarity_bytes", "aggregate_conv1": "aggregate_bytes",
               "soft_argmin": "softargmin_bytes", "softmax_max_regress": "softmax_bytes",
               "view_weight_resample": "softmax_bytes"}[kind]
        ach = st[key] / (avg[dom] * 1e-3) / 1e9
        roof = {"kernel": "k_" + kind, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "launch_ms": avg[dom], "traffic": None}
    roof.update(wl.step_fracs(ms_per_step))
    roof["dominant_phase"] = dom
    result["roofline"] = roof
    return result


if __name__ == "__main__":
    main()
