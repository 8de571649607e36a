"""Drop-in `Infer_AdaMVSNet` for MI355X: the constructor, forward() signature,
output dict and state-dict keys of reference models/adamvs.py:537-620, with the
depth-inference hot path (reference models/adamvs.py:426-533 and the functions
it calls in models/module.py) running in libadamvs_hip.so.

    from models.adamvs import Infer_AdaMVSNet          # with ada-mvs_amd/ on sys.path
    model = Infer_AdaMVSNet(num_depth, ndepths, depth_intervals_ratio, share_cr, cr_base_chs)
    model = nn.DataParallel(model).cuda(); model.load_state_dict(ckpt["model"]); model.eval()
    out = model(imgs, proj_matrices, depth_values)     # out["depth"], out["photometric_confidence"], ...

FeatureNet0 (upstream of the hot path) runs on hand-written kernels as well
(adamvs_feature_net0).  Image sizes that are no multiple of 32 raise, as they
break the reference's own size rule (its predict-time loader crops to
multiples of 32, datasets/preprocess.py:68-83).  There is no CPU fallback and
no PyTorch fallback on the GPU: CPU tensors, a missing library, or train mode raise.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ada_mvs_amd import hip_ops, packing
from ada_mvs_amd._lib import AdaMVSHipError, PRECISIONS as _PRECISIONS
from .module import Conv2d, ConvBnReLU, ConvGRUCell, ConvReLU, DeConv2dFuse, PackedCache, module_state

STAGE_SCALE = {"stage1": 4, "stage2": 2, "stage3": 1}


def _pooled_context(in_ch, out_ch, pool):
    return nn.Sequential(nn.AvgPool2d((pool, pool), stride=(pool, pool)),
                         Conv2d(in_ch, out_ch, 1, stride=1, padding=0, dilation=1))


class FeatureNet0(PackedCache, nn.Module):
    """2D U-Net with pooled-context branches, three output scales (C = 32/16/8 at 1/4, 1/2, 1/1);
    reference models/adamvs.py:49-152.  forward() = adamvs_feature_net0 (csrc/featnet.hip) on GPU tensors whose
    height and width are multiples of 32 (the reference's own size rule) and base_channels = 8; anything else raises."""

    def __init__(self, base_channels, num_stage=3, stride=4):
        super().__init__()
        c = base_channels
        self.stride, self.base_channels, self.num_stage = stride, c, num_stage
        self.conv0 = nn.Sequential(Conv2d(3, c, 3, 1, padding=1), Conv2d(c, c, 3, 1, padding=1))
        self.conv1 = nn.Sequential(Conv2d(c, 2 * c, 5, stride=2, padding=2), Conv2d(2 * c, 2 * c, 3, 1, padding=1),
                                   Conv2d(2 * c, 2 * c, 3, 1, padding=1))
        self.conv2 = nn.Sequential(Conv2d(2 * c, 4 * c, 5, stride=2, padding=2), Conv2d(4 * c, 4 * c, 3, 1, padding=1),
                                   Conv2d(4 * c, 4 * c, 3, 1, padding=1))
        self.branch1_1 = _pooled_context(4 * c, 2 * c, 4)
        self.branch1_2 = _pooled_context(4 * c, 2 * c, 8)
        self.out1 = nn.Conv2d(8 * c, 4 * c, 1, bias=False)
        self.deconv1 = DeConv2dFuse(4 * c, 2 * c, 3)
        self.deconv2 = DeConv2dFuse(2 * c, c, 3)
        self.branch2_1 = _pooled_context(2 * c, c, 4)
        self.branch2_2 = _pooled_context(2 * c, c, 8)
        self.branch3_1 = _pooled_context(c, c // 2, 4)
        self.branch3_2 = _pooled_context(c, c // 2, 8)
        self.out2 = nn.Conv2d(4 * c, 2 * c, 1, bias=False)
        self.out3 = nn.Conv2d(2 * c, c, 1, bias=False)
        self.out_channels = [4 * c, 2 * c, c]
        self._cache_init()
        self.workspace_limit_bytes = 32 << 30      # forward_cl runs larger batches in chunks

    def packed(self, device):
        def build():
            flat, offsets = packing.pack_feature_net(module_state(self), "")
            return hip_ops.PackedFeature(flat, offsets, device)
        return self.cached(device, build)

    def hip_supported(self, x):
        return x.is_cuda and self.base_channels == 8 and x.shape[-2] % 32 == 0 and x.shape[-1] % 32 == 0

    def _require_hip(self, x):
        """forward() / forward_cl() are the HIP kernels or nothing: no PyTorch path is taken silently.  (The reference
        has the same size rule, quirk Q9: its pooled-context branches and three /2 levels need multiples of 32, which
        crop_input guarantees; base_channels is 8 in every model it builds.)  forward_torch() is the explicitly named
        PyTorch evaluation of the same layers, for comparisons."""
        if not self.hip_supported(x):
            raise AdaMVSHipError("%s: needs a GPU tensor [N,3,H,W] with H, W multiples of 32 and base_channels 8 (got %s on %s, "
                                 "base_channels %d); there is no fallback path" % (type(self).__name__, tuple(x.shape), x.device, self.base_channels))

    def forward_cl(self, x):
        """[N,3,H,W] -> channel-last stage maps ([N,hw/16,32], [N,hw/4,16], [N,hw,8]) for the plane sweep.
        x may also be [B,V,3,H,W], as the reference's forward() receives the images: the maps then hold the V*B images in
        view-major order (image v*B + b), the order the plane sweep takes, read in place -- no transposed copy."""
        self._require_hip(x)
        by_view = x.dim() == 5
        N = x.shape[0] * x.shape[1] if by_view else x.shape[0]
        # intermediate maps take ~65 floats per pixel and image: bound the workspace, not the batch
        per_image = hip_ops.feature_net0_workspace_bytes(1, x.shape[-2], x.shape[-1])
        chunk = max(1, int(self.workspace_limit_bytes // per_image))
        if N <= chunk:
            return hip_ops.feature_net0(x, self.packed(x.device), views=(0, N) if by_view else None)
        H, W = x.shape[-2], x.shape[-1]      # chunks write into their slices of the whole maps (no concatenation copy)
        maps = tuple(torch.empty(N, (H // s) * (W // s), c, device=x.device, dtype=torch.float32) for s, c in ((4, 32), (2, 16), (1, 8)))
        for i in range(0, N, chunk):
            n = min(chunk, N - i)
            hip_ops.feature_net0(x if by_view else x[i:i + n], self.packed(x.device), out=tuple(m[i:i + n] for m in maps),
                                 views=(i, n) if by_view else None)
        return maps

    @staticmethod
    def _with_context(feat, branch_a, branch_b):
        size = feat.shape[2:]
        a = F.interpolate(branch_a(feat), size=size, mode="bilinear", align_corners=False)
        b = F.interpolate(branch_b(feat), size=size, mode="bilinear", align_corners=False)
        return torch.cat((a, b, feat), 1)

    def forward(self, x):
        self._require_hip(x)
        H, W = x.shape[-2:]
        s1, s2, s3 = hip_ops.feature_net0(x, self.packed(x.device))
        return {"stage1": hip_ops.unpack_features(s1, H // 4, W // 4), "stage2": hip_ops.unpack_features(s2, H // 2, W // 2),
                "stage3": hip_ops.unpack_features(s3, H, W)}

    def forward_torch(self, x):
        c0 = self.conv0(x)
        c1 = self.conv1(c0)
        c2 = self.conv2(c1)
        out = {"stage1": self.out1(self._with_context(c2, self.branch1_1, self.branch1_2))}
        f = self.deconv1(c1, c2)
        out["stage2"] = self.out2(self._with_context(f, self.branch2_1, self.branch2_2))
        f = self.deconv2(c0, f)
        out["stage3"] = self.out3(self._with_context(f, self.branch3_1, self.branch3_2))
        return out


class CostRegNet2D(PackedCache, nn.Module):
    """Depth-as-channel 2D hourglass (reference models/adamvs.py:198-238); forward = adamvs_cost_reg_net_2d."""

    def __init__(self, in_channels, base_channels=8):
        super().__init__()
        d = in_channels
        for i, stride in enumerate((1, 2, 1, 2, 1, 2, 1)):
            setattr(self, "conv%d" % i, ConvBnReLU(d, d, stride=stride))
        for i in (7, 9, 11):
            setattr(self, "conv%d" % i, nn.Sequential(
                nn.ConvTranspose2d(d, d, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
                nn.BatchNorm2d(d), nn.ReLU(inplace=True)))
        self.prob = nn.Conv2d(d, d, 3, stride=1, padding=1)
        self.precision = "fp32"            # or "bf16x3": split-bf16 MFMA, fp32-equivalent to ~1e-5 (needs d % 32 == 0)
        self._cache_init()

    def effective_precision(self):
        return self.precision if self.prob.weight.shape[0] % 32 == 0 else "fp32"

    def packed(self, device):
        prec = self.effective_precision()
        return self.cached((device, prec), lambda: packing.pack_cost_reg_net_2d(module_state(self), "", prec).to(device))

    def forward(self, x):
        N, D, h, w = x.shape
        prec = self.effective_precision()
        x_cl = hip_ops.pack_features(x)
        dr = packing.reg_width(D, prec)          # the kernels' width for D hypotheses: zero channels in, pad scores dropped
        if dr != D:
            x_cl = torch.nn.functional.pad(x_cl, (0, dr - D))
        score = hip_ops.cost_reg_net_2d(x_cl, self.packed(x.device), h, w, _PRECISIONS[prec])
        if dr != D:
            score = score[..., :D].contiguous()
        return hip_ops.unpack_features(score, h, w)


class SliceCostRegNetRED(PackedCache, nn.Module):
    """One recurrent regularisation step (reference models/adamvs.py:400-424); forward = adamvs_slice_reg_step."""

    def __init__(self, in_channels, up=True, base_channels=8):
        super().__init__()
        c = base_channels
        self.base_channels, self.in_channels, self.up = c, in_channels, up
        self.conv1 = ConvReLU(in_channels, c, 3, 1, 1)
        self.conv_gru1 = ConvGRUCell(c, c, 3)
        self.conv2 = ConvReLU(c, 2 * c, 3, 2, 1)
        self.conv_gru2 = ConvGRUCell(2 * c, 2 * c, 3)
        self.upconv1 = nn.ConvTranspose2d(2 * c, c, kernel_size=3, stride=2, padding=1, output_padding=1)
        if up:
            self.upconv2d = nn.ConvTranspose2d(c, 1, kernel_size=3, stride=2, padding=1, output_padding=1)
        else:
            self.upconv2d = nn.Conv2d(c, 1, kernel_size=3, stride=1, padding=1)
        self.precision = "fp32"            # or "bf16x3" (split-bf16 MFMA for conv1 and the ConvGRU convolutions)
        self._cache_init()

    def packed(self, device):
        if self.base_channels != 8:
            raise AdaMVSHipError("SliceCostRegNetRED: the reference hard-codes 8/16 GRU widths (adamvs.py:448-449)")
        prec = self.precision

        def build():
            flat, offsets = packing.pack_slice_reg_net(module_state(self), "", prec)
            return hip_ops.PackedFuse(flat, offsets, device)
        return self.cached((device, prec), build)

    def forward(self, cost, state1, state2):
        B, C, h, w = cost.shape
        s1 = hip_ops.pack_features(state1)
        s2 = hip_ops.pack_features(state2)
        reg = hip_ops.slice_reg_step(hip_ops.pack_features(cost), s1, s2, self.packed(cost.device), B, C, h, w, self.up,
                                     _PRECISIONS[self.precision])
        return reg, hip_ops.unpack_features(s1, h, w), hip_ops.unpack_features(s2, h // 2, w // 2)


class InferDepthNet0(nn.Module):
    """One cascade stage (reference models/adamvs.py:426-533) = adamvs_depth_stage_forward."""

    def __init__(self, in_depths, in_channels, in_up=True, base_channels=8):
        super().__init__()
        self.in_up = in_up
        self.reg = CostRegNet2D(in_depths, base_channels)
        self.reg_fuse = SliceCostRegNetRED(in_channels, in_up, base_channels)
        self.mirror_list_lengths = True        # pair_confidence carries the reference's S*D duplicate entries (quirk Q1)
        self._workspace = {}                   # one workspace per (device, concurrent tile group); shared by DataParallel replicas

    def _apply(self, fn, *a, **k):             # .cuda()/.to(): the workspaces belong to the old placement
        self._workspace.clear()
        return super()._apply(fn, *a, **k)

    def run(self, feat_cl, B, C, h, w, rt, depth_values, prev_conf, group=0, twin=False, planes=None, num_depth=None,
            workspaces=None, phases=None, outputs=None, timing_only=False):
        """feat_cl [V*B, h*w, C] view-major channel-last; rt [B,S,12]; depth_values [B,D,h,w] -- or None with
        planes = hip_ops.plane_source(...) and num_depth: the hypothesis planes are then generated inside the kernels;
        prev_conf None (stage 1) or [S,B,hp,wp].  -> (view_weight [S,B,h,w], pair_depth|None, depth, conf).
        twin: the train/test model's placement of the 1e-5 in the weighted aggregation (adamvs.py:262-300).
        workspaces: a table shared with the other stages of a cascade (they run one after the other on one stream, so
        one buffer of the largest stage's size serves all three); default: this stage's own.
        phases / outputs: run a subset of the stage's phases on given output tensors (ada_mvs_amd.dist: pass A of
        stage 1 for a subset of the source views, then pass B on the gathered view weights)."""
        S = feat_cl.shape[0] // B - 1
        first = prev_conf is None
        prev_hw = (0, 0) if first else tuple(prev_conf.shape[-2:])
        if planes is None:
            D, mode, half_span, span_dev = depth_values.shape[1], 0, 0.0, None
        else:
            (mode, half_span, depth_values), D = planes[:3], num_depth
            span_dev = planes[3] if len(planes) > 3 else None
        if first and D != self.reg.prob.weight.shape[0]:
            # CostRegNet2D is D -> D (reference adamvs.py:198-228, built with in_depths = ndepths[0]): the reference's conv0 fails on
            # any other plane count.  Here the network may run zero-padded to a wider tiling (packing.reg_width), so a count that
            # happens to equal the PADDED width would pass every size check of the library and weight the surplus planes 0.
            raise AdaMVSHipError("stage 1 was given %d hypothesis planes but its CostRegNet2D is built for %d (ndepths[0])"
                                 % (D, self.reg.prob.weight.shape[0]))
        desc = hip_ops.stage_desc(B, S, C, h, w, D, self.in_up, first, prev_hw, _PRECISIONS[self.reg.effective_precision()],
                                  _PRECISIONS[self.reg_fuse.precision], eps_in_numerator=int(twin), plane_mode=mode, half_span=half_span,
                                  half_span_dev=span_dev)
        need = hip_ops.depth_stage_workspace_bytes(desc) // 4
        table = self._workspace if workspaces is None else workspaces
        key = (feat_cl.device, group)
        ws = table.get(key)
        if ws is None or ws.numel() < need:
            table.pop(key, None)               # release the smaller buffer before asking for the larger one
            ws = None
            ws = table[key] = torch.empty(need, device=feat_cl.device, dtype=torch.float32)
        dev = feat_cl.device
        w_reg = self.reg.packed(dev) if first else None
        if phases is None:
            return hip_ops.depth_stage_forward(desc, feat_cl, rt, depth_values, prev_conf, w_reg, self.reg_fuse.packed(dev), ws)
        return hip_ops.depth_stage_forward(desc, feat_cl, rt, depth_values, prev_conf, w_reg, self.reg_fuse.packed(dev), ws,
                                           phases=phases, outputs=outputs, timing_only=timing_only)

    def forward(self, features, proj_matrices, depth_values, num_depth, confidence_map=None):
        assert len(features) == proj_matrices.shape[1], "Different number of images and projection matrices"
        assert depth_values.shape[1] == num_depth, "depth_values.shape[1]:{}  num_depth:{}".format(depth_values.shape[1], num_depth)
        B, C, h, w = features[0].shape
        S = len(features) - 1
        feat_cl = hip_ops.pack_features(torch.stack(list(features), 0).reshape(-1, C, h, w))
        rt = hip_ops.relative_transforms(proj_matrices)
        prev = None
        if confidence_map is not None:
            prev = torch.stack([c.reshape(B, c.shape[-2], c.shape[-1]) for c in confidence_map[:S]], 0).contiguous()
        vw, pd, depth, conf = self.run(feat_cl, B, C, h, w, rt, depth_values, prev)
        return self._as_dict(vw, pd, depth, conf, num_depth)

    def _as_dict(self, vw, pd, depth, conf, num_depth):
        S, B, h, w = vw.shape
        maps = [vw[i].reshape(B, 1, h, w) for i in range(S)]
        pair_confidence = list(maps)
        if self.mirror_list_lengths:
            # the reference appends one (identical) resampled map per view per hypothesis (adamvs.py:505-506);
            # stage 1 keeps its S raw maps in front (adamvs.py:489-490)
            pair_confidence = (maps if pd is not None else []) + maps * num_depth
        pair_result = [pd[i] for i in range(S)] if pd is not None else []
        return {"depth": depth, "photometric_confidence": conf, "pair_confidence": pair_confidence, "pair_result": pair_result}


class Infer_AdaMVSNet(nn.Module):
    """reference models/adamvs.py:537-620"""

    def __init__(self, num_depth=384, ndepths=[48, 32, 8], depth_intervals_ratio=[4, 2, 1], share_cr=False,
                 cr_base_chs=[8, 8, 8], precision="fp32"):
        super().__init__()
        assert precision in _PRECISIONS, "precision must be one of %s" % sorted(_PRECISIONS)
        assert len(ndepths) == len(depth_intervals_ratio)
        self.num_depth = num_depth
        self.share_cr = share_cr                 # accepted and ignored, as in the reference (quirk Q8)
        self.ndepths = list(ndepths)
        self.depth_intervals_ratio = list(depth_intervals_ratio)
        self.cr_base_chs = cr_base_chs
        self.num_stage = len(ndepths)
        self.materialize_planes = False          # True: hypothesis planes as a [B,D,h,w] tensor instead of generated in the kernels
        self._stage_workspace = {}               # one workspace per (device, tile group) for all stages; shared by DataParallel replicas
        self.view_shard = None                   # (rank, world): latency mode -- pass A of stage 1 over this rank's share of the
                                                 # source views, one all_gather of the view weights (ada_mvs_amd.dist.sharded_view_weights)
        self.stage_infos = {k: {"scale": float(v)} for k, v in STAGE_SCALE.items()}
        self.feature = FeatureNet0(base_channels=8, stride=4, num_stage=self.num_stage)
        ch = self.feature.out_channels
        self.DepthNet = nn.ModuleList([InferDepthNet0(in_depths=self.ndepths[0], in_channels=ch[0]),
                                       InferDepthNet0(in_depths=self.ndepths[0], in_channels=ch[1]),
                                       InferDepthNet0(in_depths=self.ndepths[0], in_up=False, in_channels=ch[2])])
        self.set_precision(precision)

    def _apply(self, fn, *a, **k):             # .cuda()/.to(): the workspaces belong to the old placement
        self._stage_workspace.clear()
        return super()._apply(fn, *a, **k)

    def set_precision(self, precision):
        """"fp32": fp32 MFMA everywhere (default; CostRegNet2D's stride-1 layers in the F(2x2, 3x3) form at D in {64,128,192,256}).  "bf16x3": CostRegNet2D, conv1 and the ConvGRU convolutions
        on the bf16 matrix cores with split operands (hi + lo bf16, three MFMAs per product), fp32 accumulation --
        agrees with fp32 to ~1e-5."""
        assert precision in _PRECISIONS
        self.precision = precision
        for net in self.DepthNet:
            net.reg.precision = net.reg_fuse.precision = precision

    # ---- the hot path on pre-extracted features ---------------------------------------------
    def infer_from_features(self, feats_cl, shapes, proj_matrices, depth_values, depth_interval, group=0, twin=False, span_dev=None):
        """feats_cl[s]: [V*B, h*w, C] channel-last view-major; shapes[s] = (B, C, h, w).
        Everything below is HIP.  `group` selects the workspace: independent
        tile groups may run concurrently on different streams (the recurrence is latency-bound per group).
        twin: semantics of the train/test model AdaMVSNet (see that class).
        span_dev: float32 device tensor [num_stage] holding hip_ops.half_span_of(ndepths[s], ratio[s] * depth_interval) per stage; the
        window planes of stages 2, 3 then read their half span from it when the kernels run and `depth_interval` is not used for
        them -- nothing of the tile's depth range is baked into the launches (graphed.py replays one capture for every tile)."""
        outputs = {}
        depth, conf = None, None
        first_maps = None
        for s in range(self.num_stage):
            name = "stage%d" % (s + 1)
            B, C, h, w = shapes[s]
            cur = depth_values if depth is None else depth
            # get_depth_range_samples (module.py:646-663) without its [B,D,h,w] tensor: uniform planes at stage 1, the
            # per-pixel window around the previous stage's depth afterwards, generated where they are used
            rt = hip_ops.relative_transforms(proj_matrices[name])
            net = self.DepthNet[s]
            span = self.depth_intervals_ratio[s] * depth_interval
            if self.materialize_planes:      # the reference's way (and the A/B of the generated planes): a [B,D,h,w] tensor
                vw, pd, depth, pconf = net.run(feats_cl[s], B, C, h, w, rt, hip_ops.depth_range_samples(cur, self.ndepths[s], span, [B, h, w]),
                                               conf, group, twin, workspaces=self._stage_workspace)
            elif s == 0 and self.view_shard is not None:
                from ada_mvs_amd import dist as adist          # the one exchange step of the path
                planes = hip_ops.plane_source(cur, self.ndepths[s], span, [B, h, w])
                vw, pd, depth, pconf = adist.stage_with_sharded_views(net, feats_cl[s], B, C, h, w, rt, planes, self.ndepths[s], group, twin,
                                                                      self._stage_workspace, *self.view_shard)
            else:
                planes = hip_ops.plane_source(cur, self.ndepths[s], span, [B, h, w], None if span_dev is None else span_dev[s:s + 1])
                vw, pd, depth, pconf = net.run(feats_cl[s], B, C, h, w, rt, None, conf, group, twin, planes=planes, num_depth=self.ndepths[s],
                                               workspaces=self._stage_workspace)
            if twin:
                # DepthNet0 hands its input confidence list on unchanged (adamvs.py:298): every later stage resamples
                # the stage-1 maps, and the lists carry S entries
                conf = vw if s == 0 else conf
                maps = [conf[i].reshape(B, 1, *conf.shape[-2:]) for i in range(conf.shape[0])]
                st = {"depth": depth, "photometric_confidence": pconf, "pair_confidence": maps,
                      "pair_result": [pd[i] for i in range(pd.shape[0])] if pd is not None else []}
            else:
                conf = vw
                st = net._as_dict(vw, pd, depth, pconf, self.ndepths[s])
            outputs[name] = st
            outputs.update(st)
        return outputs

    def extract_features(self, imgs):
        """-> (feats_cl, shapes) for infer_from_features; FeatureNet0 on all B*V images in one batch."""
        B, V = imgs.shape[:2]
        H, W = imgs.shape[-2:]
        maps = self.feature.forward_cl(imgs.contiguous())      # view-major, channel-last maps straight from [B,V,3,H,W]; raises on unsupported input
        feats_cl, shapes = [], []
        for s in range(self.num_stage):
            scale = (4, 2, 1)[s]
            feats_cl.append(maps[s])
            shapes.append((B, maps[s].shape[-1], H // scale, W // scale))
        return feats_cl, shapes

    def forward(self, imgs, proj_matrices, depth_values):
        if not imgs.is_cuda:
            raise AdaMVSHipError("Infer_AdaMVSNet runs on MI355X only: move the model and its inputs to the GPU "
                                 "(no CPU fallback for the depth-inference path)")
        if self.training:
            # the kernels fold BatchNorm's running statistics (eval semantics); the reference's predict script always
            # calls .eval() (predict_whu.py:89), in train mode its BatchNorm would normalise with batch statistics
            raise AdaMVSHipError("Infer_AdaMVSNet implements the eval-mode forward only: call .eval() (predict_whu.py:89)")
        return self._forward_infer(imgs, proj_matrices, depth_values)

    def _forward_infer(self, imgs, proj_matrices, depth_values):
        depth_min = float(depth_values[0, 0].cpu().numpy())        # batch item 0 only, host sync (adamvs.py:569-571)
        depth_max = float(depth_values[0, -1].cpu().numpy())
        depth_interval = (depth_max - depth_min) / self.num_depth
        feats_cl, shapes = self.extract_features(imgs)
        return self.infer_from_features(feats_cl, shapes, proj_matrices, depth_values, depth_interval)


class AdaMVSNet(Infer_AdaMVSNet):
    """The reference's train/test model (models/adamvs.py:311-396; DepthNet0 241-305, CostRegNetRED 157-195) in eval
    mode on the same kernels -- SURVEY.md section 8f row f4: `train_whu.py --mode test` on the fast path.  Same
    state-dict keys as Infer_AdaMVSNet (the reference loads one checkpoint into both).  What differs from the
    inference model, and is reproduced: depth_values = [min, max, interval] per tile (adamvs.py:344-347); the 1e-5
    of the weighted aggregation sits in the numerator (adamvs.py:262, 283-300); every later stage resamples the
    stage-1 view weights (DepthNet0 returns its confidence_map argument, adamvs.py:298); `pair_confidence` holds S
    maps.  The softmax / depth_regression / max of adamvs.py:302-305 is the online soft-argmin of the inference
    model without its +1e-10.  Training (backward) is out of scope: forward() raises in train mode."""

    def __init__(self, ndepths=[48, 32, 8], depth_intervals_ratio=[4, 2, 1], share_cr=False, cr_base_chs=[8, 8, 8],
                 precision="fp32"):
        super().__init__(num_depth=ndepths[0], ndepths=ndepths, depth_intervals_ratio=depth_intervals_ratio,
                         share_cr=share_cr, cr_base_chs=cr_base_chs, precision=precision)

    def forward(self, imgs, proj_matrices, depth_values):
        if self.training:
            raise AdaMVSHipError("AdaMVSNet (MI355X build) implements the eval-mode forward only: call .eval()")
        if not imgs.is_cuda:
            raise AdaMVSHipError("AdaMVSNet runs on MI355X only: move the model and its inputs to the GPU")
        depth_interval = float(depth_values[0, -1].cpu().numpy())  # batch item 0 only (adamvs.py:346)
        depth_range = depth_values[:, 0:-1].contiguous()            # [min, max]
        feats_cl, shapes = self.extract_features(imgs)
        return self.infer_from_features(feats_cl, shapes, proj_matrices, depth_range, depth_interval, twin=True)
