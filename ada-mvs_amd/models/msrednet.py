"""Drop-in `Infer_CascadeREDNet` for MI355X: constructor, forward() signature, output dict and state-dict keys of
reference models/msrednet.py:440-526 (`predict_whu.py --model msrednet`).

Per hypothesis plane: the variance cost of the reference and the warped source features (msrednet.py:396-412), the
four-level GroupNorm ConvGRU encoder-decoder (slice_RED_Regularization, msrednet.py:330-366) and the running
exp-sum / max / weighted-depth update (msrednet.py:415-436).  Every convolution runs on the fp32-MFMA k_conv_dd
kernel (adamvs_conv3x3_dd) over channel-last maps whose channel count is zero-padded to a supported width; the
GroupNorm statistics, gate / candidate epilogues and the variance cost are the kernels of csrc/msred.hip; FeatureNet
is adamvs_feature_net0 with zero context-branch weights ('unet') or adamvs_feature_net_fpn ('fpn').  No CPU fallback.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ada_mvs_amd import hip_ops, packing
from ada_mvs_amd._lib import AdaMVSHipError
from .adamvs import STAGE_SCALE, FeatureNet0
from .module import Conv2d, ConvReLU, DeConv2dFuse, PackedCache, _FusedLayer, module_state


class FeatureNet(FeatureNet0):
    """Reference models/msrednet.py:29-127, three stages.  arch_mode 'unet': the plain U-Net = FeatureNet0 without the
    pooled-context branches (adamvs_feature_net0 with zero branch weights); arch_mode 'fpn': lateral 1x1 convolutions
    added to the nearest-upsampled coarser map, 3x3 output convolutions (adamvs_feature_net_fpn)."""

    def __init__(self, base_channels, num_stage=3, stride=4, arch_mode="unet"):
        nn.Module.__init__(self)
        assert arch_mode in ["unet", "fpn"], "mode must be in 'unet' or 'fpn', but get:{}".format(arch_mode)
        assert num_stage == 3, "this build implements three stages"
        c = base_channels
        self.arch_mode, self.stride, self.base_channels, self.num_stage = arch_mode, stride, c, num_stage
        self.conv0 = nn.Sequential(Conv2d(3, c, 3, 1, padding=1), Conv2d(c, c, 3, 1, padding=1))
        self.conv1 = nn.Sequential(Conv2d(c, 2 * c, 5, stride=2, padding=2), Conv2d(2 * c, 2 * c, 3, 1, padding=1),
                                   Conv2d(2 * c, 2 * c, 3, 1, padding=1))
        self.conv2 = nn.Sequential(Conv2d(2 * c, 4 * c, 5, stride=2, padding=2), Conv2d(4 * c, 4 * c, 3, 1, padding=1),
                                   Conv2d(4 * c, 4 * c, 3, 1, padding=1))
        self.out1 = nn.Conv2d(4 * c, 4 * c, 1, bias=False)
        if arch_mode == "unet":
            self.deconv1 = DeConv2dFuse(4 * c, 2 * c, 3)
            self.deconv2 = DeConv2dFuse(2 * c, c, 3)
            self.out2 = nn.Conv2d(2 * c, 2 * c, 1, bias=False)
            self.out3 = nn.Conv2d(c, c, 1, bias=False)
        else:
            self.inner1 = nn.Conv2d(2 * c, 4 * c, 1, bias=True)
            self.inner2 = nn.Conv2d(c, 4 * c, 1, bias=True)
            self.out2 = nn.Conv2d(4 * c, 2 * c, 3, padding=1, bias=False)
            self.out3 = nn.Conv2d(4 * c, c, 3, padding=1, bias=False)
        self.out_channels = [4 * c, 2 * c, c]
        self._cache_init()
        self.workspace_limit_bytes = 32 << 30

    def packed(self, device):
        def build():
            if self.arch_mode == "fpn":
                flat, offsets = packing.pack_feature_net_fpn(module_state(self), "")
                return hip_ops.PackedFeatureFpn(flat, offsets, device)
            flat, offsets = packing.pack_feature_net(module_state(self), "", context=False)
            return hip_ops.PackedFeature(flat, offsets, device)
        return self.cached(device, build)

    def _run_hip(self, x):
        if self.arch_mode == "fpn":
            return hip_ops.feature_net_fpn(x, self.packed(x.device))
        return hip_ops.feature_net0(x, self.packed(x.device))

    def forward_cl(self, x):
        if self.arch_mode == "unet":
            return FeatureNet0.forward_cl(self, x)
        self._require_hip(x)
        per_image = hip_ops.feature_net_fpn_workspace_bytes(1, x.shape[-2], x.shape[-1])
        chunk = max(1, int(self.workspace_limit_bytes // per_image))
        parts = [self._run_hip(x[i:i + chunk]) for i in range(0, x.shape[0], chunk)]
        return parts[0] if len(parts) == 1 else tuple(torch.cat([p[k] for p in parts], 0) for k in range(3))

    def forward(self, x):
        if self.arch_mode == "unet":
            return FeatureNet0.forward(self, x)
        self._require_hip(x)
        H, W = x.shape[-2:]
        s1, s2, s3 = self._run_hip(x)
        return {"stage1": hip_ops.unpack_features(s1, H // 4, W // 4), "stage2": hip_ops.unpack_features(s2, H // 2, W // 2),
                "stage3": hip_ops.unpack_features(s3, H, W)}

    def forward_torch(self, x):
        c0 = self.conv0(x)
        c1 = self.conv1(c0)
        c2 = self.conv2(c1)
        out = {"stage1": self.out1(c2)}
        if self.arch_mode == "unet":
            f = self.deconv1(c1, c2)
            out["stage2"] = self.out2(f)
            f = self.deconv2(c0, f)
            out["stage3"] = self.out3(f)
        else:
            f = F.interpolate(c2, scale_factor=2, mode="nearest") + self.inner1(c1)
            out["stage2"] = self.out2(f)
            f = F.interpolate(f, scale_factor=2, mode="nearest") + self.inner2(c0)
            out["stage3"] = self.out3(f)
        return out


class ConvTransReLU(_FusedLayer):
    """reference models/module.py:294-301 (parameter container; runs inside the regulariser)"""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1, output_pad=1):
        super().__init__()
        self.conv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=pad,
                                       output_padding=output_pad, bias=False)


class ConvGRUCell2(_FusedLayer):
    """reference models/module.py:54-106 (parameter container)"""

    def __init__(self, input_channel, output_channel, kernel_size):
        super().__init__()
        cin = input_channel + output_channel
        self.output_channel = output_channel
        self.gate_conv = nn.Conv2d(cin, output_channel * 2, kernel_size, padding=1)
        self.reset_gate_norm = nn.GroupNorm(1, output_channel, 1e-5, True)
        self.update_gate_norm = nn.GroupNorm(1, output_channel, 1e-5, True)
        self.output_conv = nn.Conv2d(cin, output_channel, kernel_size, padding=1)
        self.output_norm = nn.GroupNorm(1, output_channel, 1e-5, True)


class slice_RED_Regularization(PackedCache, nn.Module):
    """reference models/msrednet.py:330-366, restructured around what is and is not recurrent.

    In the reference step, conv1-3 (the encoder) depend only on the cost of the plane, every ConvGRUCell2 only on its
    own state and its encoder map, and the decoder (upconv3/2/1, upconv2d) only on the four GRU outputs.  Inside a
    cell both convolutions act on cat(x, .) and are linear, conv(cat(x, h)) = Wx.x + Wh.h + b, so their x halves are
    state-independent as well.  Per stage:
      A  cost, encoder and (levels 3, 4) the x halves of the gate / candidate convolutions for ALL planes, one batched
         launch per layer (maps indexed d * B + b),
      B  four independent recurrences over the planes (one per level, each on its own stream); per plane: levels 3, 4
         the h halves of the three convolutions (the x halves enter through the `skip` operand) on the small-grid
         form of k_conv_dd, levels 1, 2 both convolutions whole on cat(x, h) with register-resident weights
         (adamvs_conv3x3_pair); two GroupNorm reductions and the two fused epilogues,
      C  the decoder for all planes, one batched launch per layer.
    Channel counts are zero-padded to widths adamvs_conv3x3_dd takes: x maps XW, state maps HW, GRU outputs RW."""

    XC = (None, 16, 32, 64)        # x channels of levels 2-4 (level 1: in_channels)
    HC = (8, 16, 32, 64)           # state channels
    HW = (16, 16, 32, 64)          # state map widths
    RW = (16, 32, 64, 64)          # width of the stored GRU outputs = what the decoder layer reading them takes

    def __init__(self, in_channels, base_channels=8):
        super().__init__()
        c = base_channels
        assert c == 8, "the reference hard-codes 8/16/32/64-channel states (msrednet.py:388-391)"
        self.in_channels, self.base_channels = in_channels, c
        self.conv_gru1 = ConvGRUCell2(in_channels, c, 3)
        self.conv_gru2 = ConvGRUCell2(2 * c, 2 * c, 3)
        self.conv_gru3 = ConvGRUCell2(4 * c, 4 * c, 3)
        self.conv_gru4 = ConvGRUCell2(8 * c, 8 * c, 3)
        self.conv1 = ConvReLU(in_channels, 2 * c, 3, 2, 1)
        self.conv2 = ConvReLU(2 * c, 4 * c, 3, 2, 1)
        self.conv3 = ConvReLU(4 * c, 8 * c, 3, 2, 1)
        self.upconv3 = ConvTransReLU(8 * c, 4 * c, 3, 2, 1, 1)
        self.upconv2 = ConvTransReLU(4 * c, 2 * c, 3, 2, 1, 1)
        self.upconv1 = ConvTransReLU(2 * c, c, 3, 2, 1, 1)
        self.upconv2d = nn.ConvTranspose2d(c, 1, kernel_size=3, stride=1, padding=1, output_padding=0)
        self._cache_init()
        self._packed = None                    # the packed weights of the device the current forward runs on
        self._streams = None
        self.concurrent_levels = True

    def packed(self, device):
        def build():
            flat, offsets = packing.pack_red_regularization(module_state(self), "", self.in_channels)
            return (flat.to(device), offsets)
        self._packed = self.cached(device, build)
        return self._packed

    def _w(self, name):
        flat, offsets = self._packed
        o, D = offsets[name]
        return flat[o:o + 9 * D * D], flat[o + 9 * D * D:o + 9 * D * D + D]

    def _wblock(self, name):
        """A padded D x D layer as one contiguous block: 9 D D fragment floats followed by D bias floats."""
        flat, offsets = self._packed
        o, D = offsets[name]
        return flat[o:o + 9 * D * D + D]

    def _wp(self, name, cin):
        """(A fragments, bias) of a register-resident pair convolution (levels 1 and 2)."""
        flat, offsets = self._packed
        o, rows = offsets[name]
        n = rows * 9 * cin
        return flat[o:o + n], flat[o + n:o + n + rows]

    def _gn(self, k):
        flat, offsets = self._packed
        o, hc = offsets["gn%d" % (k + 1)]
        return flat[o:o + 6 * hc]

    def x_widths(self):
        return (max(packing.pad16(self.in_channels), 16), 32, 64, 64)

    @staticmethod
    def _to_width(x, n, width):
        """The first n channels of x as a `width`-wide map (zero padded); x itself when it already has that width."""
        if x.shape[-1] == width:
            return x
        out = torch.zeros(x.shape[0], x.shape[1], width, device=x.device, dtype=torch.float32)
        hip_ops.channel_copy(x, 0, out, 0, n)
        return out

    # ---- A: cost, encoder and the x halves of the deep levels' GRU convolutions for all planes --------------------------
    def cost_maps(self, feat_cl, rt, planes, B, S, h, w):
        """-> (X0 [D*B, h*w, XW0], the negated variance cost of every plane (plane-major) at the encoder's width,
        and the same map compact [D*B, h*w, C] for the level-1 cell -- X0 itself when XW0 == C)."""
        D, C, xw0 = planes.shape[1], self.in_channels, self.x_widths()[0]
        X0 = (torch.empty if xw0 == C else torch.zeros)(D * B, h * w, xw0, device=feat_cl.device, dtype=torch.float32)
        xc = X0 if xw0 == C else torch.empty(D * B, h * w, C, device=feat_cl.device, dtype=torch.float32)
        hip_ops.red_variance_cost(feat_cl, rt, planes, X0, None if xc is X0 else xc, B, S, C, D, h, w, negate=True)
        return X0, xc

    def encode(self, X0, xc, h, w):
        """-> per level what its recurrence consumes.  Levels 1, 2 (8/16-channel states at the two largest resolutions):
        the compact x map [D*B, npix, Cx]; their cells run both convolutions on cat(x, h) with register-resident weights
        (adamvs_conv3x3_pair).  Levels 3, 4: (gxr, gxu, cx) [D*B, npix, HW_k], the x halves (+ bias) of the reset /
        update / candidate convolutions for every plane."""
        N = X0.shape[0]
        xw = self.x_widths()
        xcn = (self.in_channels,) + self.XC[1:]
        X, feeds = X0, []
        for k in range(4):
            hk, wk = h >> k, w >> k
            if k < 2:
                feeds.append(xc if k == 0 else self._to_width(X, 16, 16))
            else:
                lev = []
                src = narrow if k == 2 else X      # level 3: conv2's own 32-wide output (x is exactly 32 channels)
                for name in ("gxr", "gxu", "cx"):
                    wt, bs = self._w("%s%d" % (name, k + 1))
                    lev.append(self._to_width(hip_ops.conv3x3_dd(src, wt, bs, None, N, src.shape[-1], hk, wk, 0, False),
                                              self.HC[k], self.HW[k]))
                feeds.append(lev)
            if k < 3:                  # conv_{k+1}: stride 2, ReLU; its output is the next level's x
                wt, bs = self._w("conv%d" % (k + 1))
                e = hip_ops.conv3x3_dd(X, wt, bs, None, N, xw[k], hk, wk, 1, True)
                if k == 0:
                    feeds_x1 = e       # 16 real channels at width XW0: level 2 reads them compact
                if k == 1:
                    narrow = e         # 32 channels, 32 wide: what level 3's x halves read
                X = self._to_width(e, xcn[k + 1], xw[k + 1])
            if k == 1:
                feeds[1] = self._to_width(feeds_x1, 16, 16)
        return feeds

    # ---- B: one level's recurrence over the planes ----------------------------------------------------------------
    def recur_level(self, k, feed, Rk, B, h, w):
        """ConvGRUCell2 of level k (0-based) over the planes; h' of plane d into Rk[d*B:(d+1)*B, :, :HC].  One native
        call per level: the per-plane launches (6-7 kernels) are issued from C++ (adamvs_red_recur_pair / _split)."""
        hk, wk, HC = h >> k, w >> k, self.HC[k]
        gn = self._gn(k)
        if k < 2:      # levels 1, 2: gate_conv / output_conv whole, on cat(x_d, h) / cat(x_d, r*h) (compact maps)
            Cx = feed.shape[-1]
            wg, bg = self._wp("gp%d" % (k + 1), Cx + HC)
            wc, bc = self._wp("cp%d" % (k + 1), Cx + HC)
            hip_ops.red_recur_pair(feed, wg, bg, wc, bc, gn, Rk, B, hk, wk, HC)
        else:          # levels 3, 4: the h halves per plane; the x halves (feed) enter as the convolutions' skip operand
            gxr, gxu, cx = feed
            blocks = [self._wblock("%s%d" % (n, k + 1)) for n in ("ghr", "ghu", "ch")]
            hip_ops.red_recur_split(gxr, gxu, cx, blocks[0], blocks[1], blocks[2], gn, Rk, B, hk, wk, HC)

    # ---- C: decoder for all planes --------------------------------------------------------------------------------
    def decode(self, R, B, h, w):
        """R[k] [D*B, npix_k, RW[k]] (GRU outputs in the leading channels) -> reg_cost of every plane [D*B, h*w, 16] (channel 0)."""
        N = R[0].shape[0]
        w3, b3 = self._w("upconv3")
        up3 = hip_ops.conv3x3_dd(R[3], w3, b3, R[2], N, 64, h >> 3, w >> 3, 2, True)          # relu(upconv3(reg4)) + reg3
        w2, b2 = self._w("upconv2")
        up2 = hip_ops.conv3x3_dd(self._to_width(up3, 32, 32), w2, b2, R[1], N, 32, h >> 2, w >> 2, 2, True)
        w1, b1 = self._w("upconv1")
        up1 = hip_ops.conv3x3_dd(self._to_width(up2, 16, 16), w1, b1, R[0], N, 16, h >> 1, w >> 1, 2, True)
        wf, bf = self._w("upconv2d")
        return hip_ops.conv3x3_dd(up1, wf, bf, None, N, 16, h, w, 0, False)

    def regularize_maps(self, X0, B, h, w, xc=None):
        """X0 [D*B, h*w, XW0] (-cost of every plane; xc: the same compact, when XW0 != C) -> (reg_cost maps
        [D*B, h*w, 16] (channel 0), R: the GRU outputs)."""
        if h % 8 or w % 8:
            raise AdaMVSHipError("slice_RED_Regularization: map size %dx%d must be a multiple of 8 (three stride-2 levels)" % (h, w))
        dev = X0.device
        self.packed(dev)
        N = X0.shape[0]
        halves = self.encode(X0, X0 if xc is None else xc, h, w)
        alloc = lambda k: (torch.empty if self.RW[k] == self.HC[k] else torch.zeros)       # noqa: E731  (pad channels must be 0)
        R = [alloc(k)(N, (h >> k) * (w >> k), self.RW[k], device=dev, dtype=torch.float32) for k in range(4)]
        if self.concurrent_levels:
            if self._streams is None:
                self._streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
            main = torch.cuda.current_stream()
            for k in (3, 2, 1, 0):
                self._streams[k].wait_stream(main)
                with torch.cuda.stream(self._streams[k]):
                    self.recur_level(k, halves[k], R[k], B, h, w)
            for k in range(4):
                main.wait_stream(self._streams[k])
        else:
            for k in (3, 2, 1, 0):
                self.recur_level(k, halves[k], R[k], B, h, w)
        return self.decode(R, B, h, w), R

    def regularize(self, feat_cl, rt, planes, B, S, h, w):
        """All planes of a stage: -> vol [B, D, h*w] of reg_cost (the argument of exp in msrednet.py:415)."""
        self.packed(feat_cl.device)
        X0, xc = self.cost_maps(feat_cl, rt, planes, B, S, h, w)
        fin, _ = self.regularize_maps(X0, B, h, w, xc)
        vol = torch.empty(B, planes.shape[1], h * w, device=feat_cl.device, dtype=torch.float32)
        hip_ops.planes_to_volume(fin, vol, B)
        return vol

    def forward(self, *args, **kwargs):
        raise RuntimeError("slice_RED_Regularization runs a whole stage at a time (regularize); InferDepthNet drives it")


class InferDepthNet(nn.Module):
    """reference models/msrednet.py:369-436 on channel-last features."""

    def run(self, feat_cl, B, C, h, w, rt, planes, cost_regularization):
        """feat_cl [V*B, h*w, C] view-major; rt [B,S,12]; planes [B,D,h,w] -> depth, photometric_confidence [B,h,w]."""
        S = feat_cl.shape[0] // B - 1
        D = planes.shape[1]
        vol = cost_regularization.regularize(feat_cl, rt, planes, B, S, h, w)
        return hip_ops.soft_argmin(vol, planes, B, D, h, w)

    def forward(self, *args, **kwargs):
        raise RuntimeError("InferDepthNet: call run() (channel-last features); Infer_CascadeREDNet.forward drives it")


class Infer_CascadeREDNet(nn.Module):
    """reference models/msrednet.py:440-526"""

    def __init__(self, num_depth=384, ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1], share_cr=False,
                 cr_base_chs=[8, 8, 8]):
        super().__init__()
        assert len(ndepths) == len(depth_interals_ratio) == 3
        self.num_depth = num_depth
        self.share_cr = share_cr
        self.ndepths = list(ndepths)
        self.depth_interals_ratio = list(depth_interals_ratio)
        self.cr_base_chs = cr_base_chs
        self.num_stage = len(ndepths)
        self.stage_infos = {k: {"scale": float(v)} for k, v in STAGE_SCALE.items()}
        self.feature = FeatureNet(base_channels=8, stride=4, num_stage=self.num_stage, arch_mode="unet")
        if share_cr:
            # the reference passes the LIST of widths as in_channels here (msrednet.py:463), which cannot run
            raise AdaMVSHipError("Infer_CascadeREDNet: share_cr=True is not runnable in the reference either")
        self.cost_regularization = nn.ModuleList([slice_RED_Regularization(in_channels=self.feature.out_channels[i],
                                                                           base_channels=self.cr_base_chs[i])
                                                  for i in range(self.num_stage)])
        self.DepthNet = InferDepthNet()

    def forward(self, imgs, proj_matrices, depth_values):
        if not imgs.is_cuda:
            raise AdaMVSHipError("Infer_CascadeREDNet runs on MI355X only: move the model and its inputs to the GPU")
        depth_min = float(depth_values[0, 0].cpu().numpy())         # batch item 0 only, as in the reference
        depth_max = float(depth_values[0, -1].cpu().numpy())
        depth_interval = (depth_max - depth_min) / self.num_depth
        maps, shapes = self.extract_features(imgs)
        return self.infer_from_features(maps, shapes, proj_matrices, depth_values, depth_interval)

    def extract_features(self, imgs):
        """-> channel-last stage maps [V*B, h*w, C] (view-major) and their (B, C, h, w)."""
        B, V = imgs.shape[:2]
        H, W = imgs.shape[-2:]
        x = imgs.transpose(0, 1).reshape(B * V, *imgs.shape[2:]).contiguous()
        maps = self.feature.forward_cl(x)
        return maps, [(B, maps[s].shape[-1], H // sc, W // sc) for s, sc in enumerate((4, 2, 1))]

    def infer_from_features(self, maps, shapes, proj_matrices, depth_values, depth_interval):
        """The three stages on pre-extracted features; no host synchronisation (capturable in a hipGraph)."""
        outputs, depth = {}, None
        H, W = shapes[2][2], shapes[2][3]
        for s in range(self.num_stage):
            name = "stage%d" % (s + 1)
            B, C, h, w = shapes[s]
            if depth is None:
                cur = depth_values
            else:
                # msrednet.py:495-514: previous depth to full resolution, window samples there, then down to the stage's
                # resolution.  The samples are affine in the depth map, so resampling the map first is the same thing.
                cur = hip_ops.resize_bilinear(depth, (H, W))
                if (h, w) != (H, W):
                    cur = hip_ops.resize_bilinear(cur, (h, w))
            planes = hip_ops.depth_range_samples(cur, self.ndepths[s], self.depth_interals_ratio[s] * depth_interval, [B, h, w])
            rt = hip_ops.relative_transforms(proj_matrices[name])
            depth, conf = self.DepthNet.run(maps[s], B, C, h, w, rt, planes, self.cost_regularization[s])
            st = {"depth": depth, "photometric_confidence": conf}
            outputs[name] = st
            outputs.update(st)
        return outputs
