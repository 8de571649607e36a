"""Drop-in `Infer_CascadeREDNet` for MI355X: constructor, forward() signature, output dict and state-dict keys of
reference models/msrednet.py:440-526 (`predict_whu.py --model msrednet`).

Per hypothesis plane: the variance cost of the reference and the warped source features (msrednet.py:396-412), the
four-level GroupNorm ConvGRU encoder-decoder (slice_RED_Regularization, msrednet.py:330-366) and the running
exp-sum / max / weighted-depth update (msrednet.py:415-436).  Every convolution runs on the fp32-MFMA k_conv_dd
kernel (adamvs_conv3x3_dd) over channel-last maps whose channel count is zero-padded to a supported width; the
GroupNorm statistics, gate / candidate epilogues and the variance cost are the kernels of csrc/msred.hip; FeatureNet
is adamvs_feature_net0 with zero context-branch weights.  No CPU fallback.
"""
import torch
import torch.nn as nn

from ada_mvs_amd import hip_ops, packing
from ada_mvs_amd._lib import AdaMVSHipError
from .adamvs import STAGE_SCALE, FeatureNet0
from .module import Conv2d, ConvReLU, DeConv2dFuse, _FusedLayer


class FeatureNet(FeatureNet0):
    """The plain U-Net of reference models/msrednet.py:29-127 (arch_mode 'unet', three stages): FeatureNet0 without
    the pooled-context branches."""

    def __init__(self, base_channels, num_stage=3, stride=4, arch_mode="unet"):
        nn.Module.__init__(self)
        assert arch_mode == "unet" and num_stage == 3, "this build implements arch_mode 'unet' with three stages"
        c = base_channels
        self.arch_mode, self.stride, self.base_channels, self.num_stage = arch_mode, stride, c, num_stage
        self.conv0 = nn.Sequential(Conv2d(3, c, 3, 1, padding=1), Conv2d(c, c, 3, 1, padding=1))
        self.conv1 = nn.Sequential(Conv2d(c, 2 * c, 5, stride=2, padding=2), Conv2d(2 * c, 2 * c, 3, 1, padding=1),
                                   Conv2d(2 * c, 2 * c, 3, 1, padding=1))
        self.conv2 = nn.Sequential(Conv2d(2 * c, 4 * c, 5, stride=2, padding=2), Conv2d(4 * c, 4 * c, 3, 1, padding=1),
                                   Conv2d(4 * c, 4 * c, 3, 1, padding=1))
        self.out1 = nn.Conv2d(4 * c, 4 * c, 1, bias=False)
        self.deconv1 = DeConv2dFuse(4 * c, 2 * c, 3)
        self.deconv2 = DeConv2dFuse(2 * c, c, 3)
        self.out2 = nn.Conv2d(2 * c, 2 * c, 1, bias=False)
        self.out3 = nn.Conv2d(c, c, 1, bias=False)
        self.out_channels = [4 * c, 2 * c, c]
        self._packed = None
        self.workspace_limit_bytes = 32 << 30

    def packed(self, device):
        if self._packed is None or self._packed.buf.device != device:
            flat, offsets = packing.pack_feature_net(self.state_dict(), "", context=False)
            self._packed = hip_ops.PackedFeature(flat, offsets, device)
        return self._packed

    def forward_torch(self, x):
        c0 = self.conv0(x)
        c1 = self.conv1(c0)
        c2 = self.conv2(c1)
        out = {"stage1": self.out1(c2)}
        f = self.deconv1(c1, c2)
        out["stage2"] = self.out2(f)
        f = self.deconv2(c0, f)
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


class _Level:
    """Buffers of one resolution level: a = cat(x, h), xr = cat(x, r*h), f / o = convolution outputs, u."""

    def __init__(self, B, npix, D, Cx, HC, dev):
        z = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)      # noqa: E731
        self.D, self.Cx, self.HC, self.npix = D, Cx, HC, npix
        self.a, self.xr, self.f, self.o = z(B, npix, D), z(B, npix, D), z(B, npix, D), z(B, npix, D)
        self.u = z(B, npix, HC)
        self.stats = z(B, 2, 2)
        self.stats_o = z(B, 2)


class slice_RED_Regularization(nn.Module):
    """reference models/msrednet.py:330-366; the state lives in the level buffers of `begin()`."""

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
        self._packed = None

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._packed = None
        return super()._load_from_state_dict(*a, **k)

    def packed(self, device):
        if self._packed is None or self._packed[0].device != device:
            flat, offsets = packing.pack_red_regularization(self.state_dict(), "", self.in_channels)
            self._packed = (flat.to(device), offsets)
        return self._packed

    def _w(self, name):
        flat, offsets = self._packed
        o, D = offsets[name]
        return flat[o:o + 9 * D * D], flat[o + 9 * D * D:o + 9 * D * D + D]

    def begin(self, B, h, w, device):
        """Zero states and buffers for a stage of h x w maps (h, w multiples of 8)."""
        if h % 8 or w % 8:
            raise AdaMVSHipError("slice_RED_Regularization: map size %dx%d must be a multiple of 8 (three stride-2 levels)" % (h, w))
        self.packed(device)
        C = self.in_channels
        self.B, self.h, self.w = B, h, w
        xw, hw = (C, 16, 32, 64), (8, 16, 32, 64)
        self.lv = [_Level(B, (h >> k) * (w >> k), packing.pad16(xw[k] + hw[k]), xw[k], hw[k], device) for k in range(4)]
        z = lambda *s: torch.zeros(*s, device=device, dtype=torch.float32)      # noqa: E731
        n = [l.npix for l in self.lv]
        self.enc = [None] + [z(B, n[k], self.lv[k - 1].D) for k in (1, 2, 3)]    # conv_k output at level k+1, width D_k
        self.r4 = z(B, n[3], 64)                    # reg_cost4 -> upconv3
        self.skip = [z(B, n[0], 16), z(B, n[1], 32), z(B, n[2], 64)]             # reg_cost1..3 as decoder skips
        self.up3, self.up3n = z(B, n[2], 64), z(B, n[2], 32)
        self.up2, self.up2n = z(B, n[1], 32), z(B, n[1], 16)
        self.up1, self.fin = z(B, n[0], 16), z(B, n[0], 16)
        self.gn_ws = hip_ops.group_stats_workspace(B, 2, device)
        self.gn = [self._gn(k) for k in range(4)]

    def _gn(self, k):
        flat, offsets = self._packed
        o, hc = offsets["gn%d" % (k + 1)]
        return flat[o:o + 6 * hc]

    def cost_targets(self):
        """Where the (negated) cost of the next plane goes: the x part of level 1's two cat buffers."""
        return self.lv[0].a, self.lv[0].xr

    def _gru(self, k, out2, c2):
        """ConvGRUCell2 of level k (0-based) on a = cat(x, h); h' replaces h in a and goes to out2[..., c2:c2+HC]."""
        L, B = self.lv[k], self.B
        hk, wk = self.h >> k, self.w >> k
        wg, bg = self._w("gates%d" % (k + 1))
        hip_ops.conv3x3_dd(L.a, wg, bg, None, B, L.D, hk, wk, 0, False, out=L.f)
        hip_ops.group_stats(L.f, 0, L.HC, 2, L.stats, self.gn_ws)
        hip_ops.gru2_gates_apply(L.f, L.stats, self.gn[k], L.a, L.xr, L.u, L.Cx, L.HC)
        wc, bc = self._w("cand%d" % (k + 1))
        hip_ops.conv3x3_dd(L.xr, wc, bc, None, B, L.D, hk, wk, 0, False, out=L.o)
        hip_ops.group_stats(L.o, 0, L.HC, 1, L.stats_o, self.gn_ws)
        hip_ops.gru2_out_apply(L.o, L.stats_o, self.gn[k][4 * L.HC:], L.u, L.a, out2, c2, L.Cx, L.HC)

    def step(self, vol, d):
        """One plane: the -cost is already in the level-1 buffers (cost_targets); writes reg_cost into vol[:, d]."""
        B, h, w, lv = self.B, self.h, self.w, self.lv
        # encoder: conv_k reads the cat buffer of level k (zero weights on the state channels), stride 2, ReLU
        for k in (1, 2, 3):
            wk_, bk_ = self._w("conv%d" % k)
            hip_ops.conv3x3_dd(lv[k - 1].a, wk_, bk_, None, B, lv[k - 1].D, h >> (k - 1), w >> (k - 1), 1, True, out=self.enc[k])
            hip_ops.channel_copy(self.enc[k], 0, lv[k].a, 0, lv[k].Cx)
            hip_ops.channel_copy(self.enc[k], 0, lv[k].xr, 0, lv[k].Cx)
        self._gru(3, self.r4, 0)
        self._gru(2, self.skip[2], 0)
        w3, b3 = self._w("upconv3")
        hip_ops.conv3x3_dd(self.r4, w3, b3, self.skip[2], B, 64, h >> 3, w >> 3, 2, True, out=self.up3)      # relu(upconv3) + reg3
        hip_ops.channel_copy(self.up3, 0, self.up3n, 0, 32)
        self._gru(1, self.skip[1], 0)
        w2, b2 = self._w("upconv2")
        hip_ops.conv3x3_dd(self.up3n, w2, b2, self.skip[1], B, 32, h >> 2, w >> 2, 2, True, out=self.up2)
        hip_ops.channel_copy(self.up2, 0, self.up2n, 0, 16)
        self._gru(0, self.skip[0], 0)
        w1, b1 = self._w("upconv1")
        hip_ops.conv3x3_dd(self.up2n, w1, b1, self.skip[0], B, 16, h >> 1, w >> 1, 2, True, out=self.up1)
        wf, bf = self._w("upconv2d")
        hip_ops.conv3x3_dd(self.up1, wf, bf, None, B, 16, h, w, 0, False, out=self.fin)
        hip_ops.plane_to_volume(self.fin, vol, d)

    def forward(self, *args, **kwargs):
        raise RuntimeError("slice_RED_Regularization runs plane by plane inside InferDepthNet (begin / step)")


class InferDepthNet(nn.Module):
    """reference models/msrednet.py:369-436 on channel-last features."""

    def run(self, feat_cl, B, C, h, w, rt, planes, cost_regularization):
        """feat_cl [V*B, h*w, C] view-major; rt [B,S,12]; planes [B,D,h,w] -> depth, photometric_confidence [B,h,w]."""
        dev = feat_cl.device
        S = feat_cl.shape[0] // B - 1
        D = planes.shape[1]
        reg = cost_regularization
        reg.begin(B, h, w, dev)
        vol = torch.empty(B, D, h * w, device=dev, dtype=torch.float32)
        a, xr = reg.cost_targets()
        for d in range(D):
            plane = planes[:, d].contiguous()
            hip_ops.red_variance_cost(feat_cl, rt, plane, a, xr, B, S, C, h, w, negate=True)
            reg.step(vol, d)
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
        B, V = imgs.shape[:2]
        H, W = imgs.shape[-2:]
        x = imgs.transpose(0, 1).reshape(B * V, *imgs.shape[2:]).contiguous()
        maps = self.feature.forward_cl(x)
        outputs, depth = {}, None
        for s in range(self.num_stage):
            name = "stage%d" % (s + 1)
            scale = STAGE_SCALE[name]
            h, w = H // scale, W // scale
            if depth is None:
                planes = hip_ops.depth_range_samples(depth_values, self.ndepths[s], self.depth_interals_ratio[s] * depth_interval,
                                                     [B, h, w])
            else:
                # msrednet.py:495-514: previous depth to full resolution, window samples there, then down to the stage's
                # resolution.  The samples are affine in the depth map, so resampling the map first is the same thing.
                cur = hip_ops.resize_bilinear(depth, (H, W))
                if scale != 1:
                    cur = hip_ops.resize_bilinear(cur, (h, w))
                planes = hip_ops.depth_range_samples(cur, self.ndepths[s], self.depth_interals_ratio[s] * depth_interval, [B, h, w])
            rt = hip_ops.relative_transforms(proj_matrices[name])
            depth, conf = self.DepthNet.run(maps[s], B, maps[s].shape[-1], h, w, rt, planes, self.cost_regularization[s])
            st = {"depth": depth, "photometric_confidence": conf}
            outputs[name] = st
            outputs.update(st)
        return outputs
