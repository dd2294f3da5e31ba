"""`ResControlNet` operator (controlnet/flow_resnet.py) — SURVEY.md §8 row a21 (config 4): the residual-frame
variant of the ControlNet.  Same encoder / FDN / zero-conv path as `HipDualFlowControlNet`; the control pyramid is
    Bi_Dir_ResidueExtractor(prev, next, flow_fwd, flow_bwd)  +  WarpExtractor(warp_cond)      (flow_resnet.py:77-112)
with the residue extractor's own flow handling (pixel rescale ÷(512/res), learnt depthwise refinement, no hole fill:
extractors.py:149-207) and the plain conv pyramid of the warped frame (extractors.py:26-65).

Call surface (flow_resnet.py:52-64): forward(sample, timestep, encoder_hidden_states, controlnet_cond=[B,6,H,W],
flow_cond=[B,4,H,W], warp_cond=[B,3,H,W], conditioning_scale, guess_mode, return_dict).
Note: the reference never combines the two ControlNets at inference (pipeline.py takes one DualFlowControlNet, no
warp_cond); `combine_residuals` below is this package's documented rule (sum of the two residual sets), unpinned."""
import torch

from . import ops, weights
from .controlnet import HipDualFlowControlNet
from .ops import PackedConvF32


class BiDirResidueExtractor:
    """Bi_Dir_ResidueExtractor.forward — controlnet/extractors.py:149-207 (debug print at :174 dropped)."""

    def __init__(self, sd, p, device):
        def c(k):
            return PackedConvF32(sd[p + k + ".weight"], sd[p + k + ".bias"], device)

        self.pre = {s: [c(f"{s}_pre.{i}") for i in (0, 2, 4)] for s in ("prev", "next")}
        self.pyr = {s: [c(f"{s}_pyramids.{i}.0") for i in range(4)] for s in ("prev", "next")}
        # grouped 3x3 conv (2 -> 2, groups=2) = one 1->1 conv per flow component
        self.refine = []
        for i in range(4):
            w, b = sd[p + f"flow_refiners.{i}.weight"], sd[p + f"flow_refiners.{i}.bias"]
            self.refine.append([PackedConvF32(w[j:j + 1], b[j:j + 1], device) for j in range(2)])
        self.metric = [(c(f"warpers.{i}.metric_net.0"), c(f"warpers.{i}.metric_net.2")) for i in range(4)]
        self.zero = [c(f"zero_convs.{i}") for i in range(4)]

    def _pre(self, side, x):
        l = self.pre[side]
        x = ops.conv3x3_nchw_f32(x, l[0], 1, True)
        x = ops.conv3x3_nchw_f32(x, l[1], 2, True)
        return ops.conv3x3_nchw_f32(x, l[2], 2, True)

    def _refine(self, i, flow):
        parts = [ops.conv3x3_nchw_f32(flow[:, j:j + 1], self.refine[i][j], 1, False) for j in range(2)]
        return torch.cat(parts, 1)                      # plumbing: two [N,1,r,r] planes -> [N,2,r,r]

    def _warp(self, i, feat, flow, occ):
        m = ops.conv3x3_nchw_f32(feat, self.metric[i][0], 1, True)
        metric = ops.conv3x3_nchw_f32(m, self.metric[i][1], 1, False)
        return ops.splat_soft(feat, flow, metric, mask=occ), metric

    def __call__(self, prev_frame, next_frame, flow_fwd, flow_bwd):
        h = prev_frame.shape[-2]
        xp, xn = self._pre("prev", prev_frame), self._pre("next", next_frame)
        outs = []
        for i, res in enumerate((h // 8, h // 16, h // 32, h // 64)):
            xp = ops.conv3x3_nchw_f32(xp, self.pyr["prev"][i], 2, True)
            xn = ops.conv3x3_nchw_f32(xn, self.pyr["next"][i], 2, True)
            factor = float(h // res)
            ff = self._refine(i, ops.flow_resize_divide(flow_fwd, res, res, factor, factor))      # :181-187
            fb = self._refine(i, ops.flow_resize_divide(flow_bwd, res, res, factor, factor))
            occ_f = ops.occlusion_mask(ff, fb)                                                    # :189-190
            occ_b = ops.occlusion_mask(fb, ff)
            wp, cp = self._warp(i, xp, ff, occ_f)
            wn, cn = self._warp(i, xn, fb, occ_b)
            fused = ops.fuse_warped(wp, wn, cp, cn)                                               # :199-203 (no hole fill)
            outs.append(ops.conv3x3_nchw_f32(fused, self.zero[i], 1, False))
        return outs


class WarpExtractor:
    """WarpExtractor.forward — controlnet/extractors.py:50-65."""

    def __init__(self, sd, p, device):
        def c(k):
            return PackedConvF32(sd[p + k + ".weight"], sd[p + k + ".bias"], device)

        self.enc = [(c(f"enc{i}.block.0"), c(f"enc{i}.block.2")) for i in range(1, 6)]
        self.zero = [c(f"zero_convs.{i}") for i in range(4)]

    def __call__(self, x):
        feats = []
        for i, (a, b) in enumerate(self.enc):
            x = ops.conv3x3_nchw_f32(x, a, 4 if i == 0 else 2, True)
            x = ops.conv3x3_nchw_f32(x, b, 1, True)
            feats.append(x)
        return [ops.conv3x3_nchw_f32(f, z, 1, False) for f, z in zip(feats[1:], self.zero)]


class HipResControlNet(HipDualFlowControlNet):
    needs_warp_cond = True

    def __init__(self, state_dict, config=None, device="cuda"):
        super().__init__(state_dict, config, device)
        self.warp_extractor = WarpExtractor(state_dict, "warp_extractor.", device)

    def _make_extractor(self, sd, device):
        return BiDirResidueExtractor(sd, "feature_extractor.", device)

    def compute_pyramid(self, controlnet_cond, flow_cond, warp_cond=None):
        if warp_cond is None:
            raise ValueError("ResControlNet needs warp_cond [B,3,H,W] (flow_resnet.py:58)")
        cond = controlnet_cond.to(device=self.device, dtype=torch.float32).contiguous()
        flow = flow_cond.to(device=self.device, dtype=torch.float32).contiguous()
        warp = warp_cond.to(device=self.device, dtype=torch.float32).contiguous()
        p = self.feature_extractor(cond[:, :3], cond[:, 3:], flow[:, :2], flow[:, 2:])     # flow_resnet.py:80-83 (no slot swap here)
        w = self.warp_extractor(warp)
        return [ops.add_f32(a, b) for a, b in zip(p, w)]                                   # P + W, flow_resnet.py:90,106-112

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond=None, flow_cond=None, warp_cond=None,
                conditioning_scale=1.0, guess_mode=False, return_dict=True, **kw):
        if controlnet_cond is None or flow_cond is None or warp_cond is None:
            raise ValueError("controlnet_cond [B,6,H,W], flow_cond [B,4,H,W] and warp_cond [B,3,H,W] are required")
        self.set_context(encoder_hidden_states)
        self.prepare_controls(controlnet_cond, flow_cond, warp_cond)
        from .unet import as_nchw, to_nhwc_bf16
        t_dev = torch.as_tensor(timestep).to(device=self.device, dtype=torch.float32).reshape(-1)[:1].contiguous()
        down, mid = self.forward_nhwc(to_nhwc_bf16(sample.to(self.device)), t_dev, float(conditioning_scale))
        return ([as_nchw(d) for d in down], as_nchw(mid))

    __call__ = forward


def combine_residuals(flow_net_out, res_net_out):
    """Config-4 'dual ControlNet' rule of this package (the reference has none in-repo; parity unpinned):
    residuals of the flow ControlNet and of the residual ControlNet add, like diffusers' MultiControlNetModel.  Generic
    loop of `StableDiffusionDualFlowControlNetPipeline` (module-level calls); the fused loop performs the same sum inside
    the zero-conv GEMM epilogues (`HipUNet2DConditionModel.decode_nhwc(control=[...])`)."""
    d1, m1 = flow_net_out
    d2, m2 = res_net_out

    def add(a, b):
        if a.is_cuda and a.dtype == torch.bfloat16 and a.permute(0, 2, 3, 1).is_contiguous() and b.permute(0, 2, 3, 1).is_contiguous():
            return ops.add_bf16(a.permute(0, 2, 3, 1), b.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return a + b

    return [add(a, b) for a, b in zip(d1, d2)], add(m1, m2)
