"""Thin host wrappers: torch tensors (device memory + stream plumbing) -> C-ABI launchers of libdiffcodec_hip.so.

Activations are NHWC bf16 tensors of shape [N,H,W,C] (a [B,L,C] token sequence is the same memory with H=1, W=L).
No arithmetic happens in torch on this path; every function below ends in exactly one or more `lib.call`s."""
import math

import torch

from . import lib
from .lib import ConvDesc

BF16 = torch.bfloat16
F32 = torch.float32


def _meta(family, shape, flops, nbytes):
    """(family, shape label, algorithmic FLOPs, algorithmic HBM bytes) for lib.LaunchTimer — only built while a timer is set."""
    return (family, shape, float(flops), float(nbytes)) if lib.TIMER is not None else None



GN_DIRECT_MAX_PIXELS = 256   # maps up to 16x16: one-launch GroupNorm statistics
# GroupNorm statistics emitted by the producing conv / GEMM epilogue instead of a read pass.  These switches are plain module
# constants: tools/ assign them for an A/B, nothing on the product path reads the environment.
GN_EPILOGUE_STATS = True


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _chk(t, dtype, name):
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a device tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected contiguous memory")
    return t


# ------------------------------------------------------------------------------------------ weights
class PackedConv:
    """Conv / linear weight repacked for the device kernels.
    kind 'igemm':      w bf16 [Cout][k*k][Cin]
    kind 'small_cin':  w bf16 [k*k][Cin][Cout]
    kind 'small_cout': w bf16 [Cout][k*k][Cin]"""

    def __init__(self, weight, bias, device, geglu=False, mfma_small_cout=False, ln=None):
        """mfma_small_cout: route a Cout <= 8 (multiple of 4) conv through the MFMA tile kernels anyway — one mostly
        empty N-tile, still several times faster than the VALU direct conv when M*K is large (UNet conv_out).
        ln = (gamma, beta, eps) of an nn.LayerNorm that feeds this nn.Linear: folded into the weights at load time,
        W' = W diag(gamma), b' = b + W beta (fp32, then the usual bf16 rounding of W'), plus colsum(W') of the ROUNDED
        weights so that Linear(LN(x)) = rstd * (x W'^T - mean * colsum) + b' holds exactly for the stored operands
        (dc_conv_desc.ln_stats / ln_colsum).  Call `conv(..., ln_stats=row statistics of x)`."""
        w = weight.detach().float()
        if w.dim() == 2:
            w = w[:, :, None, None]
        cout, cin, kh, kw = w.shape
        assert kh == kw and kh in (1, 3)
        self.cout, self.cin, self.ksize, self.geglu = cout, cin, kh, geglu
        b = None if bias is None else bias.detach().float()
        self.ln_eps = None
        if ln is not None:
            assert kh == 1 and cin % 64 == 0 and cout % 16 == 0
            gamma, beta, eps = ln
            gamma, beta = gamma.detach().float().cpu(), beta.detach().float().cpu()
            wb = w[:, :, 0, 0].cpu() @ beta
            b = wb if b is None else b.cpu() + wb
            w = w.cpu() * gamma[None, :, None, None]
            self.ln_eps = float(eps)
        if cin <= 16:
            self.kind = "small_cin"
            wp = w.permute(2, 3, 1, 0).reshape(kh * kw, cin, cout)
        elif cout <= 8 and not (mfma_small_cout and cout % 4 == 0 and cin % 64 == 0):
            self.kind = "small_cout"
            wp = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
        else:
            self.kind = "igemm"
            if cin % 64 or (cout % 16 and not mfma_small_cout):
                raise ValueError(f"igemm needs Cin%64==0 and Cout%16==0, got {cin}->{cout}")
            wp = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
            if geglu:
                # rows interleaved in blocks of 16 hidden | 16 gate so both halves of a GEGLU pair share a wave
                f = cout // 2
                assert f % 16 == 0
                idx = torch.arange(cout).reshape(2, f // 16, 16).permute(1, 0, 2).reshape(-1)
                wp = wp[idx]
                if b is not None:
                    b = b[idx]
        self.w = wp.contiguous().to(device=device, dtype=BF16)
        self.bias = None if b is None else b.contiguous().to(device=device, dtype=F32)
        self.colsum = None
        if ln is not None:           # sums of the stored (rounded, possibly GEGLU-interleaved) rows, fp64 accumulation on the host
            self.colsum = self.w.detach().cpu().double().sum(dim=(1, 2)).float().contiguous().to(device)


class PackedConvF32:
    """fp32 OIHW weights of the control extractors (kept in checkpoint layout)."""

    def __init__(self, weight, bias, device):
        self.w = weight.detach().float().contiguous().to(device)
        self.bias = None if bias is None else bias.detach().float().contiguous().to(device)
        self.cout, self.cin = weight.shape[0], weight.shape[1]


# ------------------------------------------------------------------------------------------ conv / linear
def _pick_splitk(m, cout, kt, units=None, rows_per_image=0):
    """Split the K loop over workgroups when the output has too few tiles to fill 256 CUs and K is long (the
    weight-streaming layers at 8x8 / 16x16).  Each split stores its partial tile into its own fp32 slab and a finish
    pass sums the slabs (m*cout*4 bytes written and read once per split: cheap next to the weight stream, and
    deterministic).  `units` = what the launched kernel partitions (64-channel chunks for the halo-tile 3x3 kernel,
    64-wide K steps otherwise); an even partition is preferred, so the split is a divisor of `units`."""
    tile3 = units is not None
    units = kt if units is None else units
    bn = 160 if cout % 160 == 0 else 128
    tiles = math.ceil(m / 64) * math.ceil(cout / bn)
    # (per-shape sweep of every decode shape at model batches 32 and 2, round 3: short K loops are split only at tiny m, where nothing
    # else fills the chip; a 1x1 launch with >= 2048 rows needs >= 128 K steps before a split pays)
    if tiles >= 512 or (kt < 64 and (m >= 2048 or kt < 32)) or (not tile3 and m >= 2048 and kt < 128):
        return 1
    divs = [s for s in range(1, min(units, 20) + 1) if units % s == 0 and kt // s >= 8]
    target = 1024 if m >= 2048 else 256      # measured (tools/bench_splitk.py): ~1 workgroup per CU at small m, 2+ above
    if tile3 and rows_per_image >= 1024:     # 32x32 and 64x64 maps of a one- or two-frame decode (the halo-tile kernel's own pixel tiles
        target = 256                         # already give 64-256 workgroups): one workgroup per CU, not four (38 -> 26 us at 2x32x32x640)
    for s in divs:
        if tiles * s >= target:
            return s
    return divs[-1] if divs else 1


def _pick_splitk_strided(m, cout, kt):
    """The stride-2 Downsample2D convs (gather GEMM, 64-row tiles): measured per shape (tools/ab/bench_down.py, model batches 32 and 2) the
    best split is the smallest divisor of the K steps that yields ~1024 workgroups with at least 8 steps each — 1 / 2 / 4 at the three
    levels of a 16-frame step (the general rule gave 1 / 1 / 3: 136 -> 109 us at 32x32x640), 5 at the top level of a one-frame decode
    (53 -> 25 us), which the general rule leaves unsplit because its K loop is short."""
    bn = 160 if cout % 160 == 0 else 128
    tiles = math.ceil(m / 64) * math.ceil(cout / bn)
    divs = [s for s in range(1, min(kt, 20) + 1) if kt % s == 0 and kt // s >= 8]
    for s in divs:
        if tiles * s >= 1024:
            return s
    return divs[-1] if divs else 1


def row_stats_parts(cout):
    """Partials per row written by a 1x1 / linear launch with `stats_out` (dc_gemm_row_stats_parts)."""
    return lib.load().dc_gemm_row_stats_parts(int(cout))


def row_stats(x):
    """(sum, sum of squares) of every row of x [..., C] bf16 -> fp32 [rows, 1, 2]: the `ln_stats` operand of a linear with a
    folded LayerNorm when the launch that produced x could not emit the statistics itself."""
    _chk(x, BF16, "x")
    c = x.shape[-1]
    m = x.numel() // c
    st = torch.empty((m, 1, 2), device=x.device, dtype=F32)
    lib.call("dc_row_stats_bf16", x.data_ptr(), st.data_ptr(), m, c, _stream(),
             meta=_meta("row_stats_kernel (LayerNorm statistics)", f"M={m} C={c}", 3.0 * x.numel(), 2.0 * x.numel()))
    return st


def ln_finalize(partials, c, eps):
    """row-statistics partials [M, parts, 2] -> (mean, rstd) [M, 2] of a LayerNorm over c channels: the `ln_stats` operand."""
    _chk(partials, F32, "partials")
    m, parts, two = partials.shape
    assert two == 2
    mr = torch.empty((m, 2), device=partials.device, dtype=F32)
    lib.call("dc_ln_finalize", partials.data_ptr(), mr.data_ptr(), m, parts, int(c), float(eps), _stream())
    return mr


def rowpanel_takes(rows, rows_per_sample, cin, cout):
    """Mirror of dc_gemm_rowpanel_wanted (csrc/gemm_rowpanel.hip) for plain / residual / folded-LN 1x1 launches: the K = 320 kernel that
    keeps a 256-row panel in registers — the only GEMM of the family that can apply a GroupNorm affine on load (`gn_ab` without SiLU).
    The C side stays the authority: a launch this rule admits and the library does not take fails with DC_ERR_INVALID."""
    return cin == 320 and cout >= 320 and cout % 64 == 0 and rows >= 65536 and rows % 256 == 0 and rows_per_sample % 256 == 0


def conv(x1, pc, *, x2=None, gn_ab=None, gn_silu=False, row_add=None, residual=None, stride=1, pad=1,
         upsample=False, out_scale=1.0, out_f32=False, splitk=None, act=0, out=None, ln_stats=None, stats_out=None,
         gn_part=False, ln_partials=None):
    """F.conv2d (k=1|3) / nn.Linear on NHWC bf16 with the fusions of `dc_conv_desc`.  `out`: optional preallocated
    contiguous destination (e.g. one batch half of a larger buffer) on the igemm path.
    ln_stats: (mean, rstd) [M, 2] of the rows of x1 (`ln_finalize`) for a `pc` built with ln=(gamma, beta, eps): LayerNorm folded
    into this linear.
    stats_out: fp32 [M, row_stats_parts(cout), 2] to receive the row statistics of the OUTPUT (the next block's ln_stats).
    ln_partials: (partials [M, parts, 2], eps) — the RAW row statistics a producer's `stats_out` (or `row_stats`) wrote, instead of
    `ln_stats`: the LayerNorm finalize pass is folded into this launch (dc_conv_desc.ln_parts; same bits as
    `ln_stats=ln_finalize(partials, cin, eps)`), which saves a launch per LayerNorm wherever the row-panel kernel takes the GEMM.
    gn_part: ask the epilogue for the GroupNorm partial sums of the OUTPUT (dc_conv_desc.gn_part_out); when the launch can
    emit them the returned tensor carries them as `.gn_part` ([chunks, N, Cout, 2] fp32) and `group_norm_ab` skips its read
    pass over the tensor; otherwise the request is ignored and `group_norm_ab` measures the tensor as before."""
    _chk(x1, BF16, "x1")
    n, h, w, c1 = x1.shape
    c2 = 0
    if x2 is not None:
        _chk(x2, BF16, "x2")
        assert x2.shape[:3] == x1.shape[:3]
        c2 = x2.shape[3]
    assert c1 + c2 == pc.cin, f"channel mismatch {c1}+{c2} vs {pc.cin}"
    k = pc.ksize
    if k == 1:
        ho, wo = h, w
    else:
        hin, win = (2 * h, 2 * w) if upsample else (h, w)
        ho = (hin + (2 if pad else 1) - 3) // stride + 1
        wo = (win + (2 if pad else 1) - 3) // stride + 1
    if pc.kind == "small_cin":
        assert x2 is None and gn_ab is None and residual is None and row_add is None and not upsample and not out_f32 and out is None
        out = torch.empty((n, ho, wo, pc.cout), device=x1.device, dtype=BF16)
        lib.call("dc_conv_small_cin_bf16", x1.data_ptr(), pc.w.data_ptr(), _ptr(pc.bias), out.data_ptr(), n, h, w, c1,
                 pc.cout, k, stride, pad if k == 3 else 0, ho, wo, _stream())
        if out_scale != 1.0:
            raise ValueError("out_scale unsupported on the small-cin path")
        return out
    if pc.kind == "small_cout":
        assert x2 is None and residual is None and row_add is None and not upsample and stride == 1 and (k == 1 or pad == 1) and out is None
        out = torch.empty((n, h, w, pc.cout), device=x1.device, dtype=F32 if out_f32 else BF16)
        gb = 0 if gn_ab is None else gn_ab.shape[0]
        lib.call("dc_conv_small_cout_bf16", x1.data_ptr(), pc.w.data_ptr(), _ptr(pc.bias), _ptr(gn_ab), int(gn_silu), gb,
                 out.data_ptr(), int(out_f32), n, h, w, c1, pc.cout, k, _stream())
        return out
    cout_eff = pc.cout // 2 if pc.geglu else pc.cout
    if out is None:
        out = torch.empty((n, ho, wo, cout_eff), device=x1.device, dtype=F32 if out_f32 else BF16)
    else:
        assert out.is_contiguous() and out.numel() == n * ho * wo * cout_eff and out.dtype == (F32 if out_f32 else BF16)
        out = out.view(n, ho, wo, cout_eff)
    m = n * ho * wo
    kt = (9 if k == 3 else 1) * (pc.cin // 64)
    ln_parts, ln_eps, ln_scratch = 0, 0.0, None
    if ln_partials is not None:
        if ln_stats is not None:
            raise ValueError("pass ln_stats or ln_partials, not both")
        ln_stats, ln_eps = ln_partials
        _chk(ln_stats, F32, "ln_partials")
        assert ln_stats.dim() == 3 and ln_stats.shape[0] == m and ln_stats.shape[2] == 2
        ln_parts = int(ln_stats.shape[1])
        # scratch for (mean, rstd): only the tile kernels need it — the row-panel kernel finalizes in its prologue (the C side mirrors
        # `rowpanel_takes` and returns DC_ERR_INVALID if the two ever disagree)
        takes_rp = (k == 1 and x2 is None and (gn_ab is None or not gn_silu) and pc.bias is not None and not out_f32
                    and ln_parts <= 16 and rowpanel_takes(m, ho * wo, pc.cin, pc.cout))
        ln_scratch = None if takes_rp else torch.empty((m, 2), device=x1.device, dtype=F32)
    if ln_stats is not None or stats_out is not None:
        splitk = 1                          # the folded LayerNorm / row statistics live in the unsplit bf16 epilogue
    if splitk is None:
        tile3 = k == 3 and stride == 1 and pad == 1 and (wo % 16 == 0 and ho % 4 == 0 or wo == 8 and ho % 8 == 0 and not upsample)
        if k == 3 and stride == 2 and not pc.geglu:
            splitk = _pick_splitk_strided(m, pc.cout, kt)
        else:
            splitk = 1 if pc.geglu else _pick_splitk(m, pc.cout, kt, pc.cin // 64 if tile3 else None, rows_per_image=ho * wo)
    ws = torch.empty((splitk, m, pc.cout), device=x1.device, dtype=F32) if splitk > 1 else None
    if gn_ab is not None:
        _chk(gn_ab, F32, "gn_ab")
        assert gn_ab.shape[1] == pc.cin
    if residual is not None:
        _chk(residual, BF16, "residual")
        assert residual.numel() == m * pc.cout
    if (ln_stats is None) != (pc.ln_eps is None):
        raise ValueError("ln_stats must be given exactly when the weights carry a folded LayerNorm")
    if ln_stats is not None:
        _chk(ln_stats, F32, "ln_stats")
        assert (ln_parts > 0 or tuple(ln_stats.shape) == (m, 2)) and x2 is None and gn_ab is None and k == 1
    if stats_out is not None:
        _chk(stats_out, F32, "stats_out")
        assert tuple(stats_out.shape) == (m, row_stats_parts(pc.cout), 2) and k == 1 and not pc.geglu and not out_f32
    ras = 0
    if row_add is not None:
        assert row_add.dtype == F32 and row_add.is_cuda and row_add.shape == (n, pc.cout) and row_add.stride(1) == 1
        ras = row_add.stride(0)
    d = ConvDesc(x1=x1.data_ptr(), x2=_ptr(x2), w=pc.w.data_ptr(), bias=_ptr(pc.bias), gn_ab=_ptr(gn_ab),
                 row_add=_ptr(row_add), residual=_ptr(residual), out=out.data_ptr(), splitk_ws=_ptr(ws),
                 N=n, H=h, W=w, C1=c1, C2=c2, Cout=pc.cout, ksize=k, stride=stride, pad=int(pad), upsample=int(upsample),
                 Ho=ho, Wo=wo, gn_silu=int(gn_silu), epilogue=1 if pc.geglu else 0, out_f32=int(out_f32),
                 out_scale=float(out_scale), splitk=int(splitk), gn_batch=0 if gn_ab is None else gn_ab.shape[0],
                 act=int(act), row_add_stride=int(ras), ln_stats=_ptr(ln_stats), ln_colsum=_ptr(pc.colsum if ln_stats is not None else None),
                 stats_out=_ptr(stats_out), gn_part_out=0, ln_parts=ln_parts, ln_eps=float(ln_eps), ln_scratch=_ptr(ln_scratch))
    part = None
    if gn_part and GN_EPILOGUE_STATS and not out_f32 and not pc.geglu and splitk == 1 and ln_stats is None:
        chunks = lib.load().dc_conv_gn_part_chunks(d)
        if chunks > 0:
            part = torch.empty((chunks, n, pc.cout, 2), device=x1.device, dtype=F32)
            d.gn_part_out = part.data_ptr()
    meta = None
    if lib.TIMER is not None:
        kk = pc.cin * k * k
        if k == 1 and (gn_ab is None or (not gn_silu and x2 is None and rowpanel_takes(m, ho * wo, pc.cin, pc.cout))):
            fam = "gemm_dma_kernel + gemm_wide_kernel + gemm_p8_kernel + gemm_rowpanel_kernel (1x1 conv / linear GEMM family)"
        elif k == 3 and stride == 1 and pad == 1 and (wo % 16 == 0 and ho % 4 == 0 or wo == 8 and ho % 8 == 0 and not upsample):
            fam = "conv3x3_tile_kernel (3x3 stride-1 halo-tile conv)"
        else:
            fam = "igemm_kernel (strided / GN-on-load gather GEMM)"
        # algorithmic bytes: every operand once — activations (low-res input for the fused upsample), weights, output, residual
        nbytes = 2.0 * (n * h * w * pc.cin + pc.cout * kk + m * cout_eff * (2 if out_f32 else 1) + (m * pc.cout if residual is not None else 0))
        meta = _meta(fam, f"{k}x{k} s{stride} up{int(upsample)} M={m} N={pc.cout} K={kk} gn={int(gn_ab is not None)} geglu={int(pc.geglu)} "
                          f"ln={int(ln_stats is not None)} splitk={splitk}", 2.0 * m * pc.cout * kk, nbytes)
    lib.call("dc_conv_igemm_bf16", d, _stream(), meta=meta)
    if part is not None:
        out.gn_part = part
    return out


# GroupNorm+SiLU in front of a conv: folded into the conv's load stage only when the conv has a single output-channel
# tile (Cout <= 160).  With more N-tiles every workgroup of a pixel tile would redo the same exp/rcp work (Cout/160 x
# the 1.4x halo overlap); one HBM-bound elementwise pass (which also resolves the skip concat) is cheaper.
FUSE_GN_MAX_COUT = 160
# ... except at model batches of at most this many samples (one- and two-frame decodes), where every launch is latency-bound and
# the separate pass costs a launch of its own: fused everywhere, 175 -> 170 ms per single frame
FUSE_GN_SMALL_BATCH = 4


def conv_gn_silu(x, pc, ab, x2=None, **kw):
    if pc.kind == "igemm" and pc.ksize == 3 and pc.cout > FUSE_GN_MAX_COUT and x.shape[0] > FUSE_GN_SMALL_BATCH:
        return conv(gn_apply(x, ab, silu=True, x2=x2), pc, **kw)
    return conv(x, pc, x2=x2, gn_ab=ab, gn_silu=True, **kw)


def linear(x, pc, **kw):
    """nn.Linear on rows: x [..., Cin] bf16 -> [..., Cout]."""
    shp = x.shape
    y = conv(x.reshape(1, 1, -1, shp[-1]), pc, **kw)
    return y.reshape(*shp[:-1], y.shape[-1])


def conv3x3_nchw_f32(x, pc, stride=1, silu=False):
    """fp32 NCHW conv of the control extractors; x may be a channel-slice view of a contiguous NCHW tensor."""
    assert x.dtype == F32 and x.is_cuda
    n, c, h, w = x.shape
    assert x.stride(3) == 1 and x.stride(2) == w and x.stride(1) == h * w, "need dense CHW planes"
    assert stride in (1, 2, 4)
    ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    y = torch.empty((n, pc.cout, ho, wo), device=x.device, dtype=F32)
    lib.call("dc_conv3x3_nchw_f32", x.data_ptr(), x.stride(0), pc.w.data_ptr(), _ptr(pc.bias), y.data_ptr(), n, c, h, w,
             pc.cout, stride, int(silu), _stream(),
             meta=_meta("conv3x3_f32_mfma_kernel + conv3x3_nchw_f32_kernel (fp32 extractor conv: MFMA form / VALU form)", f"N={n} {c}->{pc.cout} {h}x{w} s{stride}",
                        2.0 * y.numel() * c * 9, 4.0 * (n * c * h * w + y.numel() + pc.w.numel())))
    return y


# ------------------------------------------------------------------------------------------ norms
def gn_stats(x):
    """-> (partials [chunks,N,C,2] fp32, chunks)"""
    _chk(x, BF16, "x")
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    chunks = lib.load().dc_gn_stats_chunks(hw, c)
    part = torch.empty((chunks, n, c, 2), device=x.device, dtype=F32)
    lib.call("dc_gn_stats_nhwc_bf16", x.data_ptr(), part.data_ptr(), n, hw, c, _stream(),
             meta=_meta("gn_stats_kernel (GroupNorm statistics)", f"N={n} HW={hw} C={c}", 3.0 * x.numel(), 2.0 * x.numel()))
    return part


def gn_finalize(part1, gamma, beta, groups, hw, eps, part2=None):
    ch1, n, c1 = part1.shape[0], part1.shape[1], part1.shape[2]
    ch2, c2 = (0, 0) if part2 is None else (part2.shape[0], part2.shape[2])
    ab = torch.empty((n, c1 + c2, 2), device=part1.device, dtype=F32)
    lib.call("dc_gn_finalize", part1.data_ptr(), c1, ch1, _ptr(part2), c2, ch2, _ptr(gamma), _ptr(beta), ab.data_ptr(), n,
             groups, hw, float(eps), _stream())
    return ab


def group_norm_ab(x, gamma, beta, groups, eps, x2=None):
    """GroupNorm statistics of cat[x, x2] folded with the affine into per-(sample,channel) (scale, shift)."""
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    c2 = 0 if x2 is None else x2.shape[-1]
    p1 = getattr(x, "gn_part", None)
    p2 = None if x2 is None else getattr(x2, "gn_part", None)
    if p1 is not None and (x2 is None or p2 is not None):        # the producers' epilogues already measured the tensors
        return gn_finalize(p1, gamma, beta, groups, hw, eps, p2)
    if hw > GN_DIRECT_MAX_PIXELS and (p1 is not None or p2 is not None):
        return gn_finalize(p1 if p1 is not None else gn_stats(x), gamma, beta, groups, hw, eps,
                           None if x2 is None else (p2 if p2 is not None else gn_stats(x2)))
    if hw <= GN_DIRECT_MAX_PIXELS and (c + c2) % groups == 0 and ((c + c2) // groups) % 2 == 0 and c % 2 == 0 and c2 % 2 == 0:
        _chk(x, BF16, "x")
        if x2 is not None:
            _chk(x2, BF16, "x2")
        ab = torch.empty((n, c + c2, 2), device=x.device, dtype=F32)
        lib.call("dc_gn_direct_nhwc_bf16", x.data_ptr(), c, _ptr(x2), c2, _ptr(gamma), _ptr(beta), ab.data_ptr(), n, hw,
                 groups, float(eps), _stream())
        return ab
    return gn_finalize(gn_stats(x), gamma, beta, groups, hw, eps, None if x2 is None else gn_stats(x2))


def gn_apply(x, ab, silu=False, x2=None):
    _chk(x, BF16, "x")
    n, c1 = x.shape[0], x.shape[-1]
    c2 = 0 if x2 is None else x2.shape[-1]
    hw = x.numel() // (n * c1)
    y = torch.empty(x.shape[:-1] + (c1 + c2,), device=x.device, dtype=BF16)
    lib.call("dc_gn_apply_nhwc_bf16", x.data_ptr(), c1, _ptr(x2), c2, ab.data_ptr(), y.data_ptr(), n, hw, int(silu), _stream(),
             meta=_meta("gn_apply_kernel (GroupNorm affine + SiLU)", f"N={n} HW={hw} C={c1}+{c2}", 4.0 * y.numel(), 4.0 * y.numel()))
    return y


def fdn_modulate(x, ab, gamma, beta):
    _chk(x, BF16, "x")
    _chk(gamma, BF16, "gamma")
    _chk(beta, BF16, "beta")
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    y = torch.empty_like(x)
    lib.call("dc_fdn_modulate_nhwc_bf16", x.data_ptr(), ab.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), n,
             gamma.shape[0], hw, c, _stream(),
             meta=_meta("fdn_modulate_kernel (FDN)", f"N={n} HW={hw} C={c}", 4.0 * x.numel(), 2.0 * (2 * x.numel() + 2 * gamma.numel())))
    return y


def layer_norm(x, gamma, beta, eps=1e-5):
    _chk(x, BF16, "x")
    c = x.shape[-1]
    y = torch.empty_like(x)
    lib.call("dc_layernorm_bf16", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), x.numel() // c, c, float(eps), _stream(),
             meta=_meta("layernorm_kernel", f"M={x.numel() // c} C={c}", 8.0 * x.numel(), 4.0 * x.numel()))
    return y


# ------------------------------------------------------------------------------------------ attention
def attention(q, k, v, heads, scale=None, out=None):
    """q [B,Nq,*] k,v [B,Nk,*] bf16 row-strided views (last-dim slices of a fused projection are fine)."""
    b, nq, c = q.shape
    nk = k.shape[1]
    d = c // heads
    for t in (q, k, v):
        assert t.dtype == BF16 and t.is_cuda and t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1)
    if out is None:
        out = torch.empty((b, nq, c), device=q.device, dtype=BF16)
    assert out.shape == (b, nq, c) and out.dtype == BF16 and out.is_contiguous()
    lib.call("dc_attention_bf16", q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), b, heads, nq, nk, d,
             q.stride(1), k.stride(1), v.stride(1), out.stride(1), float(scale if scale is not None else d ** -0.5), _stream(),
             meta=_meta("attn_kernel (flash attention)", f"B={b} H={heads} Nq={nq} Nk={nk} d={d}", 4.0 * b * heads * nq * nk * d,
                        2.0 * (2 * b * nq * c + 2 * b * nk * c)))
    return out


def attention_causal(q, k, v, heads, scale=None):
    """Causal self-attention over a short context (CLIP text, T <= 128): same operand convention as `attention`."""
    b, t, c = q.shape
    d = c // heads
    for x in (q, k, v):
        assert x.dtype == BF16 and x.is_cuda and x.stride(2) == 1 and x.stride(0) == x.shape[1] * x.stride(1) and x.shape[1] == t
    out = torch.empty((b, t, c), device=q.device, dtype=BF16)
    lib.call("dc_attention_causal_small_bf16", q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), b, heads, t, d,
             q.stride(1), k.stride(1), v.stride(1), out.stride(1), float(scale if scale is not None else d ** -0.5), _stream())
    return out


def embed_tokens(ids, tok_emb, pos_emb):
    """ids int64 [B,T] -> bf16 [B,T,C] = tok_emb[ids] + pos_emb[:T]."""
    assert ids.dtype == torch.int64 and ids.is_cuda and ids.is_contiguous() and ids.dim() == 2
    _chk(tok_emb, BF16, "tok_emb")
    _chk(pos_emb, BF16, "pos_emb")
    b, t = ids.shape
    assert t <= pos_emb.shape[0]
    out = torch.empty((b, t, tok_emb.shape[1]), device=ids.device, dtype=BF16)
    lib.call("dc_embed_tokens_bf16", ids.data_ptr(), tok_emb.data_ptr(), pos_emb.data_ptr(), out.data_ptr(), b, t,
             tok_emb.shape[1], tok_emb.shape[0], _stream())
    return out


def softmax_rows(s, scale):
    _chk(s, F32, "s")
    rows, cols = s.shape
    p = torch.empty((rows, cols), device=s.device, dtype=BF16)
    lib.call("dc_softmax_rows_f32_to_bf16", s.data_ptr(), p.data_ptr(), rows, cols, float(scale), _stream())
    return p


# ------------------------------------------------------------------------------------------ layout / misc
def nchw_f32_to_nhwc_bf16(x):
    _chk(x, F32, "x")
    n, c, h, w = x.shape
    y = torch.empty((n, h, w, c), device=x.device, dtype=BF16)
    lib.call("dc_nchw_f32_to_nhwc_bf16", x.data_ptr(), y.data_ptr(), n, c, h, w, _stream())
    return y


def nhwc_bf16_to_nchw_f32(x):
    _chk(x, BF16, "x")
    n, h, w, c = x.shape
    y = torch.empty((n, c, h, w), device=x.device, dtype=F32)
    lib.call("dc_nhwc_bf16_to_nchw_f32", x.data_ptr(), y.data_ptr(), n, c, h, w, _stream())
    return y


def nhwc_f32_to_nchw_f32(x):
    _chk(x, F32, "x")
    n, h, w, c = x.shape
    y = torch.empty((n, c, h, w), device=x.device, dtype=F32)
    lib.call("dc_nhwc_f32_to_nchw_f32", x.data_ptr(), y.data_ptr(), n, c, h, w, _stream())
    return y


def f32_to_bf16(x):
    _chk(x, F32, "x")
    y = torch.empty(x.shape, device=x.device, dtype=BF16)
    lib.call("dc_f32_to_bf16", x.data_ptr(), y.data_ptr(), x.numel(), _stream())
    return y


def silu_f32(x):
    _chk(x, F32, "x")
    y = torch.empty_like(x)
    lib.call("dc_silu_f32", x.data_ptr(), y.data_ptr(), x.numel(), _stream())
    return y


def add_bf16(a, b):
    _chk(a, BF16, "a")
    _chk(b, BF16, "b")
    y = torch.empty_like(a)
    lib.call("dc_add_bf16", a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream())
    return y


def timestep_embedding(t_dev, n, dim, step_dev=None):
    _chk(t_dev, F32, "t")
    out = torch.empty((n, dim), device=t_dev.device, dtype=F32)
    lib.call("dc_timestep_embedding_f32", t_dev.data_ptr(), _ptr(step_dev), out.data_ptr(), n, dim, _stream())
    return out


def freeu_lowfreq(x, s):
    _chk(x, BF16, "x")
    n, h, w, c = x.shape
    y = torch.empty_like(x)
    lib.call("dc_freeu_lowfreq_nhwc_bf16", x.data_ptr(), y.data_ptr(), n, h, w, c, float(s), _stream())
    return y


def freeu_backbone(x, b):
    _chk(x, BF16, "x")
    c = x.shape[-1]
    y = torch.empty_like(x)
    lib.call("dc_freeu_backbone_nhwc_bf16", x.data_ptr(), y.data_ptr(), x.numel() // c, c, float(b), _stream())
    return y


def lincomb(terms):
    """sum_i coef_i * tensor_i over up to four same-shaped contiguous fp32 device tensors: [(coef, tensor), ...]."""
    terms = [(float(c), t) for c, t in terms if t is not None]
    assert 1 <= len(terms) <= 4
    for _, t in terms:
        _chk(t, F32, "term")
    y = torch.empty_like(terms[0][1])
    pad = terms + [(0.0, None)] * (4 - len(terms))
    lib.call("dc_lincomb4_f32", pad[0][1].data_ptr(), _ptr(pad[1][1]), _ptr(pad[2][1]), _ptr(pad[3][1]), pad[0][0], pad[1][0],
             pad[2][0], pad[3][0], y.data_ptr(), y.numel(), _stream())
    return y


def transpose_bf16(x):
    """[B,R,C] -> [B,C,R]"""
    _chk(x, BF16, "x")
    b, r, c = x.shape
    y = torch.empty((b, c, r), device=x.device, dtype=BF16)
    lib.call("dc_transpose_bf16", x.data_ptr(), y.data_ptr(), b, r, c, _stream())
    return y


def vae_sample_latents(moments_nhwc_f32, noise, scale):
    _chk(moments_nhwc_f32, F32, "moments")
    _chk(noise, F32, "noise")
    n, c, h, w = noise.shape
    lat = torch.empty_like(noise)
    lib.call("dc_vae_sample_latents", moments_nhwc_f32.data_ptr(), noise.data_ptr(), lat.data_ptr(), float(scale), n, c, h, w, _stream())
    return lat


# ------------------------------------------------------------------------------------------ splat stage (fp32 NCHW)
def _splat_ws(n, h, w, device):
    """Scratch of the deterministic splat (cell bins; dc_splat_ws_bytes)."""
    return torch.empty((int(lib.load().dc_splat_ws_bytes(int(n), int(h), int(w))),), device=device, dtype=torch.uint8)


def splat_soft(x, flow, metric, mask=None):
    for t, nm in ((x, "in"), (flow, "flow"), (metric, "metric")):
        _chk(t, F32, nm)
    n, c, h, w = x.shape
    assert flow.shape == (n, 2, h, w) and metric.shape == (n, 1, h, w)
    out = torch.empty_like(x)
    ws = _splat_ws(n, h, w, x.device)
    # algorithmic traffic (gather form): flow + metric + bins once per (target, channel) source visit (~1 source per
    # corner list: 4 visits), in read once per visit, out written once
    lib.call("dc_splat_soft_f32", x.data_ptr(), flow.data_ptr(), metric.data_ptr(), _ptr(mask), out.data_ptr(), ws.data_ptr(),
             n, c, h, w, _stream(),
             meta=_meta("splat kernels (softsplat 'soft')", f"N={n} C={c} {h}x{w}", 10.0 * n * (c + 1) * h * w,
                        4.0 * n * h * w * (4 * c + c + 12)))
    return out


def splat_sum(x, flow):
    _chk(x, F32, "in")
    _chk(flow, F32, "flow")
    n, c, h, w = x.shape
    out = torch.empty_like(x)
    ws = _splat_ws(n, h, w, x.device)
    lib.call("dc_splat_sum_f32", x.data_ptr(), flow.data_ptr(), out.data_ptr(), ws.data_ptr(), n, c, h, w, _stream())
    return out


def occlusion_mask(flow_a, flow_b):
    _chk(flow_a, F32, "flow_a")
    _chk(flow_b, F32, "flow_b")
    n, _, h, w = flow_a.shape
    m = torch.empty((n, 1, h, w), device=flow_a.device, dtype=F32)
    ws = _splat_ws(n, h, w, flow_a.device)
    lib.call("dc_occlusion_mask_f32", flow_a.data_ptr(), flow_b.data_ptr(), m.data_ptr(), ws.data_ptr(), n, h, w, _stream())
    return m


def flow_resize_normalize(flow2, th, tw):
    """flow2: [N,2,H,W] fp32, possibly a channel-slice view of the [N,4,H,W] control."""
    assert flow2.dtype == F32 and flow2.is_cuda and flow2.shape[1] == 2
    n, _, h, w = flow2.shape
    assert flow2.stride(3) == 1 and flow2.stride(2) == w and flow2.stride(1) == h * w
    out = torch.empty((n, 2, th, tw), device=flow2.device, dtype=F32)
    lib.call("dc_flow_resize_normalize_f32", flow2.data_ptr(), flow2.stride(0), out.data_ptr(), n, h, w, th, tw, _stream())
    return out


def fuse_warped(wf, wl, cf, cb, of=None, ob=None):
    n, c, h, w = wf.shape
    out = torch.empty_like(wf)
    lib.call("dc_fuse_warped_f32", wf.data_ptr(), wl.data_ptr(), cf.data_ptr(), cb.data_ptr(), _ptr(of), _ptr(ob),
             out.data_ptr(), n, c, h, w, _stream())
    return out


def flow_resize_divide(flow2, th, tw, div_x, div_y):
    assert flow2.dtype == F32 and flow2.is_cuda and flow2.shape[1] == 2
    n, _, h, w = flow2.shape
    assert flow2.stride(3) == 1 and flow2.stride(2) == w and flow2.stride(1) == h * w
    out = torch.empty((n, 2, th, tw), device=flow2.device, dtype=F32)
    lib.call("dc_flow_resize_divide_f32", flow2.data_ptr(), flow2.stride(0), out.data_ptr(), n, h, w, th, tw, float(div_x),
             float(div_y), _stream())
    return out


def add_f32(a, b):
    _chk(a, F32, "a")
    _chk(b, F32, "b")
    y = torch.empty_like(a)
    lib.call("dc_add_f32", a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream())
    return y


# ------------------------------------------------------------------------------------------ scheduler step / io
def cfg_ddim_step(eps, latents, model_in, coef_dev, step_dev, guidance, cfg):
    b, c, h, w = latents.shape
    lib.call("dc_cfg_ddim_step", eps.data_ptr(), latents.data_ptr(), model_in.data_ptr(), coef_dev.data_ptr(),
             step_dev.data_ptr(), float(guidance), int(cfg), b, c, h, w, _stream(),
             meta=_meta("cfg_ddim_kernel (CFG + DDIM step)", f"B={b} {h}x{w}", 10.0 * latents.numel(),
                        4.0 * eps.numel() + 8.0 * latents.numel() + 2.0 * model_in.numel()))


def cfg_unipc_step(eps, latents, m0, m1, last, model_in, coef_dev, step_dev, guidance, cfg):
    """CFG combine + UniPC corrector / predictor on the fp32 state (dc_cfg_unipc_step); every tensor updated in place."""
    b, c, h, w = latents.shape
    for t in (latents, m0, m1, last):
        _chk(t, F32, "state")
        assert t.shape == latents.shape and t.is_contiguous()
    lib.call("dc_cfg_unipc_step", eps.data_ptr(), latents.data_ptr(), m0.data_ptr(), m1.data_ptr(), last.data_ptr(),
             model_in.data_ptr(), coef_dev.data_ptr(), step_dev.data_ptr(), float(guidance), int(cfg), b, c, h, w, _stream(),
             meta=_meta("cfg_unipc_kernel (CFG + UniPC step)", f"B={b} {h}x{w}", 20.0 * latents.numel(),
                        4.0 * eps.numel() + 32.0 * latents.numel() + 2.0 * model_in.numel()))


def latents_to_model_input(latents, mul=1.0, rep=1, out=None):
    _chk(latents, F32, "latents")
    b, c, h, w = latents.shape
    if out is None:
        out = torch.empty((rep * b, h, w, c), device=latents.device, dtype=BF16)
    lib.call("dc_latents_to_model_input", latents.data_ptr(), out.data_ptr(), float(mul), rep, b, c, h, w, _stream())
    return out


def postprocess_image(x_nhwc_f32, want_u8=False):
    """x [n,h,w,c] fp32; may be a channel-slice view of a wider contiguous tensor (conv_out run with a padded channel)."""
    x = x_nhwc_f32
    n, h, w, c = x.shape
    if not (x.is_cuda and x.dtype == F32 and x.stride(3) == 1 and x.stride(1) == w * x.stride(2) and x.stride(0) == h * x.stride(1)):
        raise ValueError("x: expected a device fp32 NHWC tensor or a channel-slice view of one")
    o32 = torch.empty((n, c, h, w), device=x.device, dtype=F32)
    o8 = torch.empty((n, h, w, c), device=x.device, dtype=torch.uint8) if want_u8 else None
    lib.call("dc_postprocess_image", x.data_ptr(), o32.data_ptr(), _ptr(o8), n, c, h, w, int(x.stride(2)), _stream())
    return o32, o8


# ------------------------------------------------------------------------------------------ input side
def flow_hw2_resize_scale(flow_hw2, th, tw, out=None):
    """resize_flow_to (controlnet/utils.py:21-28) on the device: [H,W,2] fp32 (.flo payload layout) -> [2,th,tw] fp32."""
    _chk(flow_hw2, F32, "flow_hw2")
    h, w, two = flow_hw2.shape
    assert two == 2
    if out is None:
        out = torch.empty((2, th, tw), device=flow_hw2.device, dtype=F32)
    assert out.shape == (2, th, tw) and out.dtype == F32 and out.is_contiguous()
    lib.call("dc_flow_hw2_resize_scale_f32", flow_hw2.data_ptr(), h, w, out.data_ptr(), th, tw, _stream())
    return out


def pack_sixch(img0_u8, img1_u8):
    """load_pair_to_sixch's tensor half (utils.py:36-39): two [H,W,3] uint8 images -> [1,6,H,W] fp32 in [0,1]."""
    for t in (img0_u8, img1_u8):
        assert t.dtype == torch.uint8 and t.is_cuda and t.is_contiguous() and t.dim() == 3 and t.shape[2] == 3
    assert img0_u8.shape == img1_u8.shape
    h, w, _ = img0_u8.shape
    out = torch.empty((1, 6, h, w), device=img0_u8.device, dtype=F32)
    lib.call("dc_pack_sixch_u8_f32", img0_u8.data_ptr(), img1_u8.data_ptr(), out.data_ptr(), h, w, _stream())
    return out


def blend_tiles_ramp(tiles_nchw, coords, full_hw, feather, scale=255.0):
    """tiles [T,C,th,tw] fp32 on the device (in [0,1] with scale=255, or already 0..255 pixel values with scale=1: tiles that
    were quantised to uint8 images first, as the reference's notebook does), coords [(y1,y2,x1,x2)] full-size windows ->
    uint8 [H,W,C] (device).  Same weights and fp32 op order as tiling.merge_ramp."""
    import numpy as np
    _chk(tiles_nchw, F32, "tiles")
    t, c, th, tw = tiles_nchw.shape
    h, w = full_hw
    assert len(coords) == t and all(y2 - y1 == th and x2 - x1 == tw for (y1, y2, x1, x2) in coords)
    f = int(feather)
    dev = tiles_nchw.device
    ramp = (0.5 - 0.5 * np.cos(np.pi * (np.arange(f, dtype=np.float32) + 0.5) / f)).astype(np.float32) if f > 0 else np.zeros(1, np.float32)
    ramp_d = torch.from_numpy(ramp).to(dev)
    coords_d = torch.tensor(coords, dtype=torch.int32).reshape(-1, 4).to(dev)
    out = torch.empty((h, w, c), device=dev, dtype=torch.uint8)
    lib.call("dc_blend_tiles_ramp_u8", tiles_nchw.data_ptr(), coords_d.data_ptr(), t, c, th, tw, ramp_d.data_ptr(), f,
             out.data_ptr(), h, w, float(scale), _stream())
    return out
