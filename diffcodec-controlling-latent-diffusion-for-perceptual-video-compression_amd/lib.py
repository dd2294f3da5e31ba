"""ctypes binding of the C-ABI declared in include/diffcodec_hip.h.

The product path has no CPU fallback: if the shared object is missing or a launcher returns non-zero, this
module raises.  (Loading the library and resolving symbols needs no GPU; launching does.)"""
import ctypes
import os
from ctypes import POINTER, c_float, c_int, c_longlong, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdiffcodec_hip.so")

vp, i32, i64, f32 = c_void_p, c_int, c_longlong, c_float


class ConvDesc(ctypes.Structure):
    """mirror of `dc_conv_desc` (include/diffcodec_hip.h)"""
    _fields_ = [
        ("x1", vp), ("x2", vp), ("w", vp), ("bias", vp), ("gn_ab", vp), ("row_add", vp), ("residual", vp),
        ("out", vp), ("splitk_ws", vp),
        ("N", i32), ("H", i32), ("W", i32), ("C1", i32), ("C2", i32), ("Cout", i32),
        ("ksize", i32), ("stride", i32), ("pad", i32), ("upsample", i32), ("Ho", i32), ("Wo", i32),
        ("gn_silu", i32), ("epilogue", i32), ("out_f32", i32), ("out_scale", f32), ("splitk", i32), ("gn_batch", i32),
        ("act", i32), ("row_add_stride", i64),
        ("ln_stats", vp), ("ln_colsum", vp), ("stats_out", vp), ("gn_part_out", vp),
        ("ln_parts", i32), ("ln_eps", f32), ("ln_scratch", vp),
    ]


# name -> argtypes (every symbol include/diffcodec_hip.h declares; tests/test_abi.py cross-checks the header)
SIGNATURES = {
    "dc_splat_soft_f32": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "dc_splat_sum_f32": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "dc_splat_ws_bytes": [i32, i32, i32],
    "dc_occlusion_mask_f32": [vp, vp, vp, vp, i32, i32, i32, vp],
    "dc_flow_resize_normalize_f32": [vp, i64, vp, i32, i32, i32, i32, i32, vp],
    "dc_flow_resize_divide_f32": [vp, i64, vp, i32, i32, i32, i32, i32, f32, f32, vp],
    "dc_fuse_warped_f32": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "dc_conv3x3_nchw_f32": [vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "dc_nchw_f32_to_nhwc_bf16": [vp, vp, i32, i32, i32, i32, vp],
    "dc_nhwc_bf16_to_nchw_f32": [vp, vp, i32, i32, i32, i32, vp],
    "dc_nhwc_f32_to_nchw_f32": [vp, vp, i32, i32, i32, i32, vp],
    "dc_f32_to_bf16": [vp, vp, i64, vp],
    "dc_conv_igemm_bf16": [POINTER(ConvDesc), vp],
    "dc_conv_igemm_ws_bytes": [POINTER(ConvDesc)],
    "dc_gemm_row_stats_parts": [i32],
    "dc_conv_gn_part_chunks": [POINTER(ConvDesc)],
    "dc_row_stats_bf16": [vp, vp, i64, i32, vp],
    "dc_ln_finalize": [vp, vp, i64, i32, i32, f32, vp],
    "dc_conv_small_cin_bf16": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "dc_conv_small_cout_bf16": [vp, vp, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "dc_gn_stats_nhwc_bf16": [vp, vp, i32, i64, i32, vp],
    "dc_gn_stats_chunks": [i64, i32],
    "dc_gn_finalize": [vp, i32, i32, vp, i32, i32, vp, vp, vp, i32, i32, i64, f32, vp],
    "dc_gn_direct_nhwc_bf16": [vp, i32, vp, i32, vp, vp, vp, i32, i64, i32, f32, vp],
    "dc_gn_apply_nhwc_bf16": [vp, i32, vp, i32, vp, vp, i32, i64, i32, vp],
    "dc_fdn_modulate_nhwc_bf16": [vp, vp, vp, vp, vp, i32, i32, i64, i32, vp],
    "dc_layernorm_bf16": [vp, vp, vp, vp, i64, i32, f32, vp],
    "dc_attention_bf16": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, i64, i64, i64, f32, vp],
    "dc_attention_causal_small_bf16": [vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i64, i64, f32, vp],
    "dc_embed_tokens_bf16": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "dc_softmax_rows_f32_to_bf16": [vp, vp, i64, i32, f32, vp],
    "dc_timestep_embedding_f32": [vp, vp, vp, i32, i32, vp],
    "dc_freeu_lowfreq_nhwc_bf16": [vp, vp, i32, i32, i32, i32, f32, vp],
    "dc_freeu_backbone_nhwc_bf16": [vp, vp, i64, i32, f32, vp],
    "dc_lincomb4_f32": [vp, vp, vp, vp, f32, f32, f32, f32, vp, i64, vp],
    "dc_transpose_bf16": [vp, vp, i32, i32, i32, vp],
    "dc_vae_sample_latents": [vp, vp, vp, f32, i32, i32, i32, i32, vp],
    "dc_silu_f32": [vp, vp, i64, vp],
    "dc_add_bf16": [vp, vp, vp, i64, vp],
    "dc_add_f32": [vp, vp, vp, i64, vp],
    "dc_cfg_ddim_step": [vp, vp, vp, vp, vp, f32, i32, i32, i32, i32, i32, vp],
    "dc_cfg_unipc_step": [vp, vp, vp, vp, vp, vp, vp, vp, f32, i32, i32, i32, i32, i32, vp],
    "dc_latents_to_model_input": [vp, vp, f32, i32, i32, i32, i32, i32, vp],
    "dc_postprocess_image": [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "dc_flow_hw2_resize_scale_f32": [vp, i32, i32, vp, i32, i32, vp],
    "dc_pack_sixch_u8_f32": [vp, vp, vp, i32, i32, vp],
    "dc_blend_tiles_ramp_u8": [vp, vp, i32, i32, i32, i32, vp, i32, vp, i32, i32, f32, vp],
}

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """Load libdiffcodec_hip.so (built by `python -m diffcodec_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH                                          # tools/ assign lib.LIB_PATH to A/B another build of the same ABI
    if not os.path.exists(path):
        raise HipLibraryMissing(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'`. "
                                "There is no CPU fallback on the product path.")
    lib = ctypes.CDLL(path)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes = args
        fn.restype = c_longlong if name in ("dc_conv_igemm_ws_bytes", "dc_splat_ws_bytes") else c_int
    lib.dc_gn_stats_chunks.restype = c_int
    _lib = lib
    return lib


class HipLaunchError(RuntimeError):
    pass


class LaunchTimer:
    """In-situ per-launch timing for bench.py's roofline leg (never active in a timed region): while `lib.TIMER` is set, every
    C-ABI call is bracketed by two HIP events recorded on the launch stream (torch's current stream — the stream the launcher
    is given), so the elapsed time is the kernel's own duration inside the real step: real operands, real cache state, host
    launch gaps excluded.  `meta` = (family label, shape label, algorithmic FLOPs, algorithmic bytes) from the ops wrapper."""

    def __init__(self):
        self.records = []          # (name, meta, start event, end event)

    def summary(self):
        """{family: dict(calls, ms, flops, bytes, shapes={shape: [calls, ms, flops, bytes]})} — call after a device sync."""
        out = {}
        for name, meta, e0, e1 in self.records:
            fam, shape, fl, by = meta if meta is not None else (name, "", 0.0, 0.0)
            ms = e0.elapsed_time(e1)
            f = out.setdefault(fam, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0, shapes={}))
            f["calls"] += 1
            f["ms"] += ms
            f["flops"] += fl
            f["bytes"] += by
            sh = f["shapes"].setdefault(shape, [0, 0.0, 0.0, 0.0])
            sh[0] += 1
            sh[1] += ms
            sh[2] += fl
            sh[3] += by
        return out


TIMER = None


def call(name, *args, meta=None):
    t = TIMER
    if t is None:
        rc = getattr(load(), name)(*args)
    else:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(load(), name)(*args)
        e1.record()
        t.records.append((name, meta, e0, e1))
    if rc != 0:
        raise HipLaunchError(f"{name} returned {rc} ({'invalid argument' if rc == -1 else 'launch failure'})")
