// Shared device/host helpers for the gfx950 (CDNA4, wave64) kernels of the DiffCodec decode path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#define DC_OK 0
#define DC_ERR_INVALID (-1)
#define DC_ERR_LAUNCH (-2)

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define DC_WAVE 64

// Developer A/B switches of the launchers (tile shape, ring depth, kernel selection).  The product library is built WITHOUT
// DC_DEV_KNOBS, so every DC_KNOB folds to its default and no dispatch decision depends on the environment; scratch builds of
// tools/ (-DDC_DEV_KNOBS) read the variable once.
#ifdef DC_DEV_KNOBS
#include <cstdlib>
#define DC_KNOB(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define DC_KNOB(name, dflt) (dflt)
#endif

__device__ __forceinline__ float dc_bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t dc_f2bf(float v) { return (bf16_t)v; }   // v_cvt_pk_bf16_f32: RNE, NaN-preserving

// SiLU / sigmoid forms use v_rcp_f32 (1 ulp) instead of an IEEE division: `x / (1 + e)` compiles to the ~10-instruction
// v_div_scale / v_div_fmas / v_div_fixup sequence per element, which made the GroupNorm-apply pass and every GN-on-load halo
// VALU-bound (gn_apply ran at 3.1 TB/s); the result is rounded to bf16 right after, 2^-16 below its last bit.
__device__ __forceinline__ float dc_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// dc_conv_desc.act: 1 = SiLU, 2 = quick-GELU x * sigmoid(1.702 x) (CLIP text MLP)
__device__ __forceinline__ float dc_act(float x, int act) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(act == 2 ? -1.702f * x : -x)); }
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the bf16 output rounding): 1 rcp + 1 exp + 7 fma
// instead of libm erff's branchy polynomial (~3x the VALU work in the GEGLU epilogue).
__device__ __forceinline__ float dc_erf_fast(float x)
{
#pragma clang fp contract(off)
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float p = 1.061405429f;
    p = __builtin_fmaf(p, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = __builtin_fmaf(-(p * t), e, 1.0f);
    return copysignf(r, x);
}
#ifndef DC_GELU_VARIANT
#define DC_GELU_VARIANT 0       // developer experiments only (tools/bench_gemm.py A/B): 1 = identity (upper bound of what a cheaper
#endif                          // GELU could buy), never built into the product library
__device__ __forceinline__ float dc_gelu_erf(float x)
{
#pragma clang fp contract(off)
#if DC_GELU_VARIANT == 1
    return x;
#else
    return 0.5f * x * (1.0f + dc_erf_fast(x * 0.70710678118654752440f));
#endif
}

// Epilogue arithmetic shared by the GEMM / conv kernels, with the fused multiply-adds WRITTEN OUT.  Left to -ffp-contract, hipcc
// fuses `v * scale + residual` (and the LayerNorm fold's `v - mean * colsum`) in some instantiations / unrolled copies of the same
// source expression and not in others (first build of gemm_rowpanel.hip: 109 of 21 M outputs one bf16 ulp away from gemm_dma.hip
// with out_scale = 0.75, all in the second row tile) — so the same row could round differently depending on which kernel, or
// which copy of a loop body, produced it.  One definition, one rounding sequence, every kernel: contraction is switched off inside
// these helpers (hipcc's default -ffp-contract=fast-honor-pragmas fuses only operations that BOTH carry the contract flag, so a
// multiply in here cannot be fused with an add at the call site either).
__device__ __forceinline__ f32x4 dc_scale_res(f32x4 v, float scale, bf16x4 rr)
{
#pragma clang fp contract(off)
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = __builtin_fmaf(v[r], scale, (float)rr[r]);
    return o;
}
// Linear(LN(x)) = rstd * (x W'^T - mean * colsum(W')) (+ b'): the per-row half of the folded LayerNorm
__device__ __forceinline__ f32x4 dc_ln_fold(f32x4 v, float mean, float rstd, f32x4 cs)
{
#pragma clang fp contract(off)
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = __builtin_fmaf(-mean, cs[r], v[r]) * rstd;
    return o;
}

// GroupNorm affine of one packed bf16 pair: (x * a + b) per element as ONE fused multiply-add each, rounded to bf16 — the arithmetic of
// dc_gn_apply_nhwc_bf16, shared with the kernels that apply the affine on load so that both give the same bits.
// g = (a_lo, b_lo, a_hi, b_hi): the (scale, shift) pairs of two consecutive channels as dc_gn_finalize lays them out.
__device__ __forceinline__ uint32_t dc_gn_affine_pair(uint32_t raw, f32x4 g)
{
#pragma clang fp contract(off)
    const float lo = __builtin_fmaf(__uint_as_float(raw << 16), g[0], g[1]);
    const float hi = __builtin_fmaf(__uint_as_float(raw & 0xffff0000u), g[2], g[3]);
    const bf16x2 pk = {(bf16_t)lo, (bf16_t)hi};
    return *(const uint32_t*)&pk;
}

// (sum, sum of squares) partials of one row -> (mean, rstd) of a LayerNorm over 1 / inv_c channels: the ONE definition behind
// dc_ln_finalize and the consumers that finalize in their prologue (dc_conv_desc.ln_parts), so both give the same bits.
__device__ __forceinline__ void dc_ln_mean_rstd_of_sums(float a1, float a2, float inv_c, float eps, float& mean, float& rstd)
{
#pragma clang fp contract(off)
    mean = a1 * inv_c;
    rstd = rsqrtf(fmaxf(a2 * inv_c - mean * mean, 0.f) + eps);
}
__device__ __forceinline__ void dc_ln_mean_rstd(const float* __restrict__ row_partials, int parts, float inv_c, float eps, float& mean,
                                                float& rstd)
{
#pragma clang fp contract(off)
    float a1 = 0.f, a2 = 0.f;
    for (int i = 0; i < parts; ++i) {                       // independent 8-byte loads, summed in index order
        const float2 v = ((const float2*)row_partials)[i];
        a1 += v.x;
        a2 += v.y;
    }
    dc_ln_mean_rstd_of_sums(a1, a2, inv_c, eps, mean, rstd);
}
// The same in two phases for the row-panel GEMM's prologue (dc_conv_desc.ln_parts): the partial pairs of all of a lane's rows are FETCHED
// first (independent loads in flight together), then SUMMED in index order as above.  (Only a kernel whose workgroup owns whole rows
// folds the finalize: in the tile GEMMs every N tile would redo it — measured on the 256-row kernel: N = 10240, K = 1280 GEGLU 235 ->
// 373 us — so those launches keep the standalone dc_ln_finalize pass.)
constexpr int DC_LN_PARTS_MAX = 16;                         // = dc_row_stats_parts_rule(1280)
struct dc_ln_row_partials {
    float2 v[DC_LN_PARTS_MAX];
};
__device__ __forceinline__ void dc_ln_fetch_partials(dc_ln_row_partials& r, const float* __restrict__ row_partials, int parts)
{
#pragma unroll
    for (int i = 0; i < DC_LN_PARTS_MAX; ++i)
        if (i < parts) r.v[i] = ((const float2*)row_partials)[i];
}
__device__ __forceinline__ void dc_ln_mean_rstd(const dc_ln_row_partials& r, int parts, float inv_c, float eps, float& mean, float& rstd)
{
#pragma clang fp contract(off)
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < DC_LN_PARTS_MAX; ++i)
        if (i < parts) {
            a1 += r.v[i].x;
            a2 += r.v[i].y;
        }
    dc_ln_mean_rstd_of_sums(a1, a2, inv_c, eps, mean, rstd);
}

__device__ __forceinline__ float dc_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float dc_wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// One K-step hand-over of an LDS ring filled by LDS-DMA: wait until all but the N youngest vector-memory operations of
// this wave have landed AND until every LDS read this wave has issued has returned, then the workgroup barrier.
// The lgkmcnt(0) is what makes the ring safe to refill: hipcc software-pipelines the last fragment reads of a K-step past
// a raw s_barrier (the MFMAs that consume them are register-only, so nothing orders them against it), and a wave that
// arrives at the barrier with reads still in flight lets another wave's DMA refill the stage under them — rare wrong
// tiles that differ from run to run (found with tools/find_nondeterminism.py on the unrolled TM=2 tile conv).  The
// sched_barrier keeps the step's MFMAs in front of the wait (so the wait is normally already satisfied), the compiler
// barrier after s_barrier keeps the next step's LDS reads behind it.
template <int N>
__device__ __forceinline__ void dc_ring_sync()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// GroupNorm statistics from a conv / GEMM epilogue: `s` / `q` = this lane's (sum, sum of squares) over its pixel rows for the 4
// channels nb..nb+3 of one 16-channel n-tile.  The 16 lanes that differ in lane bits 0-3 hold other pixels of the same
// channels: fixed-order DPP row reduction, then the lane with bits 0-3 clear stores [4 channels][2] floats.
// Sum over the 16 lanes of a DPP row (lanes that differ in lane bits 0-3) on the VALU: quad swaps, half-row mirror, row
// mirror — every lane ends with the row total (each in its own, fixed association order), no LDS traffic (a `__shfl_xor`
// compiles to ds_bpermute_b32).
__device__ __forceinline__ float dc_row16_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));   // row_mirror
    return v;
}

__device__ __forceinline__ void dc_gn_partial_store(f32x4 s, f32x4 q, float* __restrict__ dst, bool writer)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s[r] = dc_row16_sum(s[r]);
        q[r] = dc_row16_sum(q[r]);
    }
    if (writer) {
        *(f32x4*)dst = f32x4{s[0], q[0], s[1], q[1]};
        *(f32x4*)(dst + 4) = f32x4{s[2], q[2], s[3], q[3]};
    }
}

// Staged bf16 output tiles ([row][columns] with a 16-byte-multiple pitch, written as 8-byte pieces by the MFMA layout: 16 lanes =
// 16 consecutive rows at one column offset): whatever the pitch, rows r and r + 8 fall on the same banks (the pitch in dwords is a
// multiple of 4, so only 8 of the 16 rows land on distinct bank pairs) — a 2-way conflict on every ds_write_b64.  Rows 8-15 of each
// 16-row group therefore swap the two 8-byte halves of every 16-byte piece (XOR 8 on the byte offset), which moves them onto the
// free bank pairs; the row pass swaps the halves back in registers.
__device__ __forceinline__ int dc_stage_swz(int row) { return ((row >> 3) & 1) << 3; }
__device__ __forceinline__ u32x4 dc_stage_unswz(u32x4 v, int row)
{
    return ((row >> 3) & 1) ? u32x4{v[2], v[3], v[0], v[1]} : v;
}

static inline int dc_launch_status()
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DC_OK : DC_ERR_LAUNCH;
}
// Raise a kernel's dynamic-LDS limit once per (kernel, device): one bit per device ordinal in a per-kernel mask.  The only
// process-wide state of the library; lock-free (setting the attribute twice is harmless, so two host threads racing on the
// first launch merely repeat the call) and correct when one process drives several devices.
static inline void dc_set_max_dyn_lds(const void* kern, int bytes, std::atomic<unsigned long long>& done)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
}
static inline int dc_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
// Partials per output row of a `stats_out` launch (the value dc_gemm_row_stats_parts reports to the host): one per wave column
// slice of the tile grid of the 128-row kernel; kernels that produce fewer (gemm_rowpanel.hip: one) zero-fill the rest.
__host__ __device__ static inline int dc_row_stats_parts_rule(int Cout)
{
    if (Cout <= 0) return 0;
    const int bn = (Cout % 160 == 0) ? 160 : 128;       // the N tile dc_gemm_dma_launch picks for a plain (non-GEGLU) epilogue
    return 2 * ((Cout + bn - 1) / bn);                  // two wave column slices per N tile
}
