// HBM-bound normalisation kernels for gfx950 (NHWC bf16 activations, fp32 statistics).
// GroupNorm is split into per-(sample,channel) sums -> per-(sample,channel) scale/shift, so that the affine
// (+SiLU) can be applied inside the consuming conv's load stage (igemm.hip) and so that the channel concat of
// UNet skip connections never has to be materialised.  Replaces nn.GroupNorm / nn.LayerNorm inside the
// diffusers blocks called from flownet.py:87-118, pipeline.py:358-367,391 and FDN (control_utils.py:24-34).
#include "dc_common.h"
#include "../../include/diffcodec_hip.h"

namespace {

// Per-(chunk, n, c) partial sum and sum of squares — no atomics, no memset: grid = (pixel chunks, N); each thread owns
// one 8-channel vector column and strides over the chunk's pixels; the pixel-parallel partials are combined through LDS
// and each workgroup writes its own [C][2] slab.  The finalize kernel sums the slabs.
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ part,
                                                       long long HW, int C, int pix_per_block, int N)
{
    extern __shared__ float red[];                 // [ppb][C][2]
    const int nv = C >> 3;
    const int n = blockIdx.y;
    const long long p_begin = (long long)blockIdx.x * pix_per_block;
    const long long p_end = min(HW, p_begin + pix_per_block);
    const int tpp = min(nv, 256);                  // threads per pixel
    const int ppb = 256 / tpp;                     // pixels in flight
    const int vl = threadIdx.x % tpp, pl = threadIdx.x / tpp;
    if (pl < ppb) {
        for (int v = vl; v < nv; v += tpp) {
            float s[8], ss[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
            for (long long p0 = p_begin + pl; p0 < p_end; p0 += 4LL * ppb) {      // four 16-byte loads in flight per thread
                u32x4 raw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long p = p0 + (long long)u * ppb;
                    raw[u] = p < p_end ? *(const u32x4*)(x + ((long long)n * HW + p) * C + v * 8) : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float lo = __uint_as_float(raw[u][j] << 16);
                        const float hi = __uint_as_float(raw[u][j] & 0xffff0000u);
                        s[2 * j] += lo;
                        ss[2 * j] += lo * lo;
                        s[2 * j + 1] += hi;
                        ss[2 * j + 1] += hi * hi;
                    }
            }
            float* r = red + ((long long)pl * C + v * 8) * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                r[2 * j] = s[j];
                r[2 * j + 1] = ss[j];
            }
        }
    }
    __syncthreads();
    float* out = part + ((long long)blockIdx.x * N + n) * C * 2;
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        float a = 0.f;
        for (int k = 0; k < ppb; ++k) a += red[(long long)k * C * 2 + i];
        out[i] = a;
    }
}

// One 256-thread workgroup per (sample, group): threads stride over the group's (chunk, channel) partials — at most a few
// independent 8-byte loads each, all in flight together (a single wave walking 640 partials ten deep took 9 us per launch,
// 1,370 launches per 16-frame step) —, fixed-order wave + LDS reduction, then the first cpg threads emit the per-channel
// scale/shift.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ s1, int C1, int chunks1,
                                                          const float* __restrict__ s2, int C2, int chunks2,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ ab, int N, int groups, float inv_count, float eps)
{
    __shared__ float red[8];
    const int C = C1 + C2;
    const int n = blockIdx.x / groups, g = blockIdx.x - n * groups;
    const int cpg = C / groups;
    const int g0 = g * cpg;
    const int tid = threadIdx.x;
    float s = 0.f, ss = 0.f;
    // channels of the group that live in the first / second source
    const int a0 = min(g0, C1), a1 = min(g0 + cpg, C1);            // [a0,a1) in source 1
    const int b0 = max(g0, C1) - C1, b1 = max(g0 + cpg, C1) - C1;  // [b0,b1) in source 2
    const int w1 = a1 - a0, w2 = b1 - b0;
#pragma unroll 4
    for (int i = tid; i < w1 * chunks1; i += 256) {
        const int ch = i / w1, k = a0 + (i - ch * w1);
        const float2 v = *(const float2*)(s1 + (((long long)ch * N + n) * C1 + k) * 2);
        s += v.x;
        ss += v.y;
    }
#pragma unroll 4
    for (int i = tid; i < w2 * chunks2; i += 256) {
        const int ch = i / w2, k = b0 + (i - ch * w2);
        const float2 v = *(const float2*)(s2 + (((long long)ch * N + n) * C2 + k) * 2);
        s += v.x;
        ss += v.y;
    }
    s = dc_wave_sum(s);
    ss = dc_wave_sum(ss);
    if ((tid & 63) == 0) {
        red[tid >> 6] = s;
        red[4 + (tid >> 6)] = ss;
    }
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    ss = (red[4] + red[5]) + (red[6] + red[7]);
    const float mean = s * inv_count;
    const float var = fmaxf(ss * inv_count - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    for (int k = tid; k < cpg; k += 256) {
        const int c = g0 + k;
        const float ga = gamma ? gamma[c] : 1.f;
        const float be = beta ? beta[c] : 0.f;
        float2 o;
        o.x = rstd * ga;
        o.y = be - mean * rstd * ga;
        *(float2*)(ab + ((long long)n * C + c) * 2) = o;
    }
}

// Small maps (<= 16x16): the whole statistic in ONE launch, one workgroup per (sample, group), reading that group's
// channels of cat[x1, x2] directly (bf16 pairs; a pixel's group slice is contiguous).  Measured (tools/bench_gn.py):
// 4-8 us against 8-15 us for the slab + finalize pair at 8x8 / 16x16; slower from 32x32 up (strided reads, few
// workgroups), where the pair stays.  One-launch variants that finalise in the last workgroup to arrive were tried twice
// and dropped: with fp32 atomics on the sums (24-105 us at 32 samples) and with plain slabs + one ticket atomic per
// workgroup (48-178 us) — the agent-scope `__threadfence()` the hand-off needs (L2 write-back + invalidate across the
// eight XCDs) costs far more than the dependent launch it saves.
__global__ __launch_bounds__(256) void gn_direct_kernel(const bf16_t* __restrict__ x1, int C1, const bf16_t* __restrict__ x2,
                                                        int C2, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ ab, long long HW, int groups, int tp, float inv_count,
                                                        float eps)
{
    __shared__ float red[8];
    const int C = C1 + C2;
    const int n = blockIdx.x / groups, g = blockIdx.x - n * groups;
    const int cpg = C / groups, g0 = g * cpg;
    const int a0 = min(g0, C1), a1 = min(g0 + cpg, C1);            // [a0,a1) of source 1
    const int b0 = max(g0, C1) - C1, b1 = max(g0 + cpg, C1) - C1;  // [b0,b1) of source 2
    const int j = threadIdx.x % tp, pl = threadIdx.x / tp, pstep = 256 / tp;
    float s = 0.f, ss = 0.f;
    auto scan = [&](const bf16_t* __restrict__ x, int Cs, int c0, int c1) {
        const int npairs = (c1 - c0) >> 1;
        for (long long p = pl; p < HW; p += pstep) {
            const bf16_t* row = x + ((long long)n * HW + p) * Cs + c0;
            for (int q = j; q < npairs; q += tp) {
                const uint32_t raw = *(const uint32_t*)(row + 2 * q);
                const float lo = __uint_as_float(raw << 16), hi = __uint_as_float(raw & 0xffff0000u);
                s += lo + hi;
                ss += lo * lo + hi * hi;
            }
        }
    };
    if (a1 > a0) scan(x1, C1, a0, a1);
    if (b1 > b0) scan(x2, C2, b0, b1);
    s = dc_wave_sum(s);
    ss = dc_wave_sum(ss);
    if ((threadIdx.x & 63) == 0) {
        red[(threadIdx.x >> 6) * 2] = s;
        red[(threadIdx.x >> 6) * 2 + 1] = ss;
    }
    __syncthreads();
    s = red[0] + red[2] + red[4] + red[6];
    ss = red[1] + red[3] + red[5] + red[7];
    const float mean = s * inv_count;
    const float var = fmaxf(ss * inv_count - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    for (int k = threadIdx.x; k < cpg; k += 256) {
        const int c = g0 + k;
        const float ga = gamma ? gamma[c] : 1.f;
        const float be = beta ? beta[c] : 0.f;
        float2 o;
        o.x = rstd * ga;
        o.y = be - mean * rstd * ga;
        *(float2*)(ab + ((long long)n * C + c) * 2) = o;
    }
}

// y = (x*a+b) [SiLU] of cat[x1,x2].  grid = (pixel chunks, N); a thread owns one 8-channel vector column of one sample (its
// 16 scale/shift floats stay in registers) and walks the chunk's pixels four at a time, so four 16-byte loads are in flight
// per thread and the only per-pixel work is the arithmetic: no index divisions, no re-reads of the affine.
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x1, int C1,
                                                       const bf16_t* __restrict__ x2, int C2,
                                                       const float* __restrict__ ab, bf16_t* __restrict__ y,
                                                       long long HW, int pix_per_block, int silu)
{
    const int C = C1 + C2;
    const int nv = C >> 3;
    const int n = blockIdx.y;
    const long long p_begin = (long long)blockIdx.x * pix_per_block;
    const long long p_end = min(HW, p_begin + pix_per_block);
    const int tpp = min(nv, 256);                  // threads per pixel
    const int ppb = 256 / tpp;                     // pixels in flight per trip
    const int vl = threadIdx.x % tpp, pl = threadIdx.x / tpp;
    if (pl >= ppb) return;
    for (int v = vl; v < nv; v += tpp) {
        const int c = v * 8;
        f32x4 g[4];
        const float* abp = ab + ((long long)n * C + c) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = *(const f32x4*)(abp + 4 * j);
        const bool first = c < C1;
        const bf16_t* src = first ? x1 + (long long)n * HW * C1 + c : x2 + (long long)n * HW * C2 + (c - C1);
        const int cs = first ? C1 : C2;
        bf16_t* dst = y + (long long)n * HW * C + c;
        for (long long p0 = p_begin + pl; p0 < p_end; p0 += (long long)UNR * ppb) {
            u32x4 raw[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long long p = p0 + (long long)u * ppb;
                if (p < p_end) raw[u] = *(const u32x4*)(src + p * cs);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long long p = p0 + (long long)u * ppb;
                if (p >= p_end) break;
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!silu) {
                        o[j] = dc_gn_affine_pair(raw[u][j], g[j]);         // (shared with the GEMM that applies the affine on load)
                        continue;
                    }
                    float lo = __uint_as_float(raw[u][j] << 16) * g[j][0] + g[j][1];
                    float hi = __uint_as_float(raw[u][j] & 0xffff0000u) * g[j][2] + g[j][3];
                    lo = dc_silu(lo);
                    hi = dc_silu(hi);
                    bf16x2 pk = {(bf16_t)lo, (bf16_t)hi};
                    o[j] = *(uint32_t*)&pk;
                }
                const u32x4 ov = {o[0], o[1], o[2], o[3]};
                if (NT) __builtin_nontemporal_store(ov, (u32x4*)(dst + p * C));
                else *(u32x4*)(dst + p * C) = ov;
            }
        }
    }
}

__global__ __launch_bounds__(256) void fdn_modulate_kernel(const bf16_t* __restrict__ x, const float* __restrict__ ab,
                                                           const bf16_t* __restrict__ gamma,
                                                           const bf16_t* __restrict__ beta, bf16_t* __restrict__ y,
                                                           int Bp, long long HW, int C, long long total_vec)
{
    const int nv = C >> 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (long long)gridDim.x * 256) {
        const long long pix = i / nv;
        const int c = (int)(i - pix * nv) * 8;
        const int n = (int)(pix / HW);
        const long long ppix = (long long)(n % Bp) * HW + (pix - (long long)n * HW);
        const bf16x8 xv = *(const bf16x8*)(x + pix * C + c);
        const bf16x8 gv = *(const bf16x8*)(gamma + ppix * C + c);
        const bf16x8 bv = *(const bf16x8*)(beta + ppix * C + c);
        const float* abp = ab + ((long long)n * C + c) * 2;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float nrm = (float)xv[j] * abp[2 * j] + abp[2 * j + 1];
            o[j] = (bf16_t)(nrm * (1.0f + (float)gv[j]) + (float)bv[j]);
        }
        *(bf16x8*)(y + pix * C + c) = o;
    }
}

// LayerNorm: one wave per row, values kept in registers (C <= 64*8*MAXV).
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                        long long M, int C, float eps)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = C >> 3;
    float v[MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int vi = lane + 64 * k;
        if (vi < nv) {
            const bf16x8 r = *(const bf16x8*)(x + row * C + vi * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[k][j] = (float)r[j];
                s += v[k][j];
            }
        }
    }
    const float mean = dc_wave_sum(s) / C;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int vi = lane + 64 * k;
        if (vi < nv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[k][j] - mean;
                ss += d * d;
            }
        }
    }
    const float rstd = rsqrtf(dc_wave_sum(ss) / C + eps);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int vi = lane + 64 * k;
        if (vi < nv) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = vi * 8 + j;
                o[j] = (bf16_t)((v[k][j] - mean) * rstd * gamma[c] + beta[c]);
            }
            *(bf16x8*)(y + row * C + vi * 8) = o;
        }
    }
}

}  // namespace

// chunk count used by both the stats launcher and its callers (partials buffer = chunks * N * C * 2 floats)
extern "C" int dc_gn_stats_chunks(long long HW, int C)
{
    const int nv = C >> 3;
    const int tpp = nv < 256 ? nv : 256;
    const int ppb = 256 / tpp;                      // pixels processed concurrently by one workgroup
    long long chunks = HW / (8LL * ppb);            // >= 8 pixels per pixel-lane
    if (chunks < 1) chunks = 1;
    if (chunks > 64) chunks = 64;
    return (int)chunks;
}

extern "C" int dc_gn_stats_nhwc_bf16(const void* x, float* partials, int N, long long HW, int C, void* stream)
{
    if (!x || !partials || N <= 0 || HW <= 0 || C <= 0 || (C & 7) || C > 8192) return DC_ERR_INVALID;
    const int chunks = dc_gn_stats_chunks(HW, C);
    const int ppb_pix = (int)((HW + chunks - 1) / chunks);
    const int nv = C >> 3, tpp = nv < 256 ? nv : 256, ppb = 256 / tpp;
    const dim3 grid(chunks, N);
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), (size_t)ppb * C * 2 * sizeof(float), (hipStream_t)stream,
                       (const bf16_t*)x, partials, HW, C, ppb_pix, N);
    return dc_launch_status();
}

extern "C" int dc_gn_finalize(const float* sums1, int C1, int chunks1, const float* sums2, int C2, int chunks2,
                              const float* gamma, const float* beta, float* ab, int N, int groups, long long HW, float eps,
                              void* stream)
{
    const int C = C1 + C2;
    if (!sums1 || !ab || N <= 0 || groups <= 0 || C <= 0 || C % groups || (C2 && !sums2) || chunks1 <= 0) return DC_ERR_INVALID;
    const float inv_count = 1.0f / ((float)HW * (float)(C / groups));
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(N * groups), dim3(256), 0, (hipStream_t)stream,
                       sums1, C1, chunks1, sums2, C2, chunks2, gamma, beta, ab, N, groups, inv_count, eps);
    return dc_launch_status();
}

extern "C" int dc_gn_direct_nhwc_bf16(const void* x1, int C1, const void* x2, int C2, const float* gamma, const float* beta,
                                      float* ab, int N, long long HW, int groups, float eps, void* stream)
{
    const int C = C1 + C2;
    if (!x1 || !ab || N <= 0 || HW <= 0 || groups <= 0 || C <= 0 || C % groups || (C2 && !x2)) return DC_ERR_INVALID;
    const int cpg = C / groups;
    if ((cpg & 1) || (C1 & 1) || (C2 & 1)) return DC_ERR_INVALID;       // bf16 pairs: group and source widths must be even
    const int npairs = cpg >> 1;
    const int tp = npairs <= 8 ? 8 : (npairs <= 16 ? 16 : 32);          // lanes per pixel
    const float inv_count = 1.0f / ((float)HW * (float)cpg);
    hipLaunchKernelGGL(gn_direct_kernel, dim3(N * groups), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x1, C1,
                       (const bf16_t*)x2, C2, gamma, beta, ab, HW, groups, tp, inv_count, eps);
    return dc_launch_status();
}

extern "C" int dc_gn_apply_nhwc_bf16(const void* x1, int C1, const void* x2, int C2, const float* ab, void* y, int N,
                                     long long HW, int silu, void* stream)
{
    const int C = C1 + C2;
    if (!x1 || !ab || !y || N <= 0 || HW <= 0 || (C1 & 7) || (C2 & 7) || (C2 && !x2)) return DC_ERR_INVALID;
    const int nv = C >> 3, tpp = nv < 256 ? nv : 256, ppb = 256 / tpp;
    // >= 16 pixels per pixel-lane and workgroup (four trips of four), at most ~4096 workgroups in all — but small maps (the 8x8 and
    // 16x16 levels: 128-512 workgroups by that rule, half the CUs idle and the launch latency-bound) go down to ONE trip of four
    // pixels per lane until ~1024 workgroups exist
    long long chunks = HW / (16LL * ppb);
    if (chunks < 1) chunks = 1;
    while (chunks * N < 1024 && chunks * 2 * 4 * ppb <= HW) chunks *= 2;
    const long long cap = 4096 / N > 0 ? 4096 / N : 1;
    if (chunks > cap) chunks = cap;
    if (chunks < 1) chunks = 1;
    const int pix_per_block = (int)((HW + chunks - 1) / chunks);
    static const int unr = DC_KNOB("DC_GN_UNR", 4), nt = DC_KNOB("DC_GN_NT", 0);     // developer A/B knobs
    const dim3 grid((unsigned)dc_cdiv(HW, pix_per_block), N);
#define DC_GN_APPLY(U, T) hipLaunchKernelGGL((gn_apply_kernel<U, T>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x1, C1, \
                                             (const bf16_t*)x2, C2, ab, (bf16_t*)y, HW, pix_per_block, silu)
    if (unr == 8) { if (nt) DC_GN_APPLY(8, true); else DC_GN_APPLY(8, false); }
    else if (unr == 2) { if (nt) DC_GN_APPLY(2, true); else DC_GN_APPLY(2, false); }
    else { if (nt) DC_GN_APPLY(4, true); else DC_GN_APPLY(4, false); }
#undef DC_GN_APPLY
    return dc_launch_status();
}

extern "C" int dc_fdn_modulate_nhwc_bf16(const void* x, const float* ab, const void* gamma, const void* beta, void* y,
                                         int N, int Bp, long long HW, int C, void* stream)
{
    if (!x || !ab || !gamma || !beta || !y || N <= 0 || Bp <= 0 || HW <= 0 || (C & 7)) return DC_ERR_INVALID;
    const long long total_vec = (long long)N * HW * (C >> 3);
    const int grid = (int)min((long long)4096, (total_vec + 255) / 256);
    hipLaunchKernelGGL(fdn_modulate_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ab,
                       (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, Bp, HW, C, total_vec);
    return dc_launch_status();
}

// Row statistics for the LayerNorm folded into a linear (dc_conv_desc.ln_stats): (sum, sum of squares) of every row of
// x [M][C] bf16 — used when the launch that produced x could not emit them itself.  One wave per row, 16 bytes per lane
// and trip, fixed-order reductions.
__global__ __launch_bounds__(256) void row_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ stats, long long M, int C)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = C >> 3;
    float s1 = 0.f, s2 = 0.f;
    for (int vi = lane; vi < nv; vi += 64) {
        const bf16x8 r = *(const bf16x8*)(x + row * C + vi * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = (float)r[j];
            s1 += f;
            s2 += f * f;
        }
    }
    s1 = dc_wave_sum(s1);
    s2 = dc_wave_sum(s2);
    if (lane == 0) {
        stats[row * 2] = s1;
        stats[row * 2 + 1] = s2;
    }
}

// (sum, sum of squares) partials [M][parts][2] -> (mean, rstd) [M][2] of a LayerNorm over C channels.
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ partials, float* __restrict__ mr, long long M,
                                                          int parts, float inv_c, float eps)
{
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= M) return;
    float2 o;
    dc_ln_mean_rstd(partials + row * parts * 2, parts, inv_c, eps, o.x, o.y);
    *(float2*)(mr + row * 2) = o;
}

extern "C" int dc_ln_finalize(const float* partials, float* mean_rstd, long long M, int parts, int C, float eps, void* stream)
{
    if (!partials || !mean_rstd || M <= 0 || parts <= 0 || C <= 0) return DC_ERR_INVALID;
    hipLaunchKernelGGL(ln_finalize_kernel, dim3(dc_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, partials, mean_rstd, M, parts,
                       1.0f / (float)C, eps);
    return dc_launch_status();
}

extern "C" int dc_row_stats_bf16(const void* x, float* stats, long long M, int C, void* stream)
{
    if (!x || !stats || M <= 0 || C <= 0 || (C & 7)) return DC_ERR_INVALID;
    hipLaunchKernelGGL(row_stats_kernel, dim3(dc_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, stats, M, C);
    return dc_launch_status();
}

extern "C" int dc_layernorm_bf16(const void* x, const float* gamma, const float* beta, void* y, long long M, int C,
                                 float eps, void* stream)
{
    if (!x || !gamma || !beta || !y || M <= 0 || C <= 0 || (C & 7) || C > 64 * 8 * 4) return DC_ERR_INVALID;
    const dim3 grid(dc_cdiv(M, 4));
    hipStream_t st = (hipStream_t)stream;
    if (C <= 512) hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, M, C, eps);
    else if (C <= 1024) hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, M, C, eps);
    else hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, M, C, eps);
    return dc_launch_status();
}
