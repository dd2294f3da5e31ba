// MFMA implicit-GEMM convolution / linear for gfx950:  out[m][n] = sum_k X(m,k) * W[n][k]
//   m = (sample, oy, ox) pixel row of an NHWC bf16 activation, n = output channel,
//   k = (tap, cin) with cin contiguous -> the A-operand gather is a 128-byte row piece per (pixel, tap).
// Replaces F.conv2d / nn.Linear inside the diffusers blocks the reference calls (flownet.py:83-124,
// pipeline.py:358-367,391) — see include/diffcodec_hip.h.
//
// Structure (per workgroup, 256 threads = 4 waves, one output tile BM x BN, BK = 64):
//   global -> registers (16 B / lane, 8 lanes cover one 128-B row piece) -> [GN affine + SiLU in fp32] ->
//   LDS (chunk-major image, XOR-swizzled so that both the ds_write_b128 and the MFMA-fragment ds_read_b128
//   are bank-conflict-free) -> v_mfma_f32_16x16x32_bf16, fp32 accumulate.  LDS is double-buffered; the next
//   K-tile's global loads are issued before the current tile's MFMAs (one barrier per K-step).
//   The MFMA computes the transposed tile (A-operand = weights, B-operand = pixels) so that each lane ends up
//   with 4 consecutive output channels of one pixel -> 8-byte NHWC stores, float4 bias loads.
#include "dc_common.h"
#include "../../include/diffcodec_hip.h"

// conv3x3_tile.hip: LDS-staged 2D-tile kernel for 3x3 stride-1 convs with tile-aligned outputs
int dc_conv3x3_tile_supported(const dc_conv_desc& d);
int dc_conv3x3_tile_launch(const dc_conv_desc& d, hipStream_t st);
// gemm_dma.hip: LDS-DMA pipelined GEMM for 1x1 convs / linears without a load-side transform
int dc_gemm_dma_supported(const dc_conv_desc& d);
int dc_gemm_dma_launch(const dc_conv_desc& d, hipStream_t st);
int dc_gemm_dma_gn_chunks(const dc_conv_desc& d);
int dc_conv3x3_tile_gn_chunks(const dc_conv_desc& d);

namespace {

constexpr int BK = 64;

template <int WM, int WN, int TM, int TN, bool KS3, bool GN>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const dc_conv_desc d)
{
    constexpr int BM = WM * TM * 16;
    constexpr int BN = WN * TN * 16;
    constexpr int PA = BM / 32;          // A (pixel) rows per thread per K-step
    constexpr int PB = BN / 32;          // B (weight) rows per thread per K-step
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BN * 128;
    constexpr int BUF_BYTES = A_BYTES + B_BYTES;
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int q = tid & 7;               // 16-byte chunk of the 128-byte K piece
    const int r0 = tid >> 3;             // row within a 32-row pass

    const int HoWo = d.Ho * d.Wo;
    const int M = d.N * HoWo;
    const int Cin = d.C1 + d.C2;
    const int nkc = Cin >> 6;
    const int taps = KS3 ? 9 : 1;
    const int KT = taps * nkc;

    // ---- tile coordinates; XCD-aware remap: blocks b and b+8 share an XCD, give each XCD a contiguous range
    const int n_tiles = (d.Cout + BN - 1) / BN;
    const int m_tiles = (M + BM - 1) / BM;
    const int nblk = n_tiles * m_tiles;
    int bid = blockIdx.x;
    {
        const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + idx;
    }
    const int tile_n = bid % n_tiles;
    const int tile_m = bid / n_tiles;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    int kt_begin = 0, kt_end = KT;
    if (d.splitk > 1) {
        const int per = (KT + d.splitk - 1) / d.splitk;
        kt_begin = blockIdx.y * per;
        kt_end = min(KT, kt_begin + per);
        if (kt_begin >= kt_end) return;
    }

    // ---- per-thread A row bookkeeping
    int a_n[PA], a_oy[PA], a_ox[PA];
    bool a_ok[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int m = m0 + r0 + 32 * i;
        a_ok[i] = m < M;
        const int mm = a_ok[i] ? m : 0;
        a_n[i] = mm / HoWo;
        const int rem = mm - a_n[i] * HoWo;
        a_oy[i] = rem / d.Wo;
        a_ox[i] = rem - a_oy[i] * d.Wo;
    }
    const bf16_t* __restrict__ x1 = (const bf16_t*)d.x1;
    const bf16_t* __restrict__ x2 = (const bf16_t*)d.x2;
    const bf16_t* __restrict__ wgt = (const bf16_t*)d.w;

    u32x4 ra[PA], rb[PB];
    unsigned va = 0;                     // validity bits of the staged A rows
    int cur_c = 0;
    int a_nb[PA];                        // gn_ab row of each staged A row's sample
#pragma unroll
    for (int i = 0; i < PA; ++i) a_nb[i] = GN ? a_n[i] % d.gn_batch : 0;

    auto issue_loads = [&](int kt) {
        const int tap = KS3 ? kt / nkc : 0;
        const int cc = KS3 ? kt - tap * nkc : kt;
        const int c = cc * 64 + q * 8;           // channel within cat[x1,x2]
        cur_c = c;
        const bool second = c >= d.C1;
        const bf16_t* __restrict__ src = second ? x2 : x1;
        const int cs = second ? d.C2 : d.C1;
        const int co = second ? c - d.C1 : c;
        const int ky = KS3 ? tap / 3 : 0;
        const int kx = KS3 ? tap - ky * 3 : 0;
        va = 0;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            bool ok = a_ok[i];
            long long pix;
            if (KS3) {
                int uy = a_oy[i] * d.stride + ky - d.pad;
                int ux = a_ox[i] * d.stride + kx - d.pad;
                if (d.upsample) {
                    ok = ok && uy >= 0 && ux >= 0 && uy < 2 * d.H && ux < 2 * d.W;
                    uy >>= 1;
                    ux >>= 1;
                } else {
                    ok = ok && uy >= 0 && ux >= 0 && uy < d.H && ux < d.W;
                }
                pix = ((long long)a_n[i] * d.H + uy) * d.W + ux;
            } else {
                pix = (long long)(m0 + r0 + 32 * i);
            }
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) {
                v = *(const u32x4*)(src + pix * cs + co);
                va |= 1u << i;
            }
            ra[i] = v;
        }
        const long long wk = (long long)tap * Cin + cc * 64 + q * 8;
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int n = n0 + r0 + 32 * i;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (n < d.Cout) v = *(const u32x4*)(wgt + (long long)n * taps * Cin + wk);
            rb[i] = v;
        }
    };

    auto store_lds = [&](int buf) {
        char* sA = smem + buf * BUF_BYTES;
        char* sB = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            u32x4 v = ra[i];
            if (GN) {
                if (va & (1u << i)) {
                    // (scale, shift) of this row's sample: 64 B, L1-resident (shared by the 32 rows of a pass)
                    const float* __restrict__ abp = d.gn_ab + ((long long)a_nb[i] * Cin + cur_c) * 2;
                    f32x4 g[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = *(const f32x4*)(abp + 4 * j);
                    uint32_t o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t u = v[j];
                        float lo = __uint_as_float(u << 16);
                        float hi = __uint_as_float(u & 0xffff0000u);
                        lo = lo * g[j][0] + g[j][1];
                        hi = hi * g[j][2] + g[j][3];
                        if (d.gn_silu) {
                            lo = dc_silu(lo);
                            hi = dc_silu(hi);
                        }
                        bf16x2 p = {(bf16_t)lo, (bf16_t)hi};
                        o[j] = *(uint32_t*)&p;
                    }
                    v = u32x4{o[0], o[1], o[2], o[3]};
                }
            }
            const int r = r0 + 32 * i;
            *(u32x4*)(sA + q * (BM * 16) + ((r ^ q) << 4)) = v;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int r = r0 + 32 * i;
            *(u32x4*)(sB + q * (BN * 16) + ((r ^ q) << 4)) = rb[i];
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const char* sA = smem + buf * BUF_BYTES;
        const char* sB = sA + A_BYTES;
        const int fr = lane & 15;
        const int fq = lane >> 4;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int qq = 4 * s + fq;
            bf16x8 wf[TN], xf[TM];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int row = (wn * TN + tn) * 16 + fr;
                wf[tn] = *(const bf16x8*)(sB + qq * (BN * 16) + ((row ^ qq) << 4));
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int row = (wm * TM + tm) * 16 + fr;
                xf[tm] = *(const bf16x8*)(sA + qq * (BM * 16) + ((row ^ qq) << 4));
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tn], xf[tm], acc[tn][tm], 0, 0, 0);
        }
    };

    // ---- main loop
    issue_loads(kt_begin);
    store_lds(0);
    __syncthreads();
    int buf = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = kt + 1 < kt_end;
        if (more) issue_loads(kt + 1);
        compute(buf);
        if (more) store_lds(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    const long long slab = (long long)d.N * d.Ho * d.Wo * d.Cout;   // elements per split-K slab
    // ---- epilogue: lane holds out[m = .. + (lane&15)][n = .. + 4*(lane>>4) + 0..3]
    const int fr = lane & 15;
    const int fq = lane >> 4;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m = m0 + (wm * TM + tm) * 16 + fr;
        if (m >= M) continue;
        const int nimg = m / HoWo;
        if (d.epilogue == 1) {
            const int half = d.Cout >> 1;
            bf16_t* __restrict__ o = (bf16_t*)d.out;
#pragma unroll
            for (int tp = 0; tp < TN / 2; ++tp) {
                const int nb = n0 + (wn * TN + 2 * tp) * 16 + 4 * fq;      // packed row of the hidden half
                if (nb >= d.Cout) continue;
                f32x4 h = acc[2 * tp][tm];
                f32x4 g = acc[2 * tp + 1][tm];
                if (d.bias) {
                    h += *(const f32x4*)(d.bias + nb);
                    g += *(const f32x4*)(d.bias + nb + 16);
                }
                const int col = ((n0 + (wn * TN + 2 * tp) * 16) >> 1) + 4 * fq;
                bf16x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)(h[r] * dc_gelu_erf(g[r]));
                *(bf16x4*)(o + (long long)m * half + col) = pk;
            }
            continue;
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int nb = n0 + (wn * TN + tn) * 16 + 4 * fq;
            if (nb >= d.Cout) continue;
            f32x4 v = acc[tn][tm];
            const long long off = (long long)m * d.Cout + nb;
            if (d.splitk > 1) {                                   // this split's own fp32 slab: plain stores, no atomics
                *(f32x4*)(d.splitk_ws + (long long)blockIdx.y * slab + off) = v;
                continue;
            }
            if (d.bias) v += *(const f32x4*)(d.bias + nb);
            if (d.row_add) v += *(const f32x4*)(d.row_add + (long long)nimg * d.row_add_stride + nb);
            if (d.act) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = dc_act(v[r], d.act);
            }
            if (d.residual) v = dc_scale_res(v, d.out_scale, *(const bf16x4*)((const bf16_t*)d.residual + off));
            else v *= d.out_scale;
            if (d.out_f32) {
                *(f32x4*)((float*)d.out + off) = v;
            } else {
                bf16x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)v[r];
                *(bf16x4*)((bf16_t*)d.out + off) = pk;
            }
        }
    }
}

// Second pass of split-K: sum of the per-split slabs -> bias / row_add / scale / residual -> output.
__global__ void splitk_finish_kernel(const dc_conv_desc d, long long total4)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const long long off = i * 4;
    const int nb = (int)(off % d.Cout);
    const long long m = off / d.Cout;
    const int nimg = (int)(m / (d.Ho * d.Wo));
    // every operand is FETCHED before anything is added (left as `v += load` per split, hipcc emits load - vmcnt(0) - add per split and
    // per epilogue operand: five to twenty dependent memory round trips in a kernel that small launches run latency-bound); the
    // additions keep their fixed order (split 0, 1, 2, ... then bias, row_add): deterministic, and the same bits as before
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bv = d.bias ? *(const f32x4*)(d.bias + nb) : zero;
    const f32x4 ra = d.row_add ? *(const f32x4*)(d.row_add + (long long)nimg * d.row_add_stride + nb) : zero;
    bf16x4 rr = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    if (d.residual) rr = *(const bf16x4*)((const bf16_t*)d.residual + off);
    const float* __restrict__ slab = d.splitk_ws + off;
    const long long ss = total4 * 4;                            // elements per split slab
    f32x4 v = *(const f32x4*)slab;
    int k = 1;
    for (; k + 4 <= d.splitk; k += 4) {
        const f32x4 a0 = *(const f32x4*)(slab + k * ss), a1 = *(const f32x4*)(slab + (k + 1) * ss);
        const f32x4 a2 = *(const f32x4*)(slab + (k + 2) * ss), a3 = *(const f32x4*)(slab + (k + 3) * ss);
        v += a0;
        v += a1;
        v += a2;
        v += a3;
    }
    for (; k < d.splitk; ++k) v += *(const f32x4*)(slab + k * ss);
    if (d.bias) v += bv;
    if (d.row_add) v += ra;
    if (d.act) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = dc_act(v[r], d.act);
    }
    if (d.residual) v = dc_scale_res(v, d.out_scale, rr);
    else v *= d.out_scale;
    if (d.out_f32) {
        *(f32x4*)((float*)d.out + off) = v;
    } else {
        bf16x4 pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)v[r];
        *(bf16x4*)((bf16_t*)d.out + off) = pk;
    }
}

template <int WM, int WN, int TM, int TN>
int launch_cfg(const dc_conv_desc& d, hipStream_t st)
{
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    const int M = d.N * d.Ho * d.Wo;
    const int nblk = dc_cdiv(M, BM) * dc_cdiv(d.Cout, BN);
    const dim3 grid(nblk, d.splitk > 1 ? d.splitk : 1);
    const size_t lds = 2 * (BM + BN) * 128;
    const bool gn = d.gn_ab != nullptr;
#define DC_IGEMM_LAUNCH(KS3, GN)                                                                              \
    do {                                                                                                      \
        auto kern = igemm_kernel<WM, WN, TM, TN, KS3, GN>;                                                    \
        static std::atomic<unsigned long long> attr_done{0};                                                  \
        dc_set_max_dyn_lds((const void*)kern, (int)lds, attr_done);                                           \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, d);                                                \
    } while (0)
    if (d.ksize == 3) {
        if (gn) DC_IGEMM_LAUNCH(true, true);
        else DC_IGEMM_LAUNCH(true, false);
    } else {
        if (gn) DC_IGEMM_LAUNCH(false, true);
        else DC_IGEMM_LAUNCH(false, false);
    }
#undef DC_IGEMM_LAUNCH
    return dc_launch_status();
}

}  // namespace

extern "C" int dc_conv_gn_part_chunks(const dc_conv_desc* dp)
{
    if (!dp || dp->splitk > 1) return 0;
    if (dc_gemm_dma_supported(*dp)) return dc_gemm_dma_gn_chunks(*dp);
    if (dc_conv3x3_tile_supported(*dp)) return dc_conv3x3_tile_gn_chunks(*dp);
    return 0;                                               // gather GEMM (strided / GN-on-load 1x1): no statistics epilogue
}

extern "C" long long dc_conv_igemm_ws_bytes(const dc_conv_desc* d)
{
    if (!d || d->splitk <= 1) return 0;
    return (long long)d->N * d->Ho * d->Wo * d->Cout * 4 * d->splitk;
}

extern "C" int dc_conv_igemm_bf16(const dc_conv_desc* dp, void* stream)
{
    if (!dp) return DC_ERR_INVALID;
    dc_conv_desc d = *dp;
    hipStream_t st = (hipStream_t)stream;
    const int Cin = d.C1 + d.C2;
    if (!d.x1 || !d.w || !d.out) return DC_ERR_INVALID;
    if (d.ksize != 1 && d.ksize != 3) return DC_ERR_INVALID;
    if (Cin <= 0 || (Cin & 63) || (d.C1 & 63) || (d.C2 && !d.x2)) return DC_ERR_INVALID;
    // Cout: multiples of 16 everywhere; the halo-tile 3x3 kernel alone also takes multiples of 4 (clamped weight rows,
    // guarded 4-channel epilogue) — UNet conv_out 320->4 runs there on one mostly empty N-tile instead of on the VALU
    if (d.Cout <= 0 || (d.Cout & 3)) return DC_ERR_INVALID;
    if ((d.Cout & 15) && !(dc_conv3x3_tile_supported(d) && !dc_gemm_dma_supported(d) && d.splitk <= 1)) return DC_ERR_INVALID;
    if (d.N <= 0 || d.H <= 0 || d.W <= 0 || d.Ho <= 0 || d.Wo <= 0) return DC_ERR_INVALID;
    if (d.ksize == 1 && (d.stride != 1 || d.upsample || d.Ho != d.H || d.Wo != d.W)) return DC_ERR_INVALID;
    if (d.ksize == 3) {
        if (d.stride != 1 && d.stride != 2) return DC_ERR_INVALID;
        const int Hin = d.upsample ? 2 * d.H : d.H, Win = d.upsample ? 2 * d.W : d.W;
        const int ho = d.pad ? (Hin + 2 - 3) / d.stride + 1 : (Hin + 1 - 3) / d.stride + 1;
        const int wo = d.pad ? (Win + 2 - 3) / d.stride + 1 : (Win + 1 - 3) / d.stride + 1;
        if (ho != d.Ho || wo != d.Wo) return DC_ERR_INVALID;
    }
    if (d.gn_ab && d.gn_batch <= 0) return DC_ERR_INVALID;
    if (d.act < 0 || d.act > 2) return DC_ERR_INVALID;
    if (d.splitk < 1) d.splitk = 1;
    if (d.row_add_stride == 0) d.row_add_stride = d.Cout;
    if (d.epilogue == 1 && (d.splitk > 1 || (d.Cout & 31) || d.residual || d.row_add || d.out_f32)) return DC_ERR_INVALID;
    if (d.splitk > 1) {
        if (!d.splitk_ws) return DC_ERR_INVALID;
        // every split must own a non-empty K range, or its slab would stay unwritten: shrink to the fixpoint of
        // splitk = ceil(KT / ceil(KT / splitk)) (KT = 64-wide K steps, the unit all three kernels partition by)
        const int nkc = (d.C1 + d.C2) >> 6;
        const bool tile = !dc_gemm_dma_supported(d) && dc_conv3x3_tile_supported(d);     // splits channel chunks, not taps
        const int KT = (d.ksize == 3 && !tile) ? 9 * nkc : nkc;
        for (;;) {
            const int per = (KT + d.splitk - 1) / d.splitk, s2 = (KT + per - 1) / per;
            if (s2 == d.splitk) break;
            d.splitk = s2;
        }
    }
    const long long M = (long long)d.N * d.Ho * d.Wo;
    // Tile choice: wide-N tile (160) when Cout is a multiple of 160 (all SD-1.5 UNet widths), else 128;
    // tall-M tile (128) only when that still yields >= 2 workgroups per CU.
    const bool n160 = (d.Cout % 160 == 0) && d.epilogue == 0;
    const int bn = n160 ? 160 : 128;
    const long long big_tiles = ((M + 127) / 128) * ((d.Cout + bn - 1) / bn) * d.splitk;
    if ((d.ln_stats || d.stats_out) && !dc_gemm_dma_supported(d)) return DC_ERR_INVALID;   // LDS-DMA GEMM epilogue only
    if (d.gn_part_out && dc_conv_gn_part_chunks(&d) == 0) return DC_ERR_INVALID;
    int rc;
    if (dc_gemm_dma_supported(d)) rc = dc_gemm_dma_launch(d, st);
    else if (dc_conv3x3_tile_supported(d)) rc = dc_conv3x3_tile_launch(d, st);
    else if (big_tiles >= 512) rc = n160 ? launch_cfg<2, 2, 4, 5>(d, st) : launch_cfg<2, 2, 4, 4>(d, st);
    else rc = n160 ? launch_cfg<2, 2, 2, 5>(d, st) : launch_cfg<2, 2, 2, 4>(d, st);
    if (rc != DC_OK) return rc;
    if (d.splitk > 1) {
        const long long total4 = M * d.Cout / 4;
        hipLaunchKernelGGL(splitk_finish_kernel, dim3(dc_cdiv(total4, 256)), dim3(256), 0, st, d, total4);
        return dc_launch_status();
    }
    return DC_OK;
}
