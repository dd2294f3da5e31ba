// Wide-tile GEMM  out[m][n] = sum_k X[m][k] * W[n][k]  for gfx950: the long-K 1x1 / linear launches (attention out-projections,
// feed-forward, time-embedding-free shortcuts; same call sites as gemm_dma.hip) where the 128-row kernel is bound by the
// bytes it pulls through the CU's vector-memory path, not by the matrix pipe (gemm_dma with the K-loop DMA switched off ran
// 1.15-1.65x faster on these shapes: tools/ab experiment of round 2).
//
// One 512-thread workgroup per CU owns a 256 x BN output tile (BN = 160 | 128): 28 % fewer operand bytes per FLOP than two
// 128-row workgroups, which is what the LDS-DMA path is short of.  With a single workgroup per CU nothing overlaps for free,
// so the two halves of the workgroup are run in anti-phase ("ping-pong"):
//   group 0 = waves 0-3 (rows 0-127), group 1 = waves 4-7 (rows 128-255); each group is a 2(m) x 2(n) arrangement of
//   64 x (BN/2) wave tiles — the per-wave code of gemm_dma.hip;
//   a K-step of a wave is  [barrier] L: fragment reads + its share of the LDS-DMA for stage k+2  [barrier] M: the MFMAs;
//   group 1 passes one extra barrier up front, so while one group computes the other loads: each SIMD hosts one wave of
//   each group, its matrix pipe always has a wave in its M phase, and LDS reads / DMA issue hide behind the partner's MFMAs.
// Ring of 3 stages (the whole 160 KB of LDS).  Hazards, with I(2k) / I(2k+1) the barrier intervals in which group 0 / 1 read
// stage k:  RAW — every wave waits for its own pieces of stage k+1 (counted vmcnt, stage k+2 stays in flight) and for its
// fragment reads (lgkmcnt(0)) at the END of its L phase, i.e. before the barrier that precedes every read of stage k+1 by
// either group;  WAR — stage k+2 reuses the slot of stage k-1, whose last reads (group 1, I(2k-1)) are retired by that
// same lgkmcnt(0) before the barrier that opens I(2k), the earliest interval in which a piece of stage k+2 is issued.
// No DMA is issued past the last stage (the epilogue stages its rows through the ring's LDS), so the final waits are vmcnt(0).
#include "dc_common.h"
#include "../../include/diffcodec_hip.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int N>
__device__ __forceinline__ void wait_vm_lgkm()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wg_barrier()
{
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// EPI as in gemm_dma.hip: 1 bias, 2 bias + scale + residual, 3 folded LayerNorm + bias, 4 GEGLU, 5 folded LayerNorm + GEGLU.
// ST (EPI 1 / 2): bit 0 = LayerNorm row statistics of the output requested (`stats_out`), bit 1 = GroupNorm partials (`gn_part_out`).
// A compile-time choice: the sums cost 24 vector operations per 16 x 16 block (480 per wave and tile), most launches of this kernel
// request one kind or none (the feed-forward down-projection), and run-time branches around them inside the unrolled epilogue make
// hipcc spill thousands of dwords.
template <int TN, int EPI, int ST>
__global__ __launch_bounds__(512, 2) void gemm_wide_kernel(const dc_conv_desc d)
{
    constexpr int TM = 4, NST = 3;
    constexpr int BM = 256, BN = 2 * TN * 16;
    constexpr int ROWS = BM + BN;
    constexpr int STAGE = ROWS * 128;
    constexpr int NP = ROWS / 8;                      // LDS-DMA pieces (8 rows x 128 B) per stage
    constexpr int PMAX = (NP + 7) / 8;                // pieces per wave (waves with index < NP % 8 issue PMAX, the rest PMAX - 1)
    constexpr bool e_geglu = EPI >= 4, e_ln = EPI == 3 || EPI == 5, e_res = EPI == 2;
    static_assert(NP % 8 == 0 || NP % 8 == 4, "the two wave groups must be uniform in their piece count");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                        // 0: rows 0-127, 1: rows 128-255
    const int wm = wave & 1, wn = (wave >> 1) & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int wrow = grp * 128 + wm * TM * 16;        // first row of this wave's tiles inside the workgroup tile
    static_assert(ST == 0 || EPI == 1 || EPI == 2, "statistics ride on the bias / residual epilogues only");
    constexpr bool e_stats = (ST & 1) != 0, e_gnpart = (ST & 2) != 0;

    const int HoWo = d.Ho * d.Wo;
    const int M = d.N * HoWo;
    const int K = d.C1 + d.C2;
    const int nk = K >> 6;
    const int n_tiles = (d.Cout + BN - 1) / BN;
    const int m_tiles = (M + BM - 1) / BM;
    const int nblk = n_tiles * m_tiles;
    int bid = blockIdx.x;
    {
        const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + idx;
    }
    const int tile_n = bid % n_tiles, tile_m = bid / n_tiles;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-lane DMA sources: piece g = wave + 8 i covers stage rows [8g, 8g+8); lane s -> row 8g + (s>>3), LDS slot s&7,
    //      which must hold source chunk (s&7) ^ (row&7)
    //      Addresses are a workgroup-uniform base (SGPRs: tile origin of the operand) + a 32-bit per-lane offset that also carries the K
    //      offset of the stage (one v_add_u32 per piece; a 64-bit per-lane pointer costs two registers and a 64-bit vector add per piece,
    //      issued while the partner group's MFMAs hold the vector port — gemm_p8.hip measured 2-3 % for the same change).
    uint32_t off1[PMAX], off2[PMAX];
    const int c1_steps = d.C1 >> 6;
    const int my_pieces = (NP - wave + 7) / 8;        // wave-uniform: PMAX or PMAX - 1
    const char* const bx1 = (const char*)d.x1 + (long long)m0 * d.C1 * 2;
    const char* const bx2 = d.x2 ? (const char*)d.x2 + (long long)m0 * d.C2 * 2 : nullptr;
    const char* const bw = (const char*)d.w + (long long)n0 * K * 2;
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
        int g = wave + 8 * i;
        g = g < NP ? g : NP - 1;
        const int row = g * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        if (row < BM) {
            int m = m0 + row;
            m = m < M ? m : M - 1;                    // clamp: rows past M are computed and discarded
            off1[i] = (uint32_t)((m - m0) * d.C1 + chunk * 8) * 2u;
            off2[i] = (uint32_t)((m - m0) * d.C2 + chunk * 8) * 2u;
        } else {
            int n = n0 + row - BM;
            n = n < d.Cout ? n : d.Cout - 1;
            off1[i] = (uint32_t)((n - n0) * K + chunk * 8) * 2u;
            off2[i] = 0;
        }
    }
    auto issue_stage = [&](int kt, int slot) {
        char* base = smem + slot * STAGE;
        const bool second = kt >= c1_steps;
        const uint32_t ko1 = (uint32_t)kt * 128u, ko2 = (uint32_t)(kt - c1_steps) * 128u;
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            if (i == PMAX - 1 && my_pieces < PMAX) break;                        // wave-uniform
            const int g = wave + 8 * i;
            const bool is_a = g * 8 < BM;                                        // wave-uniform
            const char* p = is_a ? ((second ? bx2 : bx1) + (size_t)(uint32_t)((second ? off2[i] : off1[i]) + (second ? ko2 : ko1)))
                                 : (bw + (size_t)(uint32_t)(off1[i] + ko1));
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(base + g * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 wf[2][TN], xf[2][TM];
    auto load_frags = [&](int slot) {
        const char* sA = smem + slot * STAGE;
        const char* sB = sA + BM * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int swz = ((4 * s + fq) ^ (fr & 7)) << 4;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) xf[s][tm] = *(const bf16x8*)(sA + (wrow + tm * 16 + fr) * 128 + swz);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) wf[s][tn] = *(const bf16x8*)(sB + ((wn * TN + tn) * 16 + fr) * 128 + swz);
        }
    };
    auto mfma_frags = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][tn], xf[s][tm], acc[tn][tm], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // bias / LayerNorm operands for this lane's outputs: fetched now, first used in the epilogue
    f32x4 bv[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int nb = n0 + (wn * TN + tn) * 16 + 4 * fq;
        bv[tn] = (d.bias && nb < d.Cout) ? *(const f32x4*)(d.bias + nb) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x2 ln_mr[TM];
    f32x4 cs[TN];
    if (e_ln) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            int m = m0 + wrow + tm * 16 + fr;
            m = m < M ? m : M - 1;
            ln_mr[tm] = *(const f32x2*)(d.ln_stats + (long long)m * 2);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int nb = n0 + (wn * TN + tn) * 16 + 4 * fq;
            cs[tn] = nb < d.Cout ? *(const f32x4*)(d.ln_colsum + nb) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // the operand loads above are ordinary VMEM operations in front of the DMA pieces in this wave's queue: retire them now so
    // that the counted waits below see DMA pieces only
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- pipeline.  P = this wave's pieces per stage; the two groups differ when NP is not a multiple of 8.
    auto pipeline = [&](auto pc) {
        constexpr int P = decltype(pc)::value;
        issue_stage(0, 0);
        if (nk > 1) issue_stage(1, 1);
        if (nk > 1) wait_vm_lgkm<P>();
        else wait_vm_lgkm<0>();                       // this wave's pieces of stage 0 have landed
        if (grp == 1) wg_barrier();                   // the extra barrier that puts group 1 half a K-step behind group 0
        for (int k = 0; k < nk; ++k) {
            wg_barrier();                             // opens this group's L phase: every wave's pieces of stage k have landed
            load_frags(k % NST);
            const bool more = k + 2 < nk;
            if (more) issue_stage(k + 2, (k + 2) % NST);
            if (more) wait_vm_lgkm<P>();              // own pieces of stage k+1 landed (stage k+2 stays in flight), reads returned
            else wait_vm_lgkm<0>();
            wg_barrier();                             // opens this group's M phase (and the other group's L phase)
            mfma_frags();
        }
        if (grp == 0) wg_barrier();                   // pairs with group 1's last L-to-M barrier
    };
    if (my_pieces == PMAX) pipeline(std::integral_constant<int, PMAX>{});
    else pipeline(std::integral_constant<int, PMAX - 1>{});

    // ---- epilogue: rows staged through LDS (every DMA has landed, every fragment read has returned: see the header), then
    //      written as whole 16-byte pieces of contiguous output rows.  Group 0 arrives while group 1 still runs its last MFMAs,
    //      which touch no LDS.
    constexpr int OC = BN;
    constexpr int PITCH = OC * 2 + 16;
    static_assert(BM * PITCH <= NST * STAGE, "output tile must fit in the stage buffers");
    bf16x4 rr[TM][TN];
    if (e_res) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int m = m0 + wrow + tm * 16 + fr;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int nb = n0 + (wn * TN + tn) * 16 + 4 * fq;
                rr[tm][tn] = (m < M && nb < d.Cout) ? *(const bf16x4*)((const bf16_t*)d.residual + (long long)m * d.Cout + nb)
                                                    : bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
            }
        }
    }
    f32x4 gs[TN], gq[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) gs[tn] = gq[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int row = wrow + tm * 16 + fr;
        const int m = m0 + row;
        if (e_geglu) {
#pragma unroll
            for (int tp = 0; tp < TN / 2; ++tp) {
                f32x4 h = acc[2 * tp][tm], g = acc[2 * tp + 1][tm];
                if (e_ln) {
                    h = dc_ln_fold(h, ln_mr[tm][0], ln_mr[tm][1], cs[2 * tp]);
                    g = dc_ln_fold(g, ln_mr[tm][0], ln_mr[tm][1], cs[2 * tp + 1]);
                }
                h += bv[2 * tp];
                g += bv[2 * tp + 1];
                bf16x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)(h[r] * dc_gelu_erf(g[r]));
                *(bf16x4*)(smem + row * PITCH + (((((wn * TN + 2 * tp) * 16) / 2 + 4 * fq) * 2) ^ dc_stage_swz(row))) = pk;
            }
        } else {
            float st1 = 0.f, st2 = 0.f;
            const float rowmask = m < M ? 1.f : 0.f;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int nl = (wn * TN + tn) * 16 + 4 * fq;
                const int nb = n0 + nl;
                f32x4 v = acc[tn][tm];
                if (e_ln) v = dc_ln_fold(v, ln_mr[tm][0], ln_mr[tm][1], cs[tn]);
                v += bv[tn];
                if (e_res) v = dc_scale_res(v, d.out_scale, rr[tm][tn]);
                else v *= d.out_scale;
                bf16x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)v[r];
                *(bf16x4*)(smem + row * PITCH + ((nl * 2) ^ dc_stage_swz(row))) = pk;
                if (e_stats) {                        // compile-time
                    const float cm = nb < d.Cout ? 1.f : 0.f;
                    st1 += cm * ((v[0] + v[1]) + (v[2] + v[3]));
                    st2 += cm * ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
                }
                if (e_gnpart) {
                    gs[tn] += rowmask * v;
                    gq[tn] += rowmask * (v * v);
                }
            }
            if (e_stats) {
                st1 += __shfl_xor(st1, 16, 64);
                st2 += __shfl_xor(st2, 16, 64);
                st1 += __shfl_xor(st1, 32, 64);
                st2 += __shfl_xor(st2, 32, 64);
                if (fq == 0 && m < M) *(f32x2*)(d.stats_out + ((long long)m * (2 * n_tiles) + tile_n * 2 + wn) * 2) = f32x2{st1, st2};
            }
        }
        __builtin_amdgcn_sched_barrier(0);            // one row group at a time (register pressure)
    }
    if (e_gnpart) {
        // chunks of 64 rows (one wave row): the launcher guarantees HoWo % 256 == 0, so a tile has one sample
        const int n_img = m0 / HoWo;
        const int chunk = ((m0 - n_img * HoWo) / 64) + grp * 2 + wm;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int nb = n0 + (wn * TN + tn) * 16 + 4 * fq;
            dc_gn_partial_store(gs[tn], gq[tn], d.gn_part_out + (((long long)chunk * d.N + n_img) * d.Cout + nb) * 2,
                                fr == 0 && nb < d.Cout);
        }
    }
    __syncthreads();
    const int out_cols = e_geglu ? d.Cout >> 1 : d.Cout;
    const int col0 = e_geglu ? n0 >> 1 : n0;
    bf16_t* __restrict__ o = (bf16_t*)d.out;
    constexpr int pieces = e_geglu ? OC / 16 : OC / 8;         // 16-byte pieces per staged row
    for (int i = tid; i < BM * pieces; i += 512) {
        const int row = i / pieces, pc = i - row * pieces;
        const int m = m0 + row, c = col0 + pc * 8;
        if (m < M && c < out_cols) *(u32x4*)(o + (long long)m * out_cols + c) = dc_stage_unswz(*(const u32x4*)(smem + row * PITCH + pc * 16), row);
    }
}

template <int TN, int EPI, int ST>
int launch_wide_st(const dc_conv_desc& d, hipStream_t st)
{
    constexpr int BN = 2 * TN * 16;
    const int M = d.N * d.Ho * d.Wo;
    const int nblk = dc_cdiv(M, 256) * dc_cdiv(d.Cout, BN);
    const size_t lds = (size_t)3 * (256 + BN) * 128;
    auto kern = gemm_wide_kernel<TN, EPI, ST>;
    static std::atomic<unsigned long long> attr_done{0};
    dc_set_max_dyn_lds((const void*)kern, (int)lds, attr_done);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), lds, st, d);
    return dc_launch_status();
}

template <int TN, int EPI>
int launch_wide(const dc_conv_desc& d, hipStream_t st)
{
    if constexpr (EPI == 1 || EPI == 2) {
        switch ((d.stats_out ? 1 : 0) | (d.gn_part_out ? 2 : 0)) {
            case 1: return launch_wide_st<TN, EPI, 1>(d, st);
            case 2: return launch_wide_st<TN, EPI, 2>(d, st);
            case 3: return launch_wide_st<TN, EPI, 3>(d, st);
            default: return launch_wide_st<TN, EPI, 0>(d, st);
        }
    } else {
        return launch_wide_st<TN, EPI, 0>(d, st);
    }
}

}  // namespace

// GroupNorm-partials chunking of the wide kernel: one chunk per 64-row wave row.
int dc_gemm_wide_gn_chunks(const dc_conv_desc& d)
{
    const long long hw = (long long)d.Ho * d.Wo;
    return hw % 256 ? 0 : (int)(hw / 64);
}

// The wide kernel takes a launch when the specialised epilogue modes apply, K is long enough for the 3-stage ring to pay for
// the unoverlapped prologue / epilogue of a one-workgroup-per-CU kernel, and the tile grid fills the chip.
// DC_GEMM_WIDE: 0 = never (A/B), 1 = default rule, 2 = whenever legal.
int dc_gemm_wide_wanted(const dc_conv_desc& d, int epi)
{
    static const int mode = DC_KNOB("DC_GEMM_WIDE", 1);
    static const int min_k = DC_KNOB("DC_GEMM_WIDE_MIN_K", 640);
    if (mode == 0 || epi < 1 || epi > 5 || d.ksize != 1 || d.gn_ab || d.splitk > 1 || d.out_f32) return 0;
    const int K = d.C1 + d.C2;
    const long long M = (long long)d.N * d.Ho * d.Wo;
    const bool geglu = epi >= 4;
    const int bn = (!geglu && d.Cout % 160 == 0) ? 160 : 128;
    if (d.Cout % bn) return 0;                              // whole N tiles only (the 128-row kernel clamps ragged ones)
    if ((d.stats_out || d.gn_part_out) && ((long long)d.Ho * d.Wo) % 256) return 0;
    if (mode == 2) return M >= 256;
    const long long tiles = ((M + 255) / 256) * (d.Cout / bn);
    // measured (tools/bench_gemm.py, same box, model batch 32): +4..13 % for K 640-2560, 0..-2 % at K 640 x N 640, -4 % at K 5120
    return K >= min_k && K <= 2560 && M % 256 == 0 && tiles >= 192;
}

int dc_gemm_wide_launch(const dc_conv_desc& d, int epi, hipStream_t st)
{
    const bool n160 = epi < 4 && d.Cout % 160 == 0;
    switch (epi) {
        case 1: return n160 ? launch_wide<5, 1>(d, st) : launch_wide<4, 1>(d, st);
        case 2: return n160 ? launch_wide<5, 2>(d, st) : launch_wide<4, 2>(d, st);
        case 3: return n160 ? launch_wide<5, 3>(d, st) : launch_wide<4, 3>(d, st);
        case 4: return launch_wide<4, 4>(d, st);
        case 5: return launch_wide<4, 5>(d, st);
        default: return DC_ERR_INVALID;
    }
}
