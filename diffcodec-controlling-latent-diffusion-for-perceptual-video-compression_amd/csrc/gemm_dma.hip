// Row-major GEMM  out[m][n] = sum_k X[m][k] * W[n][k]  for gfx950 with a multi-stage LDS-DMA pipeline — the 1x1 conv /
// nn.Linear path (Transformer2DModel.proj_out, attention to_q/k/v/out, GEGLU feed-forward, time_emb_proj, conv_shortcut,
// ControlNet zero-convs: call sites flownet.py:87-124, pipeline.py:358-367) whenever no transform is needed on load.
//
// Per workgroup: 256 threads = 4 waves as 2(m) x 2(n), tile BM x BN, BK = 64 (one 128-byte line per row per K-step).
// Both operand tiles go global -> LDS with `global_load_lds_dwordx4` (no VGPR staging, no ds_write): one wave
// instruction lands 8 rows x 128 B; the LDS image is lane-linear as the DMA requires, and the bank swizzle
// (16-byte chunk index XOR (row & 7)) is applied on the per-lane SOURCE address and again on the fragment read, so
// the v_mfma_f32_16x16x32_bf16 fragment ds_read_b128 stay conflict-free.  NST LDS stages form a ring: the DMA of
// K-step k+NST-1 is issued right after the barrier that opens step k, and a counted `s_waitcnt vmcnt(N)` (never 0 in
// the loop) leaves NST-2 stages in flight across every raw `s_barrier` — one barrier per K-step.
#include "dc_common.h"
#include "../../include/diffcodec_hip.h"
#include <cstdlib>
#include <type_traits>

// Developer-only phase stamps (tools/gemm_stamp.py builds this file with -DDC_STAMP into a scratch .so): s_memtime at
// kernel entry / after the first stage landed / after the K loop / at exit, written to the (otherwise unused) split-K
// workspace.  Never defined in the product build.
#ifdef DC_STAMP
#define DC_STAMP_AT(i)                                                                         \
    do {                                                                                       \
        if (threadIdx.x == 0 && d.splitk_ws) {                                                 \
            unsigned long long t_;                                                             \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            ((unsigned long long*)d.splitk_ws)[(long long)blockIdx.x * 8 + (i)] = t_;          \
        }                                                                                      \
    } while (0)
#else
#define DC_STAMP_AT(i)
#endif

#ifndef DC_EPI_SPECIALIZE
#define DC_EPI_SPECIALIZE 1     // developer A/B switch: 0 = every launch takes the generic run-time-flag epilogue
#endif
#ifndef DC_GEMM_PIPE
#define DC_GEMM_PIPE 1          // developer A/B switch for the scheduled K-step (see `compute`)
#endif

namespace {

typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// A_REG: the activation tile goes through registers (global_load_dwordx4 -> ds_write_b128) and only the weight tile by
// LDS-DMA.  An LDS-DMA wave-instruction costs ~100 issue cycles against ~10 for a register load + 13 for its ds_write,
// and with both operands on DMA (9 pieces per wave per K-step) the K-loop is DMA-issue-bound (900 vs 640 MFMA cycles);
// the hybrid issues 5.  Requires NST == 2.
// EPI: epilogue specialisation chosen by the launcher from the descriptor.  With every fusion a run-time flag, the unrolled
// (tm, tn) epilogue was 6,300 instructions in 336 basic blocks — a branch (and often a wait) per flag per 4 outputs — and took
// 10.7k cycles per workgroup against 9k for the whole K loop at K = 320 (phase stamps, tools/gemm_stamp.py).  The common
// fusions are compile-time here, so their epilogues are straight-line code:
//   0 generic (run-time flags: row_add, act, fp32 out, split-K, any mix)      1 bias        2 bias + scale + residual
//   3 folded LayerNorm + bias                                               4 GEGLU       5 folded LayerNorm + GEGLU
// stats_out / gn_part_out stay run-time in modes 1-2: one workgroup-uniform test outside the loops.
template <int TM, int TN, int NST, bool A_REG, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_dma_kernel(const dc_conv_desc d)
{
    constexpr bool PIPE = DC_GEMM_PIPE;
    constexpr bool GENERIC = EPI == 0;
    const bool e_geglu = GENERIC ? d.epilogue == 1 : EPI >= 4;
    const bool e_ln = GENERIC ? d.ln_stats != nullptr : (EPI == 3 || EPI == 5);
    const bool e_res = GENERIC ? d.residual != nullptr : EPI == 2;
    const bool e_rowadd = GENERIC && d.row_add != nullptr;
    const int e_act = GENERIC ? d.act : 0;
    const bool e_stats = (GENERIC || EPI == 1 || EPI == 2) && d.stats_out != nullptr;
    const bool e_gnpart = (GENERIC || EPI == 1 || EPI == 2) && d.gn_part_out != nullptr;
    constexpr int WM = 2, WN = 2;
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr int ROWS = BM + BN;
    constexpr int STAGE = ROWS * 128;
    constexpr int NGW = (A_REG ? BN : ROWS) / 32;     // DMA wave-instructions per wave per stage (8 rows each)
    constexpr int G0 = A_REG ? BM / 8 : 0;            // first DMA piece (pieces below it are the register-staged A rows)
    constexpr int PA = BM / 32;
    static_assert(ROWS % 32 == 0 && BN % 32 == 0, "rows per stage must split over 4 waves x 8-row pieces");
    static_assert(!A_REG || NST == 2, "hybrid staging is double-buffered");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 15, fq = lane >> 4;
    DC_STAMP_AT(0);

    const int HoWo = d.Ho * d.Wo;
    const int M = d.N * HoWo;
    const int K = d.C1 + d.C2;
    const int KT = K >> 6;
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

    int kt_begin = 0, kt_end = KT;
    if (d.splitk > 1) {
        const int per = (KT + d.splitk - 1) / d.splitk;
        kt_begin = blockIdx.y * per;
        kt_end = min(KT, kt_begin + per);
        if (kt_begin >= kt_end) return;
    }
    const int nk = kt_end - kt_begin;

    // ---- per-lane DMA sources.  Piece g (8 rows) of a stage: rows [8g, 8g+8); lane s -> row 8g + (s>>3), LDS slot s&7,
    //      which must hold source chunk (s&7) ^ (row&7).
    const char* src1[NGW];                            // row base in x1 (or W) incl. the swizzled chunk offset
    const char* src2[NGW];                            // row base in x2 (second K range) — A rows only
    int ldsoff[NGW];
    const int c1_steps = d.C1 >> 6;
#pragma unroll
    for (int i = 0; i < NGW; ++i) {
        const int g = G0 + wave + 4 * i;
        const int row = g * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        ldsoff[i] = g * 1024;
        if (g * 8 < BM) {
            int m = m0 + row;
            m = m < M ? m : M - 1;                    // clamp: rows past M are computed and discarded
            src1[i] = (const char*)d.x1 + ((long long)m * d.C1 + chunk * 8) * 2;
            src2[i] = d.x2 ? (const char*)d.x2 + ((long long)m * d.C2 + chunk * 8) * 2 : nullptr;
        } else {
            int n = n0 + row - BM;
            n = n < d.Cout ? n : d.Cout - 1;
            src1[i] = (const char*)d.w + ((long long)n * K + chunk * 8) * 2;
            src2[i] = nullptr;
        }
    }
#ifndef DC_EXP_NO_DMA
#define DC_EXP_NO_DMA 0         // developer experiment (wrong results): 1 = stages are issued in the prologue only
#endif
    auto issue_stage = [&](int kt, int slot) {
        if (DC_EXP_NO_DMA && kt >= kt_begin + NST - 1) return;
        kt = kt < kt_end ? kt : kt_end - 1;           // past-the-end stages re-read the last one (keeps vmcnt counts constant)
        char* base = smem + slot * STAGE;
        if (kt < c1_steps) {                          // first K range (the only one without a channel concat): no per-piece select
#pragma unroll
            for (int i = 0; i < NGW; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(src1[i] + (long long)kt * 128), (lptr_t)(base + ldsoff[i]), 16, 0, 0);
            return;
        }
#pragma unroll
        for (int i = 0; i < NGW; ++i) {
            const bool is_a = (G0 + wave + 4 * i) * 8 < BM;                    // wave-uniform
            const char* p = is_a ? src2[i] + (long long)(kt - c1_steps) * 128 : src1[i] + (long long)kt * 128;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(base + ldsoff[i]), 16, 0, 0);
        }
    };

    // register-staged A rows (A_REG): thread -> chunk q of rows r0 + 32 i
    const int aq = tid & 7, ar0 = tid >> 3;
    const char* a1[PA];
    const char* a2[PA];
    u32x4 ra[PA];
    if (A_REG) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            int m = m0 + ar0 + 32 * i;
            m = m < M ? m : M - 1;
            a1[i] = (const char*)d.x1 + ((long long)m * d.C1 + aq * 8) * 2;
            a2[i] = d.x2 ? (const char*)d.x2 + ((long long)m * d.C2 + aq * 8) * 2 : nullptr;
        }
    }
    auto load_a = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const char* p = kt >= c1_steps ? a2[i] + (long long)(kt - c1_steps) * 128 : a1[i] + (long long)kt * 128;
            ra[i] = *(const u32x4*)p;
        }
    };
    auto store_a = [&](int slot) {
        char* sA = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int r = ar0 + 32 * i;
            *(u32x4*)(sA + r * 128 + ((aq ^ (r & 7)) << 4)) = ra[i];
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // One K-step.  All 2 x (TN + TM) fragment reads of the step are issued up front (both 32-wide k halves, two register
    // sets), so the reads of the second half land while the MFMAs of the first run; left to itself hipcc keeps one weight
    // fragment live at a time and waits lgkmcnt(0) in front of every group of TM MFMAs — the full LDS latency once per 4
    // MFMAs.  The next stage's LDS-DMA pieces are issued AFTER the reads in program order (a DMA is an LDS store to the
    // compiler: it may not sink below reads that precede it, but it may be scheduled among the MFMAs that follow it) and
    // are spread between the MFMAs, one per few MFMAs: an LDS-DMA wave-instruction holds the wave's issue for 60-100 cycles,
    // which a leading block of NGW of them would add to every K-step in front of the first MFMA.
    bf16x8 wf[2][TN], xf[2][TM];
    auto load_frags = [&](int slot) {
        const char* sA = smem + slot * STAGE;
        const char* sB = sA + BM * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int swz = ((4 * s + fq) ^ (fr & 7)) << 4;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) xf[s][tm] = *(const bf16x8*)(sA + ((wm * TM + tm) * 16 + fr) * 128 + swz);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) wf[s][tn] = *(const bf16x8*)(sB + ((wn * TN + tn) * 16 + fr) * 128 + swz);
        }
    };
    auto mfma_frags = [&](bool with_dma) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][tn], xf[s][tm], acc[tn][tm], 0, 0, 0);
        if (PIPE) {
            constexpr int NMF = 2 * TN * TM;
            constexpr int PER = NMF / (NGW + 1) > 0 ? NMF / (NGW + 1) : 1;
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TN + TM), 0);          // every fragment read first
            if (with_dma) {
#pragma unroll
                for (int i = 0; i < NGW; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);            // a few MFMAs ...
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);              // ... then one LDS-DMA piece
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NMF - NGW * PER, 0);
            }
        }
    };
    auto compute = [&](int slot) {
        load_frags(slot);
        mfma_frags(false);
    };

    // bias for this lane's output channels: fetched now so its latency hides under the K loop
    f32x4 bv[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int nb = n0 + (wn * TN + tn) * 16 + 4 * fq;
        bv[tn] = (d.bias && nb < d.Cout) ? *(const f32x4*)(d.bias + nb) : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // LayerNorm folded into this linear (dc_conv_desc.ln_stats): this lane's rows' (mean, rstd) and the weight column sums,
    // fetched here next to the bias and first used in the epilogue, so the loads fly under the whole K loop (consuming them
    // here would put a full memory round trip in front of the first DMA stage of every workgroup)
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 ln_mr[TM];
    f32x4 cs[TN];
    if (e_ln) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            int m = m0 + (wm * TM + tm) * 16 + fr;
            m = m < M ? m : M - 1;
            ln_mr[tm] = *(const f32x2*)(d.ln_stats + (long long)m * 2);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int nb = n0 + (wn * TN + tn) * 16 + 4 * fq;
            cs[tn] = nb < d.Cout ? *(const f32x4*)(d.ln_colsum + nb) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // ---- pipeline
    if (A_REG) {
        issue_stage(kt_begin, 0);
        load_a(kt_begin);
        store_a(0);
        wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int k = 0; k < nk; ++k) {
            const bool more = k + 1 < nk;
            if (more) {
                issue_stage(kt_begin + k + 1, (k + 1) & 1);      // slot (k+1)&1 was last read in step k-1: free since the barrier
                load_a(kt_begin + k + 1);
            }
            compute(k & 1);
            if (more) store_a((k + 1) & 1);
            dc_ring_sync<0>();                                  // this wave's weight pieces of stage k+1 have landed, its A rows are
                                                                // written and its reads of stage k have returned
        }
    } else {
#pragma unroll
        for (int s = 0; s < NST - 1; ++s) issue_stage(kt_begin + s, s);
        for (int k = 0; k < nk; ++k) {
            dc_ring_sync<DC_EXP_NO_DMA ? 0 : NGW * (NST - 2)>();   // this wave's pieces of stage k have landed and its reads of stage k-1 have
                                                          // returned; after the barrier: everyone else's too
            if (k == 0) DC_STAMP_AT(1);
            load_frags(k % NST);
            issue_stage(kt_begin + k + NST - 1, (k + NST - 1) % NST);
            mfma_frags(true);
        }
        wait_vmcnt<0>();
    }
    DC_STAMP_AT(2);

    const long long slab = (long long)d.N * d.Ho * d.Wo * d.Cout;   // elements per split-K slab
    // ---- epilogue.  bf16 outputs go through LDS so that global stores (and residual loads) are whole 16-byte pieces of
    //      contiguous output rows: the MFMA layout gives each lane 4 channels of one pixel, i.e. 32-byte row fragments
    //      per store instruction; staged, every row leaves as BN*2 contiguous bytes.
    const bool staged = !GENERIC || (!d.out_f32 && d.splitk <= 1);   // specialised modes are staged by construction
    if (staged) {
        constexpr int OC = BN;                                  // staged columns (GEGLU halves it below)
        constexpr int PITCH = OC * 2 + 16;                      // bytes per staged row (+16: spread rows over banks)
        static_assert(BM * PITCH <= NST * STAGE, "output tile must fit in the stage buffers");
        __syncthreads();                                        // all waves are done reading the last stage
        DC_STAMP_AT(4);
        // all epilogue operands are fetched up front (independent loads in flight together): issued one (tm, tn) tile at
        // a time behind `if (bias)` / `if (residual)` they serialise into ~20 dependent L2 round trips per workgroup
        bf16x4 rr[TM][TN];
        if (e_res && !e_geglu) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int m = m0 + (wm * TM + tm) * 16 + fr;
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int nb = n0 + (wn * TN + tn) * 16 + 4 * fq;
                    rr[tm][tn] = (m < M && nb < d.Cout) ? *(const bf16x4*)((const bf16_t*)d.residual + (long long)m * d.Cout + nb)
                                                        : bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                }
            }
        }
        f32x4 gs[TN], gq[TN];                                   // GroupNorm partials over this wave's rows (gn_part_out)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) gs[tn] = gq[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int row = (wm * TM + tm) * 16 + fr;
            const int m = m0 + row;
            const int nimg = e_rowadd ? (m < M ? m : M - 1) / HoWo : 0;
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
                float st1 = 0.f, st2 = 0.f;                     // partial (sum, sum of squares) of this output row: stats_out
                const float rowmask = m < M ? 1.f : 0.f;        // rows past M are clamped duplicates: keep them out of the sums
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int nl = (wn * TN + tn) * 16 + 4 * fq;
                    const int nb = n0 + nl;
                    f32x4 v = acc[tn][tm];
                    if (e_ln) v = dc_ln_fold(v, ln_mr[tm][0], ln_mr[tm][1], cs[tn]);
                    v += bv[tn];
                    if (e_rowadd && nb < d.Cout) v += *(const f32x4*)(d.row_add + (long long)nimg * d.row_add_stride + nb);
                    if (e_act) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = dc_act(v[r], e_act);
                    }
                    if (e_res) v = dc_scale_res(v, d.out_scale, rr[tm][tn]);
                    else v *= d.out_scale;
                    bf16x4 pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)v[r];
                    *(bf16x4*)(smem + row * PITCH + ((nl * 2) ^ dc_stage_swz(row))) = pk;
                    if (GENERIC || EPI == 1 || EPI == 2) {      // the sums are a handful of FMAs: always formed, stored on request
                        const float cm = nb < d.Cout ? 1.f : 0.f;
                        st1 += cm * ((v[0] + v[1]) + (v[2] + v[3]));
                        st2 += cm * ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
                        gs[tn] += rowmask * v;
                        gq[tn] += rowmask * (v * v);
                    }
                }
                if (e_stats) {                                  // lanes fr, fr+16, fr+32, fr+48 hold the same row: fixed-order tree
                    st1 += __shfl_xor(st1, 16, 64);
                    st2 += __shfl_xor(st2, 16, 64);
                    st1 += __shfl_xor(st1, 32, 64);
                    st2 += __shfl_xor(st2, 32, 64);
                    if (fq == 0 && m < M) {
                        *(f32x2*)(d.stats_out + ((long long)m * (2 * n_tiles) + tile_n * 2 + wn) * 2) = f32x2{st1, st2};
                    }
                }
            }
        }
        if (e_gnpart) {
            // chunk = (row tile within the sample, wave row); the launcher guarantees HoWo % BM == 0, so a tile has one sample
            const int n_img = m0 / HoWo;
            const int chunk = ((m0 - n_img * HoWo) / BM) * 2 + wm;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int nb = n0 + (wn * TN + tn) * 16 + 4 * fq;
                dc_gn_partial_store(gs[tn], gq[tn], d.gn_part_out + (((long long)chunk * d.N + n_img) * d.Cout + nb) * 2,
                                    fr == 0 && nb < d.Cout);
            }
        }
        DC_STAMP_AT(5);
        __syncthreads();
        DC_STAMP_AT(6);
        // cooperative store: consecutive lanes -> consecutive 16-byte pieces of one output row
        const int out_cols = e_geglu ? d.Cout >> 1 : d.Cout;
        const int col0 = e_geglu ? n0 >> 1 : n0;
        bf16_t* __restrict__ o = (bf16_t*)d.out;
        auto store_rows = [&](auto pieces_c) {                  // 16-byte pieces per staged row: a compile-time divisor
            constexpr int pieces = decltype(pieces_c)::value;
            for (int i = tid; i < BM * pieces; i += 256) {
                const int row = i / pieces, pc = i - row * pieces;
                const int m = m0 + row, c = col0 + pc * 8;
                if (m < M && c < out_cols) *(u32x4*)(o + (long long)m * out_cols + c) = dc_stage_unswz(*(const u32x4*)(smem + row * PITCH + pc * 16), row);
            }
        };
        if (e_geglu) store_rows(std::integral_constant<int, OC / 16>{});
        else store_rows(std::integral_constant<int, OC / 8>{});
        DC_STAMP_AT(3);
        return;
    }
    // ---- direct epilogue (fp32 outputs, split-K partials): same contract as igemm.hip
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
                const int nb = n0 + (wn * TN + 2 * tp) * 16 + 4 * fq;
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

template <int TM, int TN, int NST, bool A_REG, int EPI>
int launch_gemm_e(const dc_conv_desc& d, hipStream_t st)
{
    constexpr int BM = 2 * TM * 16, BN = 2 * TN * 16;
    const int M = d.N * d.Ho * d.Wo;
    const int nblk = dc_cdiv(M, BM) * dc_cdiv(d.Cout, BN);
    const dim3 grid(nblk, d.splitk > 1 ? d.splitk : 1);
    const size_t lds = (size_t)NST * (BM + BN) * 128;
    auto kern = gemm_dma_kernel<TM, TN, NST, A_REG, EPI>;
    static std::atomic<unsigned long long> attr_done{0};
    dc_set_max_dyn_lds((const void*)kern, (int)lds, attr_done);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, d);
    return dc_launch_status();
}

// epilogue mode of a descriptor (see the kernel's EPI comment); 0 = the generic run-time-flag epilogue
int epi_mode(const dc_conv_desc& d)
{
    if (!DC_EPI_SPECIALIZE) return 0;
    if (d.out_f32 || d.splitk > 1 || d.row_add || d.act) return 0;
    if (d.epilogue == 1) return (d.residual || d.stats_out || d.gn_part_out) ? 0 : (d.ln_stats ? 5 : 4);
    if (d.ln_stats) return (d.residual || d.stats_out || d.gn_part_out) ? 0 : 3;
    return d.residual ? 2 : 1;
}

// PROD: the four tile shapes the dispatcher uses get every specialised epilogue; developer-knob shapes only the generic one
template <int TM, int TN, int NST, bool A_REG = false, bool PROD = false>
int launch_gemm(const dc_conv_desc& d, hipStream_t st)
{
    if constexpr (PROD) {
        switch (epi_mode(d)) {
            case 1: return launch_gemm_e<TM, TN, NST, A_REG, 1>(d, st);
            case 2: return launch_gemm_e<TM, TN, NST, A_REG, 2>(d, st);
            case 3: return launch_gemm_e<TM, TN, NST, A_REG, 3>(d, st);
            case 4: return launch_gemm_e<TM, TN, NST, A_REG, 4>(d, st);
            case 5: return launch_gemm_e<TM, TN, NST, A_REG, 5>(d, st);
            default: break;
        }
    }
    return launch_gemm_e<TM, TN, NST, A_REG, 0>(d, st);
}

}  // namespace

extern "C" int dc_gemm_row_stats_parts(int Cout) { return dc_row_stats_parts_rule(Cout); }

int dc_gemm_rowpanel_wanted(const dc_conv_desc& d, int epi);
// 1x1 launches of the LDS-DMA GEMM family.  A GroupNorm affine on load exists in the row-panel kernel only: such a launch belongs here
// exactly when that kernel takes it (the gather GEMM of igemm.hip serves the others).
int dc_gemm_dma_supported(const dc_conv_desc& d)
{
    if (d.ksize != 1) return 0;
    return d.gn_ab == nullptr || dc_gemm_rowpanel_wanted(d, epi_mode(d));
}

int dc_gemm_dma_gn_chunks(const dc_conv_desc& d);
// gemm_wide.hip: the 256-row ping-pong kernel for long-K launches
int dc_gemm_wide_wanted(const dc_conv_desc& d, int epi);
int dc_gemm_wide_launch(const dc_conv_desc& d, int epi, hipStream_t st);
int dc_gemm_wide_gn_chunks(const dc_conv_desc& d);
// gemm_p8.hip: the 256 x 256 four-phase kernel for the long-K, wide-N linears without residual / statistics
int dc_gemm_p8_wanted(const dc_conv_desc& d, int epi);
int dc_gemm_p8_launch(const dc_conv_desc& d, int epi, hipStream_t st);
// gemm_rowpanel.hip: the K = 320 kernel that keeps a 256-row activation panel in registers and streams only W
int dc_gemm_rowpanel_wanted(const dc_conv_desc& d, int epi);
int dc_gemm_rowpanel_launch(const dc_conv_desc& d, int epi, hipStream_t st);
int dc_gemm_rowpanel_gn_chunks(const dc_conv_desc& d);

int dc_gemm_dma_launch(const dc_conv_desc& d, hipStream_t st)
{
    const long long M = (long long)d.N * d.Ho * d.Wo;
    // the folded LayerNorm and the row statistics live in the staged (bf16, unsplit) epilogue only
    if ((d.ln_stats || d.stats_out || d.gn_part_out) && (d.out_f32 || d.splitk > 1)) return DC_ERR_INVALID;
    if (d.gn_part_out && dc_gemm_dma_gn_chunks(d) == 0) return DC_ERR_INVALID;
    {
        const int epi = epi_mode(d);
        if (dc_gemm_rowpanel_wanted(d, epi)) return dc_gemm_rowpanel_launch(d, epi, st);
        if (d.ln_stats && d.ln_parts > 0) {
            // raw LayerNorm partials and a kernel whose waves do not own whole rows: the finalize pass runs here, into the
            // caller's scratch, and the launch proceeds on (mean, rstd) pairs — the same bits as finalizing beforehand
            if (!d.ln_scratch) return DC_ERR_INVALID;
            const int rc = dc_ln_finalize(d.ln_stats, d.ln_scratch, M, d.ln_parts, d.C1 + d.C2, d.ln_eps, (void*)st);
            if (rc != DC_OK) return rc;
            dc_conv_desc q = d;
            q.ln_stats = d.ln_scratch;
            q.ln_parts = 0;
            return dc_gemm_dma_launch(q, st);
        }
        if (d.ln_stats && !d.ln_colsum) return DC_ERR_INVALID;
        if (dc_gemm_p8_wanted(d, epi)) return dc_gemm_p8_launch(d, epi, st);
        if (dc_gemm_wide_wanted(d, epi)) return dc_gemm_wide_launch(d, epi, st);
    }
    if (d.ln_stats && !d.ln_colsum) return DC_ERR_INVALID;
    if (d.stats_out && d.epilogue != 0) return DC_ERR_INVALID;
    const bool n160 = (d.Cout % 160 == 0) && d.epilogue == 0;
    const int bn = n160 ? 160 : 128;
    const long long big = ((M + 127) / 128) * ((d.Cout + bn - 1) / bn) * (d.splitk > 1 ? d.splitk : 1);
    // 2 LDS stages for the big tiles keep two workgroups resident per CU (2 waves per SIMD: one computes while the
    // other waits for its DMA); the small tiles afford 3 stages at the same residency.
    static const int force_nst = DC_KNOB("DC_GEMM_NST", 0);   // developer knobs
    static const int force_small = DC_KNOB("DC_GEMM_SMALL", 0);
    if (force_small == 1) return n160 ? launch_gemm<2, 5, 2>(d, st) : launch_gemm<2, 4, 3>(d, st);
    if (force_small == 2) return launch_gemm<2, 2, 3>(d, st);
    if (force_small == 3) return launch_gemm<2, 4, 2>(d, st);
    static const int hybrid = DC_KNOB("DC_GEMM_HYBRID", 0);   // measured: no gain over all-DMA
    if (hybrid && big >= 256) return n160 ? launch_gemm<4, 5, 2, true>(d, st) : launch_gemm<4, 4, 2, true>(d, st);
    if (hybrid) return n160 ? launch_gemm<2, 5, 2, true>(d, st) : launch_gemm<2, 4, 2, true>(d, st);
    if (force_nst == 4 && big >= 256) return n160 ? launch_gemm<4, 5, 4>(d, st) : launch_gemm<4, 4, 4>(d, st);
    if (force_nst == 3 && big >= 256) return n160 ? launch_gemm<4, 5, 3>(d, st) : launch_gemm<4, 4, 3>(d, st);
    if (big >= 256) return n160 ? launch_gemm<4, 5, 2, false, true>(d, st) : launch_gemm<4, 4, 2, false, true>(d, st);
    // Grids that cannot even put one workgroup on every CU (the 16x16 / 8x8 levels of a one- or two-frame decode) are bound by
    // the serial K loop: one LDS-DMA round trip per 64-wide step.  They take a deeper ring — the whole CU's LDS for one
    // workgroup, three or four stages in flight instead of one — with the same tile shape (so the statistics / GroupNorm partial
    // layouts are unchanged).  M = 512, N = 1280, K = 1280: 26 -> see tools/bench_gemm.py 2.
    static const int deep = DC_KNOB("DC_GEMM_DEEP", 1);        // developer A/B knob
    const long long small = ((M + 63) / 64) * ((d.Cout + bn - 1) / bn) * (d.splitk > 1 ? d.splitk : 1);
    const int KT = (d.C1 + d.C2) >> 6;
    if (deep && small <= 256 && KT >= 6) return n160 ? launch_gemm<2, 5, 4, false, true>(d, st) : launch_gemm<2, 4, 5, false, true>(d, st);
    return n160 ? launch_gemm<2, 5, 2, false, true>(d, st) : launch_gemm<2, 4, 3, false, true>(d, st);
}

// gn_part_out chunks per sample of this launch (0: not available)
int dc_gemm_dma_gn_chunks(const dc_conv_desc& d)
{
    if (d.out_f32 || d.splitk > 1 || d.epilogue != 0 || d.ln_stats) return 0;
    {   // the launch decision must be the one dc_gemm_dma_launch takes with gn_part_out set (the host sizes the buffer from here)
        dc_conv_desc q = d;
        if (!q.gn_part_out) q.gn_part_out = (float*)(uintptr_t)16;
        if (dc_gemm_rowpanel_wanted(q, epi_mode(q))) return dc_gemm_rowpanel_gn_chunks(q);
        if (dc_gemm_wide_wanted(q, epi_mode(q))) return dc_gemm_wide_gn_chunks(q);
    }
    const long long M = (long long)d.N * d.Ho * d.Wo;
    const int bn = (d.Cout % 160 == 0) ? 160 : 128;
    const long long big = ((M + 127) / 128) * ((d.Cout + bn - 1) / bn);
    const int bm = big >= 256 ? 128 : 64;
    const long long hw = (long long)d.Ho * d.Wo;
    if (hw % bm) return 0;                                  // a row tile would straddle two samples
    return (int)(hw / bm) * 2;
}
