// Row-panel GEMM  out[m][n] = sum_k X[m][k] * W[n][k]  for K = 320 on gfx950: the 64x64-resolution linears of the SD-1.5
// transformer blocks (attn to_q/k/v with the folded LayerNorm, to_out + residual, GEGLU feed-forward, proj_in / proj_out; same
// call sites as gemm_dma.hip: flownet.py:87-124, pipeline.py:358-367) at model batches whose row count fills the chip.
//
// Why a separate kernel.  With K = 320 a tiled GEMM workgroup runs five K-steps between a prologue and an epilogue in which
// nothing is in flight, and both operand tiles stream through LDS once per output tile: the 128-row kernel reached 360-590
// TFLOP/s on these shapes with the matrix pipe 22 % busy (round-2 PMC), although every one of them is within 2x of its HBM
// floor.  Here the whole K of the ACTIVATION lives in registers instead:
//   * one 512-thread workgroup per CU owns a panel of 256 rows; wave w keeps its 32 rows x 320 of X as MFMA operand fragments
//     (80 VGPRs, loaded once straight from global memory) and walks over ALL output columns;
//   * only W streams through LDS: stages of 64 output columns x 320 (40 KB, N-major, so a stage is one contiguous block of the
//     [N][K] weight matrix) in a 3-deep LDS-DMA ring shared by the 8 waves, refilled across the whole column loop — one barrier
//     per stage, the ring never drains between "tiles"; the stage's 64 bias (and LayerNorm column-sum) floats ride in the same
//     slot.  LDS-fill bytes per FLOP are 1/256 (128-row tile kernel: 1/71), LDS reads 0.5 per MFMA, none for X;
//   * the epilogue of a stage (bias / folded LayerNorm / scale + residual / GEGLU, row statistics, GroupNorm partials) runs
//     right after that stage's 80 MFMAs on 32 accumulator registers, goes through a wave-private 4 KB staging area (no
//     workgroup barrier) and leaves as whole 128-byte row pieces, while the ring keeps prefetching;
//   * residual rows go by LDS-DMA too: at the top of their stage straight into the wave's staging area (row layout, whole
//     lines), where the epilogue picks them up in the MFMA layout and overwrites them in place with the outputs; a
//     one-dword-per-half-line DMA "touch" into a scratch corner two stages earlier has pulled them into L2 by then.  No load
//     of the loop has a VGPR destination: hipcc would drain the DMA ring with a vmcnt(0) at the first use of an ordinary
//     load's result, and it may copy the destination registers of an inline-asm load while the load is still in flight
//     (seen in the first build of this kernel: v_mov of the residual registers in front of the hand-placed wait).
// Every vector-memory operation of the loop is counted by hand: per stage and wave, in issue order,
//   [touch T] [residual pieces R] [W pieces P (+1 bias, +1 colsum piece on waves 0 / 1)] ... MFMAs ... [row stores F] [GN partials G]
// so the wait in front of the barrier that opens stage s (its pieces were issued two stages earlier) is
// vmcnt(P + T + R + 2 (F + G)) and the wait in front of the first read of the residual rows is vmcnt(P).
// The accumulators see the same k order as gemm_dma.hip (ten 32-wide steps in sequence, v_mfma_f32_16x16x32_bf16) and the
// epilogue arithmetic is the same expressions, so the bf16 outputs are bit-identical to the tile kernels'.
#include "dc_common.h"
#include "../../include/diffcodec_hip.h"
#include <cstdlib>
#include <type_traits>

// Developer-only phase stamps (tools/rowpanel_stamp.py builds this file with -DDC_STAMP into a scratch .so): s_memtime of wave 0
// of every workgroup at kernel entry / after the prologue / per stage after its barrier, after its MFMAs and after its epilogue,
// written to the (otherwise unused) split-K workspace.  Never defined in the product build.
#ifdef DC_STAMP
#define RP_STAMP_AT(i)                                                                         \
    do {                                                                                       \
        if (threadIdx.x == 0 && d.splitk_ws && (i) < 64) {                                     \
            unsigned long long t_;                                                             \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            ((unsigned long long*)d.splitk_ws)[(long long)blockIdx.x * 64 + (i)] = t_;         \
        }                                                                                      \
    } while (0)
#else
#define RP_STAMP_AT(i)
#endif

// Developer experiments (scratch builds of tools/build_dev.sh only; wrong results): which resource bounds a launch?
#ifndef DC_EXP_RP_NOSTORE
#define DC_EXP_RP_NOSTORE 0     // 1: the row stores are compiled out (everything else, staging reads included, stays)
#endif
#ifndef DC_EXP_RP_NOMFMA
#define DC_EXP_RP_NOMFMA 0      // 1: one MFMA per k-step instead of eight
#endif

namespace {

typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int RP_K = 320;                          // the only K this kernel is built for
constexpr int RP_KS = RP_K / 32;                   // 32-wide MFMA k-steps
constexpr int RP_SC = 64;                          // output columns per stage
constexpr int RP_TN = RP_SC / 16, RP_TM = 2;       // 16x16 tiles per wave and stage: 4 (n) x 2 (m)
constexpr int RP_WBYTES = RP_SC * RP_K * 2;        // 40,960: the W rows of a stage
constexpr int RP_STAGE = RP_WBYTES + 512;          // + 64 bias floats + 64 colsum floats
constexpr int RP_NST = 3;
constexpr int RP_P = RP_WBYTES / 1024 / 8;         // 1 KB LDS-DMA pieces per wave and stage: 5
constexpr int RP_STG = 4096;                       // wave-private output staging: 32 rows x 128 B
constexpr int RP_TOUCH = 256;                      // wave-private landing pad of the touch DMA (never read)
constexpr int RP_LDS = RP_NST * RP_STAGE + 8 * RP_STG + 8 * RP_TOUCH;
static_assert(RP_LDS <= 160 * 1024, "LDS budget");

template <int N>
__device__ __forceinline__ void rp_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// EPI as in gemm_dma.hip: 1 bias, 2 bias + scale + residual, 3 folded LayerNorm + bias, 4 GEGLU, 5 folded LayerNorm + GEGLU.
// GN: GroupNorm partials of the output requested (modes 1-2) — compile-time because its stores enter the vmcnt arithmetic.
template <int EPI, bool GN>
__global__ __launch_bounds__(512, 2) void gemm_rowpanel_kernel(const dc_conv_desc d)
{
    constexpr bool e_geglu = EPI >= 4, e_ln = EPI == 3 || EPI == 5, e_res = EPI == 2;
    constexpr int T = e_res ? 1 : 0, R = e_res ? 4 : 0;
    constexpr int F = e_geglu ? 2 : 4, G = GN ? 2 * RP_TN : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int S = d.Cout / RP_SC;                                   // stages
    const long long mw = (long long)blockIdx.x * 256 + wave * 32;   // this wave's first row
    const bool e_stats = (EPI == 1 || EPI == 2) && d.stats_out != nullptr;
    const bool scaled = d.out_scale != 1.0f;                       // x * 1.0f is x: the multiply is skipped, not approximated
    const int extra = wave == 0 || (wave == 1 && e_ln);             // wave-uniform: one small DMA piece more per stage

    // ---- this wave's activation rows as MFMA B-operand fragments: xf[tm][k] = X[mw + 16 tm + fr][32 k + 8 fq .. + 8)
    bf16x8 xf[RP_TM][RP_KS];
#pragma unroll
    for (int tm = 0; tm < RP_TM; ++tm) {
        const char* p = (const char*)d.x1 + ((mw + tm * 16 + fr) * RP_K + fq * 8) * 2;
#pragma unroll
        for (int k = 0; k < RP_KS; ++k) xf[tm][k] = *(const bf16x8*)(p + k * 64);
    }
    if (d.gn_ab) {
        // GroupNorm affine on load (dc_conv_desc.gn_ab, no SiLU: the transformer's norm in front of proj_in): the panel lies inside one
        // sample, so every row takes the same (scale, shift) per channel; applied once to the register-resident fragments with the
        // arithmetic of the standalone pass (dc_gn_affine_pair) — the normalized tensor is never written or re-read.
        const long long hw = (long long)d.Ho * d.Wo;
        const float* __restrict__ abp = d.gn_ab + ((long long)((mw / hw) % d.gn_batch) * RP_K + fq * 8) * 2;
#pragma unroll
        for (int k = 0; k < RP_KS; ++k) {
            f32x4 g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = *(const f32x4*)(abp + k * 64 + 4 * j);
#pragma unroll
            for (int tm = 0; tm < RP_TM; ++tm) {
                u32x4 v = *(const u32x4*)&xf[tm][k];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = dc_gn_affine_pair(v[j], g[j]);
                xf[tm][k] = *(const bf16x8*)&v;
            }
        }
    }
    f32x2 ln_mr[RP_TM];
    if (e_ln) {
        if (d.ln_parts > 0) {                                       // raw partials: the finalize pass folded into this prologue — every
            dc_ln_row_partials rp[RP_TM];                           // row's pairs fetched first (loads in flight together), then summed
#pragma unroll
            for (int tm = 0; tm < RP_TM; ++tm) dc_ln_fetch_partials(rp[tm], d.ln_stats + (mw + tm * 16 + fr) * d.ln_parts * 2, d.ln_parts);
#pragma unroll
            for (int tm = 0; tm < RP_TM; ++tm) {
                float mean, rstd;
                dc_ln_mean_rstd(rp[tm], d.ln_parts, 1.0f / (float)RP_K, d.ln_eps, mean, rstd);
                ln_mr[tm] = f32x2{mean, rstd};
            }
        } else {
#pragma unroll
            for (int tm = 0; tm < RP_TM; ++tm) ln_mr[tm] = *(const f32x2*)(d.ln_stats + (mw + tm * 16 + fr) * 2);
        }
    }

    // ---- W stage image in LDS: row n (64) x 640 B, 16-byte chunk c of row n stored at chunk position c ^ (n & 7) (the XOR
    //      stays inside an aligned group of 8 chunks).  Piece g = 5 wave + i covers LDS bytes [1024 g, 1024 g + 1024): lane s
    //      lands at flat chunk 64 g + s = (row n, position q) and must fetch source chunk q ^ (n & 7) of that row.
    int woff[RP_P];
#pragma unroll
    for (int i = 0; i < RP_P; ++i) {
        const int flat = (wave * RP_P + i) * 64 + lane;
        const int n = flat / 40, q = flat - n * 40;
        woff[i] = n * 640 + ((q ^ (n & 7)) << 4);
    }
    auto issue_w = [&](int s, int slot) {
        s = s < S ? s : S - 1;                                      // past-the-end stages re-read the last one into a dead slot
        const char* src = (const char*)d.w + (long long)s * RP_WBYTES;
        char* base = smem + slot * RP_STAGE;
#pragma unroll
        for (int i = 0; i < RP_P; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + woff[i]), (lptr_t)(base + (wave * RP_P + i) * 1024), 16, 0, 0);
        if (extra) {
            const float* v = (wave == 0 ? d.bias : d.ln_colsum) + s * RP_SC + lane;
            __builtin_amdgcn_global_load_lds((gptr_t)v, (lptr_t)(base + RP_WBYTES + wave * 256), 4, 0, 0);
        }
    };

    // fragment read addresses: W row 16 tn + fr, chunk 4 k + fq -> position ((k >> 1) << 3) | (((4 (k & 1) + fq)) ^ (fr & 7))
    const int x7 = fr & 7;
    const int rd_even = fr * 640 + ((fq ^ x7) << 4);
    const int rd_odd = fr * 640 + (((4 + fq) ^ x7) << 4);

    // wave-private staging (non-GEGLU: 32 rows x 128 B; GEGLU: 32 rows x 64 B), bank-conflict-free for the 8-byte MFMA-layout
    // writes and the 16-byte row reads: chunk XOR row bits, 8-byte halves swapped on alternate 8-row groups (dc_stage_swz)
    char* stg = smem + RP_NST * RP_STAGE + wave * RP_STG;
    const bf16_t* __restrict__ resid = (const bf16_t*)d.residual;
    bf16_t* __restrict__ o = (bf16_t*)d.out;
    const int out_cols = e_geglu ? d.Cout >> 1 : d.Cout;

    float st1[RP_TM], st2[RP_TM];
#pragma unroll
    for (int tm = 0; tm < RP_TM; ++tm) st1[tm] = st2[tm] = 0.f;
    // residual rows by LDS-DMA: piece i = rows [8 i, 8 i + 8) of the wave's panel, lane l -> row 8 i + (l >> 3), staging chunk
    // position l & 7, which holds source chunk (l & 7) ^ (row & 7) (the staging swizzle; no half swap on the way in)
    const char* res_lane = nullptr;
    const char* touch_lane = nullptr;
    if (e_res) {
        res_lane = (const char*)resid + ((mw + (lane >> 3)) * d.Cout) * 2 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);
        touch_lane = (const char*)resid + ((mw + (lane & 31)) * d.Cout) * 2 + (lane >> 5) * 64;
    }

    // ---- prologue: two stages in flight, everything landed before the loop (so the steady-state counts hold from stage 1 on)
    RP_STAMP_AT(0);
    issue_w(0, 0);
    issue_w(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    RP_STAMP_AT(1);
    for (int s = 0; s < S; ++s) {
        if (s > 0) {
            __builtin_amdgcn_sched_barrier(0);
            if (extra) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(RP_P + 1 + T + R + 2 * (F + G)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(RP_P + T + R + 2 * (F + G)) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        RP_STAMP_AT(2 + 3 * s);
        // ---- top of the stage: touch (s+2), residual (s), W (s+2)
        if (e_res) {
            const int s2 = s + 2 < S ? s + 2 : S - 1;
            __builtin_amdgcn_global_load_lds((gptr_t)(touch_lane + s2 * (RP_SC * 2)),
                                             (lptr_t)(smem + RP_NST * RP_STAGE + 8 * RP_STG + wave * RP_TOUCH), 4, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(res_lane + (long long)i * 8 * d.Cout * 2 + s * (RP_SC * 2)),
                                                 (lptr_t)(stg + i * 1024), 16, 0, 0);
        }
        issue_w(s + 2, (s + 2) % RP_NST);

        // ---- 80 MFMAs: D[n][m] tiles, W fragments from the slot, X fragments from registers
        const char* sb = smem + (s % RP_NST) * RP_STAGE;
        f32x4 acc[RP_TN][RP_TM];
#pragma unroll
        for (int tn = 0; tn < RP_TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < RP_TM; ++tm) acc[tn][tm] = f32x4{0.f, 0.f, 0.f, 0.f};
        // fragment reads run one k-step ahead of the MFMAs that consume them (two register sets)
        bf16x8 wf[2][RP_TN];
        auto read_w = [&](int k, bf16x8* dst) {
#pragma unroll
            for (int tn = 0; tn < RP_TN; ++tn)
                dst[tn] = *(const bf16x8*)(sb + ((k & 1) ? rd_odd : rd_even) + tn * 10240 + (k >> 1) * 128);
        };
        read_w(0, wf[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, RP_TN, 0);      // (the scheduler fills groups in order: name this one too)
#pragma unroll
        for (int k = 0; k < RP_KS; ++k) {
            if (k + 1 < RP_KS) read_w(k + 1, wf[(k + 1) & 1]);
#pragma unroll
            for (int tn = 0; tn < RP_TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < RP_TM; ++tm)
                    if (!DC_EXP_RP_NOMFMA || (tn == 0 && tm == 0))
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[k & 1][tn], xf[tm][k], acc[tn][tm], 0, 0, 0);
                    else
                        asm volatile("" ::"v"(wf[k & 1][tn]), "v"(xf[tm][k]));      // operands stay fetched
            if (k + 1 < RP_KS) __builtin_amdgcn_sched_group_barrier(0x100, RP_TN, 0);       // the next step's reads first ...
            __builtin_amdgcn_sched_group_barrier(0x008, RP_TN * RP_TM, 0);                  // ... then this step's MFMAs
        }
        RP_STAMP_AT(3 + 3 * s);

        // ---- epilogue of the stage
        f32x4 bv[RP_TN], cs[RP_TN];
#pragma unroll
        for (int tn = 0; tn < RP_TN; ++tn) {
            bv[tn] = *(const f32x4*)(sb + RP_WBYTES + (tn * 16 + 4 * fq) * 4);
            if (e_ln) cs[tn] = *(const f32x4*)(sb + RP_WBYTES + 256 + (tn * 16 + 4 * fq) * 4);
        }
        if (e_res) {                                                // this wave's residual pieces of the stage have landed
            __builtin_amdgcn_sched_barrier(0);
            if (extra) rp_wait_vm<RP_P + 1>();
            else rp_wait_vm<RP_P>();
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 gs[RP_TN], gq[RP_TN];
#pragma unroll
        for (int tn = 0; tn < RP_TN; ++tn) gs[tn] = gq[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tm = 0; tm < RP_TM; ++tm) {
            const int row = tm * 16 + fr;
            if (e_geglu) {
#pragma unroll
                for (int tp = 0; tp < RP_TN / 2; ++tp) {
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
                    const int chunk = (tp * 2 + (fq >> 1)) ^ ((row >> 1) & 3);
                    const int half = (fq & 1) ^ ((row >> 3) & 1);
                    *(bf16x4*)(stg + row * 64 + chunk * 16 + half * 8) = pk;
                }
            } else {
#pragma unroll
                for (int tn = 0; tn < RP_TN; ++tn) {
                    f32x4 v = acc[tn][tm];
                    if (e_ln) v = dc_ln_fold(v, ln_mr[tm][0], ln_mr[tm][1], cs[tn]);
                    v += bv[tn];
                    const int chunk = (tn * 2 + (fq >> 1)) ^ x7;
                    if (e_res) v = dc_scale_res(v, d.out_scale, *(const bf16x4*)(stg + row * 128 + chunk * 16 + (fq & 1) * 8));
                    else if (scaled) v *= d.out_scale;
                    bf16x4 pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)v[r];
                    const int half = (fq & 1) ^ ((row >> 3) & 1);
                    *(bf16x4*)(stg + row * 128 + chunk * 16 + half * 8) = pk;
                    if (EPI == 1 || EPI == 2) {
                        if (e_stats) {
                            st1[tm] += (v[0] + v[1]) + (v[2] + v[3]);
                            st2[tm] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                        }
                        if (GN) {
                            gs[tn] += v;
                            gq[tn] += v * v;
                        }
                    }
                }
            }
        }
        // rows out: consecutive lanes -> consecutive 16-byte pieces of one output row (whole 128-byte lines; 64 B for GEGLU)
        if (e_geglu) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (lane >> 2) + 16 * i, c = lane & 3;
                u32x4 v = *(const u32x4*)(stg + row * 64 + ((c ^ ((row >> 1) & 3)) << 4));
                if ((row >> 3) & 1) v = u32x4{v[2], v[3], v[0], v[1]};
                if (!DC_EXP_RP_NOSTORE || v[0] == 0x12345678u) *(u32x4*)(o + (mw + row) * out_cols + s * (RP_SC / 2) + c * 8) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (lane >> 3) + 8 * i, c = lane & 7;
                u32x4 v = *(const u32x4*)(stg + row * 128 + ((c ^ (row & 7)) << 4));
                if (i & 1) v = u32x4{v[2], v[3], v[0], v[1]};
                if (!DC_EXP_RP_NOSTORE || v[0] == 0x12345678u) *(u32x4*)(o + (mw + row) * out_cols + s * RP_SC + c * 8) = v;
            }
        }
        RP_STAMP_AT(4 + 3 * s);
        if (GN) {
            // chunk = 32 rows (one wave); the launcher guarantees HoWo % 256 == 0, so a panel lies inside one sample
            const long long hw = (long long)d.Ho * d.Wo;
            const int n_img = (int)(mw / hw);
            const int chunk = (int)((mw - n_img * hw) >> 5);
#pragma unroll
            for (int tn = 0; tn < RP_TN; ++tn) {
                const int nb = s * RP_SC + tn * 16 + 4 * fq;
                dc_gn_partial_store(gs[tn], gq[tn], d.gn_part_out + (((long long)chunk * d.N + n_img) * d.Cout + nb) * 2, fr == 0);
            }
        }
    }
    if (e_stats) {
        // a wave owns whole rows: part 0 carries the row's (sum, sum of squares), the other parts the launcher promised are zero
        const int parts = dc_row_stats_parts_rule(d.Cout);
#pragma unroll
        for (int tm = 0; tm < RP_TM; ++tm) {
            float a = st1[tm], b = st2[tm];
            a += __shfl_xor(a, 16, 64);
            b += __shfl_xor(b, 16, 64);
            a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 32, 64);
            float* dst = d.stats_out + (mw + tm * 16 + fr) * parts * 2;
            for (int p = fq; p < parts; p += 4) *(f32x2*)(dst + p * 2) = p == 0 ? f32x2{a, b} : f32x2{0.f, 0.f};
        }
    }
}

template <int EPI, bool GN>
int launch_rowpanel(const dc_conv_desc& d, hipStream_t st)
{
    const long long M = (long long)d.N * d.Ho * d.Wo;
    auto kern = gemm_rowpanel_kernel<EPI, GN>;
    static std::atomic<unsigned long long> attr_done{0};
    dc_set_max_dyn_lds((const void*)kern, RP_LDS, attr_done);
    hipLaunchKernelGGL(kern, dim3((unsigned)(M / 256)), dim3(512), RP_LDS, st, d);
    return dc_launch_status();
}

}  // namespace

// GroupNorm-partials chunking of the row-panel kernel: one chunk per 32-row wave panel.
int dc_gemm_rowpanel_gn_chunks(const dc_conv_desc& d)
{
    const long long hw = (long long)d.Ho * d.Wo;
    return hw % 256 ? 0 : (int)(hw / 32);
}

// Takes a launch when K = 320 (one source; load-side transform: none, or a GroupNorm affine without SiLU), a specialised epilogue mode applies, the columns split
// into whole 64-wide stages and the row panels fill the chip (>= one workgroup per CU).  Never depends on whether the optional
// statistics outputs are set, so that the chunk query and the launch agree.
// DC_GEMM_ROWPANEL (developer builds): 0 = never.
int dc_gemm_rowpanel_wanted(const dc_conv_desc& d, int epi)
{
    static const int mode = DC_KNOB("DC_GEMM_ROWPANEL", 1);
    if (mode == 0 || epi < 1 || epi > 5 || d.ksize != 1 || d.splitk > 1 || d.out_f32) return 0;
    if (d.gn_ab && (d.gn_silu || d.gn_batch <= 0)) return 0;           // the affine on load only (the X prologue has no SiLU)
    if (d.ln_parts > DC_LN_PARTS_MAX) return 0;                        // (the dispatcher finalizes first and comes back with pairs)
    if (d.C1 != RP_K || d.C2 != 0 || d.x2 || !d.bias) return 0;
    const long long M = (long long)d.N * d.Ho * d.Wo;
    if (M % 256 || M < 256 * 256) return 0;
    if (d.Cout % RP_SC || d.Cout < 5 * RP_SC) return 0;
    if (((long long)d.Ho * d.Wo) % 256) return 0;           // GroupNorm partials need a panel inside one sample
    return 1;
}

int dc_gemm_rowpanel_launch(const dc_conv_desc& d, int epi, hipStream_t st)
{
    const bool gn = d.gn_part_out != nullptr;
    if (gn && epi > 2) return DC_ERR_INVALID;
    if ((epi == 3 || epi == 5) && !d.ln_colsum) return DC_ERR_INVALID;
    switch (epi) {
        case 1: return gn ? launch_rowpanel<1, true>(d, st) : launch_rowpanel<1, false>(d, st);
        case 2: return gn ? launch_rowpanel<2, true>(d, st) : launch_rowpanel<2, false>(d, st);
        case 3: return launch_rowpanel<3, false>(d, st);
        case 4: return launch_rowpanel<4, false>(d, st);
        case 5: return launch_rowpanel<5, false>(d, st);
        default: return DC_ERR_INVALID;
    }
}
