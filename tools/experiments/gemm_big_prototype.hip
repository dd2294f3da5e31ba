// 256 x 256 macro-tile GEMM  out[m][n] = sum_k X[m][k] * W[n][k]  for gfx950: the long-K 1x1 / linear launches of the deep levels
// (attention / feed-forward projections of the 16x16 and 32x32 transformers, K >= 1280 or a wide GEGLU; same call sites as
// gemm_dma.hip / gemm_wide.hip, flownet.py:87-118 and pipeline.py:358-367).
//
// Why another tile shape (round 4): every operand of these GEMMs enters the CU through LDS-DMA, and a `global_load_lds_dwordx4` moves
// 1 KB per wave-instruction at ~100 issue cycles — about 70 GB/s per CU whatever the kernel does (the "L2-served LDS fill" rate of
// the guide).  FLOP per filled byte is (BM * BN) / (BM + BN): 71 for the 128 x 160 tile, 98 for 256 x 160, 128 for 256 x 256.  The
// library GEMM (hipBLASLt through torch, profiles/r04_bench_library_gemm_yardstick.txt) runs these shapes at 1.05-1.24 PFLOP/s with
// that macro tile; the 256 x 160 kernel reaches 0.84-0.96.  256 x 256 accumulators are 256 registers per lane at 256 threads: ONE
// wave per SIMD with the accumulators in the AGPR half of its 512-register budget — so nothing overlaps for free, and the K step is
// software-pipelined inside each wave instead:
//   a K step of 64 is two halves of 32; the fragment reads of the NEXT half are in flight while the 64 MFMAs of the current half
//   issue; the workgroup barrier sits in the MIDDLE of the step (after it stage k+1 is visible and stage k's slot is free: the
//   LDS-DMA of stage k+2 and the first-half reads of stage k+1 are issued beside the second half's MFMAs).  Two LDS stages of
//   64 KB (512 rows x 128 B, the XOR-swizzled row image of the other GEMM kernels), one stage = 16 DMA pieces per wave.
// Same k order, same MFMA (v_mfma_f32_16x16x32_bf16, A = W fragment, B = X fragment) and the same epilogue expressions as
// gemm_dma.hip / gemm_wide.hip: bf16 outputs are bit-identical with them (tests/test_gpu_rowpanel.py).
#include "dc_common.h"
#include "../../include/diffcodec_hip.h"
#include <cstdlib>
#include <type_traits>

#ifndef DC_BIG_DMA
#define DC_BIG_DMA 1            // operand path: 1 = LDS-DMA pieces between the MFMA groups, 0 = register staging (global_load -> ds_write)
#endif
#ifndef DC_BIG_NODMA            // developer experiments (WRONG results): what the K slice costs without its DMA / fragment reads / barrier
#define DC_BIG_NODMA 0
#endif
#ifndef DC_BIG_NOREAD
#define DC_BIG_NOREAD 0
#endif
#ifndef DC_BIG_NOBAR
#define DC_BIG_NOBAR 0
#endif

namespace {

typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// EPI as in gemm_dma.hip: 1 bias, 2 bias + scale + residual, 3 folded LayerNorm + bias, 4 GEGLU, 5 folded LayerNorm + GEGLU
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_big_kernel(const dc_conv_desc d, const int m_fastest)
{
    constexpr int BM = 256, BN = 256, TM = 8, TN = 8, NST = 5;
    constexpr int STAGE = (BM + BN) * 64;             // 32 KB: 512 rows x 64 B (a K slice of 32)
    constexpr int PW = (BM + BN) / 16 / 4;            // 8 LDS-DMA pieces (16 rows x 64 B) per wave and stage
    constexpr bool e_geglu = EPI >= 4, e_ln = EPI == 3 || EPI == 5, e_res = EPI == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 15, fq = lane >> 4;

    const int M = d.N * d.Ho * d.Wo;
    const int K = d.C1;
    const int nk = K >> 5;                            // K slices of 32
    const int n_tiles = d.Cout / BN, m_tiles = M / BM;
    const int nblk = n_tiles * m_tiles;
    int bid = blockIdx.x;
    {   // XCD-aware order: each XCD walks a contiguous range of tiles (see conv3x3_tile.hip for the two orders)
        const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + idx;
    }
    const int tile_n = m_fastest ? bid / m_tiles : bid % n_tiles;
    const int tile_m = m_fastest ? bid - tile_n * m_tiles : bid / n_tiles;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- LDS image of a K slice: rows of 64 B (4 chunks of 16 B = 8 k each); the chunk index is XORed with (row >> 2) & 3, so that the
    //      16 lanes of a fragment read (rows r .. r+15 at one chunk) fall on 16 different 16-byte slots of the 256-byte bank row.
    //      Operands reach LDS through REGISTERS (global_load_dwordx4 -> ds_write_b128), not by LDS-DMA: with one wave per SIMD a
    //      `global_load_lds` piece holds the wave's only issue stream for 100-185 cycles (8 pieces = more than the slice's 1,024 MFMA
    //      cycles: the LDS-DMA prototype of this kernel ran at 0.68-0.82 PFLOP/s), a register load + LDS write pair for ~30.
    //      Piece g = wave + 4 i (i < 8) covers slice rows [16g, 16g + 16): A rows for i < 4, W rows for i >= 4; lane s loads source chunk
    //      (s & 3) ^ ((row >> 2) & 3) of row 16g + (s >> 2) and writes it at byte 16 s of the piece — the same chunk for every piece.
    const int prow = 16 * wave + (lane >> 2);
    const int chunk = (lane & 3) ^ ((prow >> 2) & 3);
    const char* const srcA = (const char*)d.x1 + ((long long)(m0 + prow) * K + chunk * 8) * 2;
    const char* const srcB = (const char*)d.w + ((long long)(n0 + prow) * K + chunk * 8) * 2;
    const long long pstep = 128LL * K;                // 64 rows further down: the next piece of this lane
    u32x4 stg[4][PW];                                 // four register sets: slices j+2 .. j+5 are in flight or waiting to be written
    auto ld_slice = [&](auto set_c, int kt) {
        constexpr int S = decltype(set_c)::value;
        kt = kt < nk ? kt : nk - 1;                   // past the end: re-read the last slice (its LDS copy is never used)
        const char* pa = srcA + (long long)kt * 64;
        const char* pb = srcB + (long long)kt * 64;
#pragma unroll
        for (int i = 0; i < PW / 2; ++i) stg[S][i] = *(const u32x4*)(pa + i * pstep);
#pragma unroll
        for (int i = 0; i < PW / 2; ++i) stg[S][PW / 2 + i] = *(const u32x4*)(pb + i * pstep);
    };
    auto st_slice = [&](auto set_c, int slot) {
        constexpr int S = decltype(set_c)::value;
        char* base = smem + slot * STAGE + wave * 1024 + lane * 16;
#pragma unroll
        for (int i = 0; i < PW / 2; ++i) *(u32x4*)(base + i * 4096) = stg[S][i];
#pragma unroll
        for (int i = 0; i < PW / 2; ++i) *(u32x4*)(base + BM * 64 + i * 4096) = stg[S][PW / 2 + i];
    };

    // The 64 accumulator tiles are PINNED to the accumulation registers (inline-asm MFMAs with "+a" operands): left to hipcc, the 256
    // accumulator registers were split between the two register files and every K slice carried 200-500 v_accvgpr copies.
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses inside a slice: lane (fr, fq) reads chunk fq ^ ((row >> 2) & 3) of row base + fr (base a multiple of 16)
    const int a_off = (wm * 128 + fr) * 64 + ((fq ^ ((fr >> 2) & 3)) << 4);
    const int b_off = BM * 64 + (wn * 128 + fr) * 64 + ((fq ^ ((fr >> 2) & 3)) << 4);
    bf16x8 xa[2][TM], wb[4];
    auto rd_xa = [&](int slot, int buf, int t) { xa[buf][t] = *(const bf16x8*)(smem + slot * STAGE + a_off + t * 1024); };
    auto rd_wb = [&](int slot, int buf, int t) { wb[buf] = *(const bf16x8*)(smem + slot * STAGE + b_off + t * 1024); };

    // ---- pipeline.  K slice s: loaded into register set s & 3 during slice s-5, written to LDS slot s % 5 during slice s-2, visible at
    //      the barrier that opens slice s-1, its fragments read (into the other register set) during slice s-1, multiplied during
    //      slice s.  The loop is unrolled by four so that every register-set index is a compile-time constant.
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
#if DC_BIG_DMA
#pragma unroll
    for (int j = 0; j < NST - 1; ++j) {
        const int kt = j < nk ? j : nk - 1;
#pragma unroll
        for (int i = 0; i < PW / 2; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(srcA + (long long)kt * 64 + i * pstep), (lptr_t)(smem + j * STAGE + wave * 1024 + i * 4096), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < PW / 2; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(srcB + (long long)kt * 64 + i * pstep), (lptr_t)(smem + j * STAGE + wave * 1024 + BM * 64 + i * 4096), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
#else
    ld_slice(I0{}, 0);
    ld_slice(I1{}, 1);
    ld_slice(I2{}, 2);
    ld_slice(I3{}, 3);
    st_slice(I0{}, 0);
    st_slice(I1{}, 1);
    ld_slice(I0{}, 4);
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < TM; ++t) rd_xa(0, 0, t);
    rd_wb(0, 0, 0);
    rd_wb(0, 1, 1);
    auto slice = [&](auto q_c, int j) {               // Q = j & 3
        constexpr int Q = decltype(q_c)::value, CUR = Q & 1;
        const int slot = j % NST;
        const int nslot = j + 1 < nk ? (j + 1) % NST : slot;
#if DC_BIG_DMA
        // LDS-DMA form: slice j+4 -> slot (j+4) % 5, one 1-KB piece per W-fragment group between the groups' MFMAs
        const int kt3 = j + NST - 1 < nk ? j + NST - 1 : nk - 1;
        char* const dbase = smem + ((j + NST - 1) % NST) * STAGE + wave * 1024;
        const char* const pa = srcA + (long long)kt3 * 64;
        const char* const pb = srcB + (long long)kt3 * 64;
#else
        // slice j+2 (in registers since slice j-3) goes to LDS, the loads of slice j+5 take the set slice j+1 left
        st_slice(std::integral_constant<int, (Q + 2) & 3>{}, (j + 2) % NST);
        ld_slice(std::integral_constant<int, (Q + 1) & 3>{}, j + 5);
#endif
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            // W fragments roll through four registers sets, two groups ahead (fragment f of every slice lives in set f & 3)
            if (!DC_BIG_NOREAD) {
                if (tn + 2 < TN) rd_wb(slot, (tn + 2) & 3, tn + 2);
                else rd_wb(nslot, (tn + 2) & 3, tn + 2 - TN);              // fragments 0 / 1 of the next slice
                rd_xa(nslot, CUR ^ 1, tn);
            }
#if DC_BIG_DMA
            if (!DC_BIG_NODMA) {
                if (tn < 4) __builtin_amdgcn_global_load_lds((gptr_t)(pa + tn * pstep), (lptr_t)(dbase + tn * 4096), 16, 0, 0);
                else __builtin_amdgcn_global_load_lds((gptr_t)(pb + (tn - 4) * pstep), (lptr_t)(dbase + BM * 64 + (tn - 4) * 4096), 16, 0, 0);
            }
#endif
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[tn][tm]) : "v"(wb[tn & 3]), "v"(xa[CUR][tm]));
        }
        // hand-over to slice j+1: own pieces / LDS writes of slice j+2 done, own reads returned
        __builtin_amdgcn_sched_barrier(0);
#if DC_BIG_DMA
        if (!DC_BIG_NODMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
#endif
        if (DC_BIG_NOBAR) return;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int j = 0; j < nk; j += 4) {                 // (K is a multiple of 128: nk is a multiple of 4)
        slice(I0{}, j);
        slice(I1{}, j + 1);
        slice(I2{}, j + 2);
        slice(I3{}, j + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA piece may be in flight when the workgroup's LDS is released

    // ---- epilogue: the tile is staged through LDS (the ring is free: every DMA has landed, every fragment read has returned, and the
    //      barrier below separates the last slice's reads from the first staged row) and written as whole 16-byte pieces of contiguous
    //      output rows; lane (fr, fq) of tile (tn, tm) holds row tm*16 + fr, columns tn*16 + 4 fq .. + 3.  Same expressions and the same
    //      staging swizzle as gemm_wide.hip.
    constexpr int OC = e_geglu ? BN / 2 : BN;
    constexpr int PITCH = OC * 2 + 16;
    static_assert(BM * PITCH <= NST * STAGE, "output tile must fit in the ring");
    const int out_cols = e_geglu ? d.Cout >> 1 : d.Cout;
    bf16_t* __restrict__ o = (bf16_t*)d.out;
    f32x4 bv[TN], cs[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int nb = n0 + wn * 128 + tn * 16 + 4 * fq;
        bv[tn] = *(const f32x4*)(d.bias + nb);
        cs[tn] = e_ln ? *(const f32x4*)(d.ln_colsum + nb) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int row = wm * 128 + tm * 16 + fr;
        const int m = m0 + row;
        f32x2 mr = {0.f, 1.f};
        if (e_ln) mr = *(const f32x2*)(d.ln_stats + (long long)m * 2);
        if (e_geglu) {
#pragma unroll
            for (int tp = 0; tp < TN / 2; ++tp) {
                f32x4 h = acc[2 * tp][tm], g = acc[2 * tp + 1][tm];
                if (e_ln) {
                    h = dc_ln_fold(h, mr[0], mr[1], cs[2 * tp]);
                    g = dc_ln_fold(g, mr[0], mr[1], cs[2 * tp + 1]);
                }
                h += bv[2 * tp];
                g += bv[2 * tp + 1];
                bf16x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)(h[r] * dc_gelu_erf(g[r]));
                *(bf16x4*)(smem + row * PITCH + (((((wn * 128 + 2 * tp * 16) >> 1) + 4 * fq) * 2) ^ dc_stage_swz(row))) = pk;
            }
        } else {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int nl = wn * 128 + tn * 16 + 4 * fq;
                f32x4 v = acc[tn][tm];
                if (e_ln) v = dc_ln_fold(v, mr[0], mr[1], cs[tn]);
                v += bv[tn];
                if (e_res) v = dc_scale_res(v, d.out_scale, *(const bf16x4*)((const bf16_t*)d.residual + (long long)m * d.Cout + n0 + nl));
                else v *= d.out_scale;
                bf16x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)v[r];
                *(bf16x4*)(smem + row * PITCH + ((nl * 2) ^ dc_stage_swz(row))) = pk;
            }
        }
    }
    __syncthreads();
    const int col0 = e_geglu ? n0 >> 1 : n0;
    constexpr int pieces = OC / 8;                    // 16-byte pieces per staged row
    for (int i = tid; i < BM * pieces; i += 256) {
        const int row = i / pieces, pc = i - row * pieces;
        *(u32x4*)(o + (long long)(m0 + row) * out_cols + col0 + pc * 8) = dc_stage_unswz(*(const u32x4*)(smem + row * PITCH + pc * 16), row);
    }
}

template <int EPI>
int launch_big(const dc_conv_desc& d, hipStream_t st)
{
    const int M = d.N * d.Ho * d.Wo;
    const int nblk = (M / 256) * (d.Cout / 256);
    const size_t lds = 5 * 512 * 64;          // the whole 160 KB of the CU
    auto kern = gemm_big_kernel<EPI>;
    static std::atomic<unsigned long long> attr_done{0};
    dc_set_max_dyn_lds((const void*)kern, (int)lds, attr_done);
    static const int force_order = DC_KNOB("DC_GEMM_BIG_ORDER", -1);   // developer A/B knob: 0 = n fastest, 1 = m fastest
    const int order = force_order >= 0 ? force_order : (d.Cout > M ? 1 : 0);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, d, order);
    return dc_launch_status();
}

}  // namespace

// DC_GEMM_BIG (developer knob): 0 = never, 1 = the rule below, 2 = whenever legal.
int dc_gemm_big_wanted(const dc_conv_desc& d, int epi)
{
    static const int mode = DC_KNOB("DC_GEMM_BIG", 0);
    if (mode == 0 || epi < 1 || epi > 5 || d.ksize != 1 || d.gn_ab || d.splitk > 1 || d.out_f32 || d.act) return 0;
    if (d.C2 != 0 || d.x2 || (d.C1 & 127) || d.C1 < 256 || !d.bias) return 0;
    if (d.stats_out || d.gn_part_out || d.row_add) return 0;            // (prototype: no statistics epilogues)
    if ((epi == 3 || epi == 5) && (!d.ln_stats || !d.ln_colsum || d.ln_parts > 0)) return 0;
    const long long M = (long long)d.N * d.Ho * d.Wo;
    if (M % 256 || d.Cout % 256) return 0;
    if (mode == 2) return 1;
    return d.C1 >= 1280 && (M / 256) * (d.Cout / 256) >= 256;
}

int dc_gemm_big_launch(const dc_conv_desc& d, int epi, hipStream_t st)
{
    switch (epi) {
        case 1: return launch_big<1>(d, st);
        case 2: return launch_big<2>(d, st);
        case 3: return launch_big<3>(d, st);
        case 4: return launch_big<4>(d, st);
        case 5: return launch_big<5>(d, st);
        default: return DC_ERR_INVALID;
    }
}
