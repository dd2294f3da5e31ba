// Flash-style attention for gfx950:  O = softmax(Q K^T * scale) V, bf16 in/out, fp32 softmax + accumulation.
// Replaces diffusers' Attention processor (F.scaled_dot_product_attention / xformers, pipeline.py:138-142)
// inside BasicTransformerBlock.attn1 / attn2 as called from flownet.py:87-118 and pipeline.py:358-367.
//
// Formulation (per wave: 32 queries; per workgroup: 4 waves = 128 queries sharing the K/V tiles in LDS):
//   S^T[key][q] = K . Q^T      v_mfma_f32_32x32x16_bf16, A = K rows from LDS, B = Q^T held in registers
//   -> every lane owns ONE query column (q = lane & 31) and 16 of the 32 keys of the tile in its 16 accumulator
//      registers: the row max / row sum are lane-local plus one exchange with lane^32, and the probabilities,
//      converted pairwise to bf16, ARE the B operand of the next MFMA (no LDS round trip):
//   O^T[d][q]  += V^T . P^T     A = V^T tile (LDS, keys permuted to the accumulator's k order), B = P^T registers
//   The online-softmax rescale factor is per query = per lane, so it is one scalar multiply on the O accumulators.
// Head dims that are not multiples of 16/32 (d = 40) are zero-padded in LDS/registers.
#include "dc_common.h"
#include "../../include/diffcodec_hip.h"
#include <cstdlib>

// Developer-only phase stamps (tools/attn_stamp.py builds this file with -DDC_STAMP into a scratch .so): per-wave s_memtime sums of
// the QK^T + row-max phase, the exp + PV phase and the staging + barrier phase of the long-context loop.
#ifdef DC_STAMP
__device__ unsigned long long dc_attn_stamp_buf[1 << 18];
extern "C" int dc_attn_stamp_read(void* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(dc_attn_stamp_buf), (size_t)n * 8); }
#define DC_NOW(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define DC_NOW_RT(t) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")   // constant 100 MHz
#define DC_STAMP_SLOTS 16
#else
#define DC_NOW(t) (void)0
#define DC_NOW_RT(t) (void)0
#endif

namespace {

constexpr int KV_TILE = 64;                  // keys per iteration (two 32-key MFMA tiles)
#ifndef DC_ATTN_PRESCALE
#define DC_ATTN_PRESCALE 1                   // developer A/B switch for the accumulator-initialised softmax (see process_tile)
#endif
// Offset-in-the-GEMM online softmax (head dims with a spare zero-padded column: d = 40, 8): the queries are pre-multiplied by
// scale*log2(e) once per block (fp32 multiply, one bf16 rounding), column D of every K row is 1.0 in LDS and element D of the
// query fragment holds minus the running offset — so the QK^T MFMA chain itself leaves t = log2e*scale*s - offset, ready for
// v_exp_f32: no per-score FMA and no accumulator initialisation.  The offset (kept bf16-exact, so the product is exact)
// follows the running maximum lazily: it moves only when a tile's maximum exceeds it by more than RESCALE_THR
// (probabilities then reach 2^THR instead of 1, harmless in fp32/bf16 range) or on a block's first tile — the
// subtract-and-rescale pass is the exception, not the rule.  The d = 40 kernel is VALU-issue-bound (64 exps + 32 converts +
// 16 max3 per 28 MFMAs and tile): the 32 packed FMAs this removes were ~13 % of its VALU time.
constexpr float RESCALE_THR = 4.0f;
#ifndef DC_ATTN_SOFTMAX_PRIO
#define DC_ATTN_SOFTMAX_PRIO 2               // wave priority during the exp / convert block (0 = off: developer A/B)
#endif
#ifndef DC_ATTN_STORE_FENCE
#define DC_ATTN_STORE_FENCE 2                // developer A/B switch: scheduling fences in front of the ping-pong form's staging stores (0 | 1 = V | 2 = V and K)
#endif
#ifndef DC_ATTN_QK_FIRST
#define DC_ATTN_QK_FIRST 1                   // ping-pong form: QK^T(t) before PV(t-1) inside the MFMA block (0: the round-3 order, developer A/B)
#endif
#ifndef DC_ATTN_V_EARLY
#define DC_ATTN_V_EARLY 0                    // developer A/B switch: 0 = group 1 of the ping-pong form stages V at the end of its MFMA block
#endif
#ifndef DC_ATTN_EARLY_STAGE
#define DC_ATTN_EARLY_STAGE 1                // developer A/B switch: 0 = the next tile is written to LDS after the PV MFMAs
#endif
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
// V stays row-major [key][d] in LDS (one ds_write_b128 per staged vector); the PV MFMA's A operand (V^T: 4 consecutive
// keys of one d per lane) is fetched with the hardware transposing read ds_read_b64_tr_b16.  Row pitch = 64/192/320 B
// (== 64 or 192 mod 256) puts the four key rows of a 4x32 block on disjoint bank ranges: conflict-free.
__host__ __device__ constexpr int v_pitch_bytes(int ndt) { return ndt * 64 <= 64 ? 64 : (ndt * 64 <= 192 ? 192 : 320); }

struct AttnArgs {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    bf16_t* o;
    int B, heads, Nq, Nk;
    long long qs, ks, vs, os;
    float scale_log2e;
    int xcd_remap;
};

// launch bound of 2 waves/SIMD (<= 256 registers) for the small heads: hipcc then emits the VGPR form of the MFMA;
// at 1 wave/SIMD it parks accumulators in AGPRs and pays ~180 v_accvgpr moves per K/V tile in the softmax.
// QB = 32-query blocks per wave: with QB = 2 every K fragment and every transposed V read feeds two MFMAs, and the
// staging / barrier cost per key tile is shared by 256 queries per workgroup instead of 128.
// SHORT (Nk <= 2 * KV_TILE, i.e. the 77-token text context of every cross-attention): both key tiles are staged once
// and stay in the two LDS buffers while the workgroup walks over SHORT_PASSES query blocks — no barrier, no K/V traffic
// and no staging latency per block; the long form pays all three once per 128*QB queries for two iterations of work.
// RAGGED: Nk is not a multiple of KV_TILE — only then does the key-masking code exist at all (left in the common kernel,
// hipcc speculates its ~90 index/compare/select VALU instructions into every tile although they matter in the last one
// only; the d = 40 kernel is VALU-issue-bound, so that was a quarter of its time).
constexpr int SHORT_PASSES = 4;
// PP (ping-pong, long context only): ONE 8-wave workgroup per CU, two waves per SIMD FROM THE SAME workgroup, run half a tile
// apart by construction.  Stamps (tools/attn_stamp.py) showed why the 4-wave form stops at ~50 % MFMA occupancy: its two
// co-resident waves (two workgroups running identical code from the same start) stay IN PHASE, so their MFMA blocks contend for
// the matrix pipe and then their exp/convert blocks contend for the VALU — per tile and SIMD the time was MFMA + VALU, not
// max(MFMA, VALU).  Here a tile is two barrier-separated slots: waves 0-3 run {PV(t-1), QK^T(t)} while waves 4-7 run
// softmax(t-1), then the roles swap.  Waves 0-3 stage K, waves 4-7 stage V; every wave executes the same number of barriers.
template <int D, int QB, bool SHORT, bool RAGGED, bool PP = false>
__global__ __launch_bounds__(PP ? 512 : 256, (D <= 80 ? 2 : 1)) void attn_kernel(const AttnArgs a)
{
    static_assert(!PP || !SHORT, "ping-pong is the long-context form");
    constexpr int NWAVE = PP ? 8 : 4, NT = NWAVE * 64;
    constexpr int ND16 = (D + 15) / 16;              // K-steps of QK^T
    constexpr int NDT = (D + 31) / 32;               // 32-row output tiles of O^T
    constexpr int NV = D / 8;                        // 16-byte vectors per K/V row
    constexpr int KP16 = (ND16 * 2) | 1;             // K row pitch in 16-B units, odd -> conflict-free ds_read_b128
    constexpr int K_PITCH = KP16 * 16;
    constexpr int K_BYTES = KV_TILE * K_PITCH;
    constexpr int V_PITCH = v_pitch_bytes(NDT);
    constexpr int VT_BYTES = KV_TILE * V_PITCH;
    constexpr int BUF = K_BYTES + VT_BYTES;
    constexpr int QW = 32 * QB;                      // queries per wave
    constexpr int OPITCH = D * 2 + 16;               // staged output row pitch (SHORT): 16-byte aligned rows, off a power of two
    static_assert(NDT <= 5, "head dim <= 160");
    // Row sums for free: when the head dim leaves a spare zero-padded column (d = 40, 80, 8, 16), V's column D is set to
    // 1.0 in LDS, so O^T row D accumulates sum_k p — the softmax denominator — inside the PV MFMA, already rescaled by
    // the running-max correction.  Saves one v_add_f32 per probability.
    // spare QK^T column available for the offset; not for the short-context form (two tiles per block: the first-tile anchor
    // pass costs more than the two tiles of FMAs it saves — measured 74 vs 68 us at 4096 x 77, d = 40)
    constexpr bool QOFF = DC_ATTN_PRESCALE && (D % 16) != 0 && !SHORT;
    constexpr int QOFF_KS = D / 16, QOFF_LH = (D % 16) / 8, QOFF_E = (D % 16) % 8;
    constexpr bool ONES = (D % 32) != 0;
    constexpr int ONES_T = D / 32, ONES_R = ((D % 32) & 3) + 4 * ((D % 32) >> 3);
    static_assert(!ONES || ((D % 32) % 8) < 4, "ones row must live in the lower lane half");
    constexpr int NLD = (KV_TILE * NV + 255) / 256;  // vectors per thread per operand per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, lh = lane >> 5;
    unsigned long long st_in = 0, rt_in = 0, rt_0 = 0, rt_e = 0, st_out = 0, rt_out = 0;   // DC_STAMP: wave entry / loop / exit (cycles, 100 MHz ticks)
    DC_NOW(st_in);
    DC_NOW_RT(rt_in);
    constexpr int QWG = NWAVE * QW * (SHORT ? SHORT_PASSES : 1);   // queries per workgroup
    const int qblocks = (a.Nq + QWG - 1) / QWG;
    // XCD-aware order: the dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs, so with the query block as the
    // fastest index the 8 blocks of one (sample, head) at Nq = 4096 landed on 8 different XCDs and every L2 fetched every head's
    // K/V (PMC, round 3: 4.5x the algorithmic bytes = 8 x K/V + Q + O).  Each XCD now walks a contiguous range of (head, block)
    // pairs: the blocks of one head run on one XCD, back to back, and share its L2 copy of K/V.
    int bid = blockIdx.x;
    if (a.xcd_remap) {
        const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + idx;
    }
    const int bh = bid / qblocks;
    const int qb = bid - bh * qblocks;
    const int b = bh / a.heads, h = bh - b * a.heads;
    int q0 = qb * QWG + wave * QW;

    const bf16_t* __restrict__ Q = a.q + (long long)b * a.Nq * a.qs + h * D;
    const bf16_t* __restrict__ K = a.k + (long long)b * a.Nk * a.ks + h * D;
    const bf16_t* __restrict__ V = a.v + (long long)b * a.Nk * a.vs + h * D;

    // zero both LDS buffers once: pad columns (d >= D) and never-written bytes must be finite zeros
    for (int i = tid * 16; i < 2 * BUF; i += NT * 16) *(u32x4*)(smem + i) = u32x4{0u, 0u, 0u, 0u};
    if (ONES) {
        __syncthreads();
        if (tid < 2 * KV_TILE) {
            *(unsigned short*)(smem + (tid >> 6) * BUF + K_BYTES + (tid & 63) * V_PITCH + D * 2) = 0x3F80;   // bf16 1.0
            if (QOFF) *(unsigned short*)(smem + (tid >> 6) * BUF + (tid & 63) * K_PITCH + D * 2) = 0x3F80;   // K column D = 1
        }
    }

    // Q^T fragments: lane holds Q[q0 + 32*u + lq][16*ks + 8*lh .. +7]
    bf16x8 qf[QB][ND16];
    f32x16 oacc[QB][NDT];
    float m_run[QB], l_run[QB];
    auto fetch_q = [&](int qbase, bf16x8 (&dst)[QB][ND16]) {   // Q^T fragments of the block at qbase
#pragma unroll
        for (int u = 0; u < QB; ++u)
#pragma unroll
            for (int ks = 0; ks < ND16; ++ks) {
                const int dcol = 16 * ks + 8 * lh;
                const int qi = qbase + 32 * u + lq;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (qi < a.Nq && dcol < D) v = *(const u32x4*)(Q + (long long)qi * a.qs + dcol);
                bf16x8 qv = *(bf16x8*)&v;
                if (QOFF) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) qv[e] = (bf16_t)((float)qv[e] * a.scale_log2e);
                }
                dst[u][ks] = qv;
            }
    };
    auto reset_acc = [&]() {
#pragma unroll
        for (int u = 0; u < QB; ++u) {
#pragma unroll
            for (int t = 0; t < NDT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[u][t][r] = 0.f;
            m_run[u] = QOFF ? 0.f : -INFINITY;                 // QOFF: the offset (log2 domain, bf16-exact), set by the first tile
            l_run[u] = 0.f;
        }
    };
    fetch_q(q0, qf);
    reset_acc();

    u32x4 rk[NLD], rv[NLD];
    const int ntiles = (a.Nk + KV_TILE - 1) / KV_TILE;

    // Staging: lane -> key (consecutive lanes, consecutive keys), wave + 4i -> 16-byte vector of the row, so "this wave has a
    // vector to move" is a scalar (wave-uniform) test and the loads run without an exec mask; keys past Nk (ragged last tile
    // only) re-read the last key: their scores are masked to -inf below and their probabilities are exactly 0.
    // V goes to LDS in 4-row x 4-vector blocks per 16 lanes: with a row pitch of 4 or 12 (mod 16) 16-byte slots those 16 pieces fall on
    // 16 different slots of the 256-byte bank row (lane -> row, one vector per instruction put rows r, r+4, r+8, r+12 on the SAME slot:
    // a 4-way conflict on every V write, 18 % of the kernel's LDS cycles in the round-2/3 PMC passes), and four consecutive lanes read
    // 64 contiguous bytes of a global row instead of one 16-byte piece each.  Staging instruction q < 4 (NV / 4): rows 16 (q & 3) +
    // (lane >> 2), vectors 4 (q >> 2) + (lane & 3); the NV % 4 left-over vectors keep the lane -> row form.  (K: its odd pitch is
    // conflict-free in the lane -> row form.)
    constexpr int NVB = 4 * (NV / 4);
    auto v_piece = [&](int q, int& row, int& vec) {           // q is wave-uniform
        row = lane, vec = q;
        if (NVB > 0 && q < NVB) row = 16 * (q & 3) + (lane >> 2), vec = 4 * (q >> 2) + (lane & 3);
    };
    auto issue_loads = [&](int t) {
        const int kb = t * KV_TILE;
        int key = kb + lane;
        if (RAGGED) key = key < a.Nk ? key : a.Nk - 1;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int vec = wave + 4 * i;
            if (vec < NV) {
                rk[i] = *(const u32x4*)(K + (long long)key * a.ks + vec * 8);
                int vrow, vvec;
                v_piece(vec, vrow, vvec);
                int vkey = kb + vrow;
                if (RAGGED) vkey = vkey < a.Nk ? vkey : a.Nk - 1;
                rv[i] = *(const u32x4*)(V + (long long)vkey * a.vs + vvec * 8);
            }
        }
    };
    auto store_lds = [&](int buf) {
        char* sK = smem + buf * BUF;
        char* sV = sK + K_BYTES;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int vec = wave + 4 * i;
            if (vec < NV) {
                *(u32x4*)(sK + lane * K_PITCH + vec * 16) = rk[i];
                int vrow, vvec;
                v_piece(vec, vrow, vvec);
                *(u32x4*)(sV + vrow * V_PITCH + vvec * 16) = rv[i];
            }
        }
    };

    __syncthreads();                  // zero-fill (and the ones column) complete before the first tile lands
    if constexpr (!PP) {
        issue_loads(0);
        store_lds(0);
    }
    // every prologue load (the Q fragments too) has landed before the loop: otherwise hipcc's wait for Q sits INSIDE the loop
    // as vmcnt(0) in front of the first QK^T MFMA and drains the next tile's prefetch on every iteration
    if constexpr (!PP) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0) only
        __syncthreads();
    }

    typedef __attribute__((ext_vector_type(2))) float f32x2;
    unsigned long long st_a = 0, st_b = 0, st_c = 0, st_d = 0, st_e = 0, s_qk = 0, s_sm = 0, s_pv = 0, s_st = 0, st_0 = 0, st_v = 0, s_vst = 0;
    DC_NOW(st_0);
    DC_NOW_RT(rt_0);
    f32x16 s[QB][2];                   // S^T of the current key tile (QK^T -> softmax)
    bf16x8 pf[QB][2][2];               // its probabilities as the PV MFMA's B operand (softmax -> PV)
    // ---- S^T tiles (2 x 32 keys) for every query block: each K fragment feeds QB MFMAs.  K(t) lives in LDS buffer `buf`.
    // 32-key halves of the current key tile that hold at least one real key: 2, or 1 in the short-context form's last tile when the
    // context ends inside its first half (the 77-token text context: keys 64..76 of tile 1 -> its second half is all padding; skipping
    // it removes a quarter of every cross-attention's MFMA and softmax work).  Wave-uniform; a constant 2 in the long-context forms.
    int jn = 2;
    auto qk_part = [&](int buf) {
        const char* sK = smem + buf * BUF;
#pragma unroll
        for (int u = 0; u < QB; ++u)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[u][j][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < ND16; ++ks) {
                if (j >= jn) continue;
                const bf16x8 kf = *(const bf16x8*)(sK + (32 * j + lq) * K_PITCH + (16 * ks + 8 * lh) * 2);
#pragma unroll
                for (int u = 0; u < QB; ++u) s[u][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[u][ks], s[u][j], 0, 0, 0);
            }

    };
    // ---- online softmax.  The running max is kept on the RAW scores (scale > 0), the scale and the max subtraction
    //      are one (packed) FMA feeding v_exp_f32 directly; key masking only exists in the ragged last tile.
    float mloc_s[QB];                  // this tile's row maxima (rowmax -> softmax)
    // key masking of the ragged last tile and the tile's row maxima: its own part so that the ping-pong form can run it behind the
    // QK^T MFMAs (it balances the two slots: 1.28k cycles of MFMA block against 1.53k of softmax block before the move)
    auto rowmax_part = [&](int t) {
        const int kb = t * KV_TILE;
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            if (RAGGED && kb + KV_TILE > a.Nk) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (j >= jn) continue;
                        const int key = kb + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (key >= a.Nk) s[u][j][r] = -INFINITY;
                    }
            }
            float mloc = s[u][0][0];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (j < jn) mloc = fmaxf(mloc, s[u][j][r]);
            {   // exchange with lane ^ 32 on the VALU (v_permlane32_swap) instead of an LDS round trip (ds_bpermute)
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
                mloc = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            mloc_s[u] = mloc;
        }
    };
    auto softmax_part = [&](int t, bool first) {
        // The exponentials run on the transcendental unit and do overlap the partner wave's MFMAs — but only if this wave wins
        // the issue arbitration: at equal priority the (older) wave whose next instruction is an MFMA waiting for the matrix pipe
        // holds the SIMD's issue slot, and the two waves' times add (tools/micro/mfma_valu_overlap.hip: v_exp beside MFMA 3.31 ms
        // at equal priority = the sum, 1.97 ms = the max with the VALU wave at s_setprio 2; plain v_fma never overlaps).
        if (DC_ATTN_SOFTMAX_PRIO) __builtin_amdgcn_s_setprio(DC_ATTN_SOFTMAX_PRIO);
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const float mloc = mloc_s[u];
            if (QOFF) {
                // s already holds t = log2-domain score - offset.  Move the offset only where this tile's maximum runs ahead of it
                // by more than the threshold (or on the first tile of a block, to anchor it at a real maximum).
                const float m_next = (float)(bf16_t)(m_run[u] + mloc);       // the new offset, bf16-exact
                const float delta = (first || mloc > RESCALE_THR) ? m_next - m_run[u] : 0.f;
                if (__any(delta != 0.f)) {
                    const float alpha = __builtin_amdgcn_exp2f(-delta);      // first tile: O and l are still zero
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[u][j][r] -= delta;
                    if (!ONES) l_run[u] *= alpha;
#pragma unroll
                    for (int tt = 0; tt < NDT; ++tt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[u][tt][r] *= alpha;
                    m_run[u] += delta;
                    if (lh == QOFF_LH) qf[u][QOFF_KS][QOFF_E] = (bf16_t)(-m_run[u]);   // next tiles: the MFMA subtracts it
                }
                float lsum = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const float p0 = __builtin_amdgcn_exp2f(s[u][j][r]), p1 = __builtin_amdgcn_exp2f(s[u][j][r + 1]);
                        if (!ONES) lsum += p0 + p1;
                        pf[u][j][r >> 3][r & 7] = (bf16_t)p0;
                        pf[u][j][r >> 3][(r & 7) + 1] = (bf16_t)p1;
                    }
                if (!ONES) l_run[u] += lsum;
                continue;
            }
            const float m_new = fmaxf(m_run[u], mloc);
            const float mc = m_new * a.scale_log2e;
            float lsum = 0.f;
            // One v_fma_f32 per score, NOT a packed v_pk_fma_f32 (this file is built with -fno-slp-vectorize so that hipcc does not
            // re-pack them): in the peeled first tile hipcc kept (m_run * scale, mc) as ONE register pair and fed the packed FMA's
            // addend from its HIGH register for both halves (`v_pk_fma_f32 v[4:5], s[36:37], v[34:35], v[54:55] op_sel:[0,0,1]
            // op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]`), and on hardware the LOW half of that instruction intermittently came
            // out as scale * s - 0 in lanes 48..63 — one probability of the tile exp2(mc) too large (round-3 "d = 16 ping-pong
            // fragility"; DESIGN.md §5 round 4 has the ISA-level experiments).  Two scalar FMAs in its place are exact on every launch.
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    if (j >= jn) continue;
                    const float t0 = __builtin_fmaf(s[u][j][r], a.scale_log2e, -mc), t1 = __builtin_fmaf(s[u][j][r + 1], a.scale_log2e, -mc);
                    const float p0 = __builtin_amdgcn_exp2f(t0), p1 = __builtin_amdgcn_exp2f(t1);
                    if (!ONES) lsum += p0 + p1;
                    pf[u][j][r >> 3][r & 7] = (bf16_t)p0;
                    pf[u][j][r >> 3][(r & 7) + 1] = (bf16_t)p1;
                }
            if (__any(m_new != m_run[u])) {                  // the max moved for some query of this block: rescale O and l
                const float alpha = __builtin_amdgcn_exp2f(m_run[u] * a.scale_log2e - mc);   // 0 on the first tile
                if (!ONES) l_run[u] *= alpha;
#pragma unroll
                for (int tt = 0; tt < NDT; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[u][tt][r] *= alpha;
                m_run[u] = m_new;
            }
            if (!ONES) l_run[u] += lsum;
        }

        if (DC_ATTN_SOFTMAX_PRIO) __builtin_amdgcn_s_setprio(0);
    };
    // ---- O^T += V^T . P^T ; A-operand element jj of lane-half lh is key 16*s2 + 8*(jj>>2) + 4*lh + (jj&3)
    // transposing read: lane (16-lane group g4, j16) addresses key row (j16>>2), d columns 4*(j16&3).. of its block
    // and receives d = block + j16 for the block's 4 keys; each V fragment feeds QB MFMAs.  The V tile lives in buffer `buf`.
    auto pv_part = [&](int buf) {
        const char* sV = smem + buf * BUF + K_BYTES;
        const char* vbase = sV + (4 * lh + ((lane & 15) >> 2)) * V_PITCH + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
#pragma unroll
        for (int tt = 0; tt < NDT; ++tt) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    if (j >= jn) continue;
                    const char* pk = vbase + (32 * j + 16 * s2) * V_PITCH + tt * 64;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)pk);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pk + 8 * V_PITCH));
                    u32x2 l2 = *(const u32x2*)&lo, h2 = *(const u32x2*)&hi;
                    u32x4 av = {l2[0], l2[1], h2[0], h2[1]};
#pragma unroll
                    for (int u = 0; u < QB; ++u)
                        oacc[u][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8*)&av, pf[u][j][s2], oacc[u][tt], 0, 0, 0);
                }
        }
    };
    auto process_tile = [&](int t, bool first, bool stage_next) {
        qk_part(t & 1);
        rowmax_part(t);
#ifdef DC_STAMP
        __builtin_amdgcn_sched_barrier(0);
        DC_NOW(st_b);
#endif
        softmax_part(t, first);
#ifdef DC_STAMP
        __builtin_amdgcn_sched_barrier(0);
        DC_NOW(st_c);
#endif
        // the next tile's K/V (in registers since the top of this tile) go to the other LDS buffer HERE, so the writes drain under
        // the PV MFMAs instead of in front of the barrier
        if (DC_ATTN_EARLY_STAGE && stage_next) store_lds((t & 1) ^ 1);
        pv_part(t & 1);
    };

    // ---- finish: O[q][d] = O^T[d][q] / l
    auto store_out = [&]() {
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        float l_tot;
        if (ONES) l_tot = __shfl(oacc[u][ONES_T][ONES_R], lq, 64);          // row D of O^T lives in the lower lane half
        else l_tot = l_run[u] + __shfl_xor(l_run[u], 32, 64);
        const float inv = 1.0f / l_tot;
        const int qi = q0 + 32 * u + lq;
        if constexpr (SHORT) {
            // rows staged in this wave's LDS slice, then written as whole 16-byte pieces of contiguous rows: a per-lane
            // 8-byte store at a row stride touches 32 lines per instruction, and the short form lives on its stores
            char* stg = smem + 2 * BUF + wave * (QW * OPITCH) + (32 * u + lq) * OPITCH;
#pragma unroll
            for (int tt = 0; tt < NDT; ++tt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dcol = 32 * tt + 8 * g + 4 * lh;
                    if (dcol < D) {
                        bf16x4 pk;
#pragma unroll
                        for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)(oacc[u][tt][4 * g + r] * inv);
                        *(bf16x4*)(stg + dcol * 2) = pk;
                    }
                }
        } else if (qi < a.Nq) {
            bf16_t* __restrict__ O = a.o + ((long long)b * a.Nq + qi) * a.os + h * D;
#pragma unroll
            for (int tt = 0; tt < NDT; ++tt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dcol = 32 * tt + 8 * g + 4 * lh;
                    if (dcol < D) {
                        bf16x4 pk;
#pragma unroll
                        for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)(oacc[u][tt][4 * g + r] * inv);
                        *(bf16x4*)(O + dcol) = pk;
                    }
                }
        }
    }
    if constexpr (SHORT) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    // this wave's staged rows are visible to all of its lanes
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const char* stw = smem + 2 * BUF + wave * (QW * OPITCH);
        for (int i = lane; i < QW * NV; i += 64) {                // NV = 16-byte pieces per row
            const int row = i / NV, pc = i - row * NV;
            const int qi = q0 + row;
            if (qi < a.Nq)
                *(u32x4*)(a.o + ((long long)b * a.Nq + qi) * a.os + h * D + pc * 8) = *(const u32x4*)(stw + row * OPITCH + pc * 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    // reads done before the next pass overwrites the slice
        __builtin_amdgcn_wave_barrier();
    }
    };

    if constexpr (PP) {
        const int grp = wave >> 2, w4 = wave & 3;              // group 0 stages K, group 1 stages V; w4 + 4i -> 16-byte vector of a row
        u32x4 rs[NLD];
        auto pp_piece = [&](int q, int& row, int& vec) {       // K: lane -> row; V: the 4 x 4 blocks (see v_piece)
            row = lane, vec = q;
            if (NVB > 0 && grp && q < NVB) row = 16 * (q & 3) + (lane >> 2), vec = 4 * (q >> 2) + (lane & 3);
        };
        auto pp_issue = [&](int t) {                           // this group's operand of key tile t -> registers
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int q = w4 + 4 * i;
                if (q < NV) {
                    int row, vec;
                    pp_piece(q, row, vec);
                    int key = t * KV_TILE + row;
                    if (RAGGED) key = key < a.Nk ? key : a.Nk - 1;
                    const bf16_t* __restrict__ src = grp ? V + (long long)key * a.vs : K + (long long)key * a.ks;
                    rs[i] = *(const u32x4*)(src + vec * 8);
                }
            }
        };
        auto pp_store = [&](int buf) {                         // ... -> its half of LDS buffer `buf`
            char* base = smem + buf * BUF + (grp ? K_BYTES : 0);
            const int pitch = grp ? V_PITCH : K_PITCH;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int q = w4 + 4 * i;
                if (q < NV) {
                    int row, vec;
                    pp_piece(q, row, vec);
                    *(u32x4*)(base + row * pitch + vec * 16) = rs[i];
                }
            }
        };
        // LDS buffer p & 1 holds the PAIR p = {K(p), V(p-1)}: exactly what the MFMA block of tile p reads.  Global slots: group 0
        // runs its MFMA block of tile t in slot 2t and its softmax in slot 2t+1; group 1 one slot later.  Pair t+1 is written in
        // slot 2t+1 (K by group 0 at the end of its softmax, V(t) by group 1 at the end of its MFMA block) into the buffer whose
        // last reader (group 1, MFMA block t-1) finished in slot 2t-1, and is read from slot 2t+2 on.
        const int T = ntiles;
        pp_issue(0);                                           // K(0) | V(0)
        if (grp == 0) pp_store(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): the Q fragments too (see the 4-wave form)
        __syncthreads();                                       // pair 0 visible (its V half is never read)
        if (grp == 1) __syncthreads();                         // group 1 runs one slot behind
        // tile 0 (peeled: no PV yet, and the softmax anchors its offset)
        if (grp == 0 && 1 < T) pp_issue(1);                    // K(1): written at the end of this tile's softmax
        if (DC_ATTN_V_EARLY && grp == 1) {
            pp_store(1);                                       // V(0), in registers since the prologue
            if (DC_ATTN_V_EARLY == 1 && 1 < T) pp_issue(1);    // V(1): written at the START of the next MFMA block
        }
        qk_part(0);
        rowmax_part(0);
        if (!DC_ATTN_V_EARLY && grp == 1) pp_store(1);         // V(0), in registers since the prologue
        __syncthreads();
        if (DC_ATTN_V_EARLY != 1 && grp == 1 && 1 < T) pp_issue(1);   // V(1)
        softmax_part(0, true);
        if (grp == 0 && 1 < T) pp_store(1);                    // K(1)
        __syncthreads();
        for (int t = 1; t < T; ++t) {
            // MFMA block of tile t
            DC_NOW(st_a);
            if (grp == 0 && t + 1 < T) pp_issue(t + 1);
            // group 1 writes V(t) (fetched during the whole previous tile) and fetches V(t+1) at the START of its MFMA block: the LDS
            // write and its wait drain under the block's 28 MFMAs instead of standing between the row maxima and the barrier (stamps,
            // round 4: group 1's MFMA block took 1,900 cycles against group 0's 1,490, and group 0 idled 1,360 cycles per tile at the
            // end barrier).  Pair t+1's buffer was last read one full tile ago, so the write may come anywhere in this block.
            if (DC_ATTN_V_EARLY && grp == 1) {
                pp_store((t + 1) & 1);                         // V(t)
                if (DC_ATTN_V_EARLY == 1 && t + 1 < T) pp_issue(t + 1);
            }
#if DC_ATTN_QK_FIRST
            // scores first: their row maxima (a dependent v_max3 chain behind the last QK^T MFMA) then issue in the shadow of the 16 PV
            // MFMAs instead of as a tail in front of the barrier; 236 VGPRs, no spill (round 4, same-box A/B: 950 -> 930-939 us)
            qk_part(t & 1);
            pv_part(t & 1);
            rowmax_part(t);
#else
            pv_part(t & 1);                                    // V(t-1), probabilities of tile t-1
            __builtin_amdgcn_sched_barrier(0);                 // PV before QK^T: the probabilities die before the new scores are born
            qk_part(t & 1);
            rowmax_part(t);
#endif
            // the staging stores stay where they are written: without the fence hipcc hoists the `s_waitcnt vmcnt(0)` + ds_write of V(t)
            // up among the block's MFMAs, where the wait stalls the matrix pipe (stamps, round 4: group 1's MFMA block 1,900 -> 1,570 cycles)
            if (DC_ATTN_STORE_FENCE >= 1) __builtin_amdgcn_sched_barrier(0);
#ifdef DC_STAMP
            DC_NOW(st_v);
#endif
            if (!DC_ATTN_V_EARLY && grp == 1) pp_store((t + 1) & 1);               // V(t)
#ifdef DC_STAMP
            __builtin_amdgcn_sched_barrier(0);
            DC_NOW(st_b);
            s_vst += st_b - st_v;
#endif
            __syncthreads();
            // softmax block of tile t
            DC_NOW(st_c);
            if (DC_ATTN_V_EARLY != 1 && grp == 1 && t + 1 < T) pp_issue(t + 1);
            softmax_part(t, false);
            if (DC_ATTN_STORE_FENCE >= 2) __builtin_amdgcn_sched_barrier(0);   // K(t+1) is written AFTER the exponentials (hipcc puts its vmcnt(0) in front of them)
            if (grp == 0 && t + 1 < T) pp_store((t + 1) & 1);  // K(t+1)
#ifdef DC_STAMP
            __builtin_amdgcn_sched_barrier(0);
            DC_NOW(st_d);
#endif
            __syncthreads();
#ifdef DC_STAMP
            DC_NOW(st_e);
            s_qk += st_b - st_a, s_sm += st_c - st_b, s_pv += st_d - st_c, s_st += st_e - st_d;   // MFMA block | barrier | softmax | barrier
#endif
        }
        pv_part(T & 1);                                        // V(T-1)
        if (grp == 0) __syncthreads();                         // group 0's last barrier; group 1 is already past its last one
#ifdef DC_STAMP
        DC_NOW(st_e);
        DC_NOW_RT(rt_e);
#endif
        store_out();
#ifdef DC_STAMP
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // the output stores have left
        DC_NOW(st_out);
        DC_NOW_RT(rt_out);
        if (lane == 0 && (long long)blockIdx.x * 8 + wave < (1 << 14)) {
            unsigned long long* o = dc_attn_stamp_buf + ((long long)blockIdx.x * 8 + wave) * DC_STAMP_SLOTS;
            o[0] = s_qk, o[1] = s_sm, o[2] = s_pv, o[3] = s_st, o[4] = st_e - st_0, o[5] = 1;
            o[6] = st_out - st_in, o[7] = rt_e - rt_0, o[8] = rt_in, o[9] = rt_out;
            o[10] = __builtin_amdgcn_s_getreg((31 << 11) | 4), o[11] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_ID, XCC_ID
            o[12] = st_0 - st_in, o[13] = st_out - st_e;          // prologue | output
            o[14] = s_vst;                                        // group 1: V(t) wait + LDS write at the end of the MFMA block
        }
#endif
    } else if constexpr (!SHORT) {
        for (int t = 0; t < ntiles; ++t) {
            const bool more = t + 1 < ntiles;
            DC_NOW(st_a);
            if (more) issue_loads(t + 1);
            process_tile(t, t == 0, more);
#ifdef DC_STAMP
            __builtin_amdgcn_sched_barrier(0);
            DC_NOW(st_d);
#endif
            if (!DC_ATTN_EARLY_STAGE && more) store_lds((t & 1) ^ 1);
            __syncthreads();
#ifdef DC_STAMP
            DC_NOW(st_e);
            s_qk += st_b - st_a, s_sm += st_c - st_b, s_pv += st_d - st_c, s_st += st_e - st_d;
#endif
        }
        store_out();
#ifdef DC_STAMP
        DC_NOW(st_e);
        if (lane == 0) {
            unsigned long long* o = dc_attn_stamp_buf + ((long long)blockIdx.x * 4 + wave) * DC_STAMP_SLOTS;
            if ((long long)blockIdx.x * 4 + wave < (1 << 14)) o[0] = s_qk, o[1] = s_sm, o[2] = s_pv, o[3] = s_st, o[4] = st_e - st_0;
        }
#endif
    } else {
        if (ntiles > 1) {                                      // second key tile: staged once, like the first
            issue_loads(1);
            store_lds(1);
            __syncthreads();
        }
        for (int pass = 0;; ++pass) {                          // K/V stay put: the waves run on independently, block after block
            const bool has_next = pass + 1 < SHORT_PASSES && q0 + 4 * QW < a.Nq;
            bf16x8 qn[QB][ND16];
            if (has_next) fetch_q(q0 + 4 * QW, qn);            // next block's queries fly while this one computes
            // (round 4, measured and NOT adopted: query blocks prefetched TWO passes ahead — 61 -> 64.5 us at 32 x 8 x 4096 x 77; and the skipped
            //  dead half below removes a quarter of the form's MFMA / softmax work without moving its time: the form is bound by neither)
            for (int t = 0; t < ntiles; ++t) {
                jn = (t * KV_TILE + 32 >= a.Nk) ? 1 : 2;       // (RAGGED is always on in this form)
                process_tile(t, t == 0, false);
            }
            store_out();
            if (!has_next) break;
            q0 += 4 * QW;
#pragma unroll
            for (int u = 0; u < QB; ++u)
#pragma unroll
                for (int ks = 0; ks < ND16; ++ks) qf[u][ks] = qn[u][ks];
            reset_acc();
        }
    }
}

template <int D, int QB, bool SHORT, bool RAGGED, bool PP = false>
int launch_qb_r(const AttnArgs& a, hipStream_t st)
{
    constexpr int ND16 = (D + 15) / 16, NDT = (D + 31) / 32;
    constexpr int KP16 = (ND16 * 2) | 1;
    constexpr int BUF = KV_TILE * KP16 * 16 + KV_TILE * v_pitch_bytes(NDT);
    const size_t lds = 2 * BUF + (SHORT ? 4 * (32 * QB) * (D * 2 + 16) : 0);
    auto kern = attn_kernel<D, QB, SHORT, RAGGED, PP>;
    static std::atomic<unsigned long long> attr_done{0};
    dc_set_max_dyn_lds((const void*)kern, (int)lds, attr_done);
    constexpr int QWG = (PP ? 256 : 128) * QB * (SHORT ? SHORT_PASSES : 1);
    const int qblocks = (a.Nq + QWG - 1) / QWG;
    hipLaunchKernelGGL(kern, dim3(a.B * a.heads * qblocks), dim3(PP ? 512 : 256), lds, st, a);
    return dc_launch_status();
}

template <int D, int QB, bool SHORT>
int launch_qb(const AttnArgs& a, hipStream_t st)
{
    if (SHORT || (a.Nk % KV_TILE) != 0) return launch_qb_r<D, QB, SHORT, true>(a, st);
    return launch_qb_r<D, QB, SHORT, false>(a, st);
}

template <int D>
int launch(const AttnArgs& a, hipStream_t st)
{
    // two query blocks per wave only for the small heads (register budget) and only when that still leaves >= 2 workgroups per CU
    static const int force_qb = DC_KNOB("DC_ATTN_QB", 0);      // developer knob
    static const int no_short = DC_KNOB("DC_ATTN_NO_SHORT", 0);   // developer knob (A/B)
    // short context (text cross-attention): keys resident, several query blocks per workgroup — when enough workgroups remain
    const bool short_ctx = !no_short && a.Nk <= 2 * KV_TILE;
    if constexpr (D <= 48) {          // d = 80 spills at two blocks per wave (measured slower)
        const long long wgs2 = (long long)a.B * a.heads * ((a.Nq + 255) / 256);
        if (force_qb == 2 || (force_qb == 0 && wgs2 >= 512)) {
            // short context: one query block per wave when that still fills the chip (measured at 32 x 8 x 4096 x 77, d = 40: 63 us
            // against 70-73 us for two blocks per wave — the keys are resident either way, and the smaller workgroup tail wins)
            if (short_ctx && force_qb == 0 && (long long)a.B * a.heads * ((a.Nq + 127) / 128) / SHORT_PASSES >= 512) return launch_qb<D, 1, true>(a, st);
            if (short_ctx && wgs2 / SHORT_PASSES >= 512) return launch_qb<D, 2, true>(a, st);
            // long context with at least one 8-wave workgroup per CU: the ping-pong form (DC_ATTN_PP=0/1: developer A/B knob)
            static const int force_pp = DC_KNOB("DC_ATTN_PP", -1);
            const long long wgs_pp = (long long)a.B * a.heads * ((a.Nq + 511) / 512);
            if (!short_ctx && (force_pp == 1 || (force_pp < 0 && wgs_pp >= 256 && a.Nk >= 4 * KV_TILE))) {
                if ((a.Nk % KV_TILE) != 0) return launch_qb_r<D, 2, false, true, true>(a, st);
                return launch_qb_r<D, 2, false, false, true>(a, st);
            }
            return launch_qb<D, 2, false>(a, st);
        }
    }
    const long long wgs1 = (long long)a.B * a.heads * ((a.Nq + 127) / 128);
    if (short_ctx && wgs1 / SHORT_PASSES >= 512) return launch_qb<D, 1, true>(a, st);
    return launch_qb<D, 1, false>(a, st);
}

// Row softmax fp32 -> bf16 (one workgroup per row; cols <= 65536).
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, bf16_t* __restrict__ p,
                                                            int cols, float scale)
{
    __shared__ float red[8];
    const long long row = blockIdx.x;
    const float* x = s + row * cols;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < cols; i += 256) m = fmaxf(m, x[i] * scale);
    m = dc_wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int i = threadIdx.x; i < cols; i += 256) sum += __expf(x[i] * scale - m);
    sum = dc_wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    bf16_t* o = p + row * cols;
    for (int i = threadIdx.x; i < cols; i += 256) o[i] = (bf16_t)(__expf(x[i] * scale - m) * inv);
}

}  // namespace

extern "C" int dc_attention_bf16(const void* q, const void* k, const void* v, void* out, int B, int heads, int Nq,
                                 int Nk, int D, long long q_stride, long long k_stride, long long v_stride,
                                 long long o_stride, float scale, void* stream)
{
    if (!q || !k || !v || !out || B <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0) return DC_ERR_INVALID;
    if ((q_stride | k_stride | v_stride | o_stride) & 7) return DC_ERR_INVALID;     // 16-byte aligned rows
    AttnArgs a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, B, heads, Nq, Nk,
               q_stride, k_stride, v_stride, o_stride, scale * 1.4426950408889634f, DC_KNOB("DC_ATTN_XCD", 1)};
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 8: return launch<8>(a, st);
        case 16: return launch<16>(a, st);
        case 32: return launch<32>(a, st);
        case 40: return launch<40>(a, st);
        case 64: return launch<64>(a, st);
        case 80: return launch<80>(a, st);
        case 128: return launch<128>(a, st);
        case 160: return launch<160>(a, st);
        default: return DC_ERR_INVALID;
    }
}

extern "C" int dc_softmax_rows_f32_to_bf16(const float* s, void* p, long long rows, int cols, float scale, void* stream)
{
    if (!s || !p || rows <= 0 || cols <= 0) return DC_ERR_INVALID;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, (bf16_t*)p, cols, scale);
    return dc_launch_status();
}
