// 256 x 256 macro-tile GEMM  out[m][n] = sum_k X[m][k] * W[n][k]  for gfx950, eight waves, four phases per 64-deep K tile: the
// long-K, wide-N linears of the transformer blocks (ff.net.0.proj with GEGLU, fused QKV; same call sites as gemm_dma.hip) where
// the 256 x 160 kernel of gemm_wide.hip is bound by what a wave must ISSUE per MFMA (one LDS-DMA piece per 6.7 MFMAs, one fragment
// read per 2.2) rather than by the matrix pipe.
//
// Geometry (the guide's 256-square schedule, cdna_hip_programming.md "The 256^2 8-phase template", re-cut for this code base's
// operand layouts and epilogues): waves 0-3 (group 0) own rows 0-127 of the tile, waves 4-7 (group 1) rows 128-255; wave column
// wc = wave & 3 owns columns 64 wc .. 64 wc + 63: a 128 x 64 wave tile = 8 x 4 MFMA blocks = 128 accumulator registers, worked as
// four 64 x 32 quadrants, one per PHASE.  A K tile (64 deep) is staged as four 16 KB HALF TILES of 128 rows x 128 B:
//   H0 = X rows  {64 rows of group 0, 64 rows of group 1} that the first two quadrants read   (rows  0-63  of each group)
//   H1 = W rows  {32 of every wave column}                      first and fourth quadrant       (cols  0-31  of each wave column)
//   H2 = W rows  {the other 32 of every wave column}            second and third quadrant       (cols 32-63)
//   H3 = X rows  {rows 64-127 of each group}                    third and fourth quadrant
// two K tiles of LDS (128 KB).  A phase of a wave is
//   R: its fragment reads (12 | 4 | 8 | 0 ds_read_b128) + its two LDS-DMA pieces of ONE half tile of a later K tile;
//      s_waitcnt lgkmcnt(0) (+ the counted vmcnt in the fourth phase); s_barrier
//   M: 16 MFMAs (one quadrant x 64 of K); s_barrier
// and group 1 runs one barrier behind group 0 (an extra barrier up front), so each SIMD — it hosts one wave of each group —
// always has one wave in M while the other is in R.  The W fragments of H1 stay in registers from the first to the fourth phase.
// DMA order: phase 1 of tile t issues H3(t+1), phases 2-4 issue H0, H1, H2 of tile t+2 into the slots of tile t, each freed by
// the reads of an earlier phase:
//   RAW  every wave waits (fourth phase, before its first barrier) until all but its 6 youngest pieces have landed, i.e. all of
//        tile t+1; both groups' waits precede the barrier that ends group 0's phase, and no wave reads tile t+1 before it;
//   WAR  a slot is refilled at the earliest in the phase after its last read, and every read has RETURNED (lgkmcnt(0)) before
//        the barrier that closes the reading phase of either group.
// Accumulation order over K is the tile kernels' (ascending 32-wide MFMA steps), the epilogue arithmetic is theirs
// (dc_common.h): outputs are bit-identical with gemm_dma.hip / gemm_wide.hip.
#include "dc_common.h"
#include "../../include/diffcodec_hip.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#ifndef P8_ABL
#define P8_ABL 0                                    // developer timing ablations (wrong results): 1 no DMA in the K loop, 2 no fragment reads, 4 no MFMAs, 8 no barriers, 16 no output stores
#endif
constexpr int P8_HALF = 128 * 128;                  // bytes of a half tile
constexpr int P8_BUF = 4 * P8_HALF;                 // one K tile
constexpr int P8_RING = 2 * P8_BUF;                 // 128 KB
constexpr int P8_STAGE_MAX = 256 * (256 * 2 + 16);  // staged output tile (135,168 B) — overlays the ring after the K loop
constexpr int P8_PAR_BIAS = P8_STAGE_MAX;           // fp32 [256] bias, [256] weight column sums, [256][2] (mean, rstd)
constexpr int P8_PAR_CS = P8_PAR_BIAS + 1024;
constexpr int P8_PAR_LN = P8_PAR_CS + 1024;
constexpr int P8_LDS = P8_PAR_LN + 2048;

__device__ __forceinline__ void p8_barrier()
{
    if (!(P8_ABL & 8)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int N>
__device__ __forceinline__ void p8_wait_vm_lgkm()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void p8_wait_lgkm()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// EPI: 1 bias, 3 folded LayerNorm + bias, 4 GEGLU, 5 folded LayerNorm + GEGLU (the numbering of gemm_dma.hip)
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_p8_kernel(const dc_conv_desc d, const int gm)
{
    constexpr bool e_geglu = EPI >= 4, e_ln = EPI == 3 || EPI == 5;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;

    const int M = d.N * d.Ho * d.Wo;
    const int K = d.C1;
    const int nk = K >> 6;
    const int n_tiles = (d.Cout + 255) >> 8;                // the last N tile may be half full (Cout % 256 == 128): clamped W rows, guarded stores
    const int nblk = n_tiles * (M >> 8);
    int bid = blockIdx.x;
    {
        const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + idx;
    }
    // Tile order inside an XCD's contiguous range: groups of `gm` row tiles, the row tile the fastest index inside a group, so the 32
    // workgroups an XCD runs together form a gm x (32 / gm) block of tiles and share gm X panels + 32 / gm W panels in its L2
    // (column-fastest order: 1 + 32 panels — 7.4x the algorithmic bytes from beyond L2 at N = 10240).
    int tile_m, tile_n;
    if (gm > 0) {
        const int per_group = gm * n_tiles;
        const int group = bid / per_group, in_group = bid - group * per_group;
        const int first_m = group * gm;
        const int rows = min(gm, (M >> 8) - first_m);
        tile_m = first_m + in_group % rows;
        tile_n = in_group / rows;
    } else {
        tile_n = bid % n_tiles;
        tile_m = bid / n_tiles;
    }
    const int m0 = tile_m << 8, n0 = tile_n << 8;

    // ---- DMA sources.  Piece j of a half tile = its rows [8j, 8j+8); this wave issues pieces `wave` and `wave + 8`; lane s ->
    //      row 8j + (s >> 3), LDS slot s & 7, which holds source chunk (s & 7) ^ (row & 7).
    //      Sources are a workgroup-uniform base (SGPRs: tile origin + K offset of the half tile) plus a 32-bit per-lane offset, so a
    //      piece costs no vector address arithmetic.
    uint32_t offx[2], offw[2];                       // H0 / H1 (H3 = + 64 rows, H2 = + 32 rows: in the uniform base)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave + 8 * i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        offx[i] = (uint32_t)(((row >> 6) * 128 + (row & 63)) * K + chunk * 8) * 2u;
        int wn = n0 + (row >> 5) * 64 + (row & 31);
        // columns past Cout (whole 64-column wave slices of a half-full last tile) are computed on valid rows and never stored: row
        // Cout - 33 for H1, so that H2 (the same offsets + 32 rows) ends on row Cout - 1
        wn = wn < d.Cout ? wn : d.Cout - 33;
        offw[i] = (uint32_t)((wn - n0) * K + chunk * 8) * 2u;
    }
    const char* const bx = (const char*)d.x1 + (long long)m0 * K * 2;
    const char* const bw = (const char*)d.w + (long long)n0 * K * 2;
    const long long x_h3 = (long long)64 * K * 2, w_h2 = (long long)32 * K * 2;
    // half tile h of K tile kt into buffer `buf`
    auto issue_half = [&](int h, int kt, int buf) {
        char* base = smem + buf * P8_BUF + h * P8_HALF + wave * 1024;
        const bool is_x = h == 0 || h == 3;
        const char* sb = (is_x ? bx : bw) + (h == 3 ? x_h3 : (h == 2 ? w_h2 : 0));
        const uint32_t ko = (uint32_t)kt * 128u;     // the K offset rides in the 32-bit lane offset (one v_add_u32 per piece), the rest in SGPRs
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(sb + (size_t)(uint32_t)((is_x ? offx[i] : offw[i]) + ko)), (lptr_t)(base + i * 8192), 16, 0, 0);
    };

    // ---- fragment read bases (byte offsets into a K tile's buffer; + 2048 per 16-row block, + half-tile offset)
    int xa_off[2], wb_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int swz = ((4 * s + fq) ^ (fr & 7)) << 4;
        xa_off[s] = (grp * 64 + fr) * 128 + swz;
        wb_off[s] = (wc * 32 + fr) * 128 + swz;
    }

    // epilogue operands: fetched in front of the DMA prologue, parked in LDS above the ring once it has been issued (no registers
    // held through the K loop; hipcc's own counted vmcnt retires exactly these three loads, the oldest of the queue)
    // (both halves of the workgroup fetch and store the same 256 entries: no divergent branch for the wait-count pass to merge)
    const int pt = tid & 255;
    const int pn = n0 + pt < d.Cout ? n0 + pt : d.Cout - 1;
    const float par_b = d.bias ? d.bias[pn] : 0.f;
    float par_c = 0.f;
    f32x2 par_mr = {0.f, 0.f};
    if (e_ln) {
        par_c = d.ln_colsum[pn];
        par_mr = *(const f32x2*)(d.ln_stats + (long long)(m0 + pt) * 2);
    }
    __builtin_amdgcn_sched_barrier(0);               // the three loads stay in front of every DMA piece

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xa[2][4], wb0[2][2], wb1[2][2];

    // prologue: all of tile 0, H0-H2 of tile 1
#pragma unroll
    for (int h = 0; h < 4; ++h) issue_half(h, 0, 0);
    if (nk > 1) {
#pragma unroll
        for (int h = 0; h < 3; ++h) issue_half(h, 1, 1);
    }
    *(float*)(smem + P8_PAR_BIAS + pt * 4) = par_b;
    *(float*)(smem + P8_PAR_CS + pt * 4) = par_c;
    *(f32x2*)(smem + P8_PAR_LN + pt * 8) = par_mr;
    if (nk > 1) p8_wait_vm_lgkm<6>();
    else p8_wait_vm_lgkm<0>();
    p8_barrier();                                    // tile 0 visible to every wave
    if (grp == 1) p8_barrier();                      // group 1 runs one barrier behind

    auto read_x = [&](const char* buf, int half_off) {
        if ((P8_ABL & 2) && buf != smem) return;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) xa[s][tm] = *(const bf16x8*)(buf + half_off + xa_off[s] + tm * 2048);
    };
    auto read_w = [&](bf16x8 (&wb)[2][2], const char* buf, int half_off) {
        if ((P8_ABL & 2) && buf != smem) return;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) wb[s][tn] = *(const bf16x8*)(buf + half_off + wb_off[s] + tn * 2048);
    };
    auto quadrant = [&](auto nh_c, auto mh_c, const bf16x8 (&wb)[2][2]) {
        constexpr int nh = decltype(nh_c)::value, mh = decltype(mh_c)::value;
        if (P8_ABL & 4) {
            asm volatile("" ::"v"(xa[0][0]), "v"(xa[1][3]), "v"(wb[0][0]), "v"(wb[1][1]));
            return;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[nh * 2 + tn][mh * 4 + tm] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[s][tn], xa[s][tm], acc[nh * 2 + tn][mh * 4 + tm], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // One K tile.  STEADY: tiles t+1 and t+2 exist (no conditions in the body).
    auto k_tile = [&](auto par_c, auto steady_c, int t) {
        constexpr int P = decltype(par_c)::value;
        constexpr bool STEADY = decltype(steady_c)::value;
        const char* buf = smem + P * P8_BUF;
        const bool n1 = STEADY || t + 1 < nk, n2 = STEADY || t + 2 < nk;
        // phase 1: X rows 0-63, W cols 0-31
        read_w(wb0, buf, 1 * P8_HALF);
        __builtin_amdgcn_sched_barrier(0);
        read_x(buf, 0);
        if (n1 && !(P8_ABL & 1)) issue_half(3, t + 1, P ^ 1);
        p8_wait_lgkm();
        p8_barrier();
        quadrant(I0{}, I0{}, wb0);
        p8_barrier();
        // phase 2: W cols 32-63
        read_w(wb1, buf, 2 * P8_HALF);
        if (n2 && !(P8_ABL & 1)) issue_half(0, t + 2, P);
        p8_wait_lgkm();
        p8_barrier();
        quadrant(I1{}, I0{}, wb1);
        p8_barrier();
        // phase 3: X rows 64-127
        read_x(buf, 3 * P8_HALF);
        if (n2 && !(P8_ABL & 1)) issue_half(1, t + 2, P);
        p8_wait_lgkm();
        p8_barrier();
        quadrant(I1{}, I1{}, wb1);
        p8_barrier();
        // phase 4: no reads (W cols 0-31 are still in registers); all of tile t+1 must have landed before the next phase reads it
        if (n2 && !(P8_ABL & 1)) issue_half(2, t + 2, P);
        if (n2) p8_wait_vm_lgkm<6>();
        else p8_wait_vm_lgkm<0>();
        p8_barrier();
        quadrant(I0{}, I1{}, wb0);
        p8_barrier();
    };
    using BT = std::integral_constant<bool, true>;
    using BF = std::integral_constant<bool, false>;
    int t = 0;
    for (; t + 3 < nk; t += 2) {
        k_tile(I0{}, BT{}, t);
        k_tile(I1{}, BT{}, t + 1);
    }
    for (; t + 1 < nk; t += 2) {
        k_tile(I0{}, BF{}, t);
        k_tile(I1{}, BF{}, t + 1);
    }
    if (t < nk) k_tile(I0{}, BF{}, t);
    if (grp == 0) p8_barrier();                      // pairs with group 1's last barrier: no LDS read or DMA is pending past it

    // ---- epilogue: rows staged through LDS, written as whole 16-byte pieces of contiguous output rows
    constexpr int OC = e_geglu ? 128 : 256;
    constexpr int PITCH = OC * 2 + 16;
    static_assert(256 * PITCH <= P8_STAGE_MAX, "staged tile must end below the epilogue operands");
#pragma unroll
    for (int tm = 0; tm < 8; ++tm) {
        const int row = grp * 128 + tm * 16 + fr;
        f32x2 mr = {0.f, 0.f};
        if (e_ln) mr = *(const f32x2*)(smem + P8_PAR_LN + row * 8);
        if (e_geglu) {
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                const int nl = wc * 64 + tp * 32 + 4 * fq;                      // value block; gate block = + 16
                f32x4 h = acc[2 * tp][tm], g = acc[2 * tp + 1][tm];
                if (e_ln) {
                    h = dc_ln_fold(h, mr[0], mr[1], *(const f32x4*)(smem + P8_PAR_CS + nl * 4));
                    g = dc_ln_fold(g, mr[0], mr[1], *(const f32x4*)(smem + P8_PAR_CS + (nl + 16) * 4));
                }
                h += *(const f32x4*)(smem + P8_PAR_BIAS + nl * 4);
                g += *(const f32x4*)(smem + P8_PAR_BIAS + (nl + 16) * 4);
                bf16x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)(h[r] * dc_gelu_erf(g[r]));
                *(bf16x4*)(smem + row * PITCH + (((wc * 32 + tp * 16 + 4 * fq) * 2) ^ dc_stage_swz(row))) = pk;
            }
        } else {
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                const int nl = wc * 64 + tn * 16 + 4 * fq;
                f32x4 v = acc[tn][tm];
                if (e_ln) v = dc_ln_fold(v, mr[0], mr[1], *(const f32x4*)(smem + P8_PAR_CS + nl * 4));
                v += *(const f32x4*)(smem + P8_PAR_BIAS + nl * 4);
                v *= d.out_scale;
                bf16x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)v[r];
                *(bf16x4*)(smem + row * PITCH + ((nl * 2) ^ dc_stage_swz(row))) = pk;
            }
        }
        __builtin_amdgcn_sched_barrier(0);           // one row group at a time (register pressure)
    }
    __syncthreads();
    const int out_cols = e_geglu ? d.Cout >> 1 : d.Cout;
    const int col0 = e_geglu ? n0 >> 1 : n0;
    bf16_t* __restrict__ o = (bf16_t*)d.out;
    constexpr int pieces = OC / 8;                   // 16-byte pieces per staged row
    for (int i = tid; i < 256 * pieces; i += 512) {
        const int row = i / pieces, pc = i - row * pieces;
        if ((P8_ABL & 16) && row != 1000) continue;
        if (col0 + pc * 8 < out_cols)
            *(u32x4*)(o + (long long)(m0 + row) * out_cols + col0 + pc * 8) = dc_stage_unswz(*(const u32x4*)(smem + row * PITCH + pc * 16), row);
    }
}

template <int EPI>
int launch_p8(const dc_conv_desc& d, hipStream_t st)
{
    const long long M = (long long)d.N * d.Ho * d.Wo;
    const int nblk = (int)(M >> 8) * ((d.Cout + 255) >> 8);
    static const int gm = DC_KNOB("DC_P8_GM", 8);           // developer A/B knob: 0 = column-fastest order
    auto kern = gemm_p8_kernel<EPI>;
    static std::atomic<unsigned long long> attr_done{0};
    dc_set_max_dyn_lds((const void*)kern, P8_LDS, attr_done);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), P8_LDS, st, d, gm);
    return dc_launch_status();
}

}  // namespace

// The 256 x 256 kernel takes a launch when its epilogue modes apply (no residual, no statistics), the operands are whole tiles,
// and the tile grid fills the chip for more than one round.  DC_GEMM_P8: 0 = never (A/B), 1 = default rule, 2 = whenever legal.
int dc_gemm_p8_wanted(const dc_conv_desc& d, int epi)
{
    static const int mode = DC_KNOB("DC_GEMM_P8", 1);
    static const int min_k = DC_KNOB("DC_GEMM_P8_MIN_K", 640);
    static const int min_tiles = DC_KNOB("DC_GEMM_P8_MIN_TILES", 448);
    if (mode == 0 || !(epi == 1 || epi == 3 || epi == 4 || epi == 5)) return 0;
    if (d.ksize != 1 || d.gn_ab || d.splitk > 1 || d.out_f32 || d.C2 != 0 || d.residual || d.stats_out || d.gn_part_out) return 0;
    if (d.ln_stats && d.ln_parts > 0) return 0;             // the dispatcher finalizes first and comes back with pairs
    const int K = d.C1;
    const long long M = (long long)d.N * d.Ho * d.Wo;
    if (K < 128 || (K & 63) || (M & 255) || (d.Cout & 127)) return 0;
    if (mode == 2) return 1;
    const long long tiles = (M >> 8) * ((d.Cout + 255) >> 8);
    if ((d.Cout & 255) && d.Cout < 1792) return 0;          // a half-full last tile only where it is <= 1/15 of the columns' work
    return K >= min_k && tiles >= min_tiles;
}

int dc_gemm_p8_launch(const dc_conv_desc& d, int epi, hipStream_t st)
{
    switch (epi) {
        case 1: return launch_p8<1>(d, st);
        case 3: return launch_p8<3>(d, st);
        case 4: return launch_p8<4>(d, st);
        case 5: return launch_p8<5>(d, st);
        default: return DC_ERR_INVALID;
    }
}
