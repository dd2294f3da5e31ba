// 3x3 stride-1 convolution as an LDS-staged 2D-tile implicit GEMM for gfx950 (the ResBlock / Upsample / FDN / VAE
// conv workhorse; replaces F.conv2d(k=3, padding=1) inside the diffusers blocks called from flownet.py:83-124 and
// pipeline.py:358-367,391).
//
// Per workgroup (256 threads, 4 waves as 2(m) x 2(n)): an output patch of TH x 16 pixels of one sample x BN output
// channels.  For every 64-channel slice of the input, the (TH+2) x 18 pixel HALO of that patch is brought into LDS
// ONCE (128-byte pixel rows, GroupNorm affine + SiLU applied in fp32 on the way, zero padding materialised), and the
// nine taps are nine K-steps that read their A fragments from the same halo image at shifted pixel positions; only
// the weight tile (BN x 64 per tap) streams per K-step, double-buffered.  Compared with a per-tap gather this cuts the
// activation traffic and the normalisation work 9x and leaves almost no address arithmetic in the K-loop.
// The nearest-2x upsample of diffusers' Upsample2D is fused: the halo is taken from the low-resolution input and the
// fragment positions are halved per tap.
//
// The weight tile of each tap goes global -> LDS by `global_load_lds_dwordx4` (LDS-DMA: no VGPR staging, no ds_write)
// into a ring of NSTB stages; a counted `s_waitcnt vmcnt` + one raw `s_barrier` per K-step keep NSTB-2 stages in
// flight across the barrier, so the K-loop is MFMA/LDS-paced instead of load-latency-paced.
//
// LDS images are row-major 128-B rows with the 16-B chunk index XORed by (row & 7): ds_write_b128 of a row piece and
// the v_mfma_f32_16x16x32_bf16 fragment ds_read_b128 are both bank-conflict-free, and rows stay whole 128-B lines.
#include "dc_common.h"
#include "../../include/diffcodec_hip.h"
#include <cstdlib>
#include <utility>

#ifndef DC_EPI_SPECIALIZE
#define DC_EPI_SPECIALIZE 1     // developer A/B switch: 0 = every launch takes the generic run-time-flag epilogue
#endif
#ifndef DC_CONV_FAST
#define DC_CONV_FAST 1          // developer A/B switch: 0 = plain maps also take the XOR-swizzled, whole-step K loop
#endif
#ifndef DC_SH_HP
#define DC_SH_HP 160            // halo pixel pitch of the narrow-map (8-wide) pipelined form (developer A/B: 144 / 176 / 192)
#endif
#ifndef DC_CONV_PIPE
#define DC_CONV_PIPE 1          // developer A/B switch for the scheduled K-step (see `mfma_frags`)
#endif

// Developer-only phase stamps (tools/conv_stamp.py builds this file with -DDC_STAMP into a scratch .so): s_memtime sums of
// the K loop's barrier waits, its fragment-read + MFMA phases and the halo swaps, plus entry / loop / exit times, written to
// the (otherwise unused) split-K workspace.  Never defined in the product build.
#ifdef DC_STAMP
#define DC_NOW(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#else
#define DC_NOW(t) (void)0
#endif

namespace {

typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

template <int... I, class F>
__device__ __forceinline__ void dc_static_for(std::integer_sequence<int, I...>, F&& f)
{
    (f(std::integral_constant<int, I>{}), ...);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// EPI: epilogue specialisation (same reasoning as gemm_dma.hip): 0 = generic run-time flags (split-K slabs, fp32 output,
// activation), 1 = bias (+ time-embedding row) -> bf16, 2 = the same + scaled residual add.  gn_part_out stays a run-time,
// workgroup-uniform test outside the loops.
//
// FAST (plain maps at least 16 wide, no fused upsample; NSTB == 2, or 4 for grids of at most one workgroup per CU, whose K loop
// is otherwise paced by one weight-DMA round trip per tap): the halo image uses 160-byte pixel rows WITHOUT the XOR
// swizzle (16 consecutive pixels at one channel chunk are still bank-conflict-free), so a tap is an immediate offset from one
// per-lane base address and the nine taps are unrolled with no address arithmetic left in the K loop; and the fragment reads
// are software-pipelined ACROSS K-steps in two half-steps (32 channels each): while the MFMAs of one half run, the reads of
// the next half are in flight, with the workgroup barrier in the middle of the step (stage k+1 must be visible before its
// first half is read).  A wave then never sits at the top of a step waiting for 18 reads with no MFMA to issue.
// UPS (FAST only, no fused GroupNorm): the nearest-2x upsample of Upsample2D on the same pipeline — the halo is taken from the
// low-resolution input ((TH/2 + 2) x 10 pixels), a tap's row offset is still an immediate, and its column offset ((x + kx - 1) >> 1)
// takes one of three per-lane base addresses.
// SH (FAST only, no fused GroupNorm): 8-wide maps — an MFMA row group covers two image rows of 8 pixels (and, with 8-row tiles, a
// tile covers two whole 8x8 images); only the per-lane base address and the immediates differ.  Direct (unstaged) stores.
template <int TM, int TN, bool GN, int NSTB, int EPI, bool FAST, bool UPS = false, bool SH = false>
__global__ __launch_bounds__(256, 2) void conv3x3_tile_kernel(const dc_conv_desc d, const int m_fastest)
{
    static_assert(!SH || (FAST && !GN && !UPS), "the narrow-map form exists on the pipelined path, without GroupNorm on load");
    static_assert(!FAST || NSTB == 2 || NSTB == 4, "the half-step pipeline indexes its weight ring with step & (NSTB - 1)");
    static_assert(!UPS || (FAST && !GN), "the upsample form exists on the pipelined path, without GroupNorm on load");
    constexpr bool GENERIC = EPI == 0;
    const bool e_split = GENERIC && d.splitk > 1;
    const bool e_res = GENERIC ? (d.residual != nullptr && d.splitk <= 1) : EPI == 2;
    const int e_act = GENERIC ? d.act : 0;
    const bool e_f32 = GENERIC && d.out_f32;
    constexpr int WM = 2, WN = 2;
    constexpr int TH = WM * TM;                       // output rows per tile (tile is TH x 16 pixels)
    constexpr int BN = WN * TN * 16;
    constexpr int HALO_MAX = UPS ? (TH / 2 + 2) * 10 : SH ? (TM == 4 ? 200 : ((TH << 1) + 2) * 10) : ((TM == 4 && !FAST) ? 200 : (TH + 2) * 18);   // TM == 4 also serves two stacked 8x8 images (2 x 10 x 10)
    constexpr int NHU = (HALO_MAX * 8 + 255) / 256;   // 16-byte halo units per thread
    constexpr int NB = BN / 32;                       // weight-tile DMA wave-instructions per wave per stage
    constexpr int HP = FAST ? (SH ? DC_SH_HP : 160) : 128;   // halo pixel-row pitch (bytes)
    constexpr int H_BYTES = HALO_MAX * HP;
    constexpr int B_BYTES = BN * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sH = smem;
    char* const sB0 = smem + H_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int q = tid & 7;
    const int fr = lane & 15, fq = lane >> 4;

    const int Cin = d.C1 + d.C2;
    const int nchunks = Cin >> 6;
    // narrow maps (Wo == 8): the 16 lanes of an MFMA row cover 2 image rows of 8 pixels (sh = 1)
    const int sh = FAST ? (SH ? 1 : 0) : (d.Wo < 16 ? 1 : 0);
    const int TW = 16 >> sh;
    // dual: 8x8 maps with the 8-row tile shape — one tile = TWO whole images (wave row wm = image), so the weight tile
    // streamed per K-step serves 128 pixels instead of 64 at the weight-bound 8x8 layers
    const bool dual = sh && TM == 4;
    const int tiles_x = dual ? 1 : d.Wo / TW, tiles_y = dual ? 1 : d.Ho / (TH << sh);
    const int n_tiles = (d.Cout + BN - 1) / BN;
    const int m_tiles = dual ? d.N / 2 : d.N * tiles_y * tiles_x;
    const int nblk = n_tiles * m_tiles;
    int bid = blockIdx.x;
    {   // XCD-aware remap: blocks b, b+8, ... share an XCD (and its L2); give each XCD a contiguous tile range
        const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + idx;
    }
    // Which index runs fastest inside an XCD's contiguous range decides what its L2 shares: n fastest (the default) keeps one pixel
    // tile's halo resident while its N tiles stream the weights; m fastest keeps one WEIGHT tile resident while the pixel tiles walk
    // past it — the right order where the weights are the bigger operand (the 8x8 / 16x16 levels: 29.5 MB of weights against 5-21 MB
    // of activations; with n fastest every XCD streamed all of them, 8 x 29.5 MB per launch).  The launcher decides (`tile_order`).
    const int tile_n = m_fastest ? bid / m_tiles : bid % n_tiles;
    int tile_m = m_fastest ? bid - tile_n * m_tiles : bid / n_tiles;
    const int n_img0 = dual ? 2 * tile_m : tile_m / (tiles_y * tiles_x);
    tile_m = dual ? 0 : tile_m - n_img0 * tiles_y * tiles_x;
    const int n_img = dual ? n_img0 + wm : n_img0;              // image of this wave's output rows
    const int oy0 = (tile_m / tiles_x) * (TH << sh), ox0 = (tile_m % tiles_x) * TW;
    const int fdx = fr & (TW - 1), fdy = fr >> (4 - sh);
    const int n0 = tile_n * BN;

    int c_begin = 0, c_end = nchunks;
    if (d.splitk > 1) {
        const int per = (nchunks + d.splitk - 1) / d.splitk;
        c_begin = blockIdx.y * per;
        c_end = min(nchunks, c_begin + per);
        if (c_begin >= c_end) return;
    }

    // ---- halo geometry (input coordinates).  upsample: output tile lives on the 2x grid, halo on the input grid
    const int up = FAST ? (UPS ? 1 : 0) : d.upsample;
    const int HWd = up ? 10 : TW + 2;
    const int HHt = up ? TH / 2 + 2 : (dual ? 10 : (TH << sh) + 2);
    const int himg = HHt * HWd;                                 // halo pixels per image (dual: two images back to back)
    const int iy0 = up ? (oy0 >> 1) - 1 : oy0 - 1;
    const int ix0 = up ? (ox0 >> 1) - 1 : ox0 - 1;
    int h_pix[NHU];                                   // input pixel index of each staged unit, -1 = zero padding, -2 = unused
    int h_lds[NHU];
#pragma unroll
    for (int i = 0; i < NHU; ++i) {
        const int u = tid + 256 * i;
        const int hp = u >> 3;
        int pix = -2;
        if (hp < (dual ? 2 : 1) * himg) {
            const int im = hp >= himg ? 1 : 0;
            const int hq = hp - im * himg;
            const int hy = hq / HWd, hx = hq - hy * HWd;
            const int iy = iy0 + hy, ix = ix0 + hx;
            pix = (iy >= 0 && iy < d.H && ix >= 0 && ix < d.W) ? ((n_img0 + im) * d.H + iy) * d.W + ix : -1;
        }
        h_pix[i] = pix;
        h_lds[i] = FAST ? hp * HP + (q << 4) : hp * 128 + ((q ^ (hp & 7)) << 4);
    }
    // weight-tile DMA: piece g (8 rows) of a stage; lane -> row 8g + (lane>>3), LDS slot lane&7 = source chunk ^ (row&7)
    const char* b_src[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = (wave + 4 * i) * 8 + (lane >> 3);
        int n = n0 + row;
        n = n < d.Cout ? n : d.Cout - 1;              // rows past Cout are computed and discarded
        b_src[i] = (const char*)d.w + ((long long)n * 9 * Cin + (((lane & 7) ^ (row & 7)) << 3)) * 2;
    }
    const bf16_t* __restrict__ x1 = (const bf16_t*)d.x1;
    const bf16_t* __restrict__ x2 = (const bf16_t*)d.x2;
    const bf16_t* __restrict__ wgt = (const bf16_t*)d.w;
    const int gn_row = GN ? (n_img0 % d.gn_batch) : 0;          // (dual tiles are never launched with a fused GN)

    u32x4 rh[NHU];
    auto issue_halo = [&](int cc) {
        const int c = cc * 64 + q * 8;
        const bool second = c >= d.C1;
        const bf16_t* __restrict__ src = second ? x2 : x1;
        const int cs = second ? d.C2 : d.C1;
        const int co = second ? c - d.C1 : c;
#pragma unroll
        for (int i = 0; i < NHU; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (FAST) v = *(const u32x4*)(src + (long long)max(h_pix[i], 0) * cs + co);   // always NHU loads (counted vmcnt)
            else if (h_pix[i] >= 0) v = *(const u32x4*)(src + (long long)h_pix[i] * cs + co);
            rh[i] = v;
        }
    };
    auto store_halo = [&](int cc) {
        f32x4 g[4];
        if (GN) {
            const float* __restrict__ abp = d.gn_ab + ((long long)gn_row * Cin + cc * 64 + q * 8) * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = *(const f32x4*)(abp + 4 * j);
        }
#pragma unroll
        for (int i = 0; i < NHU; ++i) {
            if (h_pix[i] == -2) continue;
            u32x4 v = rh[i];
            if (FAST && h_pix[i] < 0) v = u32x4{0u, 0u, 0u, 0u};
            if (GN && h_pix[i] >= 0) {
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float lo = __uint_as_float(v[j] << 16) * g[j][0] + g[j][1];
                    float hi = __uint_as_float(v[j] & 0xffff0000u) * g[j][2] + g[j][3];
                    if (d.gn_silu) {
                        lo = dc_silu(lo);
                        hi = dc_silu(hi);
                    }
                    bf16x2 p = {(bf16_t)lo, (bf16_t)hi};
                    o[j] = *(uint32_t*)&p;
                }
                v = u32x4{o[0], o[1], o[2], o[3]};
            }
            *(u32x4*)(sH + h_lds[i]) = v;
        }
    };
    const int last_step = (c_end - c_begin) * 9 - 1;
#ifndef DC_EXP_NO_DMA
#define DC_EXP_NO_DMA 0         // developer experiment (wrong results): 1 = weight stages are issued in the prologue only — the
#endif                          // K loop then runs without any DMA traffic or latency: an upper bound for pipelining changes
    auto issue_b = [&](int step, int slot) {          // step = (cc - c_begin) * 9 + tap ; past-the-end re-reads the last
        if (DC_EXP_NO_DMA && step >= NSTB - 1) return;
        step = step < last_step ? step : last_step;
        const int cc = c_begin + step / 9, tap = step % 9;
        const long long off = ((long long)tap * Cin + cc * 64) * 2;
        char* base = sB0 + slot * B_BYTES;
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + off), (lptr_t)(base + (wave + 4 * i) * 1024), 16, 0, 0);
    };

    auto issue_b_at = [&](int cc, int tap, int slot) {   // the same with (slice, tap) known: no division in the unrolled K loop
        if (DC_EXP_NO_DMA) return;
        const bool past = cc >= c_end;
        cc = past ? c_end - 1 : cc;
        tap = past ? 8 : tap;
        const long long off = ((long long)tap * Cin + cc * 64) * 2;
        char* base = sB0 + slot * B_BYTES;
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + off), (lptr_t)(base + (wave + 4 * i) * 1024), 16, 0, 0);
    };

    // FAST specialised epilogues: the bias (+ the sample's time-embedding row) is the accumulators' initial value — fetched in
    // the prologue beside the halo, consumed at once (no registers live across the K loop), and the epilogue has no operand
    // fetch left in front of its arithmetic (stamps: 7k of the epilogue's 10k cycles were those dependent L2 round trips)
    constexpr bool BIAS_INIT = FAST && !GENERIC;
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BIAS_INIT) {
            const int nb = n0 + (wn * TN + i) * 16 + 4 * fq;
            if (nb < d.Cout) {
                if (d.bias) v0 = *(const f32x4*)(d.bias + nb);
                if (d.row_add) v0 += *(const f32x4*)(d.row_add + (long long)n_img * d.row_add_stride + nb);
            }
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = v0;
    }

    // One K-step (tap): all 2 x (TN + TM) fragment reads first (two register sets), then the step's LDS-DMA pieces, which
    // the schedule spreads between the MFMAs — same reasoning as gemm_dma.hip's K-step.
    bf16x8 wf[2][TN], xf[2][TM];
    auto load_frags = [&](int tap, int buf) {
        const char* sB = sB0 + buf * B_BYTES;
        const int ky = tap / 3, kx = tap - 3 * ky;
        int prow[TM];                                   // halo pixel feeding this lane's output pixel, per m-tile
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int ty = wm * TM + tm;
            prow[tm] = up ? (((ty + ky - 1) >> 1) + 1) * 10 + ((fr + kx - 1) >> 1) + 1
                          : (dual ? wm * himg + ((tm << 1) + fdy + ky) * HWd + fdx + kx
                                  : ((ty << sh) + fdy + ky) * HWd + fdx + kx);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int qq = 4 * s + fq;
            const int bsw = (qq ^ (fr & 7)) << 4;       // weight rows: (row & 7) == (fr & 7)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                xf[s][tm] = *(const bf16x8*)(sH + prow[tm] * 128 + ((qq ^ (prow[tm] & 7)) << 4));
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                wf[s][tn] = *(const bf16x8*)(sB + ((wn * TN + tn) * 16 + fr) * 128 + bsw);
        }
    };
    auto mfma_frags = [&]() {
#ifdef DC_EXP_PRIO
        __builtin_amdgcn_s_setprio(DC_EXP_PRIO);
#endif
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][tn], xf[s][tm], acc[tn][tm], 0, 0, 0);
#ifdef DC_EXP_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if (DC_CONV_PIPE) {
            constexpr int NMF = 2 * TN * TM;
            constexpr int PER = NMF / (NB + 1) > 0 ? NMF / (NB + 1) : 1;
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TN + TM), 0);          // every fragment read first
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);                // a few MFMAs ...
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                  // ... then one LDS-DMA piece
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMF - NB * PER, 0);
        }
    };

    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ta = 0, tb = 0, tc = 0, s_sync = 0, s_work = 0, s_halo = 0;
    DC_NOW(ts0);
    if constexpr (FAST) {
        // ---- FAST main loop (see the kernel comment): half-step pipeline, taps unrolled, mid-step barrier
        // this lane's halo byte offset at tap (0,0), tm 0, half 0.  Upsample: output pixel (ty, x) reads low-resolution pixel
        // (((ty + ky - 1) >> 1) + 1, ((x + kx - 1) >> 1) + 1) of the 10-wide halo: three column variants, rows stay immediates
        int a_lane[UPS ? 3 : 1];
        if constexpr (UPS) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) a_lane[kx] = ((wm * TM / 2) * 10 + ((fr + kx - 1) >> 1) + 1) * HP + (fq << 4);
        } else if constexpr (SH) {
            // narrow maps: lane fr -> pixel (fr >> 3, fr & 7) of the 10-wide halo; 8-row tiles: wave row wm = the second image
            a_lane[0] = ((TM == 4 ? wm * 100 : (2 * wm * TM) * 10) + (fr >> 3) * 10 + (fr & 7)) * HP + (fq << 4);
        } else {
            a_lane[0] = ((wm * TM) * 18 + fr) * HP + (fq << 4);
        }
        int b_lane[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) b_lane[s] = H_BYTES + ((wn * TN) * 16 + fr) * 128 + (((4 * s + fq) ^ (fr & 7)) << 4);
        bf16x8 xa[2][TM], wb[2][TN];
        auto rd_half = [&](int tap, int s, int slot_off) {
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                if constexpr (UPS) xa[s][tm] = *(const bf16x8*)(sH + a_lane[kx] + ((((tm + ky + 1) >> 1)) * 10) * HP + s * 64);   // ((tm+ky-1)>>1)+1
                else if constexpr (SH) xa[s][tm] = *(const bf16x8*)(sH + a_lane[0] + ((2 * tm + ky) * 10 + kx) * HP + s * 64);
                else xa[s][tm] = *(const bf16x8*)(sH + a_lane[0] + ((tm + ky) * 18 + kx) * HP + s * 64);
            }
            const char* bp = smem + b_lane[s] + slot_off;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) wb[s][tn] = *(const bf16x8*)(bp + tn * 2048);
        };
        auto mfma_half = [&](int s) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[s][tn], xa[s][tm], acc[tn][tm], 0, 0, 0);
        };
        constexpr int NMF = TN * TM, NRD = TN + TM;
        issue_halo(c_begin);
        issue_b(0, 0);
        store_halo(c_begin);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        rd_half(0, 0, 0);
#pragma unroll
        for (int st = 1; st < NSTB; ++st) issue_b(st, st);
        int step = 0;
        DC_NOW(ts1);
        for (int cc = c_begin; cc < c_end; ++cc) {
            const bool more_c = cc + 1 < c_end;
            dc_static_for(std::make_integer_sequence<int, 9>{}, [&](auto tap_c) {
                constexpr int tap = decltype(tap_c)::value;
                const int slot_off = (step & (NSTB - 1)) * B_BYTES;
                DC_NOW(ta);
                // first half: MFMAs on half 0 (read during the previous step), reads of half 1 in flight beside them
                rd_half(tap, 1, slot_off);
                mfma_half(0);
                // MFMAs lead each group: the wait in front of the first one then covers only reads issued a half-step ago
#pragma unroll
                for (int i = 0; i < NRD; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, NMF / NRD, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD * (NMF / NRD), 0);
                // mid-step hand-over: this wave's pieces of stage step+1 have landed (the halo prefetch issued after them in
                // tap 0 may still be in flight at tap 1), its reads of stage `step` have returned; after the barrier
                // everyone's have: stage step+1 visible, slot of stage `step` free for stage step+2
                DC_NOW(tb);
                __builtin_amdgcn_sched_barrier(0);
                // vmcnt(N) lgkmcnt(0) as a builtin, so that hipcc's own wait-count bookkeeping sees the drain.  N = the pieces of the
                // NSTB - 2 younger stages, plus the halo prefetch while it is younger than stage step+1 (taps 1 .. NSTB-1)
                constexpr int VMN = NB * (NSTB - 2) + ((tap >= 1 && tap <= NSTB - 1) ? NHU : 0);
                static_assert(VMN < 64, "vmcnt field");
                __builtin_amdgcn_s_waitcnt(0x0070 | (VMN & 15) | ((VMN >> 4) << 14));
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                DC_NOW(tc);
#ifdef DC_STAMP
                s_work += tb - ta;
                s_sync += tc - tb;
#endif
                // second half: MFMAs on half 1; reads of the next step's half 0, the DMA of stage step+2 (and in tap 0 the
                // global loads of the next channel slice's halo, always issued so that the counted wait of tap 1 holds)
                if (tap < 8) rd_half(tap + 1, 0, ((step + 1) & (NSTB - 1)) * B_BYTES);
                issue_b_at(cc + (tap + NSTB) / 9, (tap + NSTB) % 9, step & (NSTB - 1));
                if (tap == 0) issue_halo(more_c ? cc + 1 : cc);
                mfma_half(1);
                {
                    constexpr int nrd2 = tap < 8 ? NRD : 0, nvm = tap == 0 ? NB + NHU : NB;
                    constexpr int per = (NMF - nrd2) / nvm > 0 ? (NMF - nrd2) / nvm : 1;
#pragma unroll
                    for (int i = 0; i < nrd2; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
#pragma unroll
                    for (int i = 0; i < nvm; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, per, 0);
                        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    }
                    constexpr int rest = NMF - nrd2 - nvm * per;
                    if constexpr (rest > 0) __builtin_amdgcn_sched_group_barrier(0x008, rest, 0);
                }
                __builtin_amdgcn_sched_barrier(0);       // (a group left open would take the next half-step's MFMAs)
#ifdef DC_STAMP
                DC_NOW(ta);
                s_work += ta - tc;
#endif
                ++step;
            });
            if (more_c) {
                // every wave passed the mid-step barrier of tap 8 with its halo reads returned: the image is free
                DC_NOW(ta);
                store_halo(cc + 1);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                rd_half(0, 0, (step & (NSTB - 1)) * B_BYTES);
#ifdef DC_STAMP
                DC_NOW(tb);
                s_halo += tb - ta;
#endif
            }
        }
    } else {
    // ---- main loop: per 64-channel slice, halo once, 9 taps; weight tiles ride the DMA ring
    issue_halo(c_begin);
#pragma unroll
    for (int st = 0; st < NSTB - 1; ++st) issue_b(st, st);
    store_halo(c_begin);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int step = 0;
    DC_NOW(ts1);
    for (int cc = c_begin; cc < c_end; ++cc) {
        const bool more_c = cc + 1 < c_end;
        for (int tap = 0; tap < 9; ++tap, ++step) {
            DC_NOW(ta);
            dc_ring_sync<DC_EXP_NO_DMA ? 0 : NB * (NSTB - 2)>(); // this wave's pieces of weight stage `step` have landed and its reads of
                                                                 // step-1 have returned; after the barrier everyone's have: halo image
                                                                 // visible, slot step-1 free
            DC_NOW(tb);
            load_frags(tap, step % NSTB);
            if (tap == 0 && more_c) issue_halo(cc + 1);          // lands under the next eight K-steps
            issue_b(step + NSTB - 1, (step + NSTB - 1) % NSTB);
            mfma_frags();
#ifdef DC_STAMP
            __builtin_amdgcn_sched_barrier(0);
            DC_NOW(tc);
            s_sync += tb - ta;
            s_work += tc - tb;
#endif
        }
        if (more_c) {
            DC_NOW(ta);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the halo image have returned
            __builtin_amdgcn_s_barrier();                        // every wave is past its last read of the halo image
            asm volatile("" ::: "memory");
            store_halo(cc + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // visible after the next K-step's barrier
#ifdef DC_STAMP
            DC_NOW(tb);
            s_halo += tb - ta;
#endif
        }
    }
    }
    wait_vmcnt<0>();
    DC_NOW(ts2);

    const long long slab = (long long)d.N * d.Ho * d.Wo * d.Cout;   // elements per split-K slab
    // ---- epilogue: lane holds out[pixel (ty, fr)][n = .. + 4*fq + 0..3].  Bias / time-embedding row / residual are all
    //      fetched before the arithmetic so the loads overlap instead of forming a chain of dependent L2 round trips.
    f32x4 bv[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int nb = n0 + (wn * TN + tn) * 16 + 4 * fq;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!BIAS_INIT && !e_split && nb < d.Cout) {
            if (d.bias) v = *(const f32x4*)(d.bias + nb);
            if (d.row_add) v += *(const f32x4*)(d.row_add + (long long)n_img * d.row_add_stride + nb);
        }
        bv[tn] = v;
    }
    long long mrow[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int oy = dual ? (tm << 1) + fdy : oy0 + ((wm * TM + tm) << sh) + fdy, ox = ox0 + fdx;
        mrow[tm] = ((long long)n_img * d.Ho + oy) * d.Wo + ox;
    }
    bf16x4 rr[TM][TN];
    if (e_res) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int nb = n0 + (wn * TN + tn) * 16 + 4 * fq;
                rr[tm][tn] = nb < d.Cout ? *(const bf16x4*)((const bf16_t*)d.residual + mrow[tm] * d.Cout + nb)
                                         : bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
            }
    }
    // FAST tiles (16 consecutive pixels per tile row) leave through LDS: the finished bf16 tile is staged as [pixel][BN] rows and
    // written out in 16-byte pieces, 20 lanes per 320-byte pixel row, instead of 8-byte pieces scattered over 16 rows per
    // store instruction (the epilogue was store-issue-bound: 15k cycles per workgroup against 1.2k per K-step).
    constexpr int SP = BN * 2 + 16;                              // staged row pitch (bytes)
    constexpr bool stg = FAST && !GENERIC && !SH;                     // (the generic epilogue keeps its direct stores; the launcher
                                                                 //  sends Cout % 8 != 0 to the other kernel)
    if constexpr (stg) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every wave is past its last fragment read: LDS is free
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
#ifdef DC_STAMP
    unsigned long long te1 = 0, te2 = 0, te3 = 0;
    DC_NOW(te1);
#endif
    const bool want_gn = d.gn_part_out != nullptr && !e_split;
    f32x4 gs[TN], gq[TN];                                       // GroupNorm partials of this wave's output (gn_part_out)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) gs[tn] = gq[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int nb = n0 + (wn * TN + tn) * 16 + 4 * fq;
            if (!stg && nb >= d.Cout) continue;                   // (staged: columns past Cout are dropped by the row pass)
            f32x4 v = acc[tn][tm];
            const long long off = mrow[tm] * d.Cout + nb;
            if (e_split) {                                        // this split's own fp32 slab: plain stores, no atomics
                *(f32x4*)(d.splitk_ws + (long long)blockIdx.y * slab + off) = v;
                continue;
            }
            if constexpr (!BIAS_INIT) v += bv[tn];
            if (e_act) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = dc_act(v[r], e_act);
            }
            if (e_res) v = dc_scale_res(v, d.out_scale, rr[tm][tn]);
            else v *= d.out_scale;
            if (want_gn) {
                gs[tn] += v;
                gq[tn] += v * v;
            }
            if (e_f32) {
                *(f32x4*)((float*)d.out + off) = v;
            } else {
                bf16x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)v[r];
                if constexpr (stg) *(bf16x4*)(smem + ((wm * TM + tm) * 16 + fr) * SP + ((((wn * TN + tn) * 16 + 4 * fq) * 2) ^ dc_stage_swz(fr))) = pk;
                else *(bf16x4*)((bf16_t*)d.out + off) = pk;
            }
        }
        // one pixel row group at a time: without the fence the straight-line epilogue is scheduled as one block and every
        // tile's operands are live at once (1,700 spilled registers in the specialised kernels)
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (stg) {
        constexpr int CPR = BN / 8, NCH = TH * 16 * CPR / 256;   // 16-byte pieces per row / per thread
        static_assert(TH * 16 * CPR % 256 == 0, "whole pieces per thread");
        __builtin_amdgcn_sched_barrier(0);
        DC_NOW(te2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        DC_NOW(te3);
        bf16_t* const obase = (bf16_t*)d.out + (((long long)n_img0 * d.Ho + oy0) * d.Wo + ox0) * d.Cout + n0;   // block-uniform
        u32x4 pv[NCH];
        int voff[NCH];                                           // element offset inside the tile's rows (small: 32-bit)
        int row = tid / CPR, col = tid - row * CPR;              // piece tid + 256 i, advanced incrementally
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            pv[i] = dc_stage_unswz(*(const u32x4*)(smem + row * SP + col * 16), row);
            voff[i] = n0 + col * 8 < d.Cout ? ((row >> 4) * d.Wo + (row & 15)) * d.Cout + col * 8 : -1;
            col += 256 % CPR;
            row += 256 / CPR;
            if (col >= CPR) {
                col -= CPR;
                ++row;
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if (voff[i] >= 0) *(u32x4*)(obase + voff[i]) = pv[i];
    }
    if (want_gn) {
        // chunk = (pixel tile within the sample, wave row); dual tiles: one wave row = one whole 8x8 image
        const int chunk = dual ? 0 : tile_m * 2 + wm;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int nb = n0 + (wn * TN + tn) * 16 + 4 * fq;
            dc_gn_partial_store(gs[tn], gq[tn], d.gn_part_out + (((long long)chunk * d.N + n_img) * d.Cout + nb) * 2,
                                fr == 0 && nb < d.Cout);
        }
    }
#ifdef DC_STAMP
    DC_NOW(ts3);
    if (lane == 0 && d.splitk_ws && d.splitk <= 1) {
        unsigned long long* o = (unsigned long long*)d.splitk_ws + ((long long)blockIdx.x * 4 + wave) * 8;
        o[0] = ts0, o[1] = ts1, o[2] = ts2, o[3] = ts3, o[4] = s_sync, o[5] = s_work, o[6] = s_halo;
        o[7] = ((te1 - ts2) & 0xffff) | (((te2 - te1) & 0xffff) << 16) | (((te3 - te2) & 0xffff) << 32) | (((ts3 - te3) & 0xffff) << 48);
    }
#endif
}

template <int TM, int TN, int NSTB, bool FAST = false, bool UPS = false, bool SH = false>
int launch_tile(const dc_conv_desc& d, hipStream_t st)
{
    constexpr int TH = 2 * TM, BN = 2 * TN * 16;
    const int sh = d.Wo < 16 ? 1 : 0;
    const bool dual = sh && TM == 4;
    const int nblk = (dual ? d.N / 2 : d.N * (d.Ho / (TH << sh)) * (d.Wo / (16 >> sh))) * dc_cdiv(d.Cout, BN);
    constexpr int HALO_ROWS = UPS ? (TH / 2 + 2) * 10 : SH ? (TM == 4 ? 200 : ((TH << 1) + 2) * 10) : ((TM == 4 && !FAST) ? 200 : (TH + 2) * 18);
    const dim3 grid(nblk, d.splitk > 1 ? d.splitk : 1);
#ifdef DC_EXP_ONE_WG            // developer experiment: pad the allocation so one workgroup owns the CU
    const size_t lds = 96 * 1024;
#else
    const size_t lds = HALO_ROWS * (FAST ? (SH ? DC_SH_HP : 160) : 128) + NSTB * BN * 128;
#endif
    const int epi = (!DC_EPI_SPECIALIZE || d.splitk > 1 || d.out_f32 || d.act) ? 0 : (d.residual ? 2 : 1);
    // tile order inside an XCD's range (see the kernel): pixel tiles fastest when the weight tensor is larger than the activations
    static const int force_order = DC_KNOB("DC_CONV_ORDER", -1);   // developer A/B knob: 0 = n fastest, 1 = m fastest
    const int order = force_order >= 0 ? force_order : ((long long)d.Cout * 9 > (long long)d.N * d.H * d.W ? 1 : 0);
#define DC_TILE_LAUNCH1(GN, EPI)                                                                                \
    do {                                                                                                        \
        auto kern = conv3x3_tile_kernel<TM, TN, GN, NSTB, EPI, FAST, (UPS && !GN), (SH && !GN)>;                                                    \
        static std::atomic<unsigned long long> attr_done{0};                                                    \
        dc_set_max_dyn_lds((const void*)kern, (int)lds, attr_done);                                             \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, d, order);                                           \
    } while (0)
#define DC_TILE_LAUNCH(GN)                          \
    do {                                            \
        if (epi == 1) DC_TILE_LAUNCH1(GN, 1);       \
        else if (epi == 2) DC_TILE_LAUNCH1(GN, 2);  \
        else DC_TILE_LAUNCH1(GN, 0);                \
    } while (0)
    if constexpr (UPS || SH) {
        DC_TILE_LAUNCH(false);                                  // (the dispatcher sends GroupNorm-on-load upsamples to the other form)
    } else {
        if (d.gn_ab) DC_TILE_LAUNCH(true);
        else DC_TILE_LAUNCH(false);
    }
#undef DC_TILE_LAUNCH
#undef DC_TILE_LAUNCH1
    return dc_launch_status();
}

}  // namespace

// Returns 1 if the halo-tile kernel can take this descriptor (3x3, stride 1, pad 1, tile-aligned output).
int dc_conv3x3_tile_supported(const dc_conv_desc& d)
{
    if (!(d.ksize == 3 && d.stride == 1 && d.pad == 1 && d.epilogue == 0)) return 0;
    if (d.Wo == 8) return !d.upsample && (d.Ho % 8) == 0;       // narrow: tile = 8 rows x 8 cols (TM = 2)
    return (d.Wo % 16) == 0 && (d.Ho % 4) == 0;
}

// Tile-shape decision shared by the launcher and the statistics-chunk query: 4 = 8-row tiles (TM = 4), 2 = 4-row tiles,
// 8 = two whole 8x8 images per tile (TM = 4, dual).
// N tile: 160 columns when Cout is a multiple of 160 (all SD-1.5 UNet widths).  DC_CONV_BN128=1 (developer A/B knob) takes the
// 128-column tile instead wherever Cout is also a multiple of 128 (640, 1280): it affords a 3-stage weight ring at two
// workgroups per CU where the 160-column tile has room for two stages only.
static bool use_n160(const dc_conv_desc& d)
{
    static const int force128 = DC_KNOB("DC_CONV_BN128", 0);
    return d.Cout % 160 == 0 && !(force128 && d.Cout % 128 == 0);
}

static int tile_variant(const dc_conv_desc& d)
{
    const bool n160 = use_n160(d);
    const int bn = n160 ? 160 : 128;
    const long long big = (long long)d.N * (d.Ho / 8) * (d.Wo / 16) * dc_cdiv(d.Cout, bn) * (d.splitk > 1 ? d.splitk : 1);
    // LDS budget keeps two workgroups per CU: 8-row tile with BN=160 affords a 2-stage weight ring, the others 3 stages
    if (d.Wo >= 16 && (d.Ho % 8) == 0 && big >= 512) return 4;
    // 8x8 maps: two images per tile when the batch is even, nothing is fused on load and enough tiles remain
    if (d.Wo == 8 && d.Ho == 8 && (d.N & 1) == 0 && !d.gn_ab &&
        (long long)(d.N / 2) * dc_cdiv(d.Cout, bn) * (d.splitk > 1 ? d.splitk : 1) >= 512)   // measured: below 2 workgroups/CU the 4-row tiles win
        return 8;
    return 2;
}

// gn_part_out chunks per sample of this launch (0: not available — split-K partial tiles are finished elsewhere).
int dc_conv3x3_tile_gn_chunks(const dc_conv_desc& d)
{
    if (d.splitk > 1) return 0;
    const int v = tile_variant(d);
    if (v == 8) return 1;
    const int sh = d.Wo < 16 ? 1 : 0;
    const int th = (v == 4 ? 8 : 4) << sh, tw = 16 >> sh;
    return (d.Ho / th) * (d.Wo / tw) * 2;
}

// Called by dc_conv_igemm_bf16 after validation (workspace already zeroed for splitk > 1).
int dc_conv3x3_tile_launch(const dc_conv_desc& d, hipStream_t st)
{
    const bool n160 = use_n160(d);
    const int v = tile_variant(d);
    if (DC_CONV_FAST && !d.upsample && d.Wo >= 16 && (d.Cout & 7) == 0) {            // plain maps: half-step pipeline (see the kernel comment)
        // at most one workgroup per CU (one- or two-frame decodes): a four-slot weight ring, three taps of weights in flight
        static const int deep = DC_KNOB("DC_CONV_DEEP", 1);      // developer A/B knob
        const int th = v == 4 ? 8 : 4, bn = n160 ? 160 : 128;
        const long long wgs = (long long)d.N * (d.Ho / th) * (d.Wo / 16) * dc_cdiv(d.Cout, bn) * (d.splitk > 1 ? d.splitk : 1);
        // Cout <= 32 (UNet conv_out 320 -> 4, VAE conv_out 128 -> 3 + 1): one 32-column N tile instead of a mostly empty 128-column
        // one — a fifth of the MFMA work per pixel tile (the launch is then paced by the GroupNorm+SiLU of its halo, done once)
        if (d.Cout <= 32) return v == 4 ? launch_tile<4, 1, 2, true>(d, st) : launch_tile<2, 1, 2, true>(d, st);
        if (deep && wgs <= 256) {
            if (v == 4) return n160 ? launch_tile<4, 5, 4, true>(d, st) : launch_tile<4, 4, 4, true>(d, st);
            return n160 ? launch_tile<2, 5, 4, true>(d, st) : launch_tile<2, 4, 4, true>(d, st);
        }
        if (v == 4) return n160 ? launch_tile<4, 5, 2, true>(d, st) : launch_tile<4, 4, 2, true>(d, st);
        return n160 ? launch_tile<2, 5, 2, true>(d, st) : launch_tile<2, 4, 2, true>(d, st);
    }
    if (DC_CONV_FAST && d.upsample && !d.gn_ab && d.Wo >= 16 && (d.Cout & 7) == 0) {       // Upsample2D convs: the same pipeline
        if (v == 4) return n160 ? launch_tile<4, 5, 2, true, true>(d, st) : launch_tile<4, 4, 2, true, true>(d, st);
        return n160 ? launch_tile<2, 5, 2, true, true>(d, st) : launch_tile<2, 4, 2, true, true>(d, st);
    }
    if (DC_CONV_FAST && d.Wo == 8 && !d.upsample && !d.gn_ab && (d.Cout & 7) == 0) {        // 8x8 maps: the same pipeline
        if (v == 8) return n160 ? launch_tile<4, 5, 2, true, false, true>(d, st) : launch_tile<4, 4, 2, true, false, true>(d, st);
        return n160 ? launch_tile<2, 5, 2, true, false, true>(d, st) : launch_tile<2, 4, 2, true, false, true>(d, st);
    }
    if (d.Cout <= 32 && v != 8) return v == 4 ? launch_tile<4, 1, 3>(d, st) : launch_tile<2, 1, 3>(d, st);   // (Cout = 4: not a multiple of 8)
    if (v == 4 || v == 8) return n160 ? launch_tile<4, 5, 2>(d, st) : launch_tile<4, 4, 3>(d, st);
    return n160 ? launch_tile<2, 5, 3>(d, st) : launch_tile<2, 4, 3>(d, st);
}
