// EXPERIMENT — NOT BUILT, NOT SHIPPED (tools/experiments/; build.py does not list this file).
// Kept as the starting point for next round's GEMM work: parity-green (bit-identical to gemm_dma.hip without a residual),
// on par with it on time.  To try it: copy to <package>/csrc/gemm_wide.hip, add it to build.py's
// SOURCES, declare dc_gemm_wide_supported / dc_gemm_wide_launch in igemm.hip and call them ahead of dc_gemm_dma_launch;
// include paths below assume the csrc/ location.
//
// Persistent wide-tile variant of the LDS-DMA GEMM (gemm_dma.hip) for the large-M, short-K 1x1 convs / linears of the
// transformer blocks (attention projections, proj_out, conv_shortcut: call sites flownet.py:87-124, pipeline.py:358-367).
//
// Why: the 128x160 kernel pulls 288 cache lines per workgroup K-step through the CU's vector-memory path and sits at
// that path's line rate; and for K = 320..640 its prologue (first-stage latency) and epilogue are 60 % of a workgroup's
// life.  Here ONE persistent workgroup of 8 waves per CU walks over 256x160 tiles (waves 4(m) x 2(n), wave tile 64x80:
// 28 % fewer lines per FLOP) with a three-stage LDS ring that never drains: while a tile's epilogue runs, the first two
// K-stages of the NEXT tile are already landing in the two ring slots the epilogue does not use, so every K-loop after
// the first starts on resident data.
//
// Data path as gemm_dma.hip: both operands by `global_load_lds_dwordx4`, 128-byte rows XOR-swizzled on the source
// address, one raw s_barrier per K-step.  Epilogue: bias / activation / scale in registers, tile staged through the
// free ring slot in two 128-row halves, rows leave as whole 16-byte pieces; the residual is added at that point (bf16
// result + bf16 residual: one more rounding than gemm_dma.hip — the order PyTorch's bf16 modules use).
#include "dc_common.h"
#include "../../include/diffcodec_hip.h"

namespace {

typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

constexpr int WM = 4, WN = 2, TM = 4, TN = 5, NST = 3;
constexpr int BM = WM * TM * 16, BN = WN * TN * 16;     // 256 x 160
constexpr int ROWS = BM + BN;                           // 416 rows of 128 B per stage
constexpr int STAGE = ROWS * 128;                       // 53,248 B; three stages = 159,744 B (one workgroup per CU)
constexpr int NT = 64 * WM * WN;                        // 512 threads
constexpr int NPIECES = ROWS / 8;                       // 52 DMA pieces (8 rows each) per stage
constexpr int NPW = (NPIECES + WM * WN - 1) / (WM * WN);  // 7 per wave; the 4 surplus slots re-issue the wave's previous piece
constexpr int NPA = BM / 8 / (WM * WN);                 // the first 4 of a wave's pieces are activation rows
constexpr int PITCH = BN * 2 + 16;                      // bytes per staged output row (+16: spreads rows over banks)
static_assert(BM % (8 * WM * WN) == 0, "activation pieces must split evenly over the waves");
static_assert(128 * PITCH <= STAGE, "a 128-row half tile must fit in one ring slot");
static_assert((128 * (BN / 8)) % NT == 0, "row store must be the same trip count for every thread");

struct TileOff {           // per-lane byte offsets of this wave's DMA pieces for one tile
    unsigned a1[NPA], a2[NPA], b[NPW - NPA];
};

__global__ __launch_bounds__(NT, 1) void gemm_wide_kernel(const dc_conv_desc d)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & (WM - 1), wn = wave >> 2;
    const int fr = lane & 15, fq = lane >> 4;

    const int M = d.N * d.Ho * d.Wo;                          // multiple of BM (checked by the launcher)
    const int K = d.C1 + d.C2;
    const int KT = K >> 6;                                    // >= 2
    const int n_tiles = d.Cout / BN;
    const int ntiles = n_tiles * (M / BM);
    const int c1_steps = d.C1 >> 6;

    // piece g = wave + 8 i covers stage rows [8g, 8g+8); lane s -> row 8g + (s>>3), LDS slot s&7 holding source chunk
    // (s&7) ^ (row&7).  32-bit byte offsets from the tensor bases (all operands < 4 GB, checked by the launcher).
    auto tile_offsets = [&](int t, TileOff& o) {
        const int m0 = (t / n_tiles) * BM, n0 = (t % n_tiles) * BN;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            int g = wave + WM * WN * i;
            g = g < NPIECES ? g : g - WM * WN;                // surplus slot: same piece again (same bytes to the same place)
            const int row = g * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ (row & 7);
            if (i < NPA) {
                o.a1[i] = (unsigned)(((long long)(m0 + row) * d.C1 + chunk * 8) * 2);
                o.a2[i] = (unsigned)(((long long)(m0 + row) * d.C2 + chunk * 8) * 2);
            } else {
                o.b[i - NPA] = (unsigned)(((long long)(n0 + row - BM) * K + chunk * 8) * 2);
            }
        }
    };
    auto issue_stage = [&](const TileOff& o, int kt, int slot) {
        char* base = smem + slot * STAGE;
        const bool second = kt >= c1_steps;                   // K range of x2 (channel concat read in place)
        const char* abase = second ? (const char*)d.x2 : (const char*)d.x1;
        const unsigned akoff = (unsigned)((second ? kt - c1_steps : kt) * 128);
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            int g = wave + WM * WN * i;
            g = g < NPIECES ? g : g - WM * WN;
            const char* p = i < NPA ? abase + ((second ? o.a2[i] : o.a1[i]) + akoff)
                                    : (const char*)d.w + (o.b[i - NPA] + (unsigned)(kt * 128));
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(base + g * 1024), 16, 0, 0);
        }
    };

    const int ocols = BN, pieces = BN / 8;
    bf16_t* __restrict__ out = (bf16_t*)d.out;
    const bf16_t* __restrict__ res = (const bf16_t*)d.residual;

    TileOff cur, nxt;
    int t = blockIdx.x;
    tile_offsets(t, cur);
    int gs = 0;                                               // K-steps done so far: stage k of this tile lives in slot (gs + k) % NST
    issue_stage(cur, 0, 0);
    issue_stage(cur, 1, 1);

    for (; t < ntiles; t += gridDim.x) {
        const int tn = t + gridDim.x;
        const bool has_next = tn < ntiles;
        if (has_next) tile_offsets(tn, nxt);

        f32x4 acc[TN][TM];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int k = 0; k < KT; ++k) {
            // stage k landed?  First step of a tile: everything issued so far (incl. the previous epilogue's stores: CDNA4
            // counts them in vmcnt, and they may retire out of order with the loads).  Later steps: all but the newest stage.
            if (k == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
            __builtin_amdgcn_s_barrier();                     // ... for every wave; everyone is also done reading stage k-1
            const int slot_in = (gs + k + 2) % NST;           // the slot stage k-1 just vacated
            if (k + 2 < KT) issue_stage(cur, k + 2, slot_in);
            else if (has_next) issue_stage(nxt, k + 2 - KT, slot_in);      // the ring runs on into the next tile
            else issue_stage(cur, KT - 1, slot_in);           // last tile: a discarded re-read keeps the vmcnt pattern
            {
                const char* sA = smem + ((gs + k) % NST) * STAGE;
                const char* sB = sA + BM * 128;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int swz = ((4 * s + fq) ^ (fr & 7)) << 4;
                    bf16x8 xf[TM], wf[TN];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) xf[tm] = *(const bf16x8*)(sA + ((wm * TM + tm) * 16 + fr) * 128 + swz);
#pragma unroll
                    for (int q = 0; q < TN; ++q) wf[q] = *(const bf16x8*)(sB + ((wn * TN + q) * 16 + fr) * 128 + swz);
#pragma unroll
                    for (int q = 0; q < TN; ++q)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
                            acc[q][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q], xf[tm], acc[q][tm], 0, 0, 0);
                }
            }
        }
        __syncthreads();      // every wave is done with stage KT-1: its slot is the staging buffer; the other two slots are
                              // receiving the next tile's stages 0 and 1 (or the discarded re-reads)
        char* stg = smem + ((gs + KT - 1) % NST) * STAGE;
        const int m0 = (t / n_tiles) * BM, n0 = (t % n_tiles) * BN;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            if ((wm >> 1) == ph) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const int row = ((wm & 1) * TM + tm) * 16 + fr;
#pragma unroll
                    for (int q = 0; q < TN; ++q) {
                        const int nl = (wn * TN + q) * 16 + 4 * fq;
                        f32x4 v = acc[q][tm];
                        if (d.bias) v += *(const f32x4*)(d.bias + n0 + nl);
                        if (d.act) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = dc_act(v[r], d.act);
                        }
                        if (d.out_scale != 1.0f) v *= d.out_scale;
                        bf16x4 pk;
#pragma unroll
                        for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)v[r];
                        *(bf16x4*)(stg + row * PITCH + nl * 2) = pk;
                    }
                }
            }
            __syncthreads();
            // cooperative store: consecutive lanes -> consecutive 16-byte pieces of one output row (+ residual, same layout)
#pragma unroll
            for (int i = tid; i < 128 * pieces; i += NT) {
                const int row = i / pieces, pc = i - row * pieces;
                const long long off = (long long)(m0 + ph * 128 + row) * d.Cout + n0 + pc * 8;
                u32x4 sv = *(const u32x4*)(stg + row * PITCH + pc * 16);
                if (res) {
                    const u32x4 rv = *(const u32x4*)(res + off);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float lo = __uint_as_float(sv[j] << 16) + __uint_as_float(rv[j] << 16);
                        const float hi = __uint_as_float(sv[j] & 0xffff0000u) + __uint_as_float(rv[j] & 0xffff0000u);
                        bf16x2 p = {(bf16_t)lo, (bf16_t)hi};
                        sv[j] = *(uint32_t*)&p;
                    }
                }
                *(u32x4*)(out + off) = sv;
            }
            __syncthreads();
        }
        gs = (gs + KT) % NST;
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the discarded re-reads of the last tile
    (void)ocols;
}

}  // namespace

// Takes: plain 1x1 / linear without GEGLU (no GN on load, no split-K, bf16 out, no per-image row add), M a multiple of
// 256, Cout a multiple of 160, K >= 128, operands below 4 GB, and at least two tiles per CU on average (otherwise the
// 128x160 kernel's two workgroups per CU do better).
int dc_gemm_wide_supported(const dc_conv_desc& d)
{
    if (!(d.ksize == 1 && !d.gn_ab && d.splitk <= 1 && !d.out_f32 && !d.row_add && d.epilogue == 0)) return 0;
    const long long M = (long long)d.N * d.Ho * d.Wo;
    const long long K = d.C1 + d.C2;
    if (M % BM != 0 || d.Cout % BN != 0 || K < 128) return 0;
    if (M * (d.C1 > d.C2 ? d.C1 : d.C2) * 2 >= (1LL << 32) || (long long)d.Cout * K * 2 >= (1LL << 32) || M * d.Cout * 2 >= (1LL << 40)) return 0;
    return (M / BM) * (d.Cout / BN) >= 512;
}

int dc_gemm_wide_launch(const dc_conv_desc& d, hipStream_t st)
{
    const long long M = (long long)d.N * d.Ho * d.Wo;
    const int ntiles = (int)((M / BM) * (d.Cout / BN));
    static int num_cus = 0;
    if (!num_cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return DC_ERR_LAUNCH;
        num_cus = prop.multiProcessorCount;
    }
    const size_t lds = NST * (size_t)STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_wide_kernel, dim3(ntiles < num_cus ? ntiles : num_cus), dim3(NT), lds, st, d);
    return dc_launch_status();
}
