// fp32 3x3 conv (NCHW, padding 1, stride 1 | 2, + SiLU) of the control extractors on the fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 — the many-channel layers of Bi_Dir_FeatureExtractor / ResidueExtractor / WarpExtractor
// (controlnet/extractors.py:215-262, control_utils.py:43-47: 16->32 ... 640->1280), which are GEMM-shaped but must stay fp32
// (the pyramid feeds a 0.3-pixel occlusion test and the goldens are fp32).  gfx950 has no reduced-precision fp32 path; this
// instruction is exact fp32 (a k-ordered fmaf chain) at the VALU's peak rate, but one instruction does the work of 64 v_fma and
// its operands are ONE register each, so the staging / address arithmetic that bound the VALU kernels of conv_direct.hip
// (19.7 TFLOP/s over the family in round 2) no longer sits in the FMA stream.
//
// Implicit GEMM  D[co][pixel] += W[co][(ci, tap)] * X[(ci, tap)][pixel]:
//   workgroup = 4 waves, tile = CO_T (64 | 32) output channels x PT (128 | 64) output pixels of ONE sample (rows_t x cols_t,
//   cols_t = min(Wo, PT));  K runs over chunks of 8 input channels: the chunk's input patch ([8][PH][PW] floats, zero padding
//   materialised) and weight slice ([8][9][CO_T], transposed on the way in so that lanes read consecutive channels) are staged
//   in LDS from registers that were loaded one chunk ahead (global latency hides under the previous chunk's MFMAs);
//   an MFMA k-step is a PAIR of input channels at one tap (lane half h takes channel 2 cp + h): A = one ds_read_b32 of the
//   weight slice, B = one ds_read_b32 of the patch at the lane's pixel shifted by the tap — consecutive lanes, consecutive words.
//   C/D layout (dtype-independent): lane & 31 = pixel, 16 registers = channels (r & 3) + 8 (r >> 2) + 4 (lane >> 5): every store
//   instruction writes 32 consecutive pixels of one channel row.
#include "dc_common.h"
#include "../../include/diffcodec_hip.h"

namespace {

constexpr int FM_CK = 8;                       // input channels per LDS chunk
constexpr int FM_MAXE = 26;                    // patch floats per thread per chunk (largest patch: stride 2, 128 x 1 tile: 8*3*257)

template <int STRIDE, int CO_T, int PT>
__global__ __launch_bounds__(256, 2) void conv3x3_f32_mfma_kernel(const float* __restrict__ x, long long x_batch_stride,
                                                                  const float* __restrict__ w, const float* __restrict__ bias,
                                                                  float* __restrict__ y, int Cin, int H, int W, int Cout, int Ho,
                                                                  int Wo, int cols_t, int silu)
{
    constexpr int NCH = CO_T / 32;                         // 32-channel halves of the tile: 2 | 1
    constexpr int NPX = 4 / NCH;                           // pixel slices over the waves: 2 | 4
    constexpr int NB = PT / (32 * NPX);                    // 32-pixel B tiles per wave
    constexpr int WSZ = FM_CK * 9 * CO_T;                  // floats of the weight slice
    constexpr int WPT = (FM_CK * 9 / 4 * CO_T + 255) / 256;   // float4 weight pieces per thread
    static_assert(NB >= 1, "tile too small for the wave split");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Wl = lds;                                       // [FM_CK * 9][CO_T]
    float* Pl = lds + WSZ;                                 // [FM_CK][PH][PW]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = wave % NCH, ph = wave / NCH;
    const int j = lane & 31, h = lane >> 5;
    const int rows_t = PT / cols_t;
    const int PH = (rows_t - 1) * STRIDE + 3, PW = (cols_t - 1) * STRIDE + 3;
    const int tiles_x = Wo / cols_t;
    const int ty0 = (blockIdx.x / tiles_x) * rows_t, tx0 = (blockIdx.x % tiles_x) * cols_t;
    const int co0 = blockIdx.y * CO_T, n = blockIdx.z;
    const float* __restrict__ xn = x + (long long)n * x_batch_stride;

    // ---- patch staging plan: element e = tid + 256 i of [FM_CK][PH][PW] -> global offset inside the chunk (or -1: zero padding)
    const int pe = FM_CK * PH * PW;
    int goff[FM_MAXE];
#pragma unroll
    for (int i = 0; i < FM_MAXE; ++i) {
        const int e = tid + 256 * i;
        goff[i] = -1;
        if (e < pe) {
            const int ci = e / (PH * PW), r = e - ci * (PH * PW);
            const int py = r / PW, px = r - py * PW;
            const int iy = ty0 * STRIDE - 1 + py, ix = tx0 * STRIDE - 1 + px;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) goff[i] = (ci * H + iy) * W + ix;
        }
    }
    // weight staging plan: thread -> channel co = tid % CO_T, float4 pieces q + (256 / CO_T) i of the 18 per (co, chunk)
    constexpr int TPC = 256 / CO_T;                        // threads per output channel: 4 | 8
    const int wco = tid % CO_T, wq = tid / CO_T;
    const float* __restrict__ wrow = w + ((long long)(co0 + wco) * Cin) * 9;

    float rp[FM_MAXE];
    f32x4 rw[WPT];
    auto load_chunk = [&](int c0) {
        const float* xc = xn + (long long)c0 * H * W;
#pragma unroll
        for (int i = 0; i < FM_MAXE; ++i) rp[i] = goff[i] >= 0 ? xc[goff[i]] : 0.f;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int p = wq + TPC * i;
            rw[i] = p < 18 ? *(const f32x4*)(wrow + c0 * 9 + p * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < FM_MAXE; ++i) {
            const int e = tid + 256 * i;
            if (e < pe) Pl[e] = rp[i];
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int p = wq + TPC * i;
            if (p < 18) {
#pragma unroll
                for (int e = 0; e < 4; ++e) Wl[(p * 4 + e) * CO_T + wco] = rw[i][e];
            }
        }
    };

    // ---- this lane's pixels: B tile b of the wave -> pixel index inside the tile -> patch offset of its top-left tap
    int pbase[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int p = (ph * NB + b) * 32 + j;
        const int ty = p / cols_t, tx = p - ty * cols_t;
        pbase[b] = (ty * STRIDE) * PW + tx * STRIDE + h * (PH * PW);      // lane half h reads the odd channel of the pair
    }
    const int abase = h * 9 * CO_T + ch * 32 + j;

    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    const int nchunk = Cin / FM_CK;
    load_chunk(0);
    for (int c = 0; c < nchunk; ++c) {
        __syncthreads();                                   // everyone is done reading the previous chunk
        store_chunk();
        __syncthreads();
        if (c + 1 < nchunk) load_chunk((c + 1) * FM_CK);   // flies under this chunk's MFMAs
#pragma unroll
        for (int cp = 0; cp < FM_CK / 2; ++cp) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float a = Wl[abase + (cp * 2 * 9 + tap) * CO_T];
                const int poff = cp * 2 * (PH * PW) + (tap / 3) * PW + (tap % 3);
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Pl[pbase[b] + poff], acc[b], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: bias (+ SiLU); register r = channel (r & 3) + 8 (r >> 2) + 4 h of the wave's 32, lane & 31 = pixel
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int p = (ph * NB + b) * 32 + j;
        const int ty = p / cols_t, tx = p - ty * cols_t;
        float* __restrict__ yo = y + (((long long)n * Cout + co0 + ch * 32) * Ho + ty0 + ty) * Wo + tx0 + tx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = acc[b][r] + (bias ? bias[co0 + ch * 32 + col] : 0.f);
            if (silu) v = dc_silu(v);
            yo[(long long)col * Ho * Wo] = v;
        }
    }
}

template <int STRIDE, int CO_T, int PT>
int launch_fm(const float* x, long long xbs, const float* w, const float* bias, float* y, int N, int Cin, int H, int W, int Cout,
              int Ho, int Wo, int silu, hipStream_t st)
{
    const int cols_t = Wo < PT ? Wo : PT, rows_t = PT / cols_t;
    const int PH = (rows_t - 1) * STRIDE + 3, PW = (cols_t - 1) * STRIDE + 3;
    if (FM_CK * PH * PW > FM_MAXE * 256) return DC_ERR_INVALID;
    const size_t lds = (size_t)(FM_CK * 9 * CO_T + FM_CK * PH * PW) * 4;
    auto kern = conv3x3_f32_mfma_kernel<STRIDE, CO_T, PT>;
    static std::atomic<unsigned long long> attr_done{0};
    dc_set_max_dyn_lds((const void*)kern, (int)lds, attr_done);
    const dim3 grid((Wo / cols_t) * (Ho / rows_t), Cout / CO_T, N);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, xbs, w, bias, y, Cin, H, W, Cout, Ho, Wo, cols_t, silu);
    return dc_launch_status();
}

}  // namespace

// The MFMA form takes a layer when the GEMM dimensions are whole tiles: Cin a multiple of the 8-channel chunk (>= 16), Cout a
// multiple of 32, stride 1 | 2 with exact halving, and an output map that splits into 128-pixel (8x8 maps: 64-pixel) tiles.
// DC_F32CONV_MFMA (developer builds): 0 = never.
int dc_conv_f32_mfma_wanted(int Cin, int H, int W, int Cout, int stride)
{
    static const int mode = DC_KNOB("DC_F32CONV_MFMA", 1);
    if (!mode || (stride != 1 && stride != 2) || Cin < 16 || Cin % FM_CK || Cout % 32) return 0;
    if (stride == 2 && ((H | W) & 1)) return 0;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const int pt = Ho * Wo >= 128 ? 128 : 64;
    if (Ho * Wo < 64 || (pt == 64 && Cout % 64)) return 0;   // the 64-pixel tile exists for 64-channel tiles only
    const int cols_t = Wo < pt ? Wo : pt;
    if (pt % cols_t || Wo % cols_t || Ho % (pt / cols_t)) return 0;
    const int rows_t = pt / cols_t;
    if (FM_CK * ((rows_t - 1) * stride + 3) * ((cols_t - 1) * stride + 3) > FM_MAXE * 256) return 0;
    return 1;
}

int dc_conv_f32_mfma_launch(const float* x, long long xbs, const float* w, const float* bias, float* y, int N, int Cin, int H, int W,
                            int Cout, int stride, int silu, hipStream_t st)
{
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const bool small = Ho * Wo < 128, co64 = Cout % 64 == 0;
#define FM_GO(S, C, P) return launch_fm<S, C, P>(x, xbs, w, bias, y, N, Cin, H, W, Cout, Ho, Wo, silu, st)
    if (stride == 1) {
        if (small) FM_GO(1, 64, 64);
        if (co64) FM_GO(1, 64, 128);
        FM_GO(1, 32, 128);
    }
    if (small) FM_GO(2, 64, 64);
    if (co64) FM_GO(2, 64, 128);
    FM_GO(2, 32, 128);
#undef FM_GO
}
