// Direct (non-MFMA) convolutions for the layers that are not GEMM-shaped on gfx950:
//   * fp32 NCHW 3x3 convs of the control extractors (controlnet/extractors.py:215-262, control_utils.py:43-47):
//     kept in fp32 like the reference's splat stage (softsplat.py:279); LDS-tiled 16x16 output pixels x 16 couts.
//   * NHWC bf16 convs with <= 16 input channels (UNet/ControlNet conv_in 4->320, VAE conv_in, post_quant_conv)
//   * NHWC bf16 convs with <= 8 output channels (UNet conv_out 320->4 with fused GroupNorm+SiLU, VAE conv_out,
//     quant_conv): one wave per output pixel, lanes split the (tap, channel-vector) reduction.
#include "dc_common.h"
#include "../../include/diffcodec_hip.h"
#include <cstdlib>

namespace {

constexpr int CO_T = 16;     // output channels per workgroup

// CI_T input channels staged per step (smaller for stride 4, whose 66x66 input patch is large)
template <int STRIDE, int CI_T>
__global__ __launch_bounds__(256) void conv3x3_nchw_f32_kernel(const float* __restrict__ x, long long xbs,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ y, int Cin, int H, int W, int Cout,
                                                               int Ho, int Wo, int silu)
{
    constexpr int PT = 16 * STRIDE + 2;                 // input patch edge
    __shared__ float s_in[CI_T][PT][PT + 1];
    __shared__ __attribute__((aligned(16))) float s_w[CI_T][9][CO_T];
    const int tiles_x = (Wo + 15) / 16;
    const int tx0 = (blockIdx.x % tiles_x) * 16, ty0 = (blockIdx.x / tiles_x) * 16;
    const int co0 = blockIdx.y * CO_T;
    const int n = blockIdx.z;
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int ox = tx0 + lx, oy = ty0 + ly;
    float acc[CO_T];
#pragma unroll
    for (int i = 0; i < CO_T; ++i) acc[i] = 0.f;
    const int ix0 = tx0 * STRIDE - 1, iy0 = ty0 * STRIDE - 1;
    for (int c0 = 0; c0 < Cin; c0 += CI_T) {
        __syncthreads();
        for (int i = threadIdx.x; i < CI_T * PT * PT; i += 256) {
            const int px = i % PT, py = (i / PT) % PT, ci = i / (PT * PT);
            const int gx = ix0 + px, gy = iy0 + py, c = c0 + ci;
            float v = 0.f;
            if (c < Cin && gx >= 0 && gx < W && gy >= 0 && gy < H) v = x[n * xbs + ((long long)c * H + gy) * W + gx];
            s_in[ci][py][px] = v;
        }
        for (int i = threadIdx.x; i < CI_T * 9 * CO_T; i += 256) {
            const int co = i % CO_T, tap = (i / CO_T) % 9, ci = i / (CO_T * 9);
            float v = 0.f;
            if (c0 + ci < Cin && co0 + co < Cout) v = w[(((long long)(co0 + co)) * Cin + (c0 + ci)) * 9 + tap];
            s_w[ci][tap][co] = v;
        }
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < CI_T; ++ci) {
            float in[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) in[t] = s_in[ci][ly * STRIDE + t / 3][lx * STRIDE + t % 3];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
#pragma unroll
                for (int g = 0; g < CO_T / 4; ++g) {
                    const f32x4 wv = *(const f32x4*)&s_w[ci][t][4 * g];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[4 * g + r] += in[t] * wv[r];
                }
            }
        }
    }
    if (ox < Wo && oy < Ho) {
#pragma unroll
        for (int i = 0; i < CO_T; ++i) {
            const int co = co0 + i;
            if (co < Cout) {
                float v = acc[i] + (bias ? bias[co] : 0.f);
                if (silu) v = dc_silu(v);
                y[(((long long)n * Cout + co) * Ho + oy) * Wo + ox] = v;
            }
        }
    }
}

// Register-blocked form for strides 1 and 2 (round 2).  The kernel above gives a thread ONE pixel x 16 output channels and a
// workgroup one input patch per 16 output channels: the staging of the patch (and, per (input channel, tap), four 16-byte LDS
// weight reads for 16 FMAs) weighs as much as the arithmetic.  Here a thread owns FOUR consecutive pixels of a row x 16 output
// channels (64 accumulators) and a workgroup is TROWS x TXG pixel-threads x CG output-channel groups sharing ONE input patch
// (16 x 16 pixels x 64 channels as launched): 4x the FMAs per staged input value and per weight read.  The accumulation order
// per output (input channels outer, taps inner) is unchanged.  160->320 at 64x64: 1.45 -> 0.96 ms, 64->64 at 128x128: 0.51 -> 0.34.
template <int STRIDE, int TROWS, int TXG, int CG, int CI_T>
__global__ __launch_bounds__(256) void conv3x3_nchw_f32_blk_kernel(const float* __restrict__ x, long long xbs,
                                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                                   float* __restrict__ y, int Cin, int H, int W, int Cout,
                                                                   int Ho, int Wo, int silu)
{
    static_assert(TROWS * TXG * CG == 256, "one pixel-thread x channel-group per thread");
    constexpr int TW = TXG * 4;                         // output tile width
    constexpr int PH = TROWS * STRIDE + 2, PW = TW * STRIDE + 2;
    constexpr int NC = 3 * STRIDE + 3;                  // input columns a thread needs per row
    constexpr int COT = CG * 16;                        // output channels per workgroup
    __shared__ float s_in[CI_T][PH][PW + 1];
    __shared__ __attribute__((aligned(16))) float s_w[CI_T][9][COT];
    const int tiles_x = (Wo + TW - 1) / TW;
    const int tx0 = (blockIdx.x % tiles_x) * TW, ty0 = (blockIdx.x / tiles_x) * TROWS;
    const int co0 = blockIdx.y * COT;
    const int n = blockIdx.z;
    const int cg = threadIdx.x / (TROWS * TXG), pt = threadIdx.x % (TROWS * TXG);
    const int ty = pt / TXG, tx = pt % TXG;
    const int oy = ty0 + ty, ox = tx0 + tx * 4;
    float acc[4][16];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;
    const int ix0 = tx0 * STRIDE - 1, iy0 = ty0 * STRIDE - 1;
    for (int c0 = 0; c0 < Cin; c0 += CI_T) {
        __syncthreads();
        for (int i = threadIdx.x; i < CI_T * PH * PW; i += 256) {
            const int px = i % PW, py = (i / PW) % PH, ci = i / (PW * PH);
            const int gx = ix0 + px, gy = iy0 + py, c = c0 + ci;
            float v = 0.f;
            if (c < Cin && gx >= 0 && gx < W && gy >= 0 && gy < H) v = x[n * xbs + ((long long)c * H + gy) * W + gx];
            s_in[ci][py][px] = v;
        }
        for (int i = threadIdx.x; i < CI_T * 9 * COT; i += 256) {
            const int co = i % COT, tap = (i / COT) % 9, ci = i / (COT * 9);
            float v = 0.f;
            if (c0 + ci < Cin && co0 + co < Cout) v = w[(((long long)(co0 + co)) * Cin + (c0 + ci)) * 9 + tap];
            s_w[ci][tap][co] = v;
        }
        __syncthreads();
#pragma unroll 1                                        // (unrolled, hipcc keeps several channels' windows and weights live: 256 VGPRs)
        for (int ci = 0; ci < CI_T; ++ci) {
            float in[3][NC];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c) in[r][c] = s_in[ci][ty * STRIDE + r][tx * 4 * STRIDE + c];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 wv = *(const f32x4*)&s_w[ci][t][cg * 16 + 4 * g];
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[p][4 * g + r] += in[t / 3][p * STRIDE + t % 3] * wv[r];
                }
            }
        }
    }
    if (oy >= Ho || ox >= Wo) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int co = co0 + cg * 16 + i;
        if (co >= Cout) break;
        const float b = bias ? bias[co] : 0.f;
        float v[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            v[p] = acc[p][i] + b;
            if (silu) v[p] = dc_silu(v[p]);
        }
        float* dst = y + (((long long)n * Cout + co) * Ho + oy) * Wo + ox;
        if (ox + 3 < Wo && (Wo & 3) == 0) {
            *(f32x4*)dst = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (ox + p < Wo) dst[p] = v[p];
        }
    }
}

// conv_in of the UNet / ControlNet (4 -> 320, 3x3, stride 1, pad 1): thread = (4 consecutive output pixels of a row, group of
// 8 output channels).  Each of the 36 weight vectors is loaded once and used for four pixels, the 3 x 6 input pixels of
// the strip are 8-byte loads: 4x fewer weight loads per FMA than the generic kernel below (197 -> ~80 us at 32 x 64 x 64).
__global__ __launch_bounds__(256) void conv_in4_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                       const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                       int N, int H, int W, int Cout)
{
    const int cog = Cout >> 3, wq = W >> 2;
    const long long total = (long long)N * H * wq * cog;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int g = (int)(i % cog);
    const long long s = i / cog;
    const int ox0 = (int)(s % wq) * 4, oy = (int)((s / wq) % H), n = (int)(s / ((long long)wq * H));
    float acc[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[p][j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy + ky - 1;
        if (iy < 0 || iy >= H) continue;
        float xin[6][4];                                   // input pixels ox0-1 .. ox0+4 of this row, 4 channels each
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int ix = ox0 + q - 1;
            bf16x4 v = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
            if (ix >= 0 && ix < W) v = *(const bf16x4*)(x + (((long long)n * H + iy) * W + ix) * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) xin[q][c] = (float)v[c];
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bf16x8 wv = *(const bf16x8*)(w + ((long long)((ky * 3 + kx) * 4 + c)) * Cout + g * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float wf = (float)wv[j];
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[p][j] += xin[p + kx][c] * wf;
                }
            }
    }
    float b8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) b8[j] = bias ? bias[g * 8 + j] : 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(acc[p][j] + b8[j]);
        *(bf16x8*)(out + ((((long long)n * H + oy) * W) + ox0 + p) * Cout + g * 8) = o;
    }
}

// thread = (pixel, group of 8 output channels); w [taps][Cin][Cout] bf16
__global__ __launch_bounds__(256) void conv_small_cin_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                             const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                             int N, int H, int W, int Cin, int Cout, int ks, int stride,
                                                             int pad, int Ho, int Wo)
{
    const int cog = (Cout + 7) / 8;
    const long long total = (long long)N * Ho * Wo * cog;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int g = (int)(i % cog);
    const long long m = i / cog;
    const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), n = (int)(m / ((long long)Wo * Ho));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx) {
            const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const bf16_t* xp = x + (((long long)n * H + iy) * W + ix) * Cin;
            const bf16_t* wp = w + (long long)(ky * ks + kx) * Cin * Cout + g * 8;
            if ((Cout & 7) == 0) {                       // 8 output channels = one 16-byte weight vector
                for (int c = 0; c < Cin; ++c) {
                    const float xv = (float)xp[c];
                    const bf16x8 wv = *(const bf16x8*)(wp + (long long)c * Cout);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += xv * (float)wv[j];
                }
            } else {
                for (int c = 0; c < Cin; ++c) {
                    const float xv = (float)xp[c];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (g * 8 + j < Cout) acc[j] += xv * (float)wp[(long long)c * Cout + j];
                }
            }
        }
    if ((Cout & 7) == 0) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(acc[j] + (bias ? bias[g * 8 + j] : 0.f));
        *(bf16x8*)(out + m * Cout + g * 8) = o;
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int co = g * 8 + j;
        if (co < Cout) out[m * Cout + co] = (bf16_t)(acc[j] + (bias ? bias[co] : 0.f));
    }
}

// one wave per output pixel; w [Cout][taps][Cin] bf16; stride 1, pad (ks-1)/2
template <int COUT>
__global__ __launch_bounds__(256) void conv_small_cout_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                              const float* __restrict__ bias, const float* __restrict__ ab,
                                                              int silu, int gn_batch, void* __restrict__ out, int out_f32,
                                                              int N, int H, int W, int Cin, int ks)
{
    const int lane = threadIdx.x & 63;
    const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long M = (long long)N * H * W;
    if (m >= M) return;
    const int ox = (int)(m % W), oy = (int)((m / W) % H), n = (int)(m / ((long long)W * H));
    const int nv = Cin >> 3, taps = ks * ks, padk = (ks - 1) / 2;
    float acc[COUT];
#pragma unroll
    for (int j = 0; j < COUT; ++j) acc[j] = 0.f;
    for (int idx = lane; idx < taps * nv; idx += 64) {
        const int tap = idx / nv, v = idx - tap * nv;
        const int iy = oy + tap / ks - padk, ix = ox + tap % ks - padk;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const bf16x8 xv = *(const bf16x8*)(x + (((long long)n * H + iy) * W + ix) * Cin + v * 8);
        float xf[8];
        if (ab) {
            const float* abp = ab + ((long long)(n % gn_batch) * Cin + v * 8) * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = (float)xv[j] * abp[2 * j] + abp[2 * j + 1];
                xf[j] = silu ? dc_silu(t) : t;
                xf[j] = (float)(bf16_t)xf[j];          // same rounding point as the MFMA path (bf16 operand)
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[j] = (float)xv[j];
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const bf16x8 wv = *(const bf16x8*)(w + ((long long)co * taps + tap) * Cin + v * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[co] += xf[j] * (float)wv[j];
        }
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = dc_wave_sum(acc[co]);
    if (lane == 0) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float v = acc[co] + (bias ? bias[co] : 0.f);
            if (out_f32) ((float*)out)[m * COUT + co] = v;
            else ((bf16_t*)out)[m * COUT + co] = (bf16_t)v;
        }
    }
}

}  // namespace

// conv_f32_mfma.hip
int dc_conv_f32_mfma_wanted(int Cin, int H, int W, int Cout, int stride);
int dc_conv_f32_mfma_launch(const float* x, long long xbs, const float* w, const float* bias, float* y, int N, int Cin, int H, int W,
                            int Cout, int stride, int silu, hipStream_t st);

extern "C" int dc_conv3x3_nchw_f32(const float* x, long long x_batch_stride, const float* w, const float* bias, float* y,
                                   int N, int Cin, int H, int W, int Cout, int stride, int silu, void* stream)
{
    if (!x || !w || !y || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2 && stride != 4)) return DC_ERR_INVALID;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    hipStream_t st = (hipStream_t)stream;
    // GEMM-shaped layers (>= 16 input channels, output channels in whole 32-wide tiles): exact-fp32 MFMA form, conv_f32_mfma.hip
    if (dc_conv_f32_mfma_wanted(Cin, H, W, Cout, stride)) return dc_conv_f32_mfma_launch(x, x_batch_stride, w, bias, y, N, Cin, H, W, Cout, stride, silu, st);
    // the register-blocked form (16 x 16 pixels x 64 output channels per workgroup) wins where there are >= 64 output channels to
    // share an input patch and the map is at least 64 wide (or 32 wide with >= 160 channels); measured per shape, tools/bench_f32conv.py
    static const int blk = DC_KNOB("DC_F32CONV_BLOCKED", 1);     // developer A/B knob
    if (blk && stride != 4 && Cout >= 64 && (Wo >= 64 || (Wo >= 32 && Cout >= 160))) {
        const dim3 bgrid(dc_cdiv(Wo, 16) * dc_cdiv(Ho, 16), dc_cdiv(Cout, 64), N);
        if (stride == 1) hipLaunchKernelGGL((conv3x3_nchw_f32_blk_kernel<1, 16, 4, 4, 8>), bgrid, dim3(256), 0, st, x, x_batch_stride, w, bias, y, Cin, H, W, Cout, Ho, Wo, silu);
        else hipLaunchKernelGGL((conv3x3_nchw_f32_blk_kernel<2, 16, 4, 4, 8>), bgrid, dim3(256), 0, st, x, x_batch_stride, w, bias, y, Cin, H, W, Cout, Ho, Wo, silu);
        return dc_launch_status();
    }
    const dim3 grid(dc_cdiv(Wo, 16) * dc_cdiv(Ho, 16), dc_cdiv(Cout, CO_T), N);
    if (stride == 1) hipLaunchKernelGGL((conv3x3_nchw_f32_kernel<1, 8>), grid, dim3(256), 0, st, x, x_batch_stride, w, bias, y, Cin, H, W, Cout, Ho, Wo, silu);
    else if (stride == 2) hipLaunchKernelGGL((conv3x3_nchw_f32_kernel<2, 8>), grid, dim3(256), 0, st, x, x_batch_stride, w, bias, y, Cin, H, W, Cout, Ho, Wo, silu);
    else hipLaunchKernelGGL((conv3x3_nchw_f32_kernel<4, 2>), grid, dim3(256), 0, st, x, x_batch_stride, w, bias, y, Cin, H, W, Cout, Ho, Wo, silu);
    return dc_launch_status();
}

extern "C" int dc_conv_small_cin_bf16(const void* x, const void* w, const float* bias, void* out, int N, int H, int W,
                                      int Cin, int Cout, int ksize, int stride, int pad, int Ho, int Wo, void* stream)
{
    if (!x || !w || !out || N <= 0 || Cin <= 0 || Cin > 16 || Cout <= 0 || (ksize != 1 && ksize != 3)) return DC_ERR_INVALID;
    if (Cin == 4 && ksize == 3 && stride == 1 && pad == 1 && (Cout & 7) == 0 && (W & 3) == 0 && Ho == H && Wo == W) {
        const long long strips = (long long)N * H * (W >> 2) * (Cout >> 3);
        hipLaunchKernelGGL(conv_in4_kernel, dim3(dc_cdiv(strips, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)w, bias, (bf16_t*)out, N, H, W, Cout);
        return dc_launch_status();
    }
    const long long total = (long long)N * Ho * Wo * ((Cout + 7) / 8);
    hipLaunchKernelGGL(conv_small_cin_kernel, dim3(dc_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (const bf16_t*)w, bias, (bf16_t*)out, N, H, W, Cin, Cout, ksize, stride, pad, Ho, Wo);
    return dc_launch_status();
}

extern "C" int dc_conv_small_cout_bf16(const void* x, const void* w, const float* bias, const float* gn_ab, int gn_silu,
                                       int gn_batch, void* out, int out_f32, int N, int H, int W, int Cin, int Cout,
                                       int ksize, void* stream)
{
    if (!x || !w || !out || N <= 0 || Cin <= 0 || (Cin & 7) || (ksize != 1 && ksize != 3)) return DC_ERR_INVALID;
    if (gn_ab && gn_batch <= 0) return DC_ERR_INVALID;
    const long long M = (long long)N * H * W;
    const dim3 grid(dc_cdiv(M, 4));
    hipStream_t st = (hipStream_t)stream;
#define DC_SC(CO) hipLaunchKernelGGL(conv_small_cout_kernel<CO>, grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w, bias, gn_ab, gn_silu, gn_batch, out, out_f32, N, H, W, Cin, ksize)
    switch (Cout) {
        case 3: DC_SC(3); break;
        case 4: DC_SC(4); break;
        case 8: DC_SC(8); break;
        default: return DC_ERR_INVALID;
    }
#undef DC_SC
    return dc_launch_status();
}
