// Layout / dtype / scheduler elementwise kernels (HBM-bound) for gfx950.
#include "dc_common.h"
#include "../../include/diffcodec_hip.h"

namespace {

inline int grid_for(long long total) { return (int)min((long long)8192, (total + 255) / 256); }

__global__ void nchw_f32_to_nhwc_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, int C, long long HW, long long total)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long p = (i / C) % HW, n = i / (C * HW);
        d[i] = (bf16_t)s[(n * C + c) * HW + p];
    }
}
template <typename T>
__global__ void nhwc_to_nchw_f32_kernel(const T* __restrict__ s, float* __restrict__ d, int C, long long HW, long long total)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i % HW;
        const int c = (int)((i / HW) % C);
        const long long n = i / (C * HW);
        d[i] = (float)s[(n * HW + p) * C + c];
    }
}
__global__ void f32_to_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) d[i] = (bf16_t)s[i];
}
__global__ void silu_f32_kernel(const float* __restrict__ s, float* __restrict__ d, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) d[i] = dc_silu(s[i]);
}
__global__ void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = a[i] + b[i];
}
__global__ void add_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ y, long long nvec)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const bf16x8 av = *(const bf16x8*)(a + i * 8), bv = *(const bf16x8*)(b + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)av[j] + (float)bv[j]);
        *(bf16x8*)(y + i * 8) = o;
    }
}
// diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]
__global__ void timestep_embedding_kernel(const float* __restrict__ t, const int* __restrict__ step, float* __restrict__ out, int n, int dim)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int half = dim / 2;
    if (i >= n * half) return;
    const int k = i % half, r = i / half;
    const float freq = expf(-9.210340371976184f * (float)k / (float)half);
    const float a = t[step ? *step : 0] * freq;
    out[(long long)r * dim + k] = cosf(a);
    out[(long long)r * dim + half + k] = sinf(a);
}
// pipeline.py:370-375: CFG combine, DDIM update (fp32 state), next model input (bf16 NHWC, duplicated for CFG)
__global__ void cfg_ddim_kernel(const float* __restrict__ eps, float* __restrict__ lat, bf16_t* __restrict__ model_in,
                                const float* __restrict__ coef, int* __restrict__ step, float guidance, int cfg, int B, int C,
                                long long HW)
{
    const long long total = (long long)B * C * HW;
    const int st = *step;
    const float s1mat = coef[st * 4 + 0], sat = coef[st * 4 + 1], sap = coef[st * 4 + 2], s1map = coef[st * 4 + 3];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i % HW;
        const int c = (int)((i / HW) % C);
        const long long b = i / (HW * C);
        float e;
        if (cfg) {
            const float eu = eps[(b * HW + p) * C + c];
            const float et = eps[((b + B) * HW + p) * C + c];
            e = eu + guidance * (et - eu);
        } else {
            e = eps[(b * HW + p) * C + c];
        }
        const float x = lat[i];
        const float x0 = (x - s1mat * e) / sat;
        const float xn = sap * x0 + s1map * e;
        lat[i] = xn;
        const bf16_t xb = (bf16_t)xn;
        model_in[(b * HW + p) * C + c] = xb;
        if (cfg) model_in[((b + B) * HW + p) * C + c] = xb;
    }
}
__global__ void bump_step_kernel(int* step) { *step += 1; }
__global__ void latents_to_model_input_kernel(const float* __restrict__ lat, bf16_t* __restrict__ model_in, float mul, int rep,
                                              int B, int C, long long HW)
{
    const long long total = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i % HW;
        const int c = (int)((i / HW) % C);
        const long long b = i / (HW * C);
        const bf16_t v = (bf16_t)(lat[i] * mul);
        for (int r = 0; r < rep; ++r) model_in[((b + (long long)r * B) * HW + p) * C + c] = v;
    }
}
__global__ void postprocess_kernel(const float* __restrict__ x, float* __restrict__ o32, uint8_t* __restrict__ o8, int C,
                                   long long HW, long long total, int xs)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long p = (i / C) % HW, n = i / (C * HW);
        float v = x[(i / C) * xs + c] / 2.0f + 0.5f;             // xs: input elements per pixel (>= C when the conv padded Cout)
        v = fminf(fmaxf(v, 0.f), 1.f);
        if (o32) o32[(n * C + c) * HW + p] = v;
        if (o8) o8[i] = (uint8_t)rintf(v * 255.0f);
    }
}

// FreeU (validation.py:106 -> diffusers apply_freeu / fourier_filter [recalled]): the skip feature's four lowest
// frequency bins (k in {0,-1}^2 of the 2-D DFT, i.e. the 2x2 centre block after fftshift) are scaled by `s`, everything
// else is kept.  Four DFT coefficients need no FFT: seven real sums per (sample, channel), then
//   y = x + (s-1)/(HW) * Re sum_k X[k] e^{+i k.phi}.   One thread per (sample, channel); maps are 8x8 / 16x16.
__global__ void freeu_lowfreq_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N, int H, int W, int C, float s)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    const bf16_t* xp = x + (long long)n * H * W * C + c;
    bf16_t* yp = y + (long long)n * H * W * C + c;
    const float wy = 6.283185307179586f / (float)H, wx = 6.283185307179586f / (float)W;
    float a = 0.f, cx = 0.f, sx = 0.f, cy = 0.f, sy = 0.f, cxy = 0.f, sxy = 0.f;
    for (int py = 0; py < H; ++py) {
        float syv, cyv;
        __sincosf(wy * (float)py, &syv, &cyv);
        for (int px = 0; px < W; ++px) {
            float sxv, cxv;
            __sincosf(wx * (float)px, &sxv, &cxv);
            const float v = (float)xp[((long long)py * W + px) * C];
            a += v;
            cx += v * cxv;
            sx += v * sxv;
            cy += v * cyv;
            sy += v * syv;
            cxy += v * (cxv * cyv - sxv * syv);        // cos(phi_x + phi_y)
            sxy += v * (sxv * cyv + cxv * syv);        // sin(phi_x + phi_y)
        }
    }
    const float g = (s - 1.0f) / (float)(H * W);
    for (int py = 0; py < H; ++py) {
        float syv, cyv;
        __sincosf(wy * (float)py, &syv, &cyv);
        for (int px = 0; px < W; ++px) {
            float sxv, cxv;
            __sincosf(wx * (float)px, &sxv, &cxv);
            const long long o = ((long long)py * W + px) * C;
            const float low = a + cx * cxv + sx * sxv + cy * cyv + sy * syv + cxy * (cxv * cyv - sxv * syv) + sxy * (sxv * cyv + cxv * syv);
            yp[o] = (bf16_t)((float)xp[o] + g * low);
        }
    }
}
// FreeU backbone scaling: hidden[:, :C/2] *= b  (NHWC: the first half of every pixel's channels)
__global__ void freeu_backbone_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int C, float b, long long nvec)
{
    const int nv = C >> 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % nv) * 8;
        const bf16x8 v = *(const bf16x8*)(x + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (c + j) < C / 2 ? (bf16_t)((float)v[j] * b) : v[j];
        *(bf16x8*)(y + i * 8) = o;
    }
}
// y = a*x0 + b*x1 + c*x2 + d*x3 (fp32; null pointers skipped) — scheduler state updates (UniPC predictor / corrector,
// epsilon -> x0 conversion, CFG combine in the generic loop: pipeline.py:370-375)
// y = a x0 + b x1 + c x2 + d x3, terms with `has` false skipped: ONE definition of the rounding sequence (products folded into the
// running sum by explicit fused multiply-adds, in term order) shared by dc_lincomb4_f32 — the generic scheduler path — and the
// fused UniPC step below, so the two paths are bit-identical by construction.
__device__ __forceinline__ float lincomb4(float a, float x0, float b, float x1, bool h1, float c, float x2, bool h2, float d, float x3,
                                          bool h3)
{
#pragma clang fp contract(off)
    float v = a * x0;
    if (h1) v = __builtin_fmaf(b, x1, v);
    if (h2) v = __builtin_fmaf(c, x2, v);
    if (h3) v = __builtin_fmaf(d, x3, v);
    return v;
}
__global__ void lincomb4_kernel(const float* __restrict__ x0, const float* __restrict__ x1, const float* __restrict__ x2,
                                const float* __restrict__ x3, float a, float b, float c, float d, float* __restrict__ y, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = lincomb4(a, x0[i], b, x1 ? x1[i] : 0.f, x1 != nullptr, c, x2 ? x2[i] : 0.f, x2 != nullptr, d, x3 ? x3[i] : 0.f,
                        x3 != nullptr);
}
// CFG combine + one UniPCMultistepScheduler.step (bh2, predict_x0, order <= 2: validation.py:37) as ONE pass over the latent-sized
// state, every coefficient read from a device table row indexed by the device step counter — so a captured denoising step
// replays for all steps, the order warm-up and the lower-order final step included (they are data: flags in the row).
// Row (DC_UNIPC_ROW floats): 0 1/alpha_i  1 -sigma_i/alpha_i | 2 flags (1 corrector, 2 corrector uses m1, 4 predictor uses m1)
//   3..6 corrector coefficients of (last_sample, m0, m_t, m1) | 7..9 predictor coefficients of (x, m_t, m1)
// State (fp32, NCHW like the latents): m0 / m1 = the two most recent x0-predictions, last = the sample the previous predictor
// started from.  The arithmetic is scheduler.py's sequence of dc_lincomb4_f32 launches, term for term.
#define DC_UNIPC_ROW 12
__global__ void cfg_unipc_kernel(const float* __restrict__ eps, float* __restrict__ lat, float* __restrict__ m0, float* __restrict__ m1,
                                 float* __restrict__ last, bf16_t* __restrict__ model_in, const float* __restrict__ coef,
                                 const int* __restrict__ step, float guidance, int cfg, int B, int C, long long HW)
{
    const long long total = (long long)B * C * HW;
    const float* r = coef + (long long)(*step) * DC_UNIPC_ROW;
    const float ca = r[0], ce = r[1];
    const int flags = (int)r[2];
    const bool corr = flags & 1, corr_m1 = flags & 2, pred_m1 = flags & 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i % HW;
        const int c = (int)((i / HW) % C);
        const long long b = i / (HW * C);
        float e;
        if (cfg) e = lincomb4(1.0f - guidance, eps[(b * HW + p) * C + c], guidance, eps[((b + B) * HW + p) * C + c], true, 0.f, 0.f, false, 0.f, 0.f, false);
        else e = eps[(b * HW + p) * C + c];
        float x = lat[i];
        const float mt = lincomb4(ca, x, ce, e, true, 0.f, 0.f, false, 0.f, 0.f, false);              // convert_model_output
        const float pm0 = m0[i], pm1 = m1[i];
        if (corr) x = lincomb4(r[3], last[i], r[4], pm0, true, r[5], mt, true, r[6], pm1, corr_m1);     // multistep_uni_c_bh_update
        last[i] = x;
        m1[i] = pm0;
        m0[i] = mt;
        const float xn = lincomb4(r[7], x, r[8], mt, true, r[9], pm0, pred_m1, 0.f, 0.f, false);        // multistep_uni_p_bh_update
        lat[i] = xn;
        const bf16_t xb = (bf16_t)xn;
        model_in[(b * HW + p) * C + c] = xb;
        if (cfg) model_in[((b + B) * HW + p) * C + c] = xb;
    }
}
__global__ void transpose_bf16_kernel(const bf16_t* __restrict__ s, bf16_t* __restrict__ d, int R, int C)
{
    __shared__ bf16_t tile[32][33];
    const long long base = (long long)blockIdx.z * R * C;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        if (r < R && c < C) tile[j][threadIdx.x] = s[base + (long long)r * C + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (r < R && c < C) d[base + (long long)c * R + r] = tile[threadIdx.x][j];
    }
}
__global__ void vae_sample_kernel(const float* __restrict__ mom, const float* __restrict__ noise, float* __restrict__ lat, float scale,
                                  int C, long long HW, long long total)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i % HW;
        const int c = (int)((i / HW) % C);
        const long long n = i / (HW * C);
        const float mean = mom[(n * HW + p) * 2 * C + c];
        float lv = mom[(n * HW + p) * 2 * C + C + c];
        lv = fminf(fmaxf(lv, -30.0f), 20.0f);
        lat[i] = (mean + expf(0.5f * lv) * noise[i]) * scale;
    }
}

// ---- input side (controlnet/utils.py:21-39).  The .flo payload is [H][W][2] fp32 and PIL images are [H][W][3] uint8:
// both are read in their file layout and written as NCHW fp32 planes of the control tensors, so the host only uploads
// the raw bytes.
// resize_flow_to: F.interpolate(bilinear, align_corners=True) then u *= tw/W, v *= th/H (fp32 scalars, as torch's
// in-place mul by a Python float).  Index arithmetic follows ATen's area_pixel_compute_source_index in fp32.
__global__ void flow_hw2_resize_kernel(const float* __restrict__ src, int H, int W, float* __restrict__ dst, int th, int tw,
                                       float mul_u, float mul_v, float sy, float sx)
{
#pragma clang fp contract(off)   // the source index must be the ROUNDED product (ATen): an fma into the lambda shifts it by 1/2 ulp(index)
    const long long total = (long long)th * tw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / tw), x = (int)(i - (long long)y * tw);
        const float fy = sy * (float)y, fx = sx * (float)x;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const float2 a = *(const float2*)(src + ((long long)y0 * W + x0) * 2), b = *(const float2*)(src + ((long long)y0 * W + x1) * 2);
        const float2 c = *(const float2*)(src + ((long long)y1 * W + x0) * 2), e = *(const float2*)(src + ((long long)y1 * W + x1) * 2);
        dst[i] = (hy * (hx * a.x + lx * b.x) + ly * (hx * c.x + lx * e.x)) * mul_u;
        dst[total + i] = (hy * (hx * a.y + lx * b.y) + ly * (hx * c.y + lx * e.y)) * mul_v;
    }
}

// TF.to_tensor + torch.cat of load_pair_to_sixch: dst[c][y][x] = img0[y][x][c] / 255, dst[3 + c] from img1
__global__ void pack_sixch_kernel(const uint8_t* __restrict__ img0, const uint8_t* __restrict__ img1, float* __restrict__ dst,
                                  long long HW)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dst[c * HW + i] = (float)img0[i * 3 + c] / 255.0f;
            dst[(3 + c) * HW + i] = (float)img1[i * 3 + c] / 255.0f;
        }
    }
}

// Tiled-decode blend (the weighting of tiling.merge_ramp, this package's driver policy): gather
// form, one thread per output pixel, tiles visited in index order with numpy's fp32 op order (mul, mul, add — no fma), so
// the result equals the host merge bit for bit.  ramp[i] = 0.5 - 0.5 cos(pi (i + 0.5) / feather) is computed by the
// caller; it is applied on tile edges that lie inside the frame only.
__global__ void blend_tiles_ramp_kernel(const float* __restrict__ tiles, const int* __restrict__ coords, int T, int C,
                                        int th, int tw, const float* __restrict__ ramp, int feather,
                                        uint8_t* __restrict__ out, int H, int W, float scale)
{
#pragma clang fp contract(off)
    const long long total = (long long)H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long long)y * W);
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, wsum = 0.f;
        for (int t = 0; t < T; ++t) {
            const int y1 = coords[4 * t], y2 = coords[4 * t + 1], x1 = coords[4 * t + 2], x2 = coords[4 * t + 3];
            if (y < y1 || y >= y2 || x < x1 || x >= x2) continue;
            const int ly = y - y1, lx = x - x1, ny = y2 - y1, nx = x2 - x1;
            float wy = 1.f, wx = 1.f;
            if (y1 > 0 && ly < feather) wy = ramp[ly];
            if (y2 < H && ly >= ny - feather) wy = ramp[ny - 1 - ly];
            if (x1 > 0 && lx < feather) wx = ramp[lx];
            if (x2 < W && lx >= nx - feather) wx = ramp[nx - 1 - lx];
            const float m2 = wy * wx;
            for (int c = 0; c < C; ++c) {
                const float v = tiles[(((long long)t * C + c) * th + ly) * tw + lx] * scale;
                acc[c] = acc[c] + v * m2;
            }
            wsum = wsum + m2;
        }
        for (int c = 0; c < C; ++c) {
            const float q = rintf(acc[c] / wsum);
            out[i * C + c] = (uint8_t)fminf(fmaxf(q, 0.f), 255.f);
        }
    }
}

}  // namespace

extern "C" int dc_nchw_f32_to_nhwc_bf16(const float* src, void* dst, int N, int C, int H, int W, void* stream)
{
    if (!src || !dst) return DC_ERR_INVALID;
    const long long total = (long long)N * C * H * W;
    hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, C, (long long)H * W, total);
    return dc_launch_status();
}
extern "C" int dc_nhwc_bf16_to_nchw_f32(const void* src, float* dst, int N, int C, int H, int W, void* stream)
{
    if (!src || !dst) return DC_ERR_INVALID;
    const long long total = (long long)N * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, C, (long long)H * W, total);
    return dc_launch_status();
}
extern "C" int dc_nhwc_f32_to_nchw_f32(const float* src, float* dst, int N, int C, int H, int W, void* stream)
{
    if (!src || !dst) return DC_ERR_INVALID;
    const long long total = (long long)N * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, C, (long long)H * W, total);
    return dc_launch_status();
}
extern "C" int dc_f32_to_bf16(const float* src, void* dst, long long n, void* stream)
{
    if (!src || !dst || n <= 0) return DC_ERR_INVALID;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
    return dc_launch_status();
}
extern "C" int dc_silu_f32(const float* x, float* y, long long n, void* stream)
{
    if (!x || !y || n <= 0) return DC_ERR_INVALID;
    hipLaunchKernelGGL(silu_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return dc_launch_status();
}
extern "C" int dc_add_f32(const float* a, const float* b, float* y, long long n, void* stream)
{
    if (!a || !b || !y || n <= 0) return DC_ERR_INVALID;
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
    return dc_launch_status();
}
extern "C" int dc_add_bf16(const void* a, const void* b, void* y, long long n, void* stream)
{
    if (!a || !b || !y || n <= 0 || (n & 7)) return DC_ERR_INVALID;
    hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n / 8);
    return dc_launch_status();
}
extern "C" int dc_timestep_embedding_f32(const float* t_dev, const int* step_dev, float* out, int n, int dim, void* stream)
{
    if (!t_dev || !out || n <= 0 || dim <= 0 || (dim & 1)) return DC_ERR_INVALID;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(dc_cdiv((long long)n * dim / 2, 256)), dim3(256), 0, (hipStream_t)stream, t_dev, step_dev, out, n, dim);
    return dc_launch_status();
}
extern "C" int dc_freeu_lowfreq_nhwc_bf16(const void* x, void* y, int N, int H, int W, int C, float s, void* stream)
{
    if (!x || !y || N <= 0 || H <= 1 || W <= 1 || C <= 0) return DC_ERR_INVALID;
    hipLaunchKernelGGL(freeu_lowfreq_kernel, dim3(dc_cdiv((long long)N * C, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)y, N, H, W, C, s);
    return dc_launch_status();
}
extern "C" int dc_freeu_backbone_nhwc_bf16(const void* x, void* y, long long pixels, int C, float b, void* stream)
{
    if (!x || !y || pixels <= 0 || C <= 0 || (C & 15)) return DC_ERR_INVALID;
    const long long nvec = pixels * (C >> 3);
    hipLaunchKernelGGL(freeu_backbone_kernel, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, C, b, nvec);
    return dc_launch_status();
}
extern "C" int dc_lincomb4_f32(const float* x0, const float* x1, const float* x2, const float* x3, float a, float b, float c,
                               float d, float* y, long long n, void* stream)
{
    if (!x0 || !y || n <= 0) return DC_ERR_INVALID;
    hipLaunchKernelGGL(lincomb4_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x0, x1, x2, x3, a, b, c, d, y, n);
    return dc_launch_status();
}
extern "C" int dc_transpose_bf16(const void* src, void* dst, int batch, int R, int C, void* stream)
{
    if (!src || !dst || batch <= 0 || R <= 0 || C <= 0) return DC_ERR_INVALID;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(dc_cdiv(C, 32), dc_cdiv(R, 32), batch), dim3(32, 8), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (bf16_t*)dst, R, C);
    return dc_launch_status();
}
extern "C" int dc_vae_sample_latents(const float* moments, const float* noise, float* latents, float scale, int N, int C, int H, int W, void* stream)
{
    if (!moments || !noise || !latents) return DC_ERR_INVALID;
    const long long total = (long long)N * C * H * W;
    hipLaunchKernelGGL(vae_sample_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, moments, noise, latents, scale, C, (long long)H * W, total);
    return dc_launch_status();
}
extern "C" int dc_cfg_ddim_step(const float* eps, float* latents, void* model_in, const float* coef_dev, int* step_dev,
                                float guidance, int cfg, int B, int C, int H, int W, void* stream)
{
    if (!eps || !latents || !model_in || !coef_dev || !step_dev || B <= 0) return DC_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid_for((long long)B * C * H * W)), dim3(256), 0, st, eps, latents, (bf16_t*)model_in,
                       coef_dev, step_dev, guidance, cfg, B, C, (long long)H * W);
    hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(1), 0, st, step_dev);
    return dc_launch_status();
}
extern "C" int dc_cfg_unipc_step(const float* eps, float* latents, float* m0, float* m1, float* last, void* model_in,
                                 const float* coef_dev, int* step_dev, float guidance, int cfg, int B, int C, int H, int W, void* stream)
{
    if (!eps || !latents || !m0 || !m1 || !last || !model_in || !coef_dev || !step_dev || B <= 0) return DC_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cfg_unipc_kernel, dim3(grid_for((long long)B * C * H * W)), dim3(256), 0, st, eps, latents, m0, m1, last,
                       (bf16_t*)model_in, coef_dev, step_dev, guidance, cfg, B, C, (long long)H * W);
    hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(1), 0, st, step_dev);
    return dc_launch_status();
}
extern "C" int dc_latents_to_model_input(const float* latents, void* model_in, float mul, int rep, int B, int C, int H, int W, void* stream)
{
    if (!latents || !model_in || rep <= 0) return DC_ERR_INVALID;
    hipLaunchKernelGGL(latents_to_model_input_kernel, dim3(grid_for((long long)B * C * H * W)), dim3(256), 0, (hipStream_t)stream,
                       latents, (bf16_t*)model_in, mul, rep, B, C, (long long)H * W);
    return dc_launch_status();
}
extern "C" int dc_postprocess_image(const float* x, float* out_nchw_f32, uint8_t* out_nhwc_u8, int N, int C, int H, int W,
                                    int x_pixel_stride, void* stream)
{
    if (x_pixel_stride == 0) x_pixel_stride = C;
    if (x_pixel_stride < C) return DC_ERR_INVALID;
    if (!x || (!out_nchw_f32 && !out_nhwc_u8)) return DC_ERR_INVALID;
    const long long total = (long long)N * C * H * W;
    hipLaunchKernelGGL(postprocess_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, out_nchw_f32, out_nhwc_u8, C, (long long)H * W, total, x_pixel_stride);
    return dc_launch_status();
}

extern "C" int dc_flow_hw2_resize_scale_f32(const float* src_hw2, int H, int W, float* dst_2hw, int th, int tw, void* stream)
{
    if (!src_hw2 || !dst_2hw || H <= 0 || W <= 0 || th <= 0 || tw <= 0) return DC_ERR_INVALID;
    hipLaunchKernelGGL(flow_hw2_resize_kernel, dim3(grid_for((long long)th * tw)), dim3(256), 0, (hipStream_t)stream, src_hw2, H, W,
                       dst_2hw, th, tw, (float)((double)tw / (double)W), (float)((double)th / (double)H),
                       // source-index scales in IEEE fp32 on the host (ATen's area_pixel_compute_scale<float>): a 1-ulp
                       // error here is amplified by the pixel index
                       th > 1 ? (float)(H - 1) / (float)(th - 1) : 0.f, tw > 1 ? (float)(W - 1) / (float)(tw - 1) : 0.f);
    return dc_launch_status();
}

extern "C" int dc_pack_sixch_u8_f32(const uint8_t* img0_hw3, const uint8_t* img1_hw3, float* dst_6hw, int H, int W, void* stream)
{
    if (!img0_hw3 || !img1_hw3 || !dst_6hw || H <= 0 || W <= 0) return DC_ERR_INVALID;
    hipLaunchKernelGGL(pack_sixch_kernel, dim3(grid_for((long long)H * W)), dim3(256), 0, (hipStream_t)stream, img0_hw3, img1_hw3,
                       dst_6hw, (long long)H * W);
    return dc_launch_status();
}

extern "C" int dc_blend_tiles_ramp_u8(const float* tiles_nchw, const int* coords_dev, int T, int C, int th, int tw,
                                      const float* ramp_dev, int feather, uint8_t* out_hwc, int H, int W, float scale, void* stream)
{
    if (!tiles_nchw || !coords_dev || !out_hwc || T <= 0 || C <= 0 || C > 4 || th <= 0 || tw <= 0 || H <= 0 || W <= 0 || feather < 0 ||
        (feather > 0 && !ramp_dev) || 2 * feather > th || 2 * feather > tw)
        return DC_ERR_INVALID;
    hipLaunchKernelGGL(blend_tiles_ramp_kernel, dim3(grid_for((long long)H * W)), dim3(256), 0, (hipStream_t)stream, tiles_nchw,
                       coords_dev, T, C, th, tw, ramp_dev, feather, out_hwc, H, W, scale);
    return dc_launch_status();
}
