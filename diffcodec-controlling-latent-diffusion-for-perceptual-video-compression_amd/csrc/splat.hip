// Forward (summation) splatting for gfx950, fp32 NCHW — the one native kernel the reference owns
// (`softsplat_out`, controlnet/softsplat.py:285-335) plus the wrapper math around it:
//   'soft' mode (softsplat.py:246-247,253-270), compute_mask (control_utils.py:11-17),
//   FeatureWarperSoftsplat's mask multiply (control_utils.py:69-70), flow resize+normalise
//   (control_utils.py:74-97) and the confidence fusion / hole fill of extractors.py:297-310.
//
// Round 2: a DETERMINISTIC GATHER instead of the reference's float-atomic scatter.  A source pixel p lands at
// (x + fx, y + fy) and feeds the four pixels around that point; turned around, a target pixel t is fed by the sources whose
// north-west corner cell floor(x + fx, y + fy) is t, t - (1,0), t - (0,1) or t - (1,1).  So:
//   1. bin_count : cell of every source (or -1: non-finite / no corner inside the map), integer count per cell
//   2. bin_scan  : exclusive scan of the counts per image (one workgroup per image) -> cell starts
//   3. bin_fill  : sources written into their cell's segment (arrival order) ...
//   4. bin_rank  : ... and put in ascending source order (rank = number of smaller indices in the segment)
//   5. gather    : one thread per output element walks the four segments as a 4-way merge by ascending source index and
//                  accumulates in · e · w and e · w with un-contracted fp32 multiplies and adds.
// Integer atomics only (their result does not depend on arrival order), so the output is bit-identical from run to run, and the
// accumulation order per target is ascending source raster order — exactly the order of the sequential CPU restatement the
// tests check against, which 'sum' mode therefore reproduces bit for bit.  exp(metric) and the normaliser channel are
// produced on the fly (the reference's cat[in*exp(m), exp(m)] tensor is never materialised); normalise, mask multiply and the
// occlusion test are fused into the gather.  HBM-bound, and small next to the diffusion loop (2 ms per 16-frame step).
#include "dc_common.h"
#include "../../include/diffcodec_hip.h"

// In the gather every fp32 product and sum is rounded on its own, like the reference kernel's and the oracle's.  hipcc's default
// contraction turns `num + in * w` into an fma — HIP's __fmul_rn / __fadd_rn are plain operators and do not prevent it, and
// `#pragma clang fp contract(off)` did not reach the inlined lambda bodies — so every product that feeds an addition passes through
// `rounded()`, an empty asm the optimiser cannot look through.
__device__ __forceinline__ float rounded(float v)
{
    asm volatile("" : "+v"(v));
    return v;
}

namespace {

struct BinWs {
    int* cell;      // [N][HW]     cell of source p, or -1
    int* counts;    // [N][cells+1] per-cell count, reused as the fill cursor
    int* starts;    // [N][cells+1] exclusive scan
    int* sorted;    // [N][HW]     sources by cell, arrival order
    int* sorted2;   // [N][HW]     sources by cell, ascending
};
inline long long bin_cells(int H, int W) { return (long long)(H + 1) * (W + 1); }
inline long long bin_ws_ints(int N, int H, int W) { return (long long)N * (3LL * H * W + 2 * (bin_cells(H, W) + 1)); }
inline BinWs bin_carve(void* ws, int N, int H, int W)
{
    const long long hw = (long long)H * W, cp = bin_cells(H, W) + 1;
    int* p = (int*)ws;
    BinWs b;
    b.cell = p, p += N * hw;
    b.counts = p, p += N * cp;
    b.starts = p, p += N * cp;
    b.sorted = p, p += N * hw;
    b.sorted2 = p;
    return b;
}

// landing point of source (x, y): softsplat.py:298-299
__device__ __forceinline__ void landing(const float* __restrict__ flow, long long n, long long hw, long long p, int x, int y, float& fx,
                                        float& fy)
{
    fx = (float)x + flow[(n * 2 + 0) * hw + p];
    fy = (float)y + flow[(n * 2 + 1) * hw + p];
}

__global__ __launch_bounds__(256) void bin_count_kernel(const float* __restrict__ flow, int* __restrict__ cell_of, int* __restrict__ counts,
                                                        int N, int H, int W)
{
    const long long hw = (long long)H * W, cp = (long long)(H + 1) * (W + 1) + 1;
    const long long total = (long long)N * hw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / hw, p = i - n * hw;
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        float fx, fy;
        landing(flow, n, hw, p, x, y, fx, fy);
        int cell = -1;
        // finite (softsplat.py:301-302) and at least one of the four corners inside the map: floor in [-1, W-1] x [-1, H-1]
        if (isfinite(fx) && isfinite(fy) && fx >= -1.0f && fx < (float)W && fy >= -1.0f && fy < (float)H) {
            const int nwx = (int)floorf(fx), nwy = (int)floorf(fy);
            cell = (nwy + 1) * (W + 1) + nwx + 1;
            atomicAdd(counts + n * cp + cell, 1);
        }
        cell_of[i] = cell;
    }
}

// exclusive scan of one image's counts (cells + 1 entries, the last one receives the total); the counts are zeroed for the fill
__global__ __launch_bounds__(256) void bin_scan_kernel(int* __restrict__ counts, int* __restrict__ starts, long long cp)
{
    __shared__ int part[256];
    int* c = counts + blockIdx.x * cp;
    int* s = starts + blockIdx.x * cp;
    const long long per = (cp + 255) / 256;
    const long long lo = min(cp, per * threadIdx.x), hi = min(cp, lo + per);
    int sum = 0;
    for (long long i = lo; i < hi; ++i) sum += c[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {                   // Hillis-Steele inclusive scan of the 256 partials
        const int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (long long i = lo; i < hi; ++i) {
        const int v = c[i];
        s[i] = run;
        run += v;
        c[i] = 0;
    }
}

__global__ __launch_bounds__(256) void bin_fill_kernel(const int* __restrict__ cell_of, int* __restrict__ cursor, const int* __restrict__ starts,
                                                       int* __restrict__ sorted, int N, long long hw, long long cp)
{
    const long long total = (long long)N * hw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cell = cell_of[i];
        if (cell < 0) continue;
        const long long n = i / hw;
        const int slot = starts[n * cp + cell] + atomicAdd(cursor + n * cp + cell, 1);
        sorted[n * hw + slot] = (int)(i - n * hw);
    }
}

// ascending order inside every cell segment: a source's final slot = segment start + number of smaller indices in the segment
// (segments hold one or two sources for smooth flow; a pathological flow that sends k sources to one cell costs k^2 compares)
__global__ __launch_bounds__(256) void bin_rank_kernel(const int* __restrict__ cell_of, const int* __restrict__ starts,
                                                       const int* __restrict__ sorted, int* __restrict__ sorted2, int N, long long hw,
                                                       long long cp)
{
    const long long total = (long long)N * hw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cell = cell_of[i];
        if (cell < 0) continue;
        const long long n = i / hw;
        const int p = (int)(i - n * hw);
        const int s = starts[n * cp + cell], e = starts[n * cp + cell + 1];
        int rank = 0;
        for (int j = s; j < e; ++j) rank += sorted[n * hw + j] < p ? 1 : 0;
        sorted2[n * hw + s + rank] = p;
    }
}

// The contributions to target (x, y) of image n, in ascending source order: f(p, weight).  Lists: cells (x,y) [source's NW corner],
// (x-1,y) [NE], (x,y-1) [SW], (x-1,y-1) [SE]; weights as softsplat.py:315-318, fp32, no contraction.
template <class F>
__device__ __forceinline__ void for_each_source(const float* __restrict__ flow, const int* __restrict__ starts, const int* __restrict__ sorted2,
                                                long long n, int x, int y, int H, int W, F&& f)
{
    const long long hw = (long long)H * W, cp = (long long)(H + 1) * (W + 1) + 1;
    const int* st = starts + n * cp;
    const int* so = sorted2 + n * hw;
    int idx[4], end[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int qx = x - (k & 1), qy = y - (k >> 1);            // north-west corner cell of the sources in list k
        const int cell = (qy + 1) * (W + 1) + qx + 1;
        idx[k] = st[cell];
        end[k] = st[cell + 1];
    }
    for (;;) {
        int best = -1, bp = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (idx[k] < end[k]) {
                const int p = so[idx[k]];
                if (p < bp) bp = p, best = k;
            }
        if (best < 0) break;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k == best) ++idx[k];
        const int sy = bp / W, sx = bp - sy * W;
        float fx, fy;
        landing(flow, n, hw, bp, sx, sy, fx, fy);
        const int nwx = (int)floorf(fx), nwy = (int)floorf(fy);
        const float ax = (float)(nwx + 1) - fx, bx = fx - (float)nwx;                          // (sex - fx), (fx - nwx)
        const float ay = (float)(nwy + 1) - fy, by = fy - (float)nwy;                          // (sey - fy), (fy - nwy)
        const float w = rounded(((best & 1) ? bx : ax) * ((best >> 1) ? by : ay));             // nw: ax*ay  ne: bx*ay  sw: ax*by  se: bx*by
        f(bp, w);
    }
}

// MODE 0: 'sum' (out = sum in*w) ; 1: 'soft' (out = sum in*e*w / (sum e*w + 1e-7) [* (1 - mask)])
template <int MODE>
__global__ __launch_bounds__(256) void splat_gather_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                                           const float* __restrict__ metric, const float* __restrict__ mask,
                                                           const int* __restrict__ starts, const int* __restrict__ sorted2,
                                                           float* __restrict__ out, int N, int C, int H, int W)
{
    const long long hw = (long long)H * W;
    const long long total = (long long)N * C * hw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i % hw;
        const int c = (int)((i / hw) % C);
        const long long n = i / (hw * C);
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        const float* src = in + (n * C + c) * hw;
        float num = 0.f, den = 0.f;
        for_each_source(flow, starts, sorted2, n, x, y, H, W, [&](int sp, float w) {
            if (MODE == 1) {
                const float e = expf(metric[n * hw + sp]);                        // softsplat.py:246-247
                num += rounded(rounded(src[sp] * e) * w);
                den += rounded(e * w);
            } else {
                num += rounded(src[sp] * w);
            }
        });
        float v = num;
        if (MODE == 1) {
            v = num / (den + 0.0000001f);                                         // softsplat.py:256-257,270
            if (mask) v = v * (1.0f - mask[n * hw + p]);                          // control_utils.py:69-70
        }
        out[i] = v;
    }
}

// occ = (|| b + softsplat(a, b, ones, 'soft') ||_2 > 0.3)   control_utils.py:11-17 ; metric = ones -> e = exp(1) for every source
__global__ __launch_bounds__(256) void occlusion_gather_kernel(const float* __restrict__ fa, const float* __restrict__ fb,
                                                               const int* __restrict__ starts, const int* __restrict__ sorted2,
                                                               float* __restrict__ mask, int N, int H, int W, float e)
{
    const long long hw = (long long)H * W;
    const long long total = (long long)N * hw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i % hw, n = i / hw;
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        float nx = 0.f, ny = 0.f, den = 0.f;
        for_each_source(fb, starts, sorted2, n, x, y, H, W, [&](int sp, float w) {
            nx += rounded(rounded(fa[(n * 2 + 0) * hw + sp] * e) * w);
            ny += rounded(rounded(fa[(n * 2 + 1) * hw + sp] * e) * w);
            den += rounded(e * w);
        });
        den += 0.0000001f;
        const float dx = fb[(n * 2 + 0) * hw + p] + rounded(nx / den);
        const float dy = fb[(n * 2 + 1) * hw + p] + rounded(ny / den);
        mask[i] = sqrtf(rounded(dx * dx) + rounded(dy * dy)) > 0.3f ? 1.0f : 0.0f;
    }
}

inline int grid_for(long long total) { return (int)min((long long)8192, (total + 255) / 256); }

// steps 1-4 for the flow field `flow` [N,2,H,W]
int build_bins(const float* flow, const BinWs& b, int N, int H, int W, hipStream_t st)
{
    const long long hw = (long long)H * W, cp = bin_cells(H, W) + 1;
    if (hipMemsetAsync(b.counts, 0, (size_t)(N * cp) * 4, st) != hipSuccess) return DC_ERR_LAUNCH;
    const int g = grid_for((long long)N * hw);
    hipLaunchKernelGGL(bin_count_kernel, dim3(g), dim3(256), 0, st, flow, b.cell, b.counts, N, H, W);
    hipLaunchKernelGGL(bin_scan_kernel, dim3(N), dim3(256), 0, st, b.counts, b.starts, cp);
    hipLaunchKernelGGL(bin_fill_kernel, dim3(g), dim3(256), 0, st, b.cell, b.counts, b.starts, b.sorted, N, hw, cp);
    hipLaunchKernelGGL(bin_rank_kernel, dim3(g), dim3(256), 0, st, b.cell, b.starts, b.sorted, b.sorted2, N, hw, cp);
    return DC_OK;
}

// F.interpolate(bilinear, align_corners=False) + per-component division  (control_utils.py:87-96)
__global__ __launch_bounds__(256) void flow_resize_kernel(const float* __restrict__ src, long long sbs, float* __restrict__ dst,
                                                          int N, int H, int W, int h, int w, float norm_w, float norm_h)
{
    const long long total = (long long)N * 2 * h * w;
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % w), oy = (int)((i / w) % h);
        const int c = (int)((i / ((long long)w * h)) % 2);
        const long long n = i / ((long long)w * h * 2);
        float fy = sy * ((float)oy + 0.5f) - 0.5f;
        float fx = sx * ((float)ox + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float hy = 1.0f - ly, hx = 1.0f - lx;
        const float* s = src + n * sbs + (long long)c * H * W;
        const float v = hy * (hx * s[(long long)y0 * W + x0] + lx * s[(long long)y0 * W + x1]) +
                        ly * (hx * s[(long long)y1 * W + x0] + lx * s[(long long)y1 * W + x1]);
        dst[i] = v / (c == 0 ? norm_w : norm_h);
    }
}

// extractors.py:297-310
__global__ __launch_bounds__(256) void fuse_kernel(const float* __restrict__ wf, const float* __restrict__ wl,
                                                   const float* __restrict__ cf, const float* __restrict__ cb,
                                                   const float* __restrict__ of, const float* __restrict__ ob,
                                                   float* __restrict__ fused, int N, int C, long long hw)
{
    const long long total = (long long)N * C * hw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i % hw;
        const long long n = i / (hw * C);
        const float a = fmaxf(cf[n * hw + p], 0.f), b = fmaxf(cb[n * hw + p], 0.f);
        const float wsum = (a + b) + 1e-6f;
        float v = (a / wsum) * wf[i] + (b / wsum) * wl[i];
        if (of && of[n * hw + p] + ob[n * hw + p] > 1.5f) v = 0.5f * (wf[i] + wl[i]);
        fused[i] = v;
    }
}

}  // namespace

extern "C" long long dc_splat_ws_bytes(int N, int H, int W)
{
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return bin_ws_ints(N, H, W) * 4;
}

extern "C" int dc_splat_sum_f32(const float* in, const float* flow, float* out, void* ws, int N, int C, int H, int W, void* stream)
{
    if (!in || !flow || !out || !ws || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DC_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const BinWs b = bin_carve(ws, N, H, W);
    if (int rc = build_bins(flow, b, N, H, W, st)) return rc;
    hipLaunchKernelGGL(splat_gather_kernel<0>, dim3(grid_for((long long)N * C * H * W)), dim3(256), 0, st, in, flow,
                       (const float*)nullptr, (const float*)nullptr, b.starts, b.sorted2, out, N, C, H, W);
    return dc_launch_status();
}

extern "C" int dc_splat_soft_f32(const float* in, const float* flow, const float* metric, const float* mask, float* out,
                                 void* ws, int N, int C, int H, int W, void* stream)
{
    if (!in || !flow || !metric || !out || !ws || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DC_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const BinWs b = bin_carve(ws, N, H, W);
    if (int rc = build_bins(flow, b, N, H, W, st)) return rc;
    hipLaunchKernelGGL(splat_gather_kernel<1>, dim3(grid_for((long long)N * C * H * W)), dim3(256), 0, st, in, flow, metric, mask,
                       b.starts, b.sorted2, out, N, C, H, W);
    return dc_launch_status();
}

extern "C" int dc_occlusion_mask_f32(const float* flow_a, const float* flow_b, float* mask_out, void* ws, int N,
                                     int H, int W, void* stream)
{
    if (!flow_a || !flow_b || !mask_out || !ws || N <= 0 || H <= 0 || W <= 0) return DC_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const BinWs b = bin_carve(ws, N, H, W);
    if (int rc = build_bins(flow_b, b, N, H, W, st)) return rc;
    hipLaunchKernelGGL(occlusion_gather_kernel, dim3(grid_for((long long)N * H * W)), dim3(256), 0, st, flow_a, flow_b, b.starts,
                       b.sorted2, mask_out, N, H, W, expf(1.0f));                // metric = ones (control_utils.py:12)
    return dc_launch_status();
}

extern "C" int dc_flow_resize_normalize_f32(const float* src, long long src_batch_stride, float* dst, int N, int H,
                                            int W, int h, int w, void* stream)
{
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || h <= 1 || w <= 1) return DC_ERR_INVALID;
    hipLaunchKernelGGL(flow_resize_kernel, dim3(grid_for((long long)N * 2 * h * w)), dim3(256), 0, (hipStream_t)stream, src,
                       src_batch_stride, dst, N, H, W, h, w, (float)(w - 1) / 2.0f, (float)(h - 1) / 2.0f);
    return dc_launch_status();
}

extern "C" int dc_flow_resize_divide_f32(const float* src, long long src_batch_stride, float* dst, int N, int H, int W,
                                         int h, int w, float div_x, float div_y, void* stream)
{
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0 || div_x == 0.f || div_y == 0.f) return DC_ERR_INVALID;
    hipLaunchKernelGGL(flow_resize_kernel, dim3(grid_for((long long)N * 2 * h * w)), dim3(256), 0, (hipStream_t)stream, src,
                       src_batch_stride, dst, N, H, W, h, w, div_x, div_y);
    return dc_launch_status();
}

extern "C" int dc_fuse_warped_f32(const float* warped_first, const float* warped_last, const float* conf_f,
                                  const float* conf_b, const float* occ_f, const float* occ_b, float* fused, int N,
                                  int C, int H, int W, void* stream)
{
    if (!warped_first || !warped_last || !conf_f || !conf_b || !fused || (!occ_f != !occ_b)) return DC_ERR_INVALID;
    hipLaunchKernelGGL(fuse_kernel, dim3(grid_for((long long)N * C * H * W)), dim3(256), 0, (hipStream_t)stream,
                       warped_first, warped_last, conf_f, conf_b, occ_f, occ_b, fused, N, C, (long long)H * W);
    return dc_launch_status();
}
