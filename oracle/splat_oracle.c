/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, not on the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Plain-C, single-thread CPU restatement of the reference's forward (summation) splatting kernel
 * and its 'soft' wrapper:
 *   - splat_sum_f32      follows controlnet/softsplat.py:285-335 (kernel `softsplat_out`)
 *   - splat_soft_f32     follows controlnet/softsplat.py:232-274 (mode 'soft': exp-weighted, +1e-7 normalise)
 *
 * Parity pin: the reference kernel is CUDA-only (softsplat.py:347-348 asserts on CPU tensors) and the
 * repo holds no golden vectors for it, so this restatement is pinned only through hand-computed
 * known-answer cases in tests/test_oracle_splat.py ("parity unpinned" w.r.t. a run of the CUDA kernel).
 *
 * Summation order here is source-pixel raster order (n, c, y, x); the CUDA kernel uses atomicAdd whose
 * order is undefined, so bit-equality with a GPU run is not a property of the reference itself.
 * All arithmetic is fp32 (softsplat.py:279 forces float32), with the same expression order as the kernel.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* tenIn [N,C,H,W], tenFlow [N,2,H,W], tenOut [N,C,H,W] (zero-filled here), contiguous NCHW fp32. */
void splat_sum_f32(const float* in, const float* flow, float* out, int N, int C, int H, int W)
{
    memset(out, 0, sizeof(float) * (size_t)N * C * H * W);
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    /* softsplat.py:298-299 */
                    float fx = (float)x + flow[(((size_t)n * 2 + 0) * H + y) * W + x];
                    float fy = (float)y + flow[(((size_t)n * 2 + 1) * H + y) * W + x];
                    /* softsplat.py:301-302 */
                    if (!isfinite(fx) || !isfinite(fy)) continue;
                    float v = in[(((size_t)n * C + c) * H + y) * W + x];
                    /* softsplat.py:306-313 */
                    int nwx = (int)floorf(fx), nwy = (int)floorf(fy);
                    int nex = nwx + 1, ney = nwy;
                    int swx = nwx, swy = nwy + 1;
                    int sex = nwx + 1, sey = nwy + 1;
                    /* softsplat.py:315-318 */
                    float wnw = ((float)sex - fx) * ((float)sey - fy);
                    float wne = (fx - (float)swx) * ((float)swy - fy);
                    float wsw = ((float)nex - fx) * (fy - (float)ney);
                    float wse = (fx - (float)nwx) * (fy - (float)nwy);
                    float* o = out + ((size_t)n * C + c) * H * W;
                    /* softsplat.py:320-334 */
                    if (nwx >= 0 && nwx < W && nwy >= 0 && nwy < H) o[(size_t)nwy * W + nwx] += v * wnw;
                    if (nex >= 0 && nex < W && ney >= 0 && ney < H) o[(size_t)ney * W + nex] += v * wne;
                    if (swx >= 0 && swx < W && swy >= 0 && swy < H) o[(size_t)swy * W + swx] += v * wsw;
                    if (sex >= 0 && sex < W && sey >= 0 && sey < H) o[(size_t)sey * W + sex] += v * wse;
                }
}

/* 'soft' mode (softsplat.py:246-247, 251-270): in' = cat[in*exp(metric), exp(metric)]; splat;
 * out = out'[:, :-1] / (out'[:, -1:] + 1e-7).   metric [N,1,H,W]; out [N,C,H,W]. */
void splat_soft_f32(const float* in, const float* flow, const float* metric, float* out,
                    int N, int C, int H, int W)
{
    size_t hw = (size_t)H * W;
    float* aug = (float*)malloc(sizeof(float) * (size_t)N * (C + 1) * hw);
    float* acc = (float*)malloc(sizeof(float) * (size_t)N * (C + 1) * hw);
    for (int n = 0; n < N; ++n) {
        for (size_t p = 0; p < hw; ++p) {
            float e = expf(metric[(size_t)n * hw + p]);
            for (int c = 0; c < C; ++c)
                aug[((size_t)n * (C + 1) + c) * hw + p] = in[((size_t)n * C + c) * hw + p] * e;
            aug[((size_t)n * (C + 1) + C) * hw + p] = e;
        }
    }
    splat_sum_f32(aug, flow, acc, N, C + 1, H, W);
    for (int n = 0; n < N; ++n)
        for (size_t p = 0; p < hw; ++p) {
            float den = acc[((size_t)n * (C + 1) + C) * hw + p] + 0.0000001f;
            for (int c = 0; c < C; ++c)
                out[((size_t)n * C + c) * hw + p] = acc[((size_t)n * (C + 1) + c) * hw + p] / den;
        }
    free(aug);
    free(acc);
}
