// CLIP text-encoder pieces of `encode_prompt` (reference pipeline.py:223-236 -> transformers CLIPTextModel):
// token + position embedding lookup and the causal self-attention over the 77-token context.  Both run once per
// prompt, outside the sampling loop; they are latency-trivial (77 tokens) and written for clarity, not for MFMA.
#include "dc_common.h"

namespace {

// out[b][t][:] = tok[ids[b][t]][:] + pos[t][:]   (CLIPTextEmbeddings.forward); one workgroup per token, 8 channels per lane
__global__ __launch_bounds__(128) void embed_tokens_kernel(const long long* __restrict__ ids, const bf16_t* __restrict__ tok,
                                                           const bf16_t* __restrict__ pos, bf16_t* __restrict__ out,
                                                           int T, int C, int vocab)
{
    const long long row = blockIdx.x;
    const int t = (int)(row % T);
    long long id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);           // host validates; clamp keeps a bad id from faulting
    for (int c = threadIdx.x * 8; c < C; c += blockDim.x * 8) {
        const bf16x8 a = *(const bf16x8*)(tok + id * C + c), p = *(const bf16x8*)(pos + (long long)t * C + c);
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = (bf16_t)((float)a[r] + (float)p[r]);
        *(bf16x8*)(out + row * C + c) = o;
    }
}

// Causal softmax(Q K^T * scale) V for short sequences (T <= 128, D <= 128): one workgroup per (batch, head); K rows
// padded by one float in LDS so that lane j reading row j is conflict-free; one wave per query row, 4 rows per pass.
constexpr int TMAX = 128, DMAX = 128;

__global__ __launch_bounds__(256) void attn_causal_small_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                const bf16_t* __restrict__ v, bf16_t* __restrict__ out,
                                                                int heads, int T, int D, long long qs, long long ks,
                                                                long long vs, long long os, float scale)
{
    extern __shared__ float lds[];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int DP = D + 1;
    float* Ks = lds;                      // [T][D+1]
    float* Vs = Ks + T * DP;              // [T][D]
    float* Qs = Vs + T * D;               // [4 waves][D]
    float* Ps = Qs + 4 * D;               // [4 waves][T]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < T * D; i += 256) {
        const int t = i / D, c = i - t * D;
        Ks[t * DP + c] = (float)k[((long long)b * T + t) * ks + h * D + c];
        Vs[t * D + c] = (float)v[((long long)b * T + t) * vs + h * D + c];
    }
    __syncthreads();
    for (int i0 = 0; i0 < T; i0 += 4) {                                 // uniform trip count: barriers inside
        const int i = min(i0 + wave, T - 1);                            // (tail waves redo the last row; same value stored)
        for (int c = lane; c < D; c += 64) Qs[wave * D + c] = (float)q[((long long)b * T + i) * qs + h * D + c] * scale;
        __syncthreads();
        float s[2], mx = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = lane + 64 * jj;
            float acc = -INFINITY;
            if (j <= i && j < T) {                                   // causal: key j attends iff j <= i
                acc = 0.f;
                for (int c = 0; c < D; ++c) acc += Qs[wave * D + c] * Ks[j * DP + c];
            }
            s[jj] = acc;
            mx = fmaxf(mx, acc);
        }
        mx = dc_wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = lane + 64 * jj;
            const float p = (j <= i && j < T) ? __expf(s[jj] - mx) : 0.f;
            if (j < T) Ps[wave * T + j] = p;
            sum += p;
        }
        sum = dc_wave_sum(sum);
        __syncthreads();
        const float inv = 1.0f / sum;
        for (int c = lane; c < D; c += 64) {
            float acc = 0.f;
            for (int j = 0; j <= i; ++j) acc += Ps[wave * T + j] * Vs[j * D + c];
            out[((long long)b * T + i) * os + h * D + c] = (bf16_t)(acc * inv);
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int dc_embed_tokens_bf16(const long long* ids, const void* tok_emb, const void* pos_emb, void* out, int B, int T,
                                    int C, int vocab, void* stream)
{
    if (!ids || !tok_emb || !pos_emb || !out || B <= 0 || T <= 0 || C <= 0 || (C & 7) || vocab <= 0) return DC_ERR_INVALID;
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(B * T), dim3(128), 0, (hipStream_t)stream, ids, (const bf16_t*)tok_emb,
                       (const bf16_t*)pos_emb, (bf16_t*)out, T, C, vocab);
    return dc_launch_status();
}

extern "C" int dc_attention_causal_small_bf16(const void* q, const void* k, const void* v, void* out, int B, int heads, int T,
                                              int D, long long q_stride, long long k_stride, long long v_stride,
                                              long long out_stride, float scale, void* stream)
{
    if (!q || !k || !v || !out || B <= 0 || heads <= 0 || T <= 0 || T > TMAX || D <= 0 || D > DMAX) return DC_ERR_INVALID;
    const size_t lds = sizeof(float) * ((size_t)T * (D + 1) + (size_t)T * D + 4 * D + 4 * T);
    auto kern = attn_causal_small_kernel;
    static std::atomic<unsigned long long> attr_done{0};
    dc_set_max_dyn_lds((const void*)kern, (int)(sizeof(float) * ((size_t)TMAX * (DMAX + 1) + (size_t)TMAX * DMAX + 4 * DMAX + 4 * TMAX)),
                       attr_done);
    hipLaunchKernelGGL(kern, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (bf16_t*)out, heads, T, D, q_stride, k_stride, v_stride, out_stride, scale);
    return dc_launch_status();
}
