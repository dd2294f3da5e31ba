// Shared device/host helpers for the gfx950 (CDNA4, wave64) kernels of the DiffCodec decode path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DC_OK 0
#define DC_ERR_INVALID (-1)
#define DC_ERR_LAUNCH (-2)

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define DC_WAVE 64

__device__ __forceinline__ float dc_bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t dc_f2bf(float v) { return (bf16_t)v; }   // v_cvt_pk_bf16_f32: RNE, NaN-preserving

__device__ __forceinline__ float dc_silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float dc_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float dc_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float dc_wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int dc_launch_status()
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DC_OK : DC_ERR_LAUNCH;
}
static inline int dc_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
