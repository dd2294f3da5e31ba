// Developer micro-experiment (not part of the product): do a wave issuing MFMAs and a co-resident wave of the SAME SIMD issuing VALU
// (v_exp_f32 / v_fma_f32) overlap, or do their times add?  512-thread workgroups (two waves per SIMD), one workgroup per CU.
//   mode 1: waves 0-3 run the MFMA loop, waves 4-7 idle      mode 2: waves 4-7 run the VALU loop, waves 0-3 idle      mode 3: both
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip ; run: ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int VKIND>
__global__ __launch_bounds__(512, 2) void k(float* out, int mode, int iters)
{
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
            bf16x8 x, y;
            for (int i = 0; i < 8; ++i) x[i] = (__bf16)(float)(threadIdx.x & 7), y[i] = (__bf16)1.0f;
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else if (mode & 2) {
        if (mode & 4) __builtin_amdgcn_s_setprio(2);           // mode bit 2: the VALU waves run at raised priority
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 1e-3f + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {                     // 16 independent VALU instructions per iteration
                if (VKIND == 0) v[j] = __builtin_amdgcn_exp2f(v[j]);
                else v[j] = v[j] * 1.0001f + 0.5f;
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) r += v[j];
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int VKIND>
void run(const char* name)
{
    float* d;
    hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int iters = 20000;
    float ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int mode = 1; mode <= 7; ++mode) {
        if (mode > 3 && mode != 7) continue;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<VKIND>, dim3(256), dim3(512), 0, 0, d, mode, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[mode], e0, e1);
        }
    }
    printf("%s: MFMA alone %.3f ms (4 x %d MFMA 32x32x16 per wave) | VALU alone %.3f ms (16 x %d per wave) | both %.3f ms | both, VALU waves at s_setprio 2: %.3f ms | sum %.3f max %.3f\n", name,
           ms[1], iters, ms[2], iters, ms[3], ms[7], ms[1] + ms[2], ms[1] > ms[2] ? ms[1] : ms[2]);
}

int main()
{
    run<0>("v_exp_f32");
    run<1>("v_fma_f32");
    return 0;
}
