// Micro-benchmark: do MFMA and VALU instructions of one SIMD overlap?  (gfx950)
// mode 0: MFMA only, 1: VALU only (v_fma / v_exp mix), 2: both interleaved in ONE wave, 3: waves alternate (even waves MFMA, odd VALU)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE, int NV>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed * i); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + threadIdx.x + i;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave >> 2) == 0);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave >> 2) == 1);
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int r = 0; r < NV; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], seed, seed);
        }
        if (do_m) {
            c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c7, 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int r = 0; r < NV; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], seed, seed);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    const f32x4 cs = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    out[blockIdx.x * 512 + threadIdx.x] = s + cs[0] + cs[1] + cs[2] + cs[3];
}
template <int MODE, int NV> float run(float* out, int blocks, int threads = 256) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, out, 4000, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, out, 4000, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    // per iteration: 8 MFMA (8 x 16 = 128 pipe cycles) and 2*NV*16 v_fma (2 cycles each)
    printf("1 wave/SIMD : MFMA only %7.1f us | VALU only (64 fma/iter) %7.1f us | both, one wave %7.1f us\n",
           run<0, 2>(out, 256), run<1, 2>(out, 256), run<2, 2>(out, 256));
    printf("2 waves/SIMD (512-thread blocks): MFMA only %7.1f us | VALU only %7.1f us | both in every wave %7.1f us | waves 0-3 MFMA + waves 4-7 VALU %7.1f us\n",
           run<0, 2>(out, 256, 512), run<1, 2>(out, 256, 512), run<2, 2>(out, 256, 512), run<3, 2>(out, 256, 512));
    printf("1 wave/SIMD, 128 fma/iter: VALU only %7.1f us | both %7.1f us\n", run<1, 4>(out, 256), run<2, 4>(out, 256));
    return 0;
}
