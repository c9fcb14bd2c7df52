// Micro-benchmark: how many VALU instructions fit in the shadow of one MFMA, by MFMA shape (gfx950).
// Every instruction is its own `asm volatile` statement, so the issue order is exactly the program order:
//   [MFMA, F fillers] x 8 per loop iteration, independent accumulators round-robin, operands in registers (random-ish bits).
// SHAPE 0: v_mfma_f32_16x16x32_bf16 (16 pipe cycles), SHAPE 1: v_mfma_f32_32x32x16_bf16 (32 pipe cycles).
// FILL 0: v_fma_f32, 1: v_exp_f32, 2: v_cvt_pk_bf16_f32, 3: v_max3_f32.
// Prints s_memtime ticks (shader cycles) per MFMA for one wave and the wall time of the launch.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int FILL> __device__ __forceinline__ void filler(float& x, float s) {
    if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(s));
    if (FILL == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (FILL == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(s));
    if (FILL == 3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(s));
}

template <int SHAPE, int F, int FILL, int NT>
__global__ __launch_bounds__(NT) void k(float* out, int iters, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(seed * (float)((threadIdx.x * 7 + i * 13) % 31 - 15));
        b[i] = (__bf16)(seed * (float)((threadIdx.x * 5 + i * 11) % 29 - 14));
    }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * 0.01f * (float)(threadIdx.x + i);
    f32x4 c4[8];
    f32x16 c16[4];
    for (int i = 0; i < 8; ++i) c4[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) c16[i][j] = 0.f;
    const long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (SHAPE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c4[m]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c16[m & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < F; ++f) filler<FILL>(v[(m * F + f) & 7], seed);
        }
    }
    const long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i] + c4[i][0] + c4[i][1] + c4[i][2] + c4[i][3];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += c16[i][j];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0) / (float)(iters * 8);
}

template <int SHAPE, int F, int FILL, int NT> void run(float* out, const char* what) {
    const int iters = SHAPE == 0 ? 40000 : 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE, F, FILL, NT>), dim3(256), dim3(NT), 0, 0, out, iters, 0.37f);
    hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE, F, FILL, NT>), dim3(256), dim3(NT), 0, 0, out, iters, 0.37f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    float c; hipMemcpy(&c, out + (1 << 20), 4, hipMemcpyDeviceToHost);
    const double flop = (double)iters * 8 * (SHAPE == 0 ? 16384.0 : 32768.0) * (NT / 64) * 256;
    printf("%-9s waves/SIMD %d  %d x %-8s per MFMA: %6.1f ticks per MFMA (wave 0) | launch %8.1f us | %7.1f TF/s | clock ~%.2f GHz\n",
           SHAPE == 0 ? "16x16x32" : "32x32x16", NT / 256, F, what, c, ms * 1e3, flop / (ms * 1e-3) / 1e12,
           (double)c * iters * 8 / (ms * 1e-3) / 1e9);
}

int main() {
    float* out; hipMalloc(&out, (1 << 22) + 64);
#define ROW(SHAPE, FILL, NT, W)                                                                     \
    run<SHAPE, 0, FILL, NT>(out, W); run<SHAPE, 1, FILL, NT>(out, W); run<SHAPE, 2, FILL, NT>(out, W); \
    run<SHAPE, 3, FILL, NT>(out, W); run<SHAPE, 4, FILL, NT>(out, W); run<SHAPE, 6, FILL, NT>(out, W); \
    run<SHAPE, 8, FILL, NT>(out, W);
    ROW(0, 0, 256, "v_fma") ROW(1, 0, 256, "v_fma")
    ROW(0, 1, 256, "v_exp") ROW(1, 1, 256, "v_exp")
    ROW(0, 2, 256, "v_cvt_pk") ROW(1, 2, 256, "v_cvt_pk")
    ROW(0, 3, 256, "v_max3") ROW(1, 3, 256, "v_max3")
    ROW(0, 0, 512, "v_fma") ROW(1, 0, 512, "v_fma")
    ROW(0, 1, 512, "v_exp") ROW(1, 1, 512, "v_exp")
    return 0;
}
