// Micro-benchmark: issue rates of the VALU instructions the attention softmax is made of (gfx950), one wave per SIMD
// and four waves per SIMD.  Prints cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define REP 64
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (OP == 0) { a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
                           a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7); }
            if (OP == 1) { a0 = fmaf(a0, seed, seed); a1 = fmaf(a1, seed, seed); a2 = fmaf(a2, seed, seed); a3 = fmaf(a3, seed, seed);
                           a4 = fmaf(a4, seed, seed); a5 = fmaf(a5, seed, seed); a6 = fmaf(a6, seed, seed); a7 = fmaf(a7, seed, seed); }
            if (OP == 2) { const f32x2 ss = {seed, seed}; p0 = p0 * ss + ss; p1 = p1 * ss + ss; p2 = p2 * ss + ss; p3 = p3 * ss + ss;
                           p0 = p0 * ss + ss; p1 = p1 * ss + ss; p2 = p2 * ss + ss; p3 = p3 * ss + ss; }
            if (OP == 3) { a0 = fmaxf(fmaxf(a0, seed), 0.1f); a1 = fmaxf(fmaxf(a1, seed), 0.1f); a2 = fmaxf(fmaxf(a2, seed), 0.1f); a3 = fmaxf(fmaxf(a3, seed), 0.1f);
                           a4 = fmaxf(fmaxf(a4, seed), 0.1f); a5 = fmaxf(fmaxf(a5, seed), 0.1f); a6 = fmaxf(fmaxf(a6, seed), 0.1f); a7 = fmaxf(fmaxf(a7, seed), 0.1f); }
            if (OP == 4) { a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
                           a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7); }
            if (OP == 5) { const f32x2 ss = {seed, seed}; p0 = p0 + ss; p1 = p1 + ss; p2 = p2 + ss; p3 = p3 + ss; p0 = p0 + ss; p1 = p1 + ss; p2 = p2 + ss; p3 = p3 + ss; }
        }
    }
    const long t1 = __builtin_amdgcn_s_memtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0) / (float)(iters * REP);
}
int main() {
    float* out; hipMalloc(&out, (1 << 22) + 64);
    const char* names[] = {"v_exp_f32", "v_fma_f32", "v_pk_fma_f32", "v_max3_f32", "v_rcp_f32", "v_pk_add_f32"};
    for (int occ = 1; occ <= 4; occ *= 2)
        for (int op = 0; op < 6; ++op) {
            void (*f)(float*, int, float) = op == 0 ? k<0> : op == 1 ? k<1> : op == 2 ? k<2> : op == 3 ? k<3> : op == 4 ? k<4> : k<5>;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(f, dim3(256 * occ), dim3(256), 0, 0, out, 2000, 0.5f);
            hipEventRecord(e0);
            hipLaunchKernelGGL(f, dim3(256 * occ), dim3(256), 0, 0, out, 2000, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            float c; hipMemcpy(&c, out + (1 << 20), 4, hipMemcpyDeviceToHost);
            printf("waves/SIMD %d  %-14s %6.2f memtime ticks per wave-instr (one wave); kernel %7.1f us -> %5.2f ns per wave-instr per SIMD\n",
                   occ, names[op], c, ms * 1e3, ms * 1e6 / (2000.0 * REP * occ));
        }
    return 0;
}
