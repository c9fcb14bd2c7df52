// Probe of the cross-lane row maximum the attention forward uses (v_permlane16_swap + v_permlane32_swap, attention.hip quad_rows_max):
// the maximum over lanes {l, l^16, l^32, l^48}.  Written with by-value scalars (a0, a1) it is correct; with
// __builtin_bit_cast(float, a[1]) applied to the ELEMENT EXPRESSION of the builtin's ext-vector result, ROCm 7.2's clang reads element 0
// and the "maximum" is one lane group's value ("Range contract").  hipcc --offload-arch=gfx950 permlane_bitcast_probe.hip && ./a.out
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    float v = (float)(lane * 3 % 64) + 100.f * (lane >> 4);   // distinct values; max over {l, l^16, l^32, l^48}
    const unsigned u = __builtin_bit_cast(unsigned, v);
    unsigned u2 = u;
    asm volatile("" : "+v"(u2));
    const auto a = __builtin_amdgcn_permlane16_swap(u, u2, false, false);
    const unsigned a0 = a[0], a1 = a[1];
    out[lane] = __builtin_bit_cast(float, a0);
    out[64 + lane] = __builtin_bit_cast(float, a1);
    float m1 = fmaxf(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1));
    const unsigned w = __builtin_bit_cast(unsigned, m1);
    unsigned w2 = w;
    asm volatile("" : "+v"(w2));
    const auto b = __builtin_amdgcn_permlane32_swap(w, w2, false, false);
    const unsigned b0 = b[0], b1 = b[1];
    out[128 + lane] = fmaxf(__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1));
    out[192 + lane] = v;
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        float want = 0;
        for (int g = 0; g < 4; ++g) { float x = h[192 + ((l & 15) | (g << 4))]; if (x > want) want = x; }
        if (h[128 + l] != want) { ++bad; printf("lane %d quad max got %g want %g (a0 %g a1 %g v %g)\n", l, h[128 + l], want, h[l], h[64 + l], h[192 + l]); }
    }
    printf("quad_rows_max: %d bad lanes\n", bad);
    return 0;
}
