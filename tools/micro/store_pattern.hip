// Micro-benchmark: per-CU store throughput of the GEMM epilogue's access patterns (gfx950).
//   pattern 0: one buffer_store_dwordx4 covers 16 rows x 64 B   (persistent kernel's epilogue today)
//   pattern 1: one buffer_store_dwordx4 covers  8 rows x 128 B  (full cache lines)
//   pattern 2: one store covers 4 rows x 256 B
// Each 512-thread block writes 256 x 256 bf16 tiles of a [M, N] matrix, `reps` tiles per block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// pattern 3: dwordx2 stores, 16 consecutive lanes cover one 128-B row segment, 4 rows per instruction (32 per wave tile)
template <int AUX>
__global__ __launch_bounds__(512) void store2_kernel(char* C, long ldc_b, int tiles_n, int ntiles) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tm = t / tiles_n, tn = t % tiles_n;
        char* base = C + (long)tm * 256 * ldc_b + (long)tn * 512 + (long)wm * 128 * ldc_b + wn * 128;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, 0x7fffffff, 0x00020000);
        const u32x2 v = {(unsigned)t, (unsigned)lane};
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            const int row = (it >> 2) * 16 + (lane >> 4) * 4 + (it & 3), colb = (lane & 15) * 8;
            __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)(row * ldc_b + colb), 0, AUX);
        }
    }
}

template <int PAT, int AUX>
__global__ __launch_bounds__(512) void store_kernel(char* C, long ldc_b, int tiles_n, int ntiles) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tm = t / tiles_n, tn = t % tiles_n;
        char* base = C + (long)tm * 256 * ldc_b + (long)tn * 512 + (long)wm * 128 * ldc_b + wn * 128;   // wave tile 128 rows x 64 cols (128 B)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, 0x7fffffff, 0x00020000);
        const u32x4 v = {(unsigned)t, (unsigned)lane, 3u, 4u};
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            int row, colb;
            if (PAT == 0) { row = (it >> 1) * 16 + (lane & 15); colb = (it & 1) * 64 + (lane >> 4) * 16; }
            else if (PAT == 1) { row = it * 8 + (lane & 7); colb = (lane >> 3) * 16; }
            else { row = it * 8 + (lane >> 3); colb = (lane & 7) * 16; }
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(row * ldc_b + colb), 0, AUX);
        }
    }
}

int main(int argc, char** argv) {
    const int M = 87680, N = 3072;
    const long ldc_b = (long)N * 2;
    char* C;
    hipMalloc(&C, (size_t)M * ldc_b + (1 << 20));
    const int tiles_n = N / 256, tiles_m = (M + 255) / 256 - 1, ntiles = tiles_m * tiles_n;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int nblk = argc > 1 ? atoi(argv[1]) : 256;
    auto run = [&](const char* name, void (*k)(char*, long, int, int)) {
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(nblk), dim3(512), 0, 0, C, ldc_b, tiles_n, ntiles);
        hipEventRecord(e0);
        for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k, dim3(nblk), dim3(512), 0, 0, C, ldc_b, tiles_n, ntiles);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double bytes = (double)ntiles * 256 * 512;
        printf("%-28s %8.1f us  %7.2f TB/s  %6.1f B/clk/CU@1.8GHz\n", name, ms * 1e3, bytes / ms / 1e9, bytes / nblk / (ms * 1e-3 * 1.8e9));
    };
    run("16 rows x 64 B, plain", store_kernel<0, 0>);
    run("16 rows x 64 B, sc1", store_kernel<0, 16>);
    run("16 rows x 64 B, nt", store_kernel<0, 2>);
    run("16 rows x 64 B, nt|sc1", store_kernel<0, 18>);
    run("8 rows x 128 B, plain", store_kernel<1, 0>);
    run("8 rows x 128 B, nt|sc1", store_kernel<1, 18>);
    run("8 rows x 128 B (lane-major), nt|sc1", store_kernel<2, 18>);
    run("8 rows x 128 B (lane-major), nt", store_kernel<2, 2>);
    run("dwordx2 4 rows x 128 B, nt|sc1", store2_kernel<18>);
    run("dwordx2 4 rows x 128 B, plain", store2_kernel<0>);
    return 0;
}
