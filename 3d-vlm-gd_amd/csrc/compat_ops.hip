// Stand-alone forms of the reference's loss helpers (gfx950) for callers that bind the reference's own function
// names (gd_amd/compat.py): temperature sigmoid (utils/functions.py:24-33), get_masked_patch_cost
// (utils/functions.py:402-422) and kl_divergence_map (utils/losses.py:5-15), each with its backward.
// The training step never runs these: there the same arithmetic is fused into smooth_ap_kernel and the cost-volume
// kernels and the hw x hw maps never exist.  All fp32, one 256-thread block per matrix row, coalesced row sweeps
// (rows of hw = 1369 floats are only 4-byte aligned, so the sweeps are dword accesses).
#include "gd_common.h"

__device__ __forceinline__ float cblock_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float cblock_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---- sigmoid(x, temp): exponent = clamp(-x / temp, -50, 50); y = 1 / (1 + exp(exponent)) ----
__global__ __launch_bounds__(256) void sigmoid_temp_kernel(const float* x, const float* dy, float* out, long n, float temp,
                                                           int backward) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float e0 = -x[i] / temp;
        const float e = fminf(fmaxf(e0, -50.f), 50.f);
        const float y = 1.0f / (1.0f + expf(e));
        if (!backward) out[i] = y;
        else out[i] = (e0 >= -50.f && e0 <= 50.f) ? dy[i] * y * (1.0f - y) / temp : 0.f;
    }
}

// ---- get_masked_patch_cost: rows / columns outside the masks are zeroed, then softmax(row / temperature) or
//      row / max(rowsum, eps) ----
__global__ __launch_bounds__(256) void masked_patch_cost_fwd_kernel(const float* cost, const unsigned char* m1,
                                                                    const unsigned char* m2, float* out, int R, int Ccols,
                                                                    float eps, int use_softmax, float inv_temp) {
    __shared__ float red[4];
    const int r = blockIdx.x, b = blockIdx.y;
    const float* src = cost + ((long)b * R + r) * Ccols;
    float* dst = out + ((long)b * R + r) * Ccols;
    const bool rm = m1[r] != 0;
    if (use_softmax) {
        float mx = -INFINITY;
        for (int j = threadIdx.x; j < Ccols; j += 256) {
            const float v = (rm && (!m2 || m2[j])) ? src[j] * inv_temp : 0.f;
            mx = fmaxf(mx, v);
        }
        mx = cblock_max(mx, red);
        float s = 0.f;
        for (int j = threadIdx.x; j < Ccols; j += 256) {
            const float v = (rm && (!m2 || m2[j])) ? src[j] * inv_temp : 0.f;
            s += expf(v - mx);
        }
        s = cblock_sum(s, red);
        const float inv = 1.0f / s;
        for (int j = threadIdx.x; j < Ccols; j += 256) {
            const float v = (rm && (!m2 || m2[j])) ? src[j] * inv_temp : 0.f;
            dst[j] = expf(v - mx) * inv;
        }
    } else {
        float s = 0.f;
        for (int j = threadIdx.x; j < Ccols; j += 256) s += (rm && (!m2 || m2[j])) ? src[j] : 0.f;
        s = fmaxf(cblock_sum(s, red), eps);
        for (int j = threadIdx.x; j < Ccols; j += 256) dst[j] = ((rm && (!m2 || m2[j])) ? src[j] : 0.f) / s;
    }
}

// y = the forward's output (saved), dy upstream -> dcost
__global__ __launch_bounds__(256) void masked_patch_cost_bwd_kernel(const float* cost, const float* y, const float* dy,
                                                                    const unsigned char* m1, const unsigned char* m2,
                                                                    float* dcost, int R, int Ccols, float eps, int use_softmax,
                                                                    float inv_temp) {
    __shared__ float red[4];
    const int r = blockIdx.x, b = blockIdx.y;
    const long off = ((long)b * R + r) * Ccols;
    const bool rm = m1[r] != 0;
    float dot = 0.f, s = 0.f;
    for (int j = threadIdx.x; j < Ccols; j += 256) {
        dot += dy[off + j] * y[off + j];
        s += (rm && (!m2 || m2[j])) ? cost[off + j] : 0.f;
    }
    dot = cblock_sum(dot, red);
    s = cblock_sum(s, red);
    for (int j = threadIdx.x; j < Ccols; j += 256) {
        const bool keep = rm && (!m2 || m2[j]);      // masked_cost[~mask] = 0 cuts the gradient there
        float g;
        if (use_softmax) g = y[off + j] * (dy[off + j] - dot) * inv_temp;
        else g = s > eps ? (dy[off + j] - dot) / s : dy[off + j] / eps;     // clamp_min passes no gradient once the clamp is active
        dcost[off + j] = keep ? g : 0.f;
    }
}

// ---- kl_divergence_map: mean over rows of sum_j t log(t / p), t = max(T, eps), p = max(P, eps) ----
__global__ __launch_bounds__(256) void kl_rows_kernel(const float* T, const float* Pm, float* row_ws, int Ccols, float eps) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    const float* t = T + row * Ccols;
    const float* p = Pm + row * Ccols;
    float s = 0.f;
    for (int j = threadIdx.x; j < Ccols; j += 256) {
        const float tt = fmaxf(t[j], eps), pp = fmaxf(p[j], eps);
        s += tt * logf(tt / pp);
    }
    s = cblock_sum(s, red);
    if (threadIdx.x == 0) row_ws[row] = s;
}
__global__ __launch_bounds__(256) void kl_reduce_kernel(const float* row_ws, float* loss, long rows) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = threadIdx.x; i < rows; i += 256) s += row_ws[i];      // fixed order: deterministic
    s = cblock_sum(s, red);
    if (threadIdx.x == 0) loss[0] = s / (float)rows;
}
__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* T, const float* Pm, const float* gloss, float* dT, float* dP,
                                                     long n, long rows, float eps) {
    const float g = gloss[0] / (float)rows;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float t0 = T[i], p0 = Pm[i];
        const float tt = fmaxf(t0, eps), pp = fmaxf(p0, eps);
        if (dP) dP[i] = p0 >= eps ? -g * tt / pp : 0.f;
        if (dT) dT[i] = t0 >= eps ? g * (logf(tt / pp) + 1.0f) : 0.f;
    }
}

static inline int c_blocks(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 65536 ? 65536 : b)); }

extern "C" int gd_sigmoid_temp(const float* x, const float* dy, float* out, long n, float temp, void* stream) {
    GD_REQUIRE(n > 0 && temp != 0.f, "gd_sigmoid_temp: bad arguments");
    hipLaunchKernelGGL(sigmoid_temp_kernel, dim3(c_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, dy, out, n, temp, dy ? 1 : 0);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_masked_patch_cost_fwd(const float* cost, const unsigned char* m1, const unsigned char* m2, float* out, int B,
                                        int rows, int cols, float eps, int use_softmax, float temperature, void* stream) {
    GD_REQUIRE(B > 0 && rows > 0 && cols > 0 && B <= 65535 && temperature != 0.f, "gd_masked_patch_cost_fwd: bad shape");
    hipLaunchKernelGGL(masked_patch_cost_fwd_kernel, dim3(rows, B), dim3(256), 0, (hipStream_t)stream, cost, m1, m2, out, rows,
                       cols, eps, use_softmax, 1.0f / temperature);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_masked_patch_cost_bwd(const float* cost, const float* y, const float* dy, const unsigned char* m1,
                                        const unsigned char* m2, float* dcost, int B, int rows, int cols, float eps,
                                        int use_softmax, float temperature, void* stream) {
    GD_REQUIRE(B > 0 && rows > 0 && cols > 0 && B <= 65535 && temperature != 0.f, "gd_masked_patch_cost_bwd: bad shape");
    hipLaunchKernelGGL(masked_patch_cost_bwd_kernel, dim3(rows, B), dim3(256), 0, (hipStream_t)stream, cost, y, dy, m1, m2, dcost,
                       rows, cols, eps, use_softmax, 1.0f / temperature);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_kl_divergence_map_fwd(const float* t, const float* p, long rows, int cols, float eps, float* loss,
                                        float* row_ws, void* stream) {
    GD_REQUIRE(rows > 0 && cols > 0 && rows < (1L << 31), "gd_kl_divergence_map_fwd: bad shape");
    hipLaunchKernelGGL(kl_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, t, p, row_ws, cols, eps);
    hipLaunchKernelGGL(kl_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, row_ws, loss, rows);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_kl_divergence_map_bwd(const float* t, const float* p, const float* gloss, long rows, int cols, float eps,
                                        float* dt, float* dp, void* stream) {
    GD_REQUIRE(rows > 0 && cols > 0 && (dt || dp), "gd_kl_divergence_map_bwd: bad arguments");
    const long n = rows * cols;
    hipLaunchKernelGGL(kl_bwd_kernel, dim3(c_blocks(n)), dim3(256), 0, (hipStream_t)stream, t, p, gloss, dt, dp, n, rows, eps);
    GD_LAUNCH_OK();
    return 0;
}
