// LayerNorm forward / backward-to-input (the student's norm layers are frozen: no dgamma/dbeta) and
// row L2 normalisation, one wave per row, rows held in registers (D <= 2048, D % 4 == 0).
// Replaces nn.LayerNorm in timm's Block / model.norm and F.normalize
// (src/finetune_timm_vggt.py:328).
#include "gd_common.h"
#include <stdlib.h>

#define LN_MAXV 8  // 8 x (64 lanes x 4 elements) = 2048 columns

template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 load4<bf16>(const bf16* p) {
    const bf16x4 v = *(const bf16x4*)p;
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T> __device__ __forceinline__ void store4n(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4n<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4n<bf16>(bf16* p, f32x4 v) {
    *(bf16x4*)p = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
}
template <> __device__ __forceinline__ void store4n<f16>(f16* p, f32x4 v) {      // tf32h operands: saturated
    *(f16x4*)p = f16_sat4(v[0], v[1], v[2], v[3]);
}

template <typename T, typename TO, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* x, const float* gamma, const float* beta, TO* y,
                                                     float* mean, float* rstd, int M, int D, long ldx, long ldy,
                                                     float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T* xr = x + (long)row * ldx;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
            v[i] = load4<T>(xr + c);
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float d = v[i][k] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
    TO* yr = y + (long)row * ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
            const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (v[i][k] - mu) * rs * g[k] + b[k];
            store4n<TO>(yr + c, o);
        }
    }
}

template <> __device__ __forceinline__ f32x4 load4<f16>(const f16* p) {
    const f16x4 v = *(const f16x4*)p;
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
// max over the 64 lanes of non-negative-float bit patterns (integer order = float order, and a NaN / Inf pattern beats every finite one)
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned w = (unsigned)__shfl_xor((int)v, o, 64); v = w > v ? w : v; }
    return v;
}
__device__ __forceinline__ unsigned wave_uadd(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (unsigned)__shfl_xor((int)v, o, 64);
    return v;
}
#define GD_AMAX_SLOTS 256
#define GD_RANGE_SLOTS 64      // range counters: [0, 64) saturated, [64, 128) below-normal-range — partial sums, spread so that the blocks' atomics do not queue on one line

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  dx += dres when given
// tf32h extras (gd_layernorm_bwd_ex): dy may arrive as fp16 under a device-side power-of-two scale (dys = 1 / s undoes it); dx also leaves as
// fp16(dx * *sdev) (the next product's operand); `amax` (GD_AMAX_SLOTS words) receives max |dx| bit patterns — the NEXT block's gradient scale
// without a pass of its own (one atomicMax per wave, spread over the slots, skipped when the slot already holds a larger value); `range`
// counts the fp16 copies that saturated ([0]) or fell below fp16's normal range ([1]: subnormal or flushed, the input not zero).
template <typename T, typename TD, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TD* dy, const T* x, const float* gamma, const float* mean,
                                                     const float* rstd, const T* dres, const T* dres2, T* dx, int M, int D,
                                                     long ldd, long ldx, float dyscale, f16* dx16 = nullptr, const float* sdev = nullptr,
                                                     const float* dys_dev = nullptr, unsigned* amax = nullptr, unsigned* range = nullptr) {
    const int lane = threadIdx.x & 63;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = row < M;
    if (!live) {
        if (!amax && !range) return;
        row = M - 1;      // (with the block-level reductions below every wave reaches the barrier: a wave past M recomputes the last row and stores nothing)
    }
    const float mu = mean[row], rs = rstd[row];
    const float s16 = (dx16 && sdev) ? *sdev : 1.0f;      // gd_layernorm_bwd_cast: dx also leaves as fp16(dx * s), the next product's operand
    if (dys_dev) dyscale *= *dys_dev;
    const T* xr = x + (long)row * ldx;
    const TD* dr = dy + (long)row * ldd;
    f32x4 xh[NV], g[NV];
    float s1 = 0.f, s2 = 0.f;
    unsigned mb = 0u, nsat = 0u, nlow = 0u;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
            const f32x4 xv = load4<T>(xr + c), dv = load4<TD>(dr + c), gm = *(const f32x4*)(gamma + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                xh[i][k] = (xv[k] - mu) * rs;
                g[i][k] = dv[k] * dyscale * gm[k];
                s1 += g[i][k];
                s2 += g[i][k] * xh[i][k];
            }
            // an fp16 dy is the saturating C store of the GEMM that produced it (gemm epilogue f16_sat: no counter there): an element AT the clamp
            // — or Inf / NaN — is counted here, by the consumer, so that range_report()'s "saturated" covers the dX products too
            if (sizeof(TD) == 2 && sizeof(T) == 4 && range && live) {
#pragma unroll
                for (int k = 0; k < 4; ++k) nsat += !(fabsf(dv[k]) < 65504.0f);
            }
        }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    T* or_ = dx + (long)row * ldx;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = rs * (g[i][k] - s1 - xh[i][k] * s2);
            if (dres) {
                const f32x4 r = load4<T>(dres + (long)row * ldx + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] += r[k];
            }
            if (dres2) {
                const f32x4 r = load4<T>(dres2 + (long)row * ldx + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] += r[k];
            }
            if (live) store4n<T>(or_ + c, o);
            if (amax) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { const unsigned b = gd_f2u(o[k]) & 0x7fffffffu; mb = b > mb ? b : mb; }
            }
            if (dx16) {
                const f32x4 os = o * s16;
                if (live) store4n<f16>(dx16 + (long)row * D + c, os);
                if (range && live) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned b = gd_f2u(os[k]) & 0x7fffffffu;
                        nsat += b > 0x477fe000u;                        // |v| > 65504 (Inf / NaN included)
                        nlow += (b < 0x38800000u) & (b != 0u);          // 0 < |v| < 2^-14
                    }
                }
            }
        }
    }
    // ONE atomic per block and counter, spread over the slots (a wave-level atomic of every one of the 87 680 rows on one address serialises at
    // the memory side: measured +200 us on a 190 us pass)
    if (amax || range) {
        __shared__ unsigned red[3][4];
        const int w = threadIdx.x >> 6;
        if (amax) mb = wave_umax(mb);
        if (range) { nsat = wave_uadd(nsat); nlow = wave_uadd(nlow); }
        if (lane == 0) { red[0][w] = mb; red[1][w] = nsat; red[2][w] = nlow; }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (amax) {
                const unsigned a = red[0][0] > red[0][1] ? red[0][0] : red[0][1], b = red[0][2] > red[0][3] ? red[0][2] : red[0][3], m4 = a > b ? a : b;
                unsigned* slot = amax + (blockIdx.x & (GD_AMAX_SLOTS - 1));
                if (m4 > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, m4);
            }
            if (range) {
                const unsigned ns = red[1][0] + red[1][1] + red[1][2] + red[1][3], nl = red[2][0] + red[2][1] + red[2][2] + red[2][3];
                if (ns) atomicAdd(range + (blockIdx.x & (GD_RANGE_SLOTS - 1)), ns);
                if (nl) atomicAdd(range + GD_RANGE_SLOTS + (blockIdx.x & (GD_RANGE_SLOTS - 1)), nl);
            }
        }
    }
}


// bf16 rows with 16-byte accesses: a lane holds 8 consecutive elements per 512-column slab (D = 768: lanes 0-63 + lanes 0-31)
template <int NV2>
__global__ __launch_bounds__(256) void ln_fwd8_kernel(const bf16* x, const float* gamma, const float* beta, bf16* y, float* mean,
                                                      float* rstd, int M, int D, long ldx, long ldy, float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16* xr = x + (long)row * ldx;
    float v[NV2][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV2; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < D) {
            const bf16x8 t = *(const bf16x8*)(xr + c);
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[i][k] = (float)t[k]; s += v[i][k]; }
        }
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV2; ++i)
        if ((lane + 64 * i) * 8 < D) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mu; q += d * d; }
        }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
    bf16* yr = y + (long)row * ldy;
#pragma unroll
    for (int i = 0; i < NV2; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < D) {
            const f32x4 g0 = *(const f32x4*)(gamma + c), g1 = *(const f32x4*)(gamma + c + 4);
            const f32x4 b0 = *(const f32x4*)(beta + c), b1 = *(const f32x4*)(beta + c + 4);
            bf16x8 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = (bf16)((v[i][k] - mu) * rs * g0[k] + b0[k]);
                o[k + 4] = (bf16)((v[i][k + 4] - mu) * rs * g1[k] + b1[k]);
            }
            *(bf16x8*)(yr + c) = o;
        }
    }
}
template <int NV2>
__global__ __launch_bounds__(256) void ln_bwd8_kernel(const bf16* dy, const bf16* x, const float* gamma, const float* mean,
                                                      const float* rstd, const bf16* dres, const bf16* dres2, bf16* dx, int M, int D,
                                                      long ldd, long ldx, float dyscale) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float mu = mean[row], rs = rstd[row];
    const bf16* xr = x + (long)row * ldx;
    const bf16* dr = dy + (long)row * ldd;
    float xh[NV2][8], g[NV2][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV2; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < D) {
            const bf16x8 xv = *(const bf16x8*)(xr + c), dv = *(const bf16x8*)(dr + c);
            const f32x4 g0 = *(const f32x4*)(gamma + c), g1 = *(const f32x4*)(gamma + c + 4);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                xh[i][k] = ((float)xv[k] - mu) * rs;
                g[i][k] = (float)dv[k] * dyscale * (k < 4 ? g0[k] : g1[k - 4]);
                s1 += g[i][k];
                s2 += g[i][k] * xh[i][k];
            }
        }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    bf16* or_ = dx + (long)row * ldx;
#pragma unroll
    for (int i = 0; i < NV2; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < D) {
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = rs * (g[i][k] - s1 - xh[i][k] * s2);
            if (dres) {
                const bf16x8 r = *(const bf16x8*)(dres + (long)row * ldx + c);
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] += (float)r[k];
            }
            if (dres2) {
                const bf16x8 r = *(const bf16x8*)(dres2 + (long)row * ldx + c);
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] += (float)r[k];
            }
            bf16x8 ov;
#pragma unroll
            for (int k = 0; k < 8; ++k) ov[k] = (bf16)o[k];
            *(bf16x8*)(or_ + c) = ov;
        }
    }
}

// y = x / max(||x||, eps) on fp32 rows; backward dx = (dy - y (y.dy)) / max(||x||, eps)
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* x, float* y, float* inv, int M, int D, float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = x[(long)row * D + c]; s += v * v; }
    const float iv = 1.0f / fmaxf(sqrtf(wave_sum(s)), eps);
    if (lane == 0 && inv) inv[row] = iv;
    for (int c = lane; c < D; c += 64) y[(long)row * D + c] = x[(long)row * D + c] * iv;
}
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* y, const float* dy, const float* inv, float* dx,
                                                         int M, int D) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += y[(long)row * D + c] * dy[(long)row * D + c];
    s = wave_sum(s);
    const float iv = inv[row];
    for (int c = lane; c < D; c += 64) dx[(long)row * D + c] = (dy[(long)row * D + c] - y[(long)row * D + c] * s) * iv;
}

// NV = number of 256-column slabs a wave walks: keeps the register footprint proportional to D
#define LN_DISPATCH_NV(D, CALL)                     \
    do {                                            \
        const int nv_ = ((D) + 255) / 256;          \
        if (nv_ <= 1) { CALL(1); }                  \
        else if (nv_ <= 2) { CALL(2); }             \
        else if (nv_ <= 3) { CALL(3); }             \
        else if (nv_ <= 4) { CALL(4); }             \
        else if (nv_ <= 6) { CALL(6); }             \
        else { CALL(8); }                           \
    } while (0)

extern "C" int gd_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                float* rstd, int M, int D, long ldx, long ldy, float eps, int dtype, int y_dtype,
                                void* stream) {
    GD_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * LN_MAXV, "gd_layernorm_fwd: D=%d must be a multiple of 4 and <= 2048", D);
    GD_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "gd_layernorm_fwd: row strides must be multiples of 4 elements");
    GD_REQUIRE((mean == nullptr) == (rstd == nullptr), "gd_layernorm_fwd: pass both mean and rstd or neither");
    dim3 grid(gd_cdiv(M, 4)), blk(256);
    hipStream_t s = (hipStream_t)stream;
#define F_BB(NV) hipLaunchKernelGGL((ln_fwd_kernel<bf16, bf16, NV>), grid, blk, 0, s, (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, M, D, ldx, ldy, eps)
#define F_BF(NV) hipLaunchKernelGGL((ln_fwd_kernel<bf16, float, NV>), grid, blk, 0, s, (const bf16*)x, gamma, beta, (float*)y, mean, rstd, M, D, ldx, ldy, eps)
#define F_FF(NV) hipLaunchKernelGGL((ln_fwd_kernel<float, float, NV>), grid, blk, 0, s, (const float*)x, gamma, beta, (float*)y, mean, rstd, M, D, ldx, ldy, eps)
#define F_FH(NV) hipLaunchKernelGGL((ln_fwd_kernel<float, f16, NV>), grid, blk, 0, s, (const float*)x, gamma, beta, (f16*)y, mean, rstd, M, D, ldx, ldy, eps)
    const int ln16 = gd_knobs().ln_16b;
    const bool wide = ln16 && dtype == GD_BF16 && y_dtype == GD_BF16 && D % 8 == 0 && D <= 1024 && ldx % 8 == 0 && ldy % 8 == 0 &&
                      ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0;
    if (wide && D <= 512) hipLaunchKernelGGL((ln_fwd8_kernel<1>), grid, blk, 0, s, (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, M, D, ldx, ldy, eps);
    else if (wide) hipLaunchKernelGGL((ln_fwd8_kernel<2>), grid, blk, 0, s, (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, M, D, ldx, ldy, eps);
    else if (dtype == GD_BF16 && y_dtype == GD_BF16) LN_DISPATCH_NV(D, F_BB);
    else if (dtype == GD_BF16 && y_dtype == GD_F32) LN_DISPATCH_NV(D, F_BF);
    else if (dtype == GD_F32 && y_dtype == GD_F32) LN_DISPATCH_NV(D, F_FF);
    else if (dtype == GD_F32 && y_dtype == GD_F16) LN_DISPATCH_NV(D, F_FH);      // tf32h engine: LN(x) is only ever a product operand
    else {
        gd_set_error("gd_layernorm_fwd: unsupported dtype pair %d -> %d", dtype, y_dtype);
        return -1;
    }
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                const float* rstd, const void* dres, const void* dres2, void* dx, int M, int D, long ldd,
                                long ldx, float dyscale, int dtype, int dy_dtype, void* stream) {
    GD_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * LN_MAXV, "gd_layernorm_bwd: D=%d must be a multiple of 4 and <= 2048", D);
    GD_REQUIRE(ldx % 4 == 0 && ldd % 4 == 0, "gd_layernorm_bwd: row strides must be multiples of 4 elements");
    dim3 grid(gd_cdiv(M, 4)), blk(256);
    hipStream_t s = (hipStream_t)stream;
#define B_BB(NV) hipLaunchKernelGGL((ln_bwd_kernel<bf16, bf16, NV>), grid, blk, 0, s, (const bf16*)dy, (const bf16*)x, gamma, mean, rstd, (const bf16*)dres, (const bf16*)dres2, (bf16*)dx, M, D, ldd, ldx, dyscale)
#define B_BF(NV) hipLaunchKernelGGL((ln_bwd_kernel<bf16, float, NV>), grid, blk, 0, s, (const float*)dy, (const bf16*)x, gamma, mean, rstd, (const bf16*)dres, (const bf16*)dres2, (bf16*)dx, M, D, ldd, ldx, dyscale)
#define B_FF(NV) hipLaunchKernelGGL((ln_bwd_kernel<float, float, NV>), grid, blk, 0, s, (const float*)dy, (const float*)x, gamma, mean, rstd, (const float*)dres, (const float*)dres2, (float*)dx, M, D, ldd, ldx, dyscale)
    const int ln16 = gd_knobs().ln_16b;
    const bool wide = ln16 && dtype == GD_BF16 && dy_dtype == GD_BF16 && D % 8 == 0 && D <= 1024 && ldx % 8 == 0 && ldd % 8 == 0 &&
                      ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)gamma % 16) == 0 &&
                      ((uintptr_t)dres % 16) == 0 && ((uintptr_t)dres2 % 16) == 0;
    if (wide && D <= 512) hipLaunchKernelGGL((ln_bwd8_kernel<1>), grid, blk, 0, s, (const bf16*)dy, (const bf16*)x, gamma, mean, rstd, (const bf16*)dres, (const bf16*)dres2, (bf16*)dx, M, D, ldd, ldx, dyscale);
    else if (wide) hipLaunchKernelGGL((ln_bwd8_kernel<2>), grid, blk, 0, s, (const bf16*)dy, (const bf16*)x, gamma, mean, rstd, (const bf16*)dres, (const bf16*)dres2, (bf16*)dx, M, D, ldd, ldx, dyscale);
    else if (dtype == GD_BF16 && dy_dtype == GD_BF16) LN_DISPATCH_NV(D, B_BB);
    else if (dtype == GD_BF16 && dy_dtype == GD_F32) LN_DISPATCH_NV(D, B_BF);
    else if (dtype == GD_F32 && dy_dtype == GD_F32) LN_DISPATCH_NV(D, B_FF);
    else {
        gd_set_error("gd_layernorm_bwd: unsupported dtype pair x=%d dy=%d", dtype, dy_dtype);
        return -1;
    }
    GD_LAUNCH_OK();
    return 0;
}

// fp32 LayerNorm backward that ALSO writes fp16(dx * *scale_dev) [M, D] (contiguous): the tf32h engine's next product takes dx as its left
// operand under the block's power-of-two gradient scale (gd_amax_scale) — one pass instead of backward + gd_cast_f16.
extern "C" int gd_layernorm_bwd_cast(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                     const float* dres, const float* dres2, float* dx, void* dx16, const float* scale_dev, int M, int D,
                                     long ldd, long ldx, float dyscale, void* stream) {
    GD_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * LN_MAXV, "gd_layernorm_bwd_cast: D=%d must be a multiple of 4 and <= 2048", D);
    GD_REQUIRE(ldx % 4 == 0 && ldd % 4 == 0 && dx16 != nullptr, "gd_layernorm_bwd_cast: row strides must be multiples of 4 elements; dx16 required");
    dim3 grid(gd_cdiv(M, 4)), blk(256);
    hipStream_t s = (hipStream_t)stream;
#define B_FH(NV) hipLaunchKernelGGL((ln_bwd_kernel<float, float, NV>), grid, blk, 0, s, dy, x, gamma, mean, rstd, dres, dres2, dx, M, D, ldd, ldx, dyscale, (f16*)dx16, scale_dev)
    LN_DISPATCH_NV(D, B_FH);
    GD_LAUNCH_OK();
    return 0;
}

// The tf32h block backward's LayerNorm pass with every device-side extra (kernel comment above): dy f32 or fp16 (dy_dtype) times *dy_scale_dev,
// dx16 / cast_scale_dev as gd_layernorm_bwd_cast (dx16 nullable here), amax_slots (GD_AMAX_SLOTS = 256 words, zeroed by gd_scale_from_amax after
// each use) and range_counters (2 x GD_RANGE_SLOTS = 128 words of partial sums, accumulated) nullable.
extern "C" int gd_layernorm_bwd_ex(const void* dy, int dy_dtype, const float* dy_scale_dev, const float* x, const float* gamma, const float* mean,
                                   const float* rstd, const float* dres, const float* dres2, float* dx, void* dx16, const float* cast_scale_dev,
                                   unsigned* amax_slots, unsigned* range_counters, int M, int D, long ldd, long ldx, float dyscale, void* stream) {
    GD_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * LN_MAXV, "gd_layernorm_bwd_ex: D=%d must be a multiple of 4 and <= 2048", D);
    GD_REQUIRE(ldx % 4 == 0 && ldd % 4 == 0, "gd_layernorm_bwd_ex: row strides must be multiples of 4 elements");
    GD_REQUIRE(dy_dtype == GD_F32 || dy_dtype == GD_F16, "gd_layernorm_bwd_ex: dy is f32 or fp16 (dy_dtype %d)", dy_dtype);
    GD_REQUIRE(((uintptr_t)dy & 7) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dx & 15) == 0, "gd_layernorm_bwd_ex: alignment");
    dim3 grid(gd_cdiv(M, 4)), blk(256);
    hipStream_t s = (hipStream_t)stream;
#define B_XF(NV) hipLaunchKernelGGL((ln_bwd_kernel<float, float, NV>), grid, blk, 0, s, (const float*)dy, x, gamma, mean, rstd, dres, dres2, dx, M, D, ldd, ldx, dyscale, (f16*)dx16, cast_scale_dev, dy_scale_dev, amax_slots, range_counters)
#define B_XH(NV) hipLaunchKernelGGL((ln_bwd_kernel<float, f16, NV>), grid, blk, 0, s, (const f16*)dy, x, gamma, mean, rstd, dres, dres2, dx, M, D, ldd, ldx, dyscale, (f16*)dx16, cast_scale_dev, dy_scale_dev, amax_slots, range_counters)
    if (dy_dtype == GD_F16) LN_DISPATCH_NV(D, B_XH); else LN_DISPATCH_NV(D, B_XF);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_l2norm_fwd(const float* x, float* y, float* inv, int M, int D, float eps, void* stream) {
    GD_REQUIRE(M > 0 && D > 0, "gd_l2norm_fwd: bad shape");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(gd_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, y, inv, M, D, eps);
    GD_LAUNCH_OK();
    return 0;
}
extern "C" int gd_l2norm_bwd(const float* y, const float* dy, const float* inv, float* dx, int M, int D, void* stream) {
    GD_REQUIRE(M > 0 && D > 0, "gd_l2norm_bwd: bad shape");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(gd_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, y, dy, inv, dx, M, D);
    GD_LAUNCH_OK();
    return 0;
}
