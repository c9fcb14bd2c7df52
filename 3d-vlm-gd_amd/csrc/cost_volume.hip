// Dense hw x hw cost-volume KL loss (a14-a16), fused for gfx950.
//
// Reference arithmetic (src/finetune_timm_vggt.py:509-533, src/finetune_timm_mast3r.py:522-540,
// utils/functions.py:402-422, utils/losses.py:5-15), per image pair:
//     a = normalize(F1), b = normalize(F2);  S = a b^T  (|S| <= 1)
//     dir 1 (rows i of S, teacher T1[i,:], mask m1[i]);  dir 2 (columns j of S, teacher T2[j,:], mask m2[j])
//     t = max(T / clamp_min(rowsum T, 1e-8), 1e-8) on kept rows;  p = softmax(S row)
//     KL_row = sum_j t (log t - log p) = A - B + W log Z,  A = sum t log t, B = sum t s, W = sum t, Z = sum e^s
//     masked-out rows contribute 0 (vggt: t = p = 1e-8) or hw*1e-8*log(1e-8*hw) (mast3r: p = 1/hw)
//     loss = ( mean_i KL1_i + mean_j KL2_j ) / 2
// Because |s| <= 1 no max-subtraction is needed and p >= e^-2/hw > 1e-8, so every row statistic is a
// plain sum over the tile sweep: the forward makes ONE pass over S (one MFMA contraction shared by both
// directions) and over each teacher map, and never writes the hw x hw matrix.
//
// Kernels: cv_prep (row norms, teacher row sums) -> cv_fwd_tile (128x128 S tiles, partial row/column
// statistics into slabs, deterministic) -> cv_finalize (loss + saved log Z, W).
// Backward: cv_bwd_tile recomputes S, forms G = dloss/dS and stores it (and its transpose) for two batched
// NT GEMMs (G b and G^T a), then cv_norm_bwd pulls the gradient through the L2 normalisation.
#include "gd_common.h"
#include "gemm_tile.h"

extern "C" int gd_gemm_nt(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc,
                          int batch, long sA, long sW, long sC, int ab_dtype, int c_dtype, float alpha,
                          const float* bias, const float* lora_t, const float* lora_b, int lora_rt, void* preact,
                          long ldp, int act, const void* dact_src, long ldd, int dact, const void* residual, long ldr,
                          int accumulate, void* stream);

#define CV_EPS 1e-8f

// stats layout: [P][2][hw][4] = {inv_norm, teacher_rowsum(clamped), logZ, W}
__global__ __launch_bounds__(256) void cv_prep_kernel(const void* f1, const void* f2, const float* t1,
                                                      const float* t2, float* stats, int hw, int C, int dtype) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave, which = blockIdx.y, p = blockIdx.z;
    if (row >= hw) return;
    const void* f = which ? f2 : f1;
    const float* t = (which ? t2 : t1) + ((long)p * hw + row) * hw;
    const long fo = ((long)p * hw + row) * C;
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = ld_rt(f, fo + c, dtype);
        ss += v * v;
    }
    float rs = 0.f;
    for (int j = lane; j < hw; j += 64) rs += t[j];
    ss = wave_sum(ss);
    rs = wave_sum(rs);
    if (lane == 0) {
        float* o = stats + (((long)p * 2 + which) * hw + row) * 4;
        o[0] = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        o[1] = fmaxf(rs, CV_EPS);
    }
}

struct CvTileParams {
    const void* f1; const void* f2; const float* t1; const float* t2;
    float* stats; float* part1; float* part2;
    int hw, C, tiles, nslab;
    // backward only
    const unsigned char* m1; const unsigned char* m2; const float* gloss;
    void* G1; void* G2; int hwp;
};

template <typename T>
__global__ __launch_bounds__(256) void cv_fwd_tile_kernel(CvTileParams q) {
    __shared__ __attribute__((aligned(16))) char smem[GD_TILE_SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c = lane & 15;
    const int p = blockIdx.y, hw = q.hw;
    const int wg = xcd_remap(blockIdx.x, q.tiles * q.tiles);
    const int tm = wg / q.tiles, tn = wg % q.tiles;
    const long rowb = (long)q.C * sizeof(T);
    const char* Ab = (const char*)q.f1 + (long)p * hw * rowb;
    const char* Wb = (const char*)q.f2 + (long)p * hw * rowb;
    f32x4 acc[4][4];
    mma_tile_128x128<T>(Ab, rowb, hw, Wb, rowb, hw, (int)rowb, tm, tn, smem, acc);

    const float* st1 = q.stats + ((long)p * 2 + 0) * hw * 4;
    const float* st2 = q.stats + ((long)p * 2 + 1) * hw * 4;
    // scale to cosine similarity: s = acc * inv1[i] * inv2[j]
    float inv2[4], r2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = tn * 128 + wn * 64 + j * 16 + c;
        inv2[j] = col < hw ? st2[col * 4 + 0] : 0.f;
        r2[j] = col < hw ? st2[col * 4 + 1] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tm * 128 + wm * 64 + i * 16 + g * 4 + r;
            const float inv1 = row < hw ? st1[row * 4 + 0] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j][r] *= inv1 * inv2[j];
        }

    // ---- direction 1: per-row partial sums over this wave's 64 columns ----
    const float* T1 = q.t1 + (long)p * hw * hw;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tm * 128 + wm * 64 + i * 16 + g * 4 + r;
            const bool rok = row < hw;
            const float ir1 = rok ? 1.0f / st1[row * 4 + 1] : 0.f;
            float Z = 0.f, Wt = 0.f, A = 0.f, B = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = tn * 128 + wn * 64 + j * 16 + c;
                if (rok && col < hw) {
                    const float s = acc[i][j][r];
                    const float t = fmaxf(T1[(long)row * hw + col] * ir1, CV_EPS);
                    Z += __expf(s);
                    Wt += t;
                    A += t * __logf(t);
                    B += t * s;
                }
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                Z += __shfl_xor(Z, o, 64);
                Wt += __shfl_xor(Wt, o, 64);
                A += __shfl_xor(A, o, 64);
                B += __shfl_xor(B, o, 64);
            }
            if (c == 0 && rok)
                *(f32x4*)(q.part1 + (((long)p * q.nslab + tn * 2 + wn) * hw + row) * 4) = f32x4{Z, Wt, A, B};
        }

    // ---- direction 2: per-column partial sums over this wave's 64 rows; teacher T2[j, i] is staged
    //      through LDS (coalesced along i) in two halves of 64 j-rows ----
    const float* T2 = q.t2 + (long)p * hw * hw;
    float* sT = (float*)smem;  // [64][129]
    for (int h = 0; h < 2; ++h) {
        for (int qq = 0; qq < 32; ++qq) {
            const int idx = tid + 256 * qq, jj = idx >> 7, ii = idx & 127;
            const int j = tn * 128 + h * 64 + jj, i = tm * 128 + ii;
            sT[jj * 129 + ii] = (j < hw && i < hw) ? T2[(long)j * hw + i] : 0.f;
        }
        __syncthreads();
        if (wn == h) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = tn * 128 + wn * 64 + j * 16 + c;
                const bool cok = col < hw;
                const float ir2 = 1.0f / r2[j];
                float Z = 0.f, Wt = 0.f, A = 0.f, B = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int lr = wm * 64 + i * 16 + g * 4 + r;
                        if (cok && tm * 128 + lr < hw) {
                            const float s = acc[i][j][r];
                            const float t = fmaxf(sT[(j * 16 + c) * 129 + lr] * ir2, CV_EPS);
                            Z += __expf(s);
                            Wt += t;
                            A += t * __logf(t);
                            B += t * s;
                        }
                    }
#pragma unroll
                for (int o = 16; o < 64; o <<= 1) {
                    Z += __shfl_xor(Z, o, 64);
                    Wt += __shfl_xor(Wt, o, 64);
                    A += __shfl_xor(A, o, 64);
                    B += __shfl_xor(B, o, 64);
                }
                if (g == 0 && cok)
                    *(f32x4*)(q.part2 + (((long)p * q.nslab + tm * 2 + wm) * hw + col) * 4) = f32x4{Z, Wt, A, B};
            }
        }
        __syncthreads();
    }
}

// one block per pair: reduce the slabs, emit loss[p], save logZ and W for the backward
__global__ __launch_bounds__(256) void cv_finalize_kernel(const float* part1, const float* part2,
                                                          const unsigned char* m1, const unsigned char* m2,
                                                          float* stats, float* loss, int hw, int nslab, int variant) {
    const int p = blockIdx.x, tid = threadIdx.x;
    const float masked_const = variant == 1 ? (float)hw * (CV_EPS * logf(CV_EPS * (float)hw)) : 0.f;
    double total = 0.0;
    for (int idx = tid; idx < 2 * hw; idx += 256) {
        const int d = idx >= hw, row = d ? idx - hw : idx;
        const float* part = d ? part2 : part1;
        float Z = 0.f, Wt = 0.f, A = 0.f, B = 0.f;
        for (int s = 0; s < nslab; ++s) {
            const f32x4 v = *(const f32x4*)(part + (((long)p * nslab + s) * hw + row) * 4);
            Z += v[0]; Wt += v[1]; A += v[2]; B += v[3];
        }
        const float logZ = logf(Z);
        float* st = stats + (((long)p * 2 + d) * hw + row) * 4;
        st[2] = logZ;
        st[3] = Wt;
        const bool keep = (d ? m2 : m1)[(long)p * hw + row] != 0;
        total += keep ? (double)(A - B + Wt * logZ) : (double)masked_const;
    }
    __shared__ double red[256];
    red[tid] = total;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) loss[p] = (float)(0.5 * red[0] / (double)hw);
}

// ---- backward: G = dloss/dS per tile, written as G1[i][j] = G*inv2[j] and G2[j][i] = G*inv1[i] ----
template <typename T>
__global__ __launch_bounds__(256) void cv_bwd_tile_kernel(CvTileParams q) {
    __shared__ __attribute__((aligned(16))) char smem[GD_TILE_SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c = lane & 15;
    const int p = blockIdx.y, hw = q.hw, hwp = q.hwp;
    const int wg = xcd_remap(blockIdx.x, q.tiles * q.tiles);
    const int tm = wg / q.tiles, tn = wg % q.tiles;
    const long rowb = (long)q.C * sizeof(T);
    const char* Ab = (const char*)q.f1 + (long)p * hw * rowb;
    const char* Wb = (const char*)q.f2 + (long)p * hw * rowb;
    f32x4 acc[4][4];
    mma_tile_128x128<T>(Ab, rowb, hw, Wb, rowb, hw, (int)rowb, tm, tn, smem, acc);

    const float* st1 = q.stats + ((long)p * 2 + 0) * hw * 4;
    const float* st2 = q.stats + ((long)p * 2 + 1) * hw * 4;
    const float* T1 = q.t1 + (long)p * hw * hw;
    const float* T2 = q.t2 + (long)p * hw * hw;
    const unsigned char* M1 = q.m1 + (long)p * hw;
    const unsigned char* M2 = q.m2 + (long)p * hw;
    const float coef = q.gloss[p] * 0.5f / (float)hw;
    T* G1 = (T*)q.G1 + (long)p * hw * hwp;
    T* G2 = (T*)q.G2 + (long)p * hw * hwp;
    float* sT = (float*)smem;

    f32x4 cs2[4];  // per column: inv2, 1/r2, logZ2, W2*keep2
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = tn * 128 + wn * 64 + j * 16 + c;
        if (col < hw) {
            const f32x4 v = *(const f32x4*)(st2 + col * 4);
            cs2[j] = f32x4{v[0], 1.0f / v[1], v[2], M2[col] ? v[3] : -1.f};
        } else {
            cs2[j] = f32x4{0.f, 0.f, 0.f, -1.f};
        }
    }
    // pass A: direction-1 part of G (registers), scaled s kept in acc
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tm * 128 + wm * 64 + i * 16 + g * 4 + r;
            const bool rok = row < hw;
            f32x4 v = rok ? *(const f32x4*)(st1 + row * 4) : f32x4{0.f, 1.f, 0.f, 0.f};
            const bool keep1 = rok && M1[row];
            const float ir1 = 1.0f / v[1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = tn * 128 + wn * 64 + j * 16 + c;
                const float s = acc[i][j][r] * v[0] * cs2[j][0];
                float gg = 0.f;
                if (rok && col < hw) {
                    if (keep1) {
                        const float t = fmaxf(T1[(long)row * hw + col] * ir1, CV_EPS);
                        gg += v[3] * __expf(s - v[2]) - t;
                    }
                    if (cs2[j][3] >= 0.f) gg += cs2[j][3] * __expf(s - cs2[j][2]);
                }
                acc[i][j][r] = gg;  // the dir-2 teacher term is subtracted in pass B
            }
        }
    // pass B: subtract t2 (staged through LDS), then store
    for (int h = 0; h < 2; ++h) {
        for (int qq = 0; qq < 32; ++qq) {
            const int idx = tid + 256 * qq, jj = idx >> 7, ii = idx & 127;
            const int j = tn * 128 + h * 64 + jj, i = tm * 128 + ii;
            sT[jj * 129 + ii] = (j < hw && i < hw) ? T2[(long)j * hw + i] : 0.f;
        }
        __syncthreads();
        if (wn == h) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = tn * 128 + wn * 64 + j * 16 + c;
                const bool keep2 = cs2[j][3] >= 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row0 = tm * 128 + wm * 64 + i * 16 + g * 4;
                    float o1[4], o2[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = row0 + r;
                        float gg = acc[i][j][r];
                        if (keep2 && row < hw && col < hw)
                            gg -= fmaxf(sT[(j * 16 + c) * 129 + (row - tm * 128)] * cs2[j][1], CV_EPS);
                        gg = (row < hw && col < hw) ? gg * coef : 0.f;
                        o1[r] = gg * cs2[j][0];
                        o2[r] = gg * (row < hw ? st1[row * 4] : 0.f);
                        if (row < hw && col < hwp) G1[(long)row * hwp + col] = from_f32<T>(o1[r]);
                    }
                    if (col < hw) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (row0 + r < hwp) G2[(long)col * hwp + row0 + r] = from_f32<T>(o2[r]);
                    }
                }
            }
        }
        __syncthreads();
    }
}

// out[p][c][j] = in[p][j][c] for j < hw, 0 for hw <= j < hwp   (tile transpose through LDS)
template <typename T>
__global__ __launch_bounds__(256) void cv_transpose_kernel(const T* f1, const T* f2, T* o1, T* o2, int hw, int hwp,
                                                           int C) {
    __shared__ float tile[32][33];
    const int p = blockIdx.z >> 1, which = blockIdx.z & 1;
    const T* in = (which ? f2 : f1) + (long)p * hw * C;
    T* out = (which ? o2 : o1) + (long)p * C * hwp;
    const int j0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int j = j0 + k, cc = c0 + tx;
        tile[k][tx] = (j < hw && cc < C) ? to_f32<T>(in[(long)j * C + cc]) : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int cc = c0 + k, j = j0 + tx;
        if (cc < C && j < hwp) out[(long)cc * hwp + j] = from_f32<T>(tile[tx][k]);
    }
}

// da = inv*dah - a * inv^3 * (a . dah)   (gradient through x / max(||x||, 1e-12))
template <typename T>
__global__ __launch_bounds__(256) void cv_norm_bwd_kernel(const T* f1, const T* f2, const float* d1, const float* d2,
                                                          const float* stats, T* o1, T* o2, int hw, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave, which = blockIdx.y, p = blockIdx.z;
    if (row >= hw) return;
    const long off = ((long)p * hw + row) * C;
    const T* a = (which ? f2 : f1) + off;
    const float* dh = (which ? d2 : d1) + off;
    T* o = (which ? o2 : o1) + off;
    const float inv = stats[(((long)p * 2 + which) * hw + row) * 4];
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) dot += to_f32<T>(a[c]) * dh[c];
    dot = wave_sum(dot);
    const float k = inv * inv * inv * dot;
    for (int c = lane; c < C; c += 64) o[c] = from_f32<T>(inv * dh[c] - to_f32<T>(a[c]) * k);
}

// ------------------------------------------------------------------------------------------
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int cv_tiles(int hw) { return (hw + 127) / 128; }
static inline int cv_hwp(int hw) { return (hw + 63) & ~63; }   // K of the two backward GEMMs: multiple of 128 bytes (LDS-DMA path)

extern "C" size_t gd_cost_volume_kl_workspace_bytes(int P, int hw, int C, int dtype, int backward) {
    const size_t es = (size_t)gd_dtype_size(dtype);
    if (!backward) return 2 * align256((size_t)P * 2 * cv_tiles(hw) * hw * 4 * sizeof(float));
    const size_t hwp = (size_t)cv_hwp(hw);
    return 2 * align256((size_t)P * hw * hwp * es) + 2 * align256((size_t)P * C * hwp * es) +
           2 * align256((size_t)P * hw * C * sizeof(float));
}

extern "C" int gd_cost_volume_kl_fwd(const void* f1, const void* f2, const float* t1, const float* t2,
                                     const unsigned char* m1, const unsigned char* m2, int P, int hw, int C,
                                     int variant, int dtype, float* loss, float* stats, void* workspace,
                                     void* stream) {
    GD_REQUIRE(P > 0 && hw > 0 && C > 0, "gd_cost_volume_kl_fwd: bad shape P=%d hw=%d C=%d", P, hw, C);
    GD_REQUIRE(variant == 0 || variant == 1, "gd_cost_volume_kl_fwd: variant must be 0 (vggt) or 1 (mast3r)");
    GD_REQUIRE(dtype == GD_F32 || dtype == GD_BF16, "gd_cost_volume_kl_fwd: bad dtype %d", dtype);
    GD_REQUIRE((C * gd_dtype_size(dtype)) % 16 == 0, "gd_cost_volume_kl_fwd: C*elsize must be a multiple of 16 B");
    GD_REQUIRE((double)hw * 7.3890561 * CV_EPS < 1.0, "gd_cost_volume_kl_fwd: hw too large for the clamp-free softmax");
    GD_REQUIRE(((uintptr_t)f1 & 15) == 0 && ((uintptr_t)f2 & 15) == 0 && ((uintptr_t)stats & 15) == 0 &&
                   ((uintptr_t)workspace & 15) == 0,
               "gd_cost_volume_kl_fwd: f1, f2, stats, workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int tiles = cv_tiles(hw), nslab = 2 * tiles;
    float* part1 = (float*)workspace;
    float* part2 = (float*)((char*)workspace + align256((size_t)P * nslab * hw * 4 * sizeof(float)));
    hipLaunchKernelGGL(cv_prep_kernel, dim3(gd_cdiv(hw, 4), 2, P), dim3(256), 0, s, f1, f2, t1, t2, stats, hw, C,
                       dtype);
    GD_LAUNCH_OK();
    CvTileParams q = {};
    q.f1 = f1; q.f2 = f2; q.t1 = t1; q.t2 = t2; q.stats = stats; q.part1 = part1; q.part2 = part2;
    q.hw = hw; q.C = C; q.tiles = tiles; q.nslab = nslab;
    if (dtype == GD_BF16)
        hipLaunchKernelGGL(cv_fwd_tile_kernel<bf16>, dim3(tiles * tiles, P), dim3(256), 0, s, q);
    else
        hipLaunchKernelGGL(cv_fwd_tile_kernel<float>, dim3(tiles * tiles, P), dim3(256), 0, s, q);
    GD_LAUNCH_OK();
    hipLaunchKernelGGL(cv_finalize_kernel, dim3(P), dim3(256), 0, s, part1, part2, m1, m2, stats, loss, hw, nslab,
                       variant);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_cost_volume_kl_bwd(const void* f1, const void* f2, const float* t1, const float* t2,
                                     const unsigned char* m1, const unsigned char* m2, int P, int hw, int C,
                                     int dtype, const float* gloss, const float* stats, void* df1, void* df2,
                                     void* workspace, void* stream) {
    GD_REQUIRE(P > 0 && hw > 0 && C > 0, "gd_cost_volume_kl_bwd: bad shape P=%d hw=%d C=%d", P, hw, C);
    GD_REQUIRE(dtype == GD_F32 || dtype == GD_BF16, "gd_cost_volume_kl_bwd: bad dtype %d", dtype);
    GD_REQUIRE((C * gd_dtype_size(dtype)) % 16 == 0, "gd_cost_volume_kl_bwd: C*elsize must be a multiple of 16 B");
    hipStream_t s = (hipStream_t)stream;
    const size_t es = (size_t)gd_dtype_size(dtype);
    const int tiles = cv_tiles(hw), hwp = cv_hwp(hw);
    char* w = (char*)workspace;
    void* G1 = w; w += align256((size_t)P * hw * hwp * es);
    void* G2 = w; w += align256((size_t)P * hw * hwp * es);
    void* at = w; w += align256((size_t)P * C * hwp * es);
    void* bt = w; w += align256((size_t)P * C * hwp * es);
    float* da = (float*)w; w += align256((size_t)P * hw * C * sizeof(float));
    float* db = (float*)w;
    CvTileParams q = {};
    q.f1 = f1; q.f2 = f2; q.t1 = t1; q.t2 = t2; q.stats = (float*)stats; q.hw = hw; q.C = C; q.tiles = tiles;
    q.m1 = m1; q.m2 = m2; q.gloss = gloss; q.G1 = G1; q.G2 = G2; q.hwp = hwp;
    dim3 tgrid(gd_cdiv(hwp, 32), gd_cdiv(C, 32), 2 * P);
    if (dtype == GD_BF16) {
        hipLaunchKernelGGL(cv_bwd_tile_kernel<bf16>, dim3(tiles * tiles, P), dim3(256), 0, s, q);
        hipLaunchKernelGGL(cv_transpose_kernel<bf16>, tgrid, dim3(256), 0, s, (const bf16*)f1, (const bf16*)f2,
                           (bf16*)at, (bf16*)bt, hw, hwp, C);
    } else {
        hipLaunchKernelGGL(cv_bwd_tile_kernel<float>, dim3(tiles * tiles, P), dim3(256), 0, s, q);
        hipLaunchKernelGGL(cv_transpose_kernel<float>, tgrid, dim3(256), 0, s, (const float*)f1, (const float*)f2,
                           (float*)at, (float*)bt, hw, hwp, C);
    }
    GD_LAUNCH_OK();
    // d a_hat = G1 . b ,  d b_hat = G2 . a   (contraction over the padded hw axis)
    int rc = gd_gemm_nt(G1, bt, da, hw, C, hwp, hwp, hwp, C, P, (long)hw * hwp, (long)C * hwp, (long)hw * C, dtype,
                        GD_F32, 1.0f, nullptr, nullptr, nullptr, 0, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 0, stream);
    if (rc) return rc;
    rc = gd_gemm_nt(G2, at, db, hw, C, hwp, hwp, hwp, C, P, (long)hw * hwp, (long)C * hwp, (long)hw * C, dtype,
                    GD_F32, 1.0f, nullptr, nullptr, nullptr, 0, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 0, stream);
    if (rc) return rc;
    if (dtype == GD_BF16)
        hipLaunchKernelGGL(cv_norm_bwd_kernel<bf16>, dim3(gd_cdiv(hw, 4), 2, P), dim3(256), 0, s, (const bf16*)f1,
                           (const bf16*)f2, da, db, stats, (bf16*)df1, (bf16*)df2, hw, C);
    else
        hipLaunchKernelGGL(cv_norm_bwd_kernel<float>, dim3(gd_cdiv(hw, 4), 2, P), dim3(256), 0, s, (const float*)f1,
                           (const float*)f2, da, db, stats, (float*)df1, (float*)df2, hw, C);
    GD_LAUNCH_OK();
    return 0;
}
