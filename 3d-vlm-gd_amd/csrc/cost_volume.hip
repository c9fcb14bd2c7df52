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


// stats layout: [P][2][hw][4] = {inv_norm, teacher_rowsum(clamped), logZ, W};  wa: [P][2][hw][2] = {W, A}.
// One wave per (pair, view, row).  The teacher row is read ONCE into registers (coalesced 4-byte loads: rows are only
// 4-byte aligned, hw is odd), summed, and — now that 1/rowsum is known — swept again from the registers for the two
// row statistics that do not involve the student at all:  W = sum_j t,  A = sum_j t log t,  t = max(T / rowsum, 1e-8).
// (They used to be accumulated per 128 x 128 tile next to the S-dependent terms: one v_log per element per direction.)
#define CV_TROW_MAX 24   // registers per lane for a teacher row: hw <= 1536; longer rows take a second pass over memory
__global__ __launch_bounds__(256) void cv_prep_kernel(const void* f1, const void* f2, const float* t1,
                                                      const float* t2, float* stats, float* wa, int hw, int C, int dtype) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave, which = blockIdx.y, p = blockIdx.z;
    if (row >= hw) return;
    const void* f = which ? f2 : f1;
    const float* t = (which ? t2 : t1) + ((long)p * hw + row) * hw;
    const long fo = ((long)p * hw + row) * C;
    const bool inreg = hw <= 64 * CV_TROW_MAX;
    float tv[CV_TROW_MAX];
    float rs = 0.f;
    if (inreg) {
#pragma unroll
        for (int k = 0; k < CV_TROW_MAX; ++k) {
            const int j = lane + 64 * k;
            tv[k] = j < hw ? t[j] : 0.f;
            rs += tv[k];
        }
    } else {
        for (int j = lane; j < hw; j += 64) rs += t[j];
    }
    float ss = 0.f;
    if (dtype == GD_BF16) {
        const bf16x8* fv = (const bf16x8*)((const bf16*)f + fo);
        for (int c = lane; c < C / 8; c += 64) {
            const bf16x8 v = fv[c];
#pragma unroll
            for (int k = 0; k < 8; ++k) ss += (float)v[k] * (float)v[k];
        }
    } else {
        const f32x4* fv = (const f32x4*)((const float*)f + fo);
        for (int c = lane; c < C / 4; c += 64) {
            const f32x4 v = fv[c];
            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
    }
    ss = wave_sum(ss);
    rs = fmaxf(wave_sum(rs), CV_EPS);
    const float ir = 1.0f / rs;
    float W = 0.f, A = 0.f;
    if (inreg) {
#pragma unroll
        for (int k = 0; k < CV_TROW_MAX; ++k) {
            if (lane + 64 * k < hw) {
                const float x = fmaxf(tv[k] * ir, CV_EPS);
                W += x; A += x * __logf(x);
            }
        }
    } else {
        for (int j = lane; j < hw; j += 64) {
            const float x = fmaxf(t[j] * ir, CV_EPS);
            W += x; A += x * __logf(x);
        }
    }
    W = wave_sum(W); A = wave_sum(A);
    if (lane == 0) {
        float* o = stats + (((long)p * 2 + which) * hw + row) * 4;
        o[0] = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        o[1] = rs;
        float* w2 = wa + (((long)p * 2 + which) * hw + row) * 2;
        w2[0] = W; w2[1] = A;
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

// S tile (cosine similarities, fp32) parked in LDS so that BOTH teacher sweeps read global memory row-contiguously:
// element (i, j) of a [rows][ncol] tile lives at i*ncol + (j ^ (i & 31)) — row reads and column reads are both
// bank-conflict-free.
#define CV_RING 65536
__device__ __forceinline__ int sidx(int i, int j, int ncol) { return i * ncol + (j ^ (i & 31)); }

// Per-tile row / column statistics staged in LDS once (4 floats each): {inv_norm, 1/teacher_rowsum, logZ, W or -1 when
// the row is masked out (or out of range)}.  sSt[0..127] = tile rows (view 1), sSt[128..255] = tile columns (view 2).
__device__ __forceinline__ void cv_stage_stats(const CvTileParams& q, int p, int tm, int tn, f32x4* sSt) {
    const int t = threadIdx.x, hw = q.hw;
    const int which = t >> 7, idx = (which ? tn : tm) * 128 + (t & 127);
    f32x4 o = {0.f, 0.f, 0.f, -1.f};
    if (idx < hw) {
        const f32x4 v = *(const f32x4*)(q.stats + (((long)p * 2 + which) * hw + idx) * 4);
        const unsigned char* m = which ? q.m2 : q.m1;
        const bool keep = m == nullptr || m[(long)p * hw + idx] != 0;
        o = f32x4{v[0], 1.0f / v[1], v[2], keep ? v[3] : -1.f};
    }
    sSt[t] = o;
}

template <typename T>
__device__ __forceinline__ void cv_s_tile(const CvTileParams& q, int p, int tm, int tn, char* smem, const f32x4* sSt,
                                          f32x4 (&acc)[4][4]) {
    const int hw = q.hw;
    const long rowb = (long)q.C * sizeof(T);
    const char* Ab = (const char*)q.f1 + (long)p * hw * rowb;
    const char* Wb = (const char*)q.f2 + (long)p * hw * rowb;
    if (rowb % 128 == 0) dma_mainloop<T, 2, 2, 4>(Ab, rowb, hw, Wb, rowb, hw, (int)(rowb / 128), tm, tn, smem, acc);
    else mma_tile_128x128<T>(Ab, rowb, hw, Wb, rowb, hw, (int)rowb, tm, tn, smem, acc);
    // scale to cosine similarity: s = acc * inv1[i] * inv2[j]   (stats were staged before the main loop's barriers)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1, g = lane >> 4, c = lane & 15;
    float inv2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) inv2[j] = sSt[128 + wn * 64 + j * 16 + c][0];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float inv1 = sSt[wm * 64 + i * 16 + g * 4 + r][0];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j][r] *= inv1 * inv2[j];
        }
}

// Forward tile kernel.  What is left per element once W and A come from cv_prep:
//   Z partials (sum of e^s over the tile's rows / columns): taken straight from the accumulators, ONE v_exp per element
//   for both directions, reduced with DPP / two cross-row shuffles, the two wave halves combined through LDS;
//   B partials (sum of t * s): two sweeps, S tile (then its transpose) parked in LDS, teacher tile in registers,
//   per element one multiply, one max, one FMA — no transcendentals.
// Teacher loads are 16 bytes per lane (rows are only 4-byte aligned: `f32x4_u`), 16 lanes = one contiguous 256-byte
// row segment, issued before the MFMA main loop (direction 1) / before the first sweep (direction 2).
// LDS images: S row-major (the b128 row reads of 4 rows x 256 B are conflict-free by the hardware's lane grouping);
// S^T with the 16-byte chunk index XOR-ed by (row & 15) — the transposing b128 writes of 16 lanes hit 16 different rows
// at one column offset — and the teacher columns of direction 2 permuted the same way (a sum does not care).
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ f32x4 cv_ld4(const float* T, long rowoff, int col, int hw, bool rok) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (rok) {
        if (col + 4 <= hw) { const f32x4_u u = *(const f32x4_u*)(T + rowoff + col); v = f32x4{u[0], u[1], u[2], u[3]}; }
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (col + k < hw) v[k] = T[rowoff + col + k];
        }
    }
    return v;
}

template <typename T>
__global__ __launch_bounds__(256, 2) void cv_fwd_tile_kernel(CvTileParams q) {
    __shared__ __attribute__((aligned(16))) char smem[CV_RING + 4096 + 2048];   // ring / S tile | tile statistics | Z partials
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c = lane & 15;
    const int p = blockIdx.y, hw = q.hw;
    const int wg = xcd_remap(blockIdx.x, q.tiles * q.tiles);
    const int tm = wg / q.tiles, tn = wg % q.tiles;
    const float* T1 = q.t1 + (long)p * hw * hw;
    const float* T2 = q.t2 + (long)p * hw * hw;
    f32x4* sSt = (f32x4*)(smem + CV_RING);
    float* sZr = (float*)(smem + CV_RING + 4096);   // [2 (wn)][128 rows]
    float* sZc = sZr + 256;                          // [2 (wm)][128 columns]
    cv_stage_stats(q, p, tm, tn, sSt);
    // direction-1 teacher tile (tile row = teacher row), in flight under the main loop
    f32x4 t1v[8][2], t2v[8][2];
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const int row = tm * 128 + wave * 32 + st * 4 + g;
#pragma unroll
        for (int h = 0; h < 2; ++h) t1v[st][h] = cv_ld4(T1, (long)row * hw, tn * 128 + h * 64 + 4 * c, hw, row < hw);
    }
    f32x4 acc[4][4];
    cv_s_tile<T>(q, p, tm, tn, smem, sSt, acc);     // ends with a barrier: the ring is free
    float* sS = (float*)smem;                        // [128][128] row-major
    // ---- Z partials from the accumulators (e^s of in-range elements) + S into LDS ----
    {
        float zc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = wm * 64 + i * 16 + g * 4 + r;
                const bool rok = tm * 128 + rl < hw;
                float zr = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cl = wn * 64 + j * 16 + c;
                    const float s = acc[i][j][r];
                    const float e = (rok && tn * 128 + cl < hw) ? __expf(s) : 0.f;
                    zr += e; zc[j] += e;
                    sS[rl * 128 + cl] = s;
                }
                zr = row16_sum(zr);
                if (c == 0) sZr[wn * 128 + rl] = zr;
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = zc[j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g == 0) sZc[wm * 128 + wn * 64 + j * 16 + c] = v;
        }
    }
    // direction-2 teacher tile (tile column = teacher row); columns permuted like the S^T image
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const int jl = wave * 32 + st * 4 + g, trow = tn * 128 + jl;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            t2v[st][h] = cv_ld4(T2, (long)trow * hw, tm * 128 + h * 64 + 4 * (c ^ (jl & 15)), hw, trow < hw);
    }
    __syncthreads();
    // ---- direction 1: B = sum_j t s over the tile's columns, rows wave*32 .. +32 ----
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const int rl = wave * 32 + st * 4 + g, row = tm * 128 + rl;
        const float ir1 = sSt[rl][1];
        float B = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 sv = *(const f32x4*)(sS + rl * 128 + h * 64 + 4 * c);
            const int col = tn * 128 + h * 64 + 4 * c;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (col + k < hw) B = fmaf(fmaxf(t1v[st][h][k] * ir1, CV_EPS), sv[k], B);
        }
        B = row16_sum(B);
        if (c == 0 && row < hw)
            *(f32x4*)(q.part1 + (((long)p * q.nslab + tn) * hw + row) * 4) = f32x4{sZr[rl] + sZr[128 + rl], 0.f, 0.f, B};
    }
    __syncthreads();
    // ---- S^T into the same LDS: lane (g, c) owns rows 4g..4g+3 of m-tile i for its column: one b128 per (i, j) ----
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cl = wn * 64 + j * 16 + c;                 // S^T row
            const int L = i * 4 + g;                             // 16-byte chunk inside the wm half
            *(f32x4*)(sS + cl * 128 + wm * 64 + 4 * (L ^ (cl & 15))) = acc[i][j];
        }
    __syncthreads();
    // ---- direction 2: B = sum_i t s over the tile's rows, columns wave*32 .. +32 ----
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const int jl = wave * 32 + st * 4 + g, col = tn * 128 + jl;
        const float ir2 = sSt[128 + jl][1];
        float B = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 sv = *(const f32x4*)(sS + jl * 128 + h * 64 + 4 * c);      // physical chunk c = logical c ^ (jl & 15)
            const int ti = tm * 128 + h * 64 + 4 * (c ^ (jl & 15));
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ti + k < hw) B = fmaf(fmaxf(t2v[st][h][k] * ir2, CV_EPS), sv[k], B);
        }
        B = row16_sum(B);
        if (c == 0 && col < hw)
            *(f32x4*)(q.part2 + (((long)p * q.nslab + tm) * hw + col) * 4) = f32x4{sZc[jl] + sZc[128 + jl], 0.f, 0.f, B};
    }
}

// reduce the slabs, save logZ and W for the backward, emit per-chunk partial losses (CV_FCH chunks per pair), then sum
#define CV_FCH 8
__global__ __launch_bounds__(256) void cv_finalize_kernel(const float* part1, const float* part2, const float* wa,
                                                          const unsigned char* m1, const unsigned char* m2,
                                                          float* stats, double* chunk_loss, int hw, int nslab, int variant) {
    const int p = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
    const float masked_const = variant == 1 ? (float)hw * (CV_EPS * logf(CV_EPS * (float)hw)) : 0.f;
    double total = 0.0;
    const int per = (2 * hw + CV_FCH - 1) / CV_FCH;
    for (int idx = ch * per + tid; idx < min(2 * hw, (ch + 1) * per); idx += 256) {
        const int d = idx >= hw, row = d ? idx - hw : idx;
        const float* part = d ? part2 : part1;
        float Z = 0.f, B = 0.f;
        for (int s = 0; s < nslab; ++s) {
            const f32x4 v = *(const f32x4*)(part + (((long)p * nslab + s) * hw + row) * 4);
            Z += v[0]; B += v[3];
        }
        const float Wt = wa[(((long)p * 2 + d) * hw + row) * 2], A = wa[(((long)p * 2 + d) * hw + row) * 2 + 1];
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
    if (tid == 0) chunk_loss[p * CV_FCH + ch] = red[0];
}
__global__ void cv_loss_kernel(const double* chunk_loss, float* loss, int P, int hw) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= P) return;
    double s = 0.0;
    for (int c = 0; c < CV_FCH; ++c) s += chunk_loss[p * CV_FCH + c];   // fixed order: deterministic
    loss[p] = (float)(0.5 * s / (double)hw);
}

// ---- backward: G = dloss/dS per tile, written as G1[i][j] = G*inv2[j] and G2[j][i] = G*inv1[i] (both row-contiguous
//      stores).  Teacher tiles are preloaded into registers before the main loop; the tile is processed in two
//      64-column halves with s and the accumulating G in LDS (2 x 32 KB), per-row/column statistics in LDS. ----
template <typename T>
__global__ __launch_bounds__(256, 2) void cv_bwd_tile_kernel(CvTileParams q) {
    __shared__ __attribute__((aligned(16))) char smem[CV_RING + 4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c = lane & 15;
    const int p = blockIdx.y, hw = q.hw, hwp = q.hwp;
    const int wg = xcd_remap(blockIdx.x, q.tiles * q.tiles);
    const int tm = wg / q.tiles, tn = wg % q.tiles;
    const float* T1 = q.t1 + (long)p * hw * hw;
    const float* T2 = q.t2 + (long)p * hw * hw;
    f32x4* sSt = (f32x4*)(smem + CV_RING);
    cv_stage_stats(q, p, tm, tn, sSt);
    float t1v[2][32];   // pass-A teacher values of both halves: in flight under the main loop
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int k = 0; k < 32; ++k) {   // pass A: tile row wave+4k, column = 64h + lane
            const int row = tm * 128 + wave + 4 * k, col = tn * 128 + 64 * h + lane;
            t1v[h][k] = (row < hw && col < hw) ? T1[(long)row * hw + col] : 0.f;
        }
    f32x4 acc[4][4];
    cv_s_tile<T>(q, p, tm, tn, smem, sSt, acc);
    float* sS = (float*)smem;            // [128][64]
    float* sG = (float*)smem + 128 * 64; // [128][64]
    const float coef = q.gloss[p] * 0.5f / (float)hw;
    T* G1 = (T*)q.G1 + (long)p * hw * hwp;
    T* G2 = (T*)q.G2 + (long)p * hw * hwp;

#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (wn == h) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sS[sidx(wm * 64 + i * 16 + g * 4 + r, j * 16 + c, 64)] = acc[i][j][r];
        }
        const int col0 = tn * 128 + 64 * h;
        float t2v[16][2];   // pass-B teacher values of this half: issued here, land under pass A
#pragma unroll
        for (int k = 0; k < 16; ++k)     // teacher row (tile column) 64h + wave+4k, tile rows lane, lane+64
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int col = col0 + wave + 4 * k, row = tm * 128 + lane + 64 * e;
                t2v[k][e] = (row < hw && col < hw) ? T2[(long)col * hw + row] : 0.f;
            }
        __syncthreads();
        // pass A: direction-1 term (lane = column)
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int rr = wave + 4 * k;
            const f32x4 v = sSt[rr];
            float term = 0.f;
            if (v[3] >= 0.f && col0 + lane < hw)
                term = v[3] * __expf(sS[sidx(rr, lane, 64)] - v[2]) - fmaxf(t1v[h][k] * v[1], CV_EPS);
            sG[sidx(rr, lane, 64)] = term;
        }
        __syncthreads();
        // pass B: direction-2 term (lane = tile row), then G2 rows
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int jl = wave + 4 * k, col = col0 + jl;
            const bool cok = col < hw;
            const f32x4 v = sSt[128 + 64 * h + jl];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int il = lane + 64 * e, row = tm * 128 + il;
                float gg = sG[sidx(il, jl, 64)];
                if (v[3] >= 0.f && row < hw)
                    gg += v[3] * __expf(sS[sidx(il, jl, 64)] - v[2]) - fmaxf(t2v[k][e] * v[1], CV_EPS);
                gg = (cok && row < hw) ? gg * coef : 0.f;
                sG[sidx(il, jl, 64)] = gg;
                if (cok && row < hwp) G2[(long)col * hwp + row] = from_f32<T>(gg * sSt[il][0]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int rr = wave + 4 * k, row = tm * 128 + rr, col = col0 + lane;
            if (row < hw && col < hwp) G1[(long)row * hwp + col] = from_f32<T>(sG[sidx(rr, lane, 64)] * sSt[128 + 64 * h + lane][0]);
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
    if (!backward) return 2 * align256((size_t)P * 2 * cv_tiles(hw) * hw * 4 * sizeof(float)) + align256((size_t)P * CV_FCH * sizeof(double)) +
                          align256((size_t)P * 2 * hw * 2 * sizeof(float));
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
    const int tiles = cv_tiles(hw), nslab = tiles;
    float* part1 = (float*)workspace;
    float* part2 = (float*)((char*)workspace + align256((size_t)P * nslab * hw * 4 * sizeof(float)));
    double* chunk_loss = (double*)((char*)workspace + 2 * align256((size_t)P * 2 * tiles * hw * 4 * sizeof(float)));
    float* wa = (float*)((char*)chunk_loss + align256((size_t)P * CV_FCH * sizeof(double)));
    hipLaunchKernelGGL(cv_prep_kernel, dim3(gd_cdiv(hw, 4), 2, P), dim3(256), 0, s, f1, f2, t1, t2, stats, wa, hw, C,
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
    hipLaunchKernelGGL(cv_finalize_kernel, dim3(P, CV_FCH), dim3(256), 0, s, part1, part2, wa, m1, m2, stats, chunk_loss, hw,
                       nslab, variant);
    hipLaunchKernelGGL(cv_loss_kernel, dim3(gd_cdiv(P, 64)), dim3(64), 0, s, chunk_loss, loss, P, hw);
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
