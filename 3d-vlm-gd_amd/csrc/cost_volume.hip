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
#include <type_traits>
#include "gemm_frag.h"
#include <stdlib.h>

extern "C" int gd_gemm_nt(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc,
                          int batch, long sA, long sW, long sC, int ab_dtype, int c_dtype, float alpha,
                          const float* bias, const float* lora_t, const float* lora_b, int lora_rt, void* preact,
                          long ldp, int act, const void* dact_src, long ldd, int dact, const void* residual, long ldr,
                          int accumulate, void* stream);

#define CV_EPS 1e-8f


// stats layout: [P][2][hw][4] = {inv_norm, teacher_rowsum(clamped), logZ, W};  tstats: [P][2][hw][4] = {rowsum(clamped), W, A, 0}.
// Teacher maps are [P][hw][ldt] (row stride ldt >= hw elements; ldt % 4 == 0 makes every row 16-byte aligned).
//
// cv_tstats: everything about a teacher row that does not involve the student — it depends on the teacher data only, so
// a caller that caches the targets of a pair computes it ONCE (gd_cost_volume_teacher_stats) and the step then reads each
// teacher map a single time.  One wave per (pair, view, row): the row is read ONCE into registers (coalesced dword loads),
// summed, and — now that 1/rowsum is known — swept again from the registers for  W = sum_j t,  A = sum_j t log t,
// t = max(T / rowsum, 1e-8).
#define CV_TROW_MAX 24   // registers per lane for a teacher row: hw <= 1536; longer rows take a second pass over memory
__global__ __launch_bounds__(256) void cv_tstats_kernel(const float* t1, const float* t2, float* tstats, int hw, int ldt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave, which = blockIdx.y, p = blockIdx.z;
    if (row >= hw) return;
    const float* t = (which ? t2 : t1) + ((long)p * hw + row) * ldt;
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
    if (lane == 0) *(f32x4*)(tstats + (((long)p * 2 + which) * hw + row) * 4) = f32x4{rs, W, A, 0.f};
}

// inverse L2 norms of the student feature rows (+ the teacher row sum copied next to them: the tile kernels read one
// float4 per row / column).  One wave per (pair, view, row).
__global__ __launch_bounds__(256) void cv_norm_kernel(const void* f1, const void* f2, const float* tstats, float* stats, int hw,
                                                      int C, int dtype) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave, which = blockIdx.y, p = blockIdx.z;
    if (row >= hw) return;
    const void* f = which ? f2 : f1;
    const long fo = ((long)p * hw + row) * C;
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
    if (lane == 0) {
        const long o = (((long)p * 2 + which) * hw + row) * 4;
        stats[o] = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        stats[o + 1] = tstats[o];
    }
}

// the same stats rows when the caller already holds the inverse row norms (gd_tap_mean_norm_fwd took them while it wrote the
// features): 2 P hw threads instead of a pass over the features
__global__ __launch_bounds__(256) void cv_stats_init_kernel(const float* inv1, const float* inv2, const float* tstats, float* stats, int hw,
                                                            long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // (p, which, row)
    if (i >= n) return;
    const long p = i / (2L * hw), rem = i - p * 2L * hw;
    const int which = (int)(rem / hw), row = (int)(rem - (long)which * hw);
    stats[i * 4] = (which ? inv2 : inv1)[p * hw + row];
    stats[i * 4 + 1] = tstats[i * 4];
}

struct CvTileParams {
    const void* f1; const void* f2; const float* t1; const float* t2;
    float* stats; float* part1; float* part2;
    int hw, C, tiles, nslab, ldt, P, dbg;
    // backward only
    const unsigned char* m1; const unsigned char* m2; const float* gloss;
    void* G1; void* G2; int hwp;
    const int* idx; int kcap;   // kept-row form (cv_fwd_rows_kernel): [P][2][kcap] kept-row indices (padded with a kept index), kcap % 128 == 0
    const void* fc; const int* cnt;      // kept-row backward (cv_bwd_rows_kernel): compacted kept feature rows [2][P][kcap][C], kept counts [P][2]
    const float* gscale;      // fp16 G (tf32h engine): device scalar s multiplied into G before it is rounded; the GEMMs that contract G undo it
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
    const int ldt = q.ldt;
    const float* T1 = q.t1 + (long)p * hw * ldt;
    const float* T2 = q.t2 + (long)p * hw * ldt;
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
        for (int h = 0; h < 2; ++h) t1v[st][h] = cv_ld4(T1, (long)row * ldt, tn * 128 + h * 64 + 4 * c, hw, row < hw);
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
            t2v[st][h] = cv_ld4(T2, (long)trow * ldt, tm * 128 + h * 64 + 4 * (c ^ (jl & 15)), hw, trow < hw);
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
            *(f32x2*)(q.part1 + (((long)p * q.nslab + tn) * hw + row) * 2) = f32x2{sZr[rl] + sZr[128 + rl], B};
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
            *(f32x2*)(q.part2 + (((long)p * q.nslab + tm) * hw + col) * 2) = f32x2{sZc[jl] + sZc[128 + jl], B};
    }
}

// ---------------------------------------------------------------------------------------------------
// PERSISTENT forward (the default for K rows that are a multiple of 128 bytes).  The tile kernel above serialises its
// three resources — teacher loads, LDS-DMA feature stages and the epilogue sweeps sit in ONE in-order vector-memory
// queue per wave, and the S tile is parked in LDS twice.  Here the roles are split over the waves of one 768-thread
// block per CU that walks its XCD's share of the (pair, tile) space:
//   * 4 LOADER waves own the LDS-DMA: a 4-slot ring of 128-byte K-steps (A 128 rows + B 128 rows = 32 KB per slot) keeps
//     three steps in flight ACROSS tile boundaries (the next tile's first steps are already landing during the current
//     epilogue), the tile's row / column statistics ride along as one more DMA piece, and the loaders also carry the
//     previous tile's partial sums from LDS to the slabs — they issue nothing but DMA and stores, so counted
//     `s_waitcnt vmcnt(16)` is all their synchronisation with memory;
//   * 8 COMPUTE waves (4 x 2, wave tile 32 x 64) run the MFMAs and then the whole epilogue FROM THE ACCUMULATORS: the
//     teacher tiles are loaded straight into the accumulator layout — direction 2 (tile column = teacher row) as one
//     16-byte load per 16x16 block (four consecutive rows of S are four consecutive entries of the teacher row), direction 1
//     as dword loads whose 16 lanes cover a 64-byte row segment — one tile AHEAD: a wave re-issues its teacher loads
//     right after the epilogue that consumed the registers, and they land under the next main loop.  Nothing but
//     teacher loads is in a compute wave's vector-memory queue, so they never sit in front of a feature stage.
//     S never touches LDS; per element: one v_exp (both directions' Z), and max + fma per direction.
//   One s_barrier per K-step joins all 12 waves (step n landed / slot n-1 free); the epilogue has none.
// ---------------------------------------------------------------------------------------------------
#define CV_FCH 8      // loss chunks per pair (cv_finalize_*)
#define CVP_SLOTS 4
#define CVP_STAGE 32768
#define CVP_STAT_OFF (CVP_SLOTS * CVP_STAGE)            // 2 x 256 float4 (tile parity)
#define CVP_PART_OFF (CVP_STAT_OFF + 2 * 4096)          // 2 x 1536 floats: rowsZ[2][128] rowsB[2][128] colsZ[4][128] colsB[4][128]
#define CVP_SMEM (CVP_PART_OFF + 2 * 6144)

__device__ __forceinline__ void cvp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float cvp_lds_f32(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// LDS row (64 w + 16 jb + c) of the view-2 (column) operand holds tile column 64 w + 4 c + jb: after the MFMAs a lane's four n-tiles
// are FOUR CONSECUTIVE columns, so a teacher row segment of direction 1 is one aligned 16-byte load per (row, lane) — 16 lanes cover
// 256 contiguous bytes of the teacher row — instead of four dword loads from four 64-byte segments (the same permutation the
// persistent GEMM uses for its stores, gemm_persist.h nperm64).
__device__ __forceinline__ int cv_nperm64(int rho) { return (rho & ~63) | ((rho & 15) << 2) | ((rho >> 4) & 3); }
struct CvpTile { int p, tm, tn; };
__device__ __forceinline__ CvpTile cvp_tile(int l, int tiles) {
    const int t2 = tiles * tiles;
    CvpTile t;
    t.p = l / t2;
    const int r = l - t.p * t2;
    t.tm = r / tiles;
    t.tn = r - t.tm * tiles;
    return t;
}

// BWD (round 6): the same tile walk, ring, teacher prefetch and main loop with the BACKWARD's epilogue — G = dloss/dS of both directions from the
// accumulators and the two teacher tiles in registers, stored straight from the accumulator layout as G1[i][j] = G inv2[j] (four consecutive columns per
// lane: 16 lanes write one 128-byte row segment of 16-bit G) and G2[j][i] = G inv1[i] (four consecutive rows per lane).  It replaces the one-tile-per-block
// cv_bwd_tile_kernel (dword teacher loads, two LDS round trips, 2-byte stores) wherever the forward's persistent kernel runs.
template <typename T> __device__ __forceinline__ void cv_st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void cv_st4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void cv_st4<bf16>(bf16* p, f32x4 v) { *(bf16x4*)p = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]}; }
template <> __device__ __forceinline__ void cv_st4<f16>(f16* p, f32x4 v) { *(f16x4*)p = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]}; }

// The two epilogues of the persistent kernel, from the accumulators (wave tile 32 x 64: rows wm*32 + ib*16 + 4g + r, columns wn*64 + 4c + jb).
__device__ __forceinline__ void cvp_epilogue_fwd(const f32x4 (&acc)[2][4], const f32x4 (&t1v)[2][4], const f32x4 (&t2v)[2][4], const f32x4* sSt, float* sP,
                                                 const CvpTile& t, int hw, int wm, int wn, int g, int c) {
        constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
        // Entries past the ragged edge (tile rows / columns >= hw; their operands are clamped re-reads): the row's / column's inverse norm is
        // zeroed, which makes s' = 0 — nothing enters a B sum — and e = exp2(0) = 1 exactly; the Z sums of the VALID rows and columns are
        // corrected by the count of ones they collected (those of invalid rows / columns are never stored).  No per-entry test or select.
        float inv2[4], thr2[4], badc = 0.f, badr = 0.f;
        f32x2 zc[4], b2[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const int cl = wn * 64 + 4 * c + jb;
            const f32x4 v = sSt[128 + cl];
            const bool ok = t.tn * 128 + cl < hw;
            inv2[jb] = ok ? v[0] : 0.f; thr2[jb] = CV_EPS * v[1];
            badc += ok ? 0.f : 1.f;
            zc[jb] = f32x2{0.f, 0.f}; b2[jb] = f32x2{0.f, 0.f};
        }
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            f32x4 inv1, thr1, zr = {-badc, -badc, -badc, -badc}, b1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = wm * 32 + ib * 16 + 4 * g + r;
                const f32x4 v = sSt[rl];
                const bool ok = t.tm * 128 + rl < hw;
                inv1[r] = ok ? v[0] * LOG2E : 0.f; thr1[r] = CV_EPS * v[1];
                badr += ok ? 0.f : 1.f;
            }
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const f32x4 sv = acc[ib][jb] * inv1 * inv2[jb];
                f32x4 e;
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(sv[r]);
                zr += e;
                zc[jb] += f32x2{e[0], e[1]};
                zc[jb] += f32x2{e[2], e[3]};
                f32x4 m2;
#pragma unroll
                for (int r = 0; r < 4; ++r) m2[r] = fmaxf(t2v[ib][jb][r], thr2[jb]);
                b2[jb] = __builtin_elementwise_fma(f32x2{m2[0], m2[1]}, f32x2{sv[0], sv[1]}, b2[jb]);
                b2[jb] = __builtin_elementwise_fma(f32x2{m2[2], m2[3]}, f32x2{sv[2], sv[3]}, b2[jb]);
#pragma unroll
                for (int r = 0; r < 4; ++r) b1[r] = fmaf(fmaxf(t1v[ib][r][jb], thr1[r]), sv[r], b1[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = wm * 32 + ib * 16 + 4 * g + r;
                const float z = row16_sum(zr[r]), b = row16_sum(b1[r]);
                if (c == 0) { sP[wn * 128 + rl] = z; sP[256 + wn * 128 + rl] = b * (LN2 * __builtin_amdgcn_rcpf(sSt[rl][1])); }      // (the sum re-read: a register less per row)
            }
        }
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            float z = (zc[jb][0] + zc[jb][1]) - badr, b = b2[jb][0] + b2[jb][1];
            z += __shfl_xor(z, 16, 64); z += __shfl_xor(z, 32, 64);
            b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            if (g == 0) {
                const int cl = wn * 64 + 4 * c + jb;
                sP[512 + wm * 128 + cl] = z; sP[1024 + wm * 128 + cl] = b * (LN2 * __builtin_amdgcn_rcpf(sSt[128 + cl][1]));
            }
        }
}

template <typename T>
__device__ __forceinline__ void cvp_epilogue_bwd(const CvTileParams& q, const f32x4 (&acc)[2][4], const f32x4 (&t1v)[2][4], const f32x4 (&t2v)[2][4], const f32x4* sSt,
                                                 const CvpTile& t, int hw, int wm, int wn, int g, int c, unsigned keepbits) {
        // G[i][j] = coef ( e^S (W1_i / Z1_i + W2_j / Z2_j) - max(T1[i][j], eps R1_i) / R1_i - max(T2[j][i], eps R2_j) / R2_j ), a direction's terms only for
        // its kept rows (W, R: the teacher row's clamped sums; Z: the student's softmax denominator saved by the forward; coef = dloss_p / (2 hw), times the
        // fp16 range scale of the tf32h engine).  Entries past the ragged edge: the zeroed inverse norm of their row / column makes what is stored 0.
        constexpr float LOG2E = 1.4426950408889634f;
        const int hwp = q.hwp;
        const float coef = q.gloss[t.p] * 0.5f / (float)hw * (q.gscale ? q.gscale[0] : 1.0f);
        T* G1 = (T*)q.G1 + (long)t.p * hw * hwp;
        T* G2 = (T*)q.G2 + (long)t.p * hw * hwp;
        float inv2[4], thr2[4], u2[4], q2[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const int cl = wn * 64 + 4 * c + jb;
            const f32x4 v = sSt[128 + cl];
            const bool ok = t.tn * 128 + cl < hw, keep = ok && (keepbits & (256u << jb));
            inv2[jb] = ok ? v[0] : 0.f; thr2[jb] = CV_EPS * v[1];
            u2[jb] = keep ? coef * v[3] * __builtin_amdgcn_exp2f(-v[2] * LOG2E) : 0.f;
            q2[jb] = keep ? coef * __builtin_amdgcn_rcpf(v[1]) : 0.f;
        }
        const int colg = t.tn * 128 + wn * 64 + 4 * c;
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            f32x4 inv1, inv1l, thr1, u1, q1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = wm * 32 + ib * 16 + 4 * g + r;
                const f32x4 v = sSt[rl];
                const bool ok = t.tm * 128 + rl < hw, keep = ok && (keepbits & (1u << (ib * 4 + r)));
                inv1[r] = ok ? v[0] : 0.f; inv1l[r] = inv1[r] * LOG2E; thr1[r] = CV_EPS * v[1];
                u1[r] = keep ? coef * v[3] * __builtin_amdgcn_exp2f(-v[2] * LOG2E) : 0.f;
                q1[r] = keep ? coef * __builtin_amdgcn_rcpf(v[1]) : 0.f;
            }
            const int rowg = t.tm * 128 + wm * 32 + ib * 16 + 4 * g;
            f32x4 g1[4];      // [r], element jb: four consecutive columns of one row
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const f32x4 sv = acc[ib][jb] * inv1l * inv2[jb];
                f32x4 gg;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(sv[r]);
                    gg[r] = e * (u1[r] + u2[jb]) - fmaxf(t1v[ib][r][jb], thr1[r]) * q1[r] - fmaxf(t2v[ib][jb][r], thr2[jb]) * q2[jb];
                    g1[r][jb] = gg[r] * inv2[jb];
                }
                const int col = colg + jb;
                if (col < hw && rowg < hwp) cv_st4<T>(G2 + (long)col * hwp + rowg, gg * inv1);
            }
            if (colg < hwp) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rowg + r < hw) cv_st4<T>(G1 + (long)(rowg + r) * hwp + colg, g1[r]);
            }
        }
}

template <typename T, bool DBG, bool BWD = false>      // DBG: the GD_CV_DBG anatomy switches (parts of the kernel turned off); never instantiated into the product path
__global__ __launch_bounds__(768) void cv_fwd_persist_kernel(CvTileParams q) {
    __shared__ __attribute__((aligned(16))) char smem[CVP_SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = q.hw, tiles = q.tiles, ldt = q.ldt;
    const long rowb = (long)q.C * sizeof(T);
    const int nk = (int)(rowb / 128);
    // this block's share: XCD x (blocks b and b + 8 share one) takes a contiguous range of the pair-major tile list, its
    // blocks stride through it — the ~32 tiles in flight on an XCD belong to one or two pairs: their features stay in its L2
    const int total = q.P * tiles * tiles, nbx = gridDim.x >> 3, xc = blockIdx.x & 7, kb = blockIdx.x >> 3;
    const int qT = total >> 3, rT = total & 7;
    const int beg = xc < rT ? xc * (qT + 1) : rT * (qT + 1) + (xc - rT) * qT;
    const int cnt = qT + (xc < rT ? 1 : 0);
    const int n_tiles = cnt > kb ? (cnt - kb + nbx - 1) / nbx : 0;
    if (n_tiles == 0) return;
    const int n_total = n_tiles * nk;
    const unsigned smem_base = (unsigned)(uintptr_t)smem;      // LDS byte address of the block's array (generic -> LDS offset)

    if (wave >= 8) {
        // ======================================= LOADER waves =======================================
        const int lw = wave - 8;
        const char* asrc[4];
        const char* wsrc[4];
        const char* ssrc = nullptr;
        int it_i = 0, k_i = 0;                 // (tile, K-step) of the next step to issue
        auto issue = [&](int n) {
            char* sA = smem + (n & (CVP_SLOTS - 1)) * CVP_STAGE;
            char* sB = sA + 128 * 128;
            if (k_i == 0) {
                const CvpTile t = cvp_tile(beg + kb + it_i * nbx, tiles);
                const char* Ab = (const char*)q.f1 + (long)t.p * hw * rowb;
                const char* Wb = (const char*)q.f2 + (long)t.p * hw * rowb;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = (lw * 4 + i) * 8 + (lane >> 3);
                    asrc[i] = Ab + (long)min(t.tm * 128 + row, hw - 1) * rowb + (((lane & 7) ^ swz(row)) * 16);
                    wsrc[i] = Wb + (long)min(t.tn * 128 + cv_nperm64(row), hw - 1) * rowb + (((lane & 7) ^ swz(row)) * 16);
                }
                const int e = lw * 64 + lane, which = e >> 7;
                const int idx = min((which ? t.tn : t.tm) * 128 + (e & 127), hw - 1);
                ssrc = (const char*)(q.stats + (((long)t.p * 2 + which) * hw + idx) * 4);
                // the tile's 256 row / column statistics: first (oldest) piece of the step, parity buffer of the tile
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ssrc,
                                                 (__attribute__((address_space(3))) void*)(smem + CVP_STAT_OFF + (it_i & 1) * 4096 + lw * 1024),
                                                 16, 0, 0);
            }
            if (!(DBG && (q.dbg & 64))) {      // (anatomy bit 64: no feature DMA — the teacher stream alone)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + (long)k_i * 128),
                                                     (__attribute__((address_space(3))) void*)(sA + (lw * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + (long)k_i * 128),
                                                     (__attribute__((address_space(3))) void*)(sB + (lw * 4 + i) * 1024), 16, 0, 0);
            }
            if (++k_i == nk) { k_i = 0; ++it_i; }
        };
        // previous tile's partial sums: LDS (written by the compute waves before the barrier just passed) -> slabs
        const uintptr_t p1 = (uintptr_t)q.part1, p2 = (uintptr_t)q.part2;
        auto flush = [&](int it) {
            const CvpTile t = cvp_tile(beg + kb + it * nbx, tiles);
            const unsigned pb = smem_base + CVP_PART_OFF + (it & 1) * 6144;
            const int e = lw * 64 + lane;          // 0..127 rows, 128..255 columns
            // one code path for both halves (a pointer picked by a per-lane branch became a vector load from the kernel
            // argument block + s_waitcnt vmcnt(0): it drained the loader's DMA ring once per tile)
            const bool isrow = e < 128;
            const int c = e & 127;
            const unsigned zo = isrow ? c : 512 + c, bo = isrow ? 256 + c : 1024 + c;
            float Z = cvp_lds_f32(pb + zo * 4) + cvp_lds_f32(pb + (zo + 128) * 4);
            float B = cvp_lds_f32(pb + bo * 4) + cvp_lds_f32(pb + (bo + 128) * 4);
            const float Z2 = cvp_lds_f32(pb + (512 + 256 + c) * 4) + cvp_lds_f32(pb + (512 + 384 + c) * 4);
            const float B2 = cvp_lds_f32(pb + (1024 + 256 + c) * 4) + cvp_lds_f32(pb + (1024 + 384 + c) * 4);
            if (!isrow) { Z += Z2; B += B2; }          // columns: four wave rows, fixed order ((w0 + w1) + (w2 + w3)): deterministic
            const int idx = (isrow ? t.tm : t.tn) * 128 + c, slab = isrow ? t.tn : t.tm;
            const uintptr_t base = isrow ? p1 : p2;
            if (idx < hw)       // explicitly a GLOBAL store: a flat store is out of order with respect to vmcnt
                *(__attribute__((address_space(1))) f32x2*)(base + ((((long)t.p * q.nslab + slab) * hw + idx) * 2) * sizeof(float)) = f32x2{Z, B};
        };
        for (int n = 0; n < 3 && n < n_total; ++n) issue(n);
        int kk = 0, it = 0;
        for (int n = 0; n < n_total; ++n) {
            const int rem = n_total - 1 - n;
            // every operation older than the two youngest steps (8 DMA pieces each) has landed: step n is in LDS
            if (rem >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (n + 3 < n_total) issue(n + 3);
            if (!BWD && kk == 0 && it > 0) flush(it - 1);
            if (++kk == nk) { kk = 0; ++it; }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!BWD) flush(n_tiles - 1);
        return;
    }

    // ======================================= COMPUTE waves =======================================
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c = lane & 15;
    typedef typename Mma<T>::Frag Frag;
    const int sa = swz(c);
    const int abase = (wm * 32 + c) * 128, bbase = 128 * 128 + (wn * 64 + c) * 128;
    // this lane's columns of the tile: cl(jb) = wn*64 + 4c + jb (see cv_nperm64); its rows: wm*32 + ib*16 + 4g + r
    f32x4 t1v[2][4];         // direction 1: T1[row = tm*128 + wm*32 + ib*16 + 4g + r][col = tn*128 + wn*64 + 4c .. +3]   ([ib][r], element jb)
    f32x4 t2v[2][4];         // direction 2: T2[row = tn*128 + wn*64 + 4c + jb][col = tm*128 + wm*32 + ib*16 + 4g .. +3]   ([ib][jb], element r)
    // branch-free: every address is clamped into the pair's map (rows / columns past hw re-read valid entries; the
    // epilogue multiplies them by a zeroed s), so the 16 loads of a tile go out back to back with no wait between them
    unsigned keepbits = 0xfffu;
    auto prefetch = [&](int it) {
        const CvpTile t = cvp_tile(beg + kb + it * nbx, tiles);
        const float* T1 = q.t1 + (long)t.p * hw * ldt;
        const float* T2 = q.t2 + (long)t.p * hw * ldt;
        // Rows / columns that the loss masks out (utils/functions.py:402-422 zeroes them; cv_finalize replaces their term by a
        // constant) never need their teacher entries: their loads are pointed at the first line of the pair's map instead — still
        // branch-free and back to back, but an L2 hit instead of HBM traffic.  With the MASt3R trainer's keypoint-patch masks
        // (~20 % of the patches hold a keypoint) four fifths of both teacher maps are never fetched.
        bool rk[2][4], ck[4];
        if (q.m1 && !(DBG && (q.dbg & 32))) {
            const unsigned char* M1 = q.m1 + (long)t.p * hw;
            const unsigned char* M2 = q.m2 + (long)t.p * hw;
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) rk[ib][r] = M1[min(t.tm * 128 + wm * 32 + ib * 16 + 4 * g + r, hw - 1)] != 0;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) ck[jb] = M2[min(t.tn * 128 + wn * 64 + 4 * c + jb, hw - 1)] != 0;
        } else {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) rk[ib][r] = true;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) ck[jb] = true;
        }
        if (BWD) {      // the backward's epilogue needs the masks themselves: one bit per row / column of this lane, valid until the next prefetch
            keepbits = 0;
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) keepbits |= rk[ib][r] ? 1u << (ib * 4 + r) : 0u;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) keepbits |= ck[jb] ? 256u << jb : 0u;
        }
        const int col0 = min(t.tn * 128 + wn * 64 + 4 * c, ldt - 4);          // ldt % 4 == 0: aligned, inside the row
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const int row0 = t.tm * 128 + wm * 32 + ib * 16 + 4 * g;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const int col = min(t.tn * 128 + wn * 64 + 4 * c + jb, hw - 1);
                t2v[ib][jb] = *(const f32x4*)(T2 + (ck[jb] ? (long)col * ldt + min(row0, ldt - 4) : 0L));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) t1v[ib][r] = *(const f32x4*)(T1 + (rk[ib][r] ? (long)min(row0 + r, hw - 1) * ldt + col0 : 0L));
        }
    };
    const int dbg = DBG ? q.dbg : 0;     // diagnostics (GD_CV_DBG): 1 = no teacher loads, 2 = no epilogue math, 4 = no MFMAs
    if (dbg & 1) {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) t2v[ib][jb] = t1v[ib][jb] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else prefetch(0);
    int n = 0;
    for (int it = 0; it < n_tiles; ++it) {
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < nk; ++k, ++n) {
            cvp_barrier();
            const char* sb = smem + (n & (CVP_SLOTS - 1)) * CVP_STAGE;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const int co = (((kc * 4 + g) ^ sa) * 16);
                Frag a[2], b[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t] = *(const Frag*)(sb + bbase + t * 2048 + co);
#pragma unroll
                for (int t = 0; t < 2; ++t) a[t] = *(const Frag*)(sb + abase + t * 2048 + co);
                if (!(dbg & 4)) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(a[i], b[j], acc[i][j]);
                } else acc[0][0][0] += (float)a[0][0] + (float)b[0][0] + (float)a[1][0] + (float)b[1][0] + (float)b[2][0] + (float)b[3][0];
            }
        }
        if (dbg & 2) {
            if (it + 1 < n_tiles && !(dbg & 1)) prefetch(it + 1);
            if (acc[0][0][0] == 12345.678f) ((float*)smem)[tid] = acc[1][1][1] + t1v[0][0][0] + t2v[1][1][1];
            continue;
        }
        // ---------------- epilogue, all from registers ----------------
        // Round 5: 14 -> ~7 VALU instructions per entry (the epilogue was the largest single phase of a tile: ~45 issued instructions per entry and lane
        // with its reductions, all waves of the CU in it at the same time).  (a) scores in the log2 domain: s' = acc * (inv1 log2 e) * inv2, e = exp2(s'),
        // the B sums carry s' and are scaled by ln 2 once per row / column; (b) max(t / rowsum, EPS) * s = (1 / rowsum) * max(t, EPS rowsum) * s: the
        // division leaves the loop the same way; (c) no bounds tests per entry (below); (d) the r-pairs of an accumulator as packed
        // fp32 operations (v_pk_mul / v_pk_add / v_pk_fma_f32).
        const CvpTile t = cvp_tile(beg + kb + it * nbx, tiles);
        const f32x4* sSt = (const f32x4*)(smem + CVP_STAT_OFF + (it & 1) * 4096);
        float* sP = (float*)(smem + CVP_PART_OFF + (it & 1) * 6144);
        if (BWD) cvp_epilogue_bwd<T>(q, acc, t1v, t2v, sSt, t, hw, wm, wn, g, c, keepbits);
        else cvp_epilogue_fwd(acc, t1v, t2v, sSt, sP, t, hw, wm, wn, g, c);
        if (it + 1 < n_tiles && !(dbg & 1)) prefetch(it + 1);
    }
    cvp_barrier();
}

// ---------------------------------------------------------------------------------------------------
// KEPT-ROW form of the persistent forward (sparse row masks: the MASt3R trainer's keypoint-patch masks keep <= N_kp of the hw rows).
// A masked-out row's KL term is a constant and needs neither its teacher row nor its scores — and direction 2's "rows" are S's columns —
// so the two directions are computed as TWO compacted row problems instead of one hw x hw sweep:
//   direction d (0 | 1): A = the kept rows of view d's features, gathered through an index list (the LDS-DMA takes per-lane addresses: the
//   gather is free), B = ALL rows of the other view, teacher rows = the kept rows of T_d (row-contiguous), statistics per kept row only.
// Tile space [pair][direction][kcap / 128 row tiles][hw / 128 column tiles] instead of [pair][tiles][tiles]: 2 x 3 x 11 = 66 tiles per pair at 300
// keypoints against 121, each with HALF the epilogue (one direction), every teacher byte it loads needed.  Same ring, roles and waits as
// cv_fwd_persist_kernel; what differs is marked ROWS.
template <typename T>
__global__ __launch_bounds__(768) void cv_fwd_rows_kernel(CvTileParams q) {
    constexpr bool DBG = false;
    __shared__ __attribute__((aligned(16))) char smem[CVP_SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = q.hw, tiles = q.tiles, ldt = q.ldt;
    const long rowb = (long)q.C * sizeof(T);
    const int nk = (int)(rowb / 128);
    // this block's share: XCD x (blocks b and b + 8 share one) takes a contiguous range of the pair-major tile list, its
    // blocks stride through it — the ~32 tiles in flight on an XCD belong to one or two pairs: their features stay in its L2
    const int tiles_r = q.kcap >> 7, kcap = q.kcap;                  // ROWS: row tiles over the compacted (kept) rows
    const int total = q.P * 2 * tiles_r * tiles, nbx = gridDim.x >> 3, xc = blockIdx.x & 7, kb = blockIdx.x >> 3;
    const int qT = total >> 3, rT = total & 7;
    const int beg = xc < rT ? xc * (qT + 1) : rT * (qT + 1) + (xc - rT) * qT;
    const int cnt = qT + (xc < rT ? 1 : 0);
    const int n_tiles = cnt > kb ? (cnt - kb + nbx - 1) / nbx : 0;
    if (n_tiles == 0) return;
    const int n_total = n_tiles * nk;
    const unsigned smem_base = (unsigned)(uintptr_t)smem;      // LDS byte address of the block's array (generic -> LDS offset)
    // ROWS: tile l -> (pd = 2 pair + direction, tm = row tile of the kept rows, tn = column tile); .p holds pd
    auto rtile = [&](int l) __attribute__((always_inline)) {
        const int per = tiles_r * tiles;
        CvpTile t;
        t.p = l / per;
        const int r = l - t.p * per;
        t.tm = r / tiles;
        t.tn = r - t.tm * tiles;
        return t;
    };

    if (wave >= 8) {
        // ======================================= LOADER waves =======================================
        const int lw = wave - 8;
        const char* asrc[4];
        const char* wsrc[4];
        const char* ssrc = nullptr;
        int it_i = 0, k_i = 0;                 // (tile, K-step) of the next step to issue
        auto issue = [&](int n) {
            char* sA = smem + (n & (CVP_SLOTS - 1)) * CVP_STAGE;
            char* sB = sA + 128 * 128;
            if (k_i == 0) {
                const CvpTile t = rtile(beg + kb + it_i * nbx);
                const int pr = t.p >> 1, dir = t.p & 1;
                const char* Ab = (const char*)(dir ? q.f2 : q.f1) + (long)pr * hw * rowb;      // ROWS: view d's kept rows against all rows of the other view
                const char* Wb = (const char*)(dir ? q.f1 : q.f2) + (long)pr * hw * rowb;
                const int* ix = q.idx + (long)t.p * kcap + t.tm * 128;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = (lw * 4 + i) * 8 + (lane >> 3);
                    asrc[i] = Ab + (long)ix[row] * rowb + (((lane & 7) ^ swz(row)) * 16);      // ROWS: gathered (the list is padded with valid indices)
                    wsrc[i] = Wb + (long)min(t.tn * 128 + cv_nperm64(row), hw - 1) * rowb + (((lane & 7) ^ swz(row)) * 16);
                }
                const int e = lw * 64 + lane, which = e >> 7;
                const int idx = which ? min(t.tn * 128 + (e & 127), hw - 1) : ix[e & 127];
                ssrc = (const char*)(q.stats + (((long)pr * 2 + (which ? 1 - dir : dir)) * hw + idx) * 4);
                // the tile's 256 row / column statistics: first (oldest) piece of the step, parity buffer of the tile
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ssrc,
                                                 (__attribute__((address_space(3))) void*)(smem + CVP_STAT_OFF + (it_i & 1) * 4096 + lw * 1024),
                                                 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + (long)k_i * 128),
                                                 (__attribute__((address_space(3))) void*)(sA + (lw * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + (long)k_i * 128),
                                                 (__attribute__((address_space(3))) void*)(sB + (lw * 4 + i) * 1024), 16, 0, 0);
            if (++k_i == nk) { k_i = 0; ++it_i; }
        };
        // previous tile's partial sums: LDS (written by the compute waves before the barrier just passed) -> slabs
        const uintptr_t p1 = (uintptr_t)q.part1, p2 = (uintptr_t)q.part2;
        auto flush = [&](int it) {
            const CvpTile t = rtile(beg + kb + it * nbx);
            const unsigned pb = smem_base + CVP_PART_OFF + (it & 1) * 6144;
            const int e = lw * 64 + lane;          // 0..127 rows, 128..255 columns
            // one code path for both halves (a pointer picked by a per-lane branch became a vector load from the kernel
            // argument block + s_waitcnt vmcnt(0): it drained the loader's DMA ring once per tile)
            const bool isrow = e < 128;
            const int c = e & 127;
            const unsigned zo = isrow ? c : 512 + c, bo = isrow ? 256 + c : 1024 + c;
            float Z = cvp_lds_f32(pb + zo * 4) + cvp_lds_f32(pb + (zo + 128) * 4);
            float B = cvp_lds_f32(pb + bo * 4) + cvp_lds_f32(pb + (bo + 128) * 4);
            const float Z2 = cvp_lds_f32(pb + (512 + 256 + c) * 4) + cvp_lds_f32(pb + (512 + 384 + c) * 4);
            const float B2 = cvp_lds_f32(pb + (1024 + 256 + c) * 4) + cvp_lds_f32(pb + (1024 + 384 + c) * 4);
            (void)Z2; (void)B2; (void)p2;
            // ROWS: row partials only, in the compacted row space [pd][column tile][kcap]
            if (isrow)          // explicitly a GLOBAL store: a flat store is out of order with respect to vmcnt
                *(__attribute__((address_space(1))) f32x2*)(p1 + ((((long)t.p * tiles + t.tn) * kcap + t.tm * 128 + c) * 2) * sizeof(float)) = f32x2{Z, B};
        };
        for (int n = 0; n < 3 && n < n_total; ++n) issue(n);
        int kk = 0, it = 0;
        for (int n = 0; n < n_total; ++n) {
            const int rem = n_total - 1 - n;
            // every operation older than the two youngest steps (8 DMA pieces each) has landed: step n is in LDS
            if (rem >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (n + 3 < n_total) issue(n + 3);
            if (kk == 0 && it > 0) flush(it - 1);
            if (++kk == nk) { kk = 0; ++it; }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        flush(n_tiles - 1);
        return;
    }

    // ======================================= COMPUTE waves =======================================
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c = lane & 15;
    typedef typename Mma<T>::Frag Frag;
    const int sa = swz(c);
    const int abase = (wm * 32 + c) * 128, bbase = 128 * 128 + (wn * 64 + c) * 128;
    // this lane's columns of the tile: cl(jb) = wn*64 + 4c + jb (see cv_nperm64); its rows: wm*32 + ib*16 + 4g + r
    f32x4 t1v[2][4];         // direction 1: T1[row = tm*128 + wm*32 + ib*16 + 4g + r][col = tn*128 + wn*64 + 4c .. +3]   ([ib][r], element jb)
    // branch-free: every address is clamped into the pair's map (rows / columns past hw re-read valid entries; the
    // epilogue multiplies them by a zeroed s), so the 16 loads of a tile go out back to back with no wait between them
    auto prefetch = [&](int it) {      // ROWS: direction-d teacher rows of the tile's kept rows (all needed: no mask tests), nothing for the columns
        const CvpTile t = rtile(beg + kb + it * nbx);
        const int pr = t.p >> 1, dir = t.p & 1;
        const float* T1 = (dir ? q.t2 : q.t1) + (long)pr * hw * ldt;
        const int* ix = q.idx + (long)t.p * kcap + t.tm * 128 + wm * 32 + 4 * g;
        const int col0 = min(t.tn * 128 + wn * 64 + 4 * c, ldt - 4);          // ldt % 4 == 0: aligned, inside the row
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) t1v[ib][r] = *(const f32x4*)(T1 + (long)ix[ib * 16 + r] * ldt + col0);
    };
    const int dbg = DBG ? q.dbg : 0;     // diagnostics (GD_CV_DBG): 1 = no teacher loads, 2 = no epilogue math, 4 = no MFMAs
    if (dbg & 1) {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) t1v[ib][jb] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else prefetch(0);
    int n = 0;
    for (int it = 0; it < n_tiles; ++it) {
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < nk; ++k, ++n) {
            cvp_barrier();
            const char* sb = smem + (n & (CVP_SLOTS - 1)) * CVP_STAGE;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const int co = (((kc * 4 + g) ^ sa) * 16);
                Frag a[2], b[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t] = *(const Frag*)(sb + bbase + t * 2048 + co);
#pragma unroll
                for (int t = 0; t < 2; ++t) a[t] = *(const Frag*)(sb + abase + t * 2048 + co);
                if (!(dbg & 4)) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(a[i], b[j], acc[i][j]);
                } else acc[0][0][0] += (float)a[0][0] + (float)b[0][0] + (float)a[1][0] + (float)b[1][0] + (float)b[2][0] + (float)b[3][0];
            }
        }
        if (dbg & 2) {
            if (it + 1 < n_tiles && !(dbg & 1)) prefetch(it + 1);
            if (acc[0][0][0] == 12345.678f) ((float*)smem)[tid] = acc[1][1][1] + t1v[0][0][0];
            continue;
        }
        // ---------------- epilogue, all from registers ----------------
        const CvpTile t = rtile(beg + kb + it * nbx);
        const f32x4* sSt = (const f32x4*)(smem + CVP_STAT_OFF + (it & 1) * 4096);
        float* sP = (float*)(smem + CVP_PART_OFF + (it & 1) * 6144);
        // (the slimmed epilogue of cv_fwd_persist_kernel, one direction: log2-domain scores, the division by the teacher row sum and ln 2 applied to
        //  the row's B sum, no per-entry bounds tests)
        {
            constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
            float inv2[4], badc = 0.f;      // columns past hw: inverse norm 0 -> s' = 0, e = 1 exactly, counted out of the rows' Z below
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const int cl = wn * 64 + 4 * c + jb;
                const bool ok = t.tn * 128 + cl < hw;
                inv2[jb] = ok ? sSt[128 + cl][0] : 0.f;
                badc += ok ? 0.f : 1.f;
            }
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                // (rows past the kept count are padded copies of a kept row: finite, never read by the finalize pass)
                f32x4 inv1, thr1, zr = {-badc, -badc, -badc, -badc}, b1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x4 v = sSt[wm * 32 + ib * 16 + 4 * g + r];
                    inv1[r] = v[0] * LOG2E; thr1[r] = CV_EPS * v[1];
                }
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const f32x4 sv = acc[ib][jb] * inv1 * inv2[jb];
#pragma unroll
                    for (int r = 0; r < 4; ++r) zr[r] += __builtin_amdgcn_exp2f(sv[r]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) b1[r] = fmaf(fmaxf(t1v[ib][r][jb], thr1[r]), sv[r], b1[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = wm * 32 + ib * 16 + 4 * g + r;
                    const float z = row16_sum(zr[r]), b = row16_sum(b1[r]);
                    if (c == 0) { sP[wn * 128 + rl] = z; sP[256 + wn * 128 + rl] = b * (LN2 * __builtin_amdgcn_rcpf(sSt[rl][1])); }
                }
            }
        }
        if (it + 1 < n_tiles && !(dbg & 1)) prefetch(it + 1);
    }
    cvp_barrier();
}



// kept rows of one (pair, direction) in ascending order -> idx[pd][0 .. cnt), the rest of the kcap entries padded with the last kept row (row 0
// when nothing is kept); cnt[pd] = the number kept, or -1 when it EXCEEDS kcap (the caller's bound was wrong: the loss of that pair comes out NaN and so do
// its gradients — never a silently truncated sum).  One block per (pair, direction).
__global__ __launch_bounds__(256) void cv_rows_compact_kernel(const unsigned char* m1, const unsigned char* m2, int* idx, int* cnt, int hw, int kcap) {
    __shared__ int sc[256];
    const int pd = blockIdx.x, tid = threadIdx.x;
    const unsigned char* m = ((pd & 1) ? m2 : m1) + (long)(pd >> 1) * hw;
    const int per = (hw + 255) / 256, r0 = tid * per, r1 = min(hw, r0 + per);
    int n = 0;
    for (int r = r0; r < r1; ++r) n += m[r] != 0;
    sc[tid] = n;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {      // inclusive scan
        const int v = tid >= o ? sc[tid - o] : 0;
        __syncthreads();
        sc[tid] += v;
        __syncthreads();
    }
    const int total = sc[255];
    int pos = sc[tid] - n;
    int* out = idx + (long)pd * kcap;
    int last = -1;
    for (int r = r0; r < r1; ++r)
        if (m[r] != 0) { if (pos < kcap) out[pos] = r; ++pos; last = r; }
    __shared__ int slast;
    if (tid == 0) slast = 0;
    __syncthreads();
    if (last >= 0) atomicMax(&slast, last);
    __syncthreads();
    const int kept = min(total, kcap);
    for (int e = kept + tid; e < kcap; e += 256) out[e] = slast;
    if (tid == 0) cnt[pd] = total > kcap ? -1 : kept;
}

// finalize of the kept-row form: Z, B summed over the column tiles per kept row, logZ / W saved at the row's ORIGINAL index for the backward,
// per-chunk partial losses (fixed order: deterministic); masked-out rows contribute their constant (hw - kept of them per direction)
__global__ __launch_bounds__(256) void cv_finalize_rows_kernel(const float* part, const float* tstats, const int* idx, const int* cnt, float* stats,
                                                               double* chunk_loss, int hw, int tiles, int kcap, int variant) {
    const int p = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
    const float masked_const = variant == 1 ? (float)hw * (CV_EPS * logf(CV_EPS * (float)hw)) : 0.f;
    double total = 0.0;
    const int per = (2 * kcap + CV_FCH - 1) / CV_FCH;
    for (int e = ch * per + tid; e < min(2 * kcap, (ch + 1) * per); e += 256) {
        const int d = e >= kcap, kk = d ? e - kcap : e, pd = p * 2 + d;
        if (kk >= cnt[pd]) continue;
        const int row = idx[(long)pd * kcap + kk];
        float Z = 0.f, B = 0.f;
        for (int sidx = 0; sidx < tiles; ++sidx) {
            const f32x2 v = *(const f32x2*)(part + (((long)pd * tiles + sidx) * kcap + kk) * 2);
            Z += v[0]; B += v[1];
        }
        const float Wt = tstats[((long)pd * hw + row) * 4 + 1], A = tstats[((long)pd * hw + row) * 4 + 2];
        const float logZ = logf(Z);
        float* st = stats + ((long)pd * hw + row) * 4;
        st[2] = logZ;
        st[3] = Wt;
        total += (double)(A - B + Wt * logZ);
    }
    if (ch == 0 && tid < 2) {
        const int c = cnt[p * 2 + tid];
        total += c < 0 ? (double)__builtin_nanf("") : (double)masked_const * (double)(hw - c);      // more kept rows than the caller's bound: poisoned, not truncated
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

// reduce the slabs, save logZ and W for the backward, emit per-chunk partial losses (CV_FCH chunks per pair), then sum
__global__ __launch_bounds__(256) void cv_finalize_kernel(const float* part1, const float* part2, const float* tstats,
                                                          const unsigned char* m1, const unsigned char* m2,
                                                          float* stats, double* chunk_loss, int hw, int nslab, int nslab2, int variant) {
    // (nslab2: slabs of the direction-2 partials — the number of row panels of the forward kernel)
    const int p = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
    const float masked_const = variant == 1 ? (float)hw * (CV_EPS * logf(CV_EPS * (float)hw)) : 0.f;
    double total = 0.0;
    const int per = (2 * hw + CV_FCH - 1) / CV_FCH;
    for (int idx = ch * per + tid; idx < min(2 * hw, (ch + 1) * per); idx += 256) {
        const int d = idx >= hw, row = d ? idx - hw : idx;
        const float* part = d ? part2 : part1;
        const int ns = d ? nslab2 : nslab;
        float Z = 0.f, B = 0.f;
        for (int s = 0; s < ns; ++s) {
            const f32x2 v = *(const f32x2*)(part + (((long)p * ns + s) * hw + row) * 2);
            Z += v[0]; B += v[1];
        }
        const float Wt = tstats[(((long)p * 2 + d) * hw + row) * 4 + 1], A = tstats[(((long)p * 2 + d) * hw + row) * 4 + 2];
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
    const int ldt = q.ldt;
    const float* T1 = q.t1 + (long)p * hw * ldt;
    const float* T2 = q.t2 + (long)p * hw * ldt;
    f32x4* sSt = (f32x4*)(smem + CV_RING);
    cv_stage_stats(q, p, tm, tn, sSt);
    float t1v[2][32];   // pass-A teacher values of both halves: in flight under the main loop
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int k = 0; k < 32; ++k) {   // pass A: tile row wave+4k, column = 64h + lane
            const int row = tm * 128 + wave + 4 * k, col = tn * 128 + 64 * h + lane;
            t1v[h][k] = (row < hw && col < hw) ? T1[(long)row * ldt + col] : 0.f;
        }
    f32x4 acc[4][4];
    cv_s_tile<T>(q, p, tm, tn, smem, sSt, acc);
    float* sS = (float*)smem;            // [128][64]
    float* sG = (float*)smem + 128 * 64; // [128][64]
    const float coef = q.gloss[p] * 0.5f / (float)hw * (q.gscale ? q.gscale[0] : 1.0f);
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
                t2v[k][e] = (row < hw && col < hw) ? T2[(long)col * ldt + row] : 0.f;
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

// ---- kept-row backward (sparse row masks): the two directions as separate compacted row problems, like cv_fwd_rows_kernel.  Block (tm, tn) of
//      (direction d, pair p) recomputes S_c = A_kept B^T for 128 kept rows x 128 columns (A_kept: the gathered feature rows q.fc, B: the other
//      view's features), forms G = dloss/dS_c and stores Gc[k][j] = G inv_B[j] ([kcap][hwp]) and GcT[j][k] = G inv_A[k] ([hw][kcap]): two batched
//      NT GEMMs per direction contract them with the other view's features (-> kept rows of this view's gradient, scattered by cv_rows_scatter)
//      and with the kept rows (-> the other view's gradient, dense).  Rows k >= cnt and columns j >= hw are written as zeros. ----
template <typename T>
__global__ __launch_bounds__(256, 2) void cv_bwd_rows_kernel(CvTileParams q) {
    __shared__ __attribute__((aligned(16))) char smem[CV_RING + 4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c = lane & 15;
    const int hw = q.hw, hwp = q.hwp, kcap = q.kcap, ldt = q.ldt;
    const int pdl = blockIdx.y, d = pdl / q.P, p = pdl - d * q.P;      // Gc / GcT / fc are laid out [direction][pair]; idx / cnt / stats [pair][direction]
    const int tiles_r = kcap / 128;
    const int wg = xcd_remap(blockIdx.x, tiles_r * q.tiles);
    const int tm = wg / q.tiles, tn = wg % q.tiles;
    const int* ix = q.idx + ((long)p * 2 + d) * kcap;
    const int cntk = q.cnt[p * 2 + d];
    T* Gc = (T*)q.G1 + (long)pdl * kcap * hwp;
    T* GcT = (T*)q.G2 + (long)pdl * hw * kcap;
    if (tm * 128 >= cntk) {      // nothing kept in this row tile: its slice of GcT is contracted against (duplicated) kept rows and has to be zero
        for (int e = tid; e < 128 * 128; e += 256) {
            const int col = tn * 128 + (e >> 7), kk = tm * 128 + (e & 127);
            if (col < hw) GcT[(long)col * kcap + kk] = from_f32<T>(0.f);
        }
        // ... and so are its Gc rows: the batched GEMM Gc . (other view) contracts every row of the [kcap, hwp] workspace, and what cv_rows_scatter
        // then skips (k >= cnt) must still not be computed from uninitialised memory (NaN / Inf bit patterns in a torch.empty buffer)
        for (int e = tid; e < 128 * 128; e += 256) {
            const int kk = tm * 128 + (e >> 7), col = tn * 128 + (e & 127);
            if (col < hwp) Gc[(long)kk * hwp + col] = from_f32<T>(0.f);
        }
        return;
    }
    const float* Tt = (d ? q.t2 : q.t1) + (long)p * hw * ldt;
    f32x4* sSt = (f32x4*)(smem + CV_RING);
    {
        const int which = tid >> 7, t = tid & 127;
        f32x4 o = {0.f, 0.f, 0.f, -1.f};
        if (which == 0) {
            const int kk = tm * 128 + t;
            const f32x4 v = *(const f32x4*)(q.stats + (((long)p * 2 + d) * hw + ix[kk]) * 4);
            o = f32x4{v[0], 1.0f / v[1], v[2], kk < cntk ? v[3] : -1.f};
        } else {
            const int j = tn * 128 + t;
            if (j < hw) o = f32x4{q.stats[(((long)p * 2 + (1 - d)) * hw + j) * 4], 0.f, 0.f, 0.f};
        }
        sSt[tid] = o;
    }
    float t1v[2][32];   // teacher values of both halves: in flight under the main loop
#pragma unroll
    for (int k = 0; k < 32; ++k) {   // tile row wave+4k (a kept row: its teacher row is ix[...]), column = 64h + lane
        const int kk = tm * 128 + wave + 4 * k;
        const long ro = (long)ix[kk] * ldt;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = tn * 128 + 64 * h + lane;
            t1v[h][k] = (kk < cntk && col < hw && !(q.dbg & 1)) ? Tt[ro + col] : 0.f;
        }
    }
    f32x4 acc[4][4];
    {
        const long rowb = (long)q.C * sizeof(T);
        const char* Ab = (const char*)q.fc + (long)pdl * kcap * rowb;
        const char* Wb = (const char*)(d ? q.f1 : q.f2) + (long)p * hw * rowb;
        if (q.dbg & 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();
        } else if (rowb % 128 == 0) dma_mainloop<T, 2, 2, 4>(Ab, rowb, kcap, Wb, rowb, hw, (int)(rowb / 128), tm, tn, smem, acc);
        else mma_tile_128x128<T>(Ab, rowb, kcap, Wb, rowb, hw, (int)rowb, tm, tn, smem, acc);
        float invc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) invc[j] = sSt[128 + wn * 64 + j * 16 + c][0];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float invr = sSt[wm * 64 + i * 16 + g * 4 + r][0];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j][r] *= invr * invc[j];
            }
    }
    float* sS = (float*)smem;            // [128][64]
    float* sG = (float*)smem + 128 * 64; // [128][64]
    const float coef = q.gloss[p] * 0.5f / (float)hw * (q.gscale ? q.gscale[0] : 1.0f);
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
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; ++k) {      // lane = column
            const int rr = wave + 4 * k;
            const f32x4 v = sSt[rr];
            float term = 0.f;
            if (v[3] >= 0.f && col0 + lane < hw && !(q.dbg & 8))
                term = (v[3] * __expf(sS[sidx(rr, lane, 64)] - v[2]) - fmaxf(t1v[h][k] * v[1], CV_EPS)) * coef;
            sG[sidx(rr, lane, 64)] = term;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {      // GcT rows (lane = tile row: contiguous k)
            const int jl = wave + 4 * k, col = col0 + jl;
            if (col < hw && !(q.dbg & 2)) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int il = lane + 64 * e;
                    GcT[(long)col * kcap + tm * 128 + il] = from_f32<T>(sG[sidx(il, jl, 64)] * sSt[il][0]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {      // Gc rows (lane = column)
            const int rr = wave + 4 * k, col = col0 + lane;
            if (col < hwp && !(q.dbg & 2)) Gc[(long)(tm * 128 + rr) * hwp + col] = from_f32<T>(sG[sidx(rr, lane, 64)] * sSt[128 + 64 * h + lane][0]);
        }
        __syncthreads();
    }
}

// kept feature rows of both views, gathered: fc[d][p][k][:] = f_d[p][idx[p][d][k]][:]   (16 bytes per thread)
__global__ __launch_bounds__(256) void cv_rows_gather_kernel(const char* f1, const char* f2, const int* idx, char* fc, int P, int hw, int kcap, int rowb) {
    const int v16 = rowb / 16;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // (d, p, k, chunk)
    if (i >= 2L * P * kcap * v16) return;
    const int ch = (int)(i % v16);
    const long r = i / v16;
    const int k = (int)(r % kcap), dp = (int)(r / kcap), d = dp / P, p = dp - d * P;
    const int row = idx[((long)p * 2 + d) * kcap + k];
    *(f32x4*)(fc + r * rowb + ch * 16L) = *(const f32x4*)((d ? f2 : f1) + ((long)p * hw + row) * rowb + ch * 16L);
}

// the kept rows' gradients added at their original rows: d_d[p][idx[p][d][k]][:] += dk[d][p][k][:] for k < cnt   (distinct rows: no atomics)
__global__ __launch_bounds__(256) void cv_rows_scatter_kernel(const float* dk, const int* idx, const int* cnt, float* d1, float* d2, int P, int hw,
                                                              int kcap, int C) {
    const int v4 = C / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2L * P * kcap * v4) return;
    const int ch = (int)(i % v4);
    const long r = i / v4;
    const int k = (int)(r % kcap), dp = (int)(r / kcap), d = dp / P, p = dp - d * P;
    const int c = cnt[p * 2 + d];
    const int row = idx[((long)p * 2 + d) * kcap + k];
    f32x4* o = (f32x4*)((d ? d2 : d1) + ((long)p * hw + row) * C) + ch;
    if (c < 0 && k == 0) { const float n = __builtin_nanf(""); *o = f32x4{n, n, n, n}; return; }      // bound exceeded: poison the pair's gradient
    if (k >= c) return;
    const f32x4 a = *o, b = *((const f32x4*)(dk + r * C) + ch);
    *o = f32x4{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
}

// out[p][c][j] = in[p][j][c] for j < hw, 0 for hw <= j < hwp   (64 x 64 tile transpose through LDS: 16-byte loads along c, 16-byte stores along j;
// needs C*sizeof(T) % 16 == 0 and hwp % 64 == 0 — cv_hwp gives that — and 16-byte aligned bases)
template <typename T>
__global__ __launch_bounds__(256) void cv_transpose_kernel(const T* f1, const T* f2, T* o1, T* o2, int hw, int hwp, int C) {
    constexpr int VE = 16 / (int)sizeof(T), VPR = 64 / VE, LDR = 64 + VE;      // elements per vector, vectors per tile row, padded LDS row (16-byte aligned)
    typedef T vec_t __attribute__((ext_vector_type(VE)));
    __shared__ __attribute__((aligned(16))) T tile[64 * LDR];      // [c][j]
    const int p = blockIdx.z >> 1, which = blockIdx.z & 1;
    const T* in = (which ? f2 : f1) + (long)p * hw * C;
    T* out = (which ? o2 : o1) + (long)p * C * hwp;
    const int j0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
#pragma unroll
    for (int v = threadIdx.x; v < 64 * VPR; v += 256) {
        const int j = j0 + v / VPR, cc = c0 + (v % VPR) * VE;
        vec_t x;
#pragma unroll
        for (int e = 0; e < VE; ++e) x[e] = (T)0.f;
        if (j < hw && cc < C) x = *(const vec_t*)(in + (long)j * C + cc);      // (C % VE == 0: a vector is inside the row or outside it)
#pragma unroll
        for (int e = 0; e < VE; ++e) tile[((v % VPR) * VE + e) * LDR + v / VPR] = x[e];
    }
    __syncthreads();
#pragma unroll
    for (int v = threadIdx.x; v < 64 * VPR; v += 256) {
        const int cc = c0 + v / VPR, j = j0 + (v % VPR) * VE;
        if (cc < C && j < hwp) *(vec_t*)(out + (long)cc * hwp + j) = *(const vec_t*)(tile + (v / VPR) * LDR + (v % VPR) * VE);
    }
}

// da = inv*dah - a * inv^3 * (a . dah)   (gradient through x / max(||x||, 1e-12)); one wave per row, the row held in registers between the dot
// product and the output when C <= 1024 (one pass over memory)
template <typename T> __device__ __forceinline__ f32x4 cv_ldv4(const T* p);
template <> __device__ __forceinline__ f32x4 cv_ldv4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 cv_ldv4<bf16>(const bf16* p) {
    const bf16x4 v = *(const bf16x4*)p;
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T> __device__ __forceinline__ void cv_stv4(T* p, f32x4 v);
template <> __device__ __forceinline__ void cv_stv4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void cv_stv4<bf16>(bf16* p, f32x4 v) { *(bf16x4*)p = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]}; }

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
    constexpr int NV = 4;
    if (C % 4 == 0 && C <= 256 * NV) {
        f32x4 av[NV], dv[NV];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (lane + 64 * k) * 4;
            if (c < C) {
                av[k] = cv_ldv4<T>(a + c);
                dv[k] = *(const f32x4*)(dh + c);
                dot += av[k][0] * dv[k][0] + av[k][1] * dv[k][1] + av[k][2] * dv[k][2] + av[k][3] * dv[k][3];
            }
        }
        dot = wave_sum(dot);
        const float kk = inv * inv * inv * dot;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (lane + 64 * k) * 4;
            if (c < C) cv_stv4<T>(o + c, f32x4{inv * dv[k][0] - av[k][0] * kk, inv * dv[k][1] - av[k][1] * kk, inv * dv[k][2] - av[k][2] * kk,
                                              inv * dv[k][3] - av[k][3] * kk});
        }
        return;
    }
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
    if (!backward) return align256((size_t)P * cv_tiles(hw) * hw * 2 * sizeof(float)) + align256((size_t)P * cv_tiles(hw) * hw * 2 * sizeof(float)) +
                          align256((size_t)P * CV_FCH * sizeof(double)) + align256((size_t)P * 2 * hw * 4 * sizeof(float));
    const size_t hwp = (size_t)cv_hwp(hw);
    return 2 * align256((size_t)P * hw * hwp * es) + 2 * align256((size_t)P * C * hwp * es) +
           2 * align256((size_t)P * hw * C * sizeof(float));
}

extern "C" int gd_cost_volume_teacher_stats(const float* t1, const float* t2, int P, int hw, int ldt, float* tstats,
                                            void* stream) {
    GD_REQUIRE(P > 0 && hw > 0 && ldt >= hw, "gd_cost_volume_teacher_stats: bad shape P=%d hw=%d ldt=%d", P, hw, ldt);
    GD_REQUIRE(((uintptr_t)tstats & 15) == 0, "gd_cost_volume_teacher_stats: tstats must be 16-byte aligned");
    hipLaunchKernelGGL(cv_tstats_kernel, dim3(gd_cdiv(hw, 4), 2, P), dim3(256), 0, (hipStream_t)stream, t1, t2, tstats, hw, ldt);
    GD_LAUNCH_OK();
    return 0;
}

static int cv_fwd_common(const void* f1, const void* f2, const float* inv1, const float* inv2, const float* t1, const float* t2, int ldt,
                         const float* tstats, const unsigned char* m1, const unsigned char* m2, int P, int hw, int C, int variant,
                         int dtype, float* loss, float* stats, void* workspace, void* stream) {
    GD_REQUIRE(P > 0 && hw > 0 && C > 0 && ldt >= hw, "gd_cost_volume_kl_fwd: bad shape P=%d hw=%d C=%d ldt=%d", P, hw, C, ldt);
    GD_REQUIRE(variant == 0 || variant == 1, "gd_cost_volume_kl_fwd: variant must be 0 (vggt) or 1 (mast3r)");
    GD_REQUIRE(dtype == GD_F32 || dtype == GD_BF16 || dtype == GD_F16, "gd_cost_volume_kl_fwd: bad dtype %d", dtype);
    GD_REQUIRE(dtype != GD_F16 || inv1 != nullptr, "gd_cost_volume_kl_fwd: fp16 features (tf32h engine) come with the row norms of their f32 source (..._prenorm)");
    GD_REQUIRE((C * gd_dtype_size(dtype)) % 16 == 0, "gd_cost_volume_kl_fwd: C*elsize must be a multiple of 16 B");
    GD_REQUIRE((double)hw * 7.3890561 * CV_EPS < 1.0, "gd_cost_volume_kl_fwd: hw too large for the clamp-free softmax");
    GD_REQUIRE(((uintptr_t)f1 & 15) == 0 && ((uintptr_t)f2 & 15) == 0 && ((uintptr_t)stats & 15) == 0 &&
                   ((uintptr_t)workspace & 15) == 0 && ((uintptr_t)tstats & 15) == 0,
               "gd_cost_volume_kl_fwd: f1, f2, stats, tstats, workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int tiles = cv_tiles(hw), nslab = tiles;
    float* part1 = (float*)workspace;
    float* part2 = (float*)((char*)workspace + align256((size_t)P * nslab * hw * 2 * sizeof(float)));
    double* chunk_loss = (double*)((char*)part2 + align256((size_t)P * nslab * hw * 2 * sizeof(float)));
    int nslab2 = nslab;
    float* ts_ws = (float*)((char*)chunk_loss + align256((size_t)P * CV_FCH * sizeof(double)));
    if (!tstats) {     // no cached teacher statistics: one extra pass over the teacher maps
        hipLaunchKernelGGL(cv_tstats_kernel, dim3(gd_cdiv(hw, 4), 2, P), dim3(256), 0, s, t1, t2, ts_ws, hw, ldt);
        tstats = ts_ws;
    }
    if (inv1) {
        const long n = 2L * P * hw;
        hipLaunchKernelGGL(cv_stats_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, inv1, inv2, tstats, stats, hw, n);
    } else
        hipLaunchKernelGGL(cv_norm_kernel, dim3(gd_cdiv(hw, 4), 2, P), dim3(256), 0, s, f1, f2, tstats, stats, hw, C, dtype);
    GD_LAUNCH_OK();
    CvTileParams q = {};
    q.f1 = f1; q.f2 = f2; q.t1 = t1; q.t2 = t2; q.stats = stats; q.part1 = part1; q.part2 = part2;
    q.hw = hw; q.C = C; q.tiles = tiles; q.nslab = nslab; q.ldt = ldt; q.P = P;
    if (gd_knobs().cv_mask_skip) { q.m1 = m1; q.m2 = m2; }      // the persistent kernel skips the teacher entries of masked-out rows / columns
    // (A 256 x 256-tile forward on the persistent GEMM skeleton — half the feature bytes staged per FLOP, teacher entries streamed through registers
    // in a two-pass epilogue — was built and measured in round 3: 291 us against 263 us with every row kept, 221 against 193 with the trainer's masks;
    // shelved as tools/experiments/cv_persist256.h)
    const int persist = gd_knobs().cv_persist;     // GD_CV_PERSIST=0: the one-tile-per-block kernel (A/B)
    const long rowb = (long)C * gd_dtype_size(dtype);
    if (persist && rowb % 128 == 0 && rowb / 128 >= 3 && ldt % 4 == 0 && ((uintptr_t)t1 & 15) == 0 && ((uintptr_t)t2 & 15) == 0) {
        int ncu = 256;
        if (gd_knobs().ncu >= 8) ncu = gd_knobs().ncu / 8 * 8;
        const long total = (long)P * tiles * tiles;
        int grid = (int)(total < ncu ? (total + 7) / 8 * 8 : ncu);
        q.dbg = gd_knobs().cv_dbg;
        if (gd_knobs().cv_grid) {      // tests (gd_debug_set): few blocks, so that every block walks many tiles
            const int gv = gd_knobs().cv_grid / 8 * 8;
            if (gv >= 8 && gv < grid) grid = gv;
        }
        if (q.dbg && dtype == GD_BF16) hipLaunchKernelGGL((cv_fwd_persist_kernel<bf16, true>), dim3(grid), dim3(768), 0, s, q);
        else if (dtype == GD_BF16) hipLaunchKernelGGL((cv_fwd_persist_kernel<bf16, false>), dim3(grid), dim3(768), 0, s, q);
        else if (dtype == GD_F16) hipLaunchKernelGGL((cv_fwd_persist_kernel<f16, false>), dim3(grid), dim3(768), 0, s, q);
        else hipLaunchKernelGGL((cv_fwd_persist_kernel<float, false>), dim3(grid), dim3(768), 0, s, q);
    } else if (dtype == GD_F16)
        hipLaunchKernelGGL(cv_fwd_tile_kernel<f16>, dim3(tiles * tiles, P), dim3(256), 0, s, q);
    else if (dtype == GD_BF16)
        hipLaunchKernelGGL(cv_fwd_tile_kernel<bf16>, dim3(tiles * tiles, P), dim3(256), 0, s, q);
    else
        hipLaunchKernelGGL(cv_fwd_tile_kernel<float>, dim3(tiles * tiles, P), dim3(256), 0, s, q);
    GD_LAUNCH_OK();
    hipLaunchKernelGGL(cv_finalize_kernel, dim3(P, CV_FCH), dim3(256), 0, s, part1, part2, tstats, m1, m2, stats, chunk_loss, hw,
                       nslab, nslab2, variant);
    hipLaunchKernelGGL(cv_loss_kernel, dim3(gd_cdiv(P, 64)), dim3(64), 0, s, chunk_loss, loss, P, hw);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_cost_volume_kl_fwd(const void* f1, const void* f2, const float* t1, const float* t2, int ldt, const float* tstats,
                                     const unsigned char* m1, const unsigned char* m2, int P, int hw, int C, int variant, int dtype,
                                     float* loss, float* stats, void* workspace, void* stream) {
    return cv_fwd_common(f1, f2, nullptr, nullptr, t1, t2, ldt, tstats, m1, m2, P, hw, C, variant, dtype, loss, stats, workspace, stream);
}

extern "C" int gd_cost_volume_kl_fwd_prenorm(const void* f1, const void* f2, const float* inv_norm1, const float* inv_norm2, const float* t1,
                                             const float* t2, int ldt, const float* tstats, const unsigned char* m1,
                                             const unsigned char* m2, int P, int hw, int C, int variant, int dtype, float* loss,
                                             float* stats, void* workspace, void* stream) {
    GD_REQUIRE(inv_norm1 != nullptr && inv_norm2 != nullptr, "gd_cost_volume_kl_fwd_prenorm: inverse row norms missing");
    return cv_fwd_common(f1, f2, inv_norm1, inv_norm2, t1, t2, ldt, tstats, m1, m2, P, hw, C, variant, dtype, loss, stats, workspace, stream);
}

// the dense backward's tile pass: the persistent kernel's BWD instantiation where the forward's persistent kernel would run (rows of whole 128-byte
// K steps, 16-byte aligned teacher rows), else one tile per block
template <typename T>
static void cv_bwd_tiles_launch(CvTileParams& q, hipStream_t s) {
    const long rowb = (long)q.C * sizeof(T);
    if (gd_knobs().cv_persist && rowb % 128 == 0 && rowb / 128 >= 3 && q.ldt % 4 == 0 && ((uintptr_t)q.t1 & 15) == 0 && ((uintptr_t)q.t2 & 15) == 0 &&
        ((uintptr_t)q.stats & 15) == 0) {
        int ncu = 256;
        if (gd_knobs().ncu >= 8) ncu = gd_knobs().ncu / 8 * 8;
        const long total = (long)q.P * q.tiles * q.tiles;
        int grid = (int)(total < ncu ? (total + 7) / 8 * 8 : ncu);
        if (gd_knobs().cv_grid) {
            const int gv = gd_knobs().cv_grid / 8 * 8;
            if (gv >= 8 && gv < grid) grid = gv;
        }
        hipLaunchKernelGGL((cv_fwd_persist_kernel<T, false, true>), dim3(grid), dim3(768), 0, s, q);
    } else
        hipLaunchKernelGGL(cv_bwd_tile_kernel<T>, dim3(q.tiles * q.tiles, q.P), dim3(256), 0, s, q);
}

extern "C" int gd_cost_volume_kl_bwd(const void* f1, const void* f2, const float* t1, const float* t2, int ldt,
                                     const unsigned char* m1, const unsigned char* m2, int P, int hw, int C,
                                     int dtype, const float* gloss, const float* stats, void* df1, void* df2,
                                     void* workspace, void* stream) {
    GD_REQUIRE(P > 0 && hw > 0 && C > 0 && ldt >= hw, "gd_cost_volume_kl_bwd: bad shape P=%d hw=%d C=%d ldt=%d", P, hw, C, ldt);
    GD_REQUIRE(dtype == GD_F32 || dtype == GD_BF16, "gd_cost_volume_kl_bwd: bad dtype %d", dtype);
    GD_REQUIRE((C * gd_dtype_size(dtype)) % 16 == 0 && C % 4 == 0, "gd_cost_volume_kl_bwd: C*elsize must be a multiple of 16 B");
    GD_REQUIRE(((uintptr_t)f1 & 15) == 0 && ((uintptr_t)f2 & 15) == 0 && ((uintptr_t)df1 & 15) == 0 && ((uintptr_t)df2 & 15) == 0 &&
                   ((uintptr_t)workspace & 255) == 0,
               "gd_cost_volume_kl_bwd: features and gradients must be 16-byte aligned (vector transpose / normalisation backward), the workspace 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const size_t es = (size_t)gd_dtype_size(dtype);
    const int tiles = cv_tiles(hw), hwp = cv_hwp(hw);
    // G1 | G2, b^T | a^T and d a_hat | d b_hat are each ONE contiguous batch of 2P problems: the two contractions run as one launch (1 152 tiles = 4.5
    // rounds of the chip at the benched size instead of 2 x 2.25)
    char* w = (char*)workspace;
    void* G1 = w; void* G2 = w + (size_t)P * hw * hwp * es; w += align256((size_t)2 * P * hw * hwp * es);
    void* bt = w; void* at = w + (size_t)P * C * hwp * es; w += align256((size_t)2 * P * C * hwp * es);
    float* da = (float*)w; float* db = da + (size_t)P * hw * C;
    CvTileParams q = {};
    q.f1 = f1; q.f2 = f2; q.t1 = t1; q.t2 = t2; q.stats = (float*)stats; q.hw = hw; q.C = C; q.tiles = tiles;
    q.m1 = m1; q.m2 = m2; q.gloss = gloss; q.G1 = G1; q.G2 = G2; q.hwp = hwp; q.ldt = ldt; q.P = P;
    dim3 tgrid(hwp / 64, gd_cdiv(C, 64), 2 * P);
    if (dtype == GD_BF16) {
        cv_bwd_tiles_launch<bf16>(q, s);
        hipLaunchKernelGGL(cv_transpose_kernel<bf16>, tgrid, dim3(256), 0, s, (const bf16*)f1, (const bf16*)f2,
                           (bf16*)at, (bf16*)bt, hw, hwp, C);
    } else {
        cv_bwd_tiles_launch<float>(q, s);
        hipLaunchKernelGGL(cv_transpose_kernel<float>, tgrid, dim3(256), 0, s, (const float*)f1, (const float*)f2,
                           (float*)at, (float*)bt, hw, hwp, C);
    }
    GD_LAUNCH_OK();
    // d a_hat = G1 . b ,  d b_hat = G2 . a   (contraction over the padded hw axis)
    int rc = gd_gemm_nt(G1, bt, da, hw, C, hwp, hwp, hwp, C, 2 * P, (long)hw * hwp, (long)C * hwp, (long)hw * C, dtype,
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

// ---- backward of the tf32h engine: S is recomputed from the fp16 copies of the features the forward ran on, G = dloss/dS leaves as fp16 under a
// power-of-two scale taken from the loss gradients on the device (|G| ~ 1e-9 otherwise: far below fp16's range), the two G contractions run on the
// fp16 MFMA kernels and undo the scale in their epilogue; the gradient through the L2 normalisation stays fp32 on the fp32 features.
__global__ void cv_gscale_kernel(const float* gloss, int P, int hw, float* out2) {
    float m = 0.f;
    for (int p = threadIdx.x; p < P; p += 64) m = fmaxf(m, fabsf(gloss[p]));
    m = wave_max(m) * 0.5f / (float)hw;
    if (threadIdx.x == 0) {
        float s = 1.0f;
        if (m > 0.f && m < 3.0e38f) s = exp2f(floorf(log2f(4096.0f / m)));      // coef * s <= 4096: G = coef s (p - t) inv sits in fp16's normal range
        s = fminf(fmaxf(s, 1.0f), 7.9e28f);
        out2[0] = s;
        out2[1] = 1.0f / s;
    }
}
extern "C" int gd_gemm_nt_scaled(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc,
                                 int batch, long sA, long sW, long sC, int ab_dtype, int c_dtype, float alpha, const float* alpha_dev,
                                 const float* bias, const float* lora_t, const float* lora_b, int lora_rt, void* preact,
                                 long ldp, int act, const void* dact_src, long ldd, int dact, const void* residual, long ldr,
                                 int accumulate, void* stream);

extern "C" size_t gd_cost_volume_kl_bwd_h_workspace_bytes(int P, int hw, int C) {
    const size_t hwp = (size_t)cv_hwp(hw);
    return 2 * align256((size_t)P * hw * hwp * 2) + 2 * align256((size_t)P * C * hwp * 2) + 2 * align256((size_t)P * hw * C * sizeof(float)) + 256;
}

extern "C" int gd_cost_volume_kl_bwd_h(const float* f1, const float* f2, const void* f1h, const void* f2h, const float* t1, const float* t2, int ldt,
                                       const unsigned char* m1, const unsigned char* m2, int P, int hw, int C, const float* gloss,
                                       const float* stats, float* df1, float* df2, void* workspace, void* stream) {
    GD_REQUIRE(P > 0 && hw > 0 && C > 0 && ldt >= hw, "gd_cost_volume_kl_bwd_h: bad shape P=%d hw=%d C=%d ldt=%d", P, hw, C, ldt);
    GD_REQUIRE(C % 8 == 0, "gd_cost_volume_kl_bwd_h: C must be a multiple of 8");
    GD_REQUIRE(((uintptr_t)f1 & 15) == 0 && ((uintptr_t)f2 & 15) == 0 && ((uintptr_t)f1h & 15) == 0 && ((uintptr_t)f2h & 15) == 0 &&
                   ((uintptr_t)df1 & 15) == 0 && ((uintptr_t)df2 & 15) == 0 && ((uintptr_t)workspace & 255) == 0,
               "gd_cost_volume_kl_bwd_h: features and gradients must be 16-byte aligned, the workspace 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int tiles = cv_tiles(hw), hwp = cv_hwp(hw);
    char* w = (char*)workspace;      // (contiguous batches of 2P, as in gd_cost_volume_kl_bwd)
    void* G1 = w; void* G2 = w + (size_t)P * hw * hwp * 2; w += align256((size_t)2 * P * hw * hwp * 2);
    void* bt = w; void* at = w + (size_t)P * C * hwp * 2; w += align256((size_t)2 * P * C * hwp * 2);
    float* da = (float*)w; float* db = da + (size_t)P * hw * C; w += align256((size_t)2 * P * hw * C * sizeof(float));
    float* gs = (float*)w;
    hipLaunchKernelGGL(cv_gscale_kernel, dim3(1), dim3(64), 0, s, gloss, P, hw, gs);
    CvTileParams q = {};
    q.f1 = f1h; q.f2 = f2h; q.t1 = t1; q.t2 = t2; q.stats = (float*)stats; q.hw = hw; q.C = C; q.tiles = tiles;
    q.m1 = m1; q.m2 = m2; q.gloss = gloss; q.G1 = G1; q.G2 = G2; q.hwp = hwp; q.ldt = ldt; q.P = P; q.gscale = gs;
    dim3 tgrid(hwp / 64, gd_cdiv(C, 64), 2 * P);
    cv_bwd_tiles_launch<f16>(q, s);
    hipLaunchKernelGGL(cv_transpose_kernel<f16>, tgrid, dim3(256), 0, s, (const f16*)f1h, (const f16*)f2h, (f16*)at, (f16*)bt, hw, hwp, C);
    GD_LAUNCH_OK();
    int rc = gd_gemm_nt_scaled(G1, bt, da, hw, C, hwp, hwp, hwp, C, 2 * P, (long)hw * hwp, (long)C * hwp, (long)hw * C, GD_F16, GD_F32, 1.0f, gs + 1,
                               nullptr, nullptr, nullptr, 0, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 0, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(cv_norm_bwd_kernel<float>, dim3(gd_cdiv(hw, 4), 2, P), dim3(256), 0, s, f1, f2, da, db, stats, df1, df2, hw, C);
    GD_LAUNCH_OK();
    return 0;
}

// ---- kept-row forward (sparse row masks).  kcap: a multiple of 128 >= the number of kept rows of any (pair, view) — the MASt3R trainer's masks keep at
// most N_kp patches (src/finetune_timm_mast3r.py:515-519).  Needs the rows' inverse norms (the ..._prenorm contract) and the cached teacher statistics.
extern "C" size_t gd_cost_volume_kl_rows_workspace_bytes(int P, int hw, int kcap) {
    return align256((size_t)P * 2 * cv_tiles(hw) * kcap * 2 * sizeof(float)) + align256((size_t)P * CV_FCH * sizeof(double)) +
           align256((size_t)P * 2 * kcap * sizeof(int)) + align256((size_t)P * 2 * sizeof(int));
}
extern "C" int gd_cost_volume_kl_fwd_rows(const void* f1, const void* f2, const float* inv_norm1, const float* inv_norm2, const float* t1, const float* t2,
                                          int ldt, const float* tstats, const unsigned char* m1, const unsigned char* m2, int P, int hw, int C, int kcap,
                                          int variant, int dtype, float* loss, float* stats, void* workspace, void* stream) {
    GD_REQUIRE(P > 0 && hw > 0 && C > 0 && ldt >= hw && kcap > 0 && kcap % 128 == 0, "gd_cost_volume_kl_fwd_rows: bad shape P=%d hw=%d C=%d ldt=%d kcap=%d", P, hw, C, ldt, kcap);
    GD_REQUIRE(variant == 0 || variant == 1, "gd_cost_volume_kl_fwd_rows: variant must be 0 (vggt) or 1 (mast3r)");
    GD_REQUIRE(dtype == GD_F32 || dtype == GD_BF16 || dtype == GD_F16, "gd_cost_volume_kl_fwd_rows: bad dtype %d", dtype);
    GD_REQUIRE(inv_norm1 && inv_norm2 && tstats && m1 && m2, "gd_cost_volume_kl_fwd_rows: inverse row norms, teacher statistics and both masks are required");
    GD_REQUIRE((double)hw * 7.3890561 * CV_EPS < 1.0, "gd_cost_volume_kl_fwd_rows: hw too large for the clamp-free softmax");
    const long rowb = (long)C * gd_dtype_size(dtype);
    GD_REQUIRE(rowb % 128 == 0 && rowb / 128 >= 3 && ldt % 4 == 0 && ((uintptr_t)t1 & 15) == 0 && ((uintptr_t)t2 & 15) == 0 &&
                   ((uintptr_t)f1 & 15) == 0 && ((uintptr_t)f2 & 15) == 0 && ((uintptr_t)stats & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
               "gd_cost_volume_kl_fwd_rows: rows of C*elsize %% 128 == 0 bytes (>= 384), teacher rows padded to 16 bytes, 16-byte aligned pointers");
    hipStream_t s = (hipStream_t)stream;
    const int tiles = cv_tiles(hw);
    char* w = (char*)workspace;
    float* part = (float*)w; w += align256((size_t)P * 2 * tiles * kcap * 2 * sizeof(float));
    double* chunk_loss = (double*)w; w += align256((size_t)P * CV_FCH * sizeof(double));
    int* idx = (int*)w; w += align256((size_t)P * 2 * kcap * sizeof(int));
    int* cnt = (int*)w;
    const long n = 2L * P * hw;
    hipLaunchKernelGGL(cv_stats_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, inv_norm1, inv_norm2, tstats, stats, hw, n);
    hipLaunchKernelGGL(cv_rows_compact_kernel, dim3(2 * P), dim3(256), 0, s, m1, m2, idx, cnt, hw, kcap);
    CvTileParams q = {};
    q.f1 = f1; q.f2 = f2; q.t1 = t1; q.t2 = t2; q.stats = stats; q.part1 = part; q.part2 = part;
    q.hw = hw; q.C = C; q.tiles = tiles; q.nslab = tiles; q.ldt = ldt; q.P = P; q.idx = idx; q.kcap = kcap;
    int ncu = 256;
    if (gd_knobs().ncu >= 8) ncu = gd_knobs().ncu / 8 * 8;
    const long total = (long)P * 2 * (kcap / 128) * tiles;
    int grid = (int)(total < ncu ? (total + 7) / 8 * 8 : ncu);
    if (gd_knobs().cv_grid) {
        const int gv = gd_knobs().cv_grid / 8 * 8;
        if (gv >= 8 && gv < grid) grid = gv;
    }
    if (dtype == GD_BF16) hipLaunchKernelGGL(cv_fwd_rows_kernel<bf16>, dim3(grid), dim3(768), 0, s, q);
    else if (dtype == GD_F16) hipLaunchKernelGGL(cv_fwd_rows_kernel<f16>, dim3(grid), dim3(768), 0, s, q);
    else hipLaunchKernelGGL(cv_fwd_rows_kernel<float>, dim3(grid), dim3(768), 0, s, q);
    GD_LAUNCH_OK();
    hipLaunchKernelGGL(cv_finalize_rows_kernel, dim3(P, CV_FCH), dim3(256), 0, s, part, tstats, idx, cnt, stats, chunk_loss, hw, tiles, kcap, variant);
    hipLaunchKernelGGL(cv_loss_kernel, dim3(gd_cdiv(P, 64)), dim3(64), 0, s, chunk_loss, loss, P, hw);
    GD_LAUNCH_OK();
    return 0;
}

// ---- kept-row backward.  dtype GD_F32 / GD_BF16: f1, f2 (and df1, df2) in that type, f1h = f2h = null.  dtype GD_F16 (tf32h engine): f1, f2 fp32,
// f1h, f2h their fp16 copies (S, G and the four contractions run on those, G under the device-side scale of gd_cost_volume_kl_bwd_h), df1, df2 fp32.
// `stats` as saved by either forward (logZ and W sit at the rows' original indices).
extern "C" size_t gd_cost_volume_kl_bwd_rows_workspace_bytes(int P, int hw, int C, int kcap, int dtype) {
    const size_t es = (size_t)gd_dtype_size(dtype), hwp = (size_t)cv_hwp(hw);
    return align256((size_t)2 * P * kcap * hwp * es) + align256((size_t)2 * P * hw * kcap * es) + align256((size_t)2 * P * C * hwp * es) +
           2 * align256((size_t)2 * P * kcap * C * es) + align256((size_t)2 * P * kcap * C * sizeof(float)) +
           align256((size_t)2 * P * hw * C * sizeof(float)) + align256((size_t)P * 2 * kcap * sizeof(int)) + align256((size_t)P * 2 * sizeof(int)) + 256;
}

template <typename T>
static void cv_bwd_rows_launch(const CvTileParams& q, const void* fa, const void* fb, void* fc, void* fct, void* at, void* bt, hipStream_t s) {
    const int P = q.P, hw = q.hw, C = q.C, kcap = q.kcap, hwp = q.hwp, rowb = C * (int)sizeof(T);
    const long ng = 2L * P * kcap * (rowb / 16);
    hipLaunchKernelGGL(cv_rows_gather_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, s, (const char*)fa, (const char*)fb, q.idx, (char*)fc,
                       P, hw, kcap, rowb);
    hipLaunchKernelGGL(cv_bwd_rows_kernel<T>, dim3(kcap / 128 * q.tiles, 2 * P), dim3(256), 0, s, q);
    hipLaunchKernelGGL(cv_transpose_kernel<T>, dim3(hwp / 64, gd_cdiv(C, 64), 2 * P), dim3(256), 0, s, (const T*)fa, (const T*)fb, (T*)at, (T*)bt,
                       hw, hwp, C);
    hipLaunchKernelGGL(cv_transpose_kernel<T>, dim3(kcap / 64, gd_cdiv(C, 64), 2 * P), dim3(256), 0, s, (const T*)fc, (const T*)fc + (long)P * kcap * C,
                       (T*)fct, (T*)fct + (long)P * C * kcap, kcap, kcap, C);
}

extern "C" int gd_cost_volume_kl_bwd_rows(const void* f1, const void* f2, const void* f1h, const void* f2h, const float* t1, const float* t2, int ldt,
                                          const unsigned char* m1, const unsigned char* m2, int P, int hw, int C, int kcap, int dtype,
                                          const float* gloss, const float* stats, void* df1, void* df2, void* workspace, void* stream) {
    GD_REQUIRE(P > 0 && hw > 0 && C > 0 && ldt >= hw && kcap > 0 && kcap % 128 == 0, "gd_cost_volume_kl_bwd_rows: bad shape P=%d hw=%d C=%d ldt=%d kcap=%d", P, hw, C, ldt, kcap);
    GD_REQUIRE(dtype == GD_F32 || dtype == GD_BF16 || dtype == GD_F16, "gd_cost_volume_kl_bwd_rows: bad dtype %d", dtype);
    GD_REQUIRE(m1 && m2, "gd_cost_volume_kl_bwd_rows: both masks are required");
    GD_REQUIRE((dtype == GD_F16) == (f1h != nullptr && f2h != nullptr), "gd_cost_volume_kl_bwd_rows: fp16 feature copies go with dtype GD_F16 and only with it");
    const size_t es = (size_t)gd_dtype_size(dtype);
    GD_REQUIRE((C * es) % 16 == 0 && C % 4 == 0, "gd_cost_volume_kl_bwd_rows: C*elsize must be a multiple of 16 B");
    GD_REQUIRE(((uintptr_t)f1 & 15) == 0 && ((uintptr_t)f2 & 15) == 0 && ((uintptr_t)f1h & 15) == 0 && ((uintptr_t)f2h & 15) == 0 &&
                   ((uintptr_t)df1 & 15) == 0 && ((uintptr_t)df2 & 15) == 0 && ((uintptr_t)stats & 15) == 0 && ((uintptr_t)workspace & 255) == 0,
               "gd_cost_volume_kl_bwd_rows: features, gradients and stats must be 16-byte aligned, the workspace 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int tiles = cv_tiles(hw), hwp = cv_hwp(hw);
    // every [direction][pair] array is one contiguous batch of 2P problems: direction 0 = kept rows of view 1 against view 2, direction 1 the reverse
    char* w = (char*)workspace;
    void* Gc = w; w += align256((size_t)2 * P * kcap * hwp * es);
    void* GcT = w; w += align256((size_t)2 * P * hw * kcap * es);
    void* tb = w; w += align256((size_t)2 * P * C * hwp * es);      // transposed features of the OTHER view: [0] = view 2, [1] = view 1
    void* fc = w; w += align256((size_t)2 * P * kcap * C * es);
    void* fct = w; w += align256((size_t)2 * P * kcap * C * es);
    float* dk = (float*)w; w += align256((size_t)2 * P * kcap * C * sizeof(float));
    float* dd = (float*)w; w += align256((size_t)2 * P * hw * C * sizeof(float));      // gradients of the OTHER view's normalised rows: [0] = view 2, [1] = view 1
    int* idx = (int*)w; w += align256((size_t)P * 2 * kcap * sizeof(int));
    int* cnt = (int*)w; w += align256((size_t)P * 2 * sizeof(int));
    float* gs = (float*)w;
    float* db = dd;
    float* da = dd + (long)P * hw * C;
    void* bt = tb;
    void* at = (char*)tb + (size_t)P * C * hwp * es;
    const void* fa = dtype == GD_F16 ? f1h : f1;
    const void* fb = dtype == GD_F16 ? f2h : f2;
    hipLaunchKernelGGL(cv_rows_compact_kernel, dim3(2 * P), dim3(256), 0, s, m1, m2, idx, cnt, hw, kcap);
    if (dtype == GD_F16) hipLaunchKernelGGL(cv_gscale_kernel, dim3(1), dim3(64), 0, s, gloss, P, hw, gs);
    CvTileParams q = {};
    q.f1 = fa; q.f2 = fb; q.fc = fc; q.t1 = t1; q.t2 = t2; q.stats = (float*)stats; q.hw = hw; q.C = C; q.tiles = tiles;
    q.gloss = gloss; q.G1 = Gc; q.G2 = GcT; q.hwp = hwp; q.ldt = ldt; q.P = P; q.idx = idx; q.cnt = cnt; q.kcap = kcap;
    q.gscale = dtype == GD_F16 ? gs : nullptr;
    q.dbg = gd_knobs().cv_dbg;      // anatomy switches (timing experiments only: results are wrong with any bit set)
    if (dtype == GD_BF16) cv_bwd_rows_launch<bf16>(q, fa, fb, fc, fct, at, bt, s);
    else if (dtype == GD_F16) cv_bwd_rows_launch<f16>(q, fa, fb, fc, fct, at, bt, s);
    else cv_bwd_rows_launch<float>(q, fa, fb, fc, fct, at, bt, s);
    GD_LAUNCH_OK();
    const float* adev = dtype == GD_F16 ? gs + 1 : nullptr;
    // the other view's gradient, dense: d b_hat = GcT[0] a_kept ,  d a_hat = GcT[1] b_kept   (contraction over the kcap kept rows)
    int rc = gd_gemm_nt_scaled(GcT, fct, dd, hw, C, kcap, kcap, kcap, C, 2 * P, (long)hw * kcap, (long)C * kcap, (long)hw * C, dtype, GD_F32, 1.0f, adev,
                               nullptr, nullptr, nullptr, 0, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 0, stream);
    if (rc) return rc;
    // this view's kept rows: d a_hat[kept] = Gc[0] b ,  d b_hat[kept] = Gc[1] a   (contraction over the padded hw axis)
    rc = gd_gemm_nt_scaled(Gc, tb, dk, kcap, C, hwp, hwp, hwp, C, 2 * P, (long)kcap * hwp, (long)C * hwp, (long)kcap * C, dtype, GD_F32, 1.0f, adev,
                           nullptr, nullptr, nullptr, 0, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 0, stream);
    if (rc) return rc;
    const long nsc = 2L * P * kcap * (C / 4);
    hipLaunchKernelGGL(cv_rows_scatter_kernel, dim3((unsigned)((nsc + 255) / 256)), dim3(256), 0, s, dk, idx, cnt, da, db, P, hw, kcap, C);
    if (dtype == GD_BF16)
        hipLaunchKernelGGL(cv_norm_bwd_kernel<bf16>, dim3(gd_cdiv(hw, 4), 2, P), dim3(256), 0, s, (const bf16*)f1, (const bf16*)f2, da, db, stats,
                           (bf16*)df1, (bf16*)df2, hw, C);
    else
        hipLaunchKernelGGL(cv_norm_bwd_kernel<float>, dim3(gd_cdiv(hw, 4), 2, P), dim3(256), 0, s, (const float*)f1, (const float*)f2, da, db, stats,
                           (float*)df1, (float*)df2, hw, C);
    GD_LAUNCH_OK();
    return 0;
}
