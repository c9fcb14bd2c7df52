// Sparse distillation losses on keypoint features (gfx950): smooth-AP correspondence loss (a8/a9), the
// relative-depth head (a12) with its L1 term and the pairwise logistic ranking loss (a10/a11).
// All fp32: with temperature 0.01 the smooth-AP sigmoids are numerically fragile, and N is small.
#include "gd_common.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------------
// smooth-AP (src/finetune_timm_vggt.py:543-572 variant 0, src/finetune_timm_mast3r.py:560-589 variant 1;
// sigmoid = utils/functions.py:24-33 with temp 0.01 and exponent clamp +-50).
// sim [P,Nmax,Nmax] = desc1 desc2^T (unit descriptors, from gd_gemm_nt); one block per (pair, row i).
// Emits the per-row loss and dsim = d(sum_i loss_i / n_p) / d sim, fused (the step always needs both).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sig_t(float x, float inv_temp, float& dsig) {
    float e = -x * inv_temp;
    const bool in = e >= -50.f && e <= 50.f;
    e = fminf(fmaxf(e, -50.f), 50.f);
    const float y = 1.0f / (1.0f + expf(e));
    dsig = in ? y * (1.0f - y) * inv_temp : 0.f;
    return y;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void smooth_ap_kernel(const float* sim, const float* pts1, const float* pts2,
                                                        const int* counts, float* row_loss, float* dsim, int Nmax,
                                                        int variant, float thr, float inv_temp) {
    __shared__ float red[4];
    const int p = blockIdx.y, i = blockIdx.x, n = counts ? counts[p] : Nmax;
    const float* srow = sim + ((long)p * Nmax + i) * Nmax;
    float* drow = dsim + ((long)p * Nmax + i) * Nmax;
    if (i >= n) {
        for (int j = threadIdx.x; j < Nmax; j += 256) drow[j] = 0.f;
        if (threadIdx.x == 0) row_loss[(long)p * Nmax + i] = 0.f;
        return;
    }
    const float pos = srow[i];
    const float ax = pts1[((long)p * Nmax + i) * 3 + 0], ay = pts1[((long)p * Nmax + i) * 3 + 1],
                az = pts1[((long)p * Nmax + i) * 3 + 2];
    float A1 = 0.f, A2 = 0.f, D2 = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) {
        const float* q = pts2 + ((long)p * Nmax + j) * 3;
        const float dx = ax - q[0], dy = ay - q[1], dz = az - q[2];
        const bool neg = (j != i) && (sqrtf(dx * dx + dy * dy + dz * dz) > thr);
        if (neg) {
            float d1, d2;
            A1 += sig_t(srow[j] - 1.0f, inv_temp, d1);
            A2 += sig_t(srow[j] - pos, inv_temp, d2);
            D2 += d2;
        }
    }
    A1 = block_sum(A1, red);
    A2 = block_sum(A2, red);
    D2 = block_sum(D2, red);
    float dr1, dr2;
    const float rpos1 = (variant == 0 ? sig_t(1.0f - pos, inv_temp, dr1) : sig_t(pos - 1.0f, inv_temp, dr1)) + 1.0f;
    const float drpos1 = variant == 0 ? -dr1 : dr1;   // d rpos1 / d pos
    const float rpos2 = sig_t(1.0f - pos, inv_temp, dr2) + 1.0f;
    const float drpos2 = -dr2;
    const float den1 = rpos1 + A1, den2 = rpos2 + A2;
    const float ap1 = rpos1 / den1, ap2 = rpos2 / den2;
    const float dap1_dA = -rpos1 / (den1 * den1), dap2_dA = -rpos2 / (den2 * den2);
    const float dap1_dr = A1 / (den1 * den1), dap2_dr = A2 / (den2 * den2);
    const float w = -0.5f / (float)n;   // loss_i = 1 - (ap1+ap2)/2, mean over n rows
    for (int j = threadIdx.x; j < Nmax; j += 256) {
        float gj = 0.f;
        if (j < n && j != i) {
            const float* q = pts2 + ((long)p * Nmax + j) * 3;
            const float dx = ax - q[0], dy = ay - q[1], dz = az - q[2];
            if (sqrtf(dx * dx + dy * dy + dz * dz) > thr) {
                float d1, d2;
                sig_t(srow[j] - 1.0f, inv_temp, d1);
                sig_t(srow[j] - pos, inv_temp, d2);
                gj = w * (dap1_dA * d1 + dap2_dA * d2);
            }
        } else if (j == i) {
            gj = w * (dap1_dr * drpos1 + dap2_dr * drpos2 - dap2_dA * D2);
        }
        drow[j] = gj;
    }
    if (threadIdx.x == 0) row_loss[(long)p * Nmax + i] = (1.0f - 0.5f * (ap1 + ap2)) / (float)n;
}

// ---- "ME" variant of the matching loss (src/finetune_timm_me.py:191-220): the positives are EVERY (i, j) whose 3-D points
// are closer than thres3d_pos (a dynamic count, none or several per row), the negatives those farther than thres3d_neg;
//   per positive (i, jp):  ap1 = r1 / (r1 + sum_neg sig(s_ij - 1)),  r1 = sig(s_pos - 1) + 1
//                          ap2 = r2 / (r2 + sum_neg sig(s_ij - s_pos)),  r2 = sig(1 - s_pos) + 1;   loss = mean_pos (1 - (ap1+ap2)/2)
// One block per row i walks the row's positives in index order (deterministic); the row of dsim is accumulated in place,
// UN-normalised (the number of positives of the pair is only known once every row is done): smooth_ap_me_finalize
// divides.  row_ws: [P][Nmax][2] = {sum over the row's positives of (1 - ap), number of positives}.
#define AP_ME_NMAX 4096
__global__ __launch_bounds__(256) void smooth_ap_me_kernel(const float* sim, const float* pts1, const float* pts2, const int* counts,
                                                           float* row_ws, float* dsim, int Nmax, float thr_pos, float thr_neg,
                                                           float inv_temp) {
    __shared__ float red[4];
    __shared__ unsigned char flag[AP_ME_NMAX];   // bit 0: negative, bit 1: positive
    const int p = blockIdx.y, i = blockIdx.x, n = counts ? counts[p] : Nmax;
    const float* srow = sim + ((long)p * Nmax + i) * Nmax;
    float* drow = dsim + ((long)p * Nmax + i) * Nmax;
    for (int j = threadIdx.x; j < Nmax; j += 256) drow[j] = 0.f;
    if (i >= n) {
        if (threadIdx.x == 0) { row_ws[((long)p * Nmax + i) * 2] = 0.f; row_ws[((long)p * Nmax + i) * 2 + 1] = 0.f; }
        return;
    }
    const float ax = pts1[((long)p * Nmax + i) * 3 + 0], ay = pts1[((long)p * Nmax + i) * 3 + 1], az = pts1[((long)p * Nmax + i) * 3 + 2];
    float A1 = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) {
        const float* q = pts2 + ((long)p * Nmax + j) * 3;
        const float dx = ax - q[0], dy = ay - q[1], dz = az - q[2];
        const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
        const unsigned char f = (dist > thr_neg ? 1 : 0) | (dist < thr_pos ? 2 : 0);
        flag[j] = f;
        if (f & 1) { float d1; A1 += sig_t(srow[j] - 1.0f, inv_temp, d1); }
    }
    A1 = block_sum(A1, red);          // (its barriers also publish flag[])
    float lsum = 0.f;
    int npos = 0;
    for (int jp = 0; jp < n; ++jp) {
        if (!(flag[jp] & 2)) continue;                 // block-uniform
        const float pos = srow[jp];
        float A2 = 0.f, D2 = 0.f;
        for (int j = threadIdx.x; j < n; j += 256)
            if (flag[j] & 1) { float d2; A2 += sig_t(srow[j] - pos, inv_temp, d2); D2 += d2; }
        A2 = block_sum(A2, red);
        D2 = block_sum(D2, red);
        float dr1, dr2;
        const float r1 = sig_t(pos - 1.0f, inv_temp, dr1) + 1.0f, r2 = sig_t(1.0f - pos, inv_temp, dr2) + 1.0f;
        const float den1 = r1 + A1, den2 = r2 + A2;
        const float ap1 = r1 / den1, ap2 = r2 / den2;
        const float dap1_dA = -r1 / (den1 * den1), dap2_dA = -r2 / (den2 * den2);
        const float dap1_dr = A1 / (den1 * den1), dap2_dr = A2 / (den2 * den2);
        lsum += 1.0f - 0.5f * (ap1 + ap2);
        ++npos;
        for (int j = threadIdx.x; j < n; j += 256) {
            float gj = 0.f;
            if (flag[j] & 1) {
                float d1, d2;
                sig_t(srow[j] - 1.0f, inv_temp, d1);
                sig_t(srow[j] - pos, inv_temp, d2);
                gj = -0.5f * (dap1_dA * d1 + dap2_dA * d2);
            }
            if (j == jp) gj += -0.5f * (dap1_dr * dr1 - dap2_dr * dr2 - dap2_dA * D2);
            if (gj != 0.f) drow[j] += gj;               // this block owns the row
        }
    }
    if (threadIdx.x == 0) { row_ws[((long)p * Nmax + i) * 2] = lsum; row_ws[((long)p * Nmax + i) * 2 + 1] = (float)npos; }
}

// loss[p] = sum_rows / M_p, dsim[p] /= M_p   (M_p = positives of the pair; 0 -> loss 0, zero gradient)
__global__ __launch_bounds__(256) void smooth_ap_me_finalize_kernel(const float* row_ws, float* loss, float* dsim, int Nmax) {
    __shared__ float red[4];
    const int p = blockIdx.y;
    float ls = 0.f, cnt = 0.f;
    for (int i = threadIdx.x; i < Nmax; i += 256) { ls += row_ws[((long)p * Nmax + i) * 2]; cnt += row_ws[((long)p * Nmax + i) * 2 + 1]; }
    ls = block_sum(ls, red);
    cnt = block_sum(cnt, red);
    const float inv = cnt > 0.f ? 1.0f / cnt : 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) loss[p] = ls * inv;
    const long total = (long)Nmax * Nmax;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) dsim[(long)p * total + idx] *= inv;
}

// out[p] = sum_i rows[p, i]   (deterministic)
__global__ __launch_bounds__(256) void row_sum_kernel(const float* rows, float* out, int n) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += rows[(long)blockIdx.x * n + i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------------
// Depth head on pre-projected features: DepthAwareFeatureFusion.fusion_layer (utils/model.py:100-127) is
// tanh(w2 . GELU(LN_128(W1 x + b1)) + b2).  Every argument the losses feed it is a DIFFERENCE of keypoint
// features, and W1 (f_j - f_i) = u_j - u_i with u = W1 f, so the D->128 projection is done once per
// keypoint by gd_gemm_nt and the kernels below work on 128-vectors (2 per lane, one wave per evaluation).
// Head-parameter gradients are accumulated un-normalised in hg[set][HG_SIZE] with fp32 atomics:
//   [0:128) b1, [128:256) ln_w, [256:384) ln_b, [384:512) w2, [512] b2.
// ---------------------------------------------------------------------------------------------------
#define HG_SIZE 516
struct HeadW { float b1[2], lw[2], lb[2], w2[2], b2; };

__device__ __forceinline__ HeadW load_head(const float* b1, const float* lw, const float* lb, const float* w2,
                                           const float* b2, int lane) {
    HeadW h;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        h.b1[e] = b1[lane + 64 * e]; h.lw[e] = lw[lane + 64 * e]; h.lb[e] = lb[lane + 64 * e]; h.w2[e] = w2[lane + 64 * e];
    }
    h.b2 = b2[0];
    return h;
}

struct HeadCache { float yh[2], y[2], a[2], dg[2], rstd, s; };   // a = GELU(y), dg = GELU'(y) from one shared exponential

__device__ __forceinline__ float head_eval(const float (&z)[2], const HeadW& h, HeadCache& c) {
    const float mu = wave_sum(z[0] + z[1]) * (1.0f / 128.0f);
    const float d0 = z[0] - mu, d1 = z[1] - mu;
    const float var = wave_sum(d0 * d0 + d1 * d1) * (1.0f / 128.0f);
    c.rstd = rsqrtf(var + 1e-5f);
    c.yh[0] = d0 * c.rstd; c.yh[1] = d1 * c.rstd;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        c.y[e] = c.yh[e] * h.lw[e] + h.lb[e];
        float Phi, ex;
        gelu_parts(c.y[e], Phi, ex);
        c.a[e] = c.y[e] * Phi;
        c.dg[e] = Phi + c.y[e] * ex * 0.39894228040143268f;
    }
    const float o = wave_sum(c.a[0] * h.w2[0] + c.a[1] * h.w2[1]) + h.b2;
    c.s = tanhf(o);
    return c.s;
}

// given dL/ds: returns dz (2 per lane); when acc != null adds this evaluation's head-parameter grads
__device__ __forceinline__ void head_back(float dLds, const HeadW& h, const HeadCache& c, float (&dz)[2],
                                          float (*acc)[2], float& acc_b2) {
    const float dof = dLds * (1.0f - c.s * c.s);
    float dyh[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float dy = dof * h.w2[e] * c.dg[e];
        dyh[e] = dy * h.lw[e];
        if (acc) { acc[1][e] += dy * c.yh[e]; acc[2][e] += dy; acc[3][e] += dof * c.a[e]; }
    }
    if (acc) acc_b2 += dof;
    const float m1 = wave_sum(dyh[0] + dyh[1]) * (1.0f / 128.0f);
    const float m2 = wave_sum(dyh[0] * c.yh[0] + dyh[1] * c.yh[1]) * (1.0f / 128.0f);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        dz[e] = c.rstd * (dyh[e] - m1 - c.yh[e] * m2);
        if (acc) acc[0][e] += dz[e];
    }
}

// pairwise_logistic_ranking_loss (utils/losses.py:18-41) for one keypoint set per blockIdx.y:
// block (set, i) walks all j.  It evaluates pair (i,j) [s = head(u_j - u_i)] for the loss, the head grads and
// -dz into du_i, and the mirrored pair (j,i) [head(u_i - u_j)] only for its +dz into du_i, so du_i is owned
// by one block: no atomics on du.  Un-normalised; gd_pair_rank scales by gloss/count afterwards.
__global__ __launch_bounds__(256) void pair_rank_kernel(const float* u, const float* depth, const int* counts,
                                                        const float* b1, const float* lw, const float* lb,
                                                        const float* w2, const float* b2, float* du, float* hg,
                                                        float* loss_sum, int* pair_cnt, int Nmax, float thr) {
    __shared__ float sacc[4][5][128];
    __shared__ float sred[4][2];
    const int set = blockIdx.y, i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = counts ? counts[set] : Nmax;
    float* dui = du + ((long)set * Nmax + i) * 128;
    if (i >= n) {
        if (threadIdx.x < 128) dui[threadIdx.x] = 0.f;
        return;
    }
    const HeadW h = load_head(b1, lw, lb, w2, b2, lane);
    const float* ub = u + (long)set * Nmax * 128;
    const float ui[2] = {ub[(long)i * 128 + lane], ub[(long)i * 128 + lane + 64]};
    const float di = depth[(long)set * Nmax + i];
    float acc[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    float acc_b2 = 0.f, dacc[2] = {0.f, 0.f}, lsum = 0.f;
    int cnt = 0;
    for (int j = wave; j < n; j += 4) {
        const float dd = depth[(long)set * Nmax + j] - di;   // d_j - d_i
        if (!(fabsf(dd) > thr)) continue;                   // wave-uniform
        const float alpha = dd > 0.f ? 1.f : -1.f;
        const float uj[2] = {ub[(long)j * 128 + lane], ub[(long)j * 128 + lane + 64]};
        HeadCache c;
        float z[2] = {uj[0] - ui[0] + h.b1[0], uj[1] - ui[1] + h.b1[1]};
        float dz[2];
        // pair (i, j): alpha_ij = sign(d_j - d_i)
        float s = head_eval(z, h, c);
        lsum += log1pf(expf(-alpha * s));
        head_back(-alpha / (1.0f + expf(alpha * s)), h, c, dz, acc, acc_b2);
        dacc[0] -= dz[0]; dacc[1] -= dz[1];
        ++cnt;
        // mirrored pair (j, i): z' = u_i - u_j + b1, alpha_ji = -alpha
        z[0] = ui[0] - uj[0] + h.b1[0]; z[1] = ui[1] - uj[1] + h.b1[1];
        s = head_eval(z, h, c);
        head_back(alpha / (1.0f + expf(-alpha * s)), h, c, dz, nullptr, acc_b2);
        dacc[0] += dz[0]; dacc[1] += dz[1];
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        sacc[wave][0][lane + 64 * e] = dacc[e];
#pragma unroll
        for (int t = 0; t < 4; ++t) sacc[wave][1 + t][lane + 64 * e] = acc[t][e];
    }
    if (lane == 0) { sred[wave][0] = acc_b2; sred[wave][1] = lsum; }
    __shared__ int scnt[4];
    if (lane == 0) scnt[wave] = cnt;
    __syncthreads();
    for (int idx = threadIdx.x; idx < 5 * 128; idx += 256) {
        const int t = idx >> 7, k = idx & 127;
        const float v = sacc[0][t][k] + sacc[1][t][k] + sacc[2][t][k] + sacc[3][t][k];
        if (t == 0) dui[k] = v;
        else atomicAdd(hg + (long)set * HG_SIZE + (t - 1) * 128 + k, v);
    }
    if (threadIdx.x == 0) {
        atomicAdd(hg + (long)set * HG_SIZE + 512, sred[0][0] + sred[1][0] + sred[2][0] + sred[3][0]);
        atomicAdd(loss_sum + set, sred[0][1] + sred[1][1] + sred[2][1] + sred[3][1]);
        atomicAdd(pair_cnt + set, scnt[0] + scnt[1] + scnt[2] + scnt[3]);
    }
}

// ---- the same loss with EIGHT lanes per pair (16 of the 128 head channels per lane, 8 pairs per wave-instruction).
// The wave-per-pair kernel above spends most of its instructions on 64-lane reductions (5 per head evaluation) and on
// tanh / exp / log1p replicated over 64 lanes; here a reduction is 3 DPP steps inside an 8-lane group and the scalar
// tail is shared by 8 pairs.  Measured on MI355X (64 sets x 300 keypoints): 3.78 ms -> see profiles/README.md.
__device__ __forceinline__ float oct_sum(float v) {   // sum over the lane's 8-lane group, result in all 8 lanes
    v += dpp_f32<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v += dpp_f32<0x141, 0xF>(v);   // row_half_mirror
    return v;
}

// EPL head channels per lane: 16 -> 8 lanes per pair (8 pairs per wave-instruction), 8 -> 16 lanes per pair (4 pairs).
// The 16-channel form needs 332 registers (one wave per SIMD: every VALU dependency and LDS read stalls the SIMD); the
// 8-channel form fits three waves per SIMD for 2x the replicated per-pair scalar tail (tanh / exp / log1p).
template <int EPL> __device__ __forceinline__ float grp_sum(float v);
template <> __device__ __forceinline__ float grp_sum<16>(float v) { return oct_sum(v); }
template <> __device__ __forceinline__ float grp_sum<8>(float v) { return row16_sum(v); }

// Cached per evaluation: yh = LN-normalised z, q = w2 * GELU'(y), a = GELU(y) and the two group sums the LayerNorm
// backward needs, P1 = sum q ln_w and P2 = sum q ln_w yh — they do not depend on the upstream gradient, so the backward is
// one pass over the lane's channels with no reduction of its own.
template <int EPL> struct Head8 { float yh[EPL], q[EPL], a[EPL], rstd, s, P1, P2; };

// z = sign * d + b1 ; returns s = tanh(w2 . GELU(LN(z)) + b2) and caches what the backward needs
template <int EPL>
__device__ __forceinline__ float head8_eval(const float (&d)[EPL], float sign, const float (*sh)[128], int k0, float hb2, Head8<EPL>& c) {
    float z[EPL], sum = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) { z[e] = fmaf(sign, d[e], sh[0][k0 + e]); sum += z[e]; }
    const float mu = grp_sum<EPL>(sum) * (1.0f / 128.0f);
    float var = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) { z[e] -= mu; var = fmaf(z[e], z[e], var); }
    c.rstd = rsqrtf(grp_sum<EPL>(var) * (1.0f / 128.0f) + 1e-5f);
    float o = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        c.yh[e] = z[e] * c.rstd;
        const float y = fmaf(c.yh[e], sh[1][k0 + e], sh[2][k0 + e]);
        float Phi, ex;
        gelu_parts(y, Phi, ex);
        c.a[e] = y * Phi;
        c.q[e] = sh[3][k0 + e] * fmaf(y * ex, 0.39894228040143268f, Phi);
        const float ql = c.q[e] * sh[1][k0 + e];
        p1 += ql;
        p2 = fmaf(ql, c.yh[e], p2);
        o = fmaf(c.a[e], sh[3][k0 + e], o);
    }
    c.P1 = grp_sum<EPL>(p1) * (1.0f / 128.0f);
    c.P2 = grp_sum<EPL>(p2) * (1.0f / 128.0f);
    // tanh(x) = 1 - 2 / (1 + e^2x): the per-pair scalar tail runs replicated in every lane of the group, so it is kept to a
    // handful of instructions (v_exp / v_rcp; |error| ~1e-7, inf -> +-1 without a clamp)
    c.s = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * (grp_sum<EPL>(o) + hb2)));
    return c.s;
}

template <int EPL, bool ACC>
__device__ __forceinline__ void head8_back(float dLds, const float (*sh)[128], int k0, const Head8<EPL>& c, float (&dz)[EPL],
                                           float (*acc)[EPL], float& acc_b2) {
    const float dof = dLds * (1.0f - c.s * c.s);
    const float rd = c.rstd * dof;
    if (ACC) acc_b2 += dof;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        // dz = rstd * (dyh - mean(dyh) - yh * mean(dyh yh)),  dyh = dof q ln_w
        dz[e] = rd * (fmaf(c.q[e], sh[1][k0 + e], -c.P1) - c.yh[e] * c.P2);
        if (ACC) {
            const float dy = dof * c.q[e];
            acc[0][e] += dz[e]; acc[1][e] = fmaf(dy, c.yh[e], acc[1][e]); acc[2][e] += dy; acc[3][e] = fmaf(dof, c.a[e], acc[3][e]);
        }
    }
}

template <int EPL>
__global__ __launch_bounds__(256, EPL == 8 ? 3 : 1) void pair_rank8_kernel(const float* u, const float* depth, const int* counts,
                                                         const float* b1, const float* lw, const float* lb,
                                                         const float* w2, const float* b2, float* du, float* hg,
                                                         float* loss_sum, int* pair_cnt, int Nmax, float thr) {
    constexpr int LPP = 128 / EPL, PPW = 64 / LPP;              // lanes per pair, pairs per wave
    __shared__ __attribute__((aligned(16))) float sh[4][128];   // b1, ln_w, ln_b, w2
    __shared__ float sacc[4][5][128];
    __shared__ float sred[4][2];
    __shared__ int scnt[4];
    const int set = blockIdx.y, i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPP, ps = lane / LPP, k0 = sub * EPL;
    const int n = counts ? counts[set] : Nmax;
    float* dui = du + ((long)set * Nmax + i) * 128;
    if (i >= n) {
        if (threadIdx.x < 128) dui[threadIdx.x] = 0.f;
        return;
    }
    if (threadIdx.x < 128) {
        sh[0][threadIdx.x] = b1[threadIdx.x]; sh[1][threadIdx.x] = lw[threadIdx.x];
        sh[2][threadIdx.x] = lb[threadIdx.x]; sh[3][threadIdx.x] = w2[threadIdx.x];
    }
    __syncthreads();
    const float hb2 = b2[0];
    const float* ub = u + (long)set * Nmax * 128;
    float ui[EPL];
#pragma unroll
    for (int q = 0; q < EPL / 4; ++q) {
        const f32x4 v = *(const f32x4*)(ub + (long)i * 128 + k0 + 4 * q);
#pragma unroll
        for (int k = 0; k < 4; ++k) ui[4 * q + k] = v[k];
    }
    const float di = depth[(long)set * Nmax + i];
    float acc[4][EPL], dacc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { acc[0][e] = acc[1][e] = acc[2][e] = acc[3][e] = 0.f; dacc[e] = 0.f; }
    float acc_b2 = 0.f, lsum = 0.f;
    int cnt = 0;
    for (int jb = wave * PPW; jb < n; jb += 4 * PPW) {
        const int j = jb + ps;
        const bool inr = j < n;
        const float dd = inr ? depth[(long)set * Nmax + j] - di : 0.f;   // d_j - d_i
        const bool valid = inr && fabsf(dd) > thr;
        if (!__any(valid)) continue;
        const float alpha = dd > 0.f ? 1.f : -1.f, vf = valid ? 1.f : 0.f;
        float d[EPL];
#pragma unroll
        for (int q = 0; q < EPL / 4; ++q) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (inr) v = *(const f32x4*)(ub + (long)j * 128 + k0 + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) d[4 * q + k] = v[k] - ui[4 * q + k];
        }
        Head8<EPL> c;
        float dz[EPL];
        // pair (i, j): z = u_j - u_i + b1, alpha_ij = sign(d_j - d_i)
        float s = head8_eval<EPL>(d, 1.0f, sh, k0, hb2, c);
        const float em = __expf(-alpha * s);                       // |s| < 1: e^-as in (0.36, 2.72)
        lsum += vf * __logf(1.0f + em);
        head8_back<EPL, true>(vf * -alpha * em * __builtin_amdgcn_rcpf(1.0f + em), sh, k0, c, dz, acc, acc_b2);   // -a / (1 + e^as)
#pragma unroll
        for (int e = 0; e < EPL; ++e) dacc[e] -= dz[e];
        cnt += valid ? 1 : 0;
        // mirrored pair (j, i): z' = u_i - u_j + b1, alpha_ji = -alpha; only its dz reaches du_i
        s = head8_eval<EPL>(d, -1.0f, sh, k0, hb2, c);
        head8_back<EPL, false>(vf * alpha * __builtin_amdgcn_rcpf(1.0f + __expf(-alpha * s)), sh, k0, c, dz, nullptr, acc_b2);
#pragma unroll
        for (int e = 0; e < EPL; ++e) dacc[e] += dz[e];
    }
    // sum the pair slots of the wave (lanes with equal sub), then the 4 waves through LDS
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
#pragma unroll
        for (int o = LPP; o < 64; o <<= 1) {
            dacc[e] += __shfl_xor(dacc[e], o, 64);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t][e] += __shfl_xor(acc[t][e], o, 64);
        }
    }
    if (ps == 0) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            sacc[wave][0][k0 + e] = dacc[e];
#pragma unroll
            for (int t = 0; t < 4; ++t) sacc[wave][1 + t][k0 + e] = acc[t][e];
        }
    }
    // per-pair scalars live replicated in the lanes of a group: count them once
    const float b2s = wave_sum(sub == 0 ? acc_b2 : 0.f), ls = wave_sum(sub == 0 ? lsum : 0.f);
    const float cs = wave_sum(sub == 0 ? (float)cnt : 0.f);
    if (lane == 0) { sred[wave][0] = b2s; sred[wave][1] = ls; scnt[wave] = (int)(cs + 0.5f); }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 5 * 128; idx += 256) {
        const int t = idx >> 7, k = idx & 127;
        const float v = sacc[0][t][k] + sacc[1][t][k] + sacc[2][t][k] + sacc[3][t][k];
        if (t == 0) dui[k] = v;
        else atomicAdd(hg + (long)set * HG_SIZE + (t - 1) * 128 + k, v);
    }
    if (threadIdx.x == 0) {
        atomicAdd(hg + (long)set * HG_SIZE + 512, sred[0][0] + sred[1][0] + sred[2][0] + sred[3][0]);
        atomicAdd(loss_sum + set, sred[0][1] + sred[1][1] + sred[2][1] + sred[3][1]);
        atomicAdd(pair_cnt + set, scnt[0] + scnt[1] + scnt[2] + scnt[3]);
    }
}

// ---- tiled form: every UNORDERED pair {i, j} is evaluated once.  The kernel above gives each row i its own block and,
// to keep du_i local, evaluates (i, j) and the mirror (j, i) for every j — each head evaluation is done twice across the
// grid.  Here a block owns a 16 x 32 tile of the strict upper triangle (i < j): the four pair slots of a wave hold four
// rows i, the block sweeps the tile's 32 columns j (each wave starting at a different column), and for a pair both ordered
// terms are taken at once:  g = dz(i,j) - dz(j,i);  du_i -= g stays in registers, du_j += g is summed over the wave's four
// slots by two lane shuffles and over the four waves by LDS atomic adds; the tile's 16 + 32 partial rows then go to du with
// no-return global atomic adds (du is zero-filled first; fp32 summation order is not fixed, as for the head gradients).
#define RT_I 16
#define RT_J 32
__global__ __launch_bounds__(256, 2) void pair_rank_tile_kernel(const float* u, const float* depth, const int* counts,
                                                               const float* b1, const float* lw, const float* lb,
                                                               const float* w2, const float* b2, float* du, float* hg,
                                                               float* loss_sum, int* pair_cnt, int Nmax, float thr, int nti) {
    constexpr int EPL = 8;
    __shared__ __attribute__((aligned(16))) float sh[4][128];   // b1, ln_w, ln_b, w2
    __shared__ float sJ[RT_J][128];
    __shared__ float sacc[4][4][128];
    __shared__ float sred[4][2];
    __shared__ int scnt[4];
    const int set = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15, ps = lane >> 4, k0 = sub * EPL;
    const int ti = blockIdx.x % nti, tj = blockIdx.x / nti;
    const int n = counts ? counts[set] : Nmax;
    const int i0 = ti * RT_I, j0 = tj * RT_J;
    if (j0 + RT_J - 1 <= i0 || i0 >= n || j0 >= n) return;      // tile entirely on / below the diagonal, or past the set
    if (threadIdx.x < 128) {
        sh[0][threadIdx.x] = b1[threadIdx.x]; sh[1][threadIdx.x] = lw[threadIdx.x];
        sh[2][threadIdx.x] = lb[threadIdx.x]; sh[3][threadIdx.x] = w2[threadIdx.x];
    }
    for (int q = threadIdx.x; q < RT_J * 128; q += 256) (&sJ[0][0])[q] = 0.f;
    __syncthreads();
    const float hb2 = b2[0];
    const float* ub = u + (long)set * Nmax * 128;
    const int i = i0 + wave * 4 + ps;
    const bool iok = i < n;
    float ui[EPL];
#pragma unroll
    for (int q = 0; q < EPL / 4; ++q) {
        const f32x4 v = *(const f32x4*)(ub + (long)min(i, Nmax - 1) * 128 + k0 + 4 * q);
#pragma unroll
        for (int k = 0; k < 4; ++k) ui[4 * q + k] = v[k];
    }
    const float di = depth[(long)set * Nmax + min(i, Nmax - 1)];
    float acc[4][EPL], dacc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { acc[0][e] = acc[1][e] = acc[2][e] = acc[3][e] = 0.f; dacc[e] = 0.f; }
    float acc_b2 = 0.f, lsum = 0.f;
    int cnt = 0;
    for (int it = 0; it < RT_J; ++it) {
        const int jj = (it + 8 * wave) & (RT_J - 1), j = j0 + jj;   // uniform over the wave
        if (j >= n) continue;
        const float dd = depth[(long)set * Nmax + j] - di;         // d_j - d_i
        const bool valid = iok && i < j && fabsf(dd) > thr;
        if (!__any(valid)) continue;
        const float alpha = dd > 0.f ? 1.f : -1.f, vf = valid ? 1.f : 0.f;
        float d[EPL];
#pragma unroll
        for (int q = 0; q < EPL / 4; ++q) {
            const f32x4 v = *(const f32x4*)(ub + (long)j * 128 + k0 + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) d[4 * q + k] = v[k] - ui[4 * q + k];
        }
        Head8<EPL> c;
        float dz[EPL], g[EPL];
        // ordered pair (i, j): z = u_j - u_i + b1, alpha_ij = sign(d_j - d_i)
        float s = head8_eval<EPL>(d, 1.0f, sh, k0, hb2, c);
        float em = __expf(-alpha * s);
        lsum += vf * __logf(1.0f + em);
        head8_back<EPL, true>(vf * -alpha * em * __builtin_amdgcn_rcpf(1.0f + em), sh, k0, c, dz, acc, acc_b2);
#pragma unroll
        for (int e = 0; e < EPL; ++e) g[e] = dz[e];
        // ordered pair (j, i): z' = u_i - u_j + b1, alpha_ji = -alpha
        s = head8_eval<EPL>(d, -1.0f, sh, k0, hb2, c);
        em = __expf(alpha * s);
        lsum += vf * __logf(1.0f + em);
        head8_back<EPL, true>(vf * alpha * em * __builtin_amdgcn_rcpf(1.0f + em), sh, k0, c, dz, acc, acc_b2);
        cnt += valid ? 2 : 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            g[e] -= dz[e];                       // d loss / d(u_j - u_i)
            dacc[e] -= g[e];
            float t = g[e];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            if (ps == 0) atomicAdd(&sJ[jj][k0 + e], t);
        }
    }
    // head-parameter gradients: sum the 4 slots of the wave, then the 4 waves through LDS
#pragma unroll
    for (int e = 0; e < EPL; ++e)
#pragma unroll
        for (int o = 16; o < 64; o <<= 1)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t][e] += __shfl_xor(acc[t][e], o, 64);
    if (ps == 0) {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
#pragma unroll
            for (int t = 0; t < 4; ++t) sacc[wave][t][k0 + e] = acc[t][e];
    }
    if (iok) {
        float* dui = du + ((long)set * Nmax + i) * 128 + k0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) unsafeAtomicAdd(dui + e, dacc[e]);
    }
    const float b2s = wave_sum(sub == 0 ? acc_b2 : 0.f), ls = wave_sum(sub == 0 ? lsum : 0.f);
    const float cs = wave_sum(sub == 0 ? (float)cnt : 0.f);
    if (lane == 0) { sred[wave][0] = b2s; sred[wave][1] = ls; scnt[wave] = (int)(cs + 0.5f); }
    __syncthreads();
    for (int idx = threadIdx.x; idx < RT_J * 128; idx += 256) {
        const int jj = idx >> 7, k = idx & 127;
        if (j0 + jj < n) unsafeAtomicAdd(du + ((long)set * Nmax + j0 + jj) * 128 + k, sJ[jj][k]);
    }
    for (int idx = threadIdx.x; idx < 4 * 128; idx += 256) {
        const int t = idx >> 7, k = idx & 127;
        unsafeAtomicAdd(hg + (long)set * HG_SIZE + t * 128 + k, sacc[0][t][k] + sacc[1][t][k] + sacc[2][t][k] + sacc[3][t][k]);
    }
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(hg + (long)set * HG_SIZE + 512, sred[0][0] + sred[1][0] + sred[2][0] + sred[3][0]);
        unsafeAtomicAdd(loss_sum + set, sred[0][1] + sred[1][1] + sred[2][1] + sred[3][1]);
        atomicAdd(pair_cnt + set, scnt[0] + scnt[1] + scnt[2] + scnt[3]);
    }
}

// L1(head(f1 - f2), tanh(d1 - d2)) rows (src/finetune_timm_vggt.py:475-479): one wave per keypoint.
// u holds [P][2][Nmax][128]; writes du for both views (du1 = +dz, du2 = -dz, scaled by gscale[p]/n),
// head grads (scaled) into hg[p], loss_sum[p] += |s - t| / n.
#define DL1_KPW 4   // keypoints per wave: the head-gradient partials of 16 keypoints meet in LDS before ONE set of atomics per block
__global__ __launch_bounds__(256) void depth_l1_kernel(const float* u, const float* d1, const float* d2,
                                                       const int* counts, const float* gscale, const float* b1,
                                                       const float* lw, const float* lb, const float* w2,
                                                       const float* b2, float* du, float* hg, float* loss_sum,
                                                       int Nmax) {
    __shared__ float sacc[4][4][128];
    __shared__ float sred[4][2];
    const int p = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = counts ? counts[p] : Nmax;
    const HeadW h = load_head(b1, lw, lb, w2, b2, lane);
    float acc[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    float acc_b2 = 0.f, lsum = 0.f;
    for (int kk = 0; kk < DL1_KPW; ++kk) {
        const int k = (blockIdx.x * 4 + wave) * DL1_KPW + kk;
        if (k >= Nmax) break;
        float* du1 = du + (((long)p * 2 + 0) * Nmax + k) * 128;
        float* du2 = du + (((long)p * 2 + 1) * Nmax + k) * 128;
        if (k >= n) {
            du1[lane] = du1[lane + 64] = du2[lane] = du2[lane + 64] = 0.f;
            continue;
        }
        const float* u1 = u + (((long)p * 2 + 0) * Nmax + k) * 128;
        const float* u2 = u + (((long)p * 2 + 1) * Nmax + k) * 128;
        float z[2] = {u1[lane] - u2[lane] + h.b1[0], u1[lane + 64] - u2[lane + 64] + h.b1[1]};
        HeadCache c;
        const float s = head_eval(z, h, c);
        const float t = tanhf(d1[(long)p * Nmax + k] - d2[(long)p * Nmax + k]);
        const float diff = s - t, w = (gscale ? gscale[p] : 1.f) / (float)n;
        float dz[2];
        head_back((diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * w, h, c, dz, acc, acc_b2);
        lsum += fabsf(diff) / (float)n;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            du1[lane + 64 * e] = dz[e];
            du2[lane + 64 * e] = -dz[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) sacc[wave][tt][lane + 64 * e] = acc[tt][e];
    if (lane == 0) { sred[wave][0] = acc_b2; sred[wave][1] = lsum; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 4 * 128; idx += 256) {
        const int tt = idx >> 7, kx = idx & 127;
        const float v = sacc[0][tt][kx] + sacc[1][tt][kx] + sacc[2][tt][kx] + sacc[3][tt][kx];
        if (v != 0.f) atomicAdd(hg + (long)p * HG_SIZE + tt * 128 + kx, v);
    }
    if (threadIdx.x == 0) {
        atomicAdd(hg + (long)p * HG_SIZE + 512, sred[0][0] + sred[1][0] + sred[2][0] + sred[3][0]);
        atomicAdd(loss_sum + p, sred[0][1] + sred[1][1] + sred[2][1] + sred[3][1]);
    }
}

// du[set] *= g[set]/cnt, hg_out += hg[set] * g[set]/cnt, loss[set] = loss_sum/cnt (0 when no valid pair)
__global__ __launch_bounds__(256) void pair_rank_finalize_kernel(float* du, const float* hg, const float* loss_sum,
                                                                 const int* pair_cnt, const float* gscale,
                                                                 float* hg_out, float* hg_sets, float* loss, int Nmax) {
    const int set = blockIdx.x;
    const int cnt = pair_cnt[set];
    const float sc = cnt > 0 ? (gscale ? gscale[set] : 1.f) / (float)cnt : 0.f;
    for (long idx = threadIdx.x; idx < (long)Nmax * 128; idx += 256) du[(long)set * Nmax * 128 + idx] *= sc;
    for (int idx = threadIdx.x; idx < HG_SIZE; idx += 256) {
        const float v = idx <= 512 ? hg[(long)set * HG_SIZE + idx] * sc : 0.f;
        if (hg_sets) hg_sets[(long)set * HG_SIZE + idx] = v;
        if (hg_out && idx <= 512) atomicAdd(hg_out + idx, v);
    }
    if (threadIdx.x == 0) loss[set] = cnt > 0 ? loss_sum[set] / (float)cnt : 0.f;
}

// out[c] += sum_r in[r][c]
__global__ __launch_bounds__(256) void row_sum_cols_kernel(const float* in, float* out, int rows, int cols) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += in[(long)r * cols + c];
    out[c] += s;
}

// Eager DepthAwareFeatureFusion.forward (utils/model.py:101-127, depths = None branch) on pre-projected rows
// u = W1 f [M,128]: out[m] = tanh(w2 . GELU(LN(u[m] + b1)) + b2); one wave per row.  The backward recomputes the row,
// writes du[m] and accumulates the head-parameter gradients (block partials through LDS, one set of atomics per block).
#define DH_RPW 8   // rows per wave
__global__ __launch_bounds__(256) void depth_head_fwd_kernel(const float* u, const float* b1, const float* lw, const float* lb,
                                                             const float* w2, const float* b2, float* out, int M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const HeadW h = load_head(b1, lw, lb, w2, b2, lane);
    for (int kk = 0; kk < DH_RPW; ++kk) {
        const long m = ((long)blockIdx.x * 4 + wave) * DH_RPW + kk;
        if (m >= M) break;
        const float z[2] = {u[m * 128 + lane] + h.b1[0], u[m * 128 + lane + 64] + h.b1[1]};
        HeadCache c;
        const float s = head_eval(z, h, c);
        if (lane == 0) out[m] = s;
    }
}

__global__ __launch_bounds__(256) void depth_head_bwd_kernel(const float* u, const float* dout, const float* b1, const float* lw,
                                                             const float* lb, const float* w2, const float* b2, float* du,
                                                             float* hg, int M) {
    __shared__ float sacc[4][4][128];
    __shared__ float sred[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const HeadW h = load_head(b1, lw, lb, w2, b2, lane);
    float acc[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    float acc_b2 = 0.f;
    for (int kk = 0; kk < DH_RPW; ++kk) {
        const long m = ((long)blockIdx.x * 4 + wave) * DH_RPW + kk;
        if (m >= M) break;
        const float z[2] = {u[m * 128 + lane] + h.b1[0], u[m * 128 + lane + 64] + h.b1[1]};
        HeadCache c;
        head_eval(z, h, c);
        float dz[2];
        head_back(dout[m], h, c, dz, acc, acc_b2);
        du[m * 128 + lane] = dz[0];
        du[m * 128 + lane + 64] = dz[1];
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) sacc[wave][tt][lane + 64 * e] = acc[tt][e];
    if (lane == 0) sred[wave] = acc_b2;
    __syncthreads();
    for (int idx = threadIdx.x; idx < 4 * 128; idx += 256) {
        const int tt = idx >> 7, kx = idx & 127;
        const float v = sacc[0][tt][kx] + sacc[1][tt][kx] + sacc[2][tt][kx] + sacc[3][tt][kx];
        if (v != 0.f) atomicAdd(hg + tt * 128 + kx, v);
    }
    if (threadIdx.x == 0) atomicAdd(hg + 512, sred[0] + sred[1] + sred[2] + sred[3]);
}

// ---------------------------------------------------------------------------------------------------
extern "C" int gd_smooth_ap(const float* sim, const float* pts3d_1, const float* pts3d_2, const int* counts, int P,
                            int Nmax, int variant, float thres3d_neg, float temp, float* loss, float* dsim,
                            float* row_ws, void* stream) {
    GD_REQUIRE(P > 0 && Nmax > 0 && temp > 0.f, "gd_smooth_ap: bad arguments");
    GD_REQUIRE(variant == 0 || variant == 1, "gd_smooth_ap: variant must be 0 (vggt) or 1 (mast3r)");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(smooth_ap_kernel, dim3(Nmax, P), dim3(256), 0, s, sim, pts3d_1, pts3d_2, counts, row_ws, dsim,
                       Nmax, variant, thres3d_neg, 1.0f / temp);
    hipLaunchKernelGGL(row_sum_kernel, dim3(P), dim3(256), 0, s, row_ws, loss, Nmax);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_smooth_ap_me(const float* sim, const float* pts3d_1, const float* pts3d_2, const int* counts, int P,
                               int Nmax, float thres3d_pos, float thres3d_neg, float temp, float* loss, float* dsim,
                               float* row_ws, void* stream) {
    GD_REQUIRE(P > 0 && Nmax > 0 && temp > 0.f, "gd_smooth_ap_me: bad arguments");
    GD_REQUIRE(Nmax <= AP_ME_NMAX, "gd_smooth_ap_me: at most %d keypoints per view (got %d)", AP_ME_NMAX, Nmax);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(smooth_ap_me_kernel, dim3(Nmax, P), dim3(256), 0, s, sim, pts3d_1, pts3d_2, counts, row_ws, dsim, Nmax,
                       thres3d_pos, thres3d_neg, 1.0f / temp);
    int fb = (int)(((long)Nmax * Nmax + 255) / 256);
    if (fb > 1024) fb = 1024;
    hipLaunchKernelGGL(smooth_ap_me_finalize_kernel, dim3(fb, P), dim3(256), 0, s, row_ws, loss, dsim, Nmax);
    GD_LAUNCH_OK();
    return 0;
}

// workspace floats: hg [S,516] | loss_sum [S] | pair_cnt [S] (int)
extern "C" size_t gd_pair_rank_workspace_bytes(int S) { return (size_t)S * (HG_SIZE + 2) * sizeof(float); }

extern "C" int gd_pair_rank(const float* u, const float* depth, const int* counts, const float* gscale, int S,
                            int Nmax, float depth_threshold, const float* b1, const float* ln_w, const float* ln_b,
                            const float* w2, const float* b2, float* loss, float* du, float* head_grad,
                            float* head_grad_sets, void* workspace, void* stream) {
    GD_REQUIRE(S > 0 && Nmax > 0, "gd_pair_rank: bad shape");
    hipStream_t s = (hipStream_t)stream;
    float* hg = (float*)workspace;
    float* lsum = hg + (long)S * HG_SIZE;
    int* cnt = (int*)(lsum + S);
    hipMemsetAsync(workspace, 0, gd_pair_rank_workspace_bytes(S), s);
    const int wide = gd_knobs().pair_rank_wave;   // GD_PAIR_RANK_WAVE=1 selects the older one-wave-per-pair kernel (A/B testing)
    if (wide == 1) hipLaunchKernelGGL(pair_rank_kernel, dim3(Nmax, S), dim3(256), 0, s, u, depth, counts, b1, ln_w, ln_b, w2, b2, du,
                                 hg, lsum, cnt, Nmax, depth_threshold);
    else if (wide == 2) hipLaunchKernelGGL(pair_rank8_kernel<16>, dim3(Nmax, S), dim3(256), 0, s, u, depth, counts, b1, ln_w, ln_b, w2,
                                           b2, du, hg, lsum, cnt, Nmax, depth_threshold);   // GD_PAIR_RANK_WAVE=2: 8 lanes per pair
    else if (wide == 3) hipLaunchKernelGGL(pair_rank8_kernel<8>, dim3(Nmax, S), dim3(256), 0, s, u, depth, counts, b1, ln_w, ln_b, w2, b2, du,
                                           hg, lsum, cnt, Nmax, depth_threshold);           // GD_PAIR_RANK_WAVE=3: one block per row i
    else {
        hipMemsetAsync(du, 0, (size_t)S * Nmax * 128 * sizeof(float), s);
        const int nti = gd_cdiv(Nmax, RT_I), ntj = gd_cdiv(Nmax, RT_J);
        hipLaunchKernelGGL(pair_rank_tile_kernel, dim3(nti * ntj, S), dim3(256), 0, s, u, depth, counts, b1, ln_w, ln_b, w2, b2,
                           du, hg, lsum, cnt, Nmax, depth_threshold, nti);
    }
    hipLaunchKernelGGL(pair_rank_finalize_kernel, dim3(S), dim3(256), 0, s, du, hg, lsum, cnt, gscale, head_grad,
                       head_grad_sets, loss, Nmax);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_depth_l1(const float* u, const float* d1, const float* d2, const int* counts, const float* gscale,
                           int P, int Nmax, const float* b1, const float* ln_w, const float* ln_b, const float* w2,
                           const float* b2, float* loss, float* du, float* head_grad, float* head_grad_sets,
                           void* workspace, void* stream) {
    GD_REQUIRE(P > 0 && Nmax > 0, "gd_depth_l1: bad shape");
    GD_REQUIRE(head_grad_sets || workspace, "gd_depth_l1: head_grad_sets [P,516] or a workspace of that size is required");
    hipStream_t s = (hipStream_t)stream;
    float* hg = head_grad_sets ? head_grad_sets : (float*)workspace;   // per-pair head gradients [P, 516]
    hipMemsetAsync(hg, 0, (size_t)P * HG_SIZE * sizeof(float), s);
    hipMemsetAsync(loss, 0, (size_t)P * sizeof(float), s);
    hipLaunchKernelGGL(depth_l1_kernel, dim3(gd_cdiv(Nmax, 4 * DL1_KPW), P), dim3(256), 0, s, u, d1, d2, counts, gscale, b1, ln_w,
                       ln_b, w2, b2, du, hg, loss, Nmax);
    if (head_grad) hipLaunchKernelGGL(row_sum_cols_kernel, dim3(gd_cdiv(HG_SIZE, 256)), dim3(256), 0, s, hg, head_grad, P, HG_SIZE);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_depth_head_fwd(const float* u, int M, const float* b1, const float* ln_w, const float* ln_b,
                                 const float* w2, const float* b2, float* out, void* stream) {
    GD_REQUIRE(M > 0, "gd_depth_head_fwd: bad shape");
    hipLaunchKernelGGL(depth_head_fwd_kernel, dim3(gd_cdiv(M, 4 * DH_RPW)), dim3(256), 0, (hipStream_t)stream, u, b1, ln_w, ln_b,
                       w2, b2, out, M);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_depth_head_bwd(const float* u, const float* dout, int M, const float* b1, const float* ln_w,
                                 const float* ln_b, const float* w2, const float* b2, float* du, float* head_grad,
                                 void* stream) {
    GD_REQUIRE(M > 0 && head_grad, "gd_depth_head_bwd: bad arguments");
    hipLaunchKernelGGL(depth_head_bwd_kernel, dim3(gd_cdiv(M, 4 * DH_RPW)), dim3(256), 0, (hipStream_t)stream, u, dout, b1, ln_w,
                       ln_b, w2, b2, du, head_grad, M);
    GD_LAUNCH_OK();
    return 0;
}


// ------------------------------------------------------------------------------------------------------------------------------
// Round 6: the step's host-side glue as kernels.  Each of these replaces a handful of torch elementwise / reduce / copy launches of a few
// microseconds each on [P]-vectors and small tensors (profiles/r06_stray_kernels.txt): none of it is arithmetic the reference asks for beyond
// `loss = w_ap ap + w_depth depth + w_intra intra + w_kl kl`, `.mean()` (src/finetune_timm_vggt.py:599-616) and the chain rule through it.
// ------------------------------------------------------------------------------------------------------------------------------
// loss = mean_p keep_p (w0 t0[p] + w1 t1[p] + w2 t2[p] + w3 t3[p]), keep_p = counts == null || counts[p] > 0 (a pair whose keypoint filter left nothing
// contributes a constant zero: src/finetune_timm_mast3r.py:604-607); terms_out [4][P] = keep_p t_i[p] (the reported terms).  One block; fixed order.
__global__ __launch_bounds__(256) void loss_combine_fwd_kernel(const float* t0, const float* t1, const float* t2, const float* t3, float w0, float w1,
                                                               float w2, float w3, const int* counts, int P, float* loss, float* terms_out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int p = threadIdx.x; p < P; p += 256) {
        const bool keep = !counts || counts[p] > 0;
        const float a = t0[p], b = t1[p], c = t2[p], d = t3[p];
        if (terms_out) {
            terms_out[p] = keep ? a : 0.f; terms_out[P + p] = keep ? b : 0.f; terms_out[2 * P + p] = keep ? c : 0.f; terms_out[3 * P + p] = keep ? d : 0.f;
        }
        // (a zero weight drops its term even when the term is not finite: 0 * nan must not poison the sum the reference never forms — it multiplies too,
        //  so keep the multiplication's semantics: w * t)
        if (keep) acc += (double)(w0 * a) + (double)(w1 * b) + (double)(w2 * c) + (double)(w3 * d);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] / (double)P);
}
// grads [4][P]: d loss / d t_i[p] = g w_i keep_p / P
__global__ __launch_bounds__(256) void loss_combine_bwd_kernel(const float* g, float w0, float w1, float w2, float w3, const int* counts, int P, float* grads) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float s = (!counts || counts[p] > 0) ? *g / (float)P : 0.f;
    grads[p] = s * w0; grads[P + p] = s * w1; grads[2 * P + p] = s * w2; grads[3 * P + p] = s * w3;
}
extern "C" int gd_loss_combine_fwd(const float* t0, const float* t1, const float* t2, const float* t3, const float* w4, const int* counts, int P,
                                   float* loss, float* terms_out, void* stream) {
    GD_REQUIRE(P > 0 && t0 && t1 && t2 && t3 && w4 && loss, "gd_loss_combine_fwd: bad arguments");
    hipLaunchKernelGGL(loss_combine_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, t0, t1, t2, t3, w4[0], w4[1], w4[2], w4[3], counts, P, loss, terms_out);
    GD_LAUNCH_OK();
    return 0;
}
extern "C" int gd_loss_combine_bwd(const float* g, const float* w4, const int* counts, int P, float* grads, void* stream) {
    GD_REQUIRE(P > 0 && g && w4 && grads, "gd_loss_combine_bwd: bad arguments");
    hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3(gd_cdiv(P, 256)), dim3(256), 0, (hipStream_t)stream, g, w4[0], w4[1], w4[2], w4[3], counts, P, grads);
    GD_LAUNCH_OK();
    return 0;
}

// The backward of the two depth losses (gd_pair_rank / gd_depth_l1 emit their gradients for a unit upstream gradient): one pass that scales and adds
//   du_out = du_rank (x 0.5 g_intra[p]) + du_l1 (x g_l1[p])   [P][2][N][128] -> rows in the order of the features (view-major [2][P][N] when vm)
//   hg_out [516] = sum_p 0.5 g_intra[p] (hg_rank[2p] + hg_rank[2p + 1]) + g_l1[p] hg_l1[p]      (fixed order over p)
__global__ __launch_bounds__(256) void depth_bwd_combine_kernel(const float* __restrict__ du_r, const float* __restrict__ du_l, const float* __restrict__ hg_r,
                                                                const float* __restrict__ hg_l, const float* __restrict__ g_l1, const float* __restrict__ g_intra,
                                                                float* __restrict__ du_out, float* __restrict__ hg_out, int P, int N, int vm, int nblk_du) {
    if ((int)blockIdx.x >= nblk_du) {      // the head-gradient columns
        const int j = (blockIdx.x - nblk_du) * 256 + threadIdx.x;
        if (j >= HG_SIZE) return;
        float acc = 0.f;
        for (int p = 0; p < P; ++p)
            acc += 0.5f * g_intra[p] * (hg_r[(long)(2 * p) * HG_SIZE + j] + hg_r[(long)(2 * p + 1) * HG_SIZE + j]) + g_l1[p] * hg_l[(long)p * HG_SIZE + j];
        hg_out[j] = acc;
        return;
    }
    const long total = (long)P * 2 * N * 32;      // float4 items
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)nblk_du * 256) {
        const long row = idx >> 5;                 // (p, v, n)
        const int c4 = (int)(idx & 31);
        const int n = (int)(row % N);
        const long pv = row / N;
        const int v = (int)(pv & 1), p = (int)(pv >> 1);
        const f32x4 a = *(const f32x4*)(du_r + idx * 4), b = *(const f32x4*)(du_l + idx * 4);
        const float gi = 0.5f * g_intra[p], gl = g_l1[p];
        const f32x4 o = {a[0] * gi + b[0] * gl, a[1] * gi + b[1] * gl, a[2] * gi + b[2] * gl, a[3] * gi + b[3] * gl};
        const long orow = vm ? ((long)v * P + p) * N + n : row;
        *(f32x4*)(du_out + orow * 128 + c4 * 4) = o;
    }
}
extern "C" int gd_depth_bwd_combine(const float* du_rank, const float* du_l1, const float* hg_rank, const float* hg_l1, const float* g_l1, const float* g_intra,
                                    int P, int N, int view_major, float* du_out, float* hg_out, void* stream) {
    GD_REQUIRE(P > 0 && N > 0 && du_rank && du_l1 && hg_rank && hg_l1 && g_l1 && g_intra && du_out && hg_out, "gd_depth_bwd_combine: bad arguments");
    const long items = (long)P * 2 * N * 32;
    const int nblk = (int)((items + 255) / 256 < 2048 ? (items + 255) / 256 : 2048);
    hipLaunchKernelGGL(depth_bwd_combine_kernel, dim3(nblk + gd_cdiv(HG_SIZE, 256)), dim3(256), 0, (hipStream_t)stream, du_rank, du_l1, hg_rank, hg_l1, g_l1, g_intra,
                       du_out, hg_out, P, N, view_major ? 1 : 0, nblk);
    GD_LAUNCH_OK();
    return 0;
}

// out [B][R][C] = in [B][R][C] * g[b], out_t [B][C][R] = its transpose (the smooth-AP backward contracts dsim g from both sides): 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void scale_and_transpose_kernel(const float* __restrict__ in, const float* __restrict__ g, float* __restrict__ out,
                                                                  float* __restrict__ out_t, int R, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const float s = g[b];
    const float* ib = in + (long)b * R * C;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) { v = ib[(long)r * C + c] * s; out[(long)b * R * C + (long)r * C + c] = v; }
        tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (r < R && c < C) out_t[(long)b * R * C + (long)c * R + r] = tile[tx][k];
    }
}
extern "C" int gd_scale_and_transpose(const float* in, const float* g, float* out, float* out_t, int B, int R, int C, void* stream) {
    GD_REQUIRE(B > 0 && R > 0 && C > 0 && in && g && out && out_t, "gd_scale_and_transpose: bad arguments");
    hipLaunchKernelGGL(scale_and_transpose_kernel, dim3(gd_cdiv(C, 32), gd_cdiv(R, 32), B), dim3(256), 0, (hipStream_t)stream, in, g, out, out_t, R, C);
    GD_LAUNCH_OK();
    return 0;
}
