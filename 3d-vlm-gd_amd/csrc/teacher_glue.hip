// Teacher -> target glue on the device: the reference does these with numpy
// round-trips and host syncs between the frozen teacher and the student losses.
#include "gd_common.h"

// vggt/utils/geometry.py:12-110 unproject_depth_map_to_point_map: world = R^T (cam - t), cam = ((u-cu) d/fu, (v-cv) d/fv, d)
__global__ void unproject_kernel(const float* depth, const float* extr, const float* intr, float* out, int S, int H, int W) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)S * H * W) return;
    const int s = i / ((long)H * W), v = (i / W) % H, u = i % W;
    const float* E = extr + s * 12;
    const float* K = intr + s * 9;
    const float d = depth[i];
    const float cx = ((float)u - K[2]) * d / K[0], cy = ((float)v - K[5]) * d / K[4], cz = d;
    const float px = cx - E[3], py = cy - E[7], pz = cz - E[11];
    // R^T p, computed as cam . R - R^T t in the reference (np.dot(cam, R_c2w^T) + t_c2w): same value up to rounding
    out[i * 3 + 0] = E[0] * px + E[4] * py + E[8] * pz;
    out[i * 3 + 1] = E[1] * px + E[5] * py + E[9] * pz;
    out[i * 3 + 2] = E[2] * px + E[6] * py + E[10] * pz;
}

// utils/functions.py:425-472 get_coview_masks (both point maps converted with extrinsic 1, as the reference does)
__global__ void coview_kernel(const float* pm1, const float* pm2, const float* K1, const float* E1, const float* K2,
                              const float* E2, unsigned char* m1, unsigned char* m2, int P, int H, int W) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)P * H * W) return;
    const int p = i / ((long)H * W);
    const float* e1 = E1 + p * 12; const float* e2 = E2 + p * 12;
    const float* k1 = K1 + p * 9;  const float* k2 = K2 + p * 9;
    // The masks are bool: the arithmetic is pinned to one explicit fp32 operation order (the reference's matmuls restated as
    // sequential fused multiply-add chains, IEEE division), no compiler contraction or fast division left to chance.
    auto P34 = [](const float* K, const float* E, float (&M)[12]) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c)
                M[r * 4 + c] = __builtin_fmaf(K[r * 3 + 2], E[8 + c], __builtin_fmaf(K[r * 3 + 1], E[4 + c], __fmul_rn(K[r * 3 + 0], E[c])));
    };
    float Pa[12], Pb[12];
    P34(k2, e2, Pa);   // view 1 points -> view 2 image
    P34(k1, e1, Pb);   // view 2 points -> view 1 image
    auto test = [&](const float* pm, const float* Pm) -> unsigned char {
        const float qx = __fsub_rn(pm[0], e1[3]), qy = __fsub_rn(pm[1], e1[7]), qz = __fsub_rn(pm[2], e1[11]);
        // torch.matmul(pm - t, R.t()):  w_k = sum_j q_j R[k][j]
        const float wx = __builtin_fmaf(qz, e1[2], __builtin_fmaf(qy, e1[1], __fmul_rn(qx, e1[0])));
        const float wy = __builtin_fmaf(qz, e1[6], __builtin_fmaf(qy, e1[5], __fmul_rn(qx, e1[4])));
        const float wz = __builtin_fmaf(qz, e1[10], __builtin_fmaf(qy, e1[9], __fmul_rn(qx, e1[8])));
        // P @ [w; 1]
        const float hx = __fadd_rn(__builtin_fmaf(Pm[2], wz, __builtin_fmaf(Pm[1], wy, __fmul_rn(Pm[0], wx))), Pm[3]);
        const float hy = __fadd_rn(__builtin_fmaf(Pm[6], wz, __builtin_fmaf(Pm[5], wy, __fmul_rn(Pm[4], wx))), Pm[7]);
        const float hz = __fadd_rn(__builtin_fmaf(Pm[10], wz, __builtin_fmaf(Pm[9], wy, __fmul_rn(Pm[8], wx))), Pm[11]);
        const float den = __fadd_rn(hz, 1e-8f);
        const float u = __fdiv_rn(hx, den), v = __fdiv_rn(hy, den);
        return (u >= 0.f && u < (float)W && v >= 0.f && v < (float)H) ? 1 : 0;
    };
    m1[i] = test(pm1 + i * 3, Pa);
    m2[i] = test(pm2 + i * 3, Pb);
}

// utils/functions.py:475-507 sample_keypoints_nms up to the candidate mask: local maxima (|s - maxpool| < 1e-6) of conf*mask
__global__ void nms_kernel(const unsigned char* mask, const float* conf, unsigned char* keep, int P, int H, int W, int r) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)P * H * W) return;
    const int p = i / ((long)H * W), y = (i / W) % H, x = i % W;
    const unsigned char* mk = mask + (long)p * H * W;
    const float* cf = conf + (long)p * H * W;
    const float s = mk[y * W + x] ? cf[y * W + x] : 0.f;
    float mx = s;   // max_pool2d pads with -inf: out-of-image cells never win
    for (int yy = max(y - r, 0); yy <= min(y + r, H - 1); ++yy)
        for (int xx = max(x - r, 0); xx <= min(x + r, W - 1); ++xx) mx = fmaxf(mx, mk[yy * W + xx] ? cf[yy * W + xx] : 0.f);
    keep[i] = (fabsf(s - mx) < 1e-6f && mk[y * W + x]) ? 1 : 0;
}

// ordered compaction of a byte mask: idx[p][k] = linear index of the k-th set cell (row-major), count[p]; one block per p
__global__ __launch_bounds__(256) void compact_kernel(const unsigned char* keep, int* idx, int* count, int n, int cap) {
    __shared__ int wsum[4];
    __shared__ int base;
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + tid;
        const bool k = i < n && keep[(long)p * n + i];
        const unsigned long long b = __ballot(k);
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(b);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (k && off + before < cap) idx[(long)p * cap + off + before] = i;
        __syncthreads();
        if (tid == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) count[p] = base;
}

// mast3r/fast_nn.py:11-62 (dist='dot'): for every active query the database row with the largest dot product.
// thread = query (vector in registers), database streamed through LDS in chunks; cross-chunk argmax by one packed
// 64-bit atomicMax per (query, chunk): key = orderable(sim) << 32 | ~index (ties -> smallest index).
#define NN_MAXD 32
#define NN_CH 512
__global__ __launch_bounds__(256) void nn_argmax_kernel(const float* Q, const float* DB, const unsigned char* active,
                                                        unsigned long long* keys, int Nq, int Nb, int D) {
    __shared__ __attribute__((aligned(16))) float sdb[NN_CH * NN_MAXD];
    const int q = blockIdx.x * 256 + threadIdx.x, j0 = blockIdx.y * NN_CH, nj = min(NN_CH, Nb - j0);
    for (int i = threadIdx.x; i < nj * D; i += 256) sdb[i] = DB[(long)j0 * D + i];
    __syncthreads();
    if (q >= Nq || (active && !active[q])) return;
    float qv[NN_MAXD];
#pragma unroll
    for (int d = 0; d < NN_MAXD; ++d) qv[d] = d < D ? Q[(long)q * D + d] : 0.f;
    float best = -INFINITY;
    int bj = 0;
    for (int j = 0; j < nj; ++j) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < NN_MAXD; ++d)
            if (d < D) s += qv[d] * sdb[j * D + d];
        if (s > best) { best = s; bj = j; }
    }
    unsigned int u = __float_as_uint(best);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    atomicMax(keys + q, ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(j0 + bj)));
}
__global__ void nn_decode_kernel(const unsigned long long* keys, const unsigned char* active, int* idx, int Nq) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Nq || (active && !active[q])) return;
    idx[q] = (int)(0xFFFFFFFFu - (unsigned)(keys[q] & 0xFFFFFFFFull));
}

// utils/functions.py:218-259 point_cloud_to_depth: mean z of the points rounding to each pixel
__global__ void pc_scatter_kernel(const float* pts, const float* K, float* acc, float* cnt, int P, int Np, int w, int h) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)P * Np) return;
    const int p = i / Np;
    const float X = pts[i * 3], Y = pts[i * 3 + 1], Z = pts[i * 3 + 2];
    if (!(Z > 0.f)) return;
    const float* k = K + p * 9;
    const float fu = rintf(X / Z * k[0] + k[2]), fv = rintf(Y / Z * k[4] + k[5]);   // torch.round = half-to-even
    if (fu >= 0.f && fu < (float)w && fv >= 0.f && fv < (float)h) {
        const long o = (long)p * w * h + (long)fv * w + (long)fu;
        atomicAdd(acc + o, Z);
        atomicAdd(cnt + o, 1.0f);
    }
}
__global__ void pc_div_kernel(float* acc, const float* cnt, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) acc[i] = cnt[i] > 0.f ? acc[i] / cnt[i] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
extern "C" int gd_unproject_depth(const float* depth, const float* extrinsic, const float* intrinsic, float* out, int S,
                                  int H, int W, void* stream) {
    GD_REQUIRE(S > 0 && H > 0 && W > 0, "gd_unproject_depth: bad shape");
    hipLaunchKernelGGL(unproject_kernel, dim3(gd_cdiv((long)S * H * W, 256)), dim3(256), 0, (hipStream_t)stream, depth, extrinsic, intrinsic, out, S, H, W);
    GD_LAUNCH_OK();
    return 0;
}
extern "C" int gd_coview_masks(const float* pm1, const float* pm2, const float* K1, const float* E1, const float* K2,
                               const float* E2, unsigned char* m1, unsigned char* m2, int P, int H, int W, void* stream) {
    GD_REQUIRE(P > 0 && H > 0 && W > 0, "gd_coview_masks: bad shape");
    hipLaunchKernelGGL(coview_kernel, dim3(gd_cdiv((long)P * H * W, 256)), dim3(256), 0, (hipStream_t)stream, pm1, pm2, K1, E1, K2, E2, m1, m2, P, H, W);
    GD_LAUNCH_OK();
    return 0;
}
extern "C" int gd_nms_keypoints(const unsigned char* mask, const float* conf, int min_distance, unsigned char* keep_ws,
                                int* idx, int* count, int P, int H, int W, int cap, void* stream) {
    GD_REQUIRE(P > 0 && H > 0 && W > 0 && min_distance >= 0 && cap > 0, "gd_nms_keypoints: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(nms_kernel, dim3(gd_cdiv((long)P * H * W, 256)), dim3(256), 0, s, mask, conf, keep_ws, P, H, W, min_distance);
    hipLaunchKernelGGL(compact_kernel, dim3(P), dim3(256), 0, s, keep_ws, idx, count, H * W, cap);
    GD_LAUNCH_OK();
    return 0;
}
extern "C" int gd_nn_argmax(const float* queries, const float* database, const unsigned char* active, int* idx,
                            unsigned long long* key_ws, int Nq, int Nb, int D, void* stream) {
    GD_REQUIRE(Nq > 0 && Nb > 0 && D > 0 && D <= NN_MAXD, "gd_nn_argmax: need 0 < D <= %d (got %d)", NN_MAXD, D);
    hipStream_t s = (hipStream_t)stream;
    hipMemsetAsync(key_ws, 0, (size_t)Nq * sizeof(unsigned long long), s);
    hipLaunchKernelGGL(nn_argmax_kernel, dim3(gd_cdiv(Nq, 256), gd_cdiv(Nb, NN_CH)), dim3(256), 0, s, queries, database, active, key_ws, Nq, Nb, D);
    hipLaunchKernelGGL(nn_decode_kernel, dim3(gd_cdiv(Nq, 256)), dim3(256), 0, s, key_ws, active, idx, Nq);
    GD_LAUNCH_OK();
    return 0;
}
extern "C" int gd_point_cloud_to_depth(const float* points, const float* K, float* depth, float* cnt_ws, int P, int Np,
                                       int w, int h, void* stream) {
    GD_REQUIRE(P > 0 && Np > 0 && w > 0 && h > 0, "gd_point_cloud_to_depth: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)P * w * h;
    hipMemsetAsync(depth, 0, n * sizeof(float), s);
    hipMemsetAsync(cnt_ws, 0, n * sizeof(float), s);
    hipLaunchKernelGGL(pc_scatter_kernel, dim3(gd_cdiv((long)P * Np, 256)), dim3(256), 0, s, points, K, depth, cnt_ws, P, Np, w, h);
    hipLaunchKernelGGL(pc_div_kernel, dim3(gd_cdiv(n, 256)), dim3(256), 0, s, depth, cnt_ws, n);
    GD_LAUNCH_OK();
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// MASt3R teacher -> distillation target `tgt_attn_map` (dust3r/dust3r/model.py:346-366), after the head mean and the
// reciprocity average:  R_l = (mean_h tgt_l + (mean_h src_l)^T) / 2  per decoder layer l (raw scaled scores; as GEMMs:
// R_l = scale/(2H) (Q1 K2^T + K1 Q2^T) over the concatenated heads, see teacher_glue.mast3r_tgt_attn_map_from_qk).
//   P_l = softmax(R_l / temperature) over keys;  P_l[:, :, 0] := min(P_l) (one scalar per layer, over the whole batch);
//   out = mean_l P_l.
// Kernel 1: one wave per (b, i) row walks all layers, accumulates the row of `out` in registers and records the row
// minimum of every layer; kernel 2 turns the row minima into the per-layer minima and overwrites column 0.
// ---------------------------------------------------------------------------------------------------------------------
#define TG_ROW_MAX 32   // 64 * 32 = 2048 keys per row in registers
__global__ __launch_bounds__(256) void mast3r_target_rows_kernel(const float* R, float* out, float* rowmin, int L, long rows, int N2,
                                                                 float inv_temp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    float acc[TG_ROW_MAX];
#pragma unroll
    for (int k = 0; k < TG_ROW_MAX; ++k) acc[k] = 0.f;
    const float c2 = inv_temp * 1.4426950408889634f;
    for (int l = 0; l < L; ++l) {
        const float* r = R + ((long)l * rows + row) * N2;
        float v[TG_ROW_MAX], mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < TG_ROW_MAX; ++k) {
            const int j = lane + 64 * k;
            v[k] = j < N2 ? r[j] : -3.0e38f;
            mx = fmaxf(mx, v[k]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < TG_ROW_MAX; ++k) {
            v[k] = lane + 64 * k < N2 ? exp2f((v[k] - mx) * c2) : 0.f;
            sum += v[k];
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        float mn = 3.0e38f;
#pragma unroll
        for (int k = 0; k < TG_ROW_MAX; ++k) {
            if (lane + 64 * k < N2) { const float pr = v[k] * inv; acc[k] += pr; mn = fminf(mn, pr); }
        }
        mn = -wave_max(-mn);
        if (lane == 0) rowmin[(long)l * rows + row] = mn;
    }
    const float w = 1.0f / (float)L;
#pragma unroll
    for (int k = 0; k < TG_ROW_MAX; ++k) {
        const int j = lane + 64 * k;
        if (j < N2) out[row * N2 + j] = acc[k] * w;
    }
}

__global__ __launch_bounds__(256) void mast3r_target_col0_kernel(const float* rowmin, float* out, int L, long rows, int N2) {
    __shared__ float red[4];
    __shared__ float fill;
    float total = 0.f;
    for (int l = 0; l < L; ++l) {
        float mn = 3.0e38f;
        for (long r = threadIdx.x; r < rows; r += 256) mn = fminf(mn, rowmin[(long)l * rows + r]);
        mn = -wave_max(-mn);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mn;
        __syncthreads();
        total += fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
    }
    if (threadIdx.x == 0) fill = total / (float)L;
    __syncthreads();
    for (long r = threadIdx.x; r < rows; r += 256) out[r * N2] = fill;
}

extern "C" int gd_mast3r_attn_target(const float* recip_scores, int L, int B, int N1, int N2, float temperature, float* out,
                                     float* workspace, void* stream) {
    GD_REQUIRE(L > 0 && B > 0 && N1 > 0 && N2 > 0 && temperature > 0.f, "gd_mast3r_attn_target: bad arguments");
    GD_REQUIRE(N2 <= 64 * TG_ROW_MAX, "gd_mast3r_attn_target: at most %d keys per row (got %d)", 64 * TG_ROW_MAX, N2);
    GD_REQUIRE(workspace != nullptr, "gd_mast3r_attn_target: workspace of L*B*N1 floats required");
    const long rows = (long)B * N1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mast3r_target_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, recip_scores, out, workspace, L, rows,
                       N2, 1.0f / temperature);
    hipLaunchKernelGGL(mast3r_target_col0_kernel, dim3(1), dim3(256), 0, s, workspace, out, L, rows, N2);
    GD_LAUNCH_OK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// post_process_depth (utils/functions.py:262-345): the rasterised MASt3R depth map -> closed, hole-filled, median /
// bilateral / guided filtered, outlier-replaced, joint-bilateral filtered map.  Stencils on [P][H][W] maps, one thread
// per pixel.  SOURCE ABSENT — PARITY UNPINNED for the four kornia filters (restated from their published definitions, see
// oracle/gd_oracle.py:post_process_depth); the torch parts follow the source.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ppd_reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// out = max (sign = +1) / min (sign = -1) over the k x k window, cells outside the image ignored (max_pool2d's -inf padding)
__global__ void ppd_pool_kernel(const float* in, float* out, int H, int W, int k, float sign) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i % W, r = k / 2;
    const float* m = in + (long)blockIdx.y * H * W;
    float best = -INFINITY;
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) best = fmaxf(best, sign * m[yy * W + xx]);
        }
    out[(long)blockIdx.y * H * W + i] = sign * best;
}

// one hole-filling pass (:283-310): valid = in >= thr (first pass, thr = 1e-5) or in > 0 (second pass, strict)
__global__ void ppd_fill_kernel(const float* in, float* out, int H, int W, int ks, float thr, int strict) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i % W, r = ks / 2;
    const float* m = in + (long)blockIdx.y * H * W;
    float cnt = 0.f, val = 0.f;
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;          // conv2d zero padding
            const float v = m[yy * W + xx];
            const bool ok = strict ? v > 0.f : v >= thr;
            if (ok) { cnt += 1.f; val += v; }
        }
    const float c = m[i];
    const float valid = (strict ? c > 0.f : c >= thr) ? 1.f : 0.f;
    const float fill = fminf(fmaxf((cnt > 0.f ? 1.f : 0.f) - valid, 0.f), 1.f);
    out[(long)blockIdx.y * H * W + i] = c * valid + val / (cnt + 1e-8f) * fill;
}

// kornia median_blur: zero padding, lower median of the k x k window (k in {3, 5})
__global__ void ppd_median_kernel(const float* in, float* out, int H, int W, int k) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i % W, r = k / 2, n = k * k;
    const float* m = in + (long)blockIdx.y * H * W;
    float v[25];
    int c = 0;
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            const int yy = y + dy, xx = x + dx;
            const float t = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? m[yy * W + xx] : 0.f;
            int j = c++;
            while (j > 0 && v[j - 1] > t) { v[j] = v[j - 1]; --j; }        // insertion sort
            v[j] = t;
        }
    out[(long)blockIdx.y * H * W + i] = v[(n - 1) / 2];
}

// kornia (joint_)bilateral_blur, border 'reflect', single channel: w = gauss_space * exp(-0.5 (g_nb - g_c)^2 / sigma_c^2)
__global__ void ppd_bilateral_kernel(const float* in, const float* guide, float* out, int H, int W, int k, float sigma_color,
                                     float sigma_space) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i % W, r = k / 2;
    const float* m = in + (long)blockIdx.y * H * W;
    const float* g = guide + (long)blockIdx.y * H * W;
    const float gc = g[i], ic = -0.5f / (sigma_color * sigma_color), is = -1.0f / (2.f * sigma_space * sigma_space);
    float num = 0.f, den = 0.f;
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            const int p = ppd_reflect(y + dy, H) * W + ppd_reflect(x + dx, W);
            const float d = g[p] - gc;
            const float w = expf(is * (float)(dy * dy + dx * dx)) * expf(ic * d * d);   // (the 1-D normalisations cancel)
            num += w * m[p]; den += w;
        }
    out[(long)blockIdx.y * H * W + i] = num / den;
}

// guided filter, step 1: a, b from the reflect-padded k x k box means of I, p, I*I, I*p (k may be even: offsets -(k-1)/2 .. k/2)
__global__ void ppd_guided_ab_kernel(const float* I, const float* p, float* a, float* b, int H, int W, int k, float eps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i % W, fr = (k - 1) / 2;
    const float* Im = I + (long)blockIdx.y * H * W;
    const float* pm = p + (long)blockIdx.y * H * W;
    float sI = 0.f, sp = 0.f, sII = 0.f, sIp = 0.f;
    for (int dy = -fr; dy < k - fr; ++dy)
        for (int dx = -fr; dx < k - fr; ++dx) {
            const int q = ppd_reflect(y + dy, H) * W + ppd_reflect(x + dx, W);
            const float vi = Im[q], vp = pm[q];
            sI += vi; sp += vp; sII += vi * vi; sIp += vi * vp;
        }
    const float inv = 1.0f / (float)(k * k);
    const float mI = sI * inv, mp = sp * inv, var = sII * inv - mI * mI, cov = sIp * inv - mI * mp;
    const float av = cov / (var + eps);
    a[(long)blockIdx.y * H * W + i] = av;
    b[(long)blockIdx.y * H * W + i] = mp - av * mI;
}
// step 2: q = mean(a) * I + mean(b)
__global__ void ppd_guided_q_kernel(const float* a, const float* b, const float* I, float* q, int H, int W, int k) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i % W, fr = (k - 1) / 2;
    const long o = (long)blockIdx.y * H * W;
    float sa = 0.f, sb = 0.f;
    for (int dy = -fr; dy < k - fr; ++dy)
        for (int dx = -fr; dx < k - fr; ++dx) {
            const int p = ppd_reflect(y + dy, H) * W + ppd_reflect(x + dx, W);
            sa += a[o + p]; sb += b[o + p];
        }
    const float inv = 1.0f / (float)(k * k);
    q[o + i] = sa * inv * I[o + i] + sb * inv;
}

// 3-sigma outliers of q against its zero-padded k x k local mean / variance are replaced by the median map (:328-335)
__global__ void ppd_outlier_kernel(const float* q, const float* med, float* out, int H, int W, int k) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i % W, r = k / 2;
    const long o = (long)blockIdx.y * H * W;
    float s = 0.f, s2 = 0.f;
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const float v = q[o + yy * W + xx];
            s += v; s2 += v * v;
        }
    const float inv = 1.0f / (float)(k * k);
    const float lm = s * inv, lv = s2 * inv - lm * lm;
    const float ls = sqrtf(fmaxf(lv, 1e-6f));
    const float v = q[o + i];
    out[o + i] = fabsf(v - lm) > 3.0f * ls ? med[o + i] : v;
}

extern "C" size_t gd_post_process_depth_workspace_bytes(int P, int H, int W) { return (size_t)6 * P * H * W * sizeof(float); }

extern "C" int gd_post_process_depth(const float* depth, float* out, int P, int H, int W, int kernel_size, int bilateral_d,
                                     float sigma_color, float sigma_space, int guided_r, float guided_eps, void* workspace,
                                     void* stream) {
    GD_REQUIRE(P > 0 && P <= 65535 && H > 1 && W > 1, "gd_post_process_depth: bad shape P=%d H=%d W=%d", P, H, W);
    GD_REQUIRE((kernel_size == 3 || kernel_size == 5) && bilateral_d % 2 == 1 && bilateral_d >= 1 && bilateral_d / 2 < H &&
                   bilateral_d / 2 < W && guided_r >= 1 && guided_r <= H && guided_r <= W && sigma_color > 0.f && sigma_space > 0.f,
               "gd_post_process_depth: kernel_size must be 3 or 5, bilateral_d odd, guided_r <= min(H, W)");
    GD_REQUIRE(workspace != nullptr, "gd_post_process_depth: workspace required");
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)P * H * W;
    float* w0 = (float*)workspace; float* w1 = w0 + n; float* med = w1 + n; float* bil = med + n; float* a = bil + n; float* b = a + n;
    const dim3 grid(gd_cdiv((long)H * W, 256), P), blk(256);
    hipLaunchKernelGGL(ppd_pool_kernel, grid, blk, 0, s, depth, w0, H, W, kernel_size, 1.0f);       // dilate
    hipLaunchKernelGGL(ppd_pool_kernel, grid, blk, 0, s, w0, w1, H, W, kernel_size, -1.0f);         // erode
    hipLaunchKernelGGL(ppd_fill_kernel, grid, blk, 0, s, w1, w0, H, W, 5, 1e-5f, 0);
    hipLaunchKernelGGL(ppd_fill_kernel, grid, blk, 0, s, w0, w1, H, W, 7, 0.f, 1);
    hipLaunchKernelGGL(ppd_median_kernel, grid, blk, 0, s, w1, med, H, W, kernel_size);
    hipLaunchKernelGGL(ppd_bilateral_kernel, grid, blk, 0, s, med, med, bil, H, W, bilateral_d, sigma_color, sigma_space);
    hipLaunchKernelGGL(ppd_guided_ab_kernel, grid, blk, 0, s, bil, med, a, b, H, W, guided_r, guided_eps);   // guidance = bil, input = med
    hipLaunchKernelGGL(ppd_guided_q_kernel, grid, blk, 0, s, a, b, bil, w0, H, W, guided_r);
    hipLaunchKernelGGL(ppd_outlier_kernel, grid, blk, 0, s, w0, med, w1, H, W, kernel_size);
    hipLaunchKernelGGL(ppd_bilateral_kernel, grid, blk, 0, s, w1, med, out, H, W, bilateral_d, sigma_color * 0.5f, sigma_space);
    GD_LAUNCH_OK();
    return 0;
}
