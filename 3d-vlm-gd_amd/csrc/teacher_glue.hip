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
    auto P34 = [](const float* K, const float* E, float (&M)[12]) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) M[r * 4 + c] = K[r * 3 + 0] * E[c] + K[r * 3 + 1] * E[4 + c] + K[r * 3 + 2] * E[8 + c];
    };
    float Pa[12], Pb[12];
    P34(k2, e2, Pa);   // view 1 points -> view 2 image
    P34(k1, e1, Pb);   // view 2 points -> view 1 image
    auto test = [&](const float* pm, const float* Pm) -> unsigned char {
        const float qx = pm[0] - e1[3], qy = pm[1] - e1[7], qz = pm[2] - e1[11];
        // torch.matmul(pm - t, R.t()):  w_k = sum_j q_j R[k][j]
        const float wx = qx * e1[0] + qy * e1[1] + qz * e1[2];
        const float wy = qx * e1[4] + qy * e1[5] + qz * e1[6];
        const float wz = qx * e1[8] + qy * e1[9] + qz * e1[10];
        const float hx = Pm[0] * wx + Pm[1] * wy + Pm[2] * wz + Pm[3];
        const float hy = Pm[4] * wx + Pm[5] * wy + Pm[6] * wz + Pm[7];
        const float hz = Pm[8] * wx + Pm[9] * wy + Pm[10] * wz + Pm[11];
        const float u = hx / (hz + 1e-8f), v = hy / (hz + 1e-8f);
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
