// Memory-bound glue kernels of the student step (gfx950): image prep + patch im2col, token assembly,
// 3x3 im2col / col2im for refine_conv, keypoint bilinear gather / scatter, keypoint depth, patch masks,
// global-norm clip + AdamW on the flat trainable buffer.
#include "gd_common.h"

// ---------------------------------------------------------------------------------------------------
// a0 + patch-embed prologue: bilinear resize (torchvision tensor semantics: align_corners=False, no
// antialias; src/finetune_timm_vggt.py:270,340) -> Normalize(mean,std) (:153) -> im2col for the PxP/stride-P
// patch conv.  col[(b,gy,gx), c*P*P + py*P + px], zero padded to Kp columns.
// ---------------------------------------------------------------------------------------------------
// PC: compile-time patch size (14 / 16: the index arithmetic is ~10 integer divisions per element, by constants they
// become multiplies), 0 = run-time P.
template <typename T, int PC>
__global__ __launch_bounds__(256) void patch_im2col_kernel(const float* img, T* col, int B, int h, int w, int H,
                                                           int W, int Prt, int Kp, int sty, int stx, float m0, float m1,
                                                           float m2, float s0, float s1, float s2) {
    const int P = PC ? PC : Prt;
    // stride == P: the timm PatchEmbed grid; stride < P: overlapping patches (src/evaluate_timm.py:262-266 overrides
    // patch_embed.proj.stride for dense tracking features) — 1 + (H - P) / stride positions per axis
    const int gw = 1 + (W - P) / stx, gh = 1 + (H - P) / sty;
    const long rows = (long)B * gh * gw;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const bool same = h == H && w == W;
    // one block per output row (patch): the patch coordinates are wave-uniform, a thread only splits its column index
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const int gx = row % gw, gy = (row / gw) % gh;
        const long b = row / ((long)gw * gh);
        for (int k = threadIdx.x; k < Kp; k += 256) {
            float v = 0.f;
            if (k < 3 * P * P) {
                const int c = k / (P * P), py = (k / P) % P, px = k % P;
                const int Y = gy * sty + py, X = gx * stx + px;
                const float* src = img + (b * 3 + c) * h * w;
                float pix;
                if (same) {
                    pix = src[(long)Y * w + X];
                } else {
                    float fy = fmaxf(((float)Y + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)X + 0.5f) * sx - 0.5f, 0.f);
                    const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
                    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
                    const float wy = fy - (float)y0, wx = fx - (float)x0;
                    pix = (1.f - wy) * ((1.f - wx) * src[(long)y0 * w + x0] + wx * src[(long)y0 * w + x1]) +
                          wy * ((1.f - wx) * src[(long)y1 * w + x0] + wx * src[(long)y1 * w + x1]);
                }
                const float mu = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
                v = (pix - mu) / sd;
            }
            col[row * Kp + k] = from_f32<T>(v);
        }
    }
}

// tokens[b,0] = cls + pos[0];  tokens[b,1+i] = patch[b,i] + pos[1+i]     (timm _pos_embed)
template <typename T>
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const T* patch, const float* cls, const float* pos,
                                                              T* out, int B, int Np, int D) {
    constexpr int V = 16 / sizeof(T);           // 16 bytes of tokens per thread (D is a multiple of 8)
    const int dv = D / V;
    const long total = (long)B * (Np + 1) * dv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int d = (int)(idx % dv) * V;
        const int n = (idx / dv) % (Np + 1);
        const long b = idx / ((long)dv * (Np + 1));
        float v[V];
        if (n == 0) {
#pragma unroll
            for (int k = 0; k < V; ++k) v[k] = cls[d + k];
        } else {
            const uint4 raw = *(const uint4*)(patch + (b * Np + (n - 1)) * D + d);
            const T* e = (const T*)&raw;
#pragma unroll
            for (int k = 0; k < V; ++k) v[k] = to_f32<T>(e[k]);
        }
        uint4 o;
        T* oe = (T*)&o;
#pragma unroll
        for (int k = 0; k < V; ++k) oe[k] = from_f32<T>(v[k] + pos[(long)n * D + d + k]);
        *(uint4*)(out + (b * (Np + 1) + n) * D + d) = o;
    }
}

// ---------------------------------------------------------------------------------------------------
// refine_conv 3x3 pad 1 (src/finetune_timm_vggt.py:146,325) as im2col + GEMM.  x is the token-major grid
// x[b][y][x][:] = base + b*bstride + (y*gw+x)*D.  col[(b,y,x)][(ky*3+kx)*D + c]  (weights re-laid to match).
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const T* x, long bstride, T* col, int B, int gh, int gw,
                                                        int D) {
    const int vec = 16 / (int)sizeof(T), dv = D / vec;
    const long total = (long)B * gh * gw * 9 * dv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx % dv;
        const int tap = (idx / dv) % 9;
        const long row = idx / ((long)dv * 9);
        const int px = row % gw, py = (row / gw) % gh;
        const long b = row / ((long)gw * gh);
        const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (yy >= 0 && yy < gh && xx >= 0 && xx < gw) v = *(const uint4*)(x + b * bstride + ((long)yy * gw + xx) * D + c * vec);
        *(uint4*)(col + (row * 9 + tap) * D + c * vec) = v;
    }
}

// dx[b][y][x][c] = sum_taps dcol[(b, y-ky+1, x-kx+1)][(ky*3+kx)*D + c]
// 16 bytes of channels per thread; the nine taps are loaded UNCONDITIONALLY from clamped coordinates and zeroed by a
// 0/1 weight (a load under `if (inside)` followed by its conversion compiles to load + s_waitcnt vmcnt(0) inside an
// exec-masked block: nine serialised HBM latencies per element — this kernel ran at 1.6 TB/s that way).
template <typename T>
__global__ __launch_bounds__(256) void col2im3x3_vec_kernel(const T* dcol, T* dx, long bstride, int B, int gh, int gw, int D) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int dv = D / VEC;
    const long total = (long)B * gh * gw * dv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (idx % dv) * VEC;
        const long row = idx / dv;
        const int px = row % gw, py = (row / gw) % gh;
        const long b = row / ((long)gw * gh);
        uint4 v[9];
        float w[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = py - (tap / 3 - 1), xx = px - (tap % 3 - 1);
            w[tap] = (yy >= 0 && yy < gh && xx >= 0 && xx < gw) ? 1.f : 0.f;
            const int yc = min(max(yy, 0), gh - 1), xc = min(max(xx, 0), gw - 1);
            v[tap] = *(const uint4*)(dcol + (((b * gh + yc) * gw + xc) * 9 + tap) * D + c);
        }
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const T* e = (const T*)&v[tap];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = fmaf(w[tap], to_f32<T>(e[k]), acc[k]);
        }
        uint4 o;
        T* oe = (T*)&o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) oe[k] = from_f32<T>(acc[k]);
        *(uint4*)(dx + b * bstride + ((long)py * gw + px) * D + c) = o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void col2im3x3_kernel(const T* dcol, T* dx, long bstride, int B, int gh, int gw,
                                                        int D) {
    const long total = (long)B * gh * gw * D;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx % D;
        const long row = idx / D;
        const int px = row % gw, py = (row / gw) % gh;
        const long b = row / ((long)gw * gh);
        float acc = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = py - (tap / 3 - 1), xx = px - (tap % 3 - 1);
            if (yy >= 0 && yy < gh && xx >= 0 && xx < gw)
                acc += to_f32<T>(dcol[(((b * gh + yy) * gw + xx) * 9 + tap) * D + c]);
        }
        dx[b * bstride + ((long)py * gw + px) * D + c] = from_f32<T>(acc);
    }
}

// ---------------------------------------------------------------------------------------------------
// interpolate_features (utils/functions.py:55-76): F.grid_sample(bilinear, align_corners=True, border) at
// keypoints, on up to 4 token-major grids averaged (get_intermediate_feature, src/finetune_timm_vggt.py:291-296).
// kp [B,Nk,2] pixels (x,y) in the source frame; (kx,ky) = kp * (sx,sy) then a*k+b -> [-1,1] -> grid coords.
// ---------------------------------------------------------------------------------------------------
struct GatherParams {
    const void* grid[4]; int ngrid; long bstride; int grid_dtype;
    const float* kp; float* out; float* dgrid[4]; const float* dout;
    // gd_kp_gather_fwd_ln: the grids are the RAW block outputs, normalised where they are sampled — per-token LayerNorm statistics of every grid
    // (mean, 1 / sigma: [B][tokens per image], indexed like the grid's tokens after the same prefix offset) and the shared affine
    const float* ln_mean[4]; const float* ln_rstd[4]; const float* ln_w; const float* ln_b; long sstride;
    int B, Nk, gh, gw, D, pitch;     // pitch: tokens per grid line in memory (gw for a dense grid)
    float sx, sy, ax, bx, ay, by;
};

__device__ __forceinline__ void gather_coords(const GatherParams& p, long bk, int& x0, int& y0, int& x1, int& y1,
                                              float& wx, float& wy) {
    const float px = p.kp[bk * 2 + 0] * p.sx, py = p.kp[bk * 2 + 1] * p.sy;
    float gx = (p.ax * px + p.bx + 1.f) * 0.5f * (float)(p.gw - 1);
    float gy = (p.ay * py + p.by + 1.f) * 0.5f * (float)(p.gh - 1);
    gx = fminf(fmaxf(gx, 0.f), (float)(p.gw - 1));
    gy = fminf(fmaxf(gy, 0.f), (float)(p.gh - 1));
    x0 = (int)floorf(gx); y0 = (int)floorf(gy);
    wx = gx - (float)x0; wy = gy - (float)y0;
    x1 = min(x0 + 1, p.gw - 1); y1 = min(y0 + 1, p.gh - 1);
}

__global__ __launch_bounds__(256) void kp_gather_fwd_kernel(GatherParams p) {
    const long bk = blockIdx.x;  // one block per (b, keypoint)
    const long b = bk / p.Nk;
    int x0, y0, x1, y1; float wx, wy;
    gather_coords(p, bk, x0, y0, x1, y1, wx, wy);
    const float w00 = (1.f - wx) * (1.f - wy), w01 = wx * (1.f - wy), w10 = (1.f - wx) * wy, w11 = wx * wy;
    const float inv = 1.0f / (float)p.ngrid;
    for (int d = threadIdx.x; d < p.D; d += 256) {
        float acc = 0.f;
        for (int t = 0; t < p.ngrid; ++t) {
            const long base = b * p.bstride + d;
            acc += w00 * ld_rt(p.grid[t], base + ((long)y0 * p.pitch + x0) * p.D, p.grid_dtype) +
                   w01 * ld_rt(p.grid[t], base + ((long)y0 * p.pitch + x1) * p.D, p.grid_dtype) +
                   w10 * ld_rt(p.grid[t], base + ((long)y1 * p.pitch + x0) * p.D, p.grid_dtype) +
                   w11 * ld_rt(p.grid[t], base + ((long)y1 * p.pitch + x1) * p.D, p.grid_dtype);
        }
        p.out[bk * p.D + d] = acc * inv;
    }
}
// the same with 16-byte loads (rows that are 16-byte multiples and aligned): one thread per chunk of V channels; the scalar form
// above moved 2 bytes per load instruction and ran the four-tap gather of the step at 1.9 TB/s out of L2
template <typename T>
__global__ __launch_bounds__(128) void kp_gather_fwd_vec_kernel(GatherParams p) {
    constexpr int V = 16 / sizeof(T);
    const long bk = blockIdx.x;
    const long b = bk / p.Nk;
    int x0, y0, x1, y1; float wx, wy;
    gather_coords(p, bk, x0, y0, x1, y1, wx, wy);
    const float w00 = (1.f - wx) * (1.f - wy), w01 = wx * (1.f - wy), w10 = (1.f - wx) * wy, w11 = wx * wy;
    const float inv = 1.0f / (float)p.ngrid;
    const long o00 = ((long)y0 * p.pitch + x0) * p.D, o01 = ((long)y0 * p.pitch + x1) * p.D,
               o10 = ((long)y1 * p.pitch + x0) * p.D, o11 = ((long)y1 * p.pitch + x1) * p.D;
    for (int ch = threadIdx.x; ch * V < p.D; ch += 128) {
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
        for (int t = 0; t < p.ngrid; ++t) {
            const T* g = (const T*)p.grid[t] + b * p.bstride + ch * V;
            const uint4 v00 = *(const uint4*)(g + o00), v01 = *(const uint4*)(g + o01), v10 = *(const uint4*)(g + o10),
                        v11 = *(const uint4*)(g + o11);
            const T *e00 = (const T*)&v00, *e01 = (const T*)&v01, *e10 = (const T*)&v10, *e11 = (const T*)&v11;
#pragma unroll
            for (int k = 0; k < V; ++k)     // the scalar kernel's order of operations, value for value
                acc[k] += w00 * to_f32<T>(e00[k]) + w01 * to_f32<T>(e01[k]) + w10 * to_f32<T>(e10[k]) + w11 * to_f32<T>(e11[k]);
        }
        float* o = p.out + bk * p.D + ch * V;
#pragma unroll
        for (int k = 0; k < V; k += 4) *(f32x4*)(o + k) = f32x4{acc[k] * inv, acc[k + 1] * inv, acc[k + 2] * inv, acc[k + 3] * inv};
    }
}

// The same gather on RAW grids with the final LayerNorm applied where it is sampled (round 5): out = mean_t sum_nb w_nb ((x_t[nb] - mu_t[nb]) rstd_t[nb]) * gamma + beta
// (the bilinear weights of a keypoint sum to 1, so beta comes through once).  The taps' normed copies — four [M, D] passes of ln_fwd_kernel per step, 94 us each —
// are never written: a tapped block's output is the next block's input, whose LayerNorm forward already took the row statistics.
template <typename T>
__global__ __launch_bounds__(128) void kp_gather_fwd_ln_kernel(GatherParams p) {
    constexpr int V = 16 / sizeof(T);
    const long bk = blockIdx.x;
    const long b = bk / p.Nk;
    int x0, y0, x1, y1; float wx, wy;
    gather_coords(p, bk, x0, y0, x1, y1, wx, wy);
    const float w00 = (1.f - wx) * (1.f - wy), w01 = wx * (1.f - wy), w10 = (1.f - wx) * wy, w11 = wx * wy;
    const float inv = 1.0f / (float)p.ngrid;
    const long t00 = (long)y0 * p.pitch + x0, t01 = (long)y0 * p.pitch + x1, t10 = (long)y1 * p.pitch + x0, t11 = (long)y1 * p.pitch + x1;
    // per grid: the four neighbours' weights folded with their 1 / sigma, and the sum of weight * mu / sigma that comes off every channel
    float a00[4], a01[4], a10[4], a11[4], sub[4];
    for (int t = 0; t < p.ngrid; ++t) {
        const float* mu = p.ln_mean[t] + b * p.sstride;
        const float* rs = p.ln_rstd[t] + b * p.sstride;
        a00[t] = w00 * rs[t00]; a01[t] = w01 * rs[t01]; a10[t] = w10 * rs[t10]; a11[t] = w11 * rs[t11];
        sub[t] = a00[t] * mu[t00] + a01[t] * mu[t01] + a10[t] * mu[t10] + a11[t] * mu[t11];
    }
    for (int ch = threadIdx.x; ch * V < p.D; ch += 128) {
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
        for (int t = 0; t < p.ngrid; ++t) {
            const T* g = (const T*)p.grid[t] + b * p.bstride + ch * V;
            const uint4 v00 = *(const uint4*)(g + t00 * p.D), v01 = *(const uint4*)(g + t01 * p.D), v10 = *(const uint4*)(g + t10 * p.D),
                        v11 = *(const uint4*)(g + t11 * p.D);
            const T *e00 = (const T*)&v00, *e01 = (const T*)&v01, *e10 = (const T*)&v10, *e11 = (const T*)&v11;
#pragma unroll
            for (int k = 0; k < V; ++k)
                acc[k] += (a00[t] * to_f32<T>(e00[k]) + a01[t] * to_f32<T>(e01[k]) + a10[t] * to_f32<T>(e10[k]) + a11[t] * to_f32<T>(e11[k])) - sub[t];
        }
        float* o = p.out + bk * p.D + ch * V;
        const float* gw_ = p.ln_w + ch * V;
        const float* gb_ = p.ln_b + ch * V;
#pragma unroll
        for (int k = 0; k < V; k += 4)
            *(f32x4*)(o + k) = f32x4{fmaf(acc[k] * inv, gw_[k], gb_[k]), fmaf(acc[k + 1] * inv, gw_[k + 1], gb_[k + 1]),
                                     fmaf(acc[k + 2] * inv, gw_[k + 2], gb_[k + 2]), fmaf(acc[k + 3] * inv, gw_[k + 3], gb_[k + 3])};
    }
}

// scatter: dgrid[t] (fp32, batch stride p.bstride elements, pre-zeroed) += w * dout / ngrid
__global__ __launch_bounds__(256) void kp_gather_bwd_kernel(GatherParams p) {
    const long bk = blockIdx.x;
    const long b = bk / p.Nk;
    int x0, y0, x1, y1; float wx, wy;
    gather_coords(p, bk, x0, y0, x1, y1, wx, wy);
    const float inv = 1.0f / (float)p.ngrid;
    const float w00 = (1.f - wx) * (1.f - wy) * inv, w01 = wx * (1.f - wy) * inv, w10 = (1.f - wx) * wy * inv,
                w11 = wx * wy * inv;
    const long gs = p.bstride;
    for (int d = threadIdx.x; d < p.D; d += 256) {
        const float g = p.dout[bk * p.D + d];
        for (int t = 0; t < p.ngrid; ++t) {
            float* dg = p.dgrid[t] + b * gs + d;
            atomicAdd(dg + ((long)y0 * p.pitch + x0) * p.D, w00 * g);
            atomicAdd(dg + ((long)y0 * p.pitch + x1) * p.D, w01 * g);
            atomicAdd(dg + ((long)y1 * p.pitch + x0) * p.D, w10 * g);
            atomicAdd(dg + ((long)y1 * p.pitch + x1) * p.D, w11 * g);
        }
    }
}

// extract_kp_depth (utils/functions.py:348-372): 3x3 replicate-padded mean of depth at integer keypoints
__global__ void kp_depth_kernel(const float* depth, const float* kp, float* out, int B, int Nk, int H, int W) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * Nk) return;
    const long b = i / Nk;
    long lin = (long)(kp[i * 2 + 1] * (float)W + kp[i * 2 + 0]);  // reference: (y*W + x).long()
    lin = lin < 0 ? 0 : (lin >= (long)H * W ? (long)H * W - 1 : lin);  // padded (out-of-image) keypoints stay in bounds
    const int y = (int)(lin / W), x = (int)(lin % W);
    float acc = 0.f;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = min(max(y + dy, 0), H - 1), xx = min(max(x + dx, 0), W - 1);
            acc += depth[(b * H + yy) * W + xx];
        }
    out[i] = acc / 9.0f;
}

// get_patch_mask_from_kp_tensor (utils/functions.py:375-399); mask [B, ph*pw] uint8, pre-zeroed
__global__ void patch_mask_kernel(const float* kp, unsigned char* mask, int B, int Nk, int H, int W, int P) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * Nk) return;
    const long b = i / Nk;
    const float x = kp[i * 2 + 0], y = kp[i * 2 + 1];
    if (x >= 0.f && x < (float)W && y >= 0.f && y < (float)H) {
        const int pw = W / P, ph = H / P;
        const int xi = (int)x / P, yi = (int)y / P;
        if (xi < pw && yi < ph) mask[b * ph * pw + yi * pw + xi] = 1;
    }
}

// ---------------------------------------------------------------------------------------------------
// gradient_clip_val=1.0 (global L2 norm, src/main.py:153) + AdamW (src/finetune_timm_vggt.py:642-648) on the
// flat fp32 trainable buffer.  sumsq: partial[blockIdx] ; adamw reads the total from partial[0..nblk).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, long n, double* partial) {
    double acc = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += (double)g[i] * g[i];
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, long n,
                                                    const double* partial, int nblk, float max_norm, float lr,
                                                    float wd, float b1, float b2, float eps, float bc1, float bc2,
                                                    float gscale, float* norm_out) {
    double tot = 0.0;
    for (int i = 0; i < nblk; ++i) tot += partial[i];
    const float gnorm = (float)sqrt(tot) * gscale;
    float coef = max_norm > 0.f ? fminf(max_norm / (gnorm + 1e-6f), 1.0f) : 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = gnorm;
    coef *= gscale;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * coef;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        pi -= (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
        p[i] = pi;
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* in, TO* out, long n, float scale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        out[i] = from_f32<TO>(to_f32<TI>(in[i]) * scale);
}

// get_feature_cost (src/finetune_timm_mast3r.py:321-337, src/finetune_timm_vggt.py:342-353): mean of the tap
// outputs with the prefix token dropped -> contiguous [B, hw, D];  backward scatters dout/ngrid back.
struct TapMeanParams { const void* grid[4]; void* dgrid[4]; int ngrid; long bstride; int prefix; };
// 16 bytes per thread (8 bf16 / 4 f32): D is a multiple of the vector width, so a vector never straddles the prefix rows
template <typename T>
__global__ __launch_bounds__(256) void tap_mean_fwd_kernel(TapMeanParams p, T* out, int B, int hw, int D) {
    constexpr int V = 16 / sizeof(T);
    const long rowv = (long)hw * D / V, total = (long)B * rowv;
    const float inv = 1.0f / (float)p.ngrid;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long b = idx / rowv, rem = (idx - b * rowv) * V;
        const long src = b * p.bstride + (long)p.prefix * D + rem;
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
        for (int t = 0; t < p.ngrid; ++t) {
            const uint4 raw = *(const uint4*)((const T*)p.grid[t] + src);
            const T* e = (const T*)&raw;
#pragma unroll
            for (int k = 0; k < V; ++k) acc[k] += to_f32<T>(e[k]);
        }
        uint4 o;
        T* oe = (T*)&o;
#pragma unroll
        for (int k = 0; k < V; ++k) oe[k] = from_f32<T>(acc[k] * inv);
        *(uint4*)(out + b * (long)hw * D + rem) = o;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void tap_mean_bwd_kernel(TapMeanParams p, const T* dout, int B, int hw, int D, float scale) {
    constexpr int V = 16 / sizeof(T);
    const long rowv = p.bstride / V, total = (long)B * rowv;   // bstride = (prefix + hw) * D
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long b = idx / rowv, rem = (idx - b * rowv) * V;
        uint4 o = {0u, 0u, 0u, 0u};
        if (rem >= (long)p.prefix * D) {
            const uint4 raw = *(const uint4*)(dout + b * (long)hw * D + rem - (long)p.prefix * D);
            const T* e = (const T*)&raw;
            T* oe = (T*)&o;
#pragma unroll
            for (int k = 0; k < V; ++k) oe[k] = from_f32<T>(to_f32<T>(e[k]) * scale);
        }
        for (int t = 0; t < p.ngrid; ++t) *(uint4*)((T*)p.dgrid[t] + b * p.bstride + rem) = o;
    }
}

static inline int ew_blocks(long total) { long b = (total + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

// ---- refine_conv 3x3 as ONE GEMM over an overlapping-row view (no im2col) ----------------------------------------------
// Grid lines get one zero SEPARATOR column (pitch = gw + 1) and every grid row r = (b, y, x) is stored with its vertical
// neighbours: R[r] = (X[r - pitch], X[r], X[r + pitch]) (3 D wide, zeros outside the image / at separators).  With one zero guard
// row in front, A[r][(dx+1) 3D + (dy+1) D + c] = X[r + dy pitch + dx][c] = Rbuf[r * 3D + k]: an AFFINE view with row stride 3 D and
// K = 9 D whose rows overlap — the horizontal taps are the neighbouring rows of the buffer itself.  The 3x3 conv, its transpose
// (on a stacked dY) and its weight gradient are plain GEMMs on that view; the buffer is 3 x the tokens instead of im2col's 9 x.
// src: token layout (src_pitch = gw, first grid token at src_row0) or a pitched grid (src_pitch = gw + 1); TS f32 | bf16 -> TD.
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void stack3_kernel(const TS* src, TD* dst, int B, int gh, int gw, int D, long src_bstride,
                                                     long src_row0, int src_pitch) {
    constexpr int VEC = 16 / sizeof(TD);
    const int pitch = gw + 1, cpr = D / VEC;                    // chunks per D-wide slot
    const long rows = (long)B * gh * pitch + 2, total = rows * 3 * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % cpr);
        const long t = i / cpr;
        const int slot = (int)(t % 3);
        const long R = t / 3, r = R - 1;
        TD v[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] = from_f32<TD>(0.f);
        if (r >= 0 && r < rows - 2) {
            const int b = (int)(r / ((long)gh * pitch)), rem = (int)(r % ((long)gh * pitch));
            const int y = rem / pitch, x = rem % pitch, yy = y + slot - 1;
            if (x < gw && yy >= 0 && yy < gh) {
                const TS* s = src + (long)b * src_bstride + src_row0 + ((long)yy * src_pitch + x) * D + ch * VEC;
#pragma unroll
                for (int k = 0; k < VEC; ++k) v[k] = from_f32<TD>(to_f32<TS>(s[k]));
            }
        }
        TD* d = dst + (R * 3 + slot) * D + ch * VEC;
#pragma unroll
        for (int k = 0; k < VEC; ++k) d[k] = v[k];
    }
}

// pitched grid [B, gh, gw + 1, D] -> token layout [B, prefix + gh*gw, D] (prefix rows zero): the conv's input gradient
template <typename T>
__global__ __launch_bounds__(256) void unpitch_kernel(const T* src, T* dst, int B, int gh, int gw, int D, int prefix) {
    constexpr int VEC = 16 / sizeof(T);
    const int cpr = D / VEC, Nt = prefix + gh * gw;
    const long total = (long)B * Nt * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % cpr);
        const long t = i / cpr;
        const int b = (int)(t / Nt), tok = (int)(t % Nt);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (tok >= prefix) {
            const int g = tok - prefix, y = g / gw, x = g % gw;
            v = *(const uint4*)(src + (((long)b * gh + y) * (gw + 1) + x) * D + ch * VEC);
        }
        *(uint4*)(dst + t * D + ch * VEC) = v;
    }
}

// ---------------------------------------------------------------------------------------------------
extern "C" int gd_patch_im2col_strided(const float* img, void* col, int B, int h, int w, int H, int W, int P, int stride_y,
                                       int stride_x, int Kp, const float* mean3, const float* std3, int dtype, void* stream) {
    GD_REQUIRE(B > 0 && P > 0 && H >= P && W >= P && stride_y > 0 && stride_x > 0 && Kp >= 3 * P * P,
               "gd_patch_im2col: bad geometry H=%d W=%d P=%d stride=(%d,%d) Kp=%d", H, W, P, stride_y, stride_x, Kp);
    const long rows = (long)B * (1 + (H - P) / stride_y) * (1 + (W - P) / stride_x);
#define GD_PI2C(TT, PCV) hipLaunchKernelGGL((patch_im2col_kernel<TT, PCV>), dim3((unsigned)(rows < 65536 * 4 ? rows : 65536 * 4)), dim3(256), 0, (hipStream_t)stream, img, (TT*)col, B, h, w, H, W, P, Kp, stride_y, stride_x, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2])
    GD_REQUIRE(dtype == GD_F32 || dtype == GD_BF16 || dtype == GD_F16, "gd_patch_im2col: bad output dtype %d", dtype);
    if (dtype == GD_BF16) { if (P == 14) GD_PI2C(bf16, 14); else if (P == 16) GD_PI2C(bf16, 16); else GD_PI2C(bf16, 0); }
    else if (dtype == GD_F16) { if (P == 14) GD_PI2C(f16, 14); else if (P == 16) GD_PI2C(f16, 16); else GD_PI2C(f16, 0); }      // tf32h: the projection's operand directly
    else { if (P == 14) GD_PI2C(float, 14); else if (P == 16) GD_PI2C(float, 16); else GD_PI2C(float, 0); }
#undef GD_PI2C
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_patch_im2col(const float* img, void* col, int B, int h, int w, int H, int W, int P, int Kp,
                               const float* mean3, const float* std3, int dtype, void* stream) {
    GD_REQUIRE(P > 0 && H % P == 0 && W % P == 0, "gd_patch_im2col: bad geometry H=%d W=%d P=%d Kp=%d", H, W, P, Kp);
    return gd_patch_im2col_strided(img, col, B, h, w, H, W, P, P, P, Kp, mean3, std3, dtype, stream);
}

extern "C" int gd_assemble_tokens(const void* patch, const float* cls, const float* pos, void* out, int B, int Np,
                                  int D, int dtype, void* stream) {
    GD_REQUIRE(B > 0 && Np > 0 && D > 0 && D % 8 == 0, "gd_assemble_tokens: bad shape (D must be a multiple of 8)");
    GD_REQUIRE(((uintptr_t)patch & 15) == 0 && ((uintptr_t)out & 15) == 0, "gd_assemble_tokens: patch and out must be 16-byte aligned");
    const long total = (long)B * (Np + 1) * D / (16 / gd_dtype_size(dtype));
    if (dtype == GD_BF16)
        hipLaunchKernelGGL(assemble_tokens_kernel<bf16>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const bf16*)patch, cls, pos, (bf16*)out, B, Np, D);
    else
        hipLaunchKernelGGL(assemble_tokens_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const float*)patch, cls, pos, (float*)out, B, Np, D);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_im2col3x3(const void* x, long bstride, void* col, int B, int gh, int gw, int D, int dtype,
                            void* stream) {
    GD_REQUIRE(B > 0 && gh > 0 && gw > 0 && (D * gd_dtype_size(dtype)) % 16 == 0 && (bstride * gd_dtype_size(dtype)) % 16 == 0 &&
                   ((uintptr_t)x & 15) == 0,
               "gd_im2col3x3: D and bstride must be multiples of 16 bytes, x 16-byte aligned");
    const long total = (long)B * gh * gw * 9 * (D / (16 / gd_dtype_size(dtype)));
    if (dtype == GD_BF16)
        hipLaunchKernelGGL(im2col3x3_kernel<bf16>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, bstride, (bf16*)col, B, gh, gw, D);
    else
        hipLaunchKernelGGL(im2col3x3_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, bstride, (float*)col, B, gh, gw, D);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_col2im3x3(const void* dcol, void* dx, long bstride, int B, int gh, int gw, int D, int dtype,
                            void* stream) {
    GD_REQUIRE(B > 0 && gh > 0 && gw > 0 && D > 0, "gd_col2im3x3: bad shape");
    const int es = gd_dtype_size(dtype);
    if ((D * es) % 16 == 0 && (bstride * es) % 16 == 0 && ((uintptr_t)dcol & 15) == 0 && ((uintptr_t)dx & 15) == 0) {
        const long totalv = (long)B * gh * gw * (D / (16 / es));
        if (dtype == GD_BF16)
            hipLaunchKernelGGL(col2im3x3_vec_kernel<bf16>, dim3(ew_blocks(totalv)), dim3(256), 0, (hipStream_t)stream, (const bf16*)dcol, (bf16*)dx, bstride, B, gh, gw, D);
        else
            hipLaunchKernelGGL(col2im3x3_vec_kernel<float>, dim3(ew_blocks(totalv)), dim3(256), 0, (hipStream_t)stream, (const float*)dcol, (float*)dx, bstride, B, gh, gw, D);
        GD_LAUNCH_OK();
        return 0;
    }
    const long total = (long)B * gh * gw * D;
    if (dtype == GD_BF16)
        hipLaunchKernelGGL(col2im3x3_kernel<bf16>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const bf16*)dcol, (bf16*)dx, bstride, B, gh, gw, D);
    else
        hipLaunchKernelGGL(col2im3x3_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const float*)dcol, (float*)dx, bstride, B, gh, gw, D);
    GD_LAUNCH_OK();
    return 0;
}

static int fill_gather(GatherParams& p, const void* const* grids, int ngrid, long bstride, int grid_dtype,
                       const float* kp, int B, int Nk, int gh, int gw, int D, float sx, float sy, int img_h,
                       int img_w, int patch, int stride, int pitch) {
    GD_REQUIRE(ngrid >= 1 && ngrid <= 4, "kp_gather: 1..4 grids (got %d)", ngrid);
    GD_REQUIRE(B > 0 && Nk > 0 && gh > 0 && gw > 0 && D > 0 && pitch >= gw, "kp_gather: bad shape (pitch %d < gw %d?)", pitch, gw);
    p.pitch = pitch;
    for (int t = 0; t < 4; ++t) p.grid[t] = t < ngrid ? grids[t] : nullptr;
    p.ngrid = ngrid; p.bstride = bstride; p.grid_dtype = grid_dtype; p.kp = kp;
    p.B = B; p.Nk = Nk; p.gh = gh; p.gw = gw; p.D = D; p.sx = sx; p.sy = sy;
    // utils/functions.py:56-65 (python float64 arithmetic, then cast to float32 tensors)
    const double half = patch / 2.0;
    const double last_h = ((img_h - patch) / stride) * stride + half, last_w = ((img_w - patch) / stride) * stride + half;
    p.ay = (float)(2.0 / (last_h - half)); p.ax = (float)(2.0 / (last_w - half));
    p.by = (float)(1.0 - last_h * 2.0 / (last_h - half)); p.bx = (float)(1.0 - last_w * 2.0 / (last_w - half));
    return 0;
}

extern "C" int gd_kp_gather_fwd(const void* const* grids, int ngrid, long bstride, int grid_dtype, const float* kp,
                                float* out, int B, int Nk, int gh, int gw, int D, float sx, float sy, int img_h,
                                int img_w, int patch, int stride, int pitch, void* stream) {
    GatherParams p = {};
    if (fill_gather(p, grids, ngrid, bstride, grid_dtype, kp, B, Nk, gh, gw, D, sx, sy, img_h, img_w, patch, stride, pitch)) return -1;
    p.out = out;
    bool vec = (D * gd_dtype_size(grid_dtype)) % 16 == 0 && (bstride * gd_dtype_size(grid_dtype)) % 16 == 0 && ((uintptr_t)out % 16) == 0 && D % 4 == 0;
    for (int t = 0; t < ngrid; ++t) vec = vec && ((uintptr_t)grids[t] % 16) == 0;
    if (vec && grid_dtype == GD_BF16) hipLaunchKernelGGL(kp_gather_fwd_vec_kernel<bf16>, dim3(B * Nk), dim3(128), 0, (hipStream_t)stream, p);
    else if (vec) hipLaunchKernelGGL(kp_gather_fwd_vec_kernel<float>, dim3(B * Nk), dim3(128), 0, (hipStream_t)stream, p);
    else
    hipLaunchKernelGGL(kp_gather_fwd_kernel, dim3(B * Nk), dim3(256), 0, (hipStream_t)stream, p);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_kp_gather_fwd_ln(const void* const* grids, const float* const* means, const float* const* rstds, int ngrid, long bstride,
                                   long sstride, int grid_dtype, const float* ln_w, const float* ln_b, const float* kp, float* out, int B, int Nk, int gh,
                                   int gw, int D, float sx, float sy, int img_h, int img_w, int patch, int stride, int pitch, void* stream) {
    GatherParams p = {};
    if (fill_gather(p, grids, ngrid, bstride, grid_dtype, kp, B, Nk, gh, gw, D, sx, sy, img_h, img_w, patch, stride, pitch)) return -1;
    GD_REQUIRE(grid_dtype == GD_F32 || grid_dtype == GD_BF16, "gd_kp_gather_fwd_ln: grids are f32 or bf16 (got %d)", grid_dtype);
    GD_REQUIRE(means && rstds && ln_w && ln_b && sstride > 0, "gd_kp_gather_fwd_ln: statistics and affine required");
    bool vec = (D * gd_dtype_size(grid_dtype)) % 16 == 0 && (bstride * gd_dtype_size(grid_dtype)) % 16 == 0 && ((uintptr_t)out % 16) == 0 && D % 4 == 0;
    for (int t = 0; t < ngrid; ++t) {
        GD_REQUIRE(means[t] && rstds[t], "gd_kp_gather_fwd_ln: statistics of grid %d missing", t);
        vec = vec && ((uintptr_t)grids[t] % 16) == 0;
        p.ln_mean[t] = means[t]; p.ln_rstd[t] = rstds[t];
    }
    GD_REQUIRE(vec, "gd_kp_gather_fwd_ln: rows and batch strides must be 16-byte multiples, pointers 16-byte aligned, D %% 4 == 0");
    p.out = out; p.ln_w = ln_w; p.ln_b = ln_b; p.sstride = sstride;
    if (grid_dtype == GD_BF16) hipLaunchKernelGGL(kp_gather_fwd_ln_kernel<bf16>, dim3(B * Nk), dim3(128), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(kp_gather_fwd_ln_kernel<float>, dim3(B * Nk), dim3(128), 0, (hipStream_t)stream, p);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_kp_gather_bwd(float* const* dgrids, int ngrid, long bstride, const float* kp, const float* dout,
                                int B, int Nk, int gh, int gw, int D, float sx, float sy, int img_h, int img_w,
                                int patch, int stride, int pitch, void* stream) {
    GatherParams p = {};
    const void* dummy[4] = {dgrids[0], dgrids[0], dgrids[0], dgrids[0]};
    if (fill_gather(p, dummy, ngrid, bstride, GD_F32, kp, B, Nk, gh, gw, D, sx, sy, img_h, img_w, patch, stride, pitch)) return -1;
    for (int t = 0; t < 4; ++t) p.dgrid[t] = t < ngrid ? dgrids[t] : nullptr;
    p.dout = dout;
    hipLaunchKernelGGL(kp_gather_bwd_kernel, dim3(B * Nk), dim3(256), 0, (hipStream_t)stream, p);
    GD_LAUNCH_OK();
    return 0;
}

// refine_conv evaluated ONLY where it is sampled.  get_feature (src/finetune_timm_vggt.py:304-332) is
// interpolate_features(refine_conv(grid))(kp): a 3x3 convolution followed by a bilinear sample — both linear, so
//   feat[kp] = W . (sum_n w_n patch(pos_n)) + b :   interpolate the 3x3 INPUT patches of the four neighbours, then one GEMM over
// B*Nk rows instead of B*gh*gw (19 200 against 87 616 at 518^2 / 300 keypoints; 6 400 tokens per image in the reference
// geometry).  out[bk][(ky, kx, c)] = sum_{a,b} w_ab * tok[b][(y_a + ky - 1) * pitch + x_b + kx - 1][c], zero outside the grid
// (conv padding 1), neighbours and weights exactly those of kp_gather_fwd (grid_sample, align_corners, border clamp).
// One thread per 16-byte channel chunk: the 4 x 4 token block around the keypoint is read once, nine taps written.
template <typename T, typename TO>
__global__ __launch_bounds__(128) void kp_patch_gather_kernel(GatherParams p, TO* out) {
    constexpr int V = 16 / sizeof(T);
    const long bk = blockIdx.x;
    const long b = bk / p.Nk;
    int x0, y0, x1, y1; float wx, wy;
    gather_coords(p, bk, x0, y0, x1, y1, wx, wy);
    // a clamped neighbour (x1 == x0 or y1 == y0) has weight exactly 0 (wx resp. wy = 0), and the block entry read in its place
    // is a finite value (a token or the zero padding): the four taps can always be read at offsets {0, 1}
    const float w00 = (1.f - wx) * (1.f - wy), w01 = wx * (1.f - wy), w10 = (1.f - wx) * wy, w11 = wx * wy;
    const T* g = (const T*)p.grid[0] + b * p.bstride;
    for (int ch = threadIdx.x; ch * V < p.D; ch += 128) {
        float t[4][4][V];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int yy = y0 - 1 + i, xx = x0 - 1 + j;
                if (yy >= 0 && yy < p.gh && xx >= 0 && xx < p.gw) {
                    const uint4 v = *(const uint4*)(g + ((long)yy * p.pitch + xx) * p.D + ch * V);
                    const T* e = (const T*)&v;
#pragma unroll
                    for (int k = 0; k < V; ++k) t[i][j][k] = to_f32<T>(e[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < V; ++k) t[i][j][k] = 0.f;
                }
            }
        TO* o = out + bk * 9 * p.D + ch * V;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                __attribute__((aligned(16))) TO r[V];
#pragma unroll
                for (int k = 0; k < V; ++k)
                    r[k] = from_f32<TO>(w00 * t[ky][kx][k] + w01 * t[ky][kx + 1][k] + w10 * t[ky + 1][kx][k] + w11 * t[ky + 1][kx + 1][k]);
                if constexpr (V * sizeof(TO) == 16) *(uint4*)(o + (ky * 3 + kx) * p.D) = *(const uint4*)r;
                else *(uint2*)(o + (ky * 3 + kx) * p.D) = *(const uint2*)r;      // fp32 grid -> fp16 taps: 4 channels = 8 bytes
            }
    }
}

extern "C" int gd_kp_patch_gather(const void* grid, long bstride, int grid_dtype, const float* kp, void* out, int B, int Nk,
                                  int gh, int gw, int D, float sx, float sy, int img_h, int img_w, int patch, int stride,
                                  int pitch, void* stream) {
    GatherParams p = {};
    const void* one[1] = {grid};
    if (fill_gather(p, one, 1, bstride, grid_dtype, kp, B, Nk, gh, gw, D, sx, sy, img_h, img_w, patch, stride, pitch)) return -1;
    GD_REQUIRE((D * gd_dtype_size(grid_dtype)) % 16 == 0 && ((uintptr_t)grid % 16) == 0 && ((uintptr_t)out % 16) == 0,
               "gd_kp_patch_gather: token rows must be 16-byte multiples and 16-byte aligned");
    if (grid_dtype == GD_BF16)
        hipLaunchKernelGGL((kp_patch_gather_kernel<bf16, bf16>), dim3(B * Nk), dim3(128), 0, (hipStream_t)stream, p, (bf16*)out);
    else
        hipLaunchKernelGGL((kp_patch_gather_kernel<float, float>), dim3(B * Nk), dim3(128), 0, (hipStream_t)stream, p, (float*)out);
    GD_LAUNCH_OK();
    return 0;
}

// tf32h engine: the same gather from an fp32 grid with the [B*Nk, 9*D] taps written as fp16 — the operand of the K = 9D GEMM and of the weight
// gradient — instead of an fp32 block (531 MB at 64 x 300 keypoints, D 768) and a cast pass over it
extern "C" int gd_kp_patch_gather_h(const float* grid, long bstride, const float* kp, void* out16, int B, int Nk, int gh, int gw, int D, float sx,
                                    float sy, int img_h, int img_w, int patch, int stride, int pitch, void* stream) {
    GatherParams p = {};
    const void* one[1] = {grid};
    if (fill_gather(p, one, 1, bstride, GD_F32, kp, B, Nk, gh, gw, D, sx, sy, img_h, img_w, patch, stride, pitch)) return -1;
    GD_REQUIRE(D % 4 == 0 && ((uintptr_t)grid % 16) == 0 && ((uintptr_t)out16 % 16) == 0 && (bstride * 4) % 16 == 0,
               "gd_kp_patch_gather_h: D must be a multiple of 4, grid and out 16-byte aligned");
    hipLaunchKernelGGL((kp_patch_gather_kernel<float, f16>), dim3(B * Nk), dim3(128), 0, (hipStream_t)stream, p, (f16*)out16);
    GD_LAUNCH_OK();
    return 0;
}

// interpolate_features backward WITHOUT atomics: one block per (grid line y, image).  The keypoints of the image whose bilinear
// footprint touches line y are compacted IN KEYPOINT ORDER (wave ballots + a block scan: deterministic) into LDS, then every grid
// position of the line sums its contributions in that fixed order and is written exactly once, in the output dtype, zeros
// included (separator columns, prefix rows) — no zero-fill pass, no float atomics (236 MB of them per gather at the step's size,
// 1.3 TB/s), no fp32 -> bf16 cast pass, and the result is bit-reproducible.
#define KPB_MAXK 1024
template <typename TO>
__global__ __launch_bounds__(256) void kp_gather_bwd_det_kernel(GatherParams p, TO* dgrid, int prefix_rows, float scale) {
    __shared__ int s_kp[KPB_MAXK];
    __shared__ short s_x0[KPB_MAXK], s_x1[KPB_MAXK];
    __shared__ float s_w0[KPB_MAXK], s_w1[KPB_MAXK];      // line weight * (1 - wx), line weight * wx (scale folded in)
    __shared__ int s_wcnt[4], s_total;
    const int y = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_total = 0;
    __syncthreads();
    for (int base = 0; base < p.Nk; base += 256) {
        const int k = base + tid;
        bool m = false;
        int x0 = 0, y0 = 0, x1 = 0, y1 = 0; float wx = 0.f, wy = 0.f, wl = 0.f;
        if (k < p.Nk) {
            gather_coords(p, (long)b * p.Nk + k, x0, y0, x1, y1, wx, wy);
            wl = (y0 == y ? 1.f - wy : 0.f) + (y1 == y ? wy : 0.f);
            m = (y0 == y || y1 == y);
        }
        const unsigned long long bal = __ballot(m);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wcnt[wave] = __popcll(bal);
        __syncthreads();
        int off = s_total;
        for (int w = 0; w < wave; ++w) off += s_wcnt[w];
        if (m) {
            const int e = off + pre;
            s_kp[e] = k; s_x0[e] = (short)x0; s_x1[e] = (short)x1;
            s_w0[e] = wl * (1.f - wx) * scale; s_w1[e] = wl * wx * scale;
        }
        __syncthreads();
        if (tid == 0) s_total += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
    }
    const int n = s_total;
    constexpr int V = 8;                                   // channels per thread (two 16-byte loads of the fp32 keypoint gradient)
    const int nch = p.D / V;
    const int slot = tid >> 7, ch = tid & 127;             // two positions in flight, 128-thread (two-wave) groups
    TO* line = dgrid + (long)b * p.bstride + ((long)prefix_rows + (long)y * p.pitch) * p.D;
    if (y == 0) {                                          // prefix rows (cls token): zero
        for (long i = tid; i < (long)prefix_rows * p.D; i += 256) dgrid[(long)b * p.bstride + i] = from_f32<TO>(0.f);
    }
    for (int x = slot; x < p.pitch; x += 2) {
        if (ch >= nch) continue;
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
        if (x < p.gw) {
            for (int e = 0; e < n; ++e) {
                const float w = (s_x0[e] == x ? s_w0[e] : 0.f) + (s_x1[e] == x ? s_w1[e] : 0.f);
                if (s_x0[e] != x && s_x1[e] != x) continue;          // (group-uniform: x and e are)
                const float* g = p.dout + ((long)b * p.Nk + s_kp[e]) * p.D + ch * V;
                const f32x4 a = *(const f32x4*)g, c = *(const f32x4*)(g + 4);
                acc[0] += w * a[0]; acc[1] += w * a[1]; acc[2] += w * a[2]; acc[3] += w * a[3];
                acc[4] += w * c[0]; acc[5] += w * c[1]; acc[6] += w * c[2]; acc[7] += w * c[3];
            }
        }
        TO r[V];
#pragma unroll
        for (int k = 0; k < V; ++k) r[k] = from_f32<TO>(acc[k]);
        TO* o = line + (long)x * p.D + ch * V;
        if (sizeof(TO) == 2) *(uint4*)o = *(const uint4*)r;
        else { *(uint4*)o = *(const uint4*)r; *(uint4*)(o + 4) = *(const uint4*)(r + 4); }
    }
}

extern "C" int gd_kp_gather_bwd_det(void* dgrid, int out_dtype, long bstride, int prefix_rows, const float* kp, const float* dout,
                                    float scale, int B, int Nk, int gh, int gw, int D, float sx, float sy, int img_h, int img_w,
                                    int patch, int stride, int pitch, void* stream) {
    GatherParams p = {};
    const void* one[1] = {dgrid};
    if (fill_gather(p, one, 1, bstride, GD_F32, kp, B, Nk, gh, gw, D, sx, sy, img_h, img_w, patch, stride, pitch)) return -1;
    GD_REQUIRE(Nk <= KPB_MAXK && D % 8 == 0 && D <= 1024 && prefix_rows >= 0, "gd_kp_gather_bwd_det: needs Nk <= %d, D %% 8 == 0, D <= 1024", KPB_MAXK);
    GD_REQUIRE(((uintptr_t)dgrid % 16) == 0 && ((uintptr_t)dout % 16) == 0 && (bstride * gd_dtype_size(out_dtype)) % 16 == 0,
               "gd_kp_gather_bwd_det: 16-byte alignment");
    p.dout = dout;
    if (out_dtype == GD_BF16)
        hipLaunchKernelGGL((kp_gather_bwd_det_kernel<bf16>), dim3(gh, B), dim3(256), 0, (hipStream_t)stream, p, (bf16*)dgrid, prefix_rows, scale);
    else
        hipLaunchKernelGGL((kp_gather_bwd_det_kernel<float>), dim3(gh, B), dim3(256), 0, (hipStream_t)stream, p, (float*)dgrid, prefix_rows, scale);
    GD_LAUNCH_OK();
    return 0;
}

// Input gradient of gd_kp_patch_gather (the transposed 3x3 convolution restricted to where it is non-zero), WITHOUT atomics:
// U[bk][(ky, kx, c)] = dfeat[bk] . W[:, c, ky, kx] comes from ONE GEMM over B*Nk rows; every token (Y, X) then sums
//   w_ab * U[kp][(Y - y_a + 1, X - x_b + 1, c)]   over the keypoints whose 4 x 4 footprint covers it,
// in keypoint order (compacted per grid line by wave ballots + a block scan, as in kp_gather_bwd_det_kernel), and is written once in
// the output dtype — prefix rows zero.  Replaces scatter + stacked-row buffer + a K = 9D GEMM over the whole grid + un-pitching.
template <typename TU, typename TO>
__global__ __launch_bounds__(512) void kp_patch_bwd_det_kernel(GatherParams p, const TU* U, TO* dtok, int prefix_rows) {
    __shared__ int s_kp[KPB_MAXK];
    __shared__ short s_x0[KPB_MAXK];
    __shared__ signed char s_dx[KPB_MAXK], s_ky0[KPB_MAXK], s_ky1[KPB_MAXK];
    __shared__ float s_wx[KPB_MAXK], s_wa0[KPB_MAXK], s_wa1[KPB_MAXK];
    __shared__ int s_wcnt[4], s_total;
    const int y = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_total = 0;
    __syncthreads();
    // (512 threads: the first 256 compact the line's keypoints, all four 128-thread slots then walk the line's tokens — a token's sum is a chain of
    // dependent global loads, so the slots are there for the loads in flight: 344 -> 270 us at 64 x 300 keypoints, D 768, f32; eight slots: 266)
    for (int base = 0; base < p.Nk; base += 256) {
        const int k = base + tid;
        bool m = false;
        int x0 = 0, y0 = 0, x1 = 0, y1 = 0; float wx = 0.f, wy = 0.f;
        if (k < p.Nk && tid < 256) {
            gather_coords(p, (long)b * p.Nk + k, x0, y0, x1, y1, wx, wy);
            m = y >= y0 - 1 && y <= y0 + 2;            // rows of the 4 x 4 block (the forward reads offsets {0, 1} unconditionally)
        }
        const unsigned long long bal = __ballot(m);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0 && wave < 4) s_wcnt[wave] = __popcll(bal);
        __syncthreads();
        int off = s_total;
        for (int w = 0; w < min(wave, 4); ++w) off += s_wcnt[w];
        if (m) {
            const int e = off + pre;
            const int ky0 = y - y0 + 1, ky1 = y - (y0 + 1) + 1;        // tap row through neighbour a = 0 / a = 1 (rows y0, y0 + 1)
            s_kp[e] = k; s_x0[e] = (short)x0; s_dx[e] = 1;
            s_wx[e] = wx;
            s_wa0[e] = (ky0 >= 0 && ky0 <= 2) ? 1.f - wy : 0.f; s_ky0[e] = (signed char)ky0;
            s_wa1[e] = (ky1 >= 0 && ky1 <= 2) ? wy : 0.f;       s_ky1[e] = (signed char)ky1;
        }
        __syncthreads();
        if (tid == 0) s_total += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
    }
    const int n = s_total;
    constexpr int V = 8, UV = 16 / sizeof(TU);
    const int nch = p.D / V;
    const int slot = tid >> 7, ch = tid & 127;
    TO* line = dtok + (long)b * p.bstride + ((long)prefix_rows + (long)y * p.gw) * p.D;
    if (y == 0) {
        for (long i = tid; i < (long)prefix_rows * p.D; i += 512) dtok[(long)b * p.bstride + i] = from_f32<TO>(0.f);
    }
    const long urow = 9L * p.D;
    for (int x = slot; x < p.gw; x += 4) {
        if (ch >= nch) continue;
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
        for (int e = 0; e < n; ++e) {
            const int x0 = s_x0[e];
            if (x < x0 - 1 || x > x0 + 2) continue;                       // (group-uniform)
            const TU* ub = U + ((long)b * p.Nk + s_kp[e]) * urow + ch * V;
            const float wx = s_wx[e];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float wa = a ? s_wa1[e] : s_wa0[e];
                if (wa == 0.f) continue;
                const int ky = a ? s_ky1[e] : s_ky0[e];
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const int kx = x - (x0 + bb) + 1;
                    const float w = wa * (bb ? wx : 1.f - wx);
                    if (kx < 0 || kx > 2 || w == 0.f) continue;
                    const TU* u = ub + (long)(ky * 3 + kx) * p.D;
#pragma unroll
                    for (int q = 0; q < V / UV; ++q) {
                        const uint4 raw = *(const uint4*)(u + q * UV);
                        const TU* ev = (const TU*)&raw;
#pragma unroll
                        for (int k = 0; k < UV; ++k) acc[q * UV + k] += w * to_f32<TU>(ev[k]);
                    }
                }
            }
        }
        TO r[V];
#pragma unroll
        for (int k = 0; k < V; ++k) r[k] = from_f32<TO>(acc[k]);
        TO* o = line + (long)x * p.D + ch * V;
        *(uint4*)o = *(const uint4*)r;
        if (sizeof(TO) == 4) *(uint4*)(o + 4) = *(const uint4*)(r + 4);
    }
}

extern "C" int gd_kp_patch_bwd_det(const void* U, void* dtok, int dtype, long bstride, int prefix_rows, const float* kp, int B, int Nk,
                                   int gh, int gw, int D, float sx, float sy, int img_h, int img_w, int patch, int stride,
                                   void* stream) {
    GatherParams p = {};
    const void* one[1] = {dtok};
    if (fill_gather(p, one, 1, bstride, GD_F32, kp, B, Nk, gh, gw, D, sx, sy, img_h, img_w, patch, stride, gw)) return -1;
    GD_REQUIRE(Nk <= KPB_MAXK && D % 8 == 0 && D <= 1024 && prefix_rows >= 0, "gd_kp_patch_bwd_det: needs Nk <= %d, D %% 8 == 0, D <= 1024", KPB_MAXK);
    GD_REQUIRE(((uintptr_t)dtok % 16) == 0 && ((uintptr_t)U % 16) == 0 && (bstride * gd_dtype_size(dtype)) % 16 == 0,
               "gd_kp_patch_bwd_det: 16-byte alignment");
    if (dtype == GD_BF16)
        hipLaunchKernelGGL((kp_patch_bwd_det_kernel<bf16, bf16>), dim3(gh, B), dim3(512), 0, (hipStream_t)stream, p, (const bf16*)U, (bf16*)dtok, prefix_rows);
    else
        hipLaunchKernelGGL((kp_patch_bwd_det_kernel<float, float>), dim3(gh, B), dim3(512), 0, (hipStream_t)stream, p, (const float*)U, (float*)dtok, prefix_rows);
    GD_LAUNCH_OK();
    return 0;
}

// refine_conv weight packs for the step, one pass over the fp32 weight W[n][c][ky][kx] (nn.Conv2d layout):
//   wk[n][(ky, kx, c)]            — the GEMM operand of the forward / weight-gradient at the keypoints,
//   wt[c][(kx', ky', n)] = W[n][c][2 - ky'][2 - kx']  — the flipped kernel of the transposed convolution on the stacked-row view,
// both in the engine dtype.  (torch: permute + contiguous + cast, flip + permute + contiguous + cast = six kernels, ~110 us.)
template <typename T>
__global__ __launch_bounds__(256) void conv_weight_pack_kernel(const float* w, T* wk, T* wt, int D) {
    const long total = (long)D * D * 9;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        // i enumerates wk: n, tap = ky*3 + kx, c  (c fastest: coalesced stores of wk; the reads of w gather 9-strided floats)
        const int c = (int)(i % D), tap = (int)((i / D) % 9);
        const long n = i / ((long)D * 9);
        const int ky = tap / 3, kx = tap % 3;
        const float v = w[((n * D + c) * 3 + ky) * 3 + kx];
        wk[i] = from_f32<T>(v);
        // W[n][c][ky][kx] lands in wt[c][(kx', ky', n)] with ky' = 2 - ky, kx' = 2 - kx
        wt[((long)c * 9 + (2 - kx) * 3 + (2 - ky)) * D + n] = from_f32<T>(v);
    }
}
// wu[(ky, kx, c)][n] = wk[n][(ky, kx, c)]: a plain 2-D transpose of the packed weight through 32 x 32 LDS tiles (both sides coalesced;
// reading the conv weight with n fastest is a 27 KB stride: that form of the pack took 80 us)
template <typename T>
__global__ __launch_bounds__(256) void transpose2d_kernel(const T* src, T* dst, int R, int C) {     // src [R][C] -> dst [C][R]
    __shared__ T tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < C) tile[j][tx] = src[(long)(r0 + j) * C + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < R) dst[(long)(c0 + j) * R + r0 + tx] = tile[tx][j];
}
extern "C" int gd_conv_weight_pack(const float* weight, void* wk, void* wt, void* wu, int D, int dtype, void* stream) {
    GD_REQUIRE(D > 0 && weight && wk && wt, "gd_conv_weight_pack: bad arguments");
    const int blocks = gd_cdiv((long)D * D * 9, 256 * 4);
    const dim3 tg(gd_cdiv(9 * D, 32), gd_cdiv(D, 32));
    if (dtype == GD_BF16) {
        hipLaunchKernelGGL(conv_weight_pack_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight, (bf16*)wk, (bf16*)wt, D);
        if (wu) hipLaunchKernelGGL(transpose2d_kernel<bf16>, tg, dim3(256), 0, (hipStream_t)stream, (const bf16*)wk, (bf16*)wu, D, 9 * D);
    } else {
        hipLaunchKernelGGL(conv_weight_pack_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight, (float*)wk, (float*)wt, D);
        if (wu) hipLaunchKernelGGL(transpose2d_kernel<float>, tg, dim3(256), 0, (hipStream_t)stream, (const float*)wk, (float*)wu, D, 9 * D);
    }
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_kp_depth(const float* depth, const float* kp, float* out, int B, int Nk, int H, int W, void* stream) {
    GD_REQUIRE(B > 0 && Nk > 0 && H > 0 && W > 0, "gd_kp_depth: bad shape");
    hipLaunchKernelGGL(kp_depth_kernel, dim3(gd_cdiv((long)B * Nk, 256)), dim3(256), 0, (hipStream_t)stream, depth, kp, out, B, Nk, H, W);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_patch_mask(const float* kp, unsigned char* mask, int B, int Nk, int H, int W, int P, void* stream) {
    GD_REQUIRE(B > 0 && Nk > 0 && P > 0, "gd_patch_mask: bad shape");
    hipLaunchKernelGGL(patch_mask_kernel, dim3(gd_cdiv((long)B * Nk, 256)), dim3(256), 0, (hipStream_t)stream, kp, mask, B, Nk, H, W, P);
    GD_LAUNCH_OK();
    return 0;
}

#define GD_SUMSQ_BLOCKS 256
extern "C" size_t gd_adamw_workspace_bytes(void) { return GD_SUMSQ_BLOCKS * sizeof(double); }

extern "C" int gd_clip_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n,
                                  int step, float lr, float weight_decay, float beta1, float beta2, float eps,
                                  float max_norm, float grad_scale, float* grad_norm_out, void* workspace,
                                  void* stream) {
    GD_REQUIRE(n > 0 && step >= 1, "gd_clip_adamw_step: bad n/step");
    hipStream_t s = (hipStream_t)stream;
    double* partial = (double*)workspace;
    hipLaunchKernelGGL(sumsq_kernel, dim3(GD_SUMSQ_BLOCKS), dim3(256), 0, s, grads, n, partial);
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, n, partial,
                       GD_SUMSQ_BLOCKS, max_norm, lr, weight_decay, beta1, beta2, eps, bc1, bc2, grad_scale, grad_norm_out);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_clip_adamw_ranges(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, int step,
                                    float lr, float weight_decay, float beta1, float beta2, float eps, float max_norm,
                                    float grad_scale, float* grad_norm_out, void* workspace, const long* ranges, int n_ranges,
                                    void* stream) {
    GD_REQUIRE(n > 0 && step >= 1 && n_ranges >= 1 && ranges, "gd_clip_adamw_ranges: bad arguments");
    for (int r = 0; r < n_ranges; ++r)
        GD_REQUIRE(ranges[2 * r] >= 0 && ranges[2 * r] < ranges[2 * r + 1] && ranges[2 * r + 1] <= n && ranges[2 * r] % 4 == 0,
                   "gd_clip_adamw_ranges: range %d = [%ld, %ld) is not a 16-byte aligned sub-range of [0, %ld)", r, ranges[2 * r],
                   ranges[2 * r + 1], n);
    hipStream_t s = (hipStream_t)stream;
    double* partial = (double*)workspace;
    hipLaunchKernelGGL(sumsq_kernel, dim3(GD_SUMSQ_BLOCKS), dim3(256), 0, s, grads, n, partial);     // the norm is global
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    for (int r = 0; r < n_ranges; ++r) {
        const long a = ranges[2 * r], m = ranges[2 * r + 1] - a;
        hipLaunchKernelGGL(adamw_kernel, dim3(ew_blocks(m)), dim3(256), 0, s, params + a, grads + a, exp_avg + a, exp_avg_sq + a, m,
                           partial, GD_SUMSQ_BLOCKS, max_norm, lr, weight_decay, beta1, beta2, eps, bc1, bc2, grad_scale,
                           grad_norm_out);
    }
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_cast(const void* in, void* out, long n, float scale, int in_dtype, int out_dtype, void* stream) {
    GD_REQUIRE(n > 0, "gd_cast: n must be positive");
    hipStream_t s = (hipStream_t)stream;
    dim3 g(ew_blocks(n)), b(256);
    if (in_dtype == GD_F32 && out_dtype == GD_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), g, b, 0, s, (const float*)in, (bf16*)out, n, scale);
    else if (in_dtype == GD_BF16 && out_dtype == GD_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), g, b, 0, s, (const bf16*)in, (float*)out, n, scale);
    else if (in_dtype == GD_F32 && out_dtype == GD_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, s, (const float*)in, (float*)out, n, scale);
    else hipLaunchKernelGGL((cast_kernel<bf16, bf16>), g, b, 0, s, (const bf16*)in, (bf16*)out, n, scale);
    GD_LAUNCH_OK();
    return 0;
}

// The same mean, one WAVE per output row, which also hands out 1 / max(||row||, 1e-12) of the row AS STORED (rounded to T): the
// cost-volume loss normalises exactly these rows (F.normalize, src/finetune_timm_vggt.py:514-515), and taking the norm here saves
// its own pass over the features there (gd_cost_volume_kl_fwd_prenorm).  Fixed summation order: bit-reproducible.
template <typename T>
__global__ __launch_bounds__(256) void tap_mean_norm_fwd_kernel(TapMeanParams p, T* out, float* inv_norm, int B, int hw, int D, f16* out16 = nullptr) {
    constexpr int V = 16 / sizeof(T);
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * hw) return;
    const long b = row / hw, r = row - b * hw;
    const long src = b * p.bstride + ((long)p.prefix + r) * D;
    const float inv = 1.0f / (float)p.ngrid;
    float ss = 0.f;
    for (int v = lane; v < D / V; v += 64) {
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
        for (int t = 0; t < p.ngrid; ++t) {
            const uint4 raw = *(const uint4*)((const T*)p.grid[t] + src + (long)v * V);
            const T* e = (const T*)&raw;
#pragma unroll
            for (int k = 0; k < V; ++k) acc[k] += to_f32<T>(e[k]);
        }
        uint4 o;
        T* oe = (T*)&o;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            oe[k] = from_f32<T>(acc[k] * inv);
            const float x = to_f32<T>(oe[k]);
            ss += x * x;
        }
        *(uint4*)(out + row * D + (long)v * V) = o;
        if (sizeof(T) == 4 && out16) {      // tf32h engine: the fp16 copy the cost-volume products take, from the same pass
            const f16x4 h = f16_sat4(to_f32<T>(oe[0]), to_f32<T>(oe[1]), to_f32<T>(oe[2]), to_f32<T>(oe[3]));
            *(f16x4*)(out16 + row * D + (long)v * 4) = h;
        }
    }
    ss = wave_sum(ss);
    if (lane == 0) inv_norm[row] = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
}

// f32 -> three bf16 planes per row for the 3-term split product (gd_split3): x = hi + lo + O(2^-17 |x|), hi = bf16(x), lo = bf16(x - hi).
// which = 0 (left operand):  out row = [hi | lo | hi];  which = 1 (right operand): out row = [hi | hi | lo]
// so that  out_A . out_W^T = hi_a hi_w + lo_a hi_w + hi_a lo_w  over a contraction of 3 K — everything of a . w except lo_a lo_w.
__global__ __launch_bounds__(256) void split3_kernel(const float* in, bf16* out, long rows, int K, long ld_in, int which) {
    const int kv = K / 8;
    const long total = rows * kv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / kv;
        const int c = (int)(idx - r * kv) * 8;
        const f32x4 a = *(const f32x4*)(in + r * ld_in + c), b = *(const f32x4*)(in + r * ld_in + c + 4);
        const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        bf16x8 hi, lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            hi[k] = (bf16)x[k];
            lo[k] = (bf16)(x[k] - (float)hi[k]);
        }
        bf16* o = out + r * 3L * K + c;
        *(bf16x8*)o = hi;
        *(bf16x8*)(o + K) = which ? hi : lo;
        *(bf16x8*)(o + 2L * K) = which ? lo : hi;
    }
}

extern "C" int gd_split3(const float* in, void* out, long rows, int K, long ld_in, int which, void* stream) {
    GD_REQUIRE(in && out && rows > 0 && K > 0 && K % 8 == 0 && ld_in >= K && ld_in % 4 == 0 && (which == 0 || which == 1),
               "gd_split3: bad arguments (K must be a multiple of 8, ld_in of 4)");
    GD_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0, "gd_split3: in / out must be 16-byte aligned");
    hipLaunchKernelGGL(split3_kernel, dim3(ew_blocks(rows * (K / 8))), dim3(256), 0, (hipStream_t)stream, in, (bf16*)out, rows, K, ld_in, which);
    GD_LAUNCH_OK();
    return 0;
}

// ---- fp16 operands of the tf32h engine: x -> f16(sat(x * scale)); the scale of a GRADIENT tensor is a power of two taken from its own
// maximum on the device (gd_amax_scale), carried to the consuming GEMM as a device scalar (gd_gemm_nt_scaled) — no host round trip.
// range (nullable, 2 x 64 words of partial sums, accumulated): how many results saturated at +-65504 (words 0-63) / fell below fp16's normal range
// 2^-14 with a non-zero input (words 64-127: subnormal — fewer than 11 bits — or flushed to zero).  Counted per thread, one atomic per block and
// counter, and only when non-zero.
__global__ __launch_bounds__(256) void cast_f16_kernel(const float* in, f16* out, long rows, int K, long ld_in, float scale, const float* scale_dev, unsigned* range) {
    const int kv = K / 8;
    const long total = rows * kv;
    const float sc = scale_dev ? scale * *scale_dev : scale;
    unsigned nsat = 0u, nlow = 0u;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / kv;
        const int c = (int)(idx - r * kv) * 8;
        const f32x4 a = *(const f32x4*)(in + r * ld_in + c), b = *(const f32x4*)(in + r * ld_in + c + 4);
        f16x8 h;
        { const float t8[8] = {a[0] * sc, a[1] * sc, a[2] * sc, a[3] * sc, b[0] * sc, b[1] * sc, b[2] * sc, b[3] * sc}; h = f16_sat8(t8); }
        *(f16x8*)(out + r * (long)K + c) = h;
        if (range) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned u = gd_f2u((k < 4 ? a[k] : b[k - 4]) * sc) & 0x7fffffffu;
                nsat += u > 0x477fe000u;
                nlow += (u < 0x38800000u) & (u != 0u);
            }
        }
    }
    if (range) {      // one atomic per block and counter, spread over 64 slots each ([0, 64) saturated, [64, 128) below normal range)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { nsat += (unsigned)__shfl_xor((int)nsat, o, 64); nlow += (unsigned)__shfl_xor((int)nlow, o, 64); }
        __shared__ unsigned red[2][4];
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = nsat; red[1][threadIdx.x >> 6] = nlow; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned ns = red[0][0] + red[0][1] + red[0][2] + red[0][3], nl = red[1][0] + red[1][1] + red[1][2] + red[1][3];
            if (ns) atomicAdd(range + (blockIdx.x & 63), ns);
            if (nl) atomicAdd(range + 64 + (blockIdx.x & 63), nl);
        }
    }
}
// max |in| as a BIT PATTERN (non-negative floats order like their bit patterns, and an Inf / NaN pattern is larger than every finite one: a
// non-finite element survives the reduction — fmaxf would drop a NaN — and gd_scale_from_amax turns it into a NaN scale that poisons every
// consumer, as the f32 / bf16 engines' arithmetic would)
__global__ __launch_bounds__(256) void amax_kernel(const float* in, long rows, int K, long ld_in, unsigned* bits) {
    const int kv = K / 4;
    const long total = rows * kv;
    unsigned m = 0u;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / kv;
        const int c = (int)(idx - r * kv) * 4;
        const f32x4 a = *(const f32x4*)(in + r * ld_in + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) { const unsigned u = gd_f2u(a[k]) & 0x7fffffffu; m = u > m ? u : m; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned w = (unsigned)__shfl_xor((int)m, o, 64); m = w > m ? w : m; }
    __shared__ unsigned sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    // one atomic per block (one per wave of 8192 blocks serialised on the single address: 387 us for 270 MB), spread over the slots
    if (threadIdx.x == 0) {
        const unsigned a = sm[0] > sm[1] ? sm[0] : sm[1], b = sm[2] > sm[3] ? sm[2] : sm[3];
        atomicMax(bits + (blockIdx.x & 255), a > b ? a : b);
    }
}
// slots: 256 words of max-|x| bit patterns (amax_kernel, ln_bwd_kernel); s = the power of two with  target / 2 < amax * s <= target  (1 for an
// all-zero tensor, NaN for a non-finite maximum); out = {s, 1 / s}.  The slots are zeroed for their next use.
__global__ __launch_bounds__(256) void amax_scale_kernel(unsigned* slots, float target, float* out) {
    unsigned m = slots[threadIdx.x];
    slots[threadIdx.x] = 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned w = (unsigned)__shfl_xor((int)m, o, 64); m = w > m ? w : m; }
    __shared__ unsigned sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const unsigned a = sm[0] > sm[1] ? sm[0] : sm[1], b = sm[2] > sm[3] ? sm[2] : sm[3];
    const unsigned mb = a > b ? a : b;
    const float amax = __builtin_bit_cast(float, mb);
    float s = 1.0f;
    if (mb >= 0x7f800000u) s = __builtin_nanf("");
    else {
        if (amax > 0.f) s = exp2f(floorf(log2f(target / amax)));
        s = fminf(fmaxf(s, 1.0f / 16777216.0f / 16777216.0f), 16777216.0f * 16777216.0f * 16777216.0f);  // 2^-48 .. 2^72: s and 1/s stay normal floats
    }
    out[0] = s;
    out[1] = 1.0f / s;
}

extern "C" int gd_cast_f16_ex(const float* in, void* out, long rows, int K, long ld_in, float scale, const float* scale_dev, unsigned* range_counters,
                              void* stream) {
    GD_REQUIRE(in && out && rows > 0 && K > 0 && K % 8 == 0 && ld_in >= K && ld_in % 4 == 0, "gd_cast_f16: bad arguments (K must be a multiple of 8, ld_in of 4)");
    GD_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0, "gd_cast_f16: in / out must be 16-byte aligned");
    hipLaunchKernelGGL(cast_f16_kernel, dim3(ew_blocks(rows * (K / 8))), dim3(256), 0, (hipStream_t)stream, in, (f16*)out, rows, K, ld_in, scale, scale_dev, range_counters);
    GD_LAUNCH_OK();
    return 0;
}
extern "C" int gd_cast_f16(const float* in, void* out, long rows, int K, long ld_in, float scale, const float* scale_dev, void* stream) {
    return gd_cast_f16_ex(in, out, rows, K, ld_in, scale, scale_dev, nullptr, stream);
}

extern "C" int gd_scale_from_amax(unsigned* amax_slots, float target, float* scale2, void* stream) {
    GD_REQUIRE(amax_slots && scale2 && target > 0.f, "gd_scale_from_amax: bad arguments");
    hipLaunchKernelGGL(amax_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, amax_slots, target, scale2);
    GD_LAUNCH_OK();
    return 0;
}

// scale3 = {s, 1/s, unused}; amax_slots: 256 zeroed words (zeroed again on return) — a caller that keeps one such buffer saves the memset
extern "C" int gd_amax_scale(const float* in, long rows, int K, long ld_in, float target, float* scale3, unsigned* amax_slots, void* stream) {
    GD_REQUIRE(in && scale3 && amax_slots && rows > 0 && K > 0 && K % 4 == 0 && ld_in >= K && ld_in % 4 == 0 && target > 0.f, "gd_amax_scale: bad arguments");
    GD_REQUIRE(((uintptr_t)in & 15) == 0, "gd_amax_scale: in must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    { const int nb = ew_blocks(rows * (K / 4));
      hipLaunchKernelGGL(amax_kernel, dim3(nb > 2048 ? 2048 : nb), dim3(256), 0, st, in, rows, K, ld_in, amax_slots); }
    hipLaunchKernelGGL(amax_scale_kernel, dim3(1), dim3(256), 0, st, amax_slots, target, scale3);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_tap_mean_norm_fwd(const void* const* grids, int ngrid, long bstride, int prefix, void* out, float* inv_norm, int B,
                                    int hw, int D, int dtype, void* stream) {
    GD_REQUIRE(ngrid >= 1 && ngrid <= 4 && B > 0 && hw > 0 && D > 0 && D % 8 == 0 && inv_norm != nullptr, "gd_tap_mean_norm_fwd: bad arguments");
    TapMeanParams p = {};
    for (int t = 0; t < ngrid; ++t) p.grid[t] = grids[t];
    p.ngrid = ngrid; p.bstride = bstride; p.prefix = prefix;
    const unsigned blocks = (unsigned)(((long)B * hw + 3) / 4);
    if (dtype == GD_BF16) hipLaunchKernelGGL(tap_mean_norm_fwd_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, (bf16*)out, inv_norm, B, hw, D);
    else hipLaunchKernelGGL(tap_mean_norm_fwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, (float*)out, inv_norm, B, hw, D);
    GD_LAUNCH_OK();
    return 0;
}

// fp32 rows, and their fp16 copy (out16 [B, hw, D]) for the tf32h engine's cost-volume products
extern "C" int gd_tap_mean_norm_fwd_h(const void* const* grids, int ngrid, long bstride, int prefix, float* out, void* out16, float* inv_norm,
                                      int B, int hw, int D, void* stream) {
    GD_REQUIRE(ngrid >= 1 && ngrid <= 4 && B > 0 && hw > 0 && D > 0 && D % 8 == 0 && inv_norm != nullptr && out16 != nullptr, "gd_tap_mean_norm_fwd_h: bad arguments");
    TapMeanParams p = {};
    for (int t = 0; t < ngrid; ++t) p.grid[t] = grids[t];
    p.ngrid = ngrid; p.bstride = bstride; p.prefix = prefix;
    const unsigned blocks = (unsigned)(((long)B * hw + 3) / 4);
    hipLaunchKernelGGL(tap_mean_norm_fwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, out, inv_norm, B, hw, D, (f16*)out16);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_tap_mean_fwd(const void* const* grids, int ngrid, long bstride, int prefix, void* out, int B, int hw,
                               int D, int dtype, void* stream) {
    GD_REQUIRE(ngrid >= 1 && ngrid <= 4 && B > 0 && hw > 0 && D > 0, "gd_tap_mean_fwd: bad arguments");
    TapMeanParams p = {};
    for (int t = 0; t < ngrid; ++t) p.grid[t] = grids[t];
    p.ngrid = ngrid; p.bstride = bstride; p.prefix = prefix;
    GD_REQUIRE(D % 8 == 0, "gd_tap_mean_fwd: D must be a multiple of 8");
    const long total = (long)B * hw * D / (dtype == GD_BF16 ? 8 : 4);
    if (dtype == GD_BF16) hipLaunchKernelGGL(tap_mean_fwd_kernel<bf16>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, p, (bf16*)out, B, hw, D);
    else hipLaunchKernelGGL(tap_mean_fwd_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, p, (float*)out, B, hw, D);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_tap_mean_bwd(void* const* dgrids, int ngrid, int prefix, const void* dout, int B, int hw, int D,
                               float scale, int dtype, void* stream) {
    GD_REQUIRE(ngrid >= 1 && ngrid <= 4 && B > 0 && hw > 0 && D > 0 && D % 8 == 0, "gd_tap_mean_bwd: bad arguments");
    TapMeanParams p = {};
    for (int t = 0; t < ngrid; ++t) p.dgrid[t] = dgrids[t];
    p.ngrid = ngrid; p.bstride = (long)(prefix + hw) * D; p.prefix = prefix;
    const long total = (long)B * p.bstride / (dtype == GD_BF16 ? 8 : 4);
    if (dtype == GD_BF16) hipLaunchKernelGGL(tap_mean_bwd_kernel<bf16>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, p, (const bf16*)dout, B, hw, D, scale);
    else hipLaunchKernelGGL(tap_mean_bwd_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, p, (const float*)dout, B, hw, D, scale);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_stack3_rows(const void* src, void* dst, int B, int gh, int gw, int D, long src_bstride, long src_row0,
                              int src_pitch, int src_dtype, int dst_dtype, void* stream) {
    GD_REQUIRE(B > 0 && gh > 0 && gw > 0 && D > 0 && (src_pitch == gw || src_pitch == gw + 1), "gd_stack3_rows: bad shape / source pitch");
    GD_REQUIRE((D * gd_dtype_size(dst_dtype)) % 16 == 0 && ((uintptr_t)dst & 15) == 0, "gd_stack3_rows: D rows must be multiples of 16 bytes, dst aligned");
    const long total = ((long)B * gh * (gw + 1) + 2) * 3 * (D / (16 / gd_dtype_size(dst_dtype)));
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(ew_blocks(total)), b(256);
    if (src_dtype == GD_BF16 && dst_dtype == GD_BF16)
        hipLaunchKernelGGL((stack3_kernel<bf16, bf16>), g, b, 0, s, (const bf16*)src, (bf16*)dst, B, gh, gw, D, src_bstride, src_row0, src_pitch);
    else if (src_dtype == GD_F32 && dst_dtype == GD_BF16)
        hipLaunchKernelGGL((stack3_kernel<float, bf16>), g, b, 0, s, (const float*)src, (bf16*)dst, B, gh, gw, D, src_bstride, src_row0, src_pitch);
    else if (src_dtype == GD_F32 && dst_dtype == GD_F32)
        hipLaunchKernelGGL((stack3_kernel<float, float>), g, b, 0, s, (const float*)src, (float*)dst, B, gh, gw, D, src_bstride, src_row0, src_pitch);
    else { gd_set_error("gd_stack3_rows: unsupported dtype pair %d -> %d", src_dtype, dst_dtype); return -1; }
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_unpitch_tokens(const void* src, void* dst, int B, int gh, int gw, int D, int prefix, int dtype, void* stream) {
    GD_REQUIRE(B > 0 && gh > 0 && gw > 0 && D > 0 && prefix >= 0 && (D * gd_dtype_size(dtype)) % 16 == 0 &&
                   ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "gd_unpitch_tokens: bad shape / alignment");
    const long total = (long)B * (prefix + gh * gw) * (D / (16 / gd_dtype_size(dtype)));
    const dim3 g(ew_blocks(total)), b(256);
    if (dtype == GD_BF16) hipLaunchKernelGGL(unpitch_kernel<bf16>, g, b, 0, (hipStream_t)stream, (const bf16*)src, (bf16*)dst, B, gh, gw, D, prefix);
    else hipLaunchKernelGGL(unpitch_kernel<float>, g, b, 0, (hipStream_t)stream, (const float*)src, (float*)dst, B, gh, gw, D, prefix);
    GD_LAUNCH_OK();
    return 0;
}
