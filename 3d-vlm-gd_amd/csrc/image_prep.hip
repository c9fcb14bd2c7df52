// Input pipeline on device: what vggt/utils/load_fn.py:12-146 does to a decoded image — PIL's
// bicubic Image.resize on uint8 RGB, ToTensor, centre crop / white padding — and the colour augmentation of
// data_utils/dataset_mast3r_scannetpp.py:185-207 (ColorJitter + GaussianBlur).  Byte / integer work, HBM-bound; the
// resampler is bit-exact against Pillow (fixture G20 written by the reference's own function).
#include "gd_common.h"
#include <vector>

// ------------------------------------------------------------------------------------------------ PIL resampling coefficients (host)
// Pillow src/libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc, restated: support = 2 * max(scale, 1), Keys cubic
// a = -0.5, coefficients normalised in double precision and quantised to 22-bit fixed point.  No FMA contraction: the doubles
// must round exactly as Pillow's build does.
#pragma clang fp contract(off)
#define GD_PIL_BITS 22
static inline double pil_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
static inline void pil_geometry(int in_size, int out_size, double& scale, double& filterscale, double& support, int& ksize) {
    scale = filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    support = 2.0 * filterscale;
    ksize = (int)ceil(support) * 2 + 1;
}
extern "C" int gd_pil_resample_ksize(int in_size, int out_size) {
    GD_REQUIRE(in_size > 0 && out_size > 0, "gd_pil_resample_ksize: sizes must be positive");
    double scale, fs, support; int ksize;
    pil_geometry(in_size, out_size, scale, fs, support, ksize);
    return ksize;
}
extern "C" int gd_pil_resample_coeffs(int in_size, int out_size, int* xmin_out, int* count_out, int* coeffs_out) {
    GD_REQUIRE(in_size > 0 && out_size > 0 && xmin_out && count_out && coeffs_out, "gd_pil_resample_coeffs: bad arguments");
    double scale, filterscale, support; int ksize;
    pil_geometry(in_size, out_size, scale, filterscale, support, ksize);
    const double ss = 1.0 / filterscale;
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = pil_bicubic((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < ksize; ++x) {
            double v = x < xmax ? k[x] : 0.0;
            if (x < xmax && ww != 0.0) v /= ww;
            coeffs_out[(long)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << GD_PIL_BITS)) : (int)(0.5 + v * (1 << GD_PIL_BITS));
        }
        xmin_out[xx] = xmin;
        count_out[xx] = xmax;
    }
    return 0;
}
// (contraction stays off for the device code below as well: the colour kernels are compared value for value with a float32 restatement)

// ------------------------------------------------------------------------------------------------ one resampling pass (device)
// dst[r][xx][c] = clip8((2^21 + sum_x src[r][xmin[xx] + x][c] * k[xx][x]) >> 22) along one axis; the other axis and the
// channels are carried through `rows` x `C` with explicit strides, so the same kernel is the horizontal and the vertical pass.
// One thread per output byte-triple (C <= 4); consecutive threads walk the contiguous axis of dst.
struct ResampleParams {
    const unsigned char* src; unsigned char* dst;
    const int* xmin; const int* count; const int* coeffs;
    int rows, out_size, C, ksize, inner_is_rows;
    long s_row, s_pix, d_row, d_pix;
};
__global__ __launch_bounds__(256) void pil_resample_kernel(ResampleParams p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)p.rows * p.out_size) return;
    int r, xx;
    if (p.inner_is_rows) { r = (int)(idx % p.rows); xx = (int)(idx / p.rows); }    // vertical pass: rows (= image columns) contiguous
    else { xx = (int)(idx % p.out_size); r = (int)(idx / p.out_size); }
    const int x0 = p.xmin[xx], n = p.count[xx];
    const int* k = p.coeffs + (long)xx * p.ksize;
    const unsigned char* s = p.src + r * p.s_row + x0 * p.s_pix;
    int acc[4] = {1 << (GD_PIL_BITS - 1), 1 << (GD_PIL_BITS - 1), 1 << (GD_PIL_BITS - 1), 1 << (GD_PIL_BITS - 1)};
    for (int x = 0; x < n; ++x) {
        const int kv = k[x];
        for (int c = 0; c < p.C; ++c) acc[c] += (int)s[x * p.s_pix + c] * kv;
    }
    unsigned char* d = p.dst + r * p.d_row + xx * p.d_pix;
    for (int c = 0; c < p.C; ++c) {
        const int v = acc[c] >> GD_PIL_BITS;      // arithmetic shift (Pillow's clip8 indexes a table with it)
        d[c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

extern "C" int gd_pil_resize_bicubic_u8(const unsigned char* src, unsigned char* tmp, unsigned char* dst, int H, int W, int C,
                                        int new_h, int new_w, const int* xmin_h, const int* count_h, const int* coeffs_h,
                                        int ksize_h, const int* xmin_v, const int* count_v, const int* coeffs_v, int ksize_v,
                                        void* stream) {
    GD_REQUIRE(C >= 1 && C <= 4 && H > 0 && W > 0 && new_h > 0 && new_w > 0, "gd_pil_resize_bicubic_u8: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const bool need_h = new_w != W, need_v = new_h != H;
    if (!need_h && !need_v) {
        if (hipMemcpyAsync(dst, src, (size_t)H * W * C, hipMemcpyDeviceToDevice, st) != hipSuccess) { gd_set_error("gd_pil_resize_bicubic_u8: copy failed"); return -2; }
        return 0;
    }
    const unsigned char* cur = src;
    if (need_h) {                                  // Pillow: horizontal pass first, uint8 intermediate [H, new_w, C]
        unsigned char* out = need_v ? tmp : dst;
        GD_REQUIRE(out, "gd_pil_resize_bicubic_u8: tmp buffer required for a two-pass resize");
        ResampleParams p{cur, out, xmin_h, count_h, coeffs_h, H, new_w, C, ksize_h, 0, (long)W * C, C, (long)new_w * C, C};
        hipLaunchKernelGGL(pil_resample_kernel, dim3(gd_cdiv((long)H * new_w, 256)), dim3(256), 0, st, p);
        GD_LAUNCH_OK();
        cur = out;
    }
    if (need_v) {                                  // vertical pass: "rows" are the image columns
        ResampleParams p{cur, dst, xmin_v, count_v, coeffs_v, new_w, new_h, C, ksize_v, 1, C, (long)new_w * C, C, (long)new_w * C};
        hipLaunchKernelGGL(pil_resample_kernel, dim3(gd_cdiv((long)new_w * new_h, 256)), dim3(256), 0, st, p);
        GD_LAUNCH_OK();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ ToTensor + crop + pad
// dst[c][y][x] (float32 CHW, one image of the batch) = src[y - pad_top + crop_y0][x - pad_left][c] / 255, `fill` outside
// (load_fn.py:91-111: centre crop of the height, white padding; :121-139: white padding to the batch's common shape).
__global__ __launch_bounds__(256) void u8_to_chw_kernel(const unsigned char* src, float* dst, int h, int w, int C, int crop_y0,
                                                        int crop_h, int pad_top, int pad_left, int H, int W, float fill) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)C * H * W) return;
    const int x = (int)(idx % W), y = (int)((idx / W) % H), c = (int)(idx / ((long)W * H));
    const int sy = y - pad_top, sx = x - pad_left;
    float v = fill;
    if (sy >= 0 && sy < crop_h && sx >= 0 && sx < w) v = (float)src[((long)(sy + crop_y0) * w + sx) * C + c] / 255.0f;
    dst[idx] = v;
}
extern "C" int gd_u8_to_chw_float(const unsigned char* src, float* dst, int h, int w, int C, int crop_y0, int crop_h, int pad_top,
                                  int pad_left, int H, int W, float fill, void* stream) {
    GD_REQUIRE(C >= 1 && h > 0 && w > 0 && H > 0 && W > 0 && crop_y0 >= 0 && crop_h > 0 && crop_y0 + crop_h <= h,
               "gd_u8_to_chw_float: bad geometry");
    GD_REQUIRE(pad_top >= 0 && pad_left >= 0 && pad_top + crop_h <= H && pad_left + w <= W, "gd_u8_to_chw_float: the image does not fit the canvas");
    hipLaunchKernelGGL(u8_to_chw_kernel, dim3(gd_cdiv((long)C * H * W, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, h, w, C,
                       crop_y0, crop_h, pad_top, pad_left, H, W, fill);
    GD_LAUNCH_OK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ colour augmentation
// data_utils/dataset_mast3r_scannetpp.py:185-207: albumentations ColorJitter(brightness 0.2, contrast 0.2, saturation 0.2,
// hue 0.1) + GaussianBlur(blur_limit (3, 7)) on uint8 RGB.  albumentations / OpenCV are third-party dependencies that are
// absent from the reference tree and from the build image: PARITY UNPINNED.  The four jitter operations are restated from their
// published definitions (albumentations.augmentations.functional *_torchvision, uint8 path):
//   brightness: v -> clip(v * f) (truncated);  contrast: v -> clip(v * f + mean_gray * (1 - f)) with the mean of the image's
//   OpenCV grey (R 4899 + G 9617 + B 1868 + 8192) >> 14;  saturation: round(clip(v * f + grey * (1 - f)));
//   hue: RGB -> HSV (H in [0, 180)), H -> (H + 180 f) mod 180, HSV -> RGB,
// applied in the per-image order `order[4]` (a permutation of 0..3 = brightness, contrast, saturation, hue; -1 = skip).
__device__ __forceinline__ int cv_gray(int r, int g, int b) { return (r * 4899 + g * 9617 + b * 1868 + 8192) >> 14; }
__device__ __forceinline__ int clip_u8i(float v) { return (int)fminf(fmaxf(v, 0.f), 255.f); }       // np.clip(...).astype(uint8)
__device__ __forceinline__ void jitter_op(int op, float f, float mean_gray, int& r, int& g, int& b) {
    if (op == 0) { r = clip_u8i(r * f); g = clip_u8i(g * f); b = clip_u8i(b * f); }
    else if (op == 1) { const float o = mean_gray * (1.f - f); r = clip_u8i(r * f + o); g = clip_u8i(g * f + o); b = clip_u8i(b * f + o); }
    else if (op == 2) {
        const float gy = (float)cv_gray(r, g, b) * (1.f - f);
        r = (int)fminf(fmaxf(rintf(r * f + gy), 0.f), 255.f); g = (int)fminf(fmaxf(rintf(g * f + gy), 0.f), 255.f);
        b = (int)fminf(fmaxf(rintf(b * f + gy), 0.f), 255.f);
    } else if (op == 3) {
        const float fr = r, fg = g, fb = b;
        const float v = fmaxf(fr, fmaxf(fg, fb)), mn = fminf(fr, fminf(fg, fb)), d = v - mn;
        float h = 0.f;
        if (d > 0.f) {
            if (v == fr) h = (fg - fb) / d; else if (v == fg) h = 2.f + (fb - fr) / d; else h = 4.f + (fr - fg) / d;
            h *= 30.f;                                   // degrees / 2: H in [0, 180)
            if (h < 0.f) h += 180.f;
        }
        const float s = v > 0.f ? d / v : 0.f;
        int hi8 = (int)rintf(h); if (hi8 >= 180) hi8 -= 180;            // the uint8 HSV image albumentations shifts
        int hq = (int)fmodf((float)hi8 + 180.f * f, 180.f); if (hq < 0) hq += 180;     // np.mod(lut + 180 f, 180).astype(uint8)
        const float s8 = rintf(s * 255.f) / 255.f;
        const float hh = hq / 30.f; const int sector = ((int)hh) % 6; const float fq = hh - floorf(hh);
        const float p = v * (1.f - s8), q = v * (1.f - s8 * fq), t = v * (1.f - s8 * (1.f - fq));
        float R, G, B;
        switch (sector) { case 0: R = v; G = t; B = p; break; case 1: R = q; G = v; B = p; break; case 2: R = p; G = v; B = t; break;
                          case 3: R = p; G = q; B = v; break; case 4: R = t; G = p; B = v; break; default: R = v; G = p; B = q; }
        r = (int)fminf(fmaxf(rintf(R), 0.f), 255.f); g = (int)fminf(fmaxf(rintf(G), 0.f), 255.f); b = (int)fminf(fmaxf(rintf(B), 0.f), 255.f);
    }
}
// pass 0: grey sum of the image as the contrast step will see it (after the operations that precede it); pass 1: everything
__global__ __launch_bounds__(256) void color_jitter_kernel(const unsigned char* src, unsigned char* dst, int HW, const float* factors,
                                                           const int* order, unsigned long long* gray_sum, int pass) {
    const int img = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float* f = factors + img * 4;
    const int* ord = order + img * 4;
    unsigned long long part = 0;
    if (i < HW) {
        const unsigned char* s = src + ((long)img * HW + i) * 3;
        int r = s[0], g = s[1], b = s[2];
        const float mean = pass ? (float)((double)gray_sum[img] / (double)HW) : 0.f;
        bool done = false;
        for (int k = 0; k < 4 && !done; ++k) {
            const int op = ord[k];
            if (op < 0) continue;
            if (op == 1 && !pass) { done = true; break; }
            jitter_op(op, f[op], mean, r, g, b);
        }
        if (!pass) part = (unsigned long long)cv_gray(r, g, b);
        else { unsigned char* d = dst + ((long)img * HW + i) * 3; d[0] = (unsigned char)r; d[1] = (unsigned char)g; d[2] = (unsigned char)b; }
    }
    if (!pass) {
        __shared__ unsigned long long sh[256];
        sh[threadIdx.x] = part;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) atomicAdd(gray_sum + img, sh[0]);
    }
}
extern "C" int gd_color_jitter_u8(const unsigned char* src, unsigned char* dst, int n, int H, int W, const float* factors,
                                  const int* order, unsigned long long* gray_sum_ws, void* stream) {
    GD_REQUIRE(n > 0 && H > 0 && W > 0 && gray_sum_ws, "gd_color_jitter_u8: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(gray_sum_ws, 0, sizeof(unsigned long long) * n, st) != hipSuccess) { gd_set_error("gd_color_jitter_u8: memset failed"); return -2; }
    const dim3 grid(gd_cdiv((long)H * W, 256), n);
    hipLaunchKernelGGL(color_jitter_kernel, grid, dim3(256), 0, st, src, dst, H * W, factors, order, gray_sum_ws, 0);
    GD_LAUNCH_OK();
    hipLaunchKernelGGL(color_jitter_kernel, grid, dim3(256), 0, st, src, dst, H * W, factors, order, gray_sum_ws, 1);
    GD_LAUNCH_OK();
    return 0;
}

// cv2.GaussianBlur(img, (k, k), sigmaX = 0): sigma = 0.3 ((k - 1) / 2 - 1) + 0.8, normalised kernel, BORDER_REFLECT_101,
// separable, result rounded to nearest (ties to even).  ksize[img] in {0 (copy), 3, 5, 7}.  axis 0 = along x, 1 = along y.
__global__ __launch_bounds__(256) void gaussian_blur_pass_kernel(const unsigned char* src, float* tmp, unsigned char* dst, int H, int W,
                                                                 const int* ksize, int axis) {
    const int img = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)H * W * 3) return;
    const int c = (int)(i % 3), x = (int)((i / 3) % W), y = (int)(i / (3L * W));
    const int k = ksize[img];
    const long base = (long)img * H * W * 3;
    if (k <= 1) {
        if (axis == 0) tmp[base + i] = (float)src[base + i]; else dst[base + i] = (unsigned char)tmp[base + i];
        return;
    }
    const float sigma = 0.3f * ((k - 1) * 0.5f - 1.f) + 0.8f;
    const int half = k >> 1;
    float wsum = 0.f, acc = 0.f;
    for (int j = -half; j <= half; ++j) {
        const float w = expf(-(float)(j * j) / (2.f * sigma * sigma));
        int xx = x, yy = y;
        if (axis == 0) { xx = x + j; if (xx < 0) xx = -xx; if (xx >= W) xx = 2 * W - 2 - xx; }
        else { yy = y + j; if (yy < 0) yy = -yy; if (yy >= H) yy = 2 * H - 2 - yy; }
        const long si = base + ((long)yy * W + xx) * 3 + c;
        acc += w * (axis == 0 ? (float)src[si] : tmp[si]);
        wsum += w;
    }
    const float v = acc / wsum;
    if (axis == 0) tmp[base + i] = v; else dst[base + i] = (unsigned char)fminf(fmaxf(rintf(v), 0.f), 255.f);
}
extern "C" int gd_gaussian_blur_u8(const unsigned char* src, float* tmp, unsigned char* dst, int n, int H, int W, const int* ksize,
                                   void* stream) {
    GD_REQUIRE(n > 0 && H > 3 && W > 3 && tmp, "gd_gaussian_blur_u8: bad arguments");
    const dim3 grid(gd_cdiv((long)H * W * 3, 256), n);
    hipLaunchKernelGGL(gaussian_blur_pass_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, tmp, dst, H, W, ksize, 0);
    GD_LAUNCH_OK();
    hipLaunchKernelGGL(gaussian_blur_pass_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, tmp, dst, H, W, ksize, 1);
    GD_LAUNCH_OK();
    return 0;
}
