// Common device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the
// geometric-distillation hot path.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// the public C ABI: every TU sees the prototypes, so a definition that drifts from the header fails to compile
#include "gd_hip.h"
#include "gd_knobs.h"

// compile-time index loop: array indices are constants when the IR is built, so per-thread arrays that live across a persistent
// kernel's (not unrolled) tile loop are promoted to registers instead of scratch
#include <utility>
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ---- host-side error plumbing (definitions in cabi.hip) ----
void gd_set_error(const char* fmt, ...);
#define GD_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            gd_set_error(__VA_ARGS__);        \
            return -1;                        \
        }                                     \
    } while (0)
#define GD_LAUNCH_OK()                                                                   \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) {                                                          \
            gd_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return -2;                                                                   \
        }                                                                                \
    } while (0)

__host__ __device__ static inline int gd_dtype_size(int dt) { return (dt == GD_BF16 || dt == GD_F16) ? 2 : 4; }
static inline int gd_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// bit pattern of a float VALUE (__builtin_bit_cast applied directly to an ext-vector element expression v[k] reads element 0 on ROCm 7.2's clang:
// always go through a by-value float)
__device__ __forceinline__ unsigned gd_f2u(float v) { return __builtin_bit_cast(unsigned, v); }

// ---- scalar conversions ----
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }
template <> __device__ __forceinline__ float to_f32<f16>(f16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }
// fp32 -> fp16 for the tf32h engine's operands: SATURATING at +-65504 (fp16 has 5 exponent bits; see gd_cast_f16) and NaN-PROPAGATING.  Convert first
// (round to nearest even: overflow -> inf, NaN -> NaN), then clamp with gfx950's IEEE-754-2019 minimum / maximum (v_[pk_]minimum3_f16 / v_[pk_]maximum3_f16
// return NaN when an operand is NaN).  v_med3_f32 / v_min / v_max return the OTHER operand for a NaN: the fp32 median-of-three clamp this replaces turned
// NaN into -65504 and hid a poisoned activation from everything downstream (the packed pair form costs the same three instructions per two elements).
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
__device__ __forceinline__ f16 f16_clamp(f16 h) { return __builtin_elementwise_minimum(__builtin_elementwise_maximum(h, (f16)-65504.0f), (f16)65504.0f); }
__device__ __forceinline__ f16x2 f16_sat2(float a, float b) {
    const f16x2 lo = {(f16)-65504.0f, (f16)-65504.0f}, hi = {(f16)65504.0f, (f16)65504.0f};
    const f16x2 h = {(f16)a, (f16)b};
    return __builtin_elementwise_minimum(__builtin_elementwise_maximum(h, lo), hi);
}
__device__ __forceinline__ f16x4 f16_sat4(float a, float b, float c, float d) {
    const f16x2 p = f16_sat2(a, b), q = f16_sat2(c, d);
    return f16x4{p[0], p[1], q[0], q[1]};
}
template <int N> __device__ __forceinline__ void f16_satn(const float (&v)[N], f16* out) {      // N even; out may be an element pointer into an ext-vector's storage copy
    static_assert(N % 2 == 0, "pairs");
#pragma unroll
    for (int k = 0; k < N; k += 2) { const f16x2 p = f16_sat2(v[k], v[k + 1]); out[k] = p[0]; out[k + 1] = p[1]; }
}
__device__ __forceinline__ f16x8 f16_sat8(const float (&v)[8]) {
    const f16x2 a = f16_sat2(v[0], v[1]), b = f16_sat2(v[2], v[3]), c = f16_sat2(v[4], v[5]), d = f16_sat2(v[6], v[7]);
    return f16x8{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}
template <> __device__ __forceinline__ f16 from_f32<f16>(float v) { return f16_clamp((f16)v); }

// load/store one element of a runtime-typed buffer (dt = GD_F32 / GD_BF16)
__device__ __forceinline__ float ld_rt(const void* p, long i, int dt) {
    return dt == GD_BF16 ? (float)((const bf16*)p)[i] : dt == GD_F16 ? (float)((const f16*)p)[i] : ((const float*)p)[i];
}
__device__ __forceinline__ void st_rt(void* p, long i, int dt, float v) {
    if (dt == GD_BF16) ((bf16*)p)[i] = (bf16)v; else if (dt == GD_F16) ((f16*)p)[i] = from_f32<f16>(v); else ((float*)p)[i] = v;
}

// ---- wave-level reductions (64 lanes) on DPP (VALU cross-lane operands; no LDS-crossbar ds_bpermute round trips) ----
template <int CTRL, int RMASK> __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RMASK, 0xF, false));
}
// sum over each row of 16 lanes, result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v += dpp_f32<0x141, 0xF>(v);   // row_half_mirror
    v += dpp_f32<0x140, 0xF>(v);   // row_mirror
    return v;
}
// sum over the 64 lanes, result uniform (via lane 63)
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    v += dpp_f32<0x142, 0xA>(v);   // row_bcast:15 into rows 1, 3
    v += dpp_f32<0x143, 0xC>(v);   // row_bcast:31 into rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact-erf GELU and its derivative (torch.nn.GELU default, approximate='none')
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}

// Branch-free GELU / dGELU sharing ONE exponential: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute, i.e.
// at fp32 round-off level and far below bf16 resolution).  e = exp(-y^2/2) serves both the erf tail and the Gaussian
// density of the derivative.  Used where the epilogue math is a measurable share of the kernel (bf16 GEMM epilogues:
// exact erff cost +32 % on fc1 and +88 % on the dGELU-gated GEMM) and in the N^2 x 128 depth-head evaluations.
__device__ __forceinline__ void gelu_parts(float y, float& Phi, float& e) {
    const float ay = fabsf(y);
    e = __expf(-0.5f * y * y);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * 0.70710678118654752f * ay);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float half_tail = 0.5f * poly * e;          // 0.5 * erfc(|y|/sqrt2)
    Phi = y >= 0.f ? 1.0f - half_tail : half_tail;
}
__device__ __forceinline__ float gelu_fast(float y) { float P, e; gelu_parts(y, P, e); return y * P; }
__device__ __forceinline__ float dgelu_fast(float y) { float P, e; gelu_parts(y, P, e); return P + y * e * 0.39894228040143268f; }

// GELU for bf16 OUTPUTS: Phi(y) ~ sigmoid(y (c1 + c3 y^2 + c5 y^4)) — the 3-coefficient minimax fit of the erf form
// (max |y Phi| error 2.5e-5, max derivative error 1.1e-4 on the whole line; a bf16 result carries 2^-9 = 2e-3 relative), with
// the exponent's log2(e) folded into the coefficients: one v_exp, one v_rcp and 5 FMAs for Phi, and the derivative of THIS
// function costs no second exponential (the erf form needs exp(-y^2/2) on top).  c5 < 0, so the polynomial is evaluated on
// y clamped to [-8, 8] (u(8) = 27.6: sigmoid = 1 to fp32); the product uses the unclamped y.
__device__ __forceinline__ void gelu_sig_parts(float y, float& Phi, float& dPhi) {
    const float yc = __builtin_amdgcn_fmed3f(y, -8.0f, 8.0f), y2 = yc * yc;
    const float w = fmaf(y2, fmaf(y2, 0.0010142630f, -0.10677572f), -2.3011213f);      // -(c1 + c3 y^2 + c5 y^4) * log2(e)
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(yc * w));
    const float up = fmaf(y2, fmaf(y2, -0.0035151680f, 0.22203388f), 1.5950158f);       // du/dy
    Phi = s;
    dPhi = (s - s * s) * up;                                                             // dPhi/dy
}
__device__ __forceinline__ float gelu_sig(float y) { float P, d; gelu_sig_parts(y, P, d); return y * P; }

// ---- 16x16 MFMA tile abstraction ------------------------------------------------
// One "K-chunk" is 64 bytes of a row: 32 bf16 or 16 f32.  A fragment is the 16 bytes
// lane l owns: row (or column) l&15, bytes [16*(l>>4), 16*(l>>4)+16) of the chunk.
//   bf16: one v_mfma_f32_16x16x32_bf16 (lane holds k = 8*(l>>4)+j, j<8).
//   f32 : four v_mfma_f32_16x16x4_f32; MFMA j contracts element j of every lane, i.e.
//         k = 4*(l>>4)+j — the k order inside a chunk is irrelevant as long as the A and
//         B fragments use the same one (exact f32, fmaf-chain numerics).
// C/D layout (both): lane l holds C[row = 4*(l>>4)+r][col = l&15], r = 0..3.
template <typename T> struct Mma;
template <> struct Mma<bf16> {
    static constexpr int KC = 32;
    typedef bf16x8 Frag;
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
// fp16 operands (11-bit significands = TF32's; 2.5 PFLOP/s dense like bf16): the tf32h engine's matrix products
template <> struct Mma<f16> {
    static constexpr int KC = 32;
    typedef f16x8 Frag;
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int KC = 16;
    typedef f32x4 Frag;
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
};

// XCD-aware bijective remap of a linear workgroup id: blocks b and b+8 share an XCD
// (round-robin dispatch), so hand each XCD a contiguous chunk of the tile space.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = orig & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (orig >> 3);
}
