// Flash-style multi-head self-attention for the student ViT (head_dim 64), forward and backward,
// on 16x16 MFMA tiles in a TRANSPOSED formulation (gfx950):
//
//   forward, per wave 32 queries:   S^T = K Q^T   (keys on C-rows, queries on C-columns = lanes)
//                                   O^T += V^T P^T
//   With queries on the lane axis the softmax statistics (m, l) and the rescale factor are per-lane
//   scalars, the row reductions are in-register plus two cross-lane-group shuffles, and the S^T
//   accumulator tiles are directly the B operand of the second product (k-slot permutation shared
//   with the transposed V tile) — P never goes through LDS.
//
//   backward = two kernels (deterministic, no atomics):
//     attn_bwd_dq  (query owner): recompute S^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T
//     attn_bwd_dkv (key owner)  : S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS
//
// Replaces F.scaled_dot_product_attention inside timm's Attention (restated from
// vggt/layers/attention.py:51-71): softmax((q*scale) k^T) v with scale = head_dim^-0.5.
// qkv is the packed [B, N, 3, H, 64] output of the QKV GEMM; o is [B, N, H*64]; lse is [B, H, N] (natural log).
#include "gd_common.h"
#include <type_traits>
#include <stdlib.h>

#define HD 64

// ---- split-precision path (dtype code GD_F32X3): fp32 tensors in memory, every MFMA operand as a (hi, lo) pair of bf16 fragments,
// hi = bf16(x), lo = bf16(x - hi), and every product as the three bf16 MFMAs  lo_a hi_b + hi_a lo_b + hi_a hi_b  (all of a . b except
// lo_a lo_b: ~4e-6 relative, TF32 ~3e-4) — 3 x 16 MFMA cycles per 32-wide chunk against 8 x 32 for the exact-f32 MFMA.  The same
// kernels, instantiated on the tag type `x3` (a 4-byte element: pointer arithmetic is fp32's); only the traits below differ.
struct x3 { float v; };
struct X3Frag { bf16x8 hi, lo; };
template <typename T> struct IsX3 { static constexpr bool v = false; };
template <> struct IsX3<x3> { static constexpr bool v = true; };
template <> struct Mma<x3> {
    static constexpr int KC = 32;
    typedef X3Frag Frag;
    static __device__ __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, c, 0, 0, 0);      // small terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, c, 0, 0, 0);
    }
};
__device__ __forceinline__ X3Frag x3_split(const float (&x)[8]) {
    X3Frag f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        f.hi[k] = (bf16)x[k];
        f.lo[k] = (bf16)(x[k] - (float)f.hi[k]);
    }
    return f;
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // bare v_exp_f32
// max over the 4 lanes {l, l^16, l^32, l^48} with the gfx950 row-swap instructions (VALU; a ds_bpermute pair costs two
// dependent LDS round trips in the middle of the softmax)
// (The two results of a swap are taken out as SCALARS before they are reinterpreted: __builtin_bit_cast applied to an ext-vector element
// expression a[1] reads element 0 on ROCm 7.2's clang — rounds 1-3 shipped `fmaxf(bit_cast(a[0]), bit_cast(a[1]))`, which compiled to a[0] alone:
// every lane got lane group 0's maximum instead of the query's.  A uniform but arbitrary reference point still gives the right o and lse — which
// is why every bf16 / f32 test passed — but it does not bound p by 2^ATT_THR, and fp16 p overflowed to inf on peaked rows: found by round 4's
// adversarial fp16 cases, tests/test_gpu_attention.py::test_attention_reference_point_moves[float16].)
__device__ __forceinline__ float quad_rows_max(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned a0 = a[0], a1 = a[1];
    v = fmaxf(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1));
    const unsigned w = __builtin_bit_cast(unsigned, v);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    const unsigned b0 = b[0], b1 = b[1];
    return fmaxf(__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1));
}

// ---- softmax with a LAGGED reference point (forward) ---------------------------------------------------------------
// The score accumulators are started from -m (the MFMA's C operand: a persistent register quartet per query tile, no
// instruction), with q pre-multiplied by scale*log2(e), so a finished accumulator is already  s*c2 - m  and p is ONE
// v_exp per score — the online softmax's subtract / scale FMA is gone, and so is its running row sum: the row sums come
// out of the PV product as one more MFMA per k-chunk against an all-ones A fragment (they are then the sums of exactly
// the bf16-rounded p that multiply V).  m is a reference point, not the exact running maximum: the first tile sets it to
// the tile's row maximum, later tiles only RAISE it, and only when a score exceeds it by more than ATT_THR (p <= 2^THR
// otherwise): a wave-uniform branch that is almost never taken after the first tiles.  Any reference point gives the same
// o = sum p v / sum p and lse = m + log2 sum p; the first-tile rule keeps sum p >= 1, so nothing can underflow to 0 / 0.
#ifndef ATT_THR
#define ATT_THR 8.0f
#endif
// the two 16-bit element types share every layout: bf16 (bf16 engine) and fp16 (tf32h engine: TF32's significand; conversions saturate)
template <typename T> struct V16;
template <> struct V16<bf16> { typedef bf16x8 T8; typedef bf16x4 T4; };
template <> struct V16<f16> { typedef f16x8 T8; typedef f16x4 T4; };
template <typename T> __device__ __forceinline__ typename Mma<T>::Frag frag_scale(typename Mma<T>::Frag f, float a);
template <typename T> __device__ __forceinline__ typename V16<T>::T8 frag_scale16(typename V16<T>::T8 f, float a) {
    typename V16<T>::T8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = from_f32<T>((float)f[k] * a);
    return o;
}
template <> __device__ __forceinline__ bf16x8 frag_scale<bf16>(bf16x8 f, float a) { return frag_scale16<bf16>(f, a); }
template <> __device__ __forceinline__ f16x8 frag_scale<f16>(f16x8 f, float a) { return frag_scale16<f16>(f, a); }
template <> __device__ __forceinline__ f32x4 frag_scale<float>(f32x4 f, float a) { return f * a; }
template <> __device__ __forceinline__ X3Frag frag_scale<x3>(X3Frag f, float a) {      // scale the fp32 value, split again
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = ((float)f.hi[k] + (float)f.lo[k]) * a;
    return x3_split(x);
}
template <typename T> __device__ __forceinline__ typename Mma<T>::Frag frag_ones();
template <> __device__ __forceinline__ bf16x8 frag_ones<bf16>() {
    const bf16 o = (bf16)1.0f;
    return bf16x8{o, o, o, o, o, o, o, o};
}
template <> __device__ __forceinline__ f16x8 frag_ones<f16>() {
    const f16 o = (f16)1.0f;
    return f16x8{o, o, o, o, o, o, o, o};
}
template <> __device__ __forceinline__ f32x4 frag_ones<float>() { return f32x4{1.f, 1.f, 1.f, 1.f}; }
template <> __device__ __forceinline__ X3Frag frag_ones<x3>() {
    X3Frag f = {};
    f.hi = frag_ones<bf16>();
    return f;
}

// s[qt][kt] hold s*c2 - m of a 64-key tile (TAIL: keys >= N get -1e30).  Updates m / negm (and rescales o, l) when the
// tile's maximum moved the reference point, then turns the scores into p in place.
template <bool TAIL>
__device__ __forceinline__ void softmax_lagged(f32x4 (&s)[2][4], float (&m)[2], f32x4 (&negm)[2], f32x4 (&oacc)[4][2],
                                               f32x4 (&lacc)[2], bool first, int k0, int g, int N) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float tmax = -1e30f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (TAIL && k0 + kt * 16 + g * 4 + r >= N) s[qt][kt][r] = -1e30f;
                tmax = fmaxf(tmax, s[qt][kt][r]);
            }
        // the cross-lane maximum is only needed when SOME lane of the wave sees a score above the threshold (if no lane's own 16 scores exceed
        // it, no query's 64 do): the steady state pays one compare and a wave-uniform branch, not the two lane exchanges
        if (first || __any(tmax > ATT_THR)) {
            tmax = quad_rows_max(tmax);
            const bool need = first || tmax > ATT_THR;
            const float d = need ? tmax : 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[qt][kt][r] -= d;
            m[qt] += d;
            negm[qt] = f32x4{-m[qt], -m[qt], -m[qt], -m[qt]};
            if (!first) {                                  // (first tile: o and l are still zero, and d may be negative)
                const float alpha = fast_exp2(-d);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oacc[dt][qt] *= alpha;
                lacc[qt] *= alpha;
            }
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[qt][kt][r] = fast_exp2(s[qt][kt][r]);
    }
}

template <typename T> struct AT;
// LDS tiles of 64 rows.  bf16: rows are 128 bytes UNPADDED and the 16-byte chunk index is XOR-ed with
// sw(row) = ((row >> 1) & 3) << 1.  One image serves both read patterns conflict-free (PMC before: 33-43 % of the LDS cycles
// of these kernels were bank-conflict cycles with padded 144-byte rows / the GEMM's (row >> 1) & 7 swizzle):
//   * ds_read_b128 fragment reads (rows 16 k + c, chunk 4 u + g): the hardware serves lanes {0-3, 12-15, 20-27} together —
//     rows c in {0..3, 12..15} at chunk q and rows {4..11} at chunk q ^ 1; rows of equal parity share a 128-byte bank
//     half and get the XOR values {0, 2, 4, 6} resp. {4, 6, 0, 2}: eight distinct chunks per half;
//   * ds_read_b64_tr_b16 transpose reads (32 lanes = 8 consecutive rows x the chunk PAIR {2 dt, 2 dt + 1}): the four rows of
//     equal parity need four different pairs — XOR by an even number that differs between them, which an odd XOR ((row >> 1) & 7
//     has them) does not give.
struct AT16 {
    static constexpr int NF = 2;        // fragments per 64-wide contraction
    static constexpr int ROWB = 128;    // LDS row: 64 el * 2 B
    static constexpr int CPR = 8;       // 16-byte chunks per 64-element row
    static constexpr int EPC = 8;       // elements per chunk
    static __device__ __forceinline__ int sw(int row) { return ((row >> 1) & 3) << 1; }
};
template <> struct AT<bf16> : AT16 {};
template <> struct AT<f16> : AT16 {};
template <> struct AT<float> {
    static constexpr int NF = 4;
    static constexpr int ROWB = 272;    // 64 el * 4 B + 16 pad, linear
    static constexpr int CPR = 16;
    static constexpr int EPC = 4;
    static __device__ __forceinline__ int sw(int) { return 0; }
};

// x3: an LDS row is 256 bytes = 16 positions of 16 bytes: plane pl (0 hi, 1 lo), chunk q (8 bf16 each) sits at position
// (8 pl + q) ^ (row & 15) — the sixteen rows 16 k + c that a ds_read_b128 lane group reads at one logical chunk land on sixteen
// different positions; row and row + 16 share the swizzle (the transpose reads rely on it).
template <> struct AT<x3> {
    static constexpr int NF = 2;        // (hi, lo) fragment pairs per 64-wide contraction
    static constexpr int ROWB = 256;
    static constexpr int CPR = 16;      // 16-byte chunks per 64-float GLOBAL row
    static constexpr int EPC = 4;
    static __device__ __forceinline__ int sw(int row) { return row & 15; }
};

// four C-layout tiles that span 64 contraction indices (index = 16*tile + 4*g + r) -> B fragment u
template <typename T> __device__ __forceinline__ typename Mma<T>::Frag acc_to_bfrag(const f32x4 (&t)[4], int u);
template <> __device__ __forceinline__ bf16x8 acc_to_bfrag<bf16>(const f32x4 (&t)[4], int u) {
    const f32x4 a = t[2 * u], b = t[2 * u + 1];
    return bf16x8{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
}
template <> __device__ __forceinline__ f16x8 acc_to_bfrag<f16>(const f32x4 (&t)[4], int u) {
    const f32x4 a = t[2 * u], b = t[2 * u + 1];
    // plain conversions (v_cvt_pk_f16_f32; a saturating clamp per element made these issue-port-bound kernels 13-26 % slower): p <= 2^ATT_THR
    // in the forward, p <= 1 and |dS| <= |dP - delta| in the backward, where dP is a 64-term dot product of the SCALED dout (|dout| s <= 8,
    // vit.py) with v — five orders of magnitude below 65504 for any realistic v
    return f16x8{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3], (f16)b[0], (f16)b[1], (f16)b[2], (f16)b[3]};
}
template <> __device__ __forceinline__ f32x4 acc_to_bfrag<float>(const f32x4 (&t)[4], int u) { return t[u]; }
template <> __device__ __forceinline__ X3Frag acc_to_bfrag<x3>(const f32x4 (&t)[4], int u) {
    const f32x4 a = t[2 * u], b = t[2 * u + 1];
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return x3_split(x);
}

// matching A fragment from a transposed LDS tile row (64 contraction indices contiguous)
template <typename T> __device__ __forceinline__ typename Mma<T>::Frag load_tfrag(const char* row, int u, int g);
template <> __device__ __forceinline__ bf16x8 load_tfrag<bf16>(const char* row, int u, int g) {
    const bf16x4 a = *(const bf16x4*)(row + (32 * u + 4 * g) * 2);
    const bf16x4 b = *(const bf16x4*)(row + (32 * u + 16 + 4 * g) * 2);
    return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
template <> __device__ __forceinline__ f16x8 load_tfrag<f16>(const char* row, int u, int g) {
    const f16x4 a = *(const f16x4*)(row + (32 * u + 4 * g) * 2);
    const f16x4 b = *(const f16x4*)(row + (32 * u + 16 + 4 * g) * 2);
    return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
template <> __device__ __forceinline__ f32x4 load_tfrag<float>(const char* row, int u, int g) {
    return *(const f32x4*)(row + (16 * u + 4 * g) * 4);
}

// natural fragment u (16 bytes) of a [row][64] LDS tile row / global row
template <typename T> __device__ __forceinline__ typename Mma<T>::Frag load_nfrag(const char* row, int u, int g) {
    return *(const typename Mma<T>::Frag*)(row + u * 64 + g * 16);
}

// natural fragment u of row `row` of an LDS tile (chunk 4 u + g, swizzled)
template <typename T> __device__ __forceinline__ typename Mma<T>::Frag lds_nfrag(const char* tile, int row, int u, int g) {
    return *(const typename Mma<T>::Frag*)(tile + row * AT<T>::ROWB + (((u * 4 + g) ^ AT<T>::sw(row)) * 16));
}

template <> __device__ __forceinline__ X3Frag load_nfrag<x3>(const char* row, int u, int g) {      // GLOBAL fp32 row: floats 32 u + 8 g .. + 7
    const f32x4 a = *(const f32x4*)(row + (32 * u + 8 * g) * 4), b = *(const f32x4*)(row + (32 * u + 8 * g + 4) * 4);
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return x3_split(x);
}
template <> __device__ __forceinline__ X3Frag lds_nfrag<x3>(const char* tile, int row, int u, int g) {
    const char* r = tile + row * AT<x3>::ROWB;
    X3Frag f;
    f.hi = *(const bf16x8*)(r + (((u * 4 + g) ^ AT<x3>::sw(row)) * 16));
    f.lo = *(const bf16x8*)(r + (((8 + u * 4 + g) ^ AT<x3>::sw(row)) * 16));
    return f;
}

// Tile staging, split T14-style: `tile_load` issues the global loads of a 64 x 64-element tile into registers
// (rows >= nvalid read as zero) and `tile_store` writes them to LDS later — as a natural tile sN[row][64]
// and/or a transposed tile sT[col][row] — so the next tile's HBM/L2 latency hides under the current tile's MFMAs.
template <typename T, int NT = 256> struct TileRegs { uint4 v[AT<T>::CPR * 64 / NT]; };

template <typename T, int NT = 256>
__device__ __forceinline__ void tile_load(TileRegs<T, NT>& r, const char* gbase, long ld_b, int row0, int nvalid) {
    constexpr int CPR = AT<T>::CPR, NCH = CPR * 64 / NT;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = threadIdx.x + NT * i, rr = ch / CPR, cc = ch % CPR;
        r.v[i] = (row0 + rr < nvalid) ? *(const uint4*)(gbase + (long)(row0 + rr) * ld_b + cc * 16) : make_uint4(0, 0, 0, 0);
    }
}
// The same through a buffer resource (backward kernels): rows >= nvalid lie past the resource's last record and read as zero
// in hardware — no per-chunk compare / exec mask / zero-fill and no 64-bit address arithmetic in the tile loop (that was 28 of the
// dQ loop's 76 non-transcendental VALU instructions); the per-thread byte offsets are loop-invariant, the tile adds row0 * ld.
template <typename T, int NT = 256> struct TileSrc {
    __amdgpu_buffer_rsrc_t rs;
    int off[AT<T>::CPR * 64 / NT];
    int ld;
};
template <typename T, int NT = 256>
__device__ __forceinline__ void tile_src_init(TileSrc<T, NT>& src, const char* gbase, long ld_b, int nvalid) {
    constexpr int CPR = AT<T>::CPR, NCH = CPR * 64 / NT;
    src.rs = __builtin_amdgcn_make_buffer_rsrc((void*)gbase, (short)0, (int)((long)(nvalid - 1) * ld_b + 64 * (long)sizeof(T)), 0x00020000);
    src.ld = (int)ld_b;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = threadIdx.x + NT * i, rr = ch / CPR, cc = ch % CPR;
        src.off[i] = rr * (int)ld_b + cc * 16;
    }
}
template <typename T, int NT = 256>
__device__ __forceinline__ void tile_load(TileRegs<T, NT>& r, const TileSrc<T, NT>& src, int row0) {
    constexpr int NCH = AT<T>::CPR * 64 / NT;
    const int base = row0 * src.ld;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        r.v[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(src.rs, src.off[i] + base, 0, 0));
}
template <typename T, bool NAT, bool TRN, int NT = 256>
__device__ __forceinline__ void tile_store(const TileRegs<T, NT>& r, char* sN, char* sT) {
    constexpr int CPR = AT<T>::CPR, EPC = AT<T>::EPC, ROWB = AT<T>::ROWB, NCH = CPR * 64 / NT;
    if constexpr (std::is_same<T, x3>::value) {      // four floats -> four hi + four lo bf16 (8 bytes each): half `cc & 1` of bf16 chunk `cc >> 1`
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = threadIdx.x + NT * i, rr = ch / CPR, cc = ch % CPR;
            const f32x4 x = __builtin_bit_cast(f32x4, r.v[i]);
            bf16x4 hi, lo;
#pragma unroll
            for (int k = 0; k < 4; ++k) { hi[k] = (bf16)x[k]; lo[k] = (bf16)(x[k] - (float)hi[k]); }
            char* row = sN + rr * ROWB + 8 * (cc & 1);
            *(bf16x4*)(row + (((cc >> 1) ^ AT<x3>::sw(rr)) * 16)) = hi;
            *(bf16x4*)(row + (((8 + (cc >> 1)) ^ AT<x3>::sw(rr)) * 16)) = lo;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = threadIdx.x + NT * i, rr = ch / CPR, cc = ch % CPR;
        if (NAT) *(uint4*)(sN + rr * ROWB + ((cc ^ AT<T>::sw(rr)) * 16)) = r.v[i];
        if (TRN) {
            const T* e = (const T*)&r.v[i];
#pragma unroll
            for (int k = 0; k < EPC; ++k) *(T*)(sT + (cc * EPC + k) * ROWB + rr * (int)sizeof(T)) = e[k];
        }
    }
}

// The "transposed operand" A[row = column c of the tile][k-slot = tile row]:
//   bf16: read straight from the NATURAL tile with ds_read_b64_tr_b16 (hardware 4x16 transpose per 16-lane group:
//         lane 4q+p supplies the address of block row q, columns 4p..4p+3; lane i receives column i of the 4 rows)
//   f32 : 16-byte read from an explicitly transposed LDS tile.
template <typename T> struct TOp;
template <typename T> struct TOp16 {
    static constexpr bool kNeedT = false;
    static __device__ __forceinline__ typename V16<T>::T8 load(const char* sN, const char*, int dt, int u, int g, int lane) {
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        const int i = lane & 15, q = i >> 2, p = i & 3;
        const int row = 32 * u + 4 * g + q;                         // (row + 16 has the same swizzle)
        const char* a0 = sN + row * AT<bf16>::ROWB + (((2 * dt + (p >> 1)) ^ AT<bf16>::sw(row)) * 16) + 8 * (p & 1);
        const char* a1 = a0 + 16 * AT<bf16>::ROWB;
        const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
        const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 z = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
        return __builtin_bit_cast(typename V16<T>::T8, z);
    }
};
template <> struct TOp<bf16> : TOp16<bf16> {};
template <> struct TOp<f16> : TOp16<f16> {};
template <> struct TOp<x3> {      // both planes straight from the natural tile, as bf16
    static constexpr bool kNeedT = false;
    static __device__ __forceinline__ bf16x8 plane(const char* sN, int pl, int dt, int u, int g, int lane) {
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const int i = lane & 15, q = i >> 2, p = i & 3;
        const int row = 32 * u + 4 * g + q;                         // (row + 16 has the same swizzle)
        const char* a0 = sN + row * AT<x3>::ROWB + (((8 * pl + 2 * dt + (p >> 1)) ^ AT<x3>::sw(row)) * 16) + 8 * (p & 1);
        const char* a1 = a0 + 16 * AT<x3>::ROWB;
        const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
        const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
        const s16x8 z = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
        return __builtin_bit_cast(bf16x8, z);
    }
    static __device__ __forceinline__ X3Frag load(const char* sN, const char*, int dt, int u, int g, int lane) {
        X3Frag f;
        f.hi = plane(sN, 0, dt, u, g, lane);
        f.lo = plane(sN, 1, dt, u, g, lane);
        return f;
    }
};
template <> struct TOp<float> {
    static constexpr bool kNeedT = true;
    static __device__ __forceinline__ f32x4 load(const char*, const char* sT, int dt, int u, int g, int lane) {
        return load_tfrag<float>(sT + (dt * 16 + (lane & 15)) * AT<float>::ROWB, u, g);
    }
};
#define TSZ(T) (TOp<T>::kNeedT ? 64 * AT<T>::ROWB : 16)

template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4<x3>(x3* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, f32x4 v) {
    *(bf16x4*)p = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
}
template <> __device__ __forceinline__ void store4<f16>(f16* p, f32x4 v) {
    *(f16x4*)p = f16_sat4(v[0], v[1], v[2], v[3]);
}

// ------------------------------------------------------------------------------------------ forward
// (x-block, head, image) of this workgroup.  The x-blocks of one (image, head) all sweep the same K / V (or Q / dO) rows;
// in plain launch order they are dealt round-robin over the eight XCDs, so every XCD's L2 pulls every (image, head)'s
// operands over the fabric (PMC: 2.2-2.5 GB per launch against 0.4-0.5 GB of operands).  The linear launch index is
// re-mapped so that each XCD gets a contiguous run of (image, head) groups, x-block fastest.
__device__ __forceinline__ void attn_block_coords(int& xb, int& h, int& b) {
    const int nx = gridDim.x, ny = gridDim.y;
    const int lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const int r = xcd_remap(lin, nx * ny * gridDim.z);
    xb = r % nx;
    h = (r / nx) % ny;
    b = r / (nx * ny);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const T* qkv, T* o, float* lse, int N, int H, float scale) {
    constexpr int NF = AT<T>::NF, ROWB = AT<T>::ROWB;
    typedef typename Mma<T>::Frag Frag;
    __shared__ __attribute__((aligned(16))) char sK[64 * ROWB];
    __shared__ __attribute__((aligned(16))) char sV[TOp<T>::kNeedT ? 16 : 64 * ROWB];
    __shared__ __attribute__((aligned(16))) char sVt[TSZ(T)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    int xb_, h, b;
    attn_block_coords(xb_, h, b);
    const int q0 = xb_ * 128 + wave * 32;
    const long ld_b = (long)3 * H * HD * sizeof(T);
    const char* base = (const char*)qkv + (long)b * N * ld_b;
    const char* qb = base + (long)(0 * H + h) * HD * sizeof(T);
    const char* kb = base + (long)(1 * H + h) * HD * sizeof(T);
    const char* vb = base + (long)(2 * H + h) * HD * sizeof(T);

    const float c2 = scale * 1.4426950408889634f;
    Frag qf[2][NF];       // q * scale * log2(e): the scores come out of the MFMA in the exp2 domain
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + qt * 16 + c;
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            if (q < N) qf[qt][u] = frag_scale<T>(load_nfrag<T>(qb + (long)q * ld_b, u, g), c2);
            else { Frag z = {}; qf[qt][u] = z; }
        }
    }
    f32x4 oacc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m[2] = {0.f, 0.f};                                  // reference point, log2 units (see softmax_lagged)
    f32x4 negm[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 lacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // row sums (every row of the tile = the sum)
    const Frag ones = frag_ones<T>();

    TileRegs<T> rk, rv;
    tile_load<T>(rk, kb, ld_b, 0, N);
    tile_load<T>(rv, vb, ld_b, 0, N);
    // one 64-key tile; TAIL (compile-time) = the partial last tile, the only one whose keys need masking — as a run-time
    // flag the mask was if-converted into 60 compare/select instructions in EVERY tile of a VALU-bound loop
    auto key_tile = [&](int k0, auto tail_tag) {
        constexpr bool tail = decltype(tail_tag)::value;
        __syncthreads();
        tile_store<T, true, false>(rk, sK, nullptr);
        tile_store<T, !TOp<T>::kNeedT, TOp<T>::kNeedT>(rv, sV, sVt);
        __syncthreads();
        if (k0 + 64 < N) {
            tile_load<T>(rk, kb, ld_b, k0 + 64, N);
            tile_load<T>(rv, vb, ld_b, k0 + 64, N);
        }
        f32x4 s[2][4];  // [qt][key tile]
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            Frag kf[NF];
#pragma unroll
            for (int u = 0; u < NF; ++u) kf[u] = lds_nfrag<T>(sK, kt * 16 + c, u, g);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 a = negm[qt];
#pragma unroll
                for (int u = 0; u < NF; ++u) a = Mma<T>::mma(kf[u], qf[qt][u], a);
                s[qt][kt] = a;
            }
        }
        softmax_lagged<tail>(s, m, negm, oacc, lacc, k0 == 0, k0, g, N);
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            Frag pf[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                pf[qt] = acc_to_bfrag<T>(s[qt], u);
                lacc[qt] = Mma<T>::mma(ones, pf[qt], lacc[qt]);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const Frag vf = TOp<T>::load(sV, sVt, dt, u, g, lane);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) oacc[dt][qt] = Mma<T>::mma(vf, pf[qt], oacc[dt][qt]);
            }
        }
    };
    int k0 = 0;
    for (; k0 + 64 <= N; k0 += 64) key_tile(k0, std::false_type{});
    if (k0 < N) key_tile(k0, std::true_type{});
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + qt * 16 + c;
        if (q >= N) continue;
        const float lsum = lacc[qt][0];
        const float inv = 1.0f / lsum;
        T* orow = o + ((long)b * N + q) * H * HD + h * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4<T>(orow + dt * 16 + g * 4, oacc[dt][qt] * inv);
        if (g == 0) lse[((long)b * H + h) * N + q] = (m[qt] + log2f(lsum)) * 0.6931471805599453f;   // natural log
    }
}

// ------------------------------------------------------------------------------------------ forward, bf16, LDS-DMA
// Same mathematics and register layout as attn_fwd_kernel, but K / V tiles travel global -> LDS by LDS-DMA
// (global_load_lds_dwordx4) into a THREE-slot ring, two tiles ahead, with no staging registers: at N = 1370 one tile of
// prefetch distance (~0.9 us) is shorter than a loaded memory round trip and 31 % of the wave cycles sat in s_waitcnt
// vmcnt.  Rows are 128 bytes unpadded (a DMA piece is 8 rows x 128 B, lane-linear in LDS); the 16-byte chunk index is
// XOR-swizzled with (row >> 1) & 7 on the SOURCE address, as in the GEMM, so the K fragment reads (ds_read_b128) and the
// V transpose reads (ds_read_b64_tr_b16) stay conflict-free.  All LDS reads are inline asm (a ds_read the compiler can see
// gets an `s_waitcnt vmcnt` to the most recent LDS-DMA in front of it); one counted vmcnt wait + one barrier per tile.
// Key rows past N are clamped to row N-1: their scores are masked in the (compile-time) tail tile, so p = 0 for them.
#define ADS_R128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define ADS_TR64(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
// ... with the offset as a compile-time expression (the steady-state loop below: slot * tile size + row-group offset, one address register per lane)
#define ADS_R128I(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))
#define ADS_TR64I(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))
typedef __attribute__((ext_vector_type(2))) unsigned a_u32x2;
// Round 6: the tiles whose two-ahead prefetch is a full tile run in groups of THREE with the ring slot as a compile-time constant — every LDS read is
// base register + immediate (no per-tile address arithmetic on the vector ALU) and the slot / wait-kind / full-tile / rotation tests leave the loop:
// ~30 scalar instructions per tile instead of ~60 (the round-5 loop executed 1.40 scalar per MFMA, profiles/r05_pmc_attention_stall.json; the non-MFMA
// vector count per steady-state tile is what it was, ~80: 32 v_exp, 16 v_cvt_pk, 16 v_max3 + the threshold test, 4 64-bit address adds for the DMA —
// the compiler keeps the per-lane piece pointers as 64-bit pairs whatever the source says).  Bit-identical; 423.0 -> 416.6 us at 64 x 12 x 1370,
// 1016.8 -> 998.7 us at 8 x 12 x 6401, 57.58 -> 57.50 ms per step in one process (profiles/r06_probe_attn_tn.txt).  The tiles the groups do not
// cover (the last two full tiles, up to two more, the partial tail) run the generic loop below.
template <typename T>      // bf16 | f16
__global__ __launch_bounds__(256, 2) void attn_fwd_dma_kernel(const T* qkv, T* o, float* lse, int N, int H, float scale, int rot_on) {
    typedef typename Mma<T>::Frag Frag;
    constexpr int TILE = 64 * 128;                       // one K or V tile: 64 rows x 128 B
    __shared__ __attribute__((aligned(16))) char smem[6 * TILE];   // K slots 0..2 | V slots 0..2
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int xb_, h, b;
    attn_block_coords(xb_, h, b);
    const int q0 = xb_ * 128 + wave * 32;
    const long ld_b = (long)3 * H * HD * 2;
    const char* base = (const char*)qkv + (long)b * N * ld_b;
    const char* qb = base + (long)(0 * H + h) * HD * 2;
    const char* kb = base + (long)(1 * H + h) * HD * 2;
    const char* vb = base + (long)(2 * H + h) * HD * 2;

    const float c2 = scale * 1.4426950408889634f;
    Frag qf[2][2];        // q * scale * log2(e)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + qt * 16 + c;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (q < N) qf[qt][u] = frag_scale<T>(load_nfrag<T>(qb + (long)q * ld_b, u, g), c2);
            else { Frag z = {}; qf[qt][u] = z; }
        }
    }
    // DMA: wave w moves pieces 2w, 2w+1 (8 rows each) of the K tile and of the V tile
    const int prow = lane >> 3, pchunk = lane & 7;
    // per-piece source pointers of tile 0 (loop-invariant); a full tile adds the scalar k0 * ld_b — the clamped form (one 64-bit
    // multiply per piece: 10 VALU instructions, two of them quarter-rate) is only needed for the partial last tile
    const char* kp[2];
    const char* vp_[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + prow;
        const long off = (long)row * ld_b + ((pchunk ^ AT<T>::sw(row)) * 16);
        kp[i] = kb + off;
        vp_[i] = vb + off;
    }
    auto issue = [&](int k0, int slot) {
        if (k0 + 64 <= N) {
            const unsigned long t0 = (unsigned)k0 * (unsigned)ld_b;     // 32-bit scalar product (an image's qkv rows span < 2^31 bytes)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kp[i] + t0),
                                                 (__attribute__((address_space(3))) void*)(smem + slot * TILE + (wave * 2 + i) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vp_[i] + t0),
                                                 (__attribute__((address_space(3))) void*)(smem + (3 + slot) * TILE + (wave * 2 + i) * 1024), 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 8 + prow;
            const long off = (long)min(k0 + row, N - 1) * ld_b + ((pchunk ^ AT<T>::sw(row)) * 16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb + off),
                                             (__attribute__((address_space(3))) void*)(smem + slot * TILE + (wave * 2 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + off),
                                             (__attribute__((address_space(3))) void*)(smem + (3 + slot) * TILE + (wave * 2 + i) * 1024), 16, 0, 0);
        }
    };
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    // K fragment (kt, u): row kt*16 + c, logical chunk 4u + g
    const int swk = AT<T>::sw(c);
    const unsigned ka0 = c * 128 + ((0 + g) ^ swk) * 16, ka1 = c * 128 + ((4 + g) ^ swk) * 16;
    // V transpose fragment (dt, u): rows 32u + 4g + q (+16), logical chunk 2dt + (p>>1), 8-byte half p&1
    const int vq = c >> 2, vp = c & 3, swv = AT<T>::sw(4 * g + vq);
    unsigned va[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) va[dt] = (4 * g + vq) * 128 + (((2 * dt + (vp >> 1)) ^ swv) * 16) + 8 * (vp & 1);

    f32x4 oacc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m[2] = {0.f, 0.f};                                  // reference point, log2 units (see softmax_lagged)
    f32x4 negm[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 lacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const Frag ones = frag_ones<T>();
    const int ntile = (N + 63) / 64, nfull = N / 64;
    // The x-blocks of an (image, head) start together on one XCD and sweep the same K / V tiles at the same pace: in
    // lockstep they all ask one L2 channel for the same lines at the same moment.  Softmax accumulation does not care about
    // the key order, so block xb starts at full tile (2 xb) mod nfull and wraps; the partial tail tile stays last (its
    // mask is the compile-time variant of the tile body).
    const int rot = (nfull > 0 && rot_on) ? (2 * xb_) % nfull : 0;
    auto pk0 = [&](int t) { return (t < nfull ? (t + rot >= nfull ? t + rot - nfull : t + rot) : t) * 64; };
    issue(pk0(0), 0);
    if (ntile > 1) issue(pk0(1), 1);
    int slot = 0;            // t % 3, kept as a running scalar (the modulo was strength-reduced into per-address VALU fix-ups)
    auto key_tile = [&](int t, auto tail_tag) {
        constexpr bool tail = decltype(tail_tag)::value;
        const int k0 = pk0(t);
        if (t + 1 < ntile) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile t landed; tile t+1 may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // everyone's pieces of tile t landed; everyone is done with tile t-1's slot
        asm volatile("" ::: "memory");
        if (t + 2 < ntile) issue(pk0(t + 2), slot == 0 ? 2 : slot - 1);
        if (q0 >= N) return;      // (lambda) a wave whose 32 queries are all past N only moves its DMA pieces
        const unsigned kbase = lds0 + slot * TILE, vbase = lds0 + (3 + slot) * TILE;
        f32x4 kr[4][2];
        ADS_R128(kr[0][0], kbase + ka0, 0);    ADS_R128(kr[0][1], kbase + ka1, 0);
        ADS_R128(kr[1][0], kbase + ka0, 2048); ADS_R128(kr[1][1], kbase + ka1, 2048);
        ADS_R128(kr[2][0], kbase + ka0, 4096); ADS_R128(kr[2][1], kbase + ka1, 4096);
        ADS_R128(kr[3][0], kbase + ka0, 6144); ADS_R128(kr[3][1], kbase + ka1, 6144);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kr[0][0]), "+v"(kr[0][1]), "+v"(kr[1][0]), "+v"(kr[1][1]), "+v"(kr[2][0]), "+v"(kr[2][1]),
                     "+v"(kr[3][0]), "+v"(kr[3][1]));
        f32x4 s[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 a = negm[qt];
#pragma unroll
                for (int u = 0; u < 2; ++u) a = Mma<T>::mma(__builtin_bit_cast(Frag, kr[kt][u]), qf[qt][u], a);
                s[qt][kt] = a;
            }
        // V fragments: issued now, consumed after the softmax
        a_u32x2 vr[4][2][2];   // [dt][u][row half]
#define ADS_V(dt)                                                                                   \
        ADS_TR64(vr[dt][0][0], vbase + va[dt], 0);    ADS_TR64(vr[dt][0][1], vbase + va[dt], 2048);     \
        ADS_TR64(vr[dt][1][0], vbase + va[dt], 4096); ADS_TR64(vr[dt][1][1], vbase + va[dt], 6144);
        ADS_V(0) ADS_V(1) ADS_V(2) ADS_V(3)
#undef ADS_V
        softmax_lagged<tail>(s, m, negm, oacc, lacc, t == 0, k0, g, N);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vr[0][0][0]), "+v"(vr[0][0][1]), "+v"(vr[0][1][0]), "+v"(vr[0][1][1]), "+v"(vr[1][0][0]),
                     "+v"(vr[1][0][1]), "+v"(vr[1][1][0]), "+v"(vr[1][1][1]), "+v"(vr[2][0][0]), "+v"(vr[2][0][1]), "+v"(vr[2][1][0]),
                     "+v"(vr[2][1][1]), "+v"(vr[3][0][0]), "+v"(vr[3][0][1]), "+v"(vr[3][1][0]), "+v"(vr[3][1][1]));
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            Frag pf[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                pf[qt] = acc_to_bfrag<T>(s[qt], u);
                lacc[qt] = Mma<T>::mma(ones, pf[qt], lacc[qt]);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                typedef __attribute__((ext_vector_type(4))) unsigned a_u32x4;
                const a_u32x4 z = {vr[dt][u][0][0], vr[dt][u][0][1], vr[dt][u][1][0], vr[dt][u][1][1]};
                const Frag vf = __builtin_bit_cast(Frag, z);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) oacc[dt][qt] = Mma<T>::mma(vf, pf[qt], oacc[dt][qt]);
            }
        }
    };
    int t = 0;
    {
        // ---- steady state: tiles t with t + 2 < nfull, three at a time (slot = t % 3 is the compile-time S of each copy)
        const int nsteady = nfull > 2 ? (nfull - 2) / 3 * 3 : 0;
        const bool live = q0 < N;                // (a wave whose 32 queries are all past N only moves its DMA pieces)
        const unsigned voff[2] = {(unsigned)(kp[0] - kb), (unsigned)(kp[1] - kb)};      // this lane's byte offset inside a tile, per piece (< 64 rows x ld_b < 2^31)
        int kt2 = nsteady ? pk0(2) : 0;          // first key of the tile two ahead, kept as a running scalar (+64, wrapping at nfull * 64)
        const int kwrap = nfull * 64;
        auto step = [&](auto slot_c, bool first) __attribute__((always_inline)) {
            constexpr int S = decltype(slot_c)::value, S2 = (S + 2) % 3;
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // tile t landed; tile t + 1 may still fly
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            {
                const unsigned long t0 = (unsigned)kt2 * (unsigned)ld_b;
                const char* ks = kb + t0;        // scalar tile bases: the pieces go out as saddr + 32-bit voffset
                const char* vs = vb + t0;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ks + voff[i]),
                                                     (__attribute__((address_space(3))) void*)(smem + S2 * TILE + (wave * 2 + i) * 1024), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vs + voff[i]),
                                                     (__attribute__((address_space(3))) void*)(smem + (3 + S2) * TILE + (wave * 2 + i) * 1024), 16, 0, 0);
                }
                kt2 += 64;
                if (kt2 == kwrap) kt2 = 0;
            }
            if (!live) return;
            const unsigned ak0 = lds0 + ka0, ak1 = lds0 + ka1;
            f32x4 kr[4][2];
            ADS_R128I(kr[0][0], ak0, S * TILE + 0);    ADS_R128I(kr[0][1], ak1, S * TILE + 0);
            ADS_R128I(kr[1][0], ak0, S * TILE + 2048); ADS_R128I(kr[1][1], ak1, S * TILE + 2048);
            ADS_R128I(kr[2][0], ak0, S * TILE + 4096); ADS_R128I(kr[2][1], ak1, S * TILE + 4096);
            ADS_R128I(kr[3][0], ak0, S * TILE + 6144); ADS_R128I(kr[3][1], ak1, S * TILE + 6144);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kr[0][0]), "+v"(kr[0][1]), "+v"(kr[1][0]), "+v"(kr[1][1]), "+v"(kr[2][0]), "+v"(kr[2][1]),
                         "+v"(kr[3][0]), "+v"(kr[3][1]));
            f32x4 s[2][4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    f32x4 a = negm[qt];
#pragma unroll
                    for (int u = 0; u < 2; ++u) a = Mma<T>::mma(__builtin_bit_cast(Frag, kr[kt][u]), qf[qt][u], a);
                    s[qt][kt] = a;
                }
            a_u32x2 vr[4][2][2];   // [dt][u][row half]
#define ADS_VI(dt)                                                                                                                   \
            ADS_TR64I(vr[dt][0][0], lds0 + va[dt], (3 + S) * TILE + 0);    ADS_TR64I(vr[dt][0][1], lds0 + va[dt], (3 + S) * TILE + 2048);       \
            ADS_TR64I(vr[dt][1][0], lds0 + va[dt], (3 + S) * TILE + 4096); ADS_TR64I(vr[dt][1][1], lds0 + va[dt], (3 + S) * TILE + 6144);
            ADS_VI(0) ADS_VI(1) ADS_VI(2) ADS_VI(3)
#undef ADS_VI
            softmax_lagged<false>(s, m, negm, oacc, lacc, first, 0, g, N);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vr[0][0][0]), "+v"(vr[0][0][1]), "+v"(vr[0][1][0]), "+v"(vr[0][1][1]), "+v"(vr[1][0][0]),
                         "+v"(vr[1][0][1]), "+v"(vr[1][1][0]), "+v"(vr[1][1][1]), "+v"(vr[2][0][0]), "+v"(vr[2][0][1]), "+v"(vr[2][1][0]),
                         "+v"(vr[2][1][1]), "+v"(vr[3][0][0]), "+v"(vr[3][0][1]), "+v"(vr[3][1][0]), "+v"(vr[3][1][1]));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                Frag pf[2];
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    pf[qt] = acc_to_bfrag<T>(s[qt], u);
                    lacc[qt] = Mma<T>::mma(ones, pf[qt], lacc[qt]);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    typedef __attribute__((ext_vector_type(4))) unsigned a_u32x4;
                    const a_u32x4 z = {vr[dt][u][0][0], vr[dt][u][0][1], vr[dt][u][1][0], vr[dt][u][1][1]};
                    const Frag vf = __builtin_bit_cast(Frag, z);
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) oacc[dt][qt] = Mma<T>::mma(vf, pf[qt], oacc[dt][qt]);
                }
            }
        };
        for (; t < nsteady; t += 3) {
            step(std::integral_constant<int, 0>{}, t == 0);
            step(std::integral_constant<int, 1>{}, false);
            step(std::integral_constant<int, 2>{}, false);
        }
        // (nsteady is a multiple of three: the generic tiles below start at slot 0 again)
    }
    for (; (t + 1) * 64 <= N; ++t) {
        key_tile(t, std::false_type{});
        slot = slot == 2 ? 0 : slot + 1;
    }
    if (t * 64 < N) key_tile(t, std::true_type{});
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + qt * 16 + c;
        if (q >= N) continue;
        const float lsum = lacc[qt][0];
        const float inv = 1.0f / lsum;
        T* orow = o + ((long)b * N + q) * H * HD + h * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4<T>(orow + dt * 16 + g * 4, oacc[dt][qt] * inv);
        if (g == 0) lse[((long)b * H + h) * N + q] = (m[qt] + log2f(lsum)) * 0.6931471805599453f;
    }
}

// dot of two operand fragments (the 16 bytes a lane holds of a row), fp32
__device__ __forceinline__ float frag_dot(bf16x8 a, bf16x8 b) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s = fmaf((float)a[k], (float)b[k], s);
    return s;
}
__device__ __forceinline__ float frag_dot(f16x8 a, f16x8 b) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s = fmaf((float)a[k], (float)b[k], s);
    return s;
}
__device__ __forceinline__ float frag_dot(f32x4 a, f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
__device__ __forceinline__ float frag_dot(const X3Frag& a, const X3Frag& b) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s = fmaf((float)a.hi[k] + (float)a.lo[k], (float)b.hi[k] + (float)b.lo[k], s);
    return s;
}

// ------------------------------------------------------------------------------------------ backward: dQ
template <typename T>
__global__ __launch_bounds__(256, IsX3<T>::v ? 1 : 2) void attn_bwd_dq_kernel(const T* qkv, const T* o, const T* dout, const float* lse,
                                                          float* delta, T* dqkv, int N, int H, float scale) {
    constexpr int NF = AT<T>::NF, ROWB = AT<T>::ROWB;
    typedef typename Mma<T>::Frag Frag;
    __shared__ __attribute__((aligned(16))) char sK[64 * ROWB];
    __shared__ __attribute__((aligned(16))) char sKt[TSZ(T)];
    __shared__ __attribute__((aligned(16))) char sV[64 * ROWB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    int xb_, h, b;
    attn_block_coords(xb_, h, b);
    const int q0 = xb_ * 128 + wave * 32;
    const int npad = (N + 63) / 64 * 64;
    const long ld_b = (long)3 * H * HD * sizeof(T);
    const char* base = (const char*)qkv + (long)b * N * ld_b;
    const char* qb = base + (long)(0 * H + h) * HD * sizeof(T);
    const char* kb = base + (long)(1 * H + h) * HD * sizeof(T);
    const char* vb = base + (long)(2 * H + h) * HD * sizeof(T);
    const long ldo_b = (long)H * HD * sizeof(T);
    const char* dob = (const char*)dout + (long)b * N * ldo_b + (long)h * HD * sizeof(T);
    const char* ob = (const char*)o + (long)b * N * ldo_b + (long)h * HD * sizeof(T);

    // delta = rowsum(dO * O) is taken here, from the dO fragments this wave loads anyway (+ the matching O fragments), and
    // written for the dK/dV kernel that runs next on the stream — it used to be its own pass over O and dO.
    Frag qf[2][NF], dof[2][NF];
    float lq[2], dl[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + qt * 16 + c;
        const bool ok = q < N;
        float dsum = 0.f;
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            Frag z = {};
            qf[qt][u] = ok ? frag_scale<T>(load_nfrag<T>(qb + (long)q * ld_b, u, g), scale * 1.4426950408889634f) : z;   // exp2 domain
            dof[qt][u] = ok ? load_nfrag<T>(dob + (long)q * ldo_b, u, g) : z;
            const Frag of_ = ok ? load_nfrag<T>(ob + (long)q * ldo_b, u, g) : z;
            dsum += frag_dot(dof[qt][u], of_);
        }
        dsum += __shfl_xor(dsum, 16, 64);      // the four lane groups g hold the four 16-byte pieces of each K-chunk
        dsum += __shfl_xor(dsum, 32, 64);
        lq[qt] = ok ? lse[((long)b * H + h) * N + q] * 1.4426950408889634f : 0.f;   // log2 units
        dl[qt] = dsum;
        // row constants for the dK/dV kernel, in the form its accumulators start from: ws[0] = -delta, ws[1] = -lse (log2 units), rows padded to whole
        // 64-query tiles — a pad query gets -lse = -1e30, i.e. p = exp2(s - 1e30) = 0 and dS = 0 whatever row its tile slot holds
        if (g == 0 && q < npad) {
            const long wi = ((long)b * H + h) * npad + q;
            delta[wi] = ok ? -dsum : 0.f;
            delta[(long)gridDim.z * H * npad + wi] = ok ? -lq[qt] : -1e30f;
        }
    }
    f32x4 dq[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) dq[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // row constants as the initial accumulators (the MFMA's C operand): S' = q k c2 - lse2 and dP' = dO v - delta leave the
    // chains ready, p = exp2(S') and dS = p * dP' — one v_exp and one multiply per score, no subtract / scale FMA
    f32x4 nlq[2], ndl[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        nlq[qt] = f32x4{-lq[qt], -lq[qt], -lq[qt], -lq[qt]};
        ndl[qt] = f32x4{-dl[qt], -dl[qt], -dl[qt], -dl[qt]};
    }

    // (key tiles in plain order: a per-block rotated order, which helps the forward kernel a little, measured 4 % SLOWER here —
    //  the blocks of an (image, head) trail each other through the L2 and the rotation takes that away)
    const int ntile = (N + 63) / 64;
    auto pk0 = [&](int t) { return t * 64; };
    TileRegs<T> rk, rv;
    TileSrc<T> ksrc, vsrc;
    tile_src_init<T>(ksrc, kb, ld_b, N);
    tile_src_init<T>(vsrc, vb, ld_b, N);
    tile_load<T>(rk, ksrc, pk0(0));
    tile_load<T>(rv, vsrc, pk0(0));
    // TAIL (compile-time): the partial last tile.  Keys >= N are staged as zero K / V rows, so their dS never reaches dQ through the product — but
    // their p = exp2(0 - lse2) is unbounded when the row's scores are all far below zero (lse2 << 0), and a 16-bit dS = p * dP' that overflows to
    // inf makes inf * 0 = NaN in the dQ product (fp16: 5 exponent bits; found by the all_negative case of test_attention_reference_point_moves).
    // Their dS is therefore set to zero, in the tail tile only.
    auto key_tile = [&](int t, auto tail_tag) {
        constexpr bool tail = decltype(tail_tag)::value;
        __syncthreads();
        tile_store<T, true, TOp<T>::kNeedT>(rk, sK, sKt);
        tile_store<T, true, false>(rv, sV, nullptr);
        __syncthreads();
        if (t + 1 < ntile) {
            tile_load<T>(rk, ksrc, pk0(t + 1));
            tile_load<T>(rv, vsrc, pk0(t + 1));
        }
        if (q0 >= N) return;   // a wave whose 32 queries are all past N only helps staging
        f32x4 ds[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            Frag kf[NF], vf[NF];
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                kf[u] = lds_nfrag<T>(sK, kt * 16 + c, u, g);
                vf[u] = lds_nfrag<T>(sV, kt * 16 + c, u, g);
            }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 s = nlq[qt], dp = ndl[qt];
#pragma unroll
                for (int u = 0; u < NF; ++u) {
                    s = Mma<T>::mma(kf[u], qf[qt][u], s);
                    dp = Mma<T>::mma(vf[u], dof[qt][u], dp);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ds[qt][kt][r] = fast_exp2(s[r]) * dp[r];       // the 1/sqrt(d) factor is applied once, to dQ
                    if (tail && t * 64 + kt * 16 + g * 4 + r >= N) ds[qt][kt][r] = 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            Frag df[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) df[qt] = acc_to_bfrag<T>(ds[qt], u);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const Frag kf = TOp<T>::load(sK, sKt, dt, u, g, lane);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) dq[dt][qt] = Mma<T>::mma(kf, df[qt], dq[dt][qt]);
            }
        }
    };
    {
        int t = 0;
        for (; (t + 1) * 64 <= N; ++t) key_tile(t, std::false_type{});
        if (t * 64 < N) key_tile(t, std::true_type{});
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + qt * 16 + c;
        if (q >= N) continue;
        T* row = dqkv + ((long)b * N + q) * 3 * H * HD + (long)(0 * H + h) * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4<T>(row + dt * 16 + g * 4, dq[dt][qt] * scale);
    }
}

// ------------------------------------------------------------------------------------------ backward: dQ on an LDS-DMA ring (round 6)
// attn_bwd_dq_kernel with the K / V tiles on the forward's three-slot LDS-DMA ring (two tiles ahead, one counted vmcnt wait + one barrier per tile, no
// staging registers) and every LDS read inline asm, pipelined by hand under the MFMAs — the dK/dV kernel's recipe (attn_bwd_dkv_dma_kernel, below).
// Key rows past N are clamped to row N - 1 in the partial last tile (the register-staged kernel stages zeros): their dS is set to zero there, as before.
#define ADS_R128V(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define ADS_TR64V(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
template <typename T>      // bf16 | f16
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_dma_kernel(const T* qkv, const T* o, const T* dout, const float* lse, float* delta, T* dqkv, int N, int H,
                                                                 float scale) {
    typedef typename Mma<T>::Frag Frag;
    constexpr int TILE = 64 * 128;
    __shared__ __attribute__((aligned(16))) char smem[6 * TILE];      // K slots 0..2 | V slots 0..2
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int xb_, h, b;
    attn_block_coords(xb_, h, b);
    const int q0 = xb_ * 128 + wave * 32;
    const int npad = (N + 63) / 64 * 64;
    const long ld_b = (long)3 * H * HD * 2;
    const char* base = (const char*)qkv + (long)b * N * ld_b;
    const char* qb = base + (long)(0 * H + h) * HD * 2;
    const char* kb = base + (long)(1 * H + h) * HD * 2;
    const char* vb = base + (long)(2 * H + h) * HD * 2;
    const long ldo_b = (long)H * HD * 2;
    const char* dob = (const char*)dout + (long)b * N * ldo_b + (long)h * HD * 2;
    const char* ob = (const char*)o + (long)b * N * ldo_b + (long)h * HD * 2;

    // delta = rowsum(dO * O), from the dO fragments this wave loads anyway; left for the dK/dV kernel with -lse2 (see attn_bwd_dq_kernel)
    Frag qf[2][2], dof[2][2];
    f32x4 nlq[2], ndl[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + qt * 16 + c;
        const bool ok = q < N;
        float dsum = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            Frag z = {};
            qf[qt][u] = ok ? frag_scale<T>(load_nfrag<T>(qb + (long)q * ld_b, u, g), scale * 1.4426950408889634f) : z;   // exp2 domain
            dof[qt][u] = ok ? load_nfrag<T>(dob + (long)q * ldo_b, u, g) : z;
            const Frag of_ = ok ? load_nfrag<T>(ob + (long)q * ldo_b, u, g) : z;
            dsum += frag_dot(dof[qt][u], of_);
        }
        dsum += __shfl_xor(dsum, 16, 64);
        dsum += __shfl_xor(dsum, 32, 64);
        const float lq = ok ? lse[((long)b * H + h) * N + q] * 1.4426950408889634f : 0.f;   // log2 units
        if (g == 0 && q < npad) {
            const long wi = ((long)b * H + h) * npad + q;
            delta[wi] = ok ? -dsum : 0.f;
            delta[(long)gridDim.z * H * npad + wi] = ok ? -lq : -1e30f;
        }
        nlq[qt] = f32x4{-lq, -lq, -lq, -lq};
        ndl[qt] = f32x4{-dsum, -dsum, -dsum, -dsum};
    }
    f32x4 dq[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) dq[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // DMA: wave w moves pieces 2w, 2w + 1 (8 rows each) of the K tile and of the V tile: four operations per wave and tile
    const int prow = lane >> 3, pchunk = lane & 7;
    const char* kp[2];
    const char* vp_[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + prow;
        const long off = (long)row * ld_b + ((pchunk ^ AT<T>::sw(row)) * 16);
        kp[i] = kb + off;
        vp_[i] = vb + off;
    }
    auto issue = [&](int k0, int slot) {
        if (k0 + 64 <= N) {
            const unsigned long t0 = (unsigned)k0 * (unsigned)ld_b;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kp[i] + t0),
                                                 (__attribute__((address_space(3))) void*)(smem + slot * TILE + (wave * 2 + i) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vp_[i] + t0),
                                                 (__attribute__((address_space(3))) void*)(smem + (3 + slot) * TILE + (wave * 2 + i) * 1024), 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 8 + prow;
            const long off = (long)min(k0 + row, N - 1) * ld_b + ((pchunk ^ AT<T>::sw(row)) * 16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb + off),
                                             (__attribute__((address_space(3))) void*)(smem + slot * TILE + (wave * 2 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + off),
                                             (__attribute__((address_space(3))) void*)(smem + (3 + slot) * TILE + (wave * 2 + i) * 1024), 16, 0, 0);
        }
    };
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    const int swn = AT<T>::sw(c);
    const unsigned na0 = c * 128 + ((0 + g) ^ swn) * 16, na1 = c * 128 + ((4 + g) ^ swn) * 16;      // natural fragment (kt, u): row 16 kt + c, chunk 4u + g
    const int tq_ = c >> 2, tp = c & 3, swt = AT<T>::sw(4 * g + tq_);
    unsigned ta[4];                                                                                  // transposed fragment (dt, u): rows 32u + 4g + q (+16), chunk 2dt + (p >> 1)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ta[dt] = (4 * g + tq_) * 128 + (((2 * dt + (tp >> 1)) ^ swt) * 16) + 8 * (tp & 1);

    const int ntile = (N + 63) / 64;
    const bool live = q0 < N;      // a wave whose 32 queries are all past N only moves its DMA pieces
    issue(0, 0);
    if (ntile > 1) issue(64, 1);
    int slot = 0;
    auto key_tile = [&](int t, auto tail_tag) __attribute__((always_inline)) {
        constexpr bool tail = decltype(tail_tag)::value;
        if (t + 1 < ntile) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile t landed; tile t + 1 may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 2 < ntile) issue((t + 2) * 64, slot == 0 ? 2 : slot - 1);
        const unsigned kbase = lds0 + slot * TILE, vbase = lds0 + (3 + slot) * TILE;
        slot = slot == 2 ? 0 : slot + 1;
        if (!live) return;
        // RK(kt): the K and V natural fragments of key tile kt; CK(kt): S / dP MFMAs + exp for it (both query tiles); RT(u, dt): the transposed K fragments
        // of head-dim tile dt for the 32 keys of half u; MQ(u, dt): its two dQ MFMAs.  Order (at most 12 reads outstanding):
        //   RK0 RK1 | CK0 | RK2 | CK1 | RT00 RT01 | MQ00 | RT02 | MQ01 | RT03 | MQ02 | RK3 | MQ03 | CK2 | CK3 | RT10 RT11 | MQ10 | RT12 | MQ11 | RT13 | MQ12 | MQ13
        const unsigned ka0 = kbase + na0, ka1 = kbase + na1, va0 = vbase + na0, va1 = vbase + na1;
        f32x4 rk0[2], rk1[2], rv0[2], rv1[2];
        a_u32x2 tr_[2][2];
        f32x4 ds[2][2];      // [query tile][key tile & 1]
        Frag df[2];
#define DQ_RK(kt, S) ADS_R128V(rk0[S], ka0, kt * 2048); ADS_R128V(rk1[S], ka1, kt * 2048); ADS_R128V(rv0[S], va0, kt * 2048); ADS_R128V(rv1[S], va1, kt * 2048);
#define DQ_WAIT_K(n, S) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(rk0[S]), "+v"(rk1[S]), "+v"(rv0[S]), "+v"(rv1[S]) : "i"(n));
#define DQ_CK(kt, S)                                                                                                                    \
        _Pragma("unroll") for (int qt = 0; qt < 2; ++qt) {                                                                              \
            f32x4 sacc = nlq[qt], dpa = ndl[qt];                                                                                        \
            sacc = Mma<T>::mma(__builtin_bit_cast(Frag, rk0[S]), qf[qt][0], sacc);                                                      \
            dpa = Mma<T>::mma(__builtin_bit_cast(Frag, rv0[S]), dof[qt][0], dpa);                                                       \
            sacc = Mma<T>::mma(__builtin_bit_cast(Frag, rk1[S]), qf[qt][1], sacc);                                                      \
            dpa = Mma<T>::mma(__builtin_bit_cast(Frag, rv1[S]), dof[qt][1], dpa);                                                       \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                             \
                ds[qt][kt & 1][r] = fast_exp2(sacc[r]) * dpa[r];       /* the 1/sqrt(d) factor is applied once, to dQ */                \
                if (tail && t * 64 + kt * 16 + g * 4 + r >= N) ds[qt][kt & 1][r] = 0.f;                                                 \
            }                                                                                                                           \
        }
#define DQ_FRAGS()                                                                                                                      \
        _Pragma("unroll") for (int qt = 0; qt < 2; ++qt) { const f32x4 t4[4] = {ds[qt][0], ds[qt][1], ds[qt][0], ds[qt][1]}; df[qt] = acc_to_bfrag<T>(t4, 0); }
#define DQ_RT(u, dt, S) ADS_TR64V(tr_[S][0], kbase + ta[dt], u * 4096); ADS_TR64V(tr_[S][1], kbase + ta[dt], u * 4096 + 2048);
#define DQ_WAIT_T(n, S) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(tr_[S][0]), "+v"(tr_[S][1]) : "i"(n));
#define DQ_MQ(dt, S)                                                                                                                    \
        {                                                                                                                               \
            typedef __attribute__((ext_vector_type(4))) unsigned a_u32x4;                                                               \
            const a_u32x4 z = {tr_[S][0][0], tr_[S][0][1], tr_[S][1][0], tr_[S][1][1]};                                                 \
            const Frag kt_ = __builtin_bit_cast(Frag, z);                                                                               \
            _Pragma("unroll") for (int qt = 0; qt < 2; ++qt) dq[dt][qt] = Mma<T>::mma(kt_, df[qt], dq[dt][qt]);                         \
        }
        DQ_RK(0, 0) DQ_RK(1, 1)
        DQ_WAIT_K(4, 0) DQ_CK(0, 0)
        DQ_RK(2, 0)
        DQ_WAIT_K(4, 1) DQ_CK(1, 1)
        DQ_FRAGS()
        DQ_RT(0, 0, 0) DQ_RT(0, 1, 1)
        DQ_WAIT_T(2, 0) DQ_MQ(0, 0)
        DQ_RT(0, 2, 0)
        DQ_WAIT_T(2, 1) DQ_MQ(1, 1)
        DQ_RT(0, 3, 1)
        DQ_WAIT_T(2, 0) DQ_MQ(2, 0)
        DQ_RK(3, 1)
        DQ_WAIT_T(4, 1) DQ_MQ(3, 1)
        DQ_WAIT_K(4, 0) DQ_CK(2, 0)
        DQ_WAIT_K(0, 1) DQ_CK(3, 1)
        DQ_FRAGS()
        DQ_RT(1, 0, 0) DQ_RT(1, 1, 1)
        DQ_WAIT_T(2, 0) DQ_MQ(0, 0)
        DQ_RT(1, 2, 0)
        DQ_WAIT_T(2, 1) DQ_MQ(1, 1)
        DQ_RT(1, 3, 1)
        DQ_WAIT_T(2, 0) DQ_MQ(2, 0)
        DQ_WAIT_T(0, 1) DQ_MQ(3, 1)
#undef DQ_RK
#undef DQ_WAIT_K
#undef DQ_CK
#undef DQ_FRAGS
#undef DQ_RT
#undef DQ_WAIT_T
#undef DQ_MQ
    };
    {
        int t = 0;
        for (; (t + 1) * 64 <= N; ++t) key_tile(t, std::false_type{});
        if (t * 64 < N) key_tile(t, std::true_type{});
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + qt * 16 + c;
        if (q >= N) continue;
        T* row = dqkv + ((long)b * N + q) * 3 * H * HD + (long)(0 * H + h) * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4<T>(row + dt * 16 + g * 4, dq[dt][qt] * scale);
    }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
// DK = false: dV only (the first trainable block of the student: nothing below it learns, so dK — and with it dP, dS and the Q^T
// operand — is never used; half of the kernel's MFMAs)
template <typename T, int NW, bool DK = true>
__global__ __launch_bounds__(64 * NW, (NW >= 8 || IsX3<T>::v) ? 1 : 2) void attn_bwd_dkv_kernel(const T* qkv, const T* dout, const float* lse,
                                                           const float* delta, T* dqkv, int N, int H, float scale, int vfirst) {
    vfirst &= 0xff;
    constexpr int NF = AT<T>::NF, ROWB = AT<T>::ROWB;
    typedef typename Mma<T>::Frag Frag;
    __shared__ __attribute__((aligned(16))) char sQ[64 * ROWB];
    __shared__ __attribute__((aligned(16))) char sQt[TSZ(T)];
    __shared__ __attribute__((aligned(16))) char sD[64 * ROWB];
    __shared__ __attribute__((aligned(16))) char sDt[TSZ(T)];
    __shared__ __attribute__((aligned(16))) float sL[64], sDl[64];   // -lse (log2 units) and -delta of the tile's 64 queries
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    constexpr int NT = 64 * NW;
    int xb_, h, b;
    attn_block_coords(xb_, h, b);
    const int key0 = xb_ * (32 * NW) + wave * 32;
    const long ld_b = (long)3 * H * HD * sizeof(T);
    const char* base = (const char*)qkv + (long)b * N * ld_b;
    const char* qb = base + (long)(0 * H + h) * HD * sizeof(T);
    const char* kb = base + (long)(1 * H + h) * HD * sizeof(T);
    const char* vb = base + (long)(2 * H + h) * HD * sizeof(T);
    const long ldo_b = (long)H * HD * sizeof(T);
    const char* dob = (const char*)dout + (long)b * N * ldo_b + (long)h * HD * sizeof(T);

    Frag kf[2][NF], vf[2][NF];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = key0 + kt * 16 + c;
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            Frag z = {};
            kf[kt][u] = key < N ? frag_scale<T>(load_nfrag<T>(kb + (long)key * ld_b, u, g), scale * 1.4426950408889634f) : z;   // exp2 domain
            vf[kt][u] = key < N ? load_nfrag<T>(vb + (long)key * ld_b, u, g) : z;
        }
    }
    const bool wave_live = key0 < N;
    f32x4 dk[4][2], dv[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { dk[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int ntile = (N + 63) / 64;
    auto pq0 = [&](int t) { return t * 64; };
    TileRegs<T, NT> rq, rd;
    TileSrc<T, NT> qsrc, dsrc;
    tile_src_init<T, NT>(qsrc, qb, ld_b, N);
    tile_src_init<T, NT>(dsrc, dob, ldo_b, N);
    tile_load<T, NT>(rq, qsrc, pq0(0));
    tile_load<T, NT>(rd, dsrc, pq0(0));
    float rl = 0.f, rdl = 0.f;   // next tile's -lse (log2 units) / -delta rows, prefetched with the tile
    // the dQ kernel left them in that form in `delta` = ws [2][B * H][npad] (ws[0] = -delta, ws[1] = -lse2), padded to whole tiles: queries >= N have
    // -lse2 = -1e30, i.e. p = 2^(s - 1e30) = 0 (their Q / dO rows are staged as zeros) — no mask in the loop.  (`lse` itself is not read here any more.)
    const int npad = (N + 63) / 64 * 64;
    const float* wsd = delta + ((long)b * H + h) * npad;
    const float* wsl = delta + (long)gridDim.z * H * npad + ((long)b * H + h) * npad;
    if (threadIdx.x < 64) {
        const int q = pq0(0) + threadIdx.x;
        rl = wsl[q];
        rdl = wsd[q];
    }
    for (int t = 0; t < ntile; ++t) {
        __syncthreads();
        tile_store<T, true, TOp<T>::kNeedT, NT>(rq, sQ, sQt);
        tile_store<T, true, TOp<T>::kNeedT, NT>(rd, sD, sDt);
        if (t + 1 < ntile) {
            tile_load<T, NT>(rq, qsrc, pq0(t + 1));
            tile_load<T, NT>(rd, dsrc, pq0(t + 1));
        }
        if (threadIdx.x < 64) { sL[threadIdx.x] = rl; sDl[threadIdx.x] = rdl; }   // (already negated: accumulator seeds)
        if (t + 1 < ntile && threadIdx.x < 64) {
            const int q = pq0(t + 1) + threadIdx.x;
            rl = wsl[q];
            rdl = wsd[q];
        }
        __syncthreads();
        if (!wave_live) continue;   // (this wave's 32 keys are all past N: it only helps staging the tiles — 5 of the 48 waves of an
                                    //  (image, head) at N = 1370)
        f32x4 pp[2][4], dsv[2][4];  // [key tile][query tile]
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            Frag qf[NF], df[NF];
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                qf[u] = lds_nfrag<T>(sQ, qt * 16 + c, u, g);
                if (DK) df[u] = lds_nfrag<T>(sD, qt * 16 + c, u, g);
            }
            // row constants as the initial accumulators: the four queries 16 qt + 4 g + r of this lane's C rows
            const f32x4 nl = *(const f32x4*)(sL + qt * 16 + g * 4), nd = *(const f32x4*)(sDl + qt * 16 + g * 4);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x4 s = nl, dp = nd;
#pragma unroll
                for (int u = 0; u < NF; ++u) {
                    s = Mma<T>::mma(qf[u], kf[kt][u], s);    // rows: queries, cols: keys
                    if (DK) dp = Mma<T>::mma(df[u], vf[kt][u], dp);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = fast_exp2(s[r]);
                    pp[kt][qt][r] = p;
                    if (DK) dsv[kt][qt][r] = p * dp[r];            // the 1/sqrt(d) factor is applied once, to dK
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            Frag pf[2], sf[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) { pf[kt] = acc_to_bfrag<T>(pp[kt], u); if (DK) sf[kt] = acc_to_bfrag<T>(dsv[kt], u); }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const Frag dot = TOp<T>::load(sD, sDt, dt, u, g, lane);
                Frag qt_ = dot;
                if (DK) qt_ = TOp<T>::load(sQ, sQt, dt, u, g, lane);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    dv[dt][kt] = Mma<T>::mma(dot, pf[kt], dv[dt][kt]);
                    if (DK) dk[dt][kt] = Mma<T>::mma(qt_, sf[kt], dk[dt][kt]);
                }
            }
        }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = key0 + kt * 16 + c;
        if (key >= N) continue;
        // column block of dK / dV inside the packed gradient row: (dq, dk, dv) or, with vfirst, (dq, dv, dk)
        T* rk = dqkv + ((long)b * N + key) * 3 * H * HD + (long)((vfirst ? 2 : 1) * H + h) * HD;
        T* rv = dqkv + ((long)b * N + key) * 3 * H * HD + (long)((vfirst ? 1 : 2) * H + h) * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            if (DK) store4<T>(rk + dt * 16 + g * 4, dk[dt][kt] * scale);
            store4<T>(rv + dt * 16 + g * 4, dv[dt][kt]);
        }
    }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV on an LDS-DMA ring (round 6)
// attn_bwd_dkv_kernel<T, 4> spends 27 % of its time staging the Q / dO tiles (anatomy, profiles/r06_probe_attn_tn.txt: buffer loads into registers ->
// ds_write -> two block barriers per 64-query tile; 255 of 935 us at 64 x 12 x 1370).  Here the tiles — and the tile's 64 + 64 row constants, which
// the dQ kernel leaves negated and in log2 units in ws — travel global -> LDS by LDS-DMA into a THREE-slot ring, two tiles ahead, as in
// attn_fwd_dma_kernel: no staging registers, no ds_write, no address arithmetic on the vector ALU for full tiles, ONE counted vmcnt wait and ONE barrier
// per tile.  Same mathematics, register layouts and swizzle (chunk ^ AT<T>::sw(row), applied on the SOURCE address) as the register-staged kernel; every
// LDS read of the ring is inline asm (a ds_read the compiler can see gets an `s_waitcnt vmcnt(0)` to the LDS-DMA in flight in front of it).
// Query rows past N are clamped to row N - 1 in the partial last tile: their -lse2 is -1e30 (ws pads), so p = 0 and dS = 0 * finite = 0.
template <typename T, bool DK = true>      // bf16 | f16; four waves = 128 keys per block
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_dma_kernel(const T* qkv, const T* dout, const float* ws, T* dqkv, int N, int H, float scale, int vfirst) {
    typedef typename Mma<T>::Frag Frag;
    constexpr int TILE = 64 * 128;                    // one Q or dO tile: 64 rows x 128 B
    constexpr int SLOT = 2 * TILE + 512;              // Q tile | dO tile | -lse2 [64] | -delta [64]
    __shared__ __attribute__((aligned(16))) char smem[3 * SLOT];
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int xb_, h, b;
    attn_block_coords(xb_, h, b);
    const int key0 = xb_ * 128 + wave * 32;
    const long ld_b = (long)3 * H * HD * 2;
    const char* base = (const char*)qkv + (long)b * N * ld_b;
    const char* qb = base + (long)(0 * H + h) * HD * 2;
    const char* kb = base + (long)(1 * H + h) * HD * 2;
    const char* vb = base + (long)(2 * H + h) * HD * 2;
    const long ldo_b = (long)H * HD * 2;
    const char* dob = (const char*)dout + (long)b * N * ldo_b + (long)h * HD * 2;
    const int npad = (N + 63) / 64 * 64;
    const float* wsd = ws + ((long)b * H + h) * npad;                                         // -delta
    const float* wsl = ws + (long)gridDim.z * H * npad + ((long)b * H + h) * npad;            // -lse (log2 units)

    Frag kf[2][2], vf[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = key0 + kt * 16 + c;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            Frag z = {};
            kf[kt][u] = key < N ? frag_scale<T>(load_nfrag<T>(kb + (long)key * ld_b, u, g), scale * 1.4426950408889634f) : z;   // exp2 domain
            vf[kt][u] = key < N ? load_nfrag<T>(vb + (long)key * ld_b, u, g) : z;
        }
    }
    const bool wave_live = key0 < N;
    f32x4 dk[4][2], dv[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { dk[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // DMA: wave w moves pieces 2w, 2w + 1 (8 rows each) of the Q tile and of the dO tile, and one 256-byte piece of row constants (even waves -lse2,
    // odd waves -delta; waves 2 / 3 repeat what 0 / 1 send — the same bytes to the same place — so that every wave counts FIVE operations per tile)
    const int prow = lane >> 3, pchunk = lane & 7;
    const char* qp[2];
    const char* dp_[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + prow;
        qp[i] = qb + (long)row * ld_b + ((pchunk ^ AT<T>::sw(row)) * 16);
        dp_[i] = dob + (long)row * ldo_b + ((pchunk ^ AT<T>::sw(row)) * 16);
    }
    const float* cp = ((wave & 1) ? wsd : wsl) + lane;
    const int coff = 2 * TILE + (wave & 1) * 256;
    auto issue = [&](int q0t, int slot) {
        char* sb = smem + slot * SLOT;
        if (q0t + 64 <= N) {
            const unsigned long tq = (unsigned)q0t * (unsigned)ld_b, td = (unsigned)q0t * (unsigned)ldo_b;     // (an image's rows span < 2^31 bytes)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qp[i] + tq),
                                                 (__attribute__((address_space(3))) void*)(sb + (wave * 2 + i) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dp_[i] + td),
                                                 (__attribute__((address_space(3))) void*)(sb + TILE + (wave * 2 + i) * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wave * 2 + i) * 8 + prow;
                const long r = min(q0t + row, N - 1);
                const int sw16 = (pchunk ^ AT<T>::sw(row)) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qb + r * ld_b + sw16),
                                                 (__attribute__((address_space(3))) void*)(sb + (wave * 2 + i) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dob + r * ldo_b + sw16),
                                                 (__attribute__((address_space(3))) void*)(sb + TILE + (wave * 2 + i) * 1024), 16, 0, 0);
            }
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cp + q0t),
                                         (__attribute__((address_space(3))) void*)(sb + coff), 4, 0, 0);
    };
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    // natural fragment (qt, u): row qt * 16 + c, logical chunk 4u + g (row qt * 16 + c has the swizzle of row c)
    const int swn = AT<T>::sw(c);
    const unsigned na0 = c * 128 + ((0 + g) ^ swn) * 16, na1 = c * 128 + ((4 + g) ^ swn) * 16;
    // transposed fragment (dt, u): rows 32u + 4g + q (+16), logical chunk 2dt + (p >> 1), 8-byte half p & 1
    const int tq_ = c >> 2, tp = c & 3, swt = AT<T>::sw(4 * g + tq_);
    unsigned ta[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ta[dt] = (4 * g + tq_) * 128 + (((2 * dt + (tp >> 1)) ^ swt) * 16) + 8 * (tp & 1);
    const unsigned ca = 2 * TILE + g * 16;            // the four row constants 16 qt + 4g + r of this lane's C rows: one b128 at +64 qt

    const int ntile = (N + 63) / 64;
    issue(0, 0);
    if (ntile > 1) issue(64, 1);
    int slot = 0;
    for (int t = 0; t < ntile; ++t) {
        if (t + 1 < ntile) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // tile t landed; tile t + 1 may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // everyone's pieces of tile t landed; everyone is done with tile t - 1's slot
        asm volatile("" ::: "memory");
        if (t + 2 < ntile) issue((t + 2) * 64, slot == 0 ? 2 : slot - 1);
        const unsigned sbase = lds0 + slot * SLOT;
        slot = slot == 2 ? 0 : slot + 1;
        if (!wave_live) continue;   // (this wave's 32 keys are all past N: it only moves its DMA pieces)
        // ---- the tile, software-pipelined by hand over the LDS reads (inline asm reads are invisible to the scheduler; at most 14 are outstanding:
        // lgkmcnt counts to 15).  RA(qt): the natural fragments + row constants of query tile qt; CA(qt): S / dP MFMAs + exp for it; RT(u, dt): the
        // transposed dO / Q fragments of head-dim tile dt for the 32 queries of half u; MB(u, dt): its four dV / dK MFMAs.  Order:
        //   RA0 RA1 | CA0 | RA2 | CA1 | RT00 RT01 | MB00 | RT02 | MB01 | RT03 | MB02 | RA3 | MB03 | CA2 | CA3 | RT10 RT11 | MB10 | RT12 | MB11 | RT13 | MB12 | MB13
        const unsigned qa0 = sbase + na0, qa1 = sbase + na1, da0 = sbase + TILE + na0, da1 = sbase + TILE + na1, cb = sbase + ca;
        const unsigned dtb = sbase + TILE, qtb = sbase;
        f32x4 aq0[2], aq1[2], ad0[2], ad1[2], anl[2], and_[2];      // two register sets for RA
        a_u32x2 tdr[2][2], tqr[2][2];                               // two register sets for RT: [set][row half]
        f32x4 pp[2][2], dsv[2][2];                                   // [key tile][query tile & 1]: two query tiles at a time
        Frag pf[2], sf[2];
        constexpr int NRA = DK ? 6 : 3, NRT = DK ? 4 : 2;            // LDS reads per RA / RT
#define DKV_RA(qt, S)                                                                                                                   \
        ADS_R128V(aq0[S], qa0, qt * 2048); ADS_R128V(aq1[S], qa1, qt * 2048); ADS_R128V(anl[S], cb, qt * 64);                           \
        if (DK) { ADS_R128V(ad0[S], da0, qt * 2048); ADS_R128V(ad1[S], da1, qt * 2048); ADS_R128V(and_[S], cb, 256 + qt * 64); }
#define DKV_WAIT_A(n, S)                                                                                                                \
        if (DK) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(aq0[S]), "+v"(aq1[S]), "+v"(ad0[S]), "+v"(ad1[S]), "+v"(anl[S]), "+v"(and_[S]) : "i"(n)); \
        else asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(aq0[S]), "+v"(aq1[S]), "+v"(anl[S]) : "i"(n));
#define DKV_CA(qt, S)                                                                                                                   \
        {                                                                                                                               \
            const Frag qf0 = __builtin_bit_cast(Frag, aq0[S]), qf1 = __builtin_bit_cast(Frag, aq1[S]);                                  \
            _Pragma("unroll") for (int kt = 0; kt < 2; ++kt) {                                                                          \
                f32x4 sacc = anl[S], dpa = anl[S];                                                                                      \
                sacc = Mma<T>::mma(qf0, kf[kt][0], sacc);    /* rows: queries, cols: keys */                                            \
                sacc = Mma<T>::mma(qf1, kf[kt][1], sacc);                                                                               \
                if (DK) {                                                                                                               \
                    dpa = and_[S];                                                                                                      \
                    dpa = Mma<T>::mma(__builtin_bit_cast(Frag, ad0[S]), vf[kt][0], dpa);                                                \
                    dpa = Mma<T>::mma(__builtin_bit_cast(Frag, ad1[S]), vf[kt][1], dpa);                                                \
                }                                                                                                                       \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                         \
                    const float pv = fast_exp2(sacc[r]);                                                                                \
                    pp[kt][qt & 1][r] = pv;                                                                                             \
                    if (DK) dsv[kt][qt & 1][r] = pv * dpa[r];          /* the 1/sqrt(d) factor is applied once, to dK */                \
                }                                                                                                                       \
            }                                                                                                                           \
        }
        // the B fragments of half u from the two query tiles just computed (contraction index = 16 (qt & 1) + 4g + r within the half)
#define DKV_FRAGS()                                                                                                                     \
        _Pragma("unroll") for (int kt = 0; kt < 2; ++kt) {                                                                              \
            const f32x4 t4[4] = {pp[kt][0], pp[kt][1], pp[kt][0], pp[kt][1]};                                                           \
            pf[kt] = acc_to_bfrag<T>(t4, 0);                                                                                            \
            if (DK) { const f32x4 d4[4] = {dsv[kt][0], dsv[kt][1], dsv[kt][0], dsv[kt][1]}; sf[kt] = acc_to_bfrag<T>(d4, 0); }          \
        }
#define DKV_RT(u, dt, S)                                                                                                                \
        ADS_TR64V(tdr[S][0], dtb + ta[dt], u * 4096); ADS_TR64V(tdr[S][1], dtb + ta[dt], u * 4096 + 2048);                              \
        if (DK) { ADS_TR64V(tqr[S][0], qtb + ta[dt], u * 4096); ADS_TR64V(tqr[S][1], qtb + ta[dt], u * 4096 + 2048); }
#define DKV_WAIT_T(n, S)                                                                                                                \
        if (DK) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(tdr[S][0]), "+v"(tdr[S][1]), "+v"(tqr[S][0]), "+v"(tqr[S][1]) : "i"(n));    \
        else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(tdr[S][0]), "+v"(tdr[S][1]) : "i"(n));
#define DKV_MB(dt, S)                                                                                                                   \
        {                                                                                                                               \
            typedef __attribute__((ext_vector_type(4))) unsigned a_u32x4;                                                               \
            const a_u32x4 zd = {tdr[S][0][0], tdr[S][0][1], tdr[S][1][0], tdr[S][1][1]};                                                \
            const Frag dot = __builtin_bit_cast(Frag, zd);                                                                              \
            _Pragma("unroll") for (int kt = 0; kt < 2; ++kt) dv[dt][kt] = Mma<T>::mma(dot, pf[kt], dv[dt][kt]);                         \
            if (DK) {                                                                                                                   \
                const a_u32x4 zq = {tqr[S][0][0], tqr[S][0][1], tqr[S][1][0], tqr[S][1][1]};                                            \
                const Frag qt_ = __builtin_bit_cast(Frag, zq);                                                                          \
                _Pragma("unroll") for (int kt = 0; kt < 2; ++kt) dk[dt][kt] = Mma<T>::mma(qt_, sf[kt], dk[dt][kt]);                     \
            }                                                                                                                           \
        }
        DKV_RA(0, 0) DKV_RA(1, 1)
        DKV_WAIT_A(NRA, 0) DKV_CA(0, 0)
        DKV_RA(2, 0)
        DKV_WAIT_A(NRA, 1) DKV_CA(1, 1)
        DKV_FRAGS()
        DKV_RT(0, 0, 0) DKV_RT(0, 1, 1)
        DKV_WAIT_T(NRT, 0) DKV_MB(0, 0)            // (RA2 is older than the RTs: it has landed too)
        DKV_RT(0, 2, 0)
        DKV_WAIT_T(NRT, 1) DKV_MB(1, 1)
        DKV_RT(0, 3, 1)
        DKV_WAIT_T(NRT, 0) DKV_MB(2, 0)
        DKV_RA(3, 1)
        DKV_WAIT_T(NRA, 1) DKV_MB(3, 1)
        DKV_WAIT_A(NRA, 0) DKV_CA(2, 0)            // (landed long ago: the wait only ties the registers)
        DKV_WAIT_A(0, 1) DKV_CA(3, 1)
        DKV_FRAGS()
        DKV_RT(1, 0, 0) DKV_RT(1, 1, 1)
        DKV_WAIT_T(NRT, 0) DKV_MB(0, 0)
        DKV_RT(1, 2, 0)
        DKV_WAIT_T(NRT, 1) DKV_MB(1, 1)
        DKV_RT(1, 3, 1)
        DKV_WAIT_T(NRT, 0) DKV_MB(2, 0)
        DKV_WAIT_T(0, 1) DKV_MB(3, 1)
#undef DKV_RA
#undef DKV_WAIT_A
#undef DKV_CA
#undef DKV_FRAGS
#undef DKV_RT
#undef DKV_WAIT_T
#undef DKV_MB
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = key0 + kt * 16 + c;
        if (key >= N) continue;
        // column block of dK / dV inside the packed gradient row: (dq, dk, dv) or, with vfirst, (dq, dv, dk)
        T* rk = dqkv + ((long)b * N + key) * 3 * H * HD + (long)((vfirst ? 2 : 1) * H + h) * HD;
        T* rv = dqkv + ((long)b * N + key) * 3 * H * HD + (long)((vfirst ? 1 : 2) * H + h) * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            if (DK) store4<T>(rk + dt * 16 + g * 4, dk[dt][kt] * scale);
            store4<T>(rv + dt * 16 + g * 4, dv[dt][kt]);
        }
    }
}

// ------------------------------------------------------------------------------------------ teacher cross-view maps
// VGGT teacher -> distillation target.  The reference's global blocks return, per head, the two
// cross-view softmax maps  softmax(q[prefix:N/2] k[N/2+prefix:]^T * scale / temperature)  and the mirrored one
// (vggt/layers/attention.py:51-85), i.e. [2B, H, n, n] fp32 per block (480 MB per pair at n = 1369, H = 16), which are then
// averaged over heads (src/finetune_timm_vggt.py:390-392) and over the selected blocks (vggt/models/aggregator.py:273).
// Here only the averaged [2B, n, n] map ever exists: pass 1 = flash-style row statistics per (direction, head, query)
// (log2 domain), pass 2 = one block per (128 queries x 64 keys) output tile loops over the heads, recomputes the
// S^T tile on the MFMA and accumulates exp2(s - lse) in registers; `weight` (= 1 / (H * blocks)) and `accumulate`
// fold the layer mean into the same buffer.  q, k: [B, H, N, 64] (after q/k-norm and RoPE).
template <typename T>
__global__ __launch_bounds__(256, 2) void cva_stats_kernel(const T* q, const T* k, float* lse2, int B, int H, int N, int prefix,
                                                           float c2) {
    constexpr int NF = AT<T>::NF, ROWB = AT<T>::ROWB;
    typedef typename Mma<T>::Frag Frag;
    __shared__ __attribute__((aligned(16))) char sK[64 * ROWB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const int n = N / 2 - prefix, dir = blockIdx.z / B, b = blockIdx.z % B, h = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const long ld_b = (long)HD * sizeof(T);
    const char* qb = (const char*)q + (((long)b * H + h) * N + (dir ? N / 2 + prefix : prefix)) * ld_b;
    const char* kb = (const char*)k + (((long)b * H + h) * N + (dir ? prefix : N / 2 + prefix)) * ld_b;
    Frag qf[2][NF];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = q0 + qt * 16 + c;
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            Frag z = {};
            qf[qt][u] = qi < n ? load_nfrag<T>(qb + (long)qi * ld_b, u, g) : z;
        }
    }
    float m[2] = {-1e30f, -1e30f}, l[2] = {0.f, 0.f};
    TileRegs<T> rk;
    tile_load<T>(rk, kb, ld_b, 0, n);
    for (int k0 = 0; k0 < n; k0 += 64) {
        __syncthreads();
        tile_store<T, true, false>(rk, sK, nullptr);
        __syncthreads();
        if (k0 + 64 < n) tile_load<T>(rk, kb, ld_b, k0 + 64, n);
        f32x4 s[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            Frag kf[NF];
#pragma unroll
            for (int u = 0; u < NF; ++u) kf[u] = lds_nfrag<T>(sK, kt * 16 + c, u, g);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < NF; ++u) a = Mma<T>::mma(kf[u], qf[qt][u], a);
                s[qt][kt] = a;
            }
        }
        const bool tail = k0 + 64 > n;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float tmax = -1e30f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (tail && k0 + kt * 16 + g * 4 + r >= n) s[qt][kt][r] = -1e30f;
                    tmax = fmaxf(tmax, s[qt][kt][r]);
                }
            tmax = quad_rows_max(tmax);
            const float mn = fmaxf(m[qt], tmax * c2);
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ps += fast_exp2(fmaf(s[qt][kt][r], c2, -mn));
            l[qt] = l[qt] * fast_exp2(m[qt] - mn) + ps;
            m[qt] = mn;
        }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = q0 + qt * 16 + c;
        l[qt] += __shfl_xor(l[qt], 16, 64);
        l[qt] += __shfl_xor(l[qt], 32, 64);
        if (qi < n && g == 0) lse2[(((long)dir * B + b) * H + h) * n + qi] = m[qt] + log2f(l[qt]);
    }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void cva_emit_kernel(const T* q, const T* k, const float* lse2, float* out, int B, int H, int N,
                                                          int prefix, float c2, float weight, int accumulate) {
    constexpr int NF = AT<T>::NF, ROWB = AT<T>::ROWB;
    typedef typename Mma<T>::Frag Frag;
    __shared__ __attribute__((aligned(16))) char sK[64 * ROWB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const int n = N / 2 - prefix, dir = blockIdx.z / B, b = blockIdx.z % B;
    const int q0 = blockIdx.x * 128 + wave * 32, k0 = blockIdx.y * 64;
    const long ld_b = (long)HD * sizeof(T);
    const long head_b = (long)N * ld_b;
    const char* qb = (const char*)q + (((long)b * H) * N + (dir ? N / 2 + prefix : prefix)) * ld_b;
    const char* kb = (const char*)k + (((long)b * H) * N + (dir ? prefix : N / 2 + prefix)) * ld_b;
    const float* lb = lse2 + (((long)dir * B + b) * H) * n;
    f32x4 acc[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) acc[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    TileRegs<T> rk;
    tile_load<T>(rk, kb, ld_b, k0, n);
    for (int h = 0; h < H; ++h) {
        __syncthreads();
        tile_store<T, true, false>(rk, sK, nullptr);
        __syncthreads();
        if (h + 1 < H) tile_load<T>(rk, kb + (long)(h + 1) * head_b, ld_b, k0, n);
        Frag qf[2][NF];
        float ls[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int qi = q0 + qt * 16 + c;
            ls[qt] = qi < n ? lb[(long)h * n + qi] : 0.f;
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                Frag z = {};
                qf[qt][u] = qi < n ? load_nfrag<T>(qb + (long)h * head_b + (long)qi * ld_b, u, g) : z;
            }
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            Frag kf[NF];
#pragma unroll
            for (int u = 0; u < NF; ++u) kf[u] = lds_nfrag<T>(sK, kt * 16 + c, u, g);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < NF; ++u) a = Mma<T>::mma(kf[u], qf[qt][u], a);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[qt][kt][r] += fast_exp2(fmaf(a[r], c2, -ls[qt]));
            }
        }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = q0 + qt * 16 + c;
        if (qi >= n) continue;
        float* orow = out + (((long)dir * B + b) * n + qi) * n;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kj = k0 + kt * 16 + g * 4 + r;
                if (kj < n) orow[kj] = weight * acc[qt][kt][r] + (accumulate ? orow[kj] : 0.f);
            }
    }
}

extern "C" size_t gd_cross_view_attn_workspace_bytes(int B, int H, int N, int prefix) {
    const int n = N / 2 - prefix;
    return n > 0 ? (size_t)2 * B * H * n * sizeof(float) : 0;
}

extern "C" int gd_cross_view_attn(const void* q, const void* k, float* out, int B, int H, int N, int prefix, int head_dim,
                                  float scale, float temperature, float weight, int accumulate, int dtype, void* workspace,
                                  void* stream) {
    GD_REQUIRE(B > 0 && H > 0 && N > 0 && N % 2 == 0 && prefix >= 0 && N / 2 - prefix > 0,
               "gd_cross_view_attn: bad shape B=%d H=%d N=%d prefix=%d", B, H, N, prefix);
    GD_REQUIRE(head_dim == HD, "gd_cross_view_attn: head_dim must be 64 (got %d)", head_dim);
    GD_REQUIRE(dtype == GD_F32 || dtype == GD_BF16, "gd_cross_view_attn: bad dtype %d", dtype);
    GD_REQUIRE(temperature > 0.f, "gd_cross_view_attn: temperature must be positive");
    GD_REQUIRE(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && workspace != nullptr,
               "gd_cross_view_attn: q, k must be 16-byte aligned, workspace non-null");
    const int n = N / 2 - prefix;
    const float c2 = scale / temperature * 1.4426950408889634f;
    float* lse2 = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g1(gd_cdiv(n, 128), H, 2 * B), g2(gd_cdiv(n, 128), gd_cdiv(n, 64), 2 * B);
    if (dtype == GD_BF16) {
        hipLaunchKernelGGL(cva_stats_kernel<bf16>, g1, dim3(256), 0, s, (const bf16*)q, (const bf16*)k, lse2, B, H, N, prefix, c2);
        hipLaunchKernelGGL(cva_emit_kernel<bf16>, g2, dim3(256), 0, s, (const bf16*)q, (const bf16*)k, lse2, out, B, H, N, prefix, c2, weight, accumulate);
    } else {
        hipLaunchKernelGGL(cva_stats_kernel<float>, g1, dim3(256), 0, s, (const float*)q, (const float*)k, lse2, B, H, N, prefix, c2);
        hipLaunchKernelGGL(cva_emit_kernel<float>, g2, dim3(256), 0, s, (const float*)q, (const float*)k, lse2, out, B, H, N, prefix, c2, weight, accumulate);
    }
    GD_LAUNCH_OK();
    return 0;
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int gd_attention_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int head_dim, float scale,
                                int dtype, void* stream) {
    GD_REQUIRE(B > 0 && N > 0 && H > 0, "gd_attention_fwd: bad shape B=%d N=%d H=%d", B, N, H);
    GD_REQUIRE(head_dim == HD, "gd_attention_fwd: head_dim must be 64 (got %d)", head_dim);
    GD_REQUIRE((long)N * 3 * H * HD * 4 < (1L << 31), "gd_attention_fwd: one image's qkv rows must span < 2^31 bytes (32-bit tile offsets): N=%d H=%d", N, H);
    GD_REQUIRE(dtype == GD_F32 || dtype == GD_BF16 || dtype == GD_F32X3 || dtype == GD_F16, "gd_attention_fwd: bad dtype %d", dtype);
    GD_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)o & 15) == 0, "gd_attention_fwd: pointers must be 16-byte aligned");
    dim3 grid(gd_cdiv(N, 128), H, B);
    const int dma = gd_knobs().attn_dma;   // GD_ATTN_DMA=0: the register-staged forward kernel (A/B testing)
    if (dtype == GD_BF16 && dma)
        { const int ro = gd_knobs().attn_rot;
          hipLaunchKernelGGL(attn_fwd_dma_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv, (bf16*)o, lse, N, H, scale, ro); }
    else if (dtype == GD_F16 && dma)        // tf32h engine: fp16 q / k / v / p (TF32's significand), the bf16 kernel's layouts
        { const int ro = gd_knobs().attn_rot;
          hipLaunchKernelGGL(attn_fwd_dma_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)qkv, (f16*)o, lse, N, H, scale, ro); }
    else if (dtype == GD_F16)
        hipLaunchKernelGGL(attn_fwd_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)qkv, (f16*)o, lse, N, H, scale);
    else if (dtype == GD_BF16)
        hipLaunchKernelGGL(attn_fwd_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)qkv, (bf16*)o, lse, N, H, scale);
    else if (dtype == GD_F32X3)
        hipLaunchKernelGGL(attn_fwd_kernel<x3>, grid, dim3(256), 0, (hipStream_t)stream, (const x3*)qkv, (x3*)o, lse, N, H, scale);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)qkv, (float*)o, lse, N, H, scale);
    GD_LAUNCH_OK();
    return 0;
}

template <typename T>      // bf16 | f16
static void attn_bwd_launch16(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, float* delta_ws, int B, int N, int H,
                              float scale, int grad_order, bool no_dk, hipStream_t s) {
    dim3 grid(gd_cdiv(N, 128), H, B);
        if (gd_knobs().attn_dq_dma) hipLaunchKernelGGL(attn_bwd_dq_dma_kernel<T>, grid, dim3(256), 0, s, (const T*)qkv, (const T*)o, (const T*)dout, lse, delta_ws, (T*)dqkv, N, H, scale);
        else
        hipLaunchKernelGGL(attn_bwd_dq_kernel<T>, grid, dim3(256), 0, s, (const T*)qkv, (const T*)o, (const T*)dout, lse, delta_ws, (T*)dqkv, N, H, scale);
        // 128-key blocks of four waves (two blocks per CU, independent barriers, Q / dO tiles staged twice as often) when the last
        // 256-key block would be less than half full: N = 1370 pads to 1408 keys instead of 1536 (2.7 % instead of 10.8 %):
        // backward 1574 -> 1514 us at 64 x 12 x 1370; at N = 6401 (long sweeps, 0.4 % vs 2 % padding) the 8-wave form is 2 % faster.
        // GD_ATTN_DKV_NW = 4 | 8 forces one form.
        const int dkv_env = gd_knobs().attn_dkv_nw;
        const int tail = N % 256;
        const int dkv_nw = dkv_env ? dkv_env : ((N < 4096 && tail > 0 && tail <= 128) ? 4 : 8);
        const bool dkv_dma = gd_knobs().attn_dkv_dma != 0;      // GD_ATTN_DKV_DMA=0: the register-staged kernels (A/B)
        if (no_dk && dkv_nw == 4 && dkv_dma)
            hipLaunchKernelGGL((attn_bwd_dkv_dma_kernel<T, false>), dim3(gd_cdiv(N, 128), H, B), dim3(256), 0, s, (const T*)qkv, (const T*)dout, delta_ws, (T*)dqkv, N, H, scale, grad_order);
        else if (dkv_nw == 4 && dkv_dma)
            hipLaunchKernelGGL((attn_bwd_dkv_dma_kernel<T, true>), dim3(gd_cdiv(N, 128), H, B), dim3(256), 0, s, (const T*)qkv, (const T*)dout, delta_ws, (T*)dqkv, N, H, scale, grad_order);
        else if (no_dk && dkv_nw == 4)
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, 4, false>), dim3(gd_cdiv(N, 128), H, B), dim3(256), 0, s, (const T*)qkv, (const T*)dout, lse, delta_ws, (T*)dqkv, N, H, scale, grad_order);
        else if (no_dk)
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, 8, false>), dim3(gd_cdiv(N, 256), H, B), dim3(512), 0, s, (const T*)qkv, (const T*)dout, lse, delta_ws, (T*)dqkv, N, H, scale, grad_order);
        else if (dkv_nw == 4)       // (two-wave 64-key blocks: 2213 us — the staging registers spill and every block re-stages all of Q / dO)
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, 4>), dim3(gd_cdiv(N, 128), H, B), dim3(256), 0, s, (const T*)qkv, (const T*)dout, lse, delta_ws, (T*)dqkv, N, H, scale, grad_order);
        else
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, 8>), dim3(gd_cdiv(N, 256), H, B), dim3(512), 0, s, (const T*)qkv, (const T*)dout, lse, delta_ws, (T*)dqkv, N, H, scale, grad_order);
}

extern "C" int gd_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv,
                                float* delta_ws, int B, int N, int H, int head_dim, float scale, int dtype,
                                int grad_order, void* stream) {
    GD_REQUIRE(grad_order >= 0 && grad_order <= 3, "gd_attention_bwd: grad_order bit 0: 0 = (dq, dk, dv), 1 = (dq, dv, dk); bit 1: dK not needed");
    const bool no_dk = (grad_order & 2) != 0;
    grad_order &= 1;
    GD_REQUIRE(B > 0 && N > 0 && H > 0, "gd_attention_bwd: bad shape B=%d N=%d H=%d", B, N, H);
    GD_REQUIRE(head_dim == HD, "gd_attention_bwd: head_dim must be 64 (got %d)", head_dim);
    GD_REQUIRE((long)N * 3 * H * HD * 4 < (1L << 31), "gd_attention_bwd: one image's qkv rows must span < 2^31 bytes (32-bit tile offsets): N=%d H=%d", N, H);
    GD_REQUIRE(dtype == GD_F32 || dtype == GD_BF16 || dtype == GD_F32X3 || dtype == GD_F16, "gd_attention_bwd: bad dtype %d", dtype);
    GD_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)dout & 15) == 0 && ((uintptr_t)dqkv & 15) == 0,
               "gd_attention_bwd: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    GD_REQUIRE(((uintptr_t)o & 15) == 0, "gd_attention_bwd: o must be 16-byte aligned");
    dim3 grid(gd_cdiv(N, 128), H, B);
    if (dtype == GD_BF16) {
        attn_bwd_launch16<bf16>(qkv, o, dout, lse, dqkv, delta_ws, B, N, H, scale, grad_order, no_dk, s);
    } else if (dtype == GD_F16) {      // tf32h engine: dout arrives times a power of two (the caller's gd_amax_scale), dqkv leaves with it
        attn_bwd_launch16<f16>(qkv, o, dout, lse, dqkv, delta_ws, B, N, H, scale, grad_order, no_dk, s);
    } else if (dtype == GD_F32X3) {
        hipLaunchKernelGGL(attn_bwd_dq_kernel<x3>, grid, dim3(256), 0, s, (const x3*)qkv, (const x3*)o, (const x3*)dout, lse, delta_ws, (x3*)dqkv, N, H, scale);
        // four-wave 128-key blocks with the whole register file per wave (the (hi, lo) fragments double the operand registers)
        if (no_dk)
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<x3, 4, false>), dim3(gd_cdiv(N, 128), H, B), dim3(256), 0, s, (const x3*)qkv, (const x3*)dout, lse, delta_ws, (x3*)dqkv, N, H, scale, grad_order);
        else
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<x3, 4>), dim3(gd_cdiv(N, 128), H, B), dim3(256), 0, s, (const x3*)qkv, (const x3*)dout, lse, delta_ws, (x3*)dqkv, N, H, scale, grad_order);
    } else {
        hipLaunchKernelGGL(attn_bwd_dq_kernel<float>, grid, dim3(256), 0, s, (const float*)qkv, (const float*)o, (const float*)dout, lse, delta_ws, (float*)dqkv, N, H, scale);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<float, 8>), dim3(gd_cdiv(N, 256), H, B), dim3(512), 0, s, (const float*)qkv, (const float*)dout, lse, delta_ws, (float*)dqkv, N, H, scale, grad_order);
    }
    GD_LAUNCH_OK();
    return 0;
}
