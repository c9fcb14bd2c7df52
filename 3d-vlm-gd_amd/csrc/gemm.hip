// MFMA GEMMs for the student ViT and the loss contractions (gfx950).
//
//   gemm_nt : C[M,N] = epilogue( alpha * A[M,K] . W[N,K]^T )        (both operands K-contiguous:
//             nn.Linear forward as is, and dX = dY . W with a pre-transposed frozen W)
//   gemm_tn : G[N,K] += alpha * Y[M,N]^T . X[M,K]                   (weight gradients; f32 MFMA,
//             fp32 atomics across M-chunks)
//
// Tiling (gemm_nt): 128x128 block tile, 4 waves as 2x2, each wave 64x64 = 4x4 MFMA 16x16 tiles,
// K step = 128 bytes of a row (64 bf16 / 32 f32), register-staged global->LDS with the next
// tile's loads in flight under the MFMAs, LDS rows padded 128->144 B.
#include "gd_common.h"
#include "gemm_tile.h"
#include <stdlib.h>

struct GemmNtParams {
    const void* A; const void* W; void* C;
    int M, N, K;
    long lda, ldw, ldc;              // row strides in elements
    long sA, sW, sC;                 // batch strides in elements (grid.y = batch)
    int c_dtype;                     // dtype of C / preact / dact_src / residual
    float alpha;
    const float* alpha_dev;          // device scalar multiplied into alpha (gd_gemm_nt_scaled), or null
    const float* copy_scale;         // gd_gemm_nt_copy16: device scalar multiplied into the fp16 copy of the result (p.preact), or null
    const float* bias;               // [N]
    const float* lora_t; const float* lora_b; int lora_rt;   // v += sum_r t[m,r] * b[r,n]
    void* preact; long ldp;          // store v before the activation
    int act;                         // 0 none, 1 GELU(erf), 2 ReLU
    int act_deriv;                   // preact receives GELU'(v) instead of v (C-ABI act = 3): the backward then gates with dact = 3
    const void* dact_src; long ldd; int dact;                 // v *= act'(src): 1 dGELU(pre), 2 (src > 0), 3 v *= src (stored derivative)
    const void* residual; long ldr;  // v += residual
    int accumulate;                  // v += C
    int vec_epilogue;                // every epilogue tensor is 16-byte aligned with 16-byte-multiple row strides
    int c_policy;                    // 0 plain C stores, 1 write-through (sc1) C stores (GD_GEMM_CSTORE, default 1)
    unsigned long long* probe;       // gd_gemm_phase_probe accumulators (device) in -DGD_GEMM_STAGE_PROBE builds, else null
    int group_m;                     // persistent kernel: tiles of an XCD's chunk walk GM row panels per W panel (GD_GEMM_GROUP_M; 1 = row-panel-major order)
    int k_rot;                       // persistent kernel: per-tile rotation of the K-step order, krot = (tn * k_rot + tm) % nk (GD_GEMM_KROT, default 1, 0 = off: +1.3 % on the step, in-step A/B 525.6 vs 518.8 pairs/s)
};

// Fallback: register-staged main loop (any K with K*elsize % 16 == 0, any alignment of rows), scalar epilogue.
template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_regstage_kernel(GemmNtParams p) {
    constexpr int BM = 128, BN = 128;
    __shared__ __attribute__((aligned(16))) char smem[GD_TILE_SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = wg / tiles_n, tn = wg % tiles_n;
    const long batch = blockIdx.y;

    const char* Ab = (const char*)p.A + batch * p.sA * (long)sizeof(T);
    const char* Wb = (const char*)p.W + batch * p.sW * (long)sizeof(T);
    f32x4 acc[4][4];
    mma_tile_128x128<T>(Ab, p.lda * (long)sizeof(T), p.M, Wb, p.ldw * (long)sizeof(T), p.N, p.K * (int)sizeof(T),
                        tm, tn, smem, acc);

    // ---- epilogue (C layout: row = 4*(lane>>4)+r, col = lane&15 inside each 16x16 tile) ----
    const int cdt = p.c_dtype;
    char* Cb = (char*)p.C + batch * p.sC * (long)gd_dtype_size(cdt);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tm * BM + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
            if (row >= p.M) continue;
            float lt[8];
            if (p.lora_t) {
#pragma unroll
                for (int q = 0; q < 8; ++q) lt[q] = q < p.lora_rt ? p.lora_t[(long)row * p.lora_rt + q] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = tn * BN + wn * 64 + j * 16 + (lane & 15);
                if (col >= p.N) continue;
                float v = (p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha) * acc[i][j][r];
                if (p.bias) v += p.bias[col];
                if (p.lora_t) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (q < p.lora_rt) v += lt[q] * p.lora_b[(long)q * p.N + col];
                }
                if (p.preact) st_rt(p.preact, (long)row * p.ldp + col, cdt, p.act_deriv ? dgelu_f(v) : v);
                if (p.act == 1) v = gelu_f(v);
                else if (p.act == 2) v = fmaxf(v, 0.f);
                if (p.dact == 1) v *= dgelu_f(ld_rt(p.dact_src, (long)row * p.ldd + col, cdt));
                else if (p.dact == 2) v = ld_rt(p.dact_src, (long)row * p.ldd + col, cdt) > 0.f ? v : 0.f;
                else if (p.dact == 3) v *= ld_rt(p.dact_src, (long)row * p.ldd + col, cdt);
                if (p.residual) v += ld_rt(p.residual, (long)row * p.ldr + col, cdt);
                if (p.accumulate) v += ld_rt(Cb, (long)row * p.ldc + col, cdt);
                st_rt(Cb, (long)row * p.ldc + col, cdt, v);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// Main kernel: LDS-DMA (global_load_lds_dwordx4) staging into a 2-deep LDS ring, one barrier per K-step, the
// next tile's DMA in flight under the MFMAs; XOR-swizzled LDS image (swizzle applied to the per-lane SOURCE
// address, LDS destination stays lane-linear) so the ds_read_b128 fragment reads are bank-conflict-free;
// epilogue staged through LDS so that every global access of the epilogue (C, preact, residual, dact_src) is a
// 16-byte row-coalesced vector.  Requires K*elsize % 128 == 0.
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ void ld8_rt(const void* p, long i, int dt, float (&o)[8]) {
    if (dt == GD_BF16) {
        const bf16x8 v = *(const bf16x8*)((const bf16*)p + i);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (float)v[k];
    } else if (dt == GD_F16) {
        const f16x8 v = *(const f16x8*)((const f16*)p + i);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (float)v[k];
    } else {
        const f32x4 a = *(const f32x4*)((const float*)p + i), b = *(const f32x4*)((const float*)p + i + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k] = a[k]; o[4 + k] = b[k]; }
    }
}
__device__ __forceinline__ void st8_rt(void* p, long i, int dt, const float (&v)[8]) {
    if (dt == GD_BF16) {
        *(bf16x8*)((bf16*)p + i) = bf16x8{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3], (bf16)v[4], (bf16)v[5], (bf16)v[6], (bf16)v[7]};
    } else if (dt == GD_F16) {
        const f16x8 h = f16_sat8(v);
        *(f16x8*)((f16*)p + i) = h;
    } else {
        *(f32x4*)((float*)p + i) = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)((float*)p + i + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
}

// Tile configurations:  NWM x NWN waves, each wave owns WMT x 4 MFMA tiles (16*WMT rows x 64 columns).
//   <2,2,4> : 128 x 128 block, 256 threads, 64 KB LDS ring (2 blocks / CU)      — skinny N / small problems
//   <2,4,8> : 256 x 256 block, 512 threads, 128 KB LDS ring (1 block / CU, 2 waves / SIMD); wave tile 128 x 64
//             => 12 fragment reads per 32 MFMAs instead of 8 per 16: less LDS traffic per FLOP.
__device__ __forceinline__ void st8_wt(__amdgpu_buffer_rsrc_t rs, int byte_off, int dt, const float (&v)[8]) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    if (dt == GD_BF16) {
        const bf16x8 b = bf16x8{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3], (bf16)v[4], (bf16)v[5], (bf16)v[6], (bf16)v[7]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, b), rs, byte_off, 0, 16);
    } else if (dt == GD_F16) {
        const f16x8 h = f16_sat8(v);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs, byte_off, 0, 16);
    } else {
        const f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), rs, byte_off, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, b), rs, byte_off + 16, 0, 16);
    }
}

template <typename T, int NWM, int NWN, int WMT>
__global__ __launch_bounds__(64 * NWM * NWN) void gemm_nt_kernel(GemmNtParams p) {
    constexpr int NW = NWM * NWN, NT = 64 * NW, BM = NWM * WMT * 16, BN = NWN * 64;
    constexpr int STAGE = (BM + BN) * 128, EPLD = BN + 4;
    static_assert(64 * EPLD * 4 <= 2 * STAGE, "epilogue staging must fit in the ring");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + (BM + BN) * 32];   // ring + f32 LoRA tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = wg / tiles_n, tn = wg % tiles_n;
    const long batch = blockIdx.y;
    const char* Ab = (const char*)p.A + batch * p.sA * (long)sizeof(T);
    const char* Wb = (const char*)p.W + batch * p.sW * (long)sizeof(T);
    f32x4 acc[WMT][4];
    const bool lora_mma = p.lora_t && p.lora_rt == 8 && p.vec_epilogue;
    if (lora_mma) {   // one 16-byte LDS-DMA per thread per operand (2*BM == 2*BN == NT lanes); clamped rows/cols are never stored
        static_assert(2 * BM == NT && 2 * BN == NT, "LoRA tile DMA assumes square tiles");
        char* lT = smem + 2 * STAGE;
        char* lB = lT + BM * 32;
        const int trow = min(tm * BM + (tid >> 1), p.M - 1);
        const int bk = tid / (BN / 4), bc = min(tn * BN + (tid % (BN / 4)) * 4, p.N - 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.lora_t + (long)trow * 8 + (tid & 1) * 4),
                                         (__attribute__((address_space(3))) void*)(lT + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.lora_b + (long)bk * p.N + bc),
                                         (__attribute__((address_space(3))) void*)(lB + wave * 1024), 16, 0, 0);
    }
    dma_mainloop<T, NWM, NWN, WMT>(Ab, p.lda * (long)sizeof(T), p.M, Wb, p.ldw * (long)sizeof(T), p.N,
                                   p.K * (int)sizeof(T) / 128, tm, tn, smem, acc);
    const int fr = lane & 15, g = lane >> 4;

    // ---- LoRA rank update  acc += (T / alpha) . B  as ONE extra MFMA K-chunk per accumulator tile: lane (g, fr) owns
    // k = KPL*g + j of the chunk, so only k < 8 is populated.  The f32 T tile [BM][8] and B tile [8][BN] were DMA'd to
    // the LDS tail before the main loop (no registers, latency hidden); fragments are converted on the way out.
    // (The scalar form, 8 FMAs per output element on the VALU in the epilogue, cost 35 % of the K = 768 QKV GEMM.)
    if (lora_mma) {
        typedef typename Mma<T>::Frag Frag;
        constexpr int KPL = sizeof(Frag) / sizeof(T);
        const float* lT = (const float*)(smem + 2 * STAGE);
        const float* lB = lT + BM * 8;
        const float ia = 1.0f / (p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha);
        const bool live = KPL * g < 8;
        Frag bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                const float x = lB[((KPL * g + k) & 7) * BN + wn * 64 + j * 16 + fr];
                bf[j][k] = (T)(live ? x : 0.f);
            }
#pragma unroll
        for (int i = 0; i < WMT; ++i) {
            const float* tr = lT + (wm * WMT * 16 + i * 16 + fr) * 8 + ((KPL * g) & 7);
            Frag af;
#pragma unroll
            for (int k = 0; k < KPL; ++k) af[k] = (T)(live ? ia * tr[k] : 0.f);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(af, bf[j], acc[i][j]);
        }
    }

    // ---- epilogue: 64-row passes through LDS (fp32), then 8-column vectors per thread ----
    const float alpha = p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha;
    const int cdt = p.c_dtype;
    char* Cb = (char*)p.C + batch * p.sC * (long)gd_dtype_size(cdt);
    float* se = (float*)smem;  // [64][EPLD]
    const bool vec = p.vec_epilogue != 0;
    // C leaves through write-through (sc1) stores: a plain store keeps its line in the XCD's L2, and one tile round of C
    // (32 blocks x 128 KB) would flush the 4 MB L2 that the operand panels are being re-read from.
    const int csz = gd_dtype_size(cdt);
    const long tile_row0 = (long)tm * BM * p.ldc * csz;
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(Cb + tile_row0), (short)0, (int)min((long)0x7fffffff, (long)BM * p.ldc * csz), 0x00020000);
    // dact_src / residual tiles of a bf16 output are fetched ONE PASS AHEAD (16 B per item) so their HBM latency hides
    // under the staging barrier and the previous pass's arithmetic instead of being paid four times per tile.
    // One shared slot array: it carries dact_src when there is one, else the residual (a call with both reads the
    // residual directly) — 16 registers, the 256 x 256 tile has no more to spare next to its 128 accumulators.
    const bool pre = vec && cdt == GD_BF16 && (p.dact || p.residual);
    const bool pre_d = pre && p.dact, pre_r = pre && !p.dact;
    const bf16* side_src = (const bf16*)(p.dact ? p.dact_src : p.residual);
    const long side_ld = p.dact ? p.ldd : p.ldr;
    uint4 sd[4] = {};
    auto side_load = [&](int ps, int q) {
        const int item = tid + NT * q;
        const int lr = item / (BN / 8), cc = (item % (BN / 8)) * 8;
        const int row = tm * BM + ps * 64 + lr, col0 = tn * BN + cc;
        if (row < p.M && col0 + 8 <= p.N) sd[q] = *(const uint4*)(side_src + (long)row * side_ld + col0);
    };
    if (pre) {
#pragma unroll
        for (int q = 0; q < 4; ++q) side_load(0, q);
    }
#pragma unroll
    for (int ps = 0; ps < BM / 64; ++ps) {
        constexpr int RPW = WMT * 16;                       // rows per wave
        const int owner = (ps * 64) / RPW, i0 = ((ps * 64) % RPW) / 16;
        if (wm == owner) {
            // static register index i, runtime predicate: acc must never be dynamically indexed (it would go to scratch)
#pragma unroll
            for (int i = 0; i < WMT; ++i) {
                if ((i >> 2) != (i0 >> 2)) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        se[((i & 3) * 16 + g * 4 + r) * EPLD + wn * 64 + j * 16 + fr] = acc[i][j][r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int item = tid + NT * q;
            const int lr = item / (BN / 8), cc = (item % (BN / 8)) * 8;
            const int row = tm * BM + ps * 64 + lr, col0 = tn * BN + cc;
            if (row < p.M && col0 < p.N) {
            float v[8];
            {
                const f32x4 x0 = *(const f32x4*)(se + lr * EPLD + cc), x1 = *(const f32x4*)(se + lr * EPLD + cc + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[k] = alpha * x0[k]; v[4 + k] = alpha * x1[k]; }
            }
            if (vec && col0 + 8 <= p.N) {
                if (p.bias) {
                    const f32x4 b0 = *(const f32x4*)(p.bias + col0), b1 = *(const f32x4*)(p.bias + col0 + 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[k] += b0[k]; v[4 + k] += b1[k]; }
                }
                if (p.lora_t && !lora_mma) {
                    for (int qq = 0; qq < p.lora_rt; ++qq) {
                        const float tq = p.lora_t[(long)row * p.lora_rt + qq];
                        const f32x4 b0 = *(const f32x4*)(p.lora_b + (long)qq * p.N + col0),
                                    b1 = *(const f32x4*)(p.lora_b + (long)qq * p.N + col0 + 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) { v[k] += tq * b0[k]; v[4 + k] += tq * b1[k]; }
                    }
                }
                if (p.preact) {
                    if (p.act_deriv) {
                        float dv[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) dv[k] = cdt == GD_BF16 ? dgelu_fast(v[k]) : dgelu_f(v[k]);
                        st8_rt(p.preact, (long)row * p.ldp + col0, cdt, dv);
                    } else st8_rt(p.preact, (long)row * p.ldp + col0, cdt, v);
                }
                if (p.act == 1) {
                    if (cdt == GD_BF16) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = gelu_fast(v[k]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = gelu_f(v[k]);
                    }
                } else if (p.act == 2) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
                }
                if (p.dact) {
                    float s[8];
                    if (pre_d) {
                        const bf16x8 t = __builtin_bit_cast(bf16x8, sd[q]);
#pragma unroll
                        for (int k = 0; k < 8; ++k) s[k] = (float)t[k];
                    } else ld8_rt(p.dact_src, (long)row * p.ldd + col0, cdt, s);
                    if (p.dact == 3) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] *= s[k];
                    } else if (p.dact == 1 && cdt == GD_BF16) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] *= dgelu_fast(s[k]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = p.dact == 1 ? v[k] * dgelu_f(s[k]) : (s[k] > 0.f ? v[k] : 0.f);
                    }
                }
                if (p.residual) {
                    float s[8];
                    if (pre_r) {
                        const bf16x8 t = __builtin_bit_cast(bf16x8, sd[q]);
#pragma unroll
                        for (int k = 0; k < 8; ++k) s[k] = (float)t[k];
                    } else ld8_rt(p.residual, (long)row * p.ldr + col0, cdt, s);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += s[k];
                }
                if (p.accumulate) {
                    float s[8];
                    ld8_rt(Cb, (long)row * p.ldc + col0, cdt, s);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += s[k];
                }
                if (p.c_policy == 0) st8_rt(Cb, (long)row * p.ldc + col0, cdt, v);
                else st8_wt(crs, (int)((long)(ps * 64 + lr) * p.ldc * csz + (long)col0 * csz), cdt, v);
            } else {
                for (int k = 0; k < 8; ++k) {
                    const int col = col0 + k;
                    if (col >= p.N) break;
                    float x = v[k];
                    if (p.bias) x += p.bias[col];
                    if (p.lora_t && !lora_mma)   // (otherwise already in acc via the MFMA rank update)
                        for (int qq = 0; qq < p.lora_rt; ++qq) x += p.lora_t[(long)row * p.lora_rt + qq] * p.lora_b[(long)qq * p.N + col];
                    if (p.preact) st_rt(p.preact, (long)row * p.ldp + col, cdt, p.act_deriv ? dgelu_f(x) : x);
                    if (p.act == 1) x = gelu_f(x);
                    else if (p.act == 2) x = fmaxf(x, 0.f);
                    if (p.dact == 1) x *= dgelu_f(ld_rt(p.dact_src, (long)row * p.ldd + col, cdt));
                    else if (p.dact == 2) x = ld_rt(p.dact_src, (long)row * p.ldd + col, cdt) > 0.f ? x : 0.f;
                    else if (p.dact == 3) x *= ld_rt(p.dact_src, (long)row * p.ldd + col, cdt);
                    if (p.residual) x += ld_rt(p.residual, (long)row * p.ldr + col, cdt);
                    if (p.accumulate) x += ld_rt(Cb, (long)row * p.ldc + col, cdt);
                    st_rt(Cb, (long)row * p.ldc + col, cdt, x);
                }
            }
            }
            if (pre && ps + 1 < BM / 64) side_load(ps + 1, q);   // refill the slot just consumed
        }
        __syncthreads();
    }
}

#include "gemm_persist.h"
#ifdef GD_GEMM_EXPERIMENT32
#include "gemm_persist32.h"      // tools/experiments (make FLAGS+="-DGD_GEMM_EXPERIMENT32 -I../../tools/experiments")
#endif

// Phase probe of the persistent kernel: exists only in -DGD_GEMM_STAGE_PROBE builds (its s_memtime reads cost 20 % even unarmed,
// and its accumulator is process-wide state); the shipped library has neither the device code nor the global, and the entry
// point says so.
#ifdef GD_GEMM_STAGE_PROBE
static unsigned long long* g_probe = nullptr;
static unsigned long long* gd_probe_buffer() { return g_probe; }
extern "C" int gd_gemm_phase_probe(int enable, unsigned long long* out6) {
    if (out6 && g_probe) {
        hipDeviceSynchronize();
        hipMemcpy(out6, g_probe, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost);   // wait, main, epilogue, tiles, dma-issue, stage-barrier wait
    }
    if (enable && !g_probe) {
        GD_REQUIRE(hipMalloc(&g_probe, 8 * sizeof(unsigned long long)) == hipSuccess, "gd_gemm_phase_probe: hipMalloc failed");
    }
    if (enable) hipMemset(g_probe, 0, 8 * sizeof(unsigned long long));
    if (!enable && g_probe) { hipFree(g_probe); g_probe = nullptr; }
    return 0;
}
#else
static unsigned long long* gd_probe_buffer() { return nullptr; }
extern "C" int gd_gemm_phase_probe(int enable, unsigned long long* out6) {
    (void)out6;
    if (!enable) return 0;
    gd_set_error("gd_gemm_phase_probe: this library was built without -DGD_GEMM_STAGE_PROBE (debug builds only)");
    return -1;
}
#endif

// ------------------------------------------------------------------------------------------
// gemm_tn: G[N,K] += alpha * sum_m Y[m,n] X[m,k];  64x64 output tile per block, f32 MFMA 16x16x4,
// operands converted to f32 while staging (so bf16 activations give fp32-accumulated weight grads).
// ------------------------------------------------------------------------------------------
struct GemmTnParams {
    const void* Y; const void* X; float* G;
    int M, N, K;
    long ldy, ldx, ldg;
    int y_dtype, x_dtype;
    int mchunk;
    float alpha;
    const float* alpha_dev;   // device scalar multiplied into alpha (gd_gemm_tn_scaled), or null
    long sY, sX, sG;   // batch strides in elements (grid.z = batch)
    int anat;          // GD_GEMM_ANAT (timing experiments): 4 = no closing atomics (results are not written)
};

__device__ __forceinline__ void load8_as_f32(const void* base, long off, int dt, bool ok, float (&o)[8]) {
    if (!ok) {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0.f;
        return;
    }
    if (dt == GD_BF16) {
        bf16x8 v = *(const bf16x8*)((const bf16*)base + off);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
    } else if (dt == GD_F16) {
        f16x8 v = *(const f16x8*)((const f16*)base + off);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
    } else {
        const f32x4* q = (const f32x4*)((const float*)base + off);
        f32x4 a = q[0], b = q[1];
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
    }
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnParams p) {
    constexpr int TN_ = 64, TK_ = 64, MR = 32, LD = 80;  // LD: padded row stride (floats)
    __shared__ __attribute__((aligned(16))) float sY[MR * LD];
    __shared__ __attribute__((aligned(16))) float sX[MR * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_k = (p.K + TK_ - 1) / TK_;
    const int tn = blockIdx.x / tiles_k, tk = blockIdx.x % tiles_k;
    const int m_begin = blockIdx.y * p.mchunk;
    const int m_end = min(p.M, m_begin + p.mchunk);
    const int wy = wave >> 1, wx = wave & 1;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int srow = tid >> 3, scol = (tid & 7) * 8;
    const int yn = tn * TN_ + scol, xk = tk * TK_ + scol;
    float ry[8], rx[8];
    p.Y = (const char*)p.Y + (long)blockIdx.z * p.sY * gd_dtype_size(p.y_dtype);
    p.X = (const char*)p.X + (long)blockIdx.z * p.sX * gd_dtype_size(p.x_dtype);
    p.G += (long)blockIdx.z * p.sG;
    auto gload = [&](int m0) {
        const int m = m0 + srow;
        load8_as_f32(p.Y, (long)m * p.ldy + yn, p.y_dtype, m < m_end && yn < p.N, ry);
        load8_as_f32(p.X, (long)m * p.ldx + xk, p.x_dtype, m < m_end && xk < p.K, rx);
    };
    gload(m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += MR) {
        *(f32x4*)(sY + srow * LD + scol) = f32x4{ry[0], ry[1], ry[2], ry[3]};
        *(f32x4*)(sY + srow * LD + scol + 4) = f32x4{ry[4], ry[5], ry[6], ry[7]};
        *(f32x4*)(sX + srow * LD + scol) = f32x4{rx[0], rx[1], rx[2], rx[3]};
        *(f32x4*)(sX + srow * LD + scol + 4) = f32x4{rx[4], rx[5], rx[6], rx[7]};
        __syncthreads();
        if (m0 + MR < m_end) gload(m0 + MR);
#pragma unroll
        for (int s = 0; s < MR / 4; ++s) {
            const int mr = s * 4 + (lane >> 4);
            float a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = sY[mr * LD + wy * 32 + t * 16 + (lane & 15)];
                b[t] = sX[mr * LD + wx * 32 + t * 16 + (lane & 15)];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = tn * TN_ + wy * 32 + i * 16 + (lane >> 4) * 4 + r;
                const int k = tk * TK_ + wx * 32 + j * 16 + (lane & 15);
                if (n < p.N && k < p.K) atomicAdd(p.G + (long)n * p.ldg + k, (p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha) * acc[i][j][r]);
            }
}

// ------------------------------------------------------------------------------------------
// gemm_tn for bf16 operands on the bf16 MFMA: G[N,K] += alpha * Y[M,N]^T X[M,K].  Both MFMA operands contract over
// the tile ROW index (m), i.e. both are "transposed" reads of row-major tiles: served by ds_read_b64_tr_b16 from
// natural LDS tiles Y[m][n], X[m][k] (same k-slot permutation on both sides).  128 x 128 output tile, 4 waves 2x2,
// 64-row m stages, register-prefetched; M split across blockIdx.y with fp32 atomics.
// Round-3 measurements at 87 680 x 64 x 768 (adapter weight gradients of the tf32h step, 768 blocks): 89-101 us, of which the closing atomics are 35
// (GD_GEMM_ANAT=4 takes them out: 53-67 us) — 6.3 M lane atomics = 393 K sixteen-lane line operations, the same ~12 G line-ops/s the skinny kernel's
// closing atomics run at.  Two restructurings were built and were SLOWER: a second stage of register prefetch (occupancy 3 -> 2 blocks per SIMD set:
// 116-122 us) and three row groups per block in lockstep on block-wide barriers with a third of the chunks (120-126 us; 473 vs 303 us at N = K = 768)
// — independent blocks drift apart and hide each other's load round trips, lockstep groups all wait at once.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int rowb, int col0, int u, int lane) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const char* a0 = tile + (32 * u + 4 * g + q) * rowb + (col0 + 4 * pp) * 2;
    const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 16 * rowb));
    const s16x8 z = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    return __builtin_bit_cast(bf16x8, z);
}

// YF32 / XF32 (tf32h engine, T = f16): that operand is fp32 in memory and is rounded to fp16 on its way into LDS — a gradient Y times 1 / *alpha_dev
// (the scale its consumer undoes: alpha_dev = 1 / s) — so that a weight gradient contracts an fp32 tensor without a separate cast pass.
template <typename T, bool YF32 = false, bool XF32 = false>       // bf16 | f16
__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(GemmTnParams p) {
    constexpr int ROWB = 272;   // 128 bf16 + 16 B pad
    __shared__ __attribute__((aligned(16))) char smem_tn[2 * 64 * ROWB];      // the two stage tiles; the closing exchange reuses them as one 32 KB tile
    char* const sY = smem_tn;
    char* const sX = smem_tn + 64 * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 1, wx = wave & 1;
    const int tiles_k = (p.K + 127) / 128, tiles_n = (p.N + 127) / 128;
    // the (tn, tk) blocks of one M-chunk re-read its Y column blocks (tiles_k times) and X column blocks (tiles_n times):
    // each XCD gets a contiguous run of tiles, tn fastest, so an X block is fetched by one L2 and the Y blocks by few
    const int t = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    const int tile = t % (int)gridDim.x, mc = t / (int)gridDim.x;
    const int tn = tile % tiles_n, tk = tile / tiles_n;
    const int m_begin = mc * p.mchunk, m_end = min(p.M, m_begin + p.mchunk);
    typedef typename std::conditional<YF32, float, T>::type TY;
    typedef typename std::conditional<XF32, float, T>::type TX;
    const TY* Y = (const TY*)p.Y + (long)blockIdx.z * p.sY;
    const TX* X = (const TX*)p.X + (long)blockIdx.z * p.sX;
    float* G = p.G + (long)blockIdx.z * p.sG;
    typedef typename Mma<T>::Frag Frag;
    const float alpha = p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha;
    const float ys = (YF32 && p.alpha_dev) ? 1.0f / *p.alpha_dev : 1.0f;
    auto ld8 = [&](const auto* q, float sc) __attribute__((always_inline)) {      // 8 elements -> one 16-byte chunk of T
        typedef typename std::remove_cv<typename std::remove_pointer<decltype(q)>::type>::type E;
        if constexpr (std::is_same<E, float>::value) {
            const f32x4 a = *(const f32x4*)q, b = *(const f32x4*)(q + 4);
            Frag h;
#pragma unroll
            for (int k = 0; k < 4; ++k) { h[k] = from_f32<T>(a[k] * sc); h[4 + k] = from_f32<T>(b[k] * sc); }
            return __builtin_bit_cast(uint4, h);
        } else {
            return *(const uint4*)q;
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ry[4], rx[4];   // 64 rows x 16 chunks (of 8 bf16) = 1024 chunks / 256 threads per operand
    auto gload = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = tid + 256 * i, rr = ch >> 4, cc = (ch & 15) * 8;
            const int m = m0 + rr, yn = tn * 128 + cc, xk = tk * 128 + cc;
            ry[i] = (m < m_end && yn < p.N) ? ld8(Y + (long)m * p.ldy + yn, ys) : make_uint4(0, 0, 0, 0);
            rx[i] = (m < m_end && xk < p.K) ? ld8(X + (long)m * p.ldx + xk, 1.0f) : make_uint4(0, 0, 0, 0);
        }
    };
    gload(m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = tid + 256 * i, rr = ch >> 4, cc = ch & 15;
            *(uint4*)(sY + rr * ROWB + cc * 16) = ry[i];
            *(uint4*)(sX + rr * ROWB + cc * 16) = rx[i];
        }
        __syncthreads();
        if (m0 + 64 < m_end) gload(m0 + 64);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            Frag a[4], b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[t] = __builtin_bit_cast(Frag, tr_frag(sY, ROWB, wy * 64 + t * 16, u, lane));
                b[t] = __builtin_bit_cast(Frag, tr_frag(sX, ROWB, wx * 64 + t * 16, u, lane));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(a[i], b[j], acc[i][j]);
        }
    }
    // closing atomics as FULL lines: a lane's accumulator layout gives 16 consecutive k per tile row and instruction (four 64-byte pieces); through
    // LDS, 64 rows at a time, every wave instruction adds 64 consecutive k of one row — half the line operations the memory side has to serialise
    static_assert(2 * 64 * ROWB >= 64 * 128 * 4, "a 64-row half of the fp32 tile fits the two stage tiles");
    float* red = (float*)smem_tn;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (tn * 128 + h * 64 >= p.N) break;      // (uniform) no row of this half inside G
        __syncthreads();
        if (wy == h) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(i * 16 + (lane >> 4) * 4 + r) * 128 + wx * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
        }
        __syncthreads();
#pragma unroll 4
        for (int e = 0; e < 32; ++e) {
            const int idx = e * 256 + tid, n = tn * 128 + h * 64 + (idx >> 7), k = tk * 128 + (idx & 127);
            if (n < p.N && k < p.K && p.anat != 4) atomicAdd(G + (long)n * p.ldg + k, alpha * red[idx]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// gemm_tn for a SKINNY left operand (N <= 8, f32): the LoRA weight gradients  G[8, K] += T[M, 8]^T X[M, K].
// HBM-bound (X is read once, 404 MB for the q/v LoRA of one block) and far too thin for MFMA tiles: the 64x64-tile
// kernel above ran it at 1.6 TB/s.  Here a thread owns 8 consecutive columns of X (one 16-byte load per row) and all
// 8 rows of G (64 fp32 accumulators); the T row is wave-uniform (scalar loads); rows are unrolled U deep and held RAW
// (16 bytes per row for the 16-bit types) until they are used.  A block is RG row groups of K/8 threads: each group
// streams its own slice of the block's row chunk, the groups' partials are summed through LDS, and ONE set of fp32
// atomics per block closes it.  With one group per block (rounds 1-2: 512 blocks of 2-5 waves) the kernel's time was
// 39 us + bytes / 14 TB/s: a chain of 11 dependent groups of loads per wave at about one wave per SIMD, then 512 blocks'
// worth of atomics onto the same 24-72 KB.  The row groups are there for the waves per SIMD (15-16 per CU, 8 KB in flight
// each) and for FEWER closing atomics (one block per CU), not for the arithmetic.
// <RG 8, NTK 128>: K <= 1024 (1024 threads);  <RG 3, NTK 320>: K <= 2560 (960 threads);  rows U = 8 deep (128 VGPRs).
// ------------------------------------------------------------------------------------------
template <typename TX, int RG, int U, int NTK>      // NTK: the largest row-group width (threads) this instantiation is launched with
__global__ __launch_bounds__(RG * NTK) void gemm_tn_skinny_kernel(GemmTnParams p) {
    __shared__ float sG[RG * NTK * 8];   // [row group][K slot]: one G row of the block at a time
    const int nthk = blockDim.x / RG;                              // threads per row group: a multiple of 64, so a wave is inside one group
    const int tid = threadIdx.x, rg = __builtin_amdgcn_readfirstlane(tid / nthk), kt = tid - rg * nthk, kc = kt * 8;
    const bool live = kc < p.K;
    const float* __restrict__ Y = (const float*)p.Y;
    const TX* __restrict__ X = (const TX*)p.X;
    const int per = p.mchunk / RG;                                 // (the host rounds mchunk to a multiple of RG * 16)
    const int m_begin = blockIdx.y * p.mchunk + rg * per, m_end = min(p.M, m_begin + per);
    float acc[8][8];
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[n][k] = 0.f;
    const int kcl = live ? kc : 0;   // threads past K read column 0 (their results are never written): no divergent load
    auto load_x = [&](int m, float (&x)[8]) __attribute__((always_inline)) {
        if constexpr (std::is_same<TX, f16>::value) {
            const f16x8 v = *(const f16x8*)((const f16*)X + (long)m * p.ldx + kcl);
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = (float)v[k];
        } else if (sizeof(TX) == 2) {
            const bf16x8 v = *(const bf16x8*)((const bf16*)X + (long)m * p.ldx + kcl);
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = (float)v[k];
        } else {
            const f32x4 a = *(const f32x4*)((const float*)X + (long)m * p.ldx + kcl), b = *(const f32x4*)((const float*)X + (long)m * p.ldx + kcl + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { x[k] = a[k]; x[4 + k] = b[k]; }
        }
    };
    int m0 = m_begin;
    for (; m0 + U <= m_end; m0 += U) {   // full groups: the U T rows are one contiguous, wave-uniform block (scalar loads, no branches)
        const float* __restrict__ trow = Y + (long)m0 * 8;
        if constexpr (sizeof(TX) == 2) {
            uint4 raw[U];
#pragma unroll
            for (int r = 0; r < U; ++r) raw[r] = *(const uint4*)((const char*)X + ((long)(m0 + r) * p.ldx + kcl) * 2);
#pragma unroll
            for (int r = 0; r < U; ++r) {
                if (r == 8) __builtin_amdgcn_sched_barrier(0);      // (the T values of the second half are fetched after the first half's: 128 live SGPRs spill)
                float x[8];
                if constexpr (std::is_same<TX, f16>::value) {
                    const f16x8 v = __builtin_bit_cast(f16x8, raw[r]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) x[k] = (float)v[k];
                } else {
                    const bf16x8 v = __builtin_bit_cast(bf16x8, raw[r]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) x[k] = (float)v[k];
                }
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    const float t = trow[r * 8 + n];
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[n][k] = fmaf(t, x[k], acc[n][k]);
                }
            }
        } else {
            static_assert(sizeof(TX) == 2 || U == 8, "f32 rows: 8 deep");
            float x[U][8];
#pragma unroll
            for (int r = 0; r < U; ++r) load_x(m0 + r, x[r]);
#pragma unroll
            for (int r = 0; r < U; ++r)
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    const float t = trow[r * 8 + n];
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[n][k] = fmaf(t, x[r][k], acc[n][k]);
                }
        }
    }
    for (; m0 < m_end; ++m0) {
        float x[8];
        load_x(m0, x);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const float t = Y[(long)m0 * 8 + n];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[n][k] = fmaf(t, x[k], acc[n][k]);
        }
    }
    const int W = nthk * 8;
    const float alpha = p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        if (n >= p.N) break;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) sG[rg * W + kc + k] = acc[n][k];
        __syncthreads();
        for (int k = tid; k < W; k += blockDim.x)
            if (k < p.K) {
                float v = sG[k];
#pragma unroll
                for (int g = 1; g < RG; ++g) v += sG[g * W + k];      // fixed order within the block
                if (p.anat != 4) atomicAdd(p.G + (long)n * p.ldg + k, alpha * v);
            }
    }
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
static bool gd_f32_big_tiles() { return gd_knobs().gemm_f32_big == 1; }   // f32: the 128x128 config at 2 blocks/CU measured 125-138 TF/s vs 97-120 for 256x256
static bool gd_force_small_tiles() { return gd_knobs().gemm_small_tiles == 1; }   // A/B switch for benchmarking the tile configurations

// Skinny NT GEMM, N <= 8 (the rank-2r LoRA projections t = LN(x) A^T and dt = dqkv B: M = 87 680, K = 768 / 2304): pure
// streaming of A.  The 128 x 128 tile kernel spends a full tile of MFMA and W traffic on 8 useful columns (28 TFLOP/s);
// here a wave owns 16 rows, streams them as MFMA A fragments straight from global memory (two batches of four K-steps in
// flight), W lives in LDS as [K-step][8 rows][64 B] (conflict-free linear fragment reads; columns 8..15 of the MFMA tile
// are fed zeros) and the 16 x 8 result leaves as two 16-byte stores per row.
#define SK_ROWS 8
template <typename TC, typename TA = bf16>      // TA: the 16-bit operand type (bf16 | f16)
__global__ __launch_bounds__(512) void gemm_nt_skinny_kernel(GemmNtParams p) {
    typedef typename Mma<TA>::Frag Frag;
    extern __shared__ __attribute__((aligned(16))) char sW[];       // K/32 x 512 B
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nks = p.K >> 5;
    const int row0 = (blockIdx.x * 8 + wave) * 16;
    const char* ar = (const char*)p.A + (long)min(row0 + c, p.M - 1) * p.lda * 2 + 16 * g;      // (rows past M re-read row M - 1: never written)
    Frag a0[4], a1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a0[k] = *(const Frag*)(ar + k * 64);      // the first batch of A is in flight under the W staging
    // W -> LDS: 16-byte chunk j of row r goes to K-step j/4, row r, sub-chunk j%4
    for (int q = tid; q < SK_ROWS * (p.K >> 3); q += 512) {
        const int r = q / (p.K >> 3), j = q % (p.K >> 3);
        uint4 v = {0u, 0u, 0u, 0u};
        if (r < p.N) v = *(const uint4*)((const TA*)p.W + (long)r * p.ldw + j * 8);
        *(uint4*)(sW + (j >> 2) * 512 + r * 64 + (j & 3) * 16) = v;
    }
    __syncthreads();
    if (row0 >= p.M) return;
    const char* wr = sW + (c & 7) * 64 + 16 * g;
    const bool wlive = c < SK_ROWS;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int nb = nks >> 2;                                        // batches of four K-steps (K % 128 == 0)
    for (int b = 0; b < nb; b += 2) {
        if (b + 1 < nb) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a1[k] = *(const Frag*)(ar + (b + 1) * 256 + k * 64);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            Frag w = {};
            if (wlive) w = *(const Frag*)(wr + (b * 4 + k) * 512);
            if (k & 1) acc1 = Mma<TA>::mma(a0[k], w, acc1); else acc0 = Mma<TA>::mma(a0[k], w, acc0);
        }
        if (b + 2 < nb) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a0[k] = *(const Frag*)(ar + (b + 2) * 256 + k * 64);
        }
        if (b + 1 < nb) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                Frag w = {};
                if (wlive) w = *(const Frag*)(wr + ((b + 1) * 4 + k) * 512);
                if (k & 1) acc1 = Mma<TA>::mma(a1[k], w, acc1); else acc0 = Mma<TA>::mma(a1[k], w, acc0);
            }
        }
    }
    // lane (g, c) holds rows 4g + r, column c
    if (c < p.N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 4 * g + r;
            if (row < p.M) ((TC*)p.C)[(long)row * p.ldc + c] = from_f32<TC>((acc0[r] + acc1[r]) * (p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha));
        }
    }
}

static int gemm_nt_impl(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc,
                        int batch, long sA, long sW, long sC, int ab_dtype, int c_dtype, float alpha, const float* alpha_dev,
                        const float* bias, const float* lora_t, const float* lora_b, int lora_rt, void* preact,
                        long ldp, int act, const void* dact_src, long ldd, int dact, const void* residual, long ldr,
                        int accumulate, void* stream, void* copy16 = nullptr, long ldc16 = 0, const float* copy_scale = nullptr) {
    if (copy16) {       // gd_gemm_nt_copy16: f32 C = ... + residual AND an fp16 copy of it (times *copy_scale): the persistent fp16-operand kernel only
        GD_REQUIRE(ab_dtype == GD_F16 && c_dtype == GD_F32 && residual && !preact && !dact_src && act == 0 && !accumulate && batch == 1 &&
                       ((uintptr_t)copy16 & 15) == 0 && (ldc16 * 2) % 16 == 0,
                   "gd_gemm_nt_copy16: fp16 operands, f32 C with a residual and no other epilogue tensor; copy16 16-byte aligned");
        preact = copy16; ldp = ldc16;
    }
    GD_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "gd_gemm_nt: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
    GD_REQUIRE(ab_dtype == GD_F32 || ab_dtype == GD_BF16 || ab_dtype == GD_F16, "gd_gemm_nt: bad ab_dtype %d", ab_dtype);
    GD_REQUIRE(c_dtype == GD_F32 || c_dtype == GD_BF16 || c_dtype == GD_F32X3 || c_dtype == GD_F16, "gd_gemm_nt: bad c_dtype %d", c_dtype);
    // fp16 operands (tf32h engine): C / preact / dact_src / residual are all f32, or all fp16 (c_dtype GD_F16: C is the operand of the next
    // product, saturated at +-65504; preact = GELU'(v) and dact_src are factors of an elementwise product, kept at fp16's 11 bits); bf16 operands never write fp16
    GD_REQUIRE((c_dtype != GD_F16 && (ab_dtype != GD_F16 || c_dtype == GD_F32)) || (ab_dtype == GD_F16 && c_dtype == GD_F16),
               "gd_gemm_nt: fp16 operands write f32 or fp16; fp16 results come from fp16 operands only (ab_dtype %d, c_dtype %d)", ab_dtype, c_dtype);
    GD_REQUIRE(c_dtype != GD_F32X3 || ab_dtype == GD_BF16, "gd_gemm_nt: split output takes bf16 (split) operands");
    // c_dtype GD_F32X3: C is the [hi | lo | hi] bf16 operand split of the f32 result (row stride ldc >= 3N bf16 elements; the A
    // operand of the next tf32x GEMM); preact / dact_src / residual are f32.  Persistent-kernel shapes only (checked below).
    const bool csplit = c_dtype == GD_F32X3;
    if (csplit) c_dtype = GD_F32;
    const int es = gd_dtype_size(ab_dtype);
    GD_REQUIRE((K * es) % 16 == 0 && (lda * es) % 16 == 0 && (ldw * es) % 16 == 0 && (sA * es) % 16 == 0 &&
                   (sW * es) % 16 == 0,
               "gd_gemm_nt: K (%d), lda (%ld), ldw (%ld) and batch strides must be multiples of 16 bytes", K, lda, ldw);
    GD_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "gd_gemm_nt: A and W must be 16-byte aligned");
    GD_REQUIRE(lora_rt >= 0 && lora_rt <= 8, "gd_gemm_nt: lora_rt %d > 8", lora_rt);
    GD_REQUIRE(!lora_t || alpha != 0.f, "gd_gemm_nt: the LoRA rank update needs alpha != 0");
    GD_REQUIRE(batch == 1 || (!bias && !lora_t && !preact && !dact_src && !residual),
               "gd_gemm_nt: batched calls take no epilogue tensors");
    GemmNtParams p;
    p.A = A; p.W = W; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc;
    p.sA = sA; p.sW = sW; p.sC = sC; p.c_dtype = c_dtype; p.alpha = alpha; p.alpha_dev = alpha_dev; p.copy_scale = copy_scale; p.bias = bias;
    p.lora_t = lora_t; p.lora_b = lora_b; p.lora_rt = lora_rt; p.preact = preact; p.ldp = ldp; p.act = act == 3 ? 1 : act; p.act_deriv = act == 3;
    if (copy16) preact = nullptr;      // (for the dispatch below: not a pre-activation store)
    p.dact_src = dact_src; p.ldd = ldd; p.dact = dact_src ? dact : 0; p.residual = residual; p.ldr = ldr;
    p.accumulate = accumulate;
    const int cs = gd_dtype_size(c_dtype), ccs = csplit ? 2 : cs;
    auto al = [&](const void* q, long ld) { return q == nullptr || ((((uintptr_t)q) & 15) == 0 && (ld * cs) % 16 == 0); };
    p.vec_epilogue = ((uintptr_t)C & 15) == 0 && (ldc * ccs) % 16 == 0 && (sC * ccs) % 16 == 0 && al(preact, ldp) && al(dact_src, ldd) && al(residual, ldr) &&
                     (bias == nullptr || ((uintptr_t)bias & 15) == 0) && (lora_b == nullptr || (((uintptr_t)lora_b & 15) == 0 && N % 4 == 0));
    dim3 grid(gd_cdiv(M, 128) * gd_cdiv(N, 128), batch);
    const bool dma = (K * es) % 128 == 0;
    { const int cp = gd_knobs().gemm_cstore;
      p.c_policy = (cp && (long)256 * ldc * ccs < 0x7fffffffL && !accumulate) ? cp : 0; }
    p.probe = gd_probe_buffer();
    p.k_rot = gd_knobs().gemm_krot;
    p.group_m = gd_knobs().gemm_group_m > 0 ? gd_knobs().gemm_group_m : 1;
    const bool big = dma && N >= 256 && M >= (batch > 1 ? gd_knobs().gemm_batch_big_m : 1024) && !gd_force_small_tiles();
    dim3 gridb(gd_cdiv(M, 256) * gd_cdiv(N, 256), batch);
    hipStream_t st = (hipStream_t)stream;
    const int persist = gd_knobs().gemm_persist, ncu = gd_knobs().ncu;
    if ((ab_dtype == GD_BF16 || (ab_dtype == GD_F16 && c_dtype == GD_F32)) && N <= SK_ROWS && M >= 4096 && batch == 1 && K % 128 == 0 && K <= 4096 && !bias && !lora_t && !preact &&
        act == 0 && !dact_src && !residual && !accumulate) {
        const size_t lds = (size_t)(K / 32) * 512;
        if (ab_dtype == GD_F16) hipLaunchKernelGGL((gemm_nt_skinny_kernel<float, f16>), dim3(gd_cdiv(M, 128)), dim3(512), lds, st, p);
        else if (c_dtype == GD_F32) hipLaunchKernelGGL(gemm_nt_skinny_kernel<float>, dim3(gd_cdiv(M, 128)), dim3(512), lds, st, p);
        else hipLaunchKernelGGL(gemm_nt_skinny_kernel<bf16>, dim3(gd_cdiv(M, 128)), dim3(512), lds, st, p);
        GD_LAUNCH_OK();
        return 0;
    }
    const long ldmax = (ldc > ldp ? ldc : ldp) > (ldd > ldr ? ldd : ldr) ? (ldc > ldp ? ldc : ldp) : (ldd > ldr ? ldd : ldr);
    // persistent kernel (bf16 operands): the epilogue combinations of the student step, each its own instantiation
    void (*pk)(GemmNtParams) = nullptr;
    if (ab_dtype == GD_BF16 && !accumulate && !(dact_src && residual)) {
        const bool cb = c_dtype == GD_BF16;
        if (!dact_src && !residual) {
            if (act == 0 && !preact) pk = cb ? gemm_nt_persist_kernel<bf16, 0, 0, 0, false> : gemm_nt_persist_kernel<bf16, 0, 0, 0, true>;
#ifdef GD_GEMM_ANATOMY
            if (act == 0 && !preact && cb && gd_knobs().gemm_anat) {     // anatomy builds of the main loop (gemm_persist.h, ANAT)
                const int v = gd_knobs().gemm_anat;
                pk = v == 1 ? gemm_nt_persist_kernel<bf16, 0, 0, 0, false, 1> : v == 2 ? gemm_nt_persist_kernel<bf16, 0, 0, 0, false, 2>
                     : v == 3 ? gemm_nt_persist_kernel<bf16, 0, 0, 0, false, 3> : gemm_nt_persist_kernel<bf16, 0, 0, 0, false, 4>;
            }
            else
#endif
            if ((act == 1 || act == 3) && !preact) pk = cb ? gemm_nt_persist_kernel<bf16, 0, 1, 0, false> : gemm_nt_persist_kernel<bf16, 0, 1, 0, true>;
            else if (act == 1 && preact && cb) pk = gemm_nt_persist_kernel<bf16, 0, 1, 1, false>;
            else if (act == 3 && preact) pk = cb ? gemm_nt_persist_kernel<bf16, 0, 1, 2, false> : gemm_nt_persist_kernel<bf16, 0, 1, 2, true>;
        } else if (dact_src && dact == 1 && act == 0 && !preact && cb) pk = gemm_nt_persist_kernel<bf16, 1, 0, 0, false>;
        else if (dact_src && dact == 3 && act == 0 && !preact) pk = cb ? gemm_nt_persist_kernel<bf16, 3, 0, 0, false> : gemm_nt_persist_kernel<bf16, 3, 0, 0, true>;
        else if (residual && act == 0 && !preact) pk = cb ? gemm_nt_persist_kernel<bf16, 2, 0, 0, false> : gemm_nt_persist_kernel<bf16, 2, 0, 0, true>;
        if (csplit) {
            pk = nullptr;
            if (!dact_src && !residual && act == 3 && preact) pk = gemm_nt_persist_kernel<bf16, 0, 1, 2, true, 0, 1>;
            else if (!dact_src && !residual && (act == 1 || act == 3) && !preact) pk = gemm_nt_persist_kernel<bf16, 0, 1, 0, true, 0, 1>;
            else if (dact_src && dact == 3 && act == 0 && !preact) pk = gemm_nt_persist_kernel<bf16, 3, 0, 0, true, 0, 1>;
        }
    } else if (ab_dtype == GD_F16 && !accumulate && !(dact_src && residual)) {
        // fp16 operands (tf32h engine): f32 results with f32 epilogue tensors, or fp16 results with fp16 preact / dact_src (as the bf16 engine's)
        const bool ch = c_dtype == GD_F16;
#define GD_PK(S, A, P, C, L) gemm_nt_persist_kernel<f16, S, A, P, C>
        if (!dact_src && !residual) {
            if (act == 0 && !preact) pk = ch ? GD_PK(0, 0, 0, false, 0) : GD_PK(0, 0, 0, true, 0);
            else if ((act == 1 || act == 3) && !preact) pk = ch ? GD_PK(0, 1, 0, false, 0) : GD_PK(0, 1, 0, true, 0);
            else if (act == 3 && preact) pk = ch ? GD_PK(0, 1, 2, false, 0) : GD_PK(0, 1, 2, true, 0);
        } else if (dact_src && dact == 3 && act == 0 && !preact) pk = ch ? GD_PK(3, 0, 0, false, 0) : GD_PK(3, 0, 0, true, 0);
        else if (residual && act == 0 && !preact && !ch) pk = copy16 ? GD_PK(2, 0, 3, true, 0) : GD_PK(2, 0, 0, true, 0);
#undef GD_PK
    }
    // (the bf16 f32-output instantiations serve the tf32x engine: 3K-wide split operands, fp32 C / preact / dact_src / residual)
    const bool persist_ok = big && persist && pk && p.vec_epilogue && N % 8 == 0 && (!lora_t || lora_rt == 8) && 256 * ldmax * cs < 0x7fffffffL;
    GD_REQUIRE(!csplit || (persist_ok && ldc >= 3L * N && batch == 1),
               "gd_gemm_nt: split output (c_dtype 2) is served by the persistent kernel only: bf16 operands, M >= 1024, N >= 256, K %% 64 == 0, "
               "ldc >= 3N, and the GELU(+derivative) or dact 3 epilogues (M=%d N=%d K=%d act=%d dact=%d)", M, N, K, act, dact);
    GD_REQUIRE(!copy16 || persist_ok, "gd_gemm_nt_copy16: served by the persistent kernel only (M >= 1024, N >= 256, K %% 64 == 0): M=%d N=%d K=%d", M, N, K);
#ifdef GD_GEMM_EXPERIMENT32
    if (persist == 32 && big && ab_dtype == GD_BF16 && c_dtype == GD_BF16 && !bias && !lora_t && !preact && act == 0 && !dact_src && !residual &&
        !accumulate && (K * 2) % 128 == 0) {
        const int ntiles = gd_cdiv(M, 256) * gd_cdiv(N, 256);
        dim3 gridp(ntiles < ncu ? ntiles : ncu, batch);
        const int v = gd_knobs().gemm_anat;
        if (v == 1) hipLaunchKernelGGL(gemm_nt_p32_kernel<1>, gridp, dim3(512), 0, st, p);
        else if (v == 4) hipLaunchKernelGGL(gemm_nt_p32_kernel<4>, gridp, dim3(512), 0, st, p);
        else hipLaunchKernelGGL(gemm_nt_p32_kernel<0>, gridp, dim3(512), 0, st, p);
        GD_LAUNCH_OK();
        return 0;
    }
#endif
    if (persist_ok) {
        // (Tile quantisation — e.g. 1029 tiles of the N = 768 GEMMs on 256 CUs — costs far less than a round: the left-over
        // tiles run alone on an idle chip.  Handing them to the 128 x 128 kernel or cutting them into K slices was measured
        // slower / equal: DESIGN.md section 5.)
        const int ntiles = gd_cdiv(M, 256) * gd_cdiv(N, 256);
        dim3 gridp(ntiles < ncu ? ntiles : ncu, batch);
        hipLaunchKernelGGL(pk, gridp, dim3(512), 0, st, p);
        GD_LAUNCH_OK();
        return 0;
    }
    if (ab_dtype == GD_F16) {
        if (big) hipLaunchKernelGGL((gemm_nt_kernel<f16, 2, 4, 8>), gridb, dim3(512), 0, st, p);
        else if (dma) hipLaunchKernelGGL((gemm_nt_kernel<f16, 2, 2, 4>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(gemm_nt_regstage_kernel<f16>, grid, dim3(256), 0, st, p);
    } else if (ab_dtype == GD_BF16) {
        if (big) hipLaunchKernelGGL((gemm_nt_kernel<bf16, 2, 4, 8>), gridb, dim3(512), 0, st, p);
        else if (dma) hipLaunchKernelGGL((gemm_nt_kernel<bf16, 2, 2, 4>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(gemm_nt_regstage_kernel<bf16>, grid, dim3(256), 0, st, p);
    } else {
        if (big && gd_f32_big_tiles()) hipLaunchKernelGGL((gemm_nt_kernel<float, 2, 4, 8>), gridb, dim3(512), 0, st, p);
        else if (dma) hipLaunchKernelGGL((gemm_nt_kernel<float, 2, 2, 4>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(gemm_nt_regstage_kernel<float>, grid, dim3(256), 0, st, p);
    }
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_gemm_nt(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc,
                          int batch, long sA, long sW, long sC, int ab_dtype, int c_dtype, float alpha,
                          const float* bias, const float* lora_t, const float* lora_b, int lora_rt, void* preact,
                          long ldp, int act, const void* dact_src, long ldd, int dact, const void* residual, long ldr,
                          int accumulate, void* stream) {
    return gemm_nt_impl(A, W, C, M, N, K, lda, ldw, ldc, batch, sA, sW, sC, ab_dtype, c_dtype, alpha, nullptr, bias, lora_t, lora_b, lora_rt,
                        preact, ldp, act, dact_src, ldd, dact, residual, ldr, accumulate, stream);
}

extern "C" int gd_gemm_nt_scaled(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc,
                                 int batch, long sA, long sW, long sC, int ab_dtype, int c_dtype, float alpha, const float* alpha_dev,
                                 const float* bias, const float* lora_t, const float* lora_b, int lora_rt, void* preact,
                                 long ldp, int act, const void* dact_src, long ldd, int dact, const void* residual, long ldr,
                                 int accumulate, void* stream) {
    return gemm_nt_impl(A, W, C, M, N, K, lda, ldw, ldc, batch, sA, sW, sC, ab_dtype, c_dtype, alpha, alpha_dev, bias, lora_t, lora_b, lora_rt,
                        preact, ldp, act, dact_src, ldd, dact, residual, ldr, accumulate, stream);
}

static int gemm_tn_impl(const void* Y, const void* X, float* G, int M, int N, int K, long ldy, long ldx, long ldg,
                        int batch, long sY, long sX, long sG, int y_dtype, int x_dtype, float alpha, const float* alpha_dev, void* stream) {
    GD_REQUIRE(M > 0 && N > 0 && K > 0, "gd_gemm_tn: bad shape M=%d N=%d K=%d", M, N, K);
    GD_REQUIRE(N % 8 == 0 && K % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0,
               "gd_gemm_tn: N (%d), K (%d), ldy (%ld), ldx (%ld) must be multiples of 8", N, K, ldy, ldx);
    GD_REQUIRE(((uintptr_t)Y & 15) == 0 && ((uintptr_t)X & 15) == 0, "gd_gemm_tn: Y and X must be 16-byte aligned");
    GemmTnParams p;
    p.Y = Y; p.X = X; p.G = G; p.M = M; p.N = N; p.K = K; p.ldy = ldy; p.ldx = ldx; p.ldg = ldg;
    p.y_dtype = y_dtype; p.x_dtype = x_dtype; p.alpha = alpha; p.alpha_dev = alpha_dev; p.sY = sY; p.sX = sX; p.sG = sG; p.anat = gd_knobs().gemm_anat;
    GD_REQUIRE(batch >= 1 && sY % 8 == 0 && sX % 8 == 0, "gd_gemm_tn: bad batch / batch strides");
    if (N == 8 && ldy == 8 && y_dtype == GD_F32 && batch == 1 && K <= 2560 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)Y & 15) == 0) {   // LoRA weight gradients
        const int nth = gd_cdiv(gd_cdiv(K, 8), 64) * 64;      // threads per row group
        const bool narrow = nth <= 128;                       // 8 row groups of <= 128 threads; else 3 groups of <= 320; rows 8 deep
        const int rgs = narrow ? 8 : 3;
        // one block per CU (measured at 87 680 rows, tf32h step: 128 blocks 45 / 80 us (K 768 / 2304), 256: 34 / 55, 512: 43 / 66; the one-group
        // 512-block form of rounds 1-2: 58 / 75) — every block closes with N x K fp32 atomics on the same lines, so fewer, fatter blocks win
        const int nblk = gd_knobs().ncu > 0 ? gd_knobs().ncu : 256;
        int mchunk = ((gd_cdiv(M, nblk) + rgs * 8 - 1) / (rgs * 8)) * (rgs * 8);      // whole row groups of whole 8-row steps
        p.mchunk = mchunk;
        dim3 grid(1, gd_cdiv(M, mchunk), 1), blk(nth * rgs);
        hipStream_t st = (hipStream_t)stream;
        if (x_dtype == GD_BF16) { if (narrow) hipLaunchKernelGGL((gemm_tn_skinny_kernel<bf16, 8, 8, 128>), grid, blk, 0, st, p); else hipLaunchKernelGGL((gemm_tn_skinny_kernel<bf16, 3, 8, 320>), grid, blk, 0, st, p); }
        else if (x_dtype == GD_F16) { if (narrow) hipLaunchKernelGGL((gemm_tn_skinny_kernel<f16, 8, 8, 128>), grid, blk, 0, st, p); else hipLaunchKernelGGL((gemm_tn_skinny_kernel<f16, 3, 8, 320>), grid, blk, 0, st, p); }
        else { if (narrow) hipLaunchKernelGGL((gemm_tn_skinny_kernel<float, 8, 8, 128>), grid, blk, 0, st, p); else hipLaunchKernelGGL((gemm_tn_skinny_kernel<float, 3, 8, 320>), grid, blk, 0, st, p); }
        GD_LAUNCH_OK();
        return 0;
    }
    const bool h16 = y_dtype == GD_F16 && x_dtype == GD_F16 && N >= 64 && K >= 64;    // tf32h: fp16 MFMA, same kernel
    // tf32h, one operand still fp32 in memory (it is rounded to fp16 on the way into LDS; a gradient Y under the scale 1 / *alpha_dev)
    const bool hy32 = y_dtype == GD_F32 && x_dtype == GD_F16 && N >= 64 && K >= 64, hx32 = y_dtype == GD_F16 && x_dtype == GD_F32 && N >= 64 && K >= 64;
    const bool bf = (y_dtype == GD_BF16 && x_dtype == GD_BF16 && N >= 64 && K >= 64) || h16 || hy32 || hx32;   // bf16 MFMA + transpose reads
    const int tl = bf ? 128 : 64;
    const int tiles = gd_cdiv(N, tl) * gd_cdiv(K, tl);
    // enough M-chunks to fill the chip without shredding the reduction (every chunk ends in N x K fp32 atomics: at 87 680 x 768 x 64
    // 768 blocks measured 40-43 us, 1024: 47-48, 2048: 62, 256: 53)
    const int tn_blocks = gd_knobs().tn_blocks;   // A/B knob: target block count
    int splits = ((tn_blocks > 0 ? tn_blocks : (bf ? 768 : 2048)) + tiles * batch - 1) / (tiles * batch);
    int mchunk = ((gd_cdiv(M, splits) + 63) / 64) * 64;
    if (mchunk < 256) mchunk = 256;
    p.mchunk = mchunk;
    dim3 grid(tiles, gd_cdiv(M, mchunk), batch);
    if (hy32) hipLaunchKernelGGL((gemm_tn_bf16_kernel<f16, true, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (hx32) hipLaunchKernelGGL((gemm_tn_bf16_kernel<f16, false, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (h16) hipLaunchKernelGGL(gemm_tn_bf16_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (bf) hipLaunchKernelGGL(gemm_tn_bf16_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_gemm_nt_copy16(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc, float alpha,
                                 const float* alpha_dev, const float* bias, const void* residual, long ldr, void* copy16, long ldc16,
                                 const float* copy_scale_dev, void* stream) {
    return gemm_nt_impl(A, W, C, M, N, K, lda, ldw, ldc, 1, 0, 0, 0, GD_F16, GD_F32, alpha, alpha_dev, bias, nullptr, nullptr, 0, nullptr, 0, 0, nullptr,
                        0, 0, residual, ldr, 0, stream, copy16, ldc16, copy_scale_dev);
}

extern "C" int gd_gemm_tn(const void* Y, const void* X, float* G, int M, int N, int K, long ldy, long ldx, long ldg,
                          int batch, long sY, long sX, long sG, int y_dtype, int x_dtype, float alpha, void* stream) {
    return gemm_tn_impl(Y, X, G, M, N, K, ldy, ldx, ldg, batch, sY, sX, sG, y_dtype, x_dtype, alpha, nullptr, stream);
}
extern "C" int gd_gemm_tn_scaled(const void* Y, const void* X, float* G, int M, int N, int K, long ldy, long ldx, long ldg,
                                 int batch, long sY, long sX, long sG, int y_dtype, int x_dtype, float alpha, const float* alpha_dev,
                                 void* stream) {
    return gemm_tn_impl(Y, X, G, M, N, K, ldy, ldx, ldg, batch, sY, sX, sG, y_dtype, x_dtype, alpha, alpha_dev, stream);
}
