// LoRA backward of one block in ONE pass over the (dq, dv) gradient block (utils/model.py:57-71, the q / v LoRA of _LoRA_qkv):
//     dt  [M, 8]  = dqv [M, K] . bt^T [K, 8]          (gradient of the rank projections t = LN(x) A^T)
//     gbt [8, K] += t^T [8, M]  . dqv [M, K]          (gradient of the B factors, as t^T [dq | dv])
// Both products read the same [M, K = 2 * H * 64] bf16 block: the two streaming kernels they ran on (gemm_nt_skinny, gemm_tn_skinny)
// each moved its 269 MB at the step's size.  Here a 4-wave block stages 64-row x 256-column slabs in LDS (register-prefetched one
// slab ahead), and every slab feeds both MFMA products: dt from natural row fragments (wave w owns rows 16w..16w+15, all 256
// columns), gbt from hardware-transposed reads (ds_read_b64_tr_b16: wave w owns columns 64w..64w+63, all 64 rows) with t^T as the
// A operand, split into a bf16 high and low part so that the fp32 projections lose nothing (2^-17 relative).  The [8, K] partial of
// a block lives in the accumulators over all of its row chunks and ends in one round of fp32 atomics.
#include "gd_common.h"

struct LoraBwdParams {
    const void* X; long ldx;        // dqv rows: M x K of the 16-bit operand type (bf16 | fp16), row stride ldx elements
    const float* t;                 // [M, 8] f32
    const void* bt;                 // [8, K] of the operand type
    float* dt;                      // [M, 8] f32 (written)
    float* gbt;                     // [8, K] f32 (accumulated)
    int M, K;
    // tf32h engine (fp16 operands): t is multiplied by *t_mul before it is split into its high and low 16-bit parts (a gradient in the t role goes
    // in under the step's power-of-two scale s), dt and the gbt partial by *out_mul on the way out (1 / s: X or t carried s).  null = 1.
    const float* t_mul; const float* out_mul;
    int dt_scaled;                  // 1: dt leaves WITHOUT *out_mul (still in X's scaled domain: the consumers take it under s anyway)
};

#define LB_ROWS 64
#define LB_SLAB 256
#define LB_ROWB (LB_SLAB * 2 + 16)     // slab row in LDS: 512 B + 16 B pad (natural b128 reads and transpose reads both conflict-light)

__device__ __forceinline__ bf16x8 lb_tr_frag(const char* tile, int col0, int u, int lane) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const char* a0 = tile + (32 * u + 4 * g + q) * LB_ROWB + (col0 + 4 * pp) * 2;
    const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 16 * LB_ROWB));
    const s16x8 z = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    return __builtin_bit_cast(bf16x8, z);
}

template <typename T, int NSLAB>      // T: bf16 | f16
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void lora_bwd_fused_kernel(LoraBwdParams p) {
    typedef typename Mma<T>::Frag Frag;
    // one LDS block: the slab [64][LB_ROWB], the B factors [8][K], the chunk's t rows [64][8]; the closing exchange reuses it as the [8][K] fp32 partial
    constexpr int SXB = LB_ROWS * LB_ROWB, SBB = 8 * NSLAB * LB_SLAB * 2, STB = LB_ROWS * 8 * 4;
    static_assert(SXB + SBB + STB >= 8 * NSLAB * LB_SLAB * 4, "the fp32 partial fits");
    __shared__ __attribute__((aligned(16))) char smem_lb[SXB + SBB + STB];
    char* const sX = smem_lb;
    T* const sBt = (T*)(smem_lb + SXB);
    float* const sT = (float*)(smem_lb + SXB + SBB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int K = p.K;
    const T* X = (const T*)p.X;
    const float tmul = p.t_mul ? *p.t_mul : 1.0f, omul = p.out_mul ? *p.out_mul : 1.0f;
    // the B factors stay in LDS for the whole launch
    const bool want_dt = p.bt != nullptr;          // bt == NULL: only gbt += t^T . X (e.g. the LoRA-A gradient dt^T . LN(x))
    if (want_dt)
        for (int i = tid; i < 8 * K / 8; i += 256) *(uint4*)(sBt + i * 8) = *(const uint4*)((const T*)p.bt + i * 8);

    f32x4 gacc[NSLAB][4];
#pragma unroll
    for (int s = 0; s < NSLAB; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) gacc[s][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunk = (p.M + LB_ROWS - 1) / LB_ROWS;
    // slab staging through registers: 64 rows x 32 chunks of 16 bytes = 2048 chunks / 256 threads
    uint4 rx[8];
    auto gload = [&](int chunk, int s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = tid + 256 * i, rr = ch >> 5, cc = ch & 31;
            const int m = chunk * LB_ROWS + rr;
            rx[i] = m < p.M ? *(const uint4*)(X + (long)m * p.ldx + s * LB_SLAB + cc * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    int chunk = blockIdx.x;
    if (chunk < nchunk) gload(chunk, 0);
    for (; chunk < nchunk; chunk += gridDim.x) {
        f32x4 dacc = {0.f, 0.f, 0.f, 0.f};
        Frag thi[2], tlo[2];
#pragma unroll
        for (int s = 0; s < NSLAB; ++s) {          // (unrolled: gacc[s] must be a compile-time register index)
            __syncthreads();                       // every wave is done with the previous slab (and with sT of the previous chunk)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ch = tid + 256 * i, rr = ch >> 5, cc = ch & 31;
                *(uint4*)(sX + rr * LB_ROWB + cc * 16) = rx[i];
            }
            if (s == 0) {
                for (int i = tid; i < LB_ROWS * 8; i += 256) {
                    const int m = chunk * LB_ROWS + (i >> 3);
                    sT[i] = m < p.M ? p.t[(long)m * 8 + (i & 7)] : 0.f;
                }
            }
            __syncthreads();
            // prefetch the next slab (of this chunk or the first of the block's next chunk)
            if (s + 1 < NSLAB) gload(chunk, s + 1);
            else if (chunk + (int)gridDim.x < nchunk) gload(chunk + gridDim.x, 0);
            if (s == 0) {
                // t^T as the A operand of the gbt product, in the k-slot order of the transpose reads:
                // lane (g, i): rank i, slots e < 4 -> row 32u + 4g + e, e >= 4 -> row 32u + 16 + 4g + (e - 4)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int row = 32 * u + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
                        const float v = c < 8 ? sT[row * 8 + c] * tmul : 0.f;
                        const T h = from_f32<T>(v);
                        thi[u][e] = h;
                        tlo[u][e] = from_f32<T>(v - to_f32<T>(h));
                    }
            }
            // ---- dt: rows 16w..16w+15 of the chunk, this slab's 256 columns
            if (want_dt)
#pragma unroll
            for (int kc = 0; kc < LB_SLAB / 32; ++kc) {
                const Frag a = *(const Frag*)(sX + (16 * wave + c) * LB_ROWB + (32 * kc + 8 * g) * 2);
                Frag b = {};
                if (c < 8) b = *(const Frag*)(sBt + (long)c * K + s * LB_SLAB + 32 * kc + 8 * g);
                dacc = Mma<T>::mma(a, b, dacc);
            }
            // ---- gbt: columns 64w..64w+63 of this slab, all 64 rows
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const Frag b = __builtin_bit_cast(Frag, lb_tr_frag(sX, 64 * wave + 16 * nt, u, lane));
                    gacc[s][nt] = Mma<T>::mma(thi[u], b, gacc[s][nt]);
                    gacc[s][nt] = Mma<T>::mma(tlo[u], b, gacc[s][nt]);
                }
        }
        // dt of the chunk: D[row = 4g + r][col = c = rank]
        if (c < 8 && want_dt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = chunk * LB_ROWS + 16 * wave + 4 * g + r;
                if (m < p.M) p.dt[(long)m * 8 + c] = dacc[r] * (p.dt_scaled ? 1.0f : omul);
            }
        }
    }
    // gbt partial of the block: D[row = 4g + r = rank][col = c] — through LDS, so that the closing atomics leave as FULL lines (64 consecutive
    // columns of one rank per wave instruction instead of two 64-byte pieces: half the line operations the memory side serialises; gemm.hip, TN kernel)
    __syncthreads();
    float* red = (float*)smem_lb;
    if (g < 2) {
#pragma unroll
        for (int s = 0; s < NSLAB; ++s)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(4 * g + r) * K + s * LB_SLAB + 64 * wave + 16 * nt + c] = gacc[s][nt][r];
    }
    __syncthreads();
    for (int i = tid; i < 8 * K; i += 256) atomicAdd(p.gbt + i, red[i] * omul);
}

template <typename T>
static int lora_bwd_launch(const LoraBwdParams& p, hipStream_t s) {
    const int nchunk = (p.M + LB_ROWS - 1) / LB_ROWS;
    const int grid = nchunk < 512 ? nchunk : 512;
    switch (p.K / LB_SLAB) {
        case 1: hipLaunchKernelGGL((lora_bwd_fused_kernel<T, 1>), dim3(grid), dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL((lora_bwd_fused_kernel<T, 2>), dim3(grid), dim3(256), 0, s, p); break;
        case 3: hipLaunchKernelGGL((lora_bwd_fused_kernel<T, 3>), dim3(grid), dim3(256), 0, s, p); break;
        case 4: hipLaunchKernelGGL((lora_bwd_fused_kernel<T, 4>), dim3(grid), dim3(256), 0, s, p); break;
        case 6: hipLaunchKernelGGL((lora_bwd_fused_kernel<T, 6>), dim3(grid), dim3(256), 0, s, p); break;
        case 8: hipLaunchKernelGGL((lora_bwd_fused_kernel<T, 8>), dim3(grid), dim3(256), 0, s, p); break;
        default: gd_set_error("gd_lora_bwd_fused: K / 256 = %d not instantiated (1, 2, 3, 4, 6, 8)", p.K / LB_SLAB); return -1;
    }
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_lora_bwd_fused(const void* dqv, long ldx, const float* t, const void* bt, float* dt, float* gbt, int M, int K,
                                 void* stream) {
    GD_REQUIRE(M > 0 && K > 0 && K % LB_SLAB == 0 && K / LB_SLAB <= 8 && ldx % 8 == 0, "gd_lora_bwd_fused: K must be a multiple of 256 (<= 2048), ldx of 8");
    GD_REQUIRE(((uintptr_t)dqv & 15) == 0 && ((uintptr_t)bt & 15) == 0 && t && gbt && (dt || !bt), "gd_lora_bwd_fused: alignment / null pointers");
    LoraBwdParams p = {dqv, ldx, t, bt, dt, gbt, M, K, nullptr, nullptr, 0};
    return lora_bwd_launch<bf16>(p, (hipStream_t)stream);
}

// the same pass on either 16-bit operand type, with the tf32h engine's device-side scales (LoraBwdParams): dtype GD_BF16 | GD_F16
extern "C" int gd_lora_bwd_fused_scaled(const void* dqv, long ldx, const float* t, const void* bt, float* dt, float* gbt, int M, int K, int dtype,
                                        const float* t_mul_dev, const float* out_mul_dev, int dt_scaled, void* stream) {
    GD_REQUIRE(M > 0 && K > 0 && K % LB_SLAB == 0 && K / LB_SLAB <= 8 && ldx % 8 == 0, "gd_lora_bwd_fused_scaled: K must be a multiple of 256 (<= 2048), ldx of 8");
    GD_REQUIRE(((uintptr_t)dqv & 15) == 0 && ((uintptr_t)bt & 15) == 0 && t && gbt && (dt || !bt), "gd_lora_bwd_fused_scaled: alignment / null pointers");
    GD_REQUIRE(dtype == GD_BF16 || dtype == GD_F16, "gd_lora_bwd_fused_scaled: operands are bf16 or fp16 (dtype %d)", dtype);
    LoraBwdParams p = {dqv, ldx, t, bt, dt, gbt, M, K, t_mul_dev, out_mul_dev, dt_scaled};
    return dtype == GD_F16 ? lora_bwd_launch<f16>(p, (hipStream_t)stream) : lora_bwd_launch<bf16>(p, (hipStream_t)stream);
}
