// Bottleneck adapter (a3 — utils/model.py:7-25: out + up(relu(down(out))), D -> 64 -> D, no bias), fused for gfx950.
//
// As two GEMMs the adapter is pure HBM traffic: read x (down), write h, read h + x again (up + residual), write out —
// four passes over [M, D] for 2 * 2 * M * D * 64 FLOP.  Here a block keeps its 32 x D slice of x in LDS (one LDS-DMA
// fill, 16-byte chunks XOR-swizzled by row so the MFMA fragment reads are conflict-free), contracts it against the down
// weight (each wave owns 16 of the 64 bottleneck columns; its 16 x D weight slab is prefetched into registers while the
// tile is in flight), gates, parks the 32 x 64 hidden tile in LDS, and runs the up projection + residual straight out
// of LDS: x is read from HBM once and out written once.  Three blocks fit a CU, so one block's fill overlaps its
// neighbours' MFMA / store phases.
//
// The same kernel is the backward-to-input:  dX = dOut + ((dOut . up) * [h > 0]) . down  — first weight = up^T [64, D],
// gate taken from the saved forward hidden tile instead of the value's own sign, second weight = down^T [D, 64].
// The gated hidden tile (forward: h, backward: dh) is written out because the weight gradients contract it (gd_gemm_tn).
#include "gd_common.h"

#define AD_BM 32
#define AD_BOT 64

// phase 2 of one wave: out[32, D/4] = hidden . w2^T + x.  Weight row n0 + 4*fr + j feeds tile j, so a lane ends up with 4
// consecutive columns and 16 lanes cover 128 contiguous bytes of an output row.  The next column group's weight
// fragments are requested before the current group's MFMAs (L2 latency under the stores of the previous group).
template <int D, bool FULL>
__device__ __forceinline__ void ad_phase2(const char* sX, const char* sH, const bf16* __restrict__ w2, bf16* __restrict__ out,
                                          int wave, int g, int c, int row0, int M) {
    constexpr int CPR = D / 8, NG = D / 256;
    auto load_b2 = [&](int gq, bf16x8 (&b2)[4][2]) {
        const char* wr = (const char*)w2 + (long)(wave * (D / 4) + 64 * gq + 4 * c) * (AD_BOT * 2) + 16 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b2[j][ks] = *(const bf16x8*)(wr + j * (AD_BOT * 2) + 64 * ks);
    };
    bf16x8 b2[2][4][2];
    load_b2(0, b2[0]);
    bf16x8 ah[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) ah[i][ks] = *(const bf16x8*)(sH + (16 * i + c) * (AD_BOT * 2) + 64 * ks + 16 * g);
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
        if (gq + 1 < NG) load_b2(gq + 1, b2[(gq + 1) & 1]);
        const int off = 2 * (wave * (D / 4) + 64 * gq + 4 * c);   // byte offset of this lane's 4 columns inside a row
        bf16x4 xr[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * i + 4 * g + r;
                xr[i][r] = *(const bf16x4*)(sX + ((row * CPR + ((off >> 4) ^ (row & 15))) * 16) + (off & 8));
            }
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Mma<bf16>::mma(ah[i][ks], b2[gq & 1][j][ks], acc[i][j]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * i + 4 * g + r;
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16)(acc[i][j][r] + (float)xr[i][r][j]);
                if (FULL || row0 + row < M) *(bf16x4*)((char*)out + (long)(row0 + row) * (D * 2) + off) = o;
            }
    }
}

template <int D>
__global__ __launch_bounds__(256, D <= 768 ? 3 : 2) void adapter_fused_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w1,
                                                            const bf16* __restrict__ w2, const bf16* __restrict__ gate,
                                                            bf16* __restrict__ hout, bf16* __restrict__ out, int M) {
    constexpr int CPR = D / 8;                  // 16-byte chunks per row
    constexpr int KS = D / 32;                  // MFMA K-steps of phase 1
    constexpr int PIECES = AD_BM * D * 2 / 1024, PPW = PIECES / 4;
    static_assert(D % 256 == 0, "D must split into 64-column groups over 4 waves");
    __shared__ __attribute__((aligned(16))) char sX[AD_BM * D * 2];
    __shared__ __attribute__((aligned(16))) char sH[AD_BM * AD_BOT * 2];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- this wave's 16 rows of the first weight, all K-steps (L2-resident; requested before the tile fill) ----
    bf16x8 b1[KS];
    {
        const char* w1r = (const char*)w1 + (long)(16 * wave + c) * (D * 2) + 16 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b1[ks] = *(const bf16x8*)(w1r + ks * 64);
    }
    const int row0 = blockIdx.x * AD_BM;
    // ---- the x tile: LDS position (row, p) holds source chunk p ^ (row & 15) ----
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = wave * PPW + i;
        const int lin = piece * 64 + lane, row = lin / CPR, p = lin % CPR;
        const char* src = (const char*)x + (long)min(row0 + row, M - 1) * (D * 2) + ((p ^ (row & 15)) * 16);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sX + piece * 1024), 16, 0, 0);
    }
    // gate values in the accumulator layout: unconditional loads from a clamped address (a conditional load costs a full
    // memory round trip each inside its exec-masked block)
    const bool gated = gate != nullptr;
    bf16 gt[2][4];
    {
        const bf16* gp = gated ? gate : w1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = min(row0 + 16 * i + 4 * g + r, M - 1);
                gt[i][r] = gp[gated ? (long)row * AD_BOT + 16 * wave + c : 0];
            }
    }
    __syncthreads();

    // ---- phase 1: hidden[32, 16 of 64] = x_tile . w1^T  (two independent accumulation chains per row tile) ----
    f32x4 acc1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) acc1[i][0] = acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment reads in batches of 8 (4 K-steps x 2 row tiles) ahead of their MFMAs: the block is latency-bound, not
    // LDS-bandwidth-bound (every wave sweeps the whole tile once: 4 x 48 KB of reads per block)
    constexpr int KB = 4;
    static_assert(KS % KB == 0, "K-steps must split into read batches");
#pragma unroll
    for (int k0 = 0; k0 < KS; k0 += KB) {
        bf16x8 a[KB][2];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[kk][i] = *(const bf16x8*)(sX + (((16 * i + c) * CPR + ((4 * (k0 + kk) + g) ^ c)) * 16));
#pragma unroll
        for (int kk = 0; kk < KB; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc1[i][kk & 1] = Mma<bf16>::mma(a[kk][i], b1[k0 + kk], acc1[i][kk & 1]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc1[i][0][r] + acc1[i][1][r];
            v = gated ? ((float)gt[i][r] > 0.f ? v : 0.f) : fmaxf(v, 0.f);
            ((bf16*)sH)[(16 * i + 4 * g + r) * AD_BOT + 16 * wave + c] = (bf16)v;
        }
    __syncthreads();
    if (hout != nullptr && row0 + (tid >> 3) < M)
        *(uint4*)((char*)hout + (long)row0 * (AD_BOT * 2) + tid * 16) = *(const uint4*)(sH + tid * 16);

    if (row0 + AD_BM <= M) ad_phase2<D, true>(sX, sH, w2, out, wave, g, c, row0, M);
    else ad_phase2<D, false>(sX, sH, w2, out, wave, g, c, row0, M);
}

// Persistent form (one 4-wave block per CU, M >= a few thousand rows): in the kernel above every block re-reads both
// weights (192 KB at D = 768) through the vector-memory path for a 48 KB tile — four times the tile's own bytes, and
// that, not HBM, is what it is bound by.  Here a wave keeps its slab of BOTH weights in registers for the whole launch
// (one wave per SIMD: 512 registers), walks tiles blockIdx.x, blockIdx.x + gridDim.x, ..., and has the next tile's 48 KB
// in flight into registers (12 x 16 B per thread) while it works on the current one out of LDS.
template <int D>
__global__ __launch_bounds__(256, 1) void adapter_persist_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w1,
                                                                 const bf16* __restrict__ w2, const bf16* __restrict__ gate,
                                                                 bf16* __restrict__ hout, bf16* __restrict__ out, int M) {
    constexpr int CPR = D / 8, KS = D / 32, NG = D / 256;
    constexpr int PPT = AD_BM * CPR / 256;      // 16-byte chunks of a tile per thread
    __shared__ __attribute__((aligned(16))) char sX[AD_BM * D * 2];
    __shared__ __attribute__((aligned(16))) char sH[AD_BM * AD_BOT * 2];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = (M + AD_BM - 1) / AD_BM;
    const bool gated = gate != nullptr;

    // (the launch keeps gridDim.x <= ntiles, so every block owns at least one tile)
    int tile = blockIdx.x;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pre[PPT];
    static_for<PPT>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = ic;
        const int lin = i * 256 + tid, row = lin / CPR, p = lin % CPR;
        pre[i] = *(const u32x4*)((const char*)x + (long)min(tile * AD_BM + row, M - 1) * (D * 2) + p * 16);
    });
    constexpr bool B2REG = D <= 768;            // D = 1024: both weights + the prefetched tile do not fit 512 registers
    bf16x8 b1[KS], b2[B2REG ? NG : 1][4][2];
    {
        const char* w1r = (const char*)w1 + (long)(16 * wave + c) * (D * 2) + 16 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b1[ks] = *(const bf16x8*)(w1r + ks * 64);
#pragma unroll
        for (int gq = 0; gq < (B2REG ? NG : 0); ++gq)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    b2[gq][j][ks] = *(const bf16x8*)((const char*)w2 + (long)(wave * (D / 4) + 64 * gq + 4 * c + j) * (AD_BOT * 2) +
                                                     64 * ks + 16 * g);
    }
    for (; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * AD_BM;
        __syncthreads();                        // the previous tile's readers are done with sX and sH
        static_for<PPT>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = ic;
            const int lin = i * 256 + tid, row = lin / CPR, p = lin % CPR;
            *(u32x4*)(sX + (row * CPR + (p ^ (row & 15))) * 16) = pre[i];
        });
        unsigned short gt[8];
        {
            const bf16* gp = gated ? gate : w1;
            static_for<8>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = ic / 4, r = ic % 4;
                const int row = min(row0 + 16 * i + 4 * g + r, M - 1);
                gt[ic] = ((const unsigned short*)gp)[gated ? (long)row * AD_BOT + 16 * wave + c : 0];
            });
        }
        __syncthreads();
        {   // next tile into registers (the last iteration re-reads its own tile: harmless, keeps the loads unconditional)
            const int nt = tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile;
            static_for<PPT>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = ic;
                const int lin = i * 256 + tid, row = lin / CPR, p = lin % CPR;
                pre[i] = *(const u32x4*)((const char*)x + (long)min(nt * AD_BM + row, M - 1) * (D * 2) + p * 16);
            });
        }

        // ---- phase 1 ----
        f32x4 acc1[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) acc1[i][0] = acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int KB = 4;
#pragma unroll
        for (int k0 = 0; k0 < KS; k0 += KB) {
            bf16x8 a[KB][2];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[kk][i] = *(const bf16x8*)(sX + (((16 * i + c) * CPR + ((4 * (k0 + kk) + g) ^ c)) * 16));
#pragma unroll
            for (int kk = 0; kk < KB; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc1[i][kk & 1] = Mma<bf16>::mma(a[kk][i], b1[k0 + kk], acc1[i][kk & 1]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc1[i][0][r] + acc1[i][1][r];
                v = gated ? ((short)gt[4 * i + r] > 0 ? v : 0.f) : fmaxf(v, 0.f);   // bf16 > 0 <=> its bits, as int16, > 0
                ((bf16*)sH)[(16 * i + 4 * g + r) * AD_BOT + 16 * wave + c] = (bf16)v;
            }
        __syncthreads();
        if (hout != nullptr && row0 + (tid >> 3) < M)
            *(uint4*)((char*)hout + (long)row0 * (AD_BOT * 2) + tid * 16) = *(const uint4*)(sH + tid * 16);

        // ---- phase 2 ----
        if constexpr (!B2REG) {   // second weight streamed from L2 per tile, as in the one-tile kernel
            if (row0 + AD_BM <= M) ad_phase2<D, true>(sX, sH, w2, out, wave, g, c, row0, M);
            else ad_phase2<D, false>(sX, sH, w2, out, wave, g, c, row0, M);
            continue;
        }
        bf16x8 ah[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) ah[i][ks] = *(const bf16x8*)(sH + (16 * i + c) * (AD_BOT * 2) + 64 * ks + 16 * g);
        const bool full = row0 + AD_BM <= M;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            const int off = 2 * (wave * (D / 4) + 64 * gq + 4 * c);
            uint2 xr[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * i + 4 * g + r;
                    xr[i][r] = *(const uint2*)(sX + ((row * CPR + ((off >> 4) ^ (row & 15))) * 16) + (off & 8));
                }
            f32x4 acc[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = Mma<bf16>::mma(ah[i][ks], b2[B2REG ? gq : 0][j][ks], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * i + 4 * g + r;
                    const uint2 xv = xr[i][r];   // 4 bf16: widen by shifting into the high half
                    const float x0 = __builtin_bit_cast(float, xv.x << 16), x1 = __builtin_bit_cast(float, xv.x & 0xffff0000u);
                    const float x2 = __builtin_bit_cast(float, xv.y << 16), x3 = __builtin_bit_cast(float, xv.y & 0xffff0000u);
                    bf16x4 o;
                    o[0] = (bf16)(acc[i][0][r] + x0); o[1] = (bf16)(acc[i][1][r] + x1);
                    o[2] = (bf16)(acc[i][2][r] + x2); o[3] = (bf16)(acc[i][3][r] + x3);
                    if (full || row0 + row < M) *(bf16x4*)((char*)out + (long)(row0 + row) * (D * 2) + off) = o;
                }
        }
    }
}

// ---- tf32h engine: the same persistent kernel on fp32 tensors with fp16 operands -----------------------------------------------
// x32 [M, D] fp32 is read ONCE per tile into registers (24 x 16 B per thread), rounded to fp16 (times *in_scale: the backward's gradient scale)
// on its way into the LDS image the bf16 kernel uses, both products run on the fp16 MFMA, the hidden tile leaves as fp16 (forward: relu(x.down^T),
// the backward's gate and weight-gradient operand; backward: the scaled d(hidden)), and the result is out32 = x32 + alpha * (hidden . w2^T) in
// fp32 — plus, when asked for, out16 = fp16(out32 * *copy_scale), the operand of the next product.
// Round 4: the fp32 tile is ALSO parked in LDS (96 KB at D = 768: with the fp16 image and the hidden tile 148 of the 160 KB) and the residual add
// reads it from there, and BOTH weights stay in registers for the whole launch (the 32 residual registers are gone) — phase 2 issues no vector-memory
// load at all.  Before, every tile re-read its 96 KB of x from L2 and streamed the 96 KB second weight from L2, and those loads sat behind the next
// tile's 48 HBM prefetch loads of the same wave in the in-order vector-memory queue: the phase waited for the prefetch it was meant to hide.  Replaces, per call, a cast pass, the N = 64 GEMM, a second cast and the
// K = 64 GEMM (whose 129 us are pure residual / result traffic) of the unfused tf32h path.
// LN (round 5): the kernel is the only producer of a residual-stream tensor that holds WHOLE rows on chip, so it can also write what the NEXT block's
// LayerNorm 1 would compute from its result — ln16 = fp16(LayerNorm(out32; ln_w, ln_b, ln_eps)) with the row statistics the backward needs — and that
// block's 75 us LayerNorm pass (269 MB read again + 135 MB written) shrinks to the 135 MB write here.  Phase 2 writes its fp32 result back over the
// residual tile in LDS (each thread over exactly the words it read), a third phase takes the rows from there: one wave per 8 rows, a row in registers
// (12 floats per lane), mean and variance by two wave reductions as ln_fwd_kernel does — the same statistics to the last bit of rounding order.
template <int D, bool LN = false>
__global__ __launch_bounds__(256, 1) void adapter_persist_h_kernel(const float* __restrict__ x32, const f16* __restrict__ w1, const f16* __restrict__ w2,
                                                                   const f16* __restrict__ gate, f16* __restrict__ hout, float* __restrict__ out32,
                                                                   f16* __restrict__ out16, const float* in_scale, const float* alpha_dev,
                                                                   const float* copy_scale, int M, const float* __restrict__ ln_w = nullptr,
                                                                   const float* __restrict__ ln_b = nullptr, float ln_eps = 0.f, f16* __restrict__ ln16 = nullptr,
                                                                   float* __restrict__ ln_mean = nullptr, float* __restrict__ ln_rstd = nullptr) {
    constexpr int CPR = D / 8, KS = D / 32, NG = D / 256;
    constexpr int PPT = AD_BM * CPR / 256;      // fp16 16-byte chunks (8 elements = two fp32 16-byte loads) of a tile per thread
    static_assert(D <= 768, "both weights and the fp32 prefetch have to fit 512 registers");
    __shared__ __attribute__((aligned(16))) char sX[AD_BM * D * 2];
    __shared__ __attribute__((aligned(16))) char sR[AD_BM * D * 4];      // the same tile in fp32: the residual of phase 2
    __shared__ __attribute__((aligned(16))) char sH[AD_BM * AD_BOT * 2];
    __shared__ __attribute__((aligned(16))) float sLN[LN ? 2 * D : 4];   // LN: the next block's LayerNorm affine (gamma | beta)
    if (LN) {
        for (int i = threadIdx.x; i < D; i += 256) { sLN[i] = ln_w[i]; sLN[D + i] = ln_b[i]; }
    }
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = (M + AD_BM - 1) / AD_BM;
    const bool gated = !LN && gate != nullptr;      // (LN: the forward form only — ReLU gate, no scales, no fp16 copy of the un-normed result)
    const float sin = (!LN && in_scale) ? *in_scale : 1.0f, alpha = (!LN && alpha_dev) ? *alpha_dev : 1.0f, scp = (!LN && copy_scale) ? *copy_scale : 1.0f;
    if (LN) out16 = nullptr;

    int tile = blockIdx.x;
    f32x4 pre[PPT][2];
    static_for<PPT>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = ic;
        const int lin = i * 256 + tid, row = lin / CPR, p = lin % CPR;
        const float* src = x32 + (long)min(tile * AD_BM + row, M - 1) * D + p * 8;
        pre[i][0] = *(const f32x4*)src;
        pre[i][1] = *(const f32x4*)(src + 4);
    });
    // both weights stay in registers for the whole launch (96 + 96 registers at D = 768, beside the 96 of the fp32 prefetch)
    f16x8 b1[KS];
    {
        const char* w1r = (const char*)w1 + (long)(16 * wave + c) * (D * 2) + 16 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b1[ks] = *(const f16x8*)(w1r + ks * 64);
    }
    f16x8 b2[NG][4][2];
#pragma unroll
    for (int gq = 0; gq < NG; ++gq)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                b2[gq][j][ks] = *(const f16x8*)((const char*)w2 + (long)(wave * (D / 4) + 64 * gq + 4 * c + j) * (AD_BOT * 2) + 64 * ks + 16 * g);
    for (; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * AD_BM;
        __syncthreads();                        // the previous tile's readers are done with sX and sH
        static_for<PPT>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = ic;
            const int lin = i * 256 + tid, row = lin / CPR, p = lin % CPR;
            f16x8 h;
            { const float t8[8] = {pre[i][0][0] * sin, pre[i][0][1] * sin, pre[i][0][2] * sin, pre[i][0][3] * sin, pre[i][1][0] * sin, pre[i][1][1] * sin, pre[i][1][2] * sin, pre[i][1][3] * sin}; h = f16_sat8(t8); }
            *(f16x8*)(sX + (row * CPR + (p ^ (row & 15))) * 16) = h;
            *(f32x4*)(sR + (row * D + p * 8) * 4) = pre[i][0];
            *(f32x4*)(sR + (row * D + p * 8 + 4) * 4) = pre[i][1];
        });
        unsigned short gt[8];
        {
            const f16* gp = gated ? gate : w1;
            static_for<8>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = ic / 4, r = ic % 4;
                const int row = min(row0 + 16 * i + 4 * g + r, M - 1);
                gt[ic] = ((const unsigned short*)gp)[gated ? (long)row * AD_BOT + 16 * wave + c : 0];
            });
        }
        __syncthreads();
        {   // next tile into registers (the last iteration re-reads its own tile: harmless, keeps the loads unconditional)
            const int nt = tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile;
            static_for<PPT>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = ic;
                const int lin = i * 256 + tid, row = lin / CPR, p = lin % CPR;
                const float* src = x32 + (long)min(nt * AD_BM + row, M - 1) * D + p * 8;
                pre[i][0] = *(const f32x4*)src;
                pre[i][1] = *(const f32x4*)(src + 4);
            });
        }
        // ---- phase 1 ----
        f32x4 acc1[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) acc1[i][0] = acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int KB = 4;
#pragma unroll
        for (int k0 = 0; k0 < KS; k0 += KB) {
            f16x8 a[KB][2];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[kk][i] = *(const f16x8*)(sX + (((16 * i + c) * CPR + ((4 * (k0 + kk) + g) ^ c)) * 16));
#pragma unroll
            for (int kk = 0; kk < KB; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc1[i][kk & 1] = Mma<f16>::mma(a[kk][i], b1[k0 + kk], acc1[i][kk & 1]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc1[i][0][r] + acc1[i][1][r];
                // fp16 > 0 <=> its bits, as int16, > 0 (the gate is the forward's relu output: non-negative, -0 never occurs)
                v = gated ? ((short)gt[4 * i + r] > 0 ? v : 0.f) : fmaxf(v, 0.f);
                ((f16*)sH)[(16 * i + 4 * g + r) * AD_BOT + 16 * wave + c] = from_f32<f16>(v);
            }
        __syncthreads();
        if (hout != nullptr && row0 + (tid >> 3) < M)
            *(uint4*)((char*)hout + (long)row0 * (AD_BOT * 2) + tid * 16) = *(const uint4*)(sH + tid * 16);
        // ---- phase 2 ----
        f16x8 ah[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) ah[i][ks] = *(const f16x8*)(sH + (16 * i + c) * (AD_BOT * 2) + 64 * ks + 16 * g);
        const bool full = row0 + AD_BM <= M;
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            // (LN: the lane's column offset behind an optimisation barrier — hoisted out of the tile loop, the 64-bit store addresses derived from it were
            //  spilled and RELOADED here, behind the next tile's 24 HBM prefetch loads in the in-order vector-memory queue: phase 2 waited for the prefetch)
            int col = wave * (D / 4) + 64 * gq + 4 * c;
            if (LN) asm volatile("" : "+v"(col));
            f32x4 acc[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = Mma<f16>::mma(ah[i][ks], b2[gq][j][ks], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * i + 4 * g + r;
                    if (!(full || row0 + row < M)) continue;
                    // (16 lanes of a row read 256 contiguous bytes; the lane groups of a ds_read_b128 — rows 4g and 4g + 4 — land on disjoint banks)
                    const f32x4 xr = *(const f32x4*)(sR + (row * D + col) * 4);
                    const f32x4 o = {fmaf(alpha, acc[i][0][r], xr[0]), fmaf(alpha, acc[i][1][r], xr[1]),
                                     fmaf(alpha, acc[i][2][r], xr[2]), fmaf(alpha, acc[i][3][r], xr[3])};
                    *(f32x4*)(out32 + (long)(row0 + row) * D + col) = o;
                    if (out16) {
                        const f16x4 h = f16_sat4(o[0] * scp, o[1] * scp, o[2] * scp, o[3] * scp);
                        *(f16x4*)(out16 + (long)(row0 + row) * D + col) = h;
                    }
                    if (LN) *(f32x4*)(sR + (row * D + col) * 4) = o;      // (the words this thread just read: the result replaces the residual)
                }
        }
        if (LN) {
            // ---- phase 3: LayerNorm of the tile's result rows for the next block (wave w: rows 8 w .. 8 w + 7) ----
            __syncthreads();
            constexpr int NV = D / 256;      // 16-byte chunks of a row per lane
#pragma unroll 4      // (four rows in flight: a row alone is a chain of LDS and cross-lane latencies)
            for (int rr = 0; rr < AD_BM / 4; ++rr) {
                const int row = (AD_BM / 4) * wave + rr;
                if (row0 + row >= M) continue;
                // (three passes over the row in LDS instead of the row in registers: the kernel has no 12 registers to spare — both weights and the next
                //  tile's 96-register prefetch are live here — and an LDS pass of 3 KB per row costs less than the spills did: 188 bytes per lane)
                int l4 = lane * 4;
                asm volatile("" : "+v"(l4));      // (as above: nothing lane-derived of this phase may be kept in registers across the tile loop)
                const char* rp = sR + (row * D + l4) * 4;
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const f32x4 t = *(const f32x4*)(rp + k * 1024);
                    s += (t[0] + t[1]) + (t[2] + t[3]);
                }
                const float mu = wave_sum(s) * (1.0f / (float)D);
                float q = 0.f;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const f32x4 t = *(const f32x4*)(rp + k * 1024);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = t[e] - mu; q = fmaf(d, d, q); }
                }
                const float rs = rsqrtf(wave_sum(q) * (1.0f / (float)D) + ln_eps);
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int c0 = l4 + 256 * k;
                    const f32x4 t = *(const f32x4*)(rp + k * 1024);
                    const f32x4 gm = *(const f32x4*)(sLN + c0), bt = *(const f32x4*)(sLN + D + c0);
                    *(f16x4*)(ln16 + (long)(row0 + row) * D + c0) = f16_sat4(fmaf((t[0] - mu) * rs, gm[0], bt[0]), fmaf((t[1] - mu) * rs, gm[1], bt[1]),
                                                                           fmaf((t[2] - mu) * rs, gm[2], bt[2]), fmaf((t[3] - mu) * rs, gm[3], bt[3]));
                }
                if (lane == 0) { ln_mean[row0 + row] = mu; ln_rstd[row0 + row] = rs; }
            }
        }
    }
}

template <int D>
static void adapter_launch(const void* x, const void* w1, const void* w2, const void* gate, void* hout, void* out, int M,
                           hipStream_t s) {
    const int persist = gd_knobs().adapter_persist, cus = gd_knobs().ncu;
    if (persist && M >= 256 * AD_BM * 4) {
        hipLaunchKernelGGL(adapter_persist_kernel<D>, dim3(min(cus * persist, gd_cdiv(M, AD_BM))), dim3(256), 0, s, (const bf16*)x, (const bf16*)w1,
                           (const bf16*)w2, (const bf16*)gate, (bf16*)hout, (bf16*)out, M);
        return;
    }
    hipLaunchKernelGGL(adapter_fused_kernel<D>, dim3(gd_cdiv(M, AD_BM)), dim3(256), 0, s, (const bf16*)x, (const bf16*)w1,
                       (const bf16*)w2, (const bf16*)gate, (bf16*)hout, (bf16*)out, M);
}

// out[M, D] = x + gate(x . w1^T) . w2^T ;  hidden[M, 64] = gate(x . w1^T)  (bf16, may be null)
//   gate_src == null : ReLU (forward; w1 = down [64, D], w2 = up [D, 64])
//   gate_src [M, 64] : keep where gate_src > 0 (backward-to-input; x = dOut, w1 = up^T [64, D], w2 = down^T [D, 64])
extern "C" int gd_adapter_fused_supported(int D, int bottleneck, int dtype) {
    return dtype == GD_BF16 && bottleneck == AD_BOT && (D == 256 || D == 512 || D == 768 || D == 1024);
}

extern "C" int gd_adapter_fused(const void* x, const void* w1, const void* w2, const void* gate_src, void* hidden,
                                void* out, int M, int D, int bottleneck, int dtype, void* stream) {
    GD_REQUIRE(M > 0, "gd_adapter_fused: M = %d", M);
    GD_REQUIRE(gd_adapter_fused_supported(D, bottleneck, dtype),
               "gd_adapter_fused: unsupported configuration D=%d bottleneck=%d dtype=%d (bf16, bottleneck 64, D in {256,512,768,1024}); "
               "use two gd_gemm_nt calls", D, bottleneck, dtype);
    GD_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w1 & 15) == 0 && ((uintptr_t)w2 & 15) == 0 &&
                   ((uintptr_t)out & 15) == 0 && ((uintptr_t)hidden & 15) == 0 && ((uintptr_t)gate_src & 1) == 0,
               "gd_adapter_fused: x, w1, w2, out, hidden must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    switch (D) {
        case 256: adapter_launch<256>(x, w1, w2, gate_src, hidden, out, M, s); break;
        case 512: adapter_launch<512>(x, w1, w2, gate_src, hidden, out, M, s); break;
        case 768: adapter_launch<768>(x, w1, w2, gate_src, hidden, out, M, s); break;
        default: adapter_launch<1024>(x, w1, w2, gate_src, hidden, out, M, s); break;
    }
    GD_LAUNCH_OK();
    return 0;
}

// fp32 x / out with fp16 operands (tf32h engine): out32 = x32 + alpha * gate(fp16(x32 * in_scale) . w1^T) . w2^T, hidden [M, 64] fp16 (the gated
// first product, still times in_scale), out16 (optional) = fp16(out32 * copy_scale); in_scale / alpha / copy_scale are DEVICE scalars (null = 1).
// gd_adapter_fused_h plus the NEXT block's LayerNorm 1 of the result, from the rows the kernel still holds on chip: ln16 [M, D] = fp16(LayerNorm(out32)),
// ln_mean / ln_rstd [M] (what gd_layernorm_bwd* takes).  Forward form only (no gate, no scales).
extern "C" int gd_adapter_fused_h_ln(const float* x32, const void* w1, const void* w2, void* hidden, float* out32, const float* ln_w, const float* ln_b,
                                     float ln_eps, void* ln16, float* ln_mean, float* ln_rstd, int M, int D, int bottleneck, void* stream) {
    GD_REQUIRE(M > 0 && bottleneck == AD_BOT && (D == 256 || D == 512 || D == 768) && M >= 256L * AD_BM,
               "gd_adapter_fused_h_ln: unsupported configuration M=%d D=%d bottleneck=%d (bottleneck 64, D in {256,512,768}, M >= 8192)", M, D, bottleneck);
    GD_REQUIRE(ln_w && ln_b && ln16 && ln_mean && ln_rstd, "gd_adapter_fused_h_ln: the LayerNorm affine and all three outputs are required");
    GD_REQUIRE(((uintptr_t)x32 & 15) == 0 && ((uintptr_t)w1 & 15) == 0 && ((uintptr_t)w2 & 15) == 0 && ((uintptr_t)out32 & 15) == 0 &&
                   ((uintptr_t)hidden & 15) == 0 && ((uintptr_t)ln16 & 7) == 0,
               "gd_adapter_fused_h_ln: x32, w1, w2, out32, hidden must be 16-byte aligned, ln16 8-byte");
    hipStream_t s = (hipStream_t)stream;
    const int cus = gd_knobs().ncu, blocks = min(cus, gd_cdiv(M, AD_BM));
#define GD_ADL(DD) hipLaunchKernelGGL((adapter_persist_h_kernel<DD, true>), dim3(blocks), dim3(256), 0, s, x32, (const f16*)w1, (const f16*)w2, (const f16*)nullptr, \
                                      (f16*)hidden, out32, (f16*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, M, ln_w, ln_b, ln_eps, \
                                      (f16*)ln16, ln_mean, ln_rstd)
    switch (D) {
        case 256: GD_ADL(256); break;
        case 512: GD_ADL(512); break;
        default: GD_ADL(768); break;
    }
#undef GD_ADL
    GD_LAUNCH_OK();
    return 0;
}

extern "C" int gd_adapter_fused_h_supported(int D, int bottleneck, long M) { return bottleneck == AD_BOT && (D == 256 || D == 512 || D == 768) && M >= 256L * AD_BM; }
extern "C" int gd_adapter_fused_h(const float* x32, const void* w1, const void* w2, const void* gate_src, void* hidden, float* out32, void* out16,
                                  const float* in_scale, const float* alpha_dev, const float* copy_scale, int M, int D, int bottleneck, void* stream) {
    GD_REQUIRE(M > 0 && gd_adapter_fused_h_supported(D, bottleneck, M),
               "gd_adapter_fused_h: unsupported configuration M=%d D=%d bottleneck=%d (bottleneck 64, D in {256,512,768}, M >= 8192)", M, D, bottleneck);
    GD_REQUIRE(((uintptr_t)x32 & 15) == 0 && ((uintptr_t)w1 & 15) == 0 && ((uintptr_t)w2 & 15) == 0 && ((uintptr_t)out32 & 15) == 0 &&
                   ((uintptr_t)hidden & 15) == 0 && ((uintptr_t)out16 & 7) == 0 && ((uintptr_t)gate_src & 1) == 0,
               "gd_adapter_fused_h: x32, w1, w2, out32, hidden must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int cus = gd_knobs().ncu, blocks = min(cus, gd_cdiv(M, AD_BM));
#define GD_ADH(DD) hipLaunchKernelGGL(adapter_persist_h_kernel<DD>, dim3(blocks), dim3(256), 0, s, x32, (const f16*)w1, (const f16*)w2, (const f16*)gate_src, \
                                      (f16*)hidden, out32, (f16*)out16, in_scale, alpha_dev, copy_scale, M)
    switch (D) {
        case 256: GD_ADH(256); break;
        case 512: GD_ADH(512); break;
        default: GD_ADH(768); break;
    }
#undef GD_ADH
    GD_LAUNCH_OK();
    return 0;
}
