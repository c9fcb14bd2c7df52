// Persistent 256 x 256 gemm_nt, FOUR waves of 128 x 128 (one wave per SIMD, the whole 512-register file per lane: 256 accumulator
// registers + two fragment sets), gfx950.  Same LDS-DMA ring, LDS image and XOR swizzle as gemm_nt_persist_kernel; what changes:
//   * LDS fragment reads per 128-byte K step: 4 waves x 32 KB = 128 KB instead of 8 x 24 KB = 192 KB;
//   * no second wave on the SIMD: the MFMA pipe is not shared, a wave's LDS reads are software-pipelined one 64-byte chunk ahead
//     into a second fragment set (F0 / F1) instead of being hidden by a partner wave;
//   * W rows permuted n = 8 fr + j (nperm128): after the MFMAs a lane holds EIGHT consecutive output columns per row, one 16-byte
//     bf16 store per row item, 16 lanes = one 256-byte row segment.
// Round 3: experimental, plain epilogue only (C = alpha * A W^T, bf16); selected with gd_debug_set("gemm_persist", 4).
#pragma once
#include "gemm_persist.h"

__device__ __forceinline__ int nperm128(int rho) {   // LDS W-row (128 w + 16 j + fr) -> column 128 w + 8 fr + j of the 256-wide tile
    return (rho & ~127) | ((rho & 15) << 3) | ((rho >> 4) & 7);
}

#define GD_P4_ROW(i, AF, BF)                                                                                         \
    {                                                                                                                \
        const bf16x8 a_ = __builtin_bit_cast(bf16x8, AF[i]);                                                         \
        _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_)                                                             \
            acc[i][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, __builtin_bit_cast(bf16x8, BF[j_]), acc[i][j_], 0, 0, 0); \
    }
#define GD_P4_WAIT16(AF, BF)                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                              \
                 : "+v"(AF[0]), "+v"(AF[1]), "+v"(AF[2]), "+v"(AF[3]), "+v"(AF[4]), "+v"(AF[5]), "+v"(AF[6]), "+v"(AF[7]),   \
                   "+v"(BF[0]), "+v"(BF[1]), "+v"(BF[2]), "+v"(BF[3]), "+v"(BF[4]), "+v"(BF[5]), "+v"(BF[6]), "+v"(BF[7]))

// ANAT: 0 product, 1 no operand DMA in the main loop, 3 DMA ring alone (no LDS reads, no MFMAs)
template <int ANAT>
__global__ __launch_bounds__(256) void gemm_nt_p4_kernel(GemmNtParams p) {
    constexpr int NW = 4, BM = 256, BN = 256;
    constexpr int ABYTES = BM * 128, STAGE = (BM + BN) * 128, APW = BM / 8 / NW, BPW = BN / 8 / NW;   // 8 + 8 one-KB pieces per wave and stage
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, g = lane >> 4;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    const long batch = blockIdx.y;
    const char* Ab = (const char*)p.A + batch * p.sA * 2L;
    const char* Wb = (const char*)p.W + batch * p.sW * 2L;
    const long lda_b = p.lda * 2L, ldw_b = p.ldw * 2L;
    const int nk = p.K * 2 / 128;
    char* Cb = (char*)p.C + batch * p.sC * 2L;

    const char* abase_t;
    const char* wbase_t;
    unsigned aoff[APW], woff[BPW];
    int krot = 0;
    auto set_tile = [&](int tm, int tn) {
        krot = p.k_rot ? (tn * p.k_rot + tm) % nk : 0;
        abase_t = Ab + (long)tm * BM * lda_b;
        wbase_t = Wb + (long)tn * BN * ldw_b;
        const int av = min(BM, p.M - tm * BM) - 1, wv = min(BN, p.N - tn * BN) - 1;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int row = (wave * APW + i) * 8 + (lane >> 3);
            aoff[i] = (unsigned)(min(row, av) * (int)lda_b + ((lane & 7) ^ swz(row)) * 16);
        }
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const int row = (wave * BPW + i) * 8 + (lane >> 3);
            woff[i] = (unsigned)(min(nperm128(row), wv) * (int)ldw_b + ((lane & 7) ^ swz(row)) * 16);
        }
    };
    auto dma_a = [&](int kt, int buf, int i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(abase_t + kt * 128 + aoff[i]),
                                         (__attribute__((address_space(3))) void*)(smem + buf * STAGE + (wave * APW + i) * 1024), 16, 0, 0);
    };
    auto dma_w = [&](int kt, int buf, int i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase_t + kt * 128 + woff[i]),
                                         (__attribute__((address_space(3))) void*)(smem + buf * STAGE + ABYTES + (wave * BPW + i) * 1024), 16, 0, 0);
    };
    auto rot = [&](int kt0) { return kt0 + krot >= nk ? kt0 + krot - nk : kt0 + krot; };
    auto issue = [&](int kt0, int buf) {
        const int kt = rot(kt0);
#pragma unroll
        for (int i = 0; i < APW; ++i) dma_a(kt, buf, i);
#pragma unroll
        for (int i = 0; i < BPW; ++i) dma_w(kt, buf, i);
    };

    const int sa = swz(fr);
    const unsigned lds0 = lds_off(smem);
    const unsigned abase = lds0 + (wm * 128 + fr) * 128, bbase = lds0 + ABYTES + (wn * 128 + fr) * 128;
    const unsigned co0 = ((g ^ sa) * 16), co1 = (((4 + g) ^ sa) * 16);

    int t = blockIdx.x;
    if (t >= ntiles) return;
    int wg = xcd_remap(t, ntiles);
    int tm = wg / tiles_n, tn = wg % tiles_n;
    set_tile(tm, tn);
    issue(0, 0);
    if (nk > 1) issue(1, 1);

    for (;;) {
        f32x4 acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 A0[8], B0[8], A1[8], B1[8];
        // stage 0 of this tile has landed (everything older too: the previous epilogue's stores drain here in this experimental form)
        if (nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ANAT != 3) {
            const unsigned a = abase + co0, b = bbase + co0;
            GD_DSR128(B0[0], b, 0); GD_DSR128(B0[1], b, 2048); GD_DSR128(B0[2], b, 4096); GD_DSR128(B0[3], b, 6144);
            GD_DSR128(B0[4], b, 8192); GD_DSR128(B0[5], b, 10240); GD_DSR128(B0[6], b, 12288); GD_DSR128(B0[7], b, 14336);
            GD_DSR128(A0[0], a, 0); GD_DSR128(A0[1], a, 2048); GD_DSR128(A0[2], a, 4096); GD_DSR128(A0[3], a, 6144);
            GD_DSR128(A0[4], a, 8192); GD_DSR128(A0[5], a, 10240); GD_DSR128(A0[6], a, 12288); GD_DSR128(A0[7], a, 14336);
        }
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned so = (kt & 1) * STAGE, nso = ((kt + 1) & 1) * STAGE;
            if (ANAT != 3) {
                // ---- chunk 0 from F0; the stage's second chunk streams into F1 underneath, two reads per MFMA row
                const unsigned a = abase + so + co1, b = bbase + so + co1;
                GD_P4_WAIT16(A0, B0);
                __builtin_amdgcn_sched_barrier(0);
                GD_P4_ROW(0, A0, B0) GD_DSR128(B1[0], b, 0); GD_DSR128(B1[1], b, 2048); __builtin_amdgcn_sched_barrier(0);
                GD_P4_ROW(1, A0, B0) GD_DSR128(B1[2], b, 4096); GD_DSR128(B1[3], b, 6144); __builtin_amdgcn_sched_barrier(0);
                GD_P4_ROW(2, A0, B0) GD_DSR128(B1[4], b, 8192); GD_DSR128(B1[5], b, 10240); __builtin_amdgcn_sched_barrier(0);
                GD_P4_ROW(3, A0, B0) GD_DSR128(B1[6], b, 12288); GD_DSR128(B1[7], b, 14336); __builtin_amdgcn_sched_barrier(0);
                GD_P4_ROW(4, A0, B0) GD_DSR128(A1[0], a, 0); GD_DSR128(A1[1], a, 2048); __builtin_amdgcn_sched_barrier(0);
                GD_P4_ROW(5, A0, B0) GD_DSR128(A1[2], a, 4096); GD_DSR128(A1[3], a, 6144); __builtin_amdgcn_sched_barrier(0);
                GD_P4_ROW(6, A0, B0) GD_DSR128(A1[4], a, 8192); GD_DSR128(A1[5], a, 10240); __builtin_amdgcn_sched_barrier(0);
                GD_P4_ROW(7, A0, B0) GD_DSR128(A1[6], a, 12288); GD_DSR128(A1[7], a, 14336); __builtin_amdgcn_sched_barrier(0);
                // ---- chunk 1 from F1: six rows, then the stage barrier, then the last two rows cover the refill + the next reads
                GD_P4_WAIT16(A1, B1);
                __builtin_amdgcn_sched_barrier(0);
                GD_P4_ROW(0, A1, B1) GD_P4_ROW(1, A1, B1) GD_P4_ROW(2, A1, B1) GD_P4_ROW(3, A1, B1) GD_P4_ROW(4, A1, B1) GD_P4_ROW(5, A1, B1)
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stage kt+1 landed (this wave's pieces)
            __builtin_amdgcn_s_barrier();                          // ... everybody's; and slot kt&1 is free
            asm volatile("" ::: "memory");
            const bool refill = ANAT != 1 && kt + 2 < nk;
            const bool more_k = kt + 1 < nk;
            const int ktr = refill ? rot(kt + 2) : 0;
            if (ANAT != 3) {
                const unsigned a = abase + nso + co0, b = bbase + nso + co0;
                const bf16x8 a6 = __builtin_bit_cast(bf16x8, A1[6]), a7 = __builtin_bit_cast(bf16x8, A1[7]);
#define GD_P4_TAIL(q, OFF)                                                                                              \
                if (refill) { if (q < 8) dma_a(ktr, kt & 1, q); else dma_w(ktr, kt & 1, q - 8); }                          \
                if (more_k) { if (q < 8) GD_DSR128(B0[q & 7], b, OFF); else GD_DSR128(A0[q & 7], a, OFF); }                \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
                if (q < 8) acc[6][q & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a6, __builtin_bit_cast(bf16x8, B1[q & 7]), acc[6][q & 7], 0, 0, 0); \
                else acc[7][q & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a7, __builtin_bit_cast(bf16x8, B1[q & 7]), acc[7][q & 7], 0, 0, 0);       \
                __builtin_amdgcn_sched_barrier(0);
                GD_P4_TAIL(0, 0) GD_P4_TAIL(1, 2048) GD_P4_TAIL(2, 4096) GD_P4_TAIL(3, 6144) GD_P4_TAIL(4, 8192) GD_P4_TAIL(5, 10240)
                GD_P4_TAIL(6, 12288) GD_P4_TAIL(7, 14336) GD_P4_TAIL(8, 0) GD_P4_TAIL(9, 2048) GD_P4_TAIL(10, 4096) GD_P4_TAIL(11, 6144)
                GD_P4_TAIL(12, 8192) GD_P4_TAIL(13, 10240) GD_P4_TAIL(14, 12288) GD_P4_TAIL(15, 14336)
#undef GD_P4_TAIL
            } else if (refill) {
                issue(kt + 2, kt & 1);
            }
        }
        // ---- epilogue: item (i, r) = row 16 i + 4 g + r of the wave tile, this lane's eight columns 8 fr .. 8 fr + 7
        const int ctm = tm, ctn = tn;
        const int vrows = min(BM, p.M - ctm * BM);
        const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)(Cb + (long)ctm * BM * p.ldc * 2), (short)0,
                                                                              (int)min((long)0x7fffffff, (long)vrows * p.ldc * 2), 0x00020000);
        const int rloc = wm * 128 + g * 4, col0 = ctn * BN + wn * 128 + fr * 8;
        const int ldc_i = (int)p.ldc;
        const int cbase = col0 < p.N ? (rloc * ldc_i + col0) * 2 : 0x7ffffff0;
        t += gridDim.x;
        const bool more = t < ntiles;
        if (more) {
            wg = xcd_remap(t, ntiles);
            tm = wg / tiles_n; tn = wg % tiles_n;
            set_tile(tm, tn);
            issue(0, 0);
            if (nk > 1) issue(1, 1);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int coff = cbase + (i * 16 + r) * 2 * ldc_i;
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)(p.alpha * acc[i][j][r]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gd_u32x4, o), crs, coff, 0, GD_PERSIST_STORE_AUX);
            }
        if (!more) break;
    }
}
