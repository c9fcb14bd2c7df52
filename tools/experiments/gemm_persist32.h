// EXPERIMENT (round 3; not built into the library): the persistent 256 x 256 gemm_nt main loop on v_mfma_f32_32x32x16_bf16.
//
// Why: profiles/r02_micro_mfma_gap.txt — back-to-back 16x16x32 bf16 MFMAs issue at 18.0 cycles per 16-cycle instruction (19-20 with one or two
// other instructions between them), 32x32x16 at 32.0-33.0 per 32-cycle instruction with up to four fillers: the bigger instruction leaves the
// SIMD's issue port free for the loop's LDS reads and DMA issue, the smaller one pays for them.  Same LDS-DMA ring, LDS image, XOR swizzle, 8 waves
// 2 x 4 with 128 x 64 wave tiles as gemm_nt_persist_kernel; per 16-wide k-step a wave reads 4 A + 2 B fragments (16 bytes per lane: row lane % 32,
// k = 8 (lane / 32) .. + 7) and issues 8 MFMAs into 4 x 2 accumulator tiles of 32 x 32 (16 registers each), the reads of k-step s + 1 in flight under
// the MFMAs of step s; W rows in natural order, plain epilogue only (C = alpha A W^T as bf16, 2-byte stores: 32 lanes = 64 contiguous bytes of a row).
// Selected with gd_debug_set("gemm_persist", 32) in a -DGD_GEMM_EXPERIMENT32 build; ANAT as in gemm_persist.h (0 product, 1 no operand DMA in the main
// loop, 4 no C stores).
#pragma once
#include "gemm_persist.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define GD_P32_READ6(F, sb, ks)                                                                              \
    GD_DSR128(F[0], (sb) + abase + co[ks], 0);     GD_DSR128(F[1], (sb) + abase + co[ks], 4096);                \
    GD_DSR128(F[2], (sb) + abase + co[ks], 8192);  GD_DSR128(F[3], (sb) + abase + co[ks], 12288);               \
    GD_DSR128(F[4], (sb) + bbase + co[ks], 0);     GD_DSR128(F[5], (sb) + bbase + co[ks], 4096);
#define GD_P32_WAIT(F, N)                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]))
#define GD_P32_MMA(F)                                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)            \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F[i_]), __builtin_bit_cast(bf16x8, F[4 + j_]), acc[i_][j_], 0, 0, 0);

template <int ANAT>
__global__ __launch_bounds__(512) void gemm_nt_p32_kernel(GemmNtParams p) {
    constexpr int NWN = 4, NW = 8, BM = 256, BN = 256;
    constexpr int ABYTES = BM * 128, STAGE = (BM + BN) * 128, APW = BM / 8 / NW, BPW = BN / 8 / NW;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int l32 = lane & 31, h = lane >> 5;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    const long batch = blockIdx.y;
    const char* Ab = (const char*)p.A + batch * p.sA * 2L;
    const char* Wb = (const char*)p.W + batch * p.sW * 2L;
    const long lda_b = p.lda * 2L, ldw_b = p.ldw * 2L;
    const int nk = p.K * 2 / 128;
    bf16* Cb = (bf16*)p.C + batch * p.sC;

    const char* abase_t;
    const char* wbase_t;
    unsigned aoff[APW], woff[BPW];
    int krot = 0;
    auto set_tile = [&](int tm, int tn) __attribute__((always_inline)) {
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
            woff[i] = (unsigned)(min(row, wv) * (int)ldw_b + ((lane & 7) ^ swz(row)) * 16);
        }
    };
    auto issue = [&](int kt0, int buf) __attribute__((always_inline)) {
        if (ANAT == 1) return;
        const int kt = kt0 + krot >= nk ? kt0 + krot - nk : kt0 + krot;
        char* sA = smem + buf * STAGE;
        char* sB = sA + ABYTES;
#pragma unroll
        for (int i = 0; i < APW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(abase_t + kt * 128 + aoff[i]),
                                             (__attribute__((address_space(3))) void*)(sA + (wave * APW + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase_t + kt * 128 + woff[i]),
                                             (__attribute__((address_space(3))) void*)(sB + (wave * BPW + i) * 1024), 16, 0, 0);
    };
    // fragment of tile row (wm * 128 + 32 mt + l32), k-step ks (16 k = 32 bytes), half h: 16-byte chunk 2 ks + h, XOR-swizzled with (row >> 1) & 7
    const int abase = (wm * 128 + l32) * 128, bbase = ABYTES + (wn * 64 + l32) * 128;
    const int sa = (l32 >> 1) & 7;
    int co[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) co[ks] = ((2 * ks + h) ^ sa) * 16;
    const unsigned lds0 = lds_off(smem);

    int t = blockIdx.x;
    if (t >= ntiles) return;
    int wg = xcd_remap(t, ntiles);
    int tm = wg / tiles_n, tn = wg % tiles_n;
    set_tile(tm, tn);
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    for (;;) {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
        // stage 0 landed (the previous tile's C stores are older than this tile's DMA: a full drain is the simple, safe wait of an experiment)
        if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        f32x4 F0[6], F1[6];
        GD_P32_READ6(F0, lds0, 0)
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned sbo = lds0 + (kt & 1) * STAGE, nsbo = lds0 + ((kt + 1) & 1) * STAGE;
            GD_P32_READ6(F1, sbo, 1)
            GD_P32_WAIT(F0, 6);
            GD_P32_MMA(F0)
            GD_P32_READ6(F0, sbo, 2)
            GD_P32_WAIT(F1, 6);
            GD_P32_MMA(F1)
            GD_P32_READ6(F1, sbo, 3)
            GD_P32_WAIT(F0, 6);
            GD_P32_MMA(F0)
            // every LDS read of stage kt has been issued; they have to be complete before the slot is refilled
            GD_P32_WAIT(F1, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stage kt + 1 landed
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 2 < nk) issue(kt + 2, kt & 1);
            if (kt + 1 < nk) { GD_P32_READ6(F0, nsbo, 0) }
            GD_P32_MMA(F1)
        }
        const int ctm = tm, ctn = tn;
        t += gridDim.x;
        const bool more = t < ntiles;
        if (more) {      // next tile's first stages before the stores
            wg = xcd_remap(t, ntiles);
            tm = wg / tiles_n; tn = wg % tiles_n;
            set_tile(tm, tn);
            issue(0, 0);
            if (nk > 1) issue(1, 1);
        }
        // C layout of a 32 x 32 tile: lane (n = lane % 32, h = lane / 32), register v: row 8 (v / 4) + 4 h + (v % 4)
        if (ANAT != 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = ctn * BN + wn * 64 + j * 32 + l32;
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int row = ctm * BM + wm * 128 + i * 32 + 8 * (v >> 2) + 4 * h + (v & 3);
                        if (row < p.M && col < p.N) Cb[(long)row * p.ldc + col] = (bf16)(p.alpha * acc[i][j][v]);
                    }
                }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(acc[i][j]));
        }
        if (!more) break;      // (the next tile's DMA is OLDER than this tile's stores: its counted waits are conservative, never early)
    }
}
