// Shared 128x128 MFMA main loop: acc[i][j] (4x4 tiles of 16x16 per wave, waves 2x2) =
// A[tm*128 .. +128, :K] . W[tn*128 .. +128, :K]^T, both operands K-contiguous.
// K step = 128 bytes of a row (64 bf16 / 32 f32); global->register->LDS staging with the
// next tile's loads in flight under the MFMAs; LDS rows padded 128 -> 144 bytes.
// Rows >= Mrows / Nrows and bytes >= Kbytes read as zero.  256 threads.
#pragma once
#include "gd_common.h"

#define GD_TILE_ROWB 144
#define GD_TILE_SMEM (256 * GD_TILE_ROWB)

template <typename T>
__device__ __forceinline__ void mma_tile_128x128(const char* Ab, long lda_b, int Mrows, const char* Wb, long ldw_b,
                                                 int Nrows, int Kbytes, int tm, int tn, char* smem,
                                                 f32x4 (&acc)[4][4]) {
    constexpr int BM = 128, BN = 128, BKB = 128, ROWB = GD_TILE_ROWB;
    char* sA = smem;
    char* sB = smem + BM * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = (Kbytes + BKB - 1) / BKB;

    uint4 ra[4], rb[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, row = c >> 3, cc = c & 7;
            const int kb = kt * BKB + cc * 16;
            const int gr = tm * BM + row, gc = tn * BN + row;
            const bool kin = kb < Kbytes;
            ra[i] = (kin && gr < Mrows) ? *(const uint4*)(Ab + (long)gr * lda_b + kb) : make_uint4(0, 0, 0, 0);
            rb[i] = (kin && gc < Nrows) ? *(const uint4*)(Wb + (long)gc * ldw_b + kb) : make_uint4(0, 0, 0, 0);
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, row = c >> 3, cc = c & 7;
            *(uint4*)(sA + row * ROWB + cc * 16) = ra[i];
            *(uint4*)(sB + row * ROWB + cc * 16) = rb[i];
        }
    };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    typedef typename Mma<T>::Frag Frag;
    const int frow = lane & 15, fcol = (lane >> 4) * 16;
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
        swrite();
        __syncthreads();
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            Frag a[4], b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[t] = *(const Frag*)(sA + (wm * 64 + t * 16 + frow) * ROWB + kc * 64 + fcol);
                b[t] = *(const Frag*)(sB + (wn * 64 + t * 16 + frow) * ROWB + kc * 64 + fcol);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------
// LDS-DMA main loop (global_load_lds_dwordx4) with a 2-deep LDS ring and one barrier per K-step.
//   NWM x NWN waves; each wave owns WMT x 4 MFMA tiles (16*WMT rows x 64 columns); block tile BM x BN.
//   acc[i][j] = A[tm*BM + wm*16*WMT + 16 i .., :] . W[tn*BN + wn*64 + 16 j .., :]^T over nk K-steps of 128 bytes.
// LDS image per stage: A rows then W rows, 128 B per row, 16-byte chunk c of row r stored at chunk c ^ ((r>>1)&7)
// (the XOR is applied to the per-lane SOURCE address; the DMA destination stays lane-linear), which makes the
// ds_read_b128 fragment reads conflict-free.  Rows >= Mrows / Nrows are clamped (their products are never stored).
// smem must hold 2*(BM+BN)*128 bytes; on return every wave has passed the final barrier (smem reusable).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }
#ifndef GD_K_ROTATION
#define GD_K_ROTATION 0
#endif
__device__ __forceinline__ constexpr bool gd_k_rotation() { return GD_K_ROTATION != 0; }

template <typename T, int NWM, int NWN, int WMT>
__device__ __forceinline__ void dma_mainloop(const char* Ab, long lda_b, int Mrows, const char* Wb, long ldw_b,
                                             int Nrows, int nk, int tm, int tn, char* smem, f32x4 (&acc)[WMT][4]) {
    constexpr int NW = NWM * NWN, BM = NWM * WMT * 16, BN = NWN * 64;
    constexpr int ABYTES = BM * 128, STAGE = (BM + BN) * 128;
    constexpr int APW = BM / 8 / NW, BPW = BN / 8 / NW;   // 1-KB DMA pieces per wave per operand per K-step
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;

    const char* asrc[APW];
    const char* wsrc[BPW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int row = (wave * APW + i) * 8 + (lane >> 3);
        asrc[i] = Ab + (long)min(tm * BM + row, Mrows - 1) * lda_b + (((lane & 7) ^ swz(row)) * 16);
    }
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int row = (wave * BPW + i) * 8 + (lane >> 3);
        wsrc[i] = Wb + (long)min(tn * BN + row, Nrows - 1) * ldw_b + (((lane & 7) ^ swz(row)) * 16);
    }
    // K-steps are walked in a per-tile ROTATED order: tiles that share an operand panel (same tm or same tn) run
    // concurrently on one XCD and, in lockstep, would all miss the L2 on the same K-slice at the same time (measured:
    // 7.5x the compulsory L2 misses); rotated, a slice is fetched by one tile and hit by the others.
    const int rot = gd_k_rotation() ? (tn + 5 * tm) % nk : 0;
    auto issue = [&](int kt0, int buf) {
        const int kt = (kt0 + rot >= nk) ? kt0 + rot - nk : kt0 + rot;
        char* sA = smem + buf * STAGE;
        char* sB = sA + ABYTES;
#pragma unroll
        for (int i = 0; i < APW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + (long)kt * 128),
                                             (__attribute__((address_space(3))) void*)(sA + (wave * APW + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + (long)kt * 128),
                                             (__attribute__((address_space(3))) void*)(sB + (wave * BPW + i) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    typedef typename Mma<T>::Frag Frag;
    const int fr = lane & 15, g = lane >> 4;
    const int abase = (wm * WMT * 16 + fr) * 128, bbase = ABYTES + (wn * 64 + fr) * 128;
    const int sa = swz(fr);  // every row this lane reads is fr + multiple of 16: (row>>1)&7 == (fr>>1)&7

    issue(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const char* sb = smem + (kt & 1) * STAGE;
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int co = (((kc * 4 + g) ^ sa) * 16);
            Frag a[WMT], b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = *(const Frag*)(sb + bbase + t * 2048 + co);
#pragma unroll
            for (int t = 0; t < WMT; ++t) a[t] = *(const Frag*)(sb + abase + t * 2048 + co);
#pragma unroll
            for (int i = 0; i < WMT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
}
