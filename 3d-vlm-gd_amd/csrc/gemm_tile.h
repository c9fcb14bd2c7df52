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
