// SHELVED EXPERIMENT (round 3; not built into the library): cost-volume KL forward on the persistent 256 x 256 tile skeleton of gemm_persist.h.
// Result on MI355X (32 pairs, hw = 1369, C = 768, bf16; loss bit-identical to the shipped kernel, all 26 cost-volume tests green): 291 us against
// 263 us for cv_fwd_persist_kernel with every row kept, 221 against 193 with 20 % keypoint-patch masks.  The 256 x 256 main loop runs at the GEMM's
// pace (~23 us per tile), but the two-direction epilogue costs another 23-35 us per tile with the MFMA pipe idle — the same VALU work per score as
// the 128 x 128 kernel's, no longer hidden behind anything — and 6 x 6 tiles of 256 cover 1.26 x the 1369 x 1369 scores.  Deeper teacher prefetch
// (4 -> 12 items in flight) changed nothing: the epilogue is issue-bound, not latency-bound.  To integrate again: include from cost_volume.hip,
// rowtab + 4 / 2 slabs per tile row / column in the workspace, cv_finalize_kernel with two slab counts.
//
// Why: the 128 x 128 warp-specialised kernel (cv_fwd_persist_kernel) stages 393 KB of features through LDS per 25 MFLOP tile — 64 FLOP per
// staged byte against the GEMM kernel's 128 — and its anatomy shows the feature ring alone at 137 of 281 us: the
// per-CU landing rate of the LDS-DMA, not HBM, set the kernel's pace.  Here a 512-thread block computes a 256 x 256 tile of S = a b^T
// with the GEMM's main loop (same DMA ring, swizzle, permuted W rows, software-pipelined K chunks: 8 waves 2 x 4, wave tile 128 x 64,
// 128 accumulator registers) and then runs the WHOLE two-direction epilogue from the accumulators while the teacher entries stream
// through registers:
//   item (i, r) = tile row 16 i + 4 g + r of the wave tile, this lane's four consecutive columns 4 fr .. 4 fr + 3 (nperm64);
//   direction 1 (rows of S, teacher T1[row, col..col+3]): one 16-byte load per item, 16 lanes = 256 contiguous bytes of a teacher row;
//   direction 2 (columns of S, teacher T2[col, row..row+3]): one 16-byte load per (i, column) — rows 16 i + 4 g .. + 3 are four consecutive
//   entries of teacher row `col` — shared by the four items r = 0..3;
//   per m-tile i: 8 loads (4 + 4) for 16 scores, double-buffered one m-tile ahead (64 registers);
//   per score: scale (acc * inv1 * inv2 * log2 e), ONE v_exp (both directions' Z), and mul + max + fma per direction (B = sum t s);
//   row partials (Z, B) leave per item through a 16-lane DPP sum as one 8-byte store into slab [tn * 4 + wn]; column partials are
//   carried in registers over the 32 items and leave once per tile into slab [tm * 2 + wm] — no LDS, no barrier, deterministic.
// Rows / columns past hw carry inv = 0 in the staged table (their exp2(0) = 1 is subtracted as a count); rows / columns the loss masks out
// carry 1 / rowsum = 0 and their teacher loads are pointed at the pair's first line (an L2 hit), as in the 128 x 128 kernel.
#pragma once
#include "gemm_frag.h"
#include <utility>

struct Cv256Params {
    const void* f1; const void* f2;      // [P][hw][C]
    const float* t1; const float* t2;    // [P][hw][ldt]
    const float* rowtab;                 // [P][2][hwp] float2 {inv_norm, keep ? 1 / teacher_rowsum : 0}, zero beyond hw (hwp = tiles * 256)
    float* part1; float* part2;          // [P][4 tiles][hw][2], [P][2 tiles][hw][2]: {Z partial, B partial}
    int hw, C, ldt, P, tiles;
};

// the per-row table the tile kernel stages: {inv_norm, keep ? 1 / max(teacher rowsum, eps) : 0}; entries hw .. hwp-1 stay zero
__global__ __launch_bounds__(256) void cv256_rowtab_kernel(const float* stats, const unsigned char* m1, const unsigned char* m2, float* rowtab,
                                                           int hw, int hwp, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // (p, which, row < hwp)
    if (i >= n) return;
    const long pw = i / hwp;
    const int row = (int)(i - pw * hwp);
    f32x2 o = {0.f, 0.f};
    if (row < hw) {
        const float* st = stats + (pw * hw + row) * 4;
        const unsigned char* m = (pw & 1) ? m2 : m1;
        const bool keep = m == nullptr || m[(pw >> 1) * hw + row] != 0;
        o = f32x2{st[0], keep ? 1.0f / st[1] : 0.f};
    }
    *(f32x2*)(rowtab + i * 2) = o;
}

// one table entry from LDS, through inline asm: a ds_read the compiler can see gets an `s_waitcnt vmcnt(0)` in front of it as soon as an LDS-DMA is
// in flight (the next tile's prefetch, from the middle of the epilogue on)
template <int OFF>
__device__ __forceinline__ f32x2 cv256_tab(unsigned addr) {
    f32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// compile-time index loop (array indices and instruction offsets are constants when the IR is built)
template <typename F, int... I>
__device__ __forceinline__ void cv_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void cv_static_for(F&& f) { cv_static_for_impl(f, std::make_integer_sequence<int, N>{}); }      // (gd_common.h has a static_for of its own since round 4)

template <typename T>
__global__ __launch_bounds__(512) void cv_fwd_p256_kernel(Cv256Params p) {
    constexpr int NWN = 4, WMT = 8, NW = 8, BM = 256, BN = 256;
    constexpr int ABYTES = BM * 128, STAGE = (BM + BN) * 128, APW = BM / 8 / NW, BPW = BN / 8 / NW;
    constexpr int TAB_OFF = 2 * STAGE;                          // 2 (tile parity) x 512 float2
    __shared__ __attribute__((aligned(16))) char smem[TAB_OFF + 2 * 4096];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int fr = lane & 15, g = lane >> 4;
    const int hw = p.hw, tiles = p.tiles, ldt = p.ldt, hwp = tiles * 256;
    const int t2n = tiles * tiles, ntiles = p.P * t2n;
    const long rowb = (long)p.C * sizeof(T);
    const int nk = (int)(rowb / 128);

    const char* abase_t;
    const char* wbase_t;
    unsigned aoff[APW], woff[BPW];
    int krot = 0;
    auto set_tile = [&](int pp, int tm, int tn) __attribute__((always_inline)) {
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));      // re-derive the per-lane row / swizzle terms per tile (a few VALU) instead of keeping 16 registers of them alive
        krot = (tn + tm) % nk;
        abase_t = (const char*)p.f1 + ((long)pp * hw + (long)tm * BM) * rowb;
        wbase_t = (const char*)p.f2 + ((long)pp * hw + (long)tn * BN) * rowb;
        const int av = min(BM, hw - tm * BM) - 1, wv = min(BN, hw - tn * BN) - 1;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int row = (wave * APW + i) * 8 + (lane >> 3);
            aoff[i] = (unsigned)(min(row, av) * (int)rowb + ((lane & 7) ^ swz(row)) * 16);
        }
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const int row = (wave * BPW + i) * 8 + (lane >> 3);
            woff[i] = (unsigned)(min(nperm64(row), wv) * (int)rowb + ((lane & 7) ^ swz(row)) * 16);
        }
    };
    auto issue = [&](int kt0, int buf) __attribute__((always_inline)) {
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
    // the tile's 256 row and 256 column table entries (8 bytes each): four 1-KB pieces, one per wave 0..3
    auto issue_tab = [&](int pp, int tm, int tn, int slot) __attribute__((always_inline)) {
        if (wave < 4) {
            const int which = wave >> 1;
            const long e = ((long)pp * 2 + which) * hwp + (which ? tn : tm) * 256 + (wave & 1) * 128 + lane * 2;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.rowtab + e * 2),
                                             (__attribute__((address_space(3))) void*)(smem + TAB_OFF + slot * 4096 + wave * 1024), 16, 0, 0);
        }
    };
    auto decode = [&](int l, int& pp, int& tm, int& tn) __attribute__((always_inline)) {
        pp = l / t2n;
        const int r = l - pp * t2n;
        tm = r / tiles;
        tn = r - tm * tiles;
    };

    const int abase = (wm * WMT * 16 + fr) * 128, bbase = ABYTES + (wn * 64 + fr) * 128;
    const int sa = swz(fr);
    const unsigned lds0 = lds_off(smem);

    int t = blockIdx.x, slot = 0;
    if (t >= ntiles) return;
    int pp, tm, tn;
    decode(xcd_remap(t, ntiles), pp, tm, tn);
    auto prologue = [&](int pp_, int tm_, int tn_, int slot_) __attribute__((always_inline)) {
        set_tile(pp_, tm_, tn_);
        issue(0, 0);
        issue_tab(pp_, tm_, tn_, slot_);
        if (nk > 1) issue(1, 1);
    };
    prologue(pp, tm, tn, slot);
    int after = 0;      // a LOWER bound of the vector-memory operations this wave issued after the current tile's stage-1 DMA (see the epilogue)

    for (;;) {
        f32x4 acc[WMT][4];
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        wait_vm_le(after + (nk > 1 ? APW + BPW : 0));      // stage 0 and the table have landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int co0 = ((g ^ sa) * 16), co1 = (((4 + g) ^ sa) * 16);
        FragHead P, Q;
        FragTail tl;
        frag_head_issue(P, lds0 + abase + co0, lds0 + bbase + co0);
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned sbo = lds0 + (kt & 1) * STAGE, nsbo = lds0 + ((kt + 1) & 1) * STAGE;
            chunk_rows05<T>(P, tl, sbo + abase + co0, acc);
            frag_head_issue(Q, sbo + abase + co1, sbo + bbase + co1);
            chunk_rows67<T>(P, tl, acc);
            chunk_rows05<T>(Q, tl, sbo + abase + co1, acc);
            if (kt == 0) wait_vm_le(after);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 2 < nk) issue(kt + 2, kt & 1);
            if (kt + 1 < nk) frag_head_issue(P, nsbo + abase + co0, nsbo + bbase + co0);
            chunk_rows67<T>(Q, tl, acc);
        }
        // ------------------------------------------------------------------------------------------ epilogue
        const int cp = pp, ctm = tm, ctn = tn;
        int rloc = wm * 128 + 4 * g;                               // tile row of item (i, r): rloc + 16 i + r
        int cloc = wn * 64 + 4 * fr;                               // tile column jb: cloc + jb
        asm volatile("" : "+v"(rloc), "+v"(cloc));                 // opaque per tile: nothing derived from them may be hoisted out of the tile loop
        const unsigned tab = lds0 + TAB_OFF + slot * 4096;         // entry e at tab + 8 e: rows 0..255, columns 256..511
        const unsigned trow = tab + 8 * rloc, tcol = tab + 8 * (256 + cloc);
        const int row0 = ctm * 256 + rloc, col0 = ctn * 256 + cloc;
        // teacher maps of this pair and the two partial-sum slabs through buffer resources: offsets past the end read as zero / are dropped
        const int mapb = hw * ldt * 4;
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.t1 + (long)cp * hw * ldt), (short)0, mapb, 0x00020000);
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.t2 + (long)cp * hw * ldt), (short)0, mapb, 0x00020000);
        const __amdgpu_buffer_rsrc_t rp1 = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.part1 + (((long)cp * (4 * tiles) + ctn * 4 + wn) * hw) * 2), (short)0, hw * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t rp2 = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.part2 + (((long)cp * (2 * tiles) + ctm * 2 + wm) * hw) * 2), (short)0, hw * 8, 0x00020000);
        // Two passes over the accumulators (one direction's teacher entries in registers at a time: both together do not fit beside the 128
        // accumulator registers).  Pass 1: the exponentials (both directions' Z) and direction 1; pass 2: direction 2.
        float inv2[4], zc[4];
        float fnic = 0.f;
        cv_static_for<4>([&](auto JB) __attribute__((always_inline)) {
            constexpr int jb = decltype(JB)::value;
            const f32x2 v = cv256_tab<8 * jb>(tcol);
            inv2[jb] = v[0] * 1.4426950408889634f;                  // log2(e) folded into the column scale
            zc[jb] = 0.f;
            fnic += v[0] == 0.f ? 1.f : 0.f;
        });
        const int c1off = (row0 * ldt + min(col0, ldt - 4)) * 4;   // T1[row0][col0]; + (16 i + r) * ldt * 4 per item
        // ---- pass 1: direction-1 entries as a rolling set of SIXTEEN items (four m-tiles, 64 registers: 16 KB per wave, 128 KB per CU in flight —
        // with four items the epilogue ran at the latency of 32 KB in flight: 58 us per tile) — item idx's quartet is re-loaded for item idx + 16
        constexpr int DA = 3, DB = 3;      // m-tiles of teacher entries in flight per pass (4 and 4 spill beside the 128 accumulators)
        f32x4 tA[4 * DA];
        auto tloadA = [&](auto I, auto R) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value, r = decltype(R)::value;
            const bool keep = cv256_tab<8 * (16 * i + r)>(trow)[1] != 0.f;
            tA[(4 * i + r) % (4 * DA)] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, keep ? c1off + (16 * i + r) * ldt * 4 : 0, 0, 0));
        };
        cv_static_for<DA>([&](auto I) __attribute__((always_inline)) {
            cv_static_for<4>([&](auto R) __attribute__((always_inline)) { tloadA(I, R); });
        });
        float fnir = 0.f;
        const int prow8 = row0 * 8;
        cv_static_for<WMT>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            cv_static_for<4>([&](auto R) __attribute__((always_inline)) {
                constexpr int r = decltype(R)::value;
                const f32x2 rv = cv256_tab<8 * (16 * i + r)>(trow);
                const float inv1 = rv[0], ir1 = rv[1];
                fnir += rv[0] == 0.f ? 1.f : 0.f;
                float zr = -fnic, b1 = 0.f;                                          // columns past hw contribute exp2(0) = 1 each
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const float s2 = acc[i][jb][r] * (inv1 * inv2[jb]);            // s * log2(e); 0 for a row / column past hw
                    const float e = __builtin_amdgcn_exp2f(s2);
                    zr += e; zc[jb] += e;
                    b1 = fmaf(fmaxf(tA[(4 * i + r) % (4 * DA)][jb] * ir1, 1e-8f), s2, b1);
                }
                asm volatile("" : "+v"(zc[0]), "+v"(zc[1]), "+v"(zc[2]), "+v"(zc[3]));      // the running column sums, materialised per item
                if constexpr (i + DA < WMT) tloadA(std::integral_constant<int, i + DA>{}, R);
                zr = row16_sum(zr);
                b1 = row16_sum(b1) * 0.6931471805599453f;
                if (fr == 0)      // rows past hw lie beyond the slab's last record: dropped by the hardware
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(gd_u32x2, f32x2{zr, b1}), rp1, prow8, (16 * i + r) * 8, 0);
            });
        });
        // ---- pass 2: direction 2, four m-tiles of entries in flight (four sets of four quartets).  The next tile's pipeline starts right after the
        // first four sets went out: they are older than its DMA (vector memory returns in issue order), the loads of m-tiles 4..7 (16, issued
        // unconditionally) younger: `after` = 16 (a lower bound is the safe side)
        float ir2[4], b2[4];
        int c2off[4], c2step[4];                                   // byte offset of T2[col0 + jb][row0] (the map's first line when the column is masked out)
        cv_static_for<4>([&](auto JB) __attribute__((always_inline)) {
            constexpr int jb = decltype(JB)::value;
            ir2[jb] = cv256_tab<8 * jb>(tcol)[1];
            b2[jb] = 0.f;
            c2off[jb] = ir2[jb] != 0.f ? ((col0 + jb) * ldt + row0) * 4 : 0;
            c2step[jb] = ir2[jb] != 0.f ? 64 : 0;
        });
        f32x4 tB[DB][4];
        auto tloadB = [&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value, b = i % DB;
            cv_static_for<4>([&](auto JB) __attribute__((always_inline)) {
                constexpr int jb = decltype(JB)::value;
                // (everything in the per-lane offset: a per-lane value in the instruction's scalar offset makes the compiler build a waterfall loop)
                tB[b][jb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2, c2off[jb] + i * c2step[jb], 0, 0));
            });
        };
        cv_static_for<DB>([&](auto I) __attribute__((always_inline)) { tloadB(I); });
        t += gridDim.x;
        const bool more = t < ntiles;
        after = 4 * (WMT - DB) >= 16 ? 16 : 8;
        asm volatile("" ::: "memory");
        if (more) {
            decode(xcd_remap(t, ntiles), pp, tm, tn);
            slot ^= 1;
            prologue(pp, tm, tn, slot);
        }
        asm volatile("" ::: "memory");   // nothing younger may be hoisted above the DMA: `after` counts on it
        cv_static_for<WMT>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value, b = i % DB;
            cv_static_for<4>([&](auto R) __attribute__((always_inline)) {
                constexpr int r = decltype(R)::value;
                const float inv1 = cv256_tab<8 * (16 * i + r)>(trow)[0];
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const float s2 = acc[i][jb][r] * (inv1 * inv2[jb]);
                    b2[jb] = fmaf(fmaxf(tB[b][jb][r] * ir2[jb], 1e-8f), s2, b2[jb]);
                }
                asm volatile("" : "+v"(b2[0]), "+v"(b2[1]), "+v"(b2[2]), "+v"(b2[3]));
            });
            if constexpr (i + DB < WMT) tloadB(std::integral_constant<int, i + DB>{});
        });
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            float z = zc[jb] - fnir, bb = b2[jb];
            z += __shfl_xor(z, 16, 64); z += __shfl_xor(z, 32, 64);
            bb += __shfl_xor(bb, 16, 64); bb += __shfl_xor(bb, 32, 64);
            if (g == 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(gd_u32x2, f32x2{z, bb * 0.6931471805599453f}), rp2, (col0 + jb) * 8, 0, 0);
        }
        if (!more) break;
    }
}
