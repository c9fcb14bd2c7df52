// Row-panel-STATIONARY cost-volume forward (round 5) — an EXPERIMENT behind GD_CV_PANEL (1: NW = 4, 8: NW = 8; default 0 = the persistent kernels of
// cost_volume.hip).  Correct and bit-reproducible (tests/test_gpu_cost_volume.py::test_panel_forward_matches_oracle_and_round4_kernel) and SLOWER than
// what it was meant to replace: dense sweep, 32 pairs, 299 us (NW = 4) / 332 us (NW = 8) against 240 us.  DESIGN.md section 5 has the anatomy; in short
// a wave that owns only 16 rows reads the whole 16 KB column stage per K step for 16 MFMAs (LDS read bandwidth = MFMA time at 8 waves per CU), and
// at one or two waves per SIMD the phases of a tile add up instead of overlapping.  Included by cost_volume.hip after CvTileParams.
//
// The idea: cv_fwd_persist_kernel re-streams BOTH operand panels of every 128 x 128 score tile from L2 into LDS (393 KB per tile, 1.5 GB per 32-pair
// launch against 614 MB of HBM bytes) through the vector-memory pipe that also carries the teacher rows.  Here a block owns a 16 NW-row panel of view 1
// for a whole sweep of column tiles and keeps it ON CHIP — in REGISTERS, as the MFMA A fragments of the wave that owns the rows:
//   * wave w owns rows 16 w .. 16 w + 15 of the panel for the full K (768 halves = 24 fragments: 23 in registers, the last parked in LDS) and computes
//     the whole 16 x 128 strip of the tile: 8 n-blocks x 4 accumulator registers;
//   * only the view-2 (column) rows stream: a ring of NS stages of 128 rows x 128 B (16 KB), NS - 1 stages in flight ACROSS tile boundaries, 16 / NW
//     LDS-DMA pieces per wave and K step — with NW = 8 half the L2 -> LDS bytes per score, and no A-side LDS reads at all;
//   * a tile's row statistics are complete inside one wave (no cross-wave combine for direction 1); direction 2's column partials of the NW waves are
//     summed in fixed order through LDS by the next tile's first step (bit-reproducible, like the kernels this was meant to replace);
//   * teacher entries go straight into the accumulator layout one tile ahead — by INLINE-ASM loads with hand-counted `s_waitcnt vmcnt`: the wave's
//     vector-memory queue also carries its LDS-DMA pieces, and a load the compiler can see is waited for with vmcnt(0) at its first use
//     (cdna_hip_programming.md, "Pipelining across barriers"), which would drain the ring once per tile;
//   * every LDS access is inline asm for the same reason (a ds_read the compiler can see gets a vmcnt wait to the latest LDS-DMA in front of it).
// The vmcnt counts are STATIC (cva_wait_vm_c: compile-time arguments after unrolling): per tile and wave the program order is, for K step kt = 0 .. 11,
// wait(stage (tile, kt)) | barrier | the pieces of the stage NS - 1 steps ahead [+ one statistics piece with the tile's last stage], then the epilogue
// (one store), [panel reload: drained], the 16 teacher loads of the next tile — so the pieces of stage (tile, kt) have (NS - 2) stages of pieces younger
// than them, plus the teacher burst iff they were issued before it; statistics pieces and stores are left out (under-counting waits a little longer).
// Work split: the global tile list is pair-major with the column tile fastest; an XCD takes a contiguous range, and inside every pair's segment of that
// range the XCD's blocks take CONTIGUOUS sub-slices — all blocks of an XCD work on the same pair (its view-2 rows stay in that L2) and a block changes
// its row panel (one reload of the A registers: the OUTER loop of the kernel) a few times per slice.  Dense form only (the kept-row form stays on
// cv_fwd_rows_kernel: with ~1 tile per slice and direction a panel would be reloaded for every tile).
#pragma once

// LDS layout of a block of NW waves (a panel of 16 NW rows), NS ring slots of 16 KB:
//   ring | column statistics: 2 buffers (tile parity) x 128 float4 | column partials [NW waves][128 columns][Z, B] | row statistics of the panel [NW][16][2] |
//   the panel's LAST A fragment, parked [NW][64 lanes][16 B] | the block's tile list
#define CVA_STAGE 16384
#define CVA_LIST_MAX 700
template <int NW> struct CvaL {
    static constexpr int NS = NW == 8 ? 8 : 4;                       // ring slots (NS - 1 stages in flight)
    static constexpr int CST = NS * CVA_STAGE;
    static constexpr int CP = CST + 2 * 2048;
    static constexpr int RST = CP + (NW == 8 ? 2 : 1) * NW * 1024;   // (8 waves: two buffers as before; 4 waves x 2 blocks per CU: one — the flush of tile t precedes the epilogue of t + 1 by a K loop of barriers)
    static constexpr int APK = RST + NW * 128;
    static constexpr int LIST = APK + NW * 1024;
    static constexpr int SMEM = LIST + (CVA_LIST_MAX + 8) * 4;
};
static_assert(CvaL<8>::SMEM <= 163840 && 2 * CvaL<4>::SMEM <= 163840, "LDS budget (one 8-wave block or two 4-wave blocks per CU)");

// LDS row rho = 16 j + c of a stage holds tile column 8 c + j: after the MFMAs lane c's eight n-blocks are EIGHT CONSECUTIVE columns
__device__ __forceinline__ int cva_perm128(int rho) { return ((rho & 15) << 3) | (rho >> 4); }

#define CVA_GLD128(dst, ptr, off) asm volatile("global_load_dwordx4 %0, %1, off offset:" #off : "=v"(dst) : "v"(ptr) : "memory")
// the same from a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset: half the address registers of the 64-bit form
#define CVA_GLD128_S(dst, voff, sbase, off) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #off : "=v"(dst) : "v"(voff), "s"(sbase) : "memory")
#define CVA_DSW128(addr, val, off) asm volatile("ds_write_b128 %0, %1 offset:" #off : : "v"(addr), "v"(val) : "memory")
#define CVA_DSW64(addr, val, off) asm volatile("ds_write_b64 %0, %1 offset:" #off : : "v"(addr), "v"(val) : "memory")

__device__ __forceinline__ unsigned cva_lds_u32(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// four floats 1 KB apart, summed in order
__device__ __forceinline__ float cva_sum4(unsigned addr) {
    float v0, v1, v2, v3;
    GD_DSR32(v0, addr, 0); GD_DSR32(v1, addr, 1024); GD_DSR32(v2, addr, 2048); GD_DSR32(v3, addr, 3072);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    return ((v0 + v1) + v2) + v3;
}
// eight floats 1 KB apart (the eight waves' partials of one column), summed in wave order: fixed order, bit-reproducible
__device__ __forceinline__ float cva_sum8(unsigned addr) {
    float v0, v1, v2, v3;      // (four at a time: the K loop this sits in has no eight registers to spare)
    GD_DSR32(v0, addr, 0); GD_DSR32(v1, addr, 1024); GD_DSR32(v2, addr, 2048); GD_DSR32(v3, addr, 3072);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    float s = ((v0 + v1) + v2) + v3;
    asm volatile("" : "+v"(s));
    GD_DSR32(v0, addr, 4096); GD_DSR32(v1, addr, 5120); GD_DSR32(v2, addr, 6144); GD_DSR32(v3, addr, 7168);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    return (((s + v0) + v1) + v2) + v3;
}
// sum over the four lanes {l, l ^ 16, l ^ 32, l ^ 48} (the four row groups g of one column c), result in all four: VALU row swaps, no LDS
__device__ __forceinline__ float cva_quad_rows_sum(float v) {
    const unsigned u = gd_f2u(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned a0 = a[0], a1 = a[1];
    v = __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1);
    const unsigned w = gd_f2u(v);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    const unsigned b0 = b[0], b1 = b[1];
    return __builtin_bit_cast(float, b0) + __builtin_bit_cast(float, b1);
}

// a[i] <- 16 bytes at ap + 64 i, i = 0 .. N - 1, as inline-asm loads (the caller waits: the compiler does not know they are in flight)
template <int I, int N, typename Frag>
__device__ __forceinline__ void cva_load_frags(Frag (&a)[N], const char* ap) {
    if constexpr (I < N) {
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(a[I]) : "v"(ap), "n"(I * 64) : "memory");
        cva_load_frags<I + 1, N>(a, ap);
    }
}

// the same value behind an optimisation barrier: what is derived from the result cannot be hoisted out of the tile loop and kept in registers across it
// (lane-derived offsets of the epilogue / prefetch / DMA set-up: ~25 VGPRs of loop invariants that the K loop has no room for)
__device__ __forceinline__ int cva_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// this lane's index, rebuilt where it is needed (v_mbcnt pair) instead of held in a register across the K loop
__device__ __forceinline__ int cva_lane() { return cva_opaque((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))); }

// s_waitcnt vmcnt(n) for a COMPILE-TIME n (the unrolled K loop folds the chain): n is the number of this wave's vector-memory instructions younger than
// the one waited for (VMEM retires in issue order); smaller is safe (waits for a little more), larger is not
__device__ __forceinline__ void cva_wait_vm_c(int n) {
#define CVA_W(k) else if (n >= k) asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory")
    if (n >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    CVA_W(44); CVA_W(40); CVA_W(36); CVA_W(32); CVA_W(28); CVA_W(26); CVA_W(24); CVA_W(20); CVA_W(18); CVA_W(16); CVA_W(12); CVA_W(10); CVA_W(8); CVA_W(6); CVA_W(4); CVA_W(2);
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef CVA_W
}

// DENSE form (both directions from one sweep).  NK = 12 K-steps of 128 bytes (C = 768 halves); the ring runs LA = NS - 1 stages ahead, PP = 16 / NW pieces
// per wave and stage.  Static vmcnt counts (header comment): when stage (tile, kt) is waited for, the LA - 1 stages issued after it are younger
// (PP (LA - 1) pieces) plus the 16-load teacher burst iff the stage was issued before it (kt < LA); the last tile of a block issues nothing past the end of
// its list and counts what is left.  NW = 8: vmcnt(28) / vmcnt(12); NW = 4: vmcnt(24) / vmcnt(8).
// NW = 8: one block per CU, 128-row panels.  NW = 4: 64-row panels, TWO independent blocks per CU — meant to let one block's epilogue run under the
// other's MFMAs (the 8-wave block's phases simply added up); measured 299 us against 332, both behind the 240 us of the round-4 kernel.
template <typename T, int NK, int NW, int DBG = 0>      // DBG: anatomy instantiations (-DGD_CV_PANEL_ANAT builds only; bits: 1 no teacher loads, 2 no epilogue math, 4 no LDS reads / MFMAs, 8 no DMA); never the product path
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void cv_fwd_panel_kernel(CvTileParams q) {
    typedef typename Mma<T>::Frag Frag;
    typedef CvaL<NW> L;
    static_assert(sizeof(T) == 2 && NK == 12 && (NW == 4 || NW == 8), "16-bit features with 1536-byte rows");
    constexpr int NS = L::NS, LA = NS - 1;   // ring slots; stages the ring runs ahead
    constexpr int PP = 16 / NW;              // 1-KB DMA pieces per wave and stage
    constexpr int CPW = 128 / NW;            // columns per wave in the statistics DMA / partial-sum flush
    constexpr int RP = 16 * NW;              // rows of a panel
    constexpr int NTL = 16;                  // teacher loads per tile and wave
    __shared__ __attribute__((aligned(16))) char smem[L::SMEM];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = q.hw, TL = q.tiles, ldt = q.ldt;
    constexpr long rowb = (long)NK * 128;
    const int TM = (hw + RP - 1) / RP;                   // row panels per view
    const int per = TM * TL;                             // tiles that share one column operand (a pair)
    const unsigned lds0 = lds_off(smem);
    constexpr int dbg = DBG;      // compile-time: a run-time switch word cost registers the kernel does not have (the first anatomy build spilled 520 bytes)

    // ---- this block's tile list: pair-major, column tile fastest; an XCD takes a contiguous range, and inside every pair's segment of it the
    //      XCD's blocks take contiguous sub-slices ----
    {
        int* list = (int*)(smem + L::LIST);
        if (tid == 0) {
            const int total = q.P * per, nbx = gridDim.x >> 3, xc = blockIdx.x & 7, kb = blockIdx.x >> 3;
            const int qT = total >> 3, rT = total & 7;
            const int beg = xc < rT ? xc * (qT + 1) : rT * (qT + 1) + (xc - rT) * qT;
            const int end = beg + qT + (xc < rT ? 1 : 0);
            int n = 0;
            for (int sp = beg / per; sp * per < end; ++sp) {
                const int sb = max(beg, sp * per), se = min(end, (sp + 1) * per), len = se - sb;
                const int a = sb + (int)((long)len * kb / nbx), b = sb + (int)((long)len * (kb + 1) / nbx);
                for (int l = a; l < b && n < CVA_LIST_MAX; ++l) list[n++] = l;
            }
            list[CVA_LIST_MAX] = n;
        }
        __syncthreads();
    }
    const int n_tiles = __builtin_amdgcn_readfirstlane((int)cva_lds_u32(lds0 + L::LIST + CVA_LIST_MAX * 4));
    if (n_tiles == 0) return;
    auto tile_at = [&](int it) { return __builtin_amdgcn_readfirstlane((int)cva_lds_u32(lds0 + L::LIST + it * 4)); };

    // ---- column-operand DMA stream: PP 1-KB pieces per wave and stage ----
    const char* bbase = nullptr;                        // column operand of the tile whose stages are being issued (wave-uniform) ...
    unsigned boff[PP];                                  // ... + this lane's byte offset of its pieces at K-step 0 (an operand spans < 2^31 bytes: host check)
    auto set_bsrc = [&](int l) {
        const int sp = l / per, tn = (l - sp * per) % TL;
        bbase = (const char*)q.f2 + (long)sp * hw * rowb;
        const int ln = cva_lane();
#pragma unroll
        for (int i = 0; i < PP; ++i) {
            const int rho = (PP * wave + i) * 8 + (ln >> 3);
            boff[i] = (unsigned)(min(tn * 128 + cva_perm128(rho), hw - 1) * (int)rowb + (((ln & 7) ^ swz(rho)) * 16));
        }
    };
    auto issue_colstats = [&](int it) {      // statistics rows of the tile's 128 columns: 16 B each, wave w moves columns CPW w .. CPW w + CPW - 1
        const int l = tile_at(it);
        const int sp = l / per, tn = (l - sp * per) % TL;
        const int ln = cva_lane();
        const int col = min(tn * 128 + CPW * wave + (ln & (CPW - 1)), hw - 1);
        const long srow = ((long)sp * 2 + 1) * hw + col;
        if (ln < CPW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(q.stats + srow * 4),
                                             (__attribute__((address_space(3))) void*)(smem + L::CST + (it & 1) * 2048 + wave * (CPW * 16)), 16, 0, 0);
    };
    auto issue_stage = [&](int slot, int kp) {      // kp: K-step of the stage inside its tile (a constant after unrolling: folds into the pieces' offsets)
        if (dbg & 8) return;
        char* dst = smem + slot * CVA_STAGE + PP * wave * 1024;
#pragma unroll
        for (int i = 0; i < PP; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bbase + kp * 128 + boff[i]),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };

    // ---- the row panel: this wave's 16 rows as MFMA A fragments for the whole K, their statistics parked in LDS ----
    // Inline-asm loads behind an explicit drain: called between a tile's epilogue and the next tile's teacher prefetch, so the drain only waits for
    // ring stages (L2 hits), never for teacher rows.
    Frag a[2 * NK - 1];      // K chunks 0 .. 22 (chunk 23 is parked in LDS: the K loop holds 92 + 32 + 64 + 16 registers of operands as it is)
    auto load_panel = [&](int l) {
        const int sp = l / per, tm = (l - sp * per) / TL;
        const int ln = cva_lane(), g = ln >> 4, c = ln & 15;
        const int arow = min(tm * RP + 16 * wave + c, hw - 1);
        const char* ap = (const char*)q.f1 + (long)sp * hw * rowb + (long)arow * rowb + 16 * g;
        f32x4 st[4], alast;
        cva_load_frags<0, 2 * NK - 1>(a, ap);
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(alast) : "v"(ap), "n"((2 * NK - 1) * 64) : "memory");
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* sp_ = q.stats + (((long)sp * 2) * hw + min(tm * RP + 16 * wave + 4 * g + r, hw - 1)) * 4;
            CVA_GLD128(st[r], sp_, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]),
                     "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]), "+v"(a[17]), "+v"(a[18]), "+v"(a[19]),
                     "+v"(a[20]), "+v"(a[21]), "+v"(a[22]), "+v"(alast), "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]) : : "memory");
        {
            const unsigned pk = lds0 + L::APK + wave * 1024 + ln * 16;
            CVA_DSW128(pk, alast, 0);
        }
        // the rows' statistics are parked in LDS (this wave's own 128 bytes: written and read by the same wave, no barrier): eight registers
        // that the main loop does not have
        if (c == 0) {
            const unsigned ra = lds0 + L::RST + wave * 128 + (4 * g) * 8;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x2 v = {st[r][0], 1.0f / st[r][1]};
                if (r == 0) CVA_DSW64(ra, v, 0); else if (r == 1) CVA_DSW64(ra, v, 8); else if (r == 2) CVA_DSW64(ra, v, 16); else CVA_DSW64(ra, v, 24);
            }
        }
    };

    // ---- teacher tile, one tile ahead, in the accumulator layout ----
    // direction 1: t1v[r][h] = T1[row 16 w + 4 g + r][columns 8 c + 4 h .. + 3]   (element k <-> n-block j = 4 h + k)
    // direction 2: t2v[j]    = T2[row = column 8 c + j][columns = rows 16 w + 4 g .. + 3]   (element r)
    f32x4 t1v[4][2], t2v[8];
    auto prefetch = [&](int l) {
        if (dbg & 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) t1v[r][0] = t1v[r][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) t2v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            return;
        }
        l = __builtin_amdgcn_readfirstlane(l);             // (wave-uniform by construction; the loads below take their base in SGPRs)
        const int sp = l / per, rr = l - sp * per, tm = rr / TL, tn = rr - tm * TL;
        const int ln = cva_lane(), g = ln >> 4, c = ln & 15;
        const int col0 = min(tn * 128 + 8 * c, ldt - 8);                     // ldt % 4 == 0 and ldt >= 8: aligned, inside the row
        // (a pair's map spans hw * ldt * 4 < 2^31 bytes: 32-bit lane offsets from a scalar base — made scalar explicitly, halves through readfirstlane)
        auto sbase = [](const float* p) {
            const unsigned long u = (unsigned long)p;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
            return ((unsigned long)hi << 32) | lo;
        };
        unsigned long T1 = sbase(q.t1 + (long)sp * hw * ldt), T2 = sbase(q.t2 + (long)sp * hw * ldt);
        // v_readfirstlane -> SGPR -> VMEM address needs 5 wait states on gfx9; the compiler's hazard recogniser does not look inside inline asm
        // (without this the first load of a burst went out with the PREVIOUS contents of the SGPR pair: a memory fault)
        asm volatile("s_nop 4" : "+s"(T1), "+s"(T2));
        const int row0 = tm * RP + 16 * wave + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned off = (unsigned)((min(row0 + r, hw - 1) * ldt + col0) * 4);
            CVA_GLD128_S(t1v[r][0], off, T1, 0);
            CVA_GLD128_S(t1v[r][1], off, T1, 16);
        }
        const int rowc = min(row0, ldt - 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned off = (unsigned)((min(tn * 128 + 8 * c + j, hw - 1) * ldt + rowc) * 4);
            CVA_GLD128_S(t2v[j], off, T2, 0);
        }
    };

    // ---- column partials of a finished tile: LDS (written by every wave before the barrier just passed) -> slab, fixed summation order ----
    auto flush_cols = [&](int lp, int itp) {
        const int spp = lp / per, rp = lp - spp * per, tmp = rp / TL, tnp = rp - tmp * TL;
        const int ln = cva_lane();
        const int which = ln / CPW, cl = CPW * wave + (ln & (CPW - 1));
        const unsigned pb = lds0 + L::CP + (NW == 8 ? (itp & 1) * 8192 : 0) + (cl * 2 + (which & 1)) * 4;
        const float s = NW == 8 ? cva_sum8(pb) : cva_sum4(pb);
        const int col = tnp * 128 + cl;
        if (which < 2 && col < hw)
            *(__attribute__((address_space(1))) float*)((uintptr_t)q.part2 + ((((long)spp * TM + tmp) * hw + col) * 2 + which) * sizeof(float)) = s;
    };

    // ---- prologue: the first panel, statistics of tile 0, the ring's first seven stages, the first teacher tile ----
    int l_cur = tile_at(0);
    issue_colstats(0);
    set_bsrc(l_cur);
#pragma unroll
    for (int k = 0; k < LA; ++k) issue_stage(k, k);

    // fragment row c of n-block 0, the lane's chunk of the first K chunk ((g ^ swz) * 16); the second chunk's is that ^ 64
    const unsigned fb = lds0 + c * 128 + ((g ^ swz(c)) * 16);
    int it = 0;
    bool done = false;
    // Two nested loops — row panels, then the tiles of a panel — so that the A fragments are DEFINED once per panel and loop-invariant in the tile
    // loop.  (As one loop with a conditional reload, the new fragments and the old ones met in 92 phi nodes: the allocator loaded the new panel into
    // a second register range, copied it over at the back edge, and spilled what was in the way.)
    for (;;) {
    const int cur_panel = l_cur / TL;
    load_panel(l_cur);
    prefetch(l_cur);
    for (;;) {
        const int l = l_cur;
        const int sp = l / per, rr = l - sp * per, tm = rr / TL, tn = rr - tm * TL;
        const bool last = it + 1 == n_tiles;
        const int l_nxt = last ? l : tile_at(it + 1);
        const int nb = (it * NK) & (NS - 1);              // ring slot of this tile's first stage
        f32x4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            // this wave's pieces of stage (it, kt) have landed (counts: see the kernel's header comment)
            if (!last) cva_wait_vm_c(kt < LA ? PP * (LA - 1) + NTL : PP * (LA - 1));
            else cva_wait_vm_c(PP * (NK - 1 - kt < LA - 1 ? NK - 1 - kt : LA - 1) + (kt < LA ? NTL : 0));
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // ... everyone's have; the slot of stage (it, kt) - 1 is free
            {
                constexpr int dummy = 0; (void)dummy;
                const int kp = (kt + LA) % NK;
                const bool nxt = kt + LA >= NK;
                if (!nxt || !last) {
                    if (nxt && kp == 0) set_bsrc(l_nxt);
                    issue_stage((nb + kt + LA) & (NS - 1), kp);
                    if (kp == NK - 1 && !last) issue_colstats(it + 1);      // the next tile's column statistics ride with the last stage of this one
                }
            }
            if (kt == 0 && it > 0) {
                // ---- previous tile's column partials: the NW waves' sums in fixed order -> slab (wave w: columns CPW w .. + CPW - 1; lanes [0, CPW) Z, [CPW, 2 CPW) B) ----
                flush_cols(tile_at(it - 1), it - 1);
            }
            if (dbg & 4) continue;
            const unsigned sb = fb + ((nb + kt) & (NS - 1)) * CVA_STAGE;
            // B fragments in two groups of four n-blocks through the SAME 16 registers (32 registers of fragments in flight spilled: a scratch
            // reload is a vector-memory load, and the compiler waits for it with vmcnt(0) — which drains the DMA ring)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const unsigned ad = kc ? (sb ^ 64u) : sb;
                Frag af;
                if (2 * kt + kc < 2 * NK - 1) af = a[2 * kt + kc];
                else {      // the parked fragment (this wave's own 1 KB: written and read by the same wave)
                    f32x4 t;
                    GD_DSR128(t, lds0 + L::APK + wave * 1024 + cva_lane() * 16, 0);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t));
                    af = __builtin_bit_cast(Frag, t);
                }
                // B fragments two n-blocks at a time through the SAME 8 registers (more in flight spilled: a scratch reload is a vector-memory
                // load, and the compiler waits for it with vmcnt(0) — which drains the DMA ring); the SIMD's other wave covers the read latency
                f32x4 b0, b1;
#define CVA_PAIR(J, O0, O1)                                                                    \
                GD_DSR128(b0, ad, O0); GD_DSR128(b1, ad, O1);                                  \
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1));                     \
                acc[J] = Mma<T>::mma(af, __builtin_bit_cast(Frag, b0), acc[J]);                \
                acc[J + 1] = Mma<T>::mma(af, __builtin_bit_cast(Frag, b1), acc[J + 1]);        \
                __builtin_amdgcn_sched_barrier(0);
                CVA_PAIR(0, 0, 2048) CVA_PAIR(2, 4096, 6144) CVA_PAIR(4, 8192, 10240) CVA_PAIR(6, 12288, 14336)
#undef CVA_PAIR
            }
        }
        // ---------------- epilogue, all from registers ----------------
        // the tile's teacher entries have landed: younger than the burst are the 24 pieces issued during the K loop (10 in a block's last tile)
        if (!(dbg & 1)) { if (!last) cva_wait_vm_c(PP * NK); else cva_wait_vm_c(PP * (NK - LA)); }
        asm volatile("" : "+v"(t1v[0][0]), "+v"(t1v[0][1]), "+v"(t1v[1][0]), "+v"(t1v[1][1]), "+v"(t1v[2][0]), "+v"(t1v[2][1]), "+v"(t1v[3][0]), "+v"(t1v[3][1]),
                     "+v"(t2v[0]), "+v"(t2v[1]), "+v"(t2v[2]), "+v"(t2v[3]), "+v"(t2v[4]), "+v"(t2v[5]), "+v"(t2v[6]), "+v"(t2v[7]));
        if (!(dbg & 2)) {
        const int ln = cva_lane(), g = ln >> 4, c = ln & 15;      // (shadows: the epilogue's lane-derived addresses are rebuilt per tile, not kept across the K loop)
        // Two halves of four columns each (n-blocks 4 h .. 4 h + 3): the column-side temporaries of a half are 16 registers, not 32 — with 96
        // A-fragment, 32 accumulator and 64 teacher registers live, the full-width form spilled.  Column statistics of this tile: landed with the
        // previous tile's last stage, visible since that step's barrier.
        float zr[4] = {0.f, 0.f, 0.f, 0.f}, b1[4] = {0.f, 0.f, 0.f, 0.f};
        const unsigned ra = lds0 + L::RST + wave * 128 + (4 * g) * 8;
        const int row0e = tm * RP + 16 * wave + 4 * g;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {      // column pairs (n-blocks 2 qd, 2 qd + 1): the column-side temporaries of a pair are 8 registers
            float inv2[2], ir2[2], zc[2] = {0.f, 0.f}, b2[2] = {0.f, 0.f};
            bool cok[2];
            {
                const unsigned cs = lds0 + L::CST + (it & 1) * 2048 + (8 * c + 2 * qd) * 16;
                f32x2 s0, s1;      // {inverse norm, teacher row sum} of the pair's columns
                asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(s0) : "v"(cs));
                asm volatile("ds_read_b64 %0, %1 offset:16" : "=v"(s1) : "v"(cs));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1));
                inv2[0] = s0[0]; inv2[1] = s1[0];
                ir2[0] = 1.0f / s0[1]; ir2[1] = 1.0f / s1[1];
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) cok[k] = tn * 128 + 8 * c + 2 * qd + k < hw;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x2 rs;      // this row's {inverse norm, 1 / teacher row sum}: re-read per use (eight registers the loop does not have)
                asm volatile("ds_read_b64 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(rs) : "v"(ra), "n"(r * 8) : "memory");
                const float inv1r = rs[0], ir1r = rs[1];
                const bool rokr = row0e + r < hw;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int j = 2 * qd + k;
                    const bool ok = rokr && cok[k];
                    const float sv = acc[j][r] * inv1r * inv2[k];
                    const float e = ok ? __expf(sv) : 0.f;
                    const float sm = ok ? sv : 0.f;
                    zr[r] += e;
                    b1[r] = fmaf(fmaxf(t1v[r][j >> 2][j & 3] * ir1r, CV_EPS), sm, b1[r]);
                    zc[k] += e;
                    b2[k] = fmaf(fmaxf(t2v[j][r] * ir2[k], CV_EPS), sm, b2[k]);
                }
                // pin the running sums per row: the compiler otherwise keeps every addend alive to re-associate them (spilled exponentials)
                asm volatile("" : "+v"(zr[r]), "+v"(b1[r]), "+v"(zc[0]), "+v"(zc[1]), "+v"(b2[0]), "+v"(b2[1]));
            }
            // column partials of this wave's 16-row strip -> LDS [wave][column 8 c + j][Z, B] (the lanes of row group 0 write 16 contiguous bytes per pair)
            const f32x4 o0 = {cva_quad_rows_sum(zc[0]), cva_quad_rows_sum(b2[0]), cva_quad_rows_sum(zc[1]), cva_quad_rows_sum(b2[1])};
            if (g == 0) {
                const unsigned pa = lds0 + L::CP + (NW == 8 ? (it & 1) * 8192 : 0) + wave * 1024 + c * 64 + qd * 16;
                CVA_DSW128(pa, o0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);      // one column pair at a time: the scheduler otherwise overlaps the pairs and their temporaries spill an A fragment
        }
        {   // row partials of the wave's 16 rows: lane (g, c < 4) holds row 4 g + c -> 16 lanes store 128 contiguous bytes
            float zrow[4], brow[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { zrow[r] = row16_sum(zr[r]); brow[r] = row16_sum(b1[r]); }
            const float zs = c == 0 ? zrow[0] : c == 1 ? zrow[1] : c == 2 ? zrow[2] : zrow[3];
            const float bs = c == 0 ? brow[0] : c == 1 ? brow[1] : c == 2 ? brow[2] : brow[3];
            const int rl = 16 * wave + 4 * g + c;
            if (c < 4 && tm * RP + rl < hw)
                *(__attribute__((address_space(1))) f32x2*)((uintptr_t)q.part1 + ((((long)sp * TL + tn) * hw + tm * RP + rl) * 2) * sizeof(float)) = f32x2{zs, bs};
        }
        } else if (acc[0][0] == 12345.678f) ((float*)smem)[tid] = acc[1][1] + t1v[0][0][0] + t2v[0][1];
        if (last) { done = true; break; }
        l_cur = l_nxt;
        ++it;
        if (l_cur / TL != cur_panel) break;      // the next tile belongs to another row panel: reload (outer loop)
        prefetch(l_cur);
    }
    if (done) break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    flush_cols(l_cur, n_tiles - 1);
}
