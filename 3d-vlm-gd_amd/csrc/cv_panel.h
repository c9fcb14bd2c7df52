// Row-panel-STATIONARY cost-volume forward (round 5).  Included by cost_volume.hip after CvTileParams / the helpers of the persistent kernels.
//
// Why: cv_fwd_persist_kernel / cv_fwd_rows_kernel re-stream BOTH operand panels of every 128 x 128 score tile from L2 into LDS (393 KB per tile,
// 1.5 GB per 32-pair launch against 614 MB of HBM bytes) through one vector-memory pipe that also carries the teacher rows; the L2 -> LDS feature
// stream alone is ~85 of the launch's ~230 us (profiles/NOTES_r01_r03.md).  Here a block owns a 128-row panel of view 1 for a
// whole sweep of column tiles and keeps it ON CHIP — in REGISTERS, as the MFMA A fragments of the wave that owns the rows:
//   * 8 waves, wave w owns rows 16 w .. 16 w + 15 of the panel for the full K (768 halves = 24 fragments = 96 VGPRs) and computes the whole
//     16 x 128 strip of the tile: 8 n-blocks x 4 accumulator registers;
//   * only the view-2 (column) rows stream: a ring of CVA_NS = 8 stages of 128 rows x 128 B (16 KB), seven stages in flight ACROSS tile
//     boundaries (112 KB per CU), two LDS-DMA pieces per wave and K-step — half the L2 -> LDS bytes per score, and no A-side LDS reads at all;
//   * a tile's row statistics are complete inside one wave (no cross-wave combine for direction 1); direction 2's column partials of the eight
//     waves are summed in fixed order through LDS by the next tile's first step (bit-reproducible, like the kernels this replaces);
//   * teacher entries go straight into the accumulator layout one tile ahead, as before — by INLINE-ASM loads with hand-counted `s_waitcnt vmcnt`:
//     the wave's vector-memory queue now also carries its LDS-DMA pieces, and a load the compiler can see is waited for with vmcnt(0) at its first
//     use (cdna_hip_programming.md, "Pipelining across barriers"), which would drain the ring once per tile;
//   * every LDS access is inline asm for the same reason (a ds_read the compiler can see gets a vmcnt wait to the latest LDS-DMA in front of it).
// The vmcnt bookkeeping is DYNAMIC: `seq` counts the wave's unconditional vector-memory instructions (DMA pieces, statistics pieces, teacher loads),
// an 8 x 8-bit packed scalar remembers `seq` after each ring slot's pieces, and a stage is waited for with vmcnt(seq - mark) rounded DOWN to a
// multiple of 4.  Instructions that a wave may skip (stores under a row mask) are NOT counted: under-counting only waits a little longer.
// Work split: the global tile list is pair-major with the column tile fastest; an XCD takes a contiguous range (as before), and inside every
// pair's segment of that range the XCD's blocks take CONTIGUOUS sub-slices — all blocks of an XCD work on the same pair (its view-2 rows stay in
// that L2) and a block changes its row panel (one reload of the A registers) at most twice per slice.
// ROWS = the kept-row form (sparse row masks, cv_fwd_rows_kernel's problem): tile (pd = 2 pair + direction, row tile of the compacted kept rows,
// column tile), A rows gathered through the index list, one direction's epilogue, row partials only.
#pragma once

#define CVA_NS 8
#define CVA_STAGE 16384
#define CVA_CST_OFF (CVA_NS * CVA_STAGE)              // column statistics: 4 buffers x 128 float4
#define CVA_CP_OFF (CVA_CST_OFF + 4 * 2048)           // column partials: 2 buffers x [8 waves][128 columns][Z, B]
#define CVA_RST_OFF (CVA_CP_OFF + 2 * 8192)           // row statistics of the resident panel: [8 waves][16 rows][inv norm, 1 / teacher row sum]
#define CVA_LIST_OFF (CVA_RST_OFF + 1024)             // this block's tile list
#define CVA_LIST_MAX 760
#define CVA_SMEM (CVA_LIST_OFF + (CVA_LIST_MAX + 8) * 4)
static_assert(CVA_SMEM <= 163840, "LDS budget");

// LDS row rho = 16 j + c of a stage holds tile column 8 c + j: after the MFMAs lane c's eight n-blocks are EIGHT CONSECUTIVE columns
__device__ __forceinline__ int cva_perm128(int rho) { return ((rho & 15) << 3) | (rho >> 4); }

#define CVA_GLD128(dst, ptr, off) asm volatile("global_load_dwordx4 %0, %1, off offset:" #off : "=v"(dst) : "v"(ptr) : "memory")
#define CVA_DSW128(addr, val, off) asm volatile("ds_write_b128 %0, %1 offset:" #off : : "v"(addr), "v"(val) : "memory")
#define CVA_DSW64(addr, val, off) asm volatile("ds_write_b64 %0, %1 offset:" #off : : "v"(addr), "v"(val) : "memory")

// wait until at most n (rounded down to a multiple of 4, at most 60) vector-memory operations of this wave are outstanding
__device__ __forceinline__ void cva_wait_vm(int n) {
#define CVA_W(k) asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory")
    if (n >= 32) {
        if (n >= 48) { if (n >= 56) { if (n >= 60) CVA_W(60); else CVA_W(56); } else { if (n >= 52) CVA_W(52); else CVA_W(48); } }
        else { if (n >= 40) { if (n >= 44) CVA_W(44); else CVA_W(40); } else { if (n >= 36) CVA_W(36); else CVA_W(32); } }
    } else {
        if (n >= 16) { if (n >= 24) { if (n >= 28) CVA_W(28); else CVA_W(24); } else { if (n >= 20) CVA_W(20); else CVA_W(16); } }
        else { if (n >= 8) { if (n >= 12) CVA_W(12); else CVA_W(8); } else { if (n >= 4) CVA_W(4); else CVA_W(0); } }
    }
#undef CVA_W
}
__device__ __forceinline__ unsigned cva_lds_u32(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// eight floats 1 KB apart (the eight waves' partials of one column), summed in wave order: fixed order, bit-reproducible
__device__ __forceinline__ float cva_sum8(unsigned addr) {
    float v0, v1, v2, v3, v4, v5, v6, v7;
    GD_DSR32(v0, addr, 0); GD_DSR32(v1, addr, 1024); GD_DSR32(v2, addr, 2048); GD_DSR32(v3, addr, 3072);
    GD_DSR32(v4, addr, 4096); GD_DSR32(v5, addr, 5120); GD_DSR32(v6, addr, 6144); GD_DSR32(v7, addr, 7168);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
    return ((((((v0 + v1) + v2) + v3) + v4) + v5) + v6) + v7;
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

template <typename T, int NK, bool ROWS, bool DBG = false>      // DBG: anatomy builds (GD_CV_DBG bits: 1 no teacher loads, 2 no epilogue math, 4 no LDS reads / MFMAs, 8 no DMA); never the product path
__global__ __launch_bounds__(512) void cv_fwd_panel_kernel(CvTileParams q) {
    typedef typename Mma<T>::Frag Frag;
    static_assert(sizeof(T) == 2, "16-bit features");
    __shared__ __attribute__((aligned(16))) char smem[CVA_SMEM];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = q.hw, TL = q.tiles, ldt = q.ldt;
    constexpr long rowb = (long)NK * 128;
    const int tiles_r = ROWS ? (q.kcap >> 7) : TL, kcap = q.kcap;
    const int per = tiles_r * TL;                        // tiles that share one column operand (a pair / a (pair, direction))
    const unsigned lds0 = lds_off(smem);
    const int dbg = DBG ? q.dbg : 0;

    // ---- this block's tile list ----
    {
        int* list = (int*)(smem + CVA_LIST_OFF);
        if (tid == 0) {
            const int total = (ROWS ? 2 * q.P : q.P) * per, nbx = gridDim.x >> 3, xc = blockIdx.x & 7, kb = blockIdx.x >> 3;
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
    const int n_tiles = __builtin_amdgcn_readfirstlane((int)cva_lds_u32(lds0 + CVA_LIST_OFF + CVA_LIST_MAX * 4));
    if (n_tiles == 0) return;
    const int n_total = n_tiles * NK;
    auto tile_at = [&](int it) { return __builtin_amdgcn_readfirstlane((int)cva_lds_u32(lds0 + CVA_LIST_OFF + it * 4)); };

    // ---- vector-memory bookkeeping ----
    unsigned seq = 0;                      // unconditional VMEM instructions issued so far (mod 256)
    unsigned long long marks = 0;          // 8 x 8 bits: seq after the pieces of ring slot s
    unsigned markT = 0;                    // seq after the teacher loads of the tile in flight
    auto mark_set = [&](int slot) {
        const int sh = slot * 8;
        marks = (marks & ~(0xffull << sh)) | ((unsigned long long)(seq & 0xffu) << sh);
    };
    auto mark_age = [&](int slot) { return (int)((seq - (unsigned)((marks >> (slot * 8)) & 0xffu)) & 0xffu); };

    // ---- column-operand DMA stream ----
    const char* bsrc[2] = {nullptr, nullptr};
    int it_i = 0, k_i = 0, n_issue = 0;
    auto issue_colstats = [&](int it) {      // statistics rows of the tile's 128 columns: 16 B each, wave w moves columns 16 w .. 16 w + 15
        const int l = tile_at(it);
        const int sp = l / per, tn = (l - sp * per) % TL;
        const int col = min(tn * 128 + 16 * wave + (lane & 15), hw - 1);
        const long srow = ROWS ? ((long)(sp >> 1) * 2 + (1 - (sp & 1))) * hw + col : ((long)sp * 2 + 1) * hw + col;
        if (lane < 16)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(q.stats + srow * 4),
                                             (__attribute__((address_space(3))) void*)(smem + CVA_CST_OFF + (it & 3) * 2048 + wave * 256), 16, 0, 0);
        seq += 1;
    };
    auto issue_stage = [&]() {
        if (k_i == 0) {
            const int l = tile_at(it_i);
            const int sp = l / per, tn = (l - sp * per) % TL;
            const char* Wb = ROWS ? (const char*)((sp & 1) ? q.f1 : q.f2) + (long)(sp >> 1) * hw * rowb : (const char*)q.f2 + (long)sp * hw * rowb;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rho = (2 * wave + i) * 8 + (lane >> 3);
                bsrc[i] = Wb + (long)min(tn * 128 + cva_perm128(rho), hw - 1) * rowb + (((lane & 7) ^ swz(rho)) * 16);
            }
        }
        char* dst = smem + (n_issue & (CVA_NS - 1)) * CVA_STAGE + 2 * wave * 1024;
        if (!(dbg & 8))
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + (long)k_i * 128),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        seq += 2;
        mark_set(n_issue & (CVA_NS - 1));
        ++n_issue;
        if (++k_i == NK) {
            k_i = 0;
            ++it_i;
            if (it_i < n_tiles) issue_colstats(it_i);      // the next tile's column statistics ride with the last stage of this one
        }
    };

    // ---- the row panel: this wave's 16 rows as MFMA A fragments for the whole K, their statistics, (ROWS) their indices ----
    // Inline-asm loads behind an explicit drain: called between a tile's epilogue and the next tile's teacher prefetch, so the drain only waits for
    // ring stages (L2 hits), never for teacher rows.
    Frag a[2 * NK];
    int ixr[4] = {0, 0, 0, 0};
    auto load_panel = [&](int l) {
        const int sp = l / per, tm = (l - sp * per) / TL;
        int arow;
        const char* Ab;
        long srow[4];
        if constexpr (ROWS) {
            const int* ix = q.idx + (long)sp * kcap + tm * 128 + 16 * wave;
            arow = ix[c];
#pragma unroll
            for (int r = 0; r < 4; ++r) { ixr[r] = ix[4 * g + r]; srow[r] = ((long)(sp >> 1) * 2 + (sp & 1)) * hw + ixr[r]; }
            Ab = (const char*)((sp & 1) ? q.f2 : q.f1) + (long)(sp >> 1) * hw * rowb;
        } else {
            arow = min(tm * 128 + 16 * wave + c, hw - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) srow[r] = ((long)sp * 2) * hw + min(tm * 128 + 16 * wave + 4 * g + r, hw - 1);
            Ab = (const char*)q.f1 + (long)sp * hw * rowb;
        }
        const char* ap = Ab + (long)arow * rowb + 16 * g;
        f32x4 st[4];
        cva_load_frags<0, 2 * NK>(a, ap);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float* sp_ = q.stats + srow[r] * 4; CVA_GLD128(st[r], sp_, 0); }
        if constexpr (NK == 12)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]),
                         "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]), "+v"(a[17]), "+v"(a[18]), "+v"(a[19]),
                         "+v"(a[20]), "+v"(a[21]), "+v"(a[22]), "+v"(a[23]), "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]) : : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]),
                         "+v"(a[10]), "+v"(a[11]), "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]) : : "memory");
        // the rows' statistics are parked in LDS (this wave's own 128 bytes: written and read by the same wave, no barrier): eight registers
        // that the main loop does not have
        if (c == 0) {
            const unsigned ra = lds0 + CVA_RST_OFF + wave * 128 + (4 * g) * 8;
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
    f32x4 t1v[4][2], t2v[ROWS ? 1 : 8];
    auto prefetch = [&](int it) {
        if (dbg & 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) t1v[r][0] = t1v[r][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < (ROWS ? 1 : 8); ++j) t2v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            markT = seq;
            return;
        }
        const int l = tile_at(it);
        const int sp = l / per, rr = l - sp * per, tm = rr / TL, tn = rr - tm * TL;
        const int col0 = min(tn * 128 + 8 * c, ldt - 8);                     // ldt % 4 == 0 and ldt >= 8: aligned, inside the row
        if constexpr (ROWS) {
            const float* Tt = ((sp & 1) ? q.t2 : q.t1) + (long)(sp >> 1) * hw * ldt;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* ptr = Tt + (long)ixr[r] * ldt + col0;       // (the panel's kept-row indices: registers, load_panel)
                CVA_GLD128(t1v[r][0], ptr, 0);
                CVA_GLD128(t1v[r][1], ptr, 16);
            }
            seq += 8;
        } else {
            const float* T1 = q.t1 + (long)sp * hw * ldt;
            const float* T2 = q.t2 + (long)sp * hw * ldt;
            const int row0 = tm * 128 + 16 * wave + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* ptr = T1 + (long)min(row0 + r, hw - 1) * ldt + col0;
                CVA_GLD128(t1v[r][0], ptr, 0);
                CVA_GLD128(t1v[r][1], ptr, 16);
            }
            const int rowc = min(row0, ldt - 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float* ptr = T2 + (long)min(tn * 128 + 8 * c + j, hw - 1) * ldt + rowc;
                CVA_GLD128(t2v[j], ptr, 0);
            }
            seq += 16;
        }
        markT = seq;
    };

    // ---- prologue: statistics of tile 0, the ring's first seven stages, the first teacher tile ----
    int cur_panel = tile_at(0) / TL;
    load_panel(tile_at(0));
    issue_colstats(0);
    for (int n = 0; n < CVA_NS - 1 && n < n_total; ++n) issue_stage();
    prefetch(0);

    const unsigned fa = lds0 + c * 128;                   // fragment row c of n-block 0
    const int sa = swz(c);
    const unsigned co0 = ((0 + g) ^ sa) * 16, co1 = ((4 + g) ^ sa) * 16;
    int n = 0;
    for (int it = 0; it < n_tiles; ++it) {
        const int l = tile_at(it);
        const int sp = l / per, rr = l - sp * per, tm = rr / TL, tn = rr - tm * TL;
        f32x4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NK; ++kt, ++n) {
            const int slot = n & (CVA_NS - 1);
            cva_wait_vm(mark_age(slot));                  // this wave's pieces of stage n have landed
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // ... everyone's have; slot (n - 1) & 7 is free
            if (n_issue < n_total) issue_stage();
            if (!ROWS && kt == 0 && it > 0) {
                // ---- previous tile's column partials: eight waves' sums in fixed order -> slab (columns 16 w .. 16 w + 15, lanes 0-15 Z, 16-31 B) ----
                const int lp = tile_at(it - 1);
                const int spp = lp / per, rp = lp - spp * per, tmp = rp / TL, tnp = rp - tmp * TL;
                const unsigned pb = lds0 + CVA_CP_OFF + ((it - 1) & 1) * 8192 + ((16 * wave + (lane & 15)) * 2 + ((lane >> 4) & 1)) * 4;
                const float s = cva_sum8(pb);
                const int col = tnp * 128 + 16 * wave + (lane & 15);
                if (lane < 32 && col < hw)
                    *(__attribute__((address_space(1))) float*)((uintptr_t)q.part2 + ((((long)spp * q.nslab + tmp) * hw + col) * 2 + (lane >> 4)) * sizeof(float)) = s;
            }
            const unsigned sb = fa + slot * CVA_STAGE;
            if (dbg & 4) continue;
            // B fragments in two groups of four n-blocks through the SAME 16 registers (32 registers of fragments in flight spilled: a scratch
            // reload is a vector-memory load, and the compiler waits for it with vmcnt(0) — which drains the DMA ring)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const unsigned ad = sb + (kc ? co1 : co0);
                const Frag af = a[2 * kt + kc];
                f32x4 b0, b1, b2, b3;
                GD_DSR128(b0, ad, 0); GD_DSR128(b1, ad, 2048); GD_DSR128(b2, ad, 4096); GD_DSR128(b3, ad, 6144);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
                acc[0] = Mma<T>::mma(af, __builtin_bit_cast(Frag, b0), acc[0]);
                acc[1] = Mma<T>::mma(af, __builtin_bit_cast(Frag, b1), acc[1]);
                acc[2] = Mma<T>::mma(af, __builtin_bit_cast(Frag, b2), acc[2]);
                acc[3] = Mma<T>::mma(af, __builtin_bit_cast(Frag, b3), acc[3]);
                __builtin_amdgcn_sched_barrier(0);
                GD_DSR128(b0, ad, 8192); GD_DSR128(b1, ad, 10240); GD_DSR128(b2, ad, 12288); GD_DSR128(b3, ad, 14336);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
                acc[4] = Mma<T>::mma(af, __builtin_bit_cast(Frag, b0), acc[4]);
                acc[5] = Mma<T>::mma(af, __builtin_bit_cast(Frag, b1), acc[5]);
                acc[6] = Mma<T>::mma(af, __builtin_bit_cast(Frag, b2), acc[6]);
                acc[7] = Mma<T>::mma(af, __builtin_bit_cast(Frag, b3), acc[7]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (dbg & 2) {
            if (acc[0][0] == 12345.678f) ((float*)smem)[tid] = acc[1][1] + t1v[0][0][0] + t2v[0][1];
            if (it + 1 < n_tiles) {
                const int ln = tile_at(it + 1);
                if (ln / TL != cur_panel) { cur_panel = ln / TL; load_panel(ln); }
                prefetch(it + 1);
            }
            continue;
        }
        // ---------------- epilogue, all from registers ----------------
        // the tile's teacher entries have landed
        cva_wait_vm((int)((seq - markT) & 0xffu));
        if constexpr (ROWS)
            asm volatile("" : "+v"(t1v[0][0]), "+v"(t1v[0][1]), "+v"(t1v[1][0]), "+v"(t1v[1][1]), "+v"(t1v[2][0]), "+v"(t1v[2][1]), "+v"(t1v[3][0]), "+v"(t1v[3][1]));
        else
            asm volatile("" : "+v"(t1v[0][0]), "+v"(t1v[0][1]), "+v"(t1v[1][0]), "+v"(t1v[1][1]), "+v"(t1v[2][0]), "+v"(t1v[2][1]), "+v"(t1v[3][0]), "+v"(t1v[3][1]),
                         "+v"(t2v[0]), "+v"(t2v[ROWS ? 0 : 1]), "+v"(t2v[ROWS ? 0 : 2]), "+v"(t2v[ROWS ? 0 : 3]), "+v"(t2v[ROWS ? 0 : 4]), "+v"(t2v[ROWS ? 0 : 5]),
                         "+v"(t2v[ROWS ? 0 : 6]), "+v"(t2v[ROWS ? 0 : 7]));
        // Two halves of four columns each (n-blocks 4 h .. 4 h + 3): the column-side temporaries of a half are 16 registers, not 32 — with 96
        // A-fragment, 32 accumulator and 64 teacher registers live, the full-width form spilled.  Column statistics of this tile: landed with the
        // previous tile's last stage, visible since that step's barrier.
        float zr[4] = {0.f, 0.f, 0.f, 0.f}, b1[4] = {0.f, 0.f, 0.f, 0.f};
        float inv1[4], ir1[4];
        {
            const unsigned ra = lds0 + CVA_RST_OFF + wave * 128 + (4 * g) * 8;
            f32x4 r01, r23;
            GD_DSR128(r01, ra, 0); GD_DSR128(r23, ra, 16);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r01), "+v"(r23));
            inv1[0] = r01[0]; ir1[0] = r01[1]; inv1[1] = r01[2]; ir1[1] = r01[3];
            inv1[2] = r23[0]; ir1[2] = r23[1]; inv1[3] = r23[2]; ir1[3] = r23[3];
        }
        bool rok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rok[r] = ROWS ? true : (tm * 128 + 16 * wave + 4 * g + r < hw);      // (ROWS: rows past the kept count are padded copies of a kept row)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float inv2[4], ir2[4], zc[4], b2[4];
            bool cok[4];
            {
                const unsigned cs = lds0 + CVA_CST_OFF + (it & 3) * 2048 + (8 * c + 4 * h) * 16;
                f32x4 s0, s1, s2, s3;
                GD_DSR128(s0, cs, 0); GD_DSR128(s1, cs, 16); GD_DSR128(s2, cs, 32); GD_DSR128(s3, cs, 48);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
                inv2[0] = s0[0]; inv2[1] = s1[0]; inv2[2] = s2[0]; inv2[3] = s3[0];
                if constexpr (!ROWS) { ir2[0] = 1.0f / s0[1]; ir2[1] = 1.0f / s1[1]; ir2[2] = 1.0f / s2[1]; ir2[3] = 1.0f / s3[1]; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { cok[k] = tn * 128 + 8 * c + 4 * h + k < hw; zc[k] = 0.f; b2[k] = 0.f; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = 4 * h + k;
                    const bool ok = rok[r] && cok[k];
                    const float sv = acc[j][r] * inv1[r] * inv2[k];
                    const float e = ok ? __expf(sv) : 0.f;
                    const float sm = ok ? sv : 0.f;
                    zr[r] += e;
                    b1[r] = fmaf(fmaxf(t1v[r][h][k] * ir1[r], CV_EPS), sm, b1[r]);
                    if constexpr (!ROWS) {
                        zc[k] += e;
                        b2[k] = fmaf(fmaxf(t2v[ROWS ? 0 : j][r] * ir2[k], CV_EPS), sm, b2[k]);
                    }
                }
                // pin the running sums per row: the compiler otherwise keeps every addend alive to re-associate them (spilled exponentials)
                asm volatile("" : "+v"(zr[r]), "+v"(b1[r]));
                if constexpr (!ROWS) asm volatile("" : "+v"(zc[0]), "+v"(zc[1]), "+v"(zc[2]), "+v"(zc[3]), "+v"(b2[0]), "+v"(b2[1]), "+v"(b2[2]), "+v"(b2[3]));
            }
            if constexpr (!ROWS) {
                // column partials of this wave's 16-row strip -> LDS [wave][column 8 c + j][Z, B] (the lanes of row group 0 write 32 contiguous bytes per half)
                const f32x4 o0 = {cva_quad_rows_sum(zc[0]), cva_quad_rows_sum(b2[0]), cva_quad_rows_sum(zc[1]), cva_quad_rows_sum(b2[1])};
                const f32x4 o1 = {cva_quad_rows_sum(zc[2]), cva_quad_rows_sum(b2[2]), cva_quad_rows_sum(zc[3]), cva_quad_rows_sum(b2[3])};
                if (g == 0) {
                    const unsigned pa = lds0 + CVA_CP_OFF + (it & 1) * 8192 + wave * 1024 + c * 64 + h * 32;
                    CVA_DSW128(pa, o0, 0); CVA_DSW128(pa, o1, 16);
                }
            }
        }
        {   // row partials of the wave's 16 rows: lane (g, c < 4) holds row 4 g + c -> 16 lanes store 128 contiguous bytes
            float zrow[4], brow[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { zrow[r] = row16_sum(zr[r]); brow[r] = row16_sum(b1[r]); }
            const float zs = c == 0 ? zrow[0] : c == 1 ? zrow[1] : c == 2 ? zrow[2] : zrow[3];
            const float bs = c == 0 ? brow[0] : c == 1 ? brow[1] : c == 2 ? brow[2] : brow[3];
            const int rl = 16 * wave + 4 * g + c;
            if (c < 4) {
                if constexpr (ROWS)
                    *(__attribute__((address_space(1))) f32x2*)((uintptr_t)q.part1 + ((((long)sp * TL + tn) * kcap + tm * 128 + rl) * 2) * sizeof(float)) = f32x2{zs, bs};
                else if (tm * 128 + rl < hw)
                    *(__attribute__((address_space(1))) f32x2*)((uintptr_t)q.part1 + ((((long)sp * q.nslab + tn) * hw + tm * 128 + rl) * 2) * sizeof(float)) = f32x2{zs, bs};
            }
        }
        if (it + 1 < n_tiles) {
            const int ln = tile_at(it + 1);
            if (ln / TL != cur_panel) { cur_panel = ln / TL; load_panel(ln); }
            prefetch(it + 1);
        }
    }
    if constexpr (!ROWS) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int lp = tile_at(n_tiles - 1);
        const int spp = lp / per, rp = lp - spp * per, tmp = rp / TL, tnp = rp - tmp * TL;
        const unsigned pb = lds0 + CVA_CP_OFF + ((n_tiles - 1) & 1) * 8192 + ((16 * wave + (lane & 15)) * 2 + ((lane >> 4) & 1)) * 4;
        const float s = cva_sum8(pb);
        const int col = tnp * 128 + 16 * wave + (lane & 15);
        if (lane < 32 && col < hw)
            *(__attribute__((address_space(1))) float*)((uintptr_t)q.part2 + ((((long)spp * q.nslab + tmp) * hw + col) * 2 + (lane >> 4)) * sizeof(float)) = s;
    }
}
