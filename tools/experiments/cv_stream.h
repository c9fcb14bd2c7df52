// SHELVED EXPERIMENT (round 6) — the STREAMED-TEACHER form of the persistent cost-volume kernel.  Compiles against cost_volume.hip (`make experiments`,
// compile_check_cv.hip); not part of the library.
//
// Why it was built.  The anatomy of cv_fwd_persist_kernel (tools/probe_cv_streams.py, bench.py `attainable.memory_streams_us`): its teacher stream ALONE
// takes 190-220 us of the 245-260 us launch although it moves only 480 MB (2.4 TB/s; gd_cost_volume_teacher_stats streams the same maps at 5.8 TB/s with ~200 KB
// of loads in flight per CU), and the feature ring (120 us alone) hides under it.  A compute wave can only re-issue its 16 teacher loads once the epilogue
// has consumed the 64 registers they land in, so a tile's 131 KB leave as ONE burst per CU that takes 12-13 us to come back; the epilogue then waits.
//
// What it does.  Two of the four loader waves become TEACHER STREAMERS: each keeps a window of 8 slices x 4 KB = 32 dwordx4 loads per lane (128 registers;
// 64 KB in flight per CU) rolling across tile boundaries — a slice = one of the 16 teacher vectors of each of the four compute waves it serves, fetched with
// exactly the addresses the compute wave would have used — and hands the slices over through an LDS staging area (2 x 16 KB, parity of the K step): the
// streamers write a step's two slices, the step's barrier publishes them, the compute waves copy them into the same accumulator-layout registers as before.
// The feature ring gives up one slot for the staging area (three slots, two steps ahead: measured equal) and runs on two loader waves.
// Schedule (any even K-step count nk >= 8): group A (t1v) is copied out on the FIRST four K steps of its tile, each copy freeing the registers for a
// group-B load (t2v); group B is copied out on the LAST four steps, each copy followed by the NEXT tile's group-A load, which lands under the epilogue.
//
// What it measured (P = 32, 37 x 37, D = 768, bf16; whole op, us): forward 229-231 against 244 (-6 %); backward 286 against 284 (nothing); a first schedule
// that spread the slices over all twelve steps: the same.  Compile-time anatomy (ANAT): ring + MFMA only 104 (the persistent kernel's anatomy build: 130);
// + the streamers' staging writes 126; everything but the teacher loads 237-240; everything 238-241 — the stream itself is hidden (+3 us), what costs is
// the hand-over through LDS and the barrier it rides on, and the epilogue still runs with nothing beside it.  Issuing the loads from INSIDE the epilogue
// of cv_fwd_persist_kernel (row block by row block as the registers free up, no second kernel) measured 243 / 290.
// Why it is shelved: 6 % of the dense forward, none of the backward, no mask-steered load skipping yet, for a second 250-line kernel family.
#define CVS_SLOTS 3
#define CVS_STG_OFF (CVS_SLOTS * CVP_STAGE)      // [parity 2][slot-in-step 2][compute wave 8][1 KB]; ends at CVP_STAT_OFF
static_assert(CVS_STG_OFF + 32768 == CVP_STAT_OFF, "cv_stream: LDS layout");

// global address of teacher vector v (0..7: t1v[v >> 2][v & 3]; 8..15: t2v[(v - 8) >> 2][(v - 8) & 3]) of compute wave w, this lane — the addresses
// cv_fwd_persist_kernel's prefetch() uses (clamped into the pair's map: entries past the ragged edge are re-reads that the epilogue multiplies by zero)
__device__ __forceinline__ const float* cvs_addr(const float* T1, const float* T2, const CvpTile& t, int v, int w, int lane, int hw, int ldt) {
    const int wm = w >> 1, wn = w & 1, g = lane >> 4, c = lane & 15;
    if (v < 8) {
        const int ib = v >> 2, r = v & 3;
        const int row = min(t.tm * 128 + wm * 32 + ib * 16 + 4 * g + r, hw - 1), col0 = min(t.tn * 128 + wn * 64 + 4 * c, ldt - 4);
        return T1 + ((long)t.p * hw + row) * ldt + col0;
    }
    const int ib = (v - 8) >> 2, jb = (v - 8) & 3;
    const int col = min(t.tn * 128 + wn * 64 + 4 * c + jb, hw - 1), row0 = min(t.tm * 128 + wm * 32 + ib * 16 + 4 * g, ldt - 4);
    return T2 + ((long)t.p * hw + col) * ldt + row0;
}

template <typename T, bool BWD, int ANAT = 0>      // ANAT (timing only): 1 = streamers issue no loads, 2 = no staging traffic either, 4 = no epilogue
__global__ __launch_bounds__(768) void cv_stream_kernel(CvTileParams q) {
    __shared__ __attribute__((aligned(16))) char smem[CVP_SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = q.hw, tiles = q.tiles, ldt = q.ldt;
    const long rowb = (long)q.C * sizeof(T);
    const int nk = (int)(rowb / 128);
    const int total = q.P * tiles * tiles, nbx = gridDim.x >> 3, xc = blockIdx.x & 7, kb = blockIdx.x >> 3;
    const int qT = total >> 3, rT = total & 7;
    const int beg = xc < rT ? xc * (qT + 1) : rT * (qT + 1) + (xc - rT) * qT;
    const int cnt = qT + (xc < rT ? 1 : 0);
    const int n_tiles = cnt > kb ? (cnt - kb + nbx - 1) / nbx : 0;
    if (n_tiles == 0) return;
    const int n_total = n_tiles * nk;
    const unsigned smem_base = (unsigned)(uintptr_t)smem;

    if (wave >= 10) {
        // ======================================= TEACHER STREAMERS =======================================
        const int sw = wave - 10;
        __builtin_amdgcn_s_setprio(3);          // its few LDS writes per step go out ahead of the compute waves' fragment reads
        f32x4 R[8][4];                          // [slice & 7][served compute wave sw*4 + j]
        CvpTile tc = cvp_tile(beg + kb, tiles);
#pragma unroll
        for (int v = 0; v < 8; ++v)
#pragma unroll
            for (int j = 0; j < 4; ++j) R[v][j] = *(const f32x4*)cvs_addr(q.t1, q.t2, tc, v, sw * 4 + j, lane, hw, ldt);
        for (int it = 0; it < n_tiles; ++it) {
            // (past the last tile the window re-reads the last tile's group A: the number of loads in flight behind a slice stays what the compiler's wait
            // counts assume, and nobody copies them out)
            const CvpTile tn = cvp_tile(beg + kb + min(it + 1, n_tiles - 1) * nbx, tiles);
#pragma unroll
            for (int h = 0; h < 2; ++h) {          // h = 0: first four steps, group A out, group B in; h = 1: last four steps, group B out, next tile's group A in
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int s8 = 2 * k + u;      // register slot; slice h*8 + s8 goes out, slice (1-h)*8 + s8 comes in
                        char* dst = smem + CVS_STG_OFF + (k & 1) * 16384 + (u * 8 + sw * 4) * 1024 + lane * 16;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (!(ANAT & 2)) *(f32x4*)(dst + j * 1024) = R[s8][j];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (!(ANAT & 1)) R[s8][j] = *(const f32x4*)cvs_addr(q.t1, q.t2, h ? tn : tc, (1 - h) * 8 + s8, sw * 4 + j, lane, hw, ldt);
                    }
                    cvp_barrier();
                }
                if (h == 0)
                    for (int k = 8; k < nk; ++k) cvp_barrier();
            }
            tc = tn;
        }
        cvp_barrier();
        return;
    }

    if (wave >= 8) {
        // ======================================= FEATURE LOADERS (two waves, three-slot ring) =======================================
        const int lw = wave - 8;
        const char* asrc[8];
        const char* wsrc[8];
        int it_i = 0, k_i = 0, slot_i = 0;
        auto issue = [&]() {
            char* sA = smem + slot_i * CVP_STAGE;
            char* sB = sA + 128 * 128;
            slot_i = slot_i == CVS_SLOTS - 1 ? 0 : slot_i + 1;
            if (k_i == 0) {
                const CvpTile t = cvp_tile(beg + kb + it_i * nbx, tiles);
                const char* Ab = (const char*)q.f1 + (long)t.p * hw * rowb;
                const char* Wb = (const char*)q.f2 + (long)t.p * hw * rowb;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = (lw * 8 + i) * 8 + (lane >> 3);
                    asrc[i] = Ab + (long)min(t.tm * 128 + row, hw - 1) * rowb + (((lane & 7) ^ swz(row)) * 16);
                    wsrc[i] = Wb + (long)min(t.tn * 128 + cv_nperm64(row), hw - 1) * rowb + (((lane & 7) ^ swz(row)) * 16);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {      // the tile's 256 row / column statistics: the step's oldest pieces, parity buffer of the tile
                    const int e = (lw * 2 + h) * 64 + lane, which = e >> 7;
                    const int idx = min((which ? t.tn : t.tm) * 128 + (e & 127), hw - 1);
                    const char* ssrc = (const char*)(q.stats + (((long)t.p * 2 + which) * hw + idx) * 4);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ssrc,
                                                     (__attribute__((address_space(3))) void*)(smem + CVP_STAT_OFF + (it_i & 1) * 4096 + (lw * 2 + h) * 1024),
                                                     16, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + (long)k_i * 128),
                                                 (__attribute__((address_space(3))) void*)(sA + (lw * 8 + i) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + (long)k_i * 128),
                                                 (__attribute__((address_space(3))) void*)(sB + (lw * 8 + i) * 1024), 16, 0, 0);
            if (++k_i == nk) { k_i = 0; ++it_i; }
        };
        const uintptr_t p1 = (uintptr_t)q.part1, p2 = (uintptr_t)q.part2;
        auto flush = [&](int it) {      // previous tile's partial sums: LDS -> slabs (cv_fwd_persist_kernel's flush, two entries per lane)
            const CvpTile t = cvp_tile(beg + kb + it * nbx, tiles);
            const unsigned pb = smem_base + CVP_PART_OFF + (it & 1) * 6144;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = (lw * 2 + h) * 64 + lane;
                const bool isrow = e < 128;
                const int c = e & 127;
                const unsigned zo = isrow ? c : 512 + c, bo = isrow ? 256 + c : 1024 + c;
                float Z = cvp_lds_f32(pb + zo * 4) + cvp_lds_f32(pb + (zo + 128) * 4);
                float B = cvp_lds_f32(pb + bo * 4) + cvp_lds_f32(pb + (bo + 128) * 4);
                const float Z2 = cvp_lds_f32(pb + (512 + 256 + c) * 4) + cvp_lds_f32(pb + (512 + 384 + c) * 4);
                const float B2 = cvp_lds_f32(pb + (1024 + 256 + c) * 4) + cvp_lds_f32(pb + (1024 + 384 + c) * 4);
                if (!isrow) { Z += Z2; B += B2; }
                const int idx = (isrow ? t.tm : t.tn) * 128 + c, slab = isrow ? t.tn : t.tm;
                const uintptr_t base = isrow ? p1 : p2;
                if (idx < hw)
                    *(__attribute__((address_space(1))) f32x2*)(base + ((((long)t.p * q.nslab + slab) * hw + idx) * 2) * sizeof(float)) = f32x2{Z, B};
            }
        };
        for (int n = 0; n < 2 && n < n_total; ++n) issue();
        int kk = 0, it = 0;
        for (int n = 0; n < n_total; ++n) {
            // step n is in LDS once at most the youngest step's 16 pieces are outstanding (loads retire in order; the flush's stores can only make the
            // count larger)
            if (n_total - 1 - n >= 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (n + 2 < n_total) issue();
            if (!BWD && kk == 0 && it > 0) flush(it - 1);
            if (++kk == nk) { kk = 0; ++it; }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!BWD) flush(n_tiles - 1);
        return;
    }

    // ======================================= COMPUTE waves =======================================
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c = lane & 15;
    typedef typename Mma<T>::Frag Frag;
    const int sa = swz(c);
    const int abase = (wm * 32 + c) * 128, bbase = 128 * 128 + (wn * 64 + c) * 128;
    f32x4 t1v[2][4], t2v[2][4];
    int slot = 0;
    for (int it = 0; it < n_tiles; ++it) {
        const CvpTile t = cvp_tile(beg + kb + it * nbx, tiles);
        unsigned char mb[12];
        if (BWD) {      // the backward's epilogue needs the row masks themselves: twelve byte loads that land under the main loop
            const unsigned char* M1 = q.m1 ? q.m1 + (long)t.p * hw : nullptr;
            const unsigned char* M2 = q.m2 ? q.m2 + (long)t.p * hw : nullptr;
#pragma unroll
            for (int i = 0; i < 8; ++i) mb[i] = M1 ? M1[min(t.tm * 128 + wm * 32 + (i >> 2) * 16 + 4 * g + (i & 3), hw - 1)] : 1;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) mb[8 + jb] = M2 ? M2[min(t.tn * 128 + wn * 64 + 4 * c + jb, hw - 1)] : 1;
        }
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto kstep = [&](const char* sb) {
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const int co = (((kc * 4 + g) ^ sa) * 16);
                Frag a[2], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) b[u] = *(const Frag*)(sb + bbase + u * 2048 + co);
#pragma unroll
                for (int u = 0; u < 2; ++u) a[u] = *(const Frag*)(sb + abase + u * 2048 + co);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(a[i], b[j], acc[i][j]);
            }
        };
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                cvp_barrier();
                const char* sb = smem + slot * CVP_STAGE;
                slot = slot == CVS_SLOTS - 1 ? 0 : slot + 1;
#pragma unroll
                for (int u = 0; u < 2; ++u) {      // this step's two teacher slices: staging -> the accumulator-layout registers
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
                    if (!(ANAT & 2)) x = *(const f32x4*)(smem + CVS_STG_OFF + (k & 1) * 16384 + (u * 8 + wave) * 1024 + lane * 16);
                    const int s8 = 2 * k + u;
                    if (h == 0) t1v[s8 >> 2][s8 & 3] = x;
                    else t2v[s8 >> 2][s8 & 3] = x;
                }
                kstep(sb);
            }
            if (h == 0)
                for (int k = 8; k < nk; ++k) {
                    cvp_barrier();
                    const char* sb = smem + slot * CVP_STAGE;
                    slot = slot == CVS_SLOTS - 1 ? 0 : slot + 1;
                    kstep(sb);
                }
        }
        const f32x4* sSt = (const f32x4*)(smem + CVP_STAT_OFF + (it & 1) * 4096);
        float* sP = (float*)(smem + CVP_PART_OFF + (it & 1) * 6144);
        if (BWD) {
            unsigned keepbits = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) keepbits |= mb[i] ? 1u << i : 0u;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) keepbits |= mb[8 + jb] ? 256u << jb : 0u;
            cvp_epilogue_bwd<T>(q, acc, t1v, t2v, sSt, t, hw, wm, wn, g, c, keepbits);
        } else if (!(ANAT & 4))
            cvp_epilogue_fwd(acc, t1v, t2v, sSt, sP, t, hw, wm, wn, g, c);
        else if (acc[0][0][0] == 12345.678f) ((float*)smem)[tid] = acc[1][1][1] + t1v[0][0][0] + t2v[1][1][1];
    }
    cvp_barrier();
}
