// Error plumbing shared by every C-ABI entry point (see include/gd_hip.h).
#include "gd_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

void gd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gd_last_error(void) { return g_err; }

// ---- options: one table, read once ----
struct KnobDef { const char* name; const char* env; int GdKnobs::*field; int def; };
static const KnobDef KNOBS[] = {
    {"gemm_persist", "GD_GEMM_PERSIST", &GdKnobs::gemm_persist, 1},       {"gemm_small_tiles", "GD_GEMM_SMALL_TILES", &GdKnobs::gemm_small_tiles, 0},
    {"gemm_f32_big", "GD_GEMM_F32_BIG", &GdKnobs::gemm_f32_big, 0},       {"gemm_cstore", "GD_GEMM_CSTORE", &GdKnobs::gemm_cstore, 1},
    {"gemm_anat", "GD_GEMM_ANAT", &GdKnobs::gemm_anat, 0},                {"gemm_batch_big_m", "GD_GEMM_BATCH_BIG_M", &GdKnobs::gemm_batch_big_m, 384},
    {"gemm_krot", "GD_GEMM_KROT", &GdKnobs::gemm_krot, 1},                {"tn_blocks", "GD_TN_BLOCKS", &GdKnobs::tn_blocks, 0},
    {"attn_dma", "GD_ATTN_DMA", &GdKnobs::attn_dma, 1},                   {"attn_rot", "GD_ATTN_ROT", &GdKnobs::attn_rot, 1},
    {"attn_dq_dma", "GD_ATTN_DQ_DMA", &GdKnobs::attn_dq_dma, 1},
    {"attn_dkv_dma", "GD_ATTN_DKV_DMA", &GdKnobs::attn_dkv_dma, 1},
    {"attn_dkv_nw", "GD_ATTN_DKV_NW", &GdKnobs::attn_dkv_nw, 0},          {"cv_mask_skip", "GD_CV_MASK_SKIP", &GdKnobs::cv_mask_skip, 1},
    {"cv_persist", "GD_CV_PERSIST", &GdKnobs::cv_persist, 1},             {"cv_dbg", "GD_CV_DBG", &GdKnobs::cv_dbg, 0},
    {"cv_grid", "GD_CV_GRID", &GdKnobs::cv_grid, 0},                      {"pair_rank_wave", "GD_PAIR_RANK_WAVE", &GdKnobs::pair_rank_wave, 0},
    {"ln_16b", "GD_LN_16B", &GdKnobs::ln_16b, 1},                         {"adapter_persist", "GD_ADAPTER_PERSIST", &GdKnobs::adapter_persist, 1},
    {"reserve_cus", "GD_RESERVE_CUS", &GdKnobs::reserve_cus, 0},          {"gemm_group_m", "GD_GEMM_GROUP_M", &GdKnobs::gemm_group_m, 1},
};
static void gd_apply_reserve(GdKnobs& v) {
    int r = v.reserve_cus < 0 ? 0 : v.reserve_cus;
    if (r > v.ncu_dev - 8) r = v.ncu_dev - 8;      // never fewer than one CU per XCD
    v.ncu = v.ncu_dev - (r > 0 ? r : 0);
}

GdKnobs& gd_knobs_mut() {
    static GdKnobs k = [] {
        GdKnobs v = {};
        for (const KnobDef& d : KNOBS) {
            const char* e = getenv(d.env);
            v.*(d.field) = e ? atoi(e) : d.def;
        }
        v.ncu_dev = 256;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            v.ncu_dev = cus;
        gd_apply_reserve(v);
        return v;
    }();
    return k;
}

extern "C" int gd_debug_set(const char* name, int value) {
    GD_REQUIRE(name != nullptr, "gd_debug_set: null name");
    for (const KnobDef& d : KNOBS)
        if (strcmp(d.name, name) == 0) { gd_knobs_mut().*(d.field) = value; gd_apply_reserve(gd_knobs_mut()); return 0; }
    gd_set_error("gd_debug_set: unknown option '%s'", name);
    return -1;
}
extern "C" int gd_debug_get(const char* name) {
    if (name)
        for (const KnobDef& d : KNOBS)
            if (strcmp(d.name, name) == 0) return gd_knobs().*(d.field);
    gd_set_error("gd_debug_get: unknown option '%s'", name ? name : "(null)");
    return -1;
}
extern "C" int gd_abi_version(void) { return GD_ABI_VERSION; }

