// A/B options of the library, resolved ONCE (first use, C++11 thread-safe static initialisation) from GD_* environment variables
// into one plain struct; no entry point reads the environment or keeps first-use state of its own.  The defaults are the
// measured-best settings; the knobs exist so that the comparisons in profiles/README.md can be re-run.  `gd_debug_set`
// (include/gd_hip.h) overrides one option at run time for tests and anatomy tools — a process-wide switch, not for production.
#pragma once

struct GdKnobs {
    int gemm_persist;      // GD_GEMM_PERSIST      1: persistent 256x256 kernel for big bf16 shapes; 0: staged tile kernels
    int gemm_small_tiles;  // GD_GEMM_SMALL_TILES  1: force the 128x128 configuration
    int gemm_f32_big;      // GD_GEMM_F32_BIG      1: 256x256 tiles for f32 operands
    int gemm_cstore;       // GD_GEMM_CSTORE       C store policy of the staged kernels (0 LDS-staged, 1 direct)
    int gemm_group_m;      // GD_GEMM_GROUP_M      persistent kernel: row panels per W panel in an XCD's tile walk (1 = row-panel-major)
    int gemm_krot;         // GD_GEMM_KROT         per-tile K-step rotation of the persistent kernel (0 off)
    int gemm_batch_big_m;  // GD_GEMM_BATCH_BIG_M  batched gemm_nt: smallest M served by the 256 x 256 kernels (384: the kept-row cost-volume contractions, 97 vs 149 us; un-batched: 1024)
    int gemm_anat;         // GD_GEMM_ANAT         anatomy instantiations of the persistent main loop (0 = the product kernel)
    int tn_blocks;         // GD_TN_BLOCKS         target block count of the tile TN GEMM (0 auto)
    int attn_dma;          // GD_ATTN_DMA          1: LDS-DMA attention forward; 0: register-staged
    int attn_rot;          // GD_ATTN_ROT          1: forward x-block xb starts at key tile 2 xb
    int attn_dq_dma;       // GD_ATTN_DQ_DMA       1: dQ (16-bit operands) on the LDS-DMA ring (round 6); 0: register-staged tiles
    int attn_dkv_dma;      // GD_ATTN_DKV_DMA      1: dK/dV (four-wave form, 16-bit operands) on the LDS-DMA ring (round 6); 0: register-staged tiles
    int attn_dkv_nw;       // GD_ATTN_DKV_NW       0 auto | 4 | 8 waves per dK/dV block
    int cv_mask_skip;      // GD_CV_MASK_SKIP      1: masked teacher rows are not fetched
    int cv_persist;        // GD_CV_PERSIST        1: persistent cost-volume forward
    int cv_dbg;            // GD_CV_DBG            anatomy switches of the cost-volume forward (tools/cv_anatomy.py)
    int cv_grid;           // GD_CV_GRID           block count cap of the persistent cost-volume forward (tests: many tiles per block)
    int pair_rank_wave;    // GD_PAIR_RANK_WAVE    0 tiled kernel | 1 | 2 | 3 older forms
    int ln_16b;            // GD_LN_16B            1: 16-byte LayerNorm accesses
    int adapter_persist;   // GD_ADAPTER_PERSIST   blocks per CU of the persistent adapter kernel (0: one block per tile)
    int reserve_cus;       // GD_RESERVE_CUS       compute units the persistent kernels leave free (data parallelism: RCCL's kernels need CUs to run
                           //                      UNDER a backward made of one-block-per-CU launches; dp / bench.py set it when world > 1; 0 otherwise)
    int ncu_dev;           // compute units of the current device at first use (256 on MI355X)
    int ncu;               // = ncu_dev - reserve_cus: the grid of every persistent kernel
};
GdKnobs& gd_knobs_mut();                                                // cabi.hip
static inline const GdKnobs& gd_knobs() { return gd_knobs_mut(); }
