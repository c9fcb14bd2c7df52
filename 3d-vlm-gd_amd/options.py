"""Host-side A/B switches of the engine, resolved ONCE (at import) from GD_* environment variables into one table — the Python twin of
csrc/gd_knobs.h.  Defaults are the measured-best paths; the switches exist so that the comparisons in profiles/README.md can be re-run.
`set_option` overrides one at run time (tests)."""
import os

_DEFS = {
    # name: (environment variable, default)
    "conv_at_kp": ("GD_CONV_AT_KP", 1),          # refine_conv evaluated at the keypoints only (vit.conv3x3_at_keypoints); 0: on the whole grid
    "conv_dx_at_kp": ("GD_CONV_DX_AT_KP", 1),    # its input gradient from the keypoint rows (no dense transposed convolution)
    "conv_stacked": ("GD_CONV_STACKED", 1),      # dense 3x3 conv as stacked-row GEMMs; 0: the im2col form
    "lora_fused": ("GD_LORA_FUSED", 1),          # one-pass LoRA backward kernel (bf16)
    "gather_det": ("GD_GATHER_DET", 1),          # atomics-free keypoint-gather backward
    "adapter_h_fused": ("GD_ADAPTER_H_FUSED", 1),  # tf32h: the one-pass fp16-operand adapter kernel (0: cast + two GEMMs)
    "cv_bwd_rows": ("GD_CV_BWD_ROWS", 1),        # sparse row masks: the cost-volume backward in its kept-row form too (0: the dense hw x hw backward)
    "h16_dy": ("GD_H16_DY", 1),                  # tf32h: the two dX GEMMs that feed a LayerNorm backward write fp16 in the block's scaled domain (0: fp32)
    "tap_norm_fused": ("GD_TAP_NORM_FUSED", 1),  # the taps' final LayerNorm applied inside the keypoint gather from the next block's row statistics (0: a LayerNorm pass per tap)
    # tf32h: no LayerNorm pass between the projection GEMM and fc1 — the projection's epilogue leaves the fp16 rows and per-row partial sums, fc1 takes
    # the UN-normalised rows against W diag(gamma) and normalises its product in the epilogue (csrc/gemm_persist.h LNF; vit._BlockFn.forward).
    # MEASURED, round 5 (profiles/r05_ln2_fold_ab.txt): the 74 us LayerNorm pass goes, but the projection's second store per output item costs +35 us
    # (184 -> 217: its epilogue is a serial phase at the chip's HBM rate), the statistics kernel 6, and fc1's epilogue +20 ... +26 (six more VALU
    # instructions per item in a GELU epilogue that is issue-bound): 58.51 -> 58.36 ms per step in one process, 60.30 -> 60.74 ms of kernel time
    # under rocprofv3 — a wash.  Off by default; kept under test (tests/test_gpu_gemm.py, test_gpu_step.py).
    "ln2_fold": ("GD_LN2_FOLD", 0),
    # tf32h: the fused adapter kernel also writes the NEXT block's LayerNorm 1 (fp16 rows + statistics) from the rows it still holds on chip
    # (gd_adapter_fused_h_ln; forward_all only): that block then runs no LayerNorm pass of its own
    "adapter_ln": ("GD_ADAPTER_LN", 1),
    "direct_grads": ("GD_DIRECT_GRADS", 1),      # fit_step: block weight gradients accumulate straight into the flat gradient buffer
    # the blocks' weight-gradient contractions (adapter up / down, LoRA-A: streams over the activations that nothing downstream in the backward reads) on
    # a SECOND stream, with `wgrad_reserve_cus` compute units kept free of the persistent kernels during the backward (csrc/gd_knobs.h reserve_cus).
    # 0 (default): off; 1: a plain second stream; 2: a stream confined to the reserved CUs (gd_stream_create_cu_mask).  Needs direct_grads.
    # MEASURED, round 5 (tools/ab_step.py, profiles/r05_wgrad_stream_ab.txt): 1 gains 0.2 ms of 58 (the side kernels' 768 blocks take every CU and the
    # one-block-per-CU GEMM behind them waits: the streams mostly take turns); 2 LOSES 12 - 31 ms — the side kernels are not HBM-bound per CU, their time
    # scales with 256 / k, and CU-time handed to them is CU-time the dX GEMMs lose: partitioning conserves work, it does not create throughput.
    "wgrad_stream": ("GD_WGRAD_STREAM", 0),
    "wgrad_reserve_cus": ("GD_WGRAD_RESERVE_CUS", 8),
}
_VALUES = {k: int(os.environ.get(env, str(d))) for k, (env, d) in _DEFS.items()}


def option(name):
    return _VALUES[name]


def set_option(name, value):
    """Override one switch (process-wide; tests and A/B tools)."""
    if name not in _VALUES:
        raise KeyError(f"unknown option {name!r}: {sorted(_VALUES)}")
    old = _VALUES[name]
    _VALUES[name] = int(value)
    return old
