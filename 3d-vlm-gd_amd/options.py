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
    # tf32h: the fused adapter kernel also writes the NEXT block's LayerNorm 1 (fp16 rows + statistics) from the rows it still holds on chip
    # (gd_adapter_fused_h_ln; forward_all only): that block then runs no LayerNorm pass of its own
    "adapter_ln": ("GD_ADAPTER_LN", 1),
    "direct_grads": ("GD_DIRECT_GRADS", 1),      # fit_step: block weight gradients accumulate straight into the flat gradient buffer
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
