"""ctypes binding of lib/libgd_hip.so (the C ABI declared in include/gd_hip.h).

There is NO CPU fallback: if the library is missing or a kernel reports an error the call
raises.  `lib()` loads lazily so that CPU-only tooling can import the package."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GD_HIP_LIB: another build of the same library (A/B timing of two builds in one session; never set in production)
LIB_PATH = os.environ.get("GD_HIP_LIB") or os.path.join(_HERE, "lib", "libgd_hip.so")
_lib = None

F32, BF16, F16 = 0, 1, 3

c_int, c_long, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes).  Must list every symbol of include/gd_hip.h (tests check this).
ABI_VERSION = 3      # include/gd_hip.h GD_ABI_VERSION: the version this signature table was written for
SIGNATURES = {
    "gd_last_error": (ctypes.c_char_p, []),
    "gd_abi_version": (c_int, []),
    "gd_debug_set": (c_int, [ctypes.c_char_p, c_int]),
    "gd_debug_get": (c_int, [ctypes.c_char_p]),
    "gd_gemm_phase_probe": (c_int, [c_int, c_void_p]),
    "gd_loss_combine_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_float), c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "gd_loss_combine_bwd": (c_int, [c_void_p, ctypes.POINTER(c_float), c_void_p, c_int, c_void_p, c_void_p]),
    "gd_depth_bwd_combine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gd_scale_and_transpose": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gd_gemm_nt": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_long, c_long,
                           c_int, c_long, c_long, c_long, c_int, c_int, c_float,
                           c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_long, c_int,
                           c_void_p, c_long, c_int, c_void_p, c_long, c_int, c_void_p]),
    "gd_gemm_nt_scaled": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_long, c_long,
                                  c_int, c_long, c_long, c_long, c_int, c_int, c_float, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_long, c_int,
                                  c_void_p, c_long, c_int, c_void_p, c_long, c_int, c_void_p]),
    "gd_layernorm_bwd_cast": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int, c_int, c_long, c_long, c_float, c_void_p]),
    "gd_cost_volume_kl_rows_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gd_cost_volume_kl_fwd_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                           c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gd_cost_volume_kl_bwd_rows_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "gd_cost_volume_kl_bwd_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                           c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gd_cost_volume_kl_bwd_h_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gd_cost_volume_kl_bwd_h": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gd_gemm_nt_copy16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_long, c_long, c_float, c_void_p, c_void_p,
                                  c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p]),
    "gd_tap_mean_norm_fwd_h": (c_int, [c_void_p, c_int, c_long, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gd_adapter_fused_h_supported": (c_int, [c_int, c_int, c_long]),
    "gd_adapter_fused_h_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                      c_int, c_int, c_int, c_void_p]),
    "gd_adapter_fused_h": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int, c_int, c_int, c_void_p]),
    "gd_gemm_tn_scaled": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_long, c_long,
                                  c_int, c_long, c_long, c_long, c_int, c_int, c_float, c_void_p, c_void_p]),
    "gd_cast_f16": (c_int, [c_void_p, c_void_p, c_long, c_int, c_long, c_float, c_void_p, c_void_p]),
    "gd_amax_scale": (c_int, [c_void_p, c_long, c_int, c_long, c_float, c_void_p, c_void_p, c_void_p]),
    "gd_scale_from_amax": (c_int, [c_void_p, c_float, c_void_p, c_void_p]),
    "gd_cast_f16_ex": (c_int, [c_void_p, c_void_p, c_long, c_int, c_long, c_float, c_void_p, c_void_p, c_void_p]),
    "gd_layernorm_bwd_ex": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int, c_int, c_long, c_long, c_float, c_void_p]),
    "gd_gemm_tn": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_long, c_long,
                           c_int, c_long, c_long, c_long, c_int, c_int, c_float, c_void_p]),
    "gd_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_long,
                                 c_long, c_float, c_int, c_int, c_void_p]),
    "gd_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                 c_long, c_long, c_float, c_int, c_int, c_void_p]),
    "gd_adapter_fused_supported": (c_int, [c_int, c_int, c_int]),
    "gd_adapter_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                 c_void_p]),
    "gd_l2norm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "gd_l2norm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gd_patch_im2col": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_int, c_void_p]),
    "gd_patch_im2col_strided": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_int, c_void_p]),
    "gd_assemble_tokens": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "gd_im2col3x3": (c_int, [c_void_p, c_long, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gd_col2im3x3": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gd_kp_gather_fwd": (c_int, [ctypes.POINTER(c_void_p), c_int, c_long, c_int, c_void_p, c_void_p, c_int, c_int,
                                 c_int, c_int, c_int, c_float, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gd_kp_gather_fwd_ln": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_int, c_long, c_long, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gd_kp_gather_bwd": (c_int, [ctypes.POINTER(c_void_p), c_int, c_long, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                 c_int, c_float, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gd_lora_bwd_fused": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gd_lora_bwd_fused_scaled": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "gd_conv_weight_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gd_kp_patch_bwd_det": (c_int, [c_void_p, c_void_p, c_int, c_long, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                    c_float, c_int, c_int, c_int, c_int, c_void_p]),
    "gd_kp_gather_bwd_det": (c_int, [c_void_p, c_int, c_long, c_int, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int, c_int,
                                     c_float, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gd_kp_patch_gather": (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                                   c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gd_kp_patch_gather_h": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                                     c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gd_stack3_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_long, c_long, c_int, c_int, c_int, c_void_p]),
    "gd_unpitch_tokens": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gd_tap_mean_fwd": (c_int, [ctypes.POINTER(c_void_p), c_int, c_long, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                c_void_p]),
    "gd_tap_mean_norm_fwd": (c_int, [ctypes.POINTER(c_void_p), c_int, c_long, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                     c_void_p]),
    "gd_tap_mean_bwd": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_int,
                                c_void_p]),
    "gd_kp_depth": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "gd_patch_mask": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gd_smooth_ap": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float,
                             c_void_p, c_void_p, c_void_p, c_void_p]),
    "gd_smooth_ap_me": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_void_p,
                                c_void_p, c_void_p, c_void_p]),
    "gd_pair_rank_workspace_bytes": (c_size_t, [c_int]),
    "gd_pair_rank": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gd_depth_l1": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gd_depth_head_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gd_depth_head_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "gd_sigmoid_temp": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_float, c_void_p]),
    "gd_masked_patch_cost_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_float,
                                         c_void_p]),
    "gd_masked_patch_cost_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                         c_float, c_int, c_float, c_void_p]),
    "gd_kl_divergence_map_fwd": (c_int, [c_void_p, c_void_p, c_long, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "gd_kl_divergence_map_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "gd_adamw_workspace_bytes": (c_size_t, []),
    "gd_clip_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_float, c_float, c_float,
                                   c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "gd_clip_adamw_ranges": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_float, c_float, c_float,
                                     c_float, c_float, c_float, c_float, c_void_p, c_void_p, ctypes.POINTER(c_long), c_int, c_void_p]),
    "gd_comm_rccl_version": (c_int, []),
    "gd_comm_unique_id": (c_int, [c_void_p]),
    "gd_comm_init": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_void_p]),
    "gd_comm_destroy": (c_int, [c_void_p]),
    "gd_flat_allreduce": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p]),
    "gd_unproject_depth": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gd_coview_masks": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_void_p]),
    "gd_nms_keypoints": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                 c_void_p]),
    "gd_nn_argmax": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gd_point_cloud_to_depth": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "gd_post_process_depth_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gd_post_process_depth": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_float,
                                      c_void_p, c_void_p]),
    "gd_rope_2d": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_long, c_float, c_float, c_int, c_void_p]),
    "gd_mast3r_attn_target": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "gd_cross_view_attn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "gd_cross_view_attn": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float,
                                   c_int, c_int, c_void_p, c_void_p]),
    "gd_split3": (c_int, [c_void_p, c_void_p, c_long, c_int, c_long, c_int, c_void_p]),
    "gd_cast": (c_int, [c_void_p, c_void_p, c_long, c_float, c_int, c_int, c_void_p]),
    "gd_attention_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "gd_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                 c_int, c_float, c_int, c_int, c_void_p]),
    "gd_cost_volume_kl_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "gd_cost_volume_teacher_stats": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gd_cost_volume_kl_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                      c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gd_cost_volume_kl_fwd_prenorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                              c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gd_cost_volume_kl_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                      c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gd_pil_resample_ksize": (c_int, [c_int, c_int]),
    "gd_pil_resample_coeffs": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gd_pil_resize_bicubic_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                         c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gd_u8_to_chw_float": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                   c_void_p]),
    "gd_color_jitter_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gd_gaussian_blur_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
}


class GdHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GdHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the HIP path)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        got = L.gd_abi_version()
        if got != ABI_VERSION:
            raise GdHipError(f"{LIB_PATH} has C-ABI version {got}, this package binds version {ABI_VERSION}: rebuild it "
                             "(`python -c 'import __graft_entry__ as g; g.build()'`)")
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise GdHipError(f"{what} failed (rc={rc}): {lib().gd_last_error().decode()}")


def dtype_code(t):
    import torch
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise GdHipError(f"unsupported dtype {t.dtype}")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
