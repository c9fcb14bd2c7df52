// `make -C 3d-vlm-gd_amd/csrc experiments`: the shelved kernels of tools/experiments/ still COMPILE against the library's current sources
// (they are not linked into libgd_hip.so and not tested: each file's header says what was measured and why it was shelved).
#define GD_GEMM_EXPERIMENT32 1
#include "gemm.hip"             // GemmNtParams, gemm_persist.h, and — under the flag — gemm_persist32.h with its dispatch hook
#include "gemm_persist4.h"
#include "cv_persist256.h"
template __global__ void gemm_nt_p4_kernel<0>(GemmNtParams);
template __global__ void cv_fwd_p256_kernel<bf16>(Cv256Params);
