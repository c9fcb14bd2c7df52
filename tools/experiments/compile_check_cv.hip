// `make -C 3d-vlm-gd_amd/csrc experiments` (second translation unit): the shelved cost-volume kernels that build on cost_volume.hip's own device code
// (CvTileParams, the ring / epilogue helpers of cv_fwd_persist_kernel) still COMPILE against it.  Not linked into libgd_hip.so, not tested.
#include "cost_volume.hip"
#include "cv_stream.h"
#include "cv_split.h"
template __global__ void cv_stream_kernel<bf16, false, 0>(CvTileParams);
template __global__ void cv_stream_kernel<f16, true, 0>(CvTileParams);
template __global__ void cv_fwd_split_kernel<bf16, 0>(CvTileParams);
template __global__ void cv_fwd_split_kernel<float, 0>(CvTileParams);
