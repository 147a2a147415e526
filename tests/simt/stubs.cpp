// TEST INFRASTRUCTURE: the entry points of dmd_conv_f16ws.hip, which is hand-scheduled around inline gfx950 assembly and is
// not part of the host build.  Nothing is eligible for it here, so dmd_conv2d takes the conv_mfma instances instead.
#include "../../diamond_amd/csrc/dmd_common.h"

extern "C" int dmd_conv2d_f16x2_eligible(const dmd_conv_params*) { return 0; }
extern "C" int dmd_conv2d_proj_eligible(const dmd_conv_params*) { return 0; }
int dmd_launch_conv_f16ws(const dmd_conv_params&, hipStream_t) {
  dmd_set_error("conv_f16ws_kernel is not part of the host (SIMT interpreter) build");
  return 1;
}
extern "C" int dmd_pack_conv_weight_f16x2(const float*, void*, int, int, int, int, dmd_stream_t) {
  dmd_set_error("dmd_pack_conv_weight_f16x2 is not part of the host (SIMT interpreter) build; dmd_pack_jobs is");
  return 1;
}
extern "C" int dmd_ws_trace_dump(unsigned long long*, int*) { return 0; }
// marker: diamond_amd/native.py refuses to load a library that exports this (the product has no CPU path)
extern "C" int dmd_simt_host_build(void) { return 1; }
