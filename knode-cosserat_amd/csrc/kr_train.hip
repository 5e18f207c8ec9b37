// kr_train.hip - KNODE one-step-ahead training path (placeholder until the kernels land in this round).
#include "kr_internal.hpp"
using namespace kr;
extern "C" {
int kr_next_segment_physics(kr_handle*, int64_t, int, const void*, const void*, const void*, const void*,
                            const int32_t*, void*, int, void*, int, void*) {
  set_error("kr_next_segment_physics: not implemented yet");
  return KR_E_UNSUPPORTED;
}
size_t kr_mlp_ws_bytes(int, const int32_t*, int64_t) { return 0; }
int kr_mlp_forward(kr_handle*, int64_t, int, const int32_t*, const int32_t*, const float* const*, const float* const*,
                   const float*, int, float*, void*, void*) {
  set_error("kr_mlp_forward: not implemented yet");
  return KR_E_UNSUPPORTED;
}
int kr_mlp_backward(kr_handle*, int64_t, int, const int32_t*, const int32_t*, const float* const*, const float*, int,
                    const float*, const void*, float* const*, float* const*, void*) {
  set_error("kr_mlp_backward: not implemented yet");
  return KR_E_UNSUPPORTED;
}
int kr_loss_fwd_bwd(kr_handle*, int64_t, int, const float*, const float*, const float*, const int32_t*, double, float*,
                    float*, float*, void*) {
  set_error("kr_loss_fwd_bwd: not implemented yet");
  return KR_E_UNSUPPORTED;
}
}
