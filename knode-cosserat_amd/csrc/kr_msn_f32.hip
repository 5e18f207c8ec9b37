// float instantiation of the one-launch-per-step multiple-shooting kernels with the MLP on (kr_msn_impl.hpp)
#define KR_SIM_T float
#define KR_MS_NO_INST
#include "kr_msn_impl.hpp"
namespace kr {
template int launch_ms_step_nn<float>(kr_handle*, int, const StepArgs<float>&, hipStream_t);
}
