// double instantiation of the several-wavefront persistent kernel with the MLP on (kr_mswn_impl.hpp)
#define KR_MS_NO_INST
#include "kr_mswn_impl.hpp"
namespace kr {
template int nn_sim_waves_per_rod<double>(kr_handle*, int, int64_t);
template int launch_msw_nn_sim<double>(kr_handle*, int, const SimArgs<double>&, hipStream_t);
template int launch_msw_gh_sim<double>(kr_handle*, int, const SimArgs<double>&, hipStream_t);
}
