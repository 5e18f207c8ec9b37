// float instantiation of the several-wavefront persistent kernel with overlapped time steps (kr_mswo_impl.hpp)
#define KR_MS_NO_INST
#include "kr_mswo_impl.hpp"
namespace kr {
template int launch_mswo_sim<float>(kr_handle*, int, const SimArgs<float>&, hipStream_t);
}
