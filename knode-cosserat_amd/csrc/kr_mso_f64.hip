// double instantiation of the overlapped persistent kernel (kr_mso_impl.hpp)
#define KR_MS_NO_INST
#include "kr_mso_impl.hpp"
namespace kr {
template int launch_mso_sim<double>(kr_handle*, const SimArgs<double>&, hipStream_t);
template int prepare_mso_sim<double>(kr_handle*, int64_t);
}
