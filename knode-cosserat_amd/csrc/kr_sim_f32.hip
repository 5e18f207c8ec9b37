// float instantiation of the simulation kernels
#define KR_SIM_T float
#include "kr_ms_impl.hpp"
