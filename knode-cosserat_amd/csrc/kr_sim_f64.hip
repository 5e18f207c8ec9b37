// double instantiation of the simulation kernels
#define KR_SIM_T double
#include "kr_ms_impl.hpp"
