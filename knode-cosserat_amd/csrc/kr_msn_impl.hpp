// kr_msn_impl.hpp - host side of the one-launch-per-step multiple-shooting kernels WITH the residual MLP inside the
// sweeps (ms_step_kernel<T, DIAG, SCHEME, HS, NN = true>, kr_ms_impl.hpp): what kr_step_batch runs with the MLP on, and
// what kr_simulate_batch falls to where the persistent kernels do not serve the problem (N > 128, RK4 sweeps, full
// material matrices, a second hidden layer wider than 192).  Own translation units (kr_msn_f32.hip / kr_msn_f64.hip),
// compiled with -mllvm -amdgpu-spill-sgpr-to-vgpr=0 (Makefile): these kernels spill ~600 registers, and with SGPR
// spills in VGPR lanes hipcc 7.2 opens whole-wave-mode brackets around their copies and schedules ordinary VGPR spills
// INTO those brackets - a miscompile whenever a lane that was idle at the spill reads the slot later (found once in
// msw_sim_kernel, DESIGN.md section 9 and LABBOOK.md section 4).  With SGPR spills in scratch memory no bracket exists; the build runs
// tools/wwm_spill_scan.py over the assembly of every translation unit and fails on a hit.
#pragma once
#include "kr_ms_impl.hpp"

namespace kr {
template <typename T>
int launch_ms_step_nn(kr_handle* h, int scheme, const StepArgs<T>& a, hipStream_t s) {
  return launch_ms_nn<T, true>(h, scheme, a, s);
}
}  // namespace kr
