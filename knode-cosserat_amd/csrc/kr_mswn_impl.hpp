// kr_mswn_impl.hpp - several wavefronts per rod WITH the residual MLP inside the sweeps: host side of the NN
// instantiations of msw_sim_kernel (kr_msw_impl.hpp).  Own translation units (kr_mswn_f32.hip / kr_mswn_f64.hip) so
// that the largest kernels of the library compile in parallel with the others.
//
// Why: with the MLP on, one grid point costs ~20 k cycles of network evaluation (mlp_jvp.hpp) in a dependent chain
// exchange -> base chain -> act' -> JVP layers -> exchange, 43 % of which a lone wavefront spends waiting, and a
// sweep is 25 such points at four sub-intervals per wavefront.  Two wavefronts per rod cut the chain to 14 points
// (seven sub-intervals) and put two wavefronts on every SIMD at B = 1024, whose waits overlap; the distributed
// condensation between them costs a few thousand cycles per iteration, which is nothing here.  The records of the
// BDF2 history move from LDS to global memory for that (see msw_sim_kernel), so that four rods fit a CU.
#pragma once
#include "kr_ms_impl.hpp"

namespace kr {

template <typename T>
static bool msw_nn_fits(kr_handle* h, int W, int64_t B) {
  const RodConst<T>& P = consts<T>(h);
  if (P.N - 1 < 2 * (4 + 3 * (W - 1))) return false;
  // One wavefront per SIMD at most.  The kernel and its evaluator use all 512 registers of a SIMD lane; instantiations
  // limited to 256 (two wavefronts per SIMD, which B = 1024 would need) were built and measured: everything live in
  // the sweep is then spilled around every evaluator call and the scratch traffic of eight wavefronts per CU makes a
  // step 1.9 x SLOWER than one wavefront per rod (fp64 1.88 against 1.00 ms, fp32 1.01 against 0.57 ms at B = 1024).
  if (B * W > 1024) return false;
  const size_t bytes = sizeof(T) * (W == 2 ? msw_sim_lds_elems<T, 2>(P.N, true) : msw_sim_lds_elems<T, 4>(P.N, true));
  if (bytes > (size_t)h->lds_limit) return false;
  // every rod resident at once (a second round of workgroups would wait for the first to finish all its steps)
  return B <= 256 * (int64_t)((size_t)h->lds_limit / bytes);
}

template <typename T>
int nn_sim_waves_per_rod(kr_handle* h, int scheme, int64_t B) {
  const RodConst<T>& P = consts<T>(h);
  const MlpDev<T>& M = mlpdev<T>(h);
  if (h->ms_mode == 0 || h->persistent == 0 || scheme != KR_EULER || !P.diag) return 0;
  if (M.n_layers <= 0 || !M.mfma_ok || !M.jvp_ok || h->params.nn_input_history) return 0;
  if (h->waves_per_rod == 1) return 0;
  if (h->waves_per_rod == 2) return msw_nn_fits<T>(h, 2, B) ? 2 : 0;
  if (h->waves_per_rod == 4) return msw_nn_fits<T>(h, 4, B) ? 4 : 0;
  if (B * 4 <= 1024 && msw_nn_fits<T>(h, 4, B)) return 4;
  if (B * 2 <= 1024 && msw_nn_fits<T>(h, 2, B)) return 2;
  return 0;
}

template <typename T>
int launch_msw_nn_sim(kr_handle* h, int W, const SimArgs<T>& a, hipStream_t s) {
  const RodConst<T>& P = consts<T>(h);
  const MlpDev<T>& M = mlpdev<T>(h);
  if (!P.diag || (W != 2 && W != 4) || !msw_nn_fits<T>(h, W, a.B)) return 1;
  int rc = ensure_hist_ws(h, (size_t)a.B * P.N * HS_LEAN * sizeof(T));
  if (rc) return rc;
  SimArgs<T> a2 = a;
  a2.hist_ws = static_cast<T*>(h->hist_ws);
  h->last_waves_per_rod = W;
  return W == 2 ? launch_msw_sim_inst<T, true, 2, true>(P, M, a2, s) : launch_msw_sim_inst<T, true, 4, true>(P, M, a2, s);
}


// MLP off, long rods (HM = 1: history records in LDS, the two newest states read from A.states): what launch_msw_sim
// (kr_msw_impl.hpp) falls to when the leading slots of two states do not fit the LDS next to everything else (N = 400:
// 77 KB on top of 38 KB of records and 19 KB per wavefront)
template <typename T>
int launch_msw_gh_sim(kr_handle* h, int W, const SimArgs<T>& a, hipStream_t s) {
  const RodConst<T>& P = consts<T>(h);
  const MlpDev<T>& M = mlpdev<T>(h);
  if (W != 2 && W != 4) return 1;
  const size_t bytes = sizeof(T) * (W == 2 ? msw_sim_lds_elems<T, 2>(P.N, false, 1) : msw_sim_lds_elems<T, 4>(P.N, false, 1));
  if (bytes > (size_t)h->lds_limit || a.B > 256 * (int64_t)((size_t)h->lds_limit / bytes)) return 1;
  h->last_waves_per_rod = W;
  if (W == 2)
    return P.diag ? launch_msw_sim_inst<T, true, 2, false, 1, 1>(P, M, a, s) : launch_msw_sim_inst<T, false, 2, false, 1, 1>(P, M, a, s);
  return P.diag ? launch_msw_sim_inst<T, true, 4, false, 1, 1>(P, M, a, s) : launch_msw_sim_inst<T, false, 4, false, 1, 1>(P, M, a, s);
}

}  // namespace kr
