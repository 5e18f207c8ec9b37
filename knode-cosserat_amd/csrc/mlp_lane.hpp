// mlp_lane.hpp - residual MLP evaluated independently by every lane of a wave.
//
// Used inside the shooting sweep and the batched ODE kernel, where each lane
// owns a different input vector (one per rod x finite-difference column, or
// one per row).  This is the device form of CosseratRod.get_nn_output
// (reference cosserat_ode.py:90-112): arbitrary depth, Tanh / Softplus / ReLU /
// ELU / identity.
//
// Mapping: output-stationary tiles of 16 units live in VGPRs; the loop runs over
// the input units.  Weights are wave-uniform, stored transposed
// (Wt[in][out_pad]) so the 16 weights of a tile are contiguous and arrive by
// scalar loads (s_load_dwordx8/16 -> SGPR operand of v_fma); activations are
// per-lane columns buf[unit*stride + lane], bank-conflict free in LDS and
// coalesced in global memory.
#pragma once
#include "kr_internal.hpp"

namespace kr {

constexpr int MLP_TILE = 16;

// in: bufA holds the input vector (dims[0] units).  Result: the last layer's
// outputs are left in the returned buffer (bufA or bufB).
template <typename T>
__device__ __forceinline__ T* mlp_lane_eval(const MlpDev<T>& M, T* bufA, T* bufB, int stride) {
  T* bin = bufA;
  T* bout = bufB;
  for (int k = 0; k < M.n_layers; ++k) {
    const int nin = M.dims[k];
    const int nout = M.dims[k + 1];
    const int opad = M.out_pad[k];
    const int act = M.acts[k];
    const T* __restrict__ Wt = M.Wt[k];
    const T* __restrict__ bb = M.b[k];
    for (int o0 = 0; o0 < nout; o0 += MLP_TILE) {
      T acc[MLP_TILE];
#pragma unroll
      for (int t = 0; t < MLP_TILE; ++t) acc[t] = bb[o0 + t];
      const T* __restrict__ w = Wt + o0;
      for (int i = 0; i < nin; ++i) {
        const T xi = bin[i * stride];
#pragma unroll
        for (int t = 0; t < MLP_TILE; ++t) acc[t] = fma(w[t], xi, acc[t]);
        w += opad;
      }
#pragma unroll
      for (int t = 0; t < MLP_TILE; ++t) bout[(o0 + t) * stride] = activate<T>(act, acc[t]);
    }
    T* tmp = bin;
    bin = bout;
    bout = tmp;
  }
  return bin;
}

}  // namespace kr
