// kr_ode.hip - batched per-segment derivative: CosseratRodTorch.ODE_parallel
// (reference cosserat_ode_torch.py:217-322) and, row by row, CosseratRod.ODE
// (cosserat_ode.py:114-186).  Q independent rows, 53 values in / 25 out per
// row: an HBM-streaming kernel (78*sizeof(T) algorithmic bytes per row).
//
// Rows are row-major with odd lengths (19, 6, 3), so a lane-per-row load would
// touch a different cache line per lane.  Each workgroup therefore stages a
// tile of 128 rows through LDS: global -> LDS with fully coalesced contiguous
// reads, compute from LDS with a conflict-free odd row pitch, results back to
// LDS and out with contiguous writes.
#include <cstdint>
#include <type_traits>

#include "kr_internal.hpp"
#include "mlp_lane.hpp"

namespace kr {

constexpr int ODE_ROWS = 128;  // rows per workgroup = threads per workgroup

template <typename T>
__device__ __forceinline__ void tile_in(const T* __restrict__ g, T* l, int64_t row0, int64_t Q, int width, int tid) {
  const int64_t base = row0 * width;
  const int64_t lim = (Q - row0 < ODE_ROWS ? Q - row0 : ODE_ROWS) * width;
  for (int64_t i = tid; i < (int64_t)ODE_ROWS * width; i += ODE_ROWS) l[i] = i < lim ? g[base + i] : T(0);
}
template <typename T>
__device__ __forceinline__ void tile_out(T* __restrict__ g, const T* l, int64_t row0, int64_t Q, int width, int tid) {
  const int64_t base = row0 * width;
  const int64_t lim = (Q - row0 < ODE_ROWS ? Q - row0 : ODE_ROWS) * width;
  for (int64_t i = tid; i < lim; i += ODE_ROWS) g[base + i] = l[i];
}
// Full tiles of 16-byte aligned arrays: 16 bytes per lane, every load of the tile issued before the first
// LDS write (the scalar forms above keep one 4/8-byte access in flight per lane and reach a quarter of the
// HBM rate).  WIDTH * ODE_ROWS * sizeof(T) is a multiple of 16 for every row width used here.
template <typename T, int WIDTH>
struct TileVec {
  using V = typename std::conditional<sizeof(T) == 8, double __attribute__((ext_vector_type(2))),
                                      float __attribute__((ext_vector_type(4)))>::type;
  static constexpr int PER = 16 / sizeof(T);
  static constexpr int NVEC = ODE_ROWS * WIDTH / PER;
  static constexpr int TRIPS = (NVEC + ODE_ROWS - 1) / ODE_ROWS;
  static_assert((ODE_ROWS * WIDTH) % PER == 0, "tile is not a whole number of 16-byte vectors");
  V r[TRIPS];
  __device__ __forceinline__ void load(const T* __restrict__ g, int64_t row0, int tid) {
    const V* src = reinterpret_cast<const V*>(g + row0 * WIDTH);
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
      const int i = tid + k * ODE_ROWS;
      if (i < NVEC) r[k] = __builtin_nontemporal_load(src + i);
    }
  }
  __device__ __forceinline__ void to_lds(T* l, int tid) const {
    V* dst = reinterpret_cast<V*>(l);
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
      const int i = tid + k * ODE_ROWS;
      if (i < NVEC) dst[i] = r[k];
    }
  }
  static __device__ __forceinline__ void store(T* __restrict__ g, const T* l, int64_t row0, int tid) {
    V* dst = reinterpret_cast<V*>(g + row0 * WIDTH);
    const V* src = reinterpret_cast<const V*>(l);
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
      const int i = tid + k * ODE_ROWS;
      if (i < NVEC) __builtin_nontemporal_store(src[i], dst + i);
    }
  }
};

template <typename T, bool DIAG, bool NN, bool NNHIST>
__global__ __launch_bounds__(ODE_ROWS) void ode_batch_kernel(const RodConst<T> P, const MlpDev<T> M, int64_t Q,
                                                             const T* __restrict__ y, const T* __restrict__ yh,
                                                             const T* __restrict__ zh, const T* __restrict__ tf,
                                                             T* __restrict__ dys, T* __restrict__ z, T* act_ws,
                                                             int vec_ok) {
  // LDS: y[128][19] yh[128][19] zh[128][6] tf[128][3]  (47 per row; odd pitches 19/19 and 6,3 are read
  // row-per-lane: bank = (row*pitch + c) mod 32/64 - pitch 19 is conflict free, 6 and 3 are 2-way at worst)
  __shared__ __attribute__((aligned(16))) T sy[ODE_ROWS * 19];
  __shared__ __attribute__((aligned(16))) T syh[ODE_ROWS * 19];
  __shared__ __attribute__((aligned(16))) T szh[ODE_ROWS * 6];
  __shared__ __attribute__((aligned(16))) T stf[ODE_ROWS * 3];
  const int tid = threadIdx.x;
  for (int64_t row0 = (int64_t)blockIdx.x * ODE_ROWS; row0 < Q; row0 += (int64_t)gridDim.x * ODE_ROWS) {
    const bool full = vec_ok && row0 + ODE_ROWS <= Q;  // uniform per workgroup
    if (full) {
      TileVec<T, 19> a, b;
      TileVec<T, 6> c;
      TileVec<T, 3> d;
      a.load(y, row0, tid); b.load(yh, row0, tid); c.load(zh, row0, tid); d.load(tf, row0, tid);
      a.to_lds(sy, tid); b.to_lds(syh, tid); c.to_lds(szh, tid); d.to_lds(stf, tid);
    } else {
      tile_in(y, sy, row0, Q, 19, tid);
      tile_in(yh, syh, row0, Q, 19, tid);
      tile_in(zh, szh, row0, Q, 6, tid);
      tile_in(tf, stf, row0, Q, 3, tid);
    }
    __syncthreads();
    T yr[19];
#pragma unroll
    for (int c = 0; c < 19; ++c) yr[c] = sy[tid * 19 + c];
    if (row0 + tid >= Q) yr[3] = T(1);  // padding rows: keep the quaternion invertible
    RodState<T> ys_in = rows_to_state(yr);
    RodHist<T> hst;
    hst.qh = {syh[tid * 19 + 13], syh[tid * 19 + 14], syh[tid * 19 + 15]};
    hst.wh = {syh[tid * 19 + 16], syh[tid * 19 + 17], syh[tid * 19 + 18]};
    hst.vh = {szh[tid * 6 + 0], szh[tid * 6 + 1], szh[tid * 6 + 2]};
    hst.uh = {szh[tid * 6 + 3], szh[tid * 6 + 4], szh[tid * 6 + 5]};
    hist_derive(P, hst);
    const V3<T> tfv{stf[tid * 3 + 0], stf[tid * 3 + 1], stf[tid * 3 + 2]};
    const V3<T> fconst{P.rhoAg[0] + tfv.x, P.rhoAg[1] + tfv.y, P.rhoAg[2] + tfv.z};
    RodState<T> k;
    V3<T> v, u;
    ode_eval<T, DIAG>(P, ys_in, hst, fconst, k, v, u);
    T out[25];
    {
      T kr_[19];
      state_to_rows(k, kr_);
#pragma unroll
      for (int c = 0; c < 19; ++c) out[c] = kr_[c];
      out[19] = v.x; out[20] = v.y; out[21] = v.z; out[22] = u.x; out[23] = u.y; out[24] = u.z;
    }
    if constexpr (NN) {
      // per-lane activation columns in global scratch (coalesced across lanes)
      const size_t lanes = (size_t)gridDim.x * ODE_ROWS;
      const size_t gl = (size_t)blockIdx.x * ODE_ROWS + tid;
      T* bufA = act_ws + gl;
      T* bufB = act_ws + (size_t)M.max_dim * lanes + gl;
      const int st = (int)lanes;
      int o = 0;
#pragma unroll
      for (int c = 0; c < 19; ++c) bufA[(size_t)(o + c) * st] = yr[c];
      o += 19;
      if constexpr (NNHIST) {
#pragma unroll
        for (int c = 0; c < 19; ++c) bufA[(size_t)(o + c) * st] = syh[tid * 19 + c];
        o += 19;
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) bufA[(size_t)(o + c) * st] = out[19 + c];
      o += 6;
      if constexpr (NNHIST) {
#pragma unroll
        for (int c = 0; c < 6; ++c) bufA[(size_t)(o + c) * st] = szh[tid * 6 + c];
        o += 6;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) bufA[(size_t)(o + c) * st] = stf[tid * 3 + c];
      const T* res = mlp_lane_eval<T>(M, bufA, bufB, st);
#pragma unroll
      for (int c = 0; c < 25; ++c) out[c] += res[(size_t)c * st];
    }
    __syncthreads();  // everyone is done reading sy before it is reused for the results
#pragma unroll
    for (int c = 0; c < 19; ++c) sy[tid * 19 + c] = out[c];
#pragma unroll
    for (int c = 0; c < 6; ++c) szh[tid * 6 + c] = out[19 + c];
    __syncthreads();
    if (full) {
      TileVec<T, 19>::store(dys, sy, row0, tid);
      TileVec<T, 6>::store(z, szh, row0, tid);
    } else {
      tile_out(dys, sy, row0, Q, 19, tid);
      tile_out(z, szh, row0, Q, 6, tid);
    }
    __syncthreads();
  }
}



// CosseratRod.get_nn_output (cosserat_ode.py:90-112) on Q rows
template <typename T>
__global__ __launch_bounds__(ODE_ROWS) void mlp_eval_kernel(const MlpDev<T> M, int64_t Q, const T* __restrict__ x,
                                                            T* __restrict__ out, T* act_ws) {
  const size_t lanes = (size_t)gridDim.x * ODE_ROWS;
  const size_t gl = (size_t)blockIdx.x * ODE_ROWS + threadIdx.x;
  T* bufA = act_ws + gl;
  T* bufB = act_ws + (size_t)M.max_dim * lanes + gl;
  const int st = (int)lanes;
  const int nin = M.dims[0];
  for (int64_t row0 = (int64_t)blockIdx.x * ODE_ROWS; row0 < Q; row0 += (int64_t)gridDim.x * ODE_ROWS) {
    const int64_t row = row0 + threadIdx.x;
    const int64_t rr = row < Q ? row : Q - 1;
    for (int c = 0; c < nin; ++c) bufA[(size_t)c * st] = x[rr * nin + c];
    const T* res = mlp_lane_eval<T>(M, bufA, bufB, st);
    if (row < Q)
      for (int c = 0; c < 25; ++c) out[row * 25 + c] = res[(size_t)c * st];
  }
}

template <typename T>
int launch_mlp_eval(kr_handle* h, int64_t Q, const T* x, T* out, hipStream_t s) {
  const MlpDev<T>& M = mlpdev<T>(h);
  if (M.n_layers <= 0) {
    set_error("no MLP was set (kr_set_mlp)");
    return KR_E_STATE;
  }
  int64_t tiles = (Q + ODE_ROWS - 1) / ODE_ROWS;
  int grid = (int)(tiles < 2048 ? tiles : 2048);
  int rc = ensure_ws(h, (size_t)2 * M.max_dim * grid * ODE_ROWS * sizeof(T));
  if (rc) return rc;
  hipLaunchKernelGGL((mlp_eval_kernel<T>), dim3(grid), dim3(ODE_ROWS), 0, s, M, Q, x, out, static_cast<T*>(h->ws));
  KR_HIP(hipGetLastError());
  return KR_OK;
}
template int launch_mlp_eval<float>(kr_handle*, int64_t, const float*, float*, hipStream_t);
template int launch_mlp_eval<double>(kr_handle*, int64_t, const double*, double*, hipStream_t);

template <typename T>
int launch_ode_batch(kr_handle* h, int64_t Q, const T* y, const T* yh, const T* zh, const T* tf, T* dys, T* z,
                     int use_nn, hipStream_t s) {
  if (Q <= 0) return KR_OK;
  const RodConst<T>& P = consts<T>(h);
  const MlpDev<T>& M = mlpdev<T>(h);
  int64_t tiles = (Q + ODE_ROWS - 1) / ODE_ROWS;
  int grid = (int)(tiles < 2048 ? tiles : 2048);
  T* act = nullptr;
  if (use_nn) {
    if (M.n_layers <= 0) {
      set_error("use_nn requested but no MLP was set (kr_set_mlp)");
      return KR_E_STATE;
    }
    int rc = ensure_ws(h, (size_t)2 * M.max_dim * grid * ODE_ROWS * sizeof(T));
    if (rc) return rc;
    act = static_cast<T*>(h->ws);
  }
  const bool hist = h->params.nn_input_history != 0;
#define KR_LAUNCH(NNv, Hv)                                                                                  \
  do {                                                                                                      \
    if (P.diag)                                                                                             \
      hipLaunchKernelGGL((ode_batch_kernel<T, true, NNv, Hv>), dim3(grid), dim3(ODE_ROWS), 0, s, P, M, Q, y, yh, zh, \
                         tf, dys, z, act, vec_ok);                                                          \
    else                                                                                                    \
      hipLaunchKernelGGL((ode_batch_kernel<T, false, NNv, Hv>), dim3(grid), dim3(ODE_ROWS), 0, s, P, M, Q, y, yh, zh, \
                         tf, dys, z, act, vec_ok);                                                          \
  } while (0)
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const int vec_ok = al(y) && al(yh) && al(zh) && al(tf) && al(dys) && al(z);
  if (!use_nn) KR_LAUNCH(false, false);
  else if (!hist) KR_LAUNCH(true, false);
  else KR_LAUNCH(true, true);
#undef KR_LAUNCH
  KR_HIP(hipGetLastError());
  return KR_OK;
}
template int launch_ode_batch<float>(kr_handle*, int64_t, const float*, const float*, const float*, const float*,
                                     float*, float*, int, hipStream_t);
template int launch_ode_batch<double>(kr_handle*, int64_t, const double*, const double*, const double*,
                                      const double*, double*, double*, int, hipStream_t);

}  // namespace kr
