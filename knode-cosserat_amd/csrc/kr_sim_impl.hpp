// kr_sim_impl.hpp - forward simulation kernels: one implicit BDF2 time step of a
// batch of rods (the body of knode.simulate, reference knode.py:70-100) and the
// packed-state helpers.
//
// Work decomposition of the step kernel (gfx950, wave64):
//   * one wavefront = 8 rods x 8 "columns"; column 0 integrates the rod from the
//     current base-wrench guess G, columns 1..6 integrate it from G + h_c e_c
//     (forward-difference Jacobian of the shooting residual), column 7 idles as
//     a duplicate of column 0.  All 64 lanes execute the same instruction stream
//     on different data - no divergence inside a sweep.
//   * the spatial sweep (cosserat_ode.py:198-201, N-1 dependent Euler steps) is
//     sequential per lane; the 19-component state stays in VGPRs, every rod
//     parameter in SGPRs (kernel-argument struct), the BDF2 history terms of the
//     wave's 8 rods are staged once per step in LDS (12 values per grid point)
//     and read back as wave-uniform-per-rod ds_read_b128 broadcasts.
//   * after each sweep the 7x6 residuals are exchanged with lane shuffles
//     inside the 8-lane group and every lane solves the same 6x6 system in
//     registers (partial pivoting, fully unrolled).
//   * column 0 streams the swept state to HBM as 16-byte-aligned packed records
//     only once the Newton update has become small (|dG| <= sqrt(tol)), i.e. on
//     the sweep that is expected to be the accepted one.
// Included by kr_sim_f32.hip / kr_sim_f64.hip with KR_SIM_T defined, so the two precisions compile in parallel.
#pragma once
#include "kr_internal.hpp"
#include "mlp_lane.hpp"
#include "mlp_mfma.hpp"
#include "mlp_jvp.hpp"

namespace kr {

constexpr int RPW = 8;   // rods per wave
constexpr int WAVE = 64;

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
template <typename T>
struct Vec16;  // 16-byte vector of T
template <>
struct Vec16<float> {
  typedef float type __attribute__((ext_vector_type(4)));
  static constexpr int n = 4;
};
template <>
struct Vec16<double> {
  typedef double type __attribute__((ext_vector_type(2)));
  static constexpr int n = 2;
};

// packed record <-> registers.  rec[0..24] in slot order q w v u p h n m.
template <typename T>
__device__ __forceinline__ void record_from(const RodState<T>& y, V3<T> v, V3<T> u, T (&rec)[KR_SLOTS]) {
  rec[0] = y.q.x; rec[1] = y.q.y; rec[2] = y.q.z;
  rec[3] = y.w.x; rec[4] = y.w.y; rec[5] = y.w.z;
  rec[6] = v.x; rec[7] = v.y; rec[8] = v.z;
  rec[9] = u.x; rec[10] = u.y; rec[11] = u.z;
  rec[12] = y.p.x; rec[13] = y.p.y; rec[14] = y.p.z;
  rec[15] = y.h0; rec[16] = y.h1; rec[17] = y.h2; rec[18] = y.h3;
  rec[19] = y.n.x; rec[20] = y.n.y; rec[21] = y.n.z;
  rec[22] = y.m.x; rec[23] = y.m.y; rec[24] = y.m.z;
  rec[25] = T(0); rec[26] = T(0); rec[27] = T(0);
}

template <typename T>
__device__ __forceinline__ void store_record(T* dst, const T (&rec)[KR_SLOTS]) {
  using V = typename Vec16<T>::type;
  constexpr int n = Vec16<T>::n;
  V* d = reinterpret_cast<V*>(dst);
#pragma unroll
  for (int c = 0; c < KR_SLOTS / n; ++c) {
    V v;
#pragma unroll
    for (int e = 0; e < n; ++e) v[e] = rec[c * n + e];
    d[c] = v;
  }
}

// n values (a multiple of the 16-byte vector width) to a 16-byte aligned destination
template <typename T, int NV>
__device__ __forceinline__ void store_vec(T* dst, const T (&v)[NV]) {
  using V = typename Vec16<T>::type;
  constexpr int n = Vec16<T>::n;
  static_assert(NV % n == 0, "vector store needs a multiple of 16 bytes");
  V* d = reinterpret_cast<V*>(dst);
#pragma unroll
  for (int c = 0; c < NV / n; ++c) {
    V x;
#pragma unroll
    for (int e = 0; e < n; ++e) x[e] = v[c * n + e];
    d[c] = x;
  }
}

template <typename T, int HS>
__device__ __forceinline__ void load_hist_vec(const T* src, T (&hv)[HS]) {
  if constexpr (HS % Vec16<T>::n == 0) {
    using V = typename Vec16<T>::type;
    constexpr int n = Vec16<T>::n;
    const V* s = reinterpret_cast<const V*>(src);
#pragma unroll
    for (int c = 0; c < HS / n; ++c) {
      V v = s[c];
#pragma unroll
      for (int e = 0; e < n; ++e) hv[c * n + e] = v[e];
    }
  } else {
#pragma unroll
    for (int c = 0; c < HS; ++c) hv[c] = src[c];
  }
}

// History record layout (LDS or global scratch), HS values per grid point:
//   0..2 q_h  3..5 w_h  6..8 v_h  9..11 u_h  12..14 av  15..17 au
//   HS == HS_PHYS (20): 18..19 padding
//   HS == HS_NNH  (32): 18..30 history of p h n m (MLP input when nn_input_history), 31 padding
constexpr int HS_PHYS = 20;
constexpr int HS_PHYS64 = 18;  // fp64: 18 doubles are already a multiple of 16 bytes, no padding
constexpr int HS_NNH = 32;
template <typename T>
constexpr int hs_phys() { return sizeof(T) == 8 ? HS_PHYS64 : HS_PHYS; }

template <typename T, int HS>
__device__ __forceinline__ RodHist<T> hist_from(const T (&hv)[HS]) {
  RodHist<T> h;
  h.qh = {hv[0], hv[1], hv[2]};
  h.wh = {hv[3], hv[4], hv[5]};
  h.vh = {hv[6], hv[7], hv[8]};
  h.uh = {hv[9], hv[10], hv[11]};
  h.av = {hv[12], hv[13], hv[14]};
  h.au = {hv[15], hv[16], hv[17]};
  return h;
}

// history record of one grid point from the packed states of the two previous
// time levels (knode.py:74-75): raw = hc1*cur + hc2*prev, then av / au.
template <typename T, int HS>
__device__ __forceinline__ void build_hist_vals(const RodConst<T>& P, T hc1, T hc2, const T* __restrict__ c,
                                                const T* __restrict__ p, T (&hv)[HS]) {
  constexpr int NR = (HS == HS_NNH) ? 25 : 12;
  constexpr int NRP = (NR + Vec16<T>::n - 1) / Vec16<T>::n * Vec16<T>::n;  // stays inside the 28-slot record
  T cv[NRP], pv[NRP];
  load_hist_vec<T, NRP>(c, cv);
  load_hist_vec<T, NRP>(p, pv);
  T raw[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) raw[k] = hc1 * cv[k] + hc2 * pv[k];
#pragma unroll
  for (int k = 0; k < 12; ++k) hv[k] = raw[k];
  RodHist<T> h;
  h.qh = {raw[0], raw[1], raw[2]};
  h.wh = {raw[3], raw[4], raw[5]};
  h.vh = {raw[6], raw[7], raw[8]};
  h.uh = {raw[9], raw[10], raw[11]};
  hist_derive(P, h);
  hv[12] = h.av.x; hv[13] = h.av.y; hv[14] = h.av.z;
  hv[15] = h.au.x; hv[16] = h.au.y; hv[17] = h.au.z;
  if constexpr (HS == HS_NNH) {
#pragma unroll
    for (int k = 0; k < 13; ++k) hv[18 + k] = raw[12 + k];
    hv[31] = T(0);
  } else if constexpr (HS > 18) {
    hv[18] = T(0);
    hv[19] = T(0);
  }
}
template <typename T, int HS>
__device__ __forceinline__ void build_hist_point(const RodConst<T>& P, T hc1, T hc2, const T* __restrict__ c,
                                                 const T* __restrict__ p, T* dst) {
  T hv[HS];
  build_hist_vals<T, HS>(P, hc1, hc2, c, p, hv);
  using V = typename Vec16<T>::type;
  constexpr int n = Vec16<T>::n;
  V* d = reinterpret_cast<V*>(dst);
#pragma unroll
  for (int k = 0; k < HS / n; ++k) {
    V v;
#pragma unroll
    for (int e = 0; e < n; ++e) v[e] = hv[k * n + e];
    d[k] = v;
  }
}

// Will the next Newton sweep be the accepted one?  dn / dn_prev are the scaled update norms of
// this and the previous iteration.  Quadratic contraction |d_{k+1}| ~ kappa |d_k|^2, kappa from the
// last two updates (capped at 1 = the conservative sqrt(tol) rule when no estimate exists yet).
template <typename T>
__device__ __forceinline__ bool predict_final(T dn, T dn_prev, T tol, T tolA) {
  if (dn <= tolA) return true;
  if (!(dn_prev > T(0)) || !(dn < dn_prev)) return false;
  T kappa = dn * fast_rcp(dn_prev * dn_prev);  // an overflowing or NaN ratio ends up clamped below
  kappa = fmin(fmax(kappa, T(1e-3)), T(1));
  return T(4) * kappa * dn * dn <= tol;
}

// 6x6 solve, Gaussian elimination with partial pivoting on static indices.  The row exchange is a real
// branch, not a pair of selects per element: pivots rarely move (the tip-vs-base Jacobian is close to
// block triangular with a unit diagonal), and where all lanes hold the same system the branch is uniform.
template <typename T>
__device__ __forceinline__ void solve6(T (&a)[6][7], T (&x)[6]) {
  T inv[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      if (__builtin_expect(fabs(a[i][k]) > fabs(a[k][k]), 0)) {
#pragma unroll
        for (int c = k; c < 7; ++c) {
          const T t = a[k][c];
          a[k][c] = a[i][c];
          a[i][c] = t;
        }
      }
    }
    // reciprocal of the pivot by the hardware estimate + Newton steps (full precision for a normal
    // pivot; a zero pivot gives inf/nan, which the callers report as KR_ST_NONFINITE)
    inv[k] = fast_rcp(a[k][k]);
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      const T f = a[i][k] * inv[k];
#pragma unroll
      for (int c = k + 1; c < 7; ++c) a[i][c] = fma(-f, a[k][c], a[i][c]);
    }
  }
#pragma unroll
  for (int k = 5; k >= 0; --k) {
    T s = a[k][6];
#pragma unroll
    for (int c = k + 1; c < 6; ++c) s = fma(-a[k][c], x[c], s);
    x[k] = s * inv[k];
  }
}

// ---------------------------------------------------------------------------
// MLP correction inside a sweep (cosserat_ode.py:169-184)
// ---------------------------------------------------------------------------
// role of a lane in a multiple-shooting sweep, for the base + JVP evaluator (mlp_jvp.hpp); jvp = false elsewhere
struct NnRole {
  int iv = 0, col = 0;
  bool idle = false, jvp = false;
  int zrow = -1;  // dx row a lane without a column zeroes (-1: the layout of the one-wavefront kernels)
  int xrow = -1;  // dx row of a lane with a column, when it is not 16 iv + col - 1 (p columns)
  int ptab = 0;   // wavefront-uniform: where the p columns' samples sit (mlp_jvp.hpp, jvp_scale_pack; 0: there are none)
  bool lowp = false;  // wavefront-uniform: fp64 sweep whose network evaluations may run the fp32 base chain (mlp_jvp_eval_lowp)
  bool base_only = false;  // wavefront-uniform: a sweep that needs no forward-difference columns (mlp_jvp_eval_base)
};
// JVP_ONLY: the caller guarantees M.mfma_ok && M.jvp_ok and a lane role (kr_msw_impl.hpp with the MLP on) - no other
// evaluator is compiled in, so that the register limit of a two-wavefronts-per-SIMD kernel holds for all its callees
template <typename T, int HS, int VAR = 0, bool JVP_ONLY = false>
__device__ __forceinline__ void nn_correct(const MlpDev<T>& M, T* bufA, T* bufB, int stride, T* tile, int lane,
                                           const NnRole& role, const RodState<T>& y, const T (&hv)[HS], V3<T> tf,
                                           RodState<T>& ys, V3<T>& v, V3<T>& u) {
  T yr[19];
  state_to_rows(y, yr);
  if constexpr (HS != HS_NNH) {
    if (JVP_ONLY || M.mfma_ok) {  // wave-uniform: the 64 evaluations of the wave as GEMMs on the matrix cores
      T x[MM_IN];
#pragma unroll
      for (int i = 0; i < 19; ++i) x[i] = yr[i];
      x[19] = v.x; x[20] = v.y; x[21] = v.z; x[22] = u.x; x[23] = u.y; x[24] = u.z;
      x[25] = tf.x; x[26] = tf.y; x[27] = tf.z;
      T d[25];
      if constexpr (JVP_ONLY) {
        mlp_jvp_eval<T, VAR>(M, x, tile, lane, role.iv, role.col, role.idle, role.zrow, d, role.xrow, role.ptab, role.lowp, role.base_only);
      } else {
        if (role.jvp && M.jvp_ok) mlp_jvp_eval<T, VAR>(M, x, tile, lane, role.iv, role.col, role.idle, role.zrow, d, role.xrow, role.ptab, role.lowp, role.base_only);  // wave-uniform choice
        else mlp_mfma_eval<T>(M, x, tile, lane, d);
      }
      T yr2[19];
      state_to_rows(ys, yr2);
#pragma unroll
      for (int i = 0; i < 19; ++i) yr2[i] += d[i];
      ys = rows_to_state(yr2);
      v = {v.x + d[19], v.y + d[20], v.z + d[21]};
      u = {u.x + d[22], u.y + d[23], u.z + d[24]};
      return;
    }
  }
  int o = 0;
#pragma unroll
  for (int i = 0; i < 19; ++i) bufA[(o + i) * stride] = yr[i];
  o += 19;
  if constexpr (HS == HS_NNH) {
    // yh in reference row order p h n m q w  <- record entries 18..30, 0..5
#pragma unroll
    for (int i = 0; i < 13; ++i) bufA[(o + i) * stride] = hv[18 + i];
#pragma unroll
    for (int i = 0; i < 6; ++i) bufA[(o + 13 + i) * stride] = hv[i];
    o += 19;
  }
  bufA[(o + 0) * stride] = v.x; bufA[(o + 1) * stride] = v.y; bufA[(o + 2) * stride] = v.z;
  bufA[(o + 3) * stride] = u.x; bufA[(o + 4) * stride] = u.y; bufA[(o + 5) * stride] = u.z;
  o += 6;
  if constexpr (HS == HS_NNH) {
#pragma unroll
    for (int i = 0; i < 6; ++i) bufA[(o + i) * stride] = hv[6 + i];
    o += 6;
  }
  bufA[(o + 0) * stride] = tf.x; bufA[(o + 1) * stride] = tf.y; bufA[(o + 2) * stride] = tf.z;
  const T* out = mlp_lane_eval<T>(M, bufA, bufB, stride);
  T d[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) d[i] = out[i * stride];
  T yr2[19];
  state_to_rows(ys, yr2);
#pragma unroll
  for (int i = 0; i < 19; ++i) yr2[i] += d[i];
  ys = rows_to_state(yr2);
  v = {v.x + d[19], v.y + d[20], v.z + d[21]};
  u = {u.x + d[22], u.y + d[23], u.z + d[24]};
}

// ---------------------------------------------------------------------------
// the time-step kernel
// ---------------------------------------------------------------------------
template <typename T, int HS>
struct SweepCtx {
  const T* hbase;   // history of this lane's rod: [N][HS], LDS or global
  T* bufA;          // MLP activation columns of this lane (per-lane evaluator)
  T* bufB;
  int astride;
  T* tile;          // LDS exchange tile of the wave (matrix-core evaluator)
  int lane;
  NnRole role;
  V3<T> tf, fconst;
};

// one ODE evaluation incl. the optional network correction
template <typename T, bool DIAG, bool NN, int HS, bool JVP_ONLY = false>
__device__ __forceinline__ void eval_point(const RodConst<T>& P, const MlpDev<T>& M, const SweepCtx<T, HS>& C,
                                           const RodState<T>& y, const T (&hv)[HS], RodState<T>& k, V3<T>& v,
                                           V3<T>& u) {
  ode_eval<T, DIAG>(P, y, hist_from<T, HS>(hv), C.fconst, k, v, u);
  if constexpr (NN) nn_correct<T, HS, 0, JVP_ONLY>(M, C.bufA, C.bufB, C.astride, C.tile, C.lane, C.role, y, hv, C.tf, k, v, u);
}

template <typename T, bool DIAG, int SCHEME, bool HIST_LDS, bool NN, int HS>
__global__ __launch_bounds__(WAVE) void step_kernel(const RodConst<T> P, const StepArgs<T> A, const MlpDev<T> M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);

  const int lane = threadIdx.x;
  const int rl = lane >> 3;
  const int col = lane & 7;
  const int N = P.N;
  const int64_t rod_raw = (int64_t)blockIdx.x * RPW + rl;
  const bool valid = rod_raw < A.B;
  const int64_t rod = valid ? rod_raw : A.B - 1;
  const size_t rod_elems = (size_t)N * KR_SLOTS;

  SweepCtx<T, HS> C;
  // ---- BDF2 history terms (knode.py:74-75) --------------------------------
  if constexpr (HIST_LDS) {
    for (int pt = lane; pt < RPW * N; pt += WAVE) {
      const int r = pt / N;
      const int j = pt - r * N;
      int64_t rr = (int64_t)blockIdx.x * RPW + r;
      if (rr >= A.B) rr = A.B - 1;
      const size_t off = rr * rod_elems + (size_t)j * KR_SLOTS;
      build_hist_point<T, HS>(P, A.hc1, A.hc2, A.cur + off, A.prev + off, smem + (size_t)pt * HS);
    }
    __syncthreads();
    C.hbase = smem + (size_t)rl * N * HS;
  } else {
    C.hbase = A.hist_ws + rod * (size_t)N * HS;
  }

  // MLP activation columns: LDS when the launcher found room, else global scratch
  C.bufA = nullptr;
  C.bufB = nullptr;
  C.astride = 0;
  C.tile = nullptr;
  C.lane = lane;
  if constexpr (NN) {
    if (M.mfma_ok && HS != HS_NNH) {
      C.tile = smem + (HIST_LDS ? (size_t)RPW * N * HS : 0);
    } else if (A.act_ws == nullptr) {
      T* a0 = smem + (HIST_LDS ? (size_t)RPW * N * HS : 0);
      C.bufA = a0 + lane;
      C.bufB = a0 + (size_t)M.max_dim * WAVE + lane;
      C.astride = WAVE;
    } else {
      const size_t lanes = (size_t)gridDim.x * WAVE;
      const size_t gl = (size_t)blockIdx.x * WAVE + lane;
      C.bufA = A.act_ws + gl;
      C.bufB = A.act_ws + (size_t)M.max_dim * lanes + gl;
      C.astride = (int)lanes;
    }
  }

  // ---- per-rod inputs ------------------------------------------------------
  T G[6], G0[6];  // G0: the caller's guess (knode.py:89 warm start), where the damped second phase starts over
#pragma unroll
  for (int k = 0; k < 6; ++k) G0[k] = G[k] = A.G[rod * 6 + k];
  if (A.pred_order > 0) {
    // G of earlier steps = n, m at the base of the stored states; extrapolate in time
    const size_t o0 = rod * rod_elems + SL_N;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const T g0 = A.cur[o0 + k], g1 = A.prev[o0 + k];
      const T g2 = A.pred_order > 1 ? A.prev2[o0 + k] : T(0);
      G[k] = A.pred_order > 1 ? T(3) * (g0 - g1) + g2 : T(2) * g0 - g1;
    }
  }
  C.tf = {T(0), T(0), T(0)};
#pragma unroll
  for (int t = 0; t < 4; ++t) {  // cosserat_ode.py:195
    const T tt = A.tens[rod * A.tens_stride + t];
    C.tf.x += tt * P.tdirs[t * 3 + 0];
    C.tf.y += tt * P.tdirs[t * 3 + 1];
    C.tf.z += tt * P.tdirs[t * 3 + 2];
  }
  C.fconst = {P.rhoAg[0] + C.tf.x, P.rhoAg[1] + C.tf.y, P.rhoAg[2] + C.tf.z};
  // z of the last grid point is never touched by a sweep (cosserat_ode.py:198-201)
  V3<T> vlast, ulast;
  {
    const T* cl = A.cur + rod * rod_elems + (size_t)(N - 1) * KR_SLOTS;
    vlast = {cl[SL_V], cl[SL_V + 1], cl[SL_V + 2]};
    ulast = {cl[SL_U], cl[SL_U + 1], cl[SL_U + 2]};
  }
  T* out_rod = A.next + rod * rod_elems;

  // done: Newton finished for this rod; storing: column 0 streams the state out
  // during the sweep; stored: the state in HBM belongs to the current G;
  // flush: last pass, only for rods that stopped without a stored sweep.
  bool done = false, storing = (A.mode == 1), stored = false, flush = false;
  int status = KR_ST_MAXIT;
  int it = 0;
  T dn_prev = T(-1);
  // Second phase for rods the plain iteration did not solve (the reference's hybrd has a trust region): restart
  // from the caller's guess with backtracking - every update is taken as G - lam d, lam halved until the residual
  // norm has decreased (what the CPU oracle's newton_shoot does).  Same root, same stopping rule.
  bool damped = false, have_trial = false;
  // Sweep counters of THIS rod: `it` counts the plain phase, `itd` the damped one (its own cap, 8 x maxit); a rod
  // that is done stops counting although its lanes keep sweeping with the rest of the wavefront.  A.iters = it + itd.
  bool redo = false;  // this rod is in the damped phase
  int itd = 0;
  int maxit = A.maxit;
  T Gold[6], dsv[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { Gold[k] = G[k]; dsv[k] = T(0); }
  T nr_old = T(-1), lam = T(1);

  while (true) {
    // ---- one spatial sweep, all 64 lanes ----------------------------------
    T hstep[6];
    T Gl[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      hstep[k] = A.fd_eps * fmax(fabs(G[k]), T(1));
      Gl[k] = G[k] + ((col == k + 1) ? hstep[k] : T(0));
    }
    RodState<T> y;
    y.p = {P.p0[0], P.p0[1], P.p0[2]};
    y.h0 = P.h0[0]; y.h1 = P.h0[1]; y.h2 = P.h0[2]; y.h3 = P.h0[3];
    y.n = {Gl[0], Gl[1], Gl[2]};
    y.m = {Gl[3], Gl[4], Gl[5]};
    y.q = {P.q0[0], P.q0[1], P.q0[2]};
    y.w = {P.w0[0], P.w0[1], P.w0[2]};
    const bool st = valid && col == 0 && (flush ? !stored : (storing && !done));

    T hv[HS];
    load_hist_vec<T, HS>(C.hbase, hv);
    for (int j = 0; j < N - 1; ++j) {
      RodState<T> k1;
      V3<T> v, u;
      eval_point<T, DIAG, NN, HS>(P, M, C, y, hv, k1, v, u);
      if (st) {
        T rec[KR_SLOTS];
        record_from(y, v, u, rec);
        store_record(out_rod + (size_t)j * KR_SLOTS, rec);
      }
      if constexpr (SCHEME == KR_EULER) {
        load_hist_vec<T, HS>(C.hbase + (size_t)(j + 1) * HS, hv);  // next point's history (last one unused)
        y = state_axpy(y, P.ds, k1);
      } else {
        // classical RK4, cosserat_ode.py:232-242: stages 2,3 see the midpoint
        // history, stage 4 the history of point j+1
        T hn[HS], hm[HS];
        load_hist_vec<T, HS>(C.hbase + (size_t)(j + 1) * HS, hn);
        if (A.mid) {
          // the caller's own midpoint histories yh_int[:, j], zh_int[:, j] (cosserat_ode.py:225,233-234)
          const T* mp = A.mid + rod * rod_elems + (size_t)j * KR_SLOTS;
          build_hist_vals<T, HS>(P, T(1), T(0), mp, mp, hm);
        } else {
#pragma unroll
          for (int c = 0; c < HS; ++c) hm[c] = T(0.5) * (hv[c] + hn[c]);
        }
        RodState<T> k2, k3, k4;
        V3<T> v2, u2;
        RodState<T> ya = state_axpy(y, P.ds * T(0.5), k1);
        eval_point<T, DIAG, NN, HS>(P, M, C, ya, hm, k2, v2, u2);
        ya = state_axpy(y, P.ds * T(0.5), k2);
        eval_point<T, DIAG, NN, HS>(P, M, C, ya, hm, k3, v2, u2);
        ya = state_axpy(y, P.ds, k3);
        eval_point<T, DIAG, NN, HS>(P, M, C, ya, hn, k4, v2, u2);
        RodState<T> ksum = state_axpy(k1, T(2), k2);
        ksum = state_axpy(ksum, T(2), k3);
        ksum = state_axpy(ksum, T(1), k4);
        y = state_axpy(y, P.ds / T(6), ksum);
#pragma unroll
        for (int c = 0; c < HS; ++c) hv[c] = hn[c];
      }
    }
    if (st) {
      T rec[KR_SLOTS];
      record_from(y, vlast, ulast, rec);
      store_record(out_rod + (size_t)(N - 1) * KR_SLOTS, rec);
      if (A.tip) {
        T* tp = A.tip + rod * A.tip_stride;
        tp[0] = y.p.x; tp[1] = y.p.y; tp[2] = y.p.z;
      }
    }
    if (flush) break;
    if (storing && !done) stored = true;
    T res[6];
    res[0] = P.Ftip[0] - y.n.x; res[1] = P.Ftip[1] - y.n.y; res[2] = P.Ftip[2] - y.n.z;
    res[3] = P.Mtip[0] - y.m.x; res[4] = P.Mtip[1] - y.m.y; res[5] = P.Mtip[2] - y.m.z;
    if (!done && !flush) { if (redo) ++itd; else ++it; }
    const int cnt = redo ? itd : it;  // sweeps of the phase this rod is in, against that phase's cap
    if (A.mode == 1) {
      if (valid && col == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) A.r_out[rod * 6 + k] = res[k];
      }
      return;
    }

    // ---- Newton update on the 6 shooting unknowns -------------------------
    const int gbase = lane & ~7;
    T a[6][7];
#pragma unroll
    for (int k = 0; k < 6; ++k) a[k][6] = __shfl(res[k], gbase, WAVE);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const T ih = T(1) / hstep[c];
#pragma unroll
      for (int k = 0; k < 6; ++k) a[k][c] = (__shfl(res[k], gbase + c + 1, WAVE) - a[k][6]) * ih;
    }
    T nr = T(0);  // squared residual norm at the point just swept (a NaN fails the comparison below)
#pragma unroll
    for (int k = 0; k < 6; ++k) nr = fma(a[k][6], a[k][6], nr);
    T d[6];
    solve6(a, d);
    T dn = T(0), gn = T(1);
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      dn = fmax(dn, fabs(d[k]));
      gn = fmax(gn, fabs(G[k]));
      finite = finite && isfinite(d[k]);
    }
    if (!done) {
      const T keep = T(1) - T(1e-4) * lam;
      if (damped && have_trial && !(nr <= nr_old * keep * keep) && lam > T(1.0 / 1024.0) && cnt < maxit) {
        lam *= T(0.5);  // rejected trial point: shorter step along the same direction
#pragma unroll
        for (int k = 0; k < 6; ++k) G[k] = Gold[k] - lam * dsv[k];
        stored = false;
      } else if (!finite) {
        done = true;
        status = KR_ST_NONFINITE;
      } else if (storing && dn <= A.tol * gn) {
        done = true;  // the state streamed out by this sweep is the accepted one
        status = KR_ST_CONVERGED;
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) { Gold[k] = G[k]; dsv[k] = d[k]; G[k] -= d[k]; }
        nr_old = nr;
        lam = T(1);
        have_trial = damped;
        if (predict_final<T>(dn / gn, dn_prev, A.tol, A.tolA)) storing = true;
        dn_prev = dn / gn;
        stored = false;
        if (cnt >= maxit) {
          done = true;
          status = KR_ST_MAXIT;
        }
      }
    }
    if (__all(done)) {
      if (!damped && __any(valid && status != KR_ST_CONVERGED)) {
        damped = true;  // wave-uniform; only the rods that failed start over
        if (status != KR_ST_CONVERGED) {
          done = false; stored = false; storing = true; have_trial = false;
          status = KR_ST_MAXIT;
          redo = true;
          maxit = 8 * A.maxit;
          dn_prev = T(-1);
#pragma unroll
          for (int k = 0; k < 6; ++k) G[k] = G0[k];
        }
        continue;
      }
      // rods that stopped without a stored sweep (iteration cap / non-finite):
      // one more pass so that state_next is always the sweep of the returned G
      if (__any(!stored)) flush = true;
      else break;
    }
  }

  if (valid && col == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) A.G[rod * 6 + k] = G[k];
    if (A.status) A.status[rod * A.st_stride] = status;
    if (A.iters) A.iters[rod * A.st_stride] = it + itd;
  }
}

// same from register copies of the twelve leading slots (q w v u) of the two time levels
template <typename T, int HS>
__device__ __forceinline__ void build_hist_regs(const RodConst<T>& P, T hc1, T hc2, const T (&cv)[12], const T (&pv)[12],
                                                T* dst) {
  static_assert(HS == HS_PHYS || HS == HS_PHYS64, "the register form carries the physics history only");
  T hv[HS];
#pragma unroll
  for (int k = 0; k < 12; ++k) hv[k] = hc1 * cv[k] + hc2 * pv[k];
  RodHist<T> h;
  h.qh = {hv[0], hv[1], hv[2]};
  h.wh = {hv[3], hv[4], hv[5]};
  h.vh = {hv[6], hv[7], hv[8]};
  h.uh = {hv[9], hv[10], hv[11]};
  hist_derive(P, h);
  hv[12] = h.av.x; hv[13] = h.av.y; hv[14] = h.av.z;
  hv[15] = h.au.x; hv[16] = h.au.y; hv[17] = h.au.z;
  if constexpr (HS > 18) {
    hv[18] = T(0);
    hv[19] = T(0);
  }
  store_vec<T, HS>(dst, hv);
}

// history into global scratch when it does not fit in LDS
template <typename T, int HS>
__global__ void hist_kernel(const RodConst<T> P, T hc1, T hc2, int64_t B, const T* __restrict__ cur,
                            const T* __restrict__ prev, T* __restrict__ hist) {
  const int64_t total = B * (int64_t)P.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    build_hist_point<T, HS>(P, hc1, hc2, cur + i * KR_SLOTS, prev + i * KR_SLOTS, hist + i * HS);
}

template <typename T, bool DIAG, int SCHEME, bool HIST_LDS, bool NN, int HS>
static int launch_step_inst(const RodConst<T>& P, const MlpDev<T>& M, const StepArgs<T>& a, size_t smem,
                            hipStream_t s) {
  auto kern = step_kernel<T, DIAG, SCHEME, HIST_LDS, NN, HS>;
  if (int rc_lds_ = dyn_lds(reinterpret_cast<const void*>(kern), smem)) return rc_lds_;
  const int grid = (int)((a.B + RPW - 1) / RPW);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE), smem, s, P, a, M);
  KR_HIP(hipGetLastError());
  return KR_OK;
}

template <typename T, int SCHEME, bool NN, int HS>
static int launch_step_mem(kr_handle* h, StepArgs<T> a, hipStream_t s) {
  const RodConst<T>& P = consts<T>(h);
  const MlpDev<T>& M = mlpdev<T>(h);
  const int N = P.N;
  const int grid = (int)((a.B + RPW - 1) / RPW);
  const size_t hist_lds = (size_t)RPW * N * HS * sizeof(T);
  const bool mfma = NN && M.mfma_ok && HS != HS_NNH;
  const size_t act_lds = !NN ? 0 : mfma ? (size_t)WAVE * MM_TILE_LD * sizeof(T) : (size_t)2 * M.max_dim * WAVE * sizeof(T);
  const size_t limit = (size_t)h->lds_limit;
  bool hist_in_lds = hist_lds <= limit;
  bool act_in_lds = NN && (act_lds + (hist_in_lds ? hist_lds : 0) <= limit);
  if (NN && !act_in_lds && hist_in_lds && act_lds <= limit) {
    // prefer the activations in LDS (touched ~100x more often than the history)
    act_in_lds = true;
    hist_in_lds = false;
  }
  size_t ws_need = 0;
  const size_t hist_bytes = ((size_t)a.B * N * HS * sizeof(T) + 255) & ~size_t(255);
  const size_t act_bytes = (size_t)2 * M.max_dim * grid * WAVE * sizeof(T);
  if (!hist_in_lds) ws_need += hist_bytes;
  if (NN && !act_in_lds) ws_need += act_bytes;
  a.hist_ws = nullptr;
  a.act_ws = nullptr;
  if (ws_need) {
    int rc = ensure_ws(h, ws_need);
    if (rc) return rc;
    unsigned char* w = static_cast<unsigned char*>(h->ws);
    if (!hist_in_lds) {
      a.hist_ws = reinterpret_cast<T*>(w);
      w += hist_bytes;
      hipLaunchKernelGGL((hist_kernel<T, HS>), dim3(1024), dim3(256), 0, s, P, a.hc1, a.hc2, a.B, a.cur, a.prev, a.hist_ws);
      KR_HIP(hipGetLastError());
    }
    if (NN && !act_in_lds) a.act_ws = reinterpret_cast<T*>(w);
  }
  const size_t smem = (hist_in_lds ? hist_lds : 0) + (act_in_lds ? act_lds : 0);
  if (P.diag) {
    if (hist_in_lds) return launch_step_inst<T, true, SCHEME, true, NN, HS>(P, M, a, smem, s);
    return launch_step_inst<T, true, SCHEME, false, NN, HS>(P, M, a, smem, s);
  }
  if (hist_in_lds) return launch_step_inst<T, false, SCHEME, true, NN, HS>(P, M, a, smem, s);
  return launch_step_inst<T, false, SCHEME, false, NN, HS>(P, M, a, smem, s);
}

template <typename T, int SCHEME>
static int launch_step_nn(kr_handle* h, int use_nn, const StepArgs<T>& a, hipStream_t s) {
  if (!use_nn) return launch_step_mem<T, SCHEME, false, hs_phys<T>()>(h, a, s);
  if (mlpdev<T>(h).n_layers <= 0) {
    set_error("use_nn requested but no MLP was set (kr_set_mlp)");
    return KR_E_STATE;
  }
  if (h->params.nn_input_history) return launch_step_mem<T, SCHEME, true, HS_NNH>(h, a, s);
  return launch_step_mem<T, SCHEME, true, hs_phys<T>()>(h, a, s);
}

template <typename T>
static bool ms_eligible(kr_handle* h, int use_nn, const StepArgs<T>& a);
template <typename T>
static int launch_ms(kr_handle* h, int scheme, int use_nn, const StepArgs<T>& a, hipStream_t s);

template <typename T>
static int launch_msw(kr_handle* h, int W, const StepArgs<T>& a, hipStream_t s);

template <typename T>
int launch_step(kr_handle* h, int scheme, int use_nn, const StepArgs<T>& a, hipStream_t s) {
  h->last_waves_per_rod = 1;
  if (const int W = step_waves_per_rod<T>(h, scheme, use_nn, a.B, a.mode)) return launch_msw<T>(h, W, a, s);
  if (ms_eligible<T>(h, use_nn, a)) return launch_ms<T>(h, scheme, use_nn, a, s);
  if (scheme == KR_EULER) return launch_step_nn<T, KR_EULER>(h, use_nn, a, s);
  if (scheme == KR_RK4) return launch_step_nn<T, KR_RK4>(h, use_nn, a, s);
  set_error("unknown scheme");
  return KR_E_ARG;
}

// ---------------------------------------------------------------------------
// packed-state helpers
// ---------------------------------------------------------------------------
// reference row (0..24 of [y;z]) -> packed slot
__device__ __forceinline__ int slot_of_row(int r) {
  // rows: p0-2 h3-6 n7-9 m10-12 q13-15 w16-18 v19-21 u22-24
  if (r < 13) return SL_P + r;           // p h n m are contiguous in both orders
  if (r < 19) return SL_Q + (r - 13);    // q w
  return SL_V + (r - 19);                // v u
}

template <typename T>
__global__ void init_straight_kernel(const RodConst<T> P, double L, int64_t B, T* __restrict__ state) {
  const int N = P.N;
  const int64_t pts = B * (int64_t)N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pts; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % N);
    T rec[KR_SLOTS];
#pragma unroll
    for (int k = 0; k < KR_SLOTS; ++k) rec[k] = T(0);
    rec[SL_P + 2] = (T)((double)j * (L / (double)(N - 1)));  // np.linspace(0, L, N), knode.py:59
    if (j == N - 1) rec[SL_P + 2] = (T)L;
    rec[SL_H] = T(1);
    rec[SL_V + 2] = T(1);
    store_record(state + i * KR_SLOTS, rec);
  }
}

template <typename T>
__global__ void pack_kernel(int N, int64_t B, const T* __restrict__ y_fm, const T* __restrict__ z_fm,
                            T* __restrict__ state) {
  // thread per (rod, row, j): coalesced reads along j
  const int64_t total = B * 25 * (int64_t)N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % N);
    const int64_t t = i / N;
    const int r = (int)(t % 25);
    const int64_t b = t / 25;
    const T val = r < 19 ? y_fm[(b * 19 + r) * N + j] : z_fm[(b * 6 + (r - 19)) * N + j];
    state[(b * N + j) * KR_SLOTS + slot_of_row(r)] = val;
    if (r < 3) state[(b * N + j) * KR_SLOTS + 25 + r] = T(0);
  }
}

template <typename T>
__global__ void unpack_kernel(int N, int64_t B, const T* __restrict__ state, T* __restrict__ y_fm,
                              T* __restrict__ z_fm) {
  const int64_t total = B * 25 * (int64_t)N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % N);
    const int64_t t = i / N;
    const int r = (int)(t % 25);
    const int64_t b = t / 25;
    const T val = state[(b * N + j) * KR_SLOTS + slot_of_row(r)];
    if (r < 19) y_fm[(b * 19 + r) * N + j] = val;
    else z_fm[(b * 6 + (r - 19)) * N + j] = val;
  }
}

// out[b][50][N] = [y; z; yh; zh] with yh = c1*state_m1 + c2*state_m2 (knode.py:74-75,96)
template <typename T>
__global__ void unpack50_kernel(const RodConst<T> P, int64_t B, const T* __restrict__ st, const T* __restrict__ m1,
                                const T* __restrict__ m2, T* __restrict__ out) {
  const int N = P.N;
  const int64_t total = B * 50 * (int64_t)N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % N);
    const int64_t t = i / N;
    const int r = (int)(t % 50);
    const int64_t b = t / 50;
    const int64_t o = (b * N + j) * KR_SLOTS;
    T val;
    if (r < 25) val = st[o + slot_of_row(r)];
    else {
      const int s = slot_of_row(r - 25);
      val = P.c1 * m1[o + s] + P.c2 * m2[o + s];
    }
    out[i] = val;
  }
}

template <typename T>
__global__ void tip_kernel(int N, int64_t B, const T* __restrict__ state, T* __restrict__ tip) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * 3) {
    const int64_t b = i / 3;
    const int c = (int)(i % 3);
    tip[i] = state[(b * N + (N - 1)) * KR_SLOTS + SL_P + c];
  }
}

static inline int grid_for(int64_t n, int block = 256) {
  int64_t g = (n + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

template <typename T>
int launch_init_straight(kr_handle* h, int64_t B, T* state, hipStream_t s) {
  const RodConst<T>& P = consts<T>(h);
  hipLaunchKernelGGL((init_straight_kernel<T>), dim3(grid_for(B * P.N)), dim3(256), 0, s, P, h->params.L, B, state);
  KR_HIP(hipGetLastError());
  return KR_OK;
}
template <typename T>
int launch_pack(kr_handle* h, int64_t B, const T* y_fm, const T* z_fm, T* state, hipStream_t s) {
  const int N = h->params.N;
  hipLaunchKernelGGL((pack_kernel<T>), dim3(grid_for(B * 25 * N)), dim3(256), 0, s, N, B, y_fm, z_fm, state);
  KR_HIP(hipGetLastError());
  return KR_OK;
}
template <typename T>
int launch_unpack(kr_handle* h, int64_t B, const T* state, T* y_fm, T* z_fm, hipStream_t s) {
  const int N = h->params.N;
  hipLaunchKernelGGL((unpack_kernel<T>), dim3(grid_for(B * 25 * N)), dim3(256), 0, s, N, B, state, y_fm, z_fm);
  KR_HIP(hipGetLastError());
  return KR_OK;
}
template <typename T>
int launch_unpack50(kr_handle* h, int64_t B, const T* st, const T* m1, const T* m2, T* out, hipStream_t s) {
  const RodConst<T>& P = consts<T>(h);
  hipLaunchKernelGGL((unpack50_kernel<T>), dim3(grid_for(B * 50 * P.N)), dim3(256), 0, s, P, B, st, m1, m2, out);
  KR_HIP(hipGetLastError());
  return KR_OK;
}
template <typename T>
int launch_tip(kr_handle* h, int64_t B, const T* state, T* tip, hipStream_t s) {
  const int N = h->params.N;
  hipLaunchKernelGGL((tip_kernel<T>), dim3(grid_for(B * 3)), dim3(256), 0, s, N, B, state, tip);
  KR_HIP(hipGetLastError());
  return KR_OK;
}

#define KR_INST(T)                                                                              \
  template int launch_step<T>(kr_handle*, int, int, const StepArgs<T>&, hipStream_t);            \
  template int launch_init_straight<T>(kr_handle*, int64_t, T*, hipStream_t);                   \
  template int launch_pack<T>(kr_handle*, int64_t, const T*, const T*, T*, hipStream_t);        \
  template int launch_unpack<T>(kr_handle*, int64_t, const T*, T*, T*, hipStream_t);            \
  template int launch_unpack50<T>(kr_handle*, int64_t, const T*, const T*, const T*, T*, hipStream_t); \
  template int launch_tip<T>(kr_handle*, int64_t, const T*, T*, hipStream_t);

}  // namespace kr
