// kr_ms_impl.hpp - multiple-shooting form of the implicit time step.
//
// Why: at the BASELINE batch (1024 rods) the single-shooting kernel keeps only
// 128 of the chip's 1024 SIMDs busy, each with a dependent chain of
// (Newton sweeps) x (N-1 grid points).  Here ONE rod owns a whole wavefront:
// the rod is cut into P = 4 sub-intervals that are integrated concurrently,
//     interval 0 :  1 + 6  lanes  (unknowns: base wrench G, like single shooting)
//     interval i :  1 + 16 lanes  (unknowns: the state at the interval's first grid
//                                  point; p has an identity Jacobian and needs no lane)
// 7 + 3*17 = 58 lanes, chain length (N-1)/4 per sweep, 1024 wavefronts at B=1024.
// The discrete equations are the reference's (cosserat_ode.py:198-201 applied on
// every segment, boundary conditions of :194 and :204-207); only the way the
// nonlinear system is solved differs: Newton on (G, Y_1..Y_{P-1}) with the
// continuity conditions  E_i(Y_i) = Y_{i+1}  condensed onto the 6 base unknowns:
//     dY_1 = c_0 + A_0 dG,  dY_{i+1} = c_i + A_i dY_i   (c_i = E_i - Y_{i+1}, A_i = dE_i/dY_i by
//     forward differences),  [n;m]-rows of A_{P-1} dY_{P-1} = tip residual  ->  6x6 solve.
// At acceptance every |dG|, |dY_i| is <= tol, i.e. the stored trajectory satisfies
// the sweep equations with interface jumps below the solver tolerance - the same
// accuracy statement as single shooting.
//
// The initial guess comes from extrapolating the previous states in time
// (order 0/1/2), which saves one Newton sweep per step on smooth inputs.
#pragma once
#include <type_traits>
#include "kr_sim_impl.hpp"

namespace kr {

#ifndef KR_MS_UNROLL
#define KR_MS_UNROLL 2  // grid points per trip of the non-storing sweep loop (lets the scheduler overlap neighbours)
#endif
constexpr int kMsUnroll = KR_MS_UNROLL;  // (a constant, not the macro, in the pragma: -save-temps compiles the preprocessed text)
constexpr int MS_P = 4;
constexpr int MS_WPB = 4;  // wavefronts (= rods) per workgroup: the CU puts the 4 waves of a workgroup on its 4 SIMDs

// Rods never interact, so all LDS hand-offs are between lanes of ONE wavefront: order the LDS
// traffic (s_waitcnt via the workgroup-scope fence) and stop the compiler from moving code across,
// but do not execute s_barrier - the waves of a workgroup run different numbers of Newton sweeps.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
// two adjacent values at an even element offset of a 16-byte aligned array: one LDS access
template <typename T>
__device__ __forceinline__ void store_pair(T* dst, T a, T b) {
  typedef T V2 __attribute__((ext_vector_type(2)));
  V2 v;
  v[0] = a; v[1] = b;
  *reinterpret_cast<V2*>(dst) = v;
}
// value of the lane whose index differs by the quad permutation CTRL (0xB1: ^1, 0x4E: ^2), by DPP
template <int CTRL>
__device__ __forceinline__ float quad_xor(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double quad_xor(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xFFFFFFFFll), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// |u| / max(|x|, 1) in fp32 (hardware reciprocal); +inf for a NaN or infinite update
template <typename T>
__device__ __forceinline__ float update_ratio(T u, T x) {
  const float q = fabsf((float)u) * __builtin_amdgcn_rcpf(fmaxf(fabsf((float)x), 1.0f));
  return q <= 3.0e38f ? q : __builtin_inff();
}
// maximum of a non-negative value over the 64 lanes (all active), DPP within rows of 16 then 4 readlanes
__device__ __forceinline__ float wave_max_nonneg(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), decltype(ctrl)::value, 0xF, 0xF, true));
  };
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0xB1>{}));   // quad: lanes ^1
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x4E>{}));   // quad: lanes ^2
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x141>{}));  // row_half_mirror
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x140>{}));  // row_mirror
  const int b = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// sum over the 64 lanes (all active) of a double: DPP within rows of 16, then 4 x 2 readlanes
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += quad_xor<0xB1>(v);
  v += quad_xor<0x4E>(v);
  v += quad_xor<0x141>(v);  // row_half_mirror
  v += quad_xor<0x140>(v);  // row_mirror
  const long long b = __double_as_longlong(v);
  const int lo = (int)(b & 0xFFFFFFFFll), hi = (int)(b >> 32);
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int l = __builtin_amdgcn_readlane(lo, 16 * r), h = __builtin_amdgcn_readlane(hi, 16 * r);
    s += __longlong_as_double(((long long)h << 32) | (unsigned int)l);
  }
  return s;
}
// value of lane SRC in every lane
template <int SRC>
__device__ __forceinline__ float lane_bcast(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), SRC)); }
template <int SRC>
__device__ __forceinline__ double lane_bcast(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), SRC), hi = __builtin_amdgcn_readlane((int)(b >> 32), SRC);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <typename T>
__device__ __forceinline__ void load_pair(const T* src, T& a, T& b) {
  typedef T V2 __attribute__((ext_vector_type(2)));
  const V2 v = *reinterpret_cast<const V2*>(src);
  a = v[0]; b = v[1];
}
constexpr int MS_YP = 19;  // length of a state vector in LDS (row order p h n m q w)
constexpr int MS_NCOL = 6 + 16 * (MS_P - 1);  // Jacobian columns of all intervals, packed
__device__ __forceinline__ int ms_col0(int iv) { return iv == 0 ? 0 : 6 + 16 * (iv - 1); }

// reference row order r (0..18 = p h n m q w) of the packed slots
__device__ __forceinline__ int ms_slot_of_yrow(int r) { return r < 13 ? SL_P + r : SL_Q + (r - 13); }

template <typename T>
__device__ __forceinline__ T extrapolate(int order, T g0, T g1, T g2) {
  if (order <= 0) return g0;
  if (order == 1) return T(2) * g0 - g1;
  return T(3) * (g0 - g1) + g2;
}

constexpr int MS_ORDER_LP = 8;  // "order" code of the adaptive linear predictor (see ms_sim_kernel)
constexpr int MS_HLEV = 8;  // time levels of the unknowns the persistent kernel keeps (predictor order <= 7)
// polynomial extrapolation one step ahead from the newest `order + 1` levels h[0] (newest) .. h[order]:
// sum_k (-1)^k C(order+1, k+1) h[k]
template <typename T>
__device__ __forceinline__ T extrapolate_n(int order, const T (&h)[MS_HLEV]) {
  switch (order) {
    case 0: return h[0];
    case 1: return T(2) * h[0] - h[1];
    case 2: return T(3) * (h[0] - h[1]) + h[2];
    case 3: return T(4) * (h[0] + h[2]) - T(6) * h[1] - h[3];
    case 4: return T(5) * (h[0] - h[3]) + T(10) * (h[2] - h[1]) + h[4];
    case 5: return T(6) * (h[0] + h[4]) - T(15) * (h[1] + h[3]) + T(20) * h[2] - h[5];
    case 6: return T(7) * (h[0] - h[5]) + T(21) * (h[4] - h[1]) + T(35) * (h[2] - h[3]) + h[6];
    default: return T(8) * (h[0] + h[6]) - T(28) * (h[1] + h[5]) + T(56) * (h[2] + h[4]) - T(70) * h[3] - h[7];
  }
}

enum : int {
  CD_P0 = 0, CD_H0 = 3, CD_Q0 = 7, CD_W0 = 10, CD_FTIP = 13, CD_MTIP = 16, CD_TDIRS = 19, CD_RHOAG = 31,
  CD_KSEI = 34, CD_KSEV = 43, CD_BSE = 46, CD_KBTI = 55, CD_BBT = 64, CD_SIZE = 76
};
// LDS elements of one rod (a multiple of 4, so every rod's slice stays 16-byte aligned)
constexpr int MS_BO_PAD = 48;  // 304 (XB) + 48 (Tm) + 48 = 400 = 112 input-row + 288 hand-off elements
constexpr int MS_BP_ELEMS = 3 * 3 * 19 + 1 + 12;  // [interval 1..3][p direction][19 rows], padding, [3][3] p updates (+3)
template <typename T, int HS>
__host__ __device__ inline size_t ms_lds_elems(int N, bool persist, bool nn = false) {
  // XB, Tm, Es are contiguous; with the MLP on the same region doubles as the 64 x 32 exchange
  // tile of mlp_mfma.hpp.  The cold table sits in front of XB so the tile cannot clobber it.
  // (MS_BO_PAD: with the MLP on, XB + Tm + this pad hold the scratch of a BASE-ONLY network evaluation - 112 + 288 elements -
  //  so that such a sweep leaves the forward-difference columns in Es alone: ms_newton, "base-only storing sweeps")
  size_t alg = 2 * MS_YP * 8 + 48 + (nn ? MS_BO_PAD : 0) + ((WAVE * MS_YP + 3) & ~3);
  if (nn && alg < mj_scratch_elems<T>()) alg = mj_scratch_elems<T>();  // scratch of the base + JVP evaluator
  alg = (alg + 3) & ~size_t(3);
  size_t n = (size_t)N * HS + ((MS_P * MS_YP + 3) & ~3) + ((CD_SIZE + 3) & ~3) + 40 + alg;  // 40: Ti (6 x 6 inverse)
  if (persist) n += (size_t)N * 12;
  if (persist && nn) n = ((n + 3) & ~size_t(3)) + MS_BP_ELEMS;  // p-column blocks B_g (ms_newton)
  return (n + 3) & ~size_t(3);
}

#ifdef KR_MS_STAMPS
#define KR_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); (var) = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#define KR_STAMP_ADD(acc, t0) do { unsigned long long _t; KR_STAMP(_t); (acc) += _t - (t0); (t0) = _t; } while (0)
struct MsStamps { unsigned long long sweep = 0, alg = 0, prep = 0, total = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, osum = 0, retries = 0; int its = 0, olast = 0; float em[8] = {0, 0, 0, 0, 0, 0, 0, 0}; double dn[4] = {0, 0, 0, 0}; double qa = 0, qn = 0; };
#else
#define KR_STAMP(var) do { } while (0)
#define KR_STAMP_ADD(acc, t0) do { } while (0)
struct MsStamps { };
#endif

// Per-step ("cold") rod parameters live in a small LDS table so that only the ~15 doubles the sweep
// itself needs occupy scalar registers (keeping all of RodConst live across the time-step loop
// made the compiler spill SGPRs to VGPR lanes: 48 v_readlane per grid point).
template <typename T>
__device__ __forceinline__ void ms_cold_fill(const RodConst<T>& Pc, T* cold, int lane) {
  // one value per lane and round; the table is read back by everybody after a wave_sync
  for (int k = lane; k < CD_SIZE; k += WAVE) {
    T v;
    if (k < CD_H0) v = Pc.p0[k - CD_P0];
    else if (k < CD_Q0) v = Pc.h0[k - CD_H0];
    else if (k < CD_W0) v = Pc.q0[k - CD_Q0];
    else if (k < CD_FTIP) v = Pc.w0[k - CD_W0];
    else if (k < CD_MTIP) v = Pc.Ftip[k - CD_FTIP];
    else if (k < CD_TDIRS) v = Pc.Mtip[k - CD_MTIP];
    else if (k < CD_RHOAG) v = Pc.tdirs[k - CD_TDIRS];
    else if (k < CD_KSEI) v = Pc.rhoAg[k - CD_RHOAG];
    else if (k < CD_KSEV) v = Pc.Ksei[k - CD_KSEI];
    else if (k < CD_BSE) v = Pc.Kse_vstar[k - CD_KSEV];
    else if (k < CD_KBTI) v = Pc.Bse[k - CD_BSE];
    else if (k < CD_BBT) v = Pc.Kbti[k - CD_KBTI];
    else v = Pc.Bbt[k - CD_BBT];
    cold[k] = v;
  }
}
template <typename T>
__device__ __forceinline__ V3<T> cold_matvec(const T* A, V3<T> x) {
  return {A[0] * x.x + A[1] * x.y + A[2] * x.z, A[3] * x.x + A[4] * x.y + A[5] * x.z,
          A[6] * x.x + A[7] * x.y + A[8] * x.z};
}
// hist_derive with the matrices read from the cold table (DIAG: only their diagonals, like ode_eval)
template <typename T, bool DIAG>
__device__ __forceinline__ void hist_derive_cold(const T* cold, RodHist<T>& h) {
  if constexpr (DIAG) {
    h.av = {cold[CD_KSEI + 0] * (cold[CD_KSEV + 0] - cold[CD_BSE + 0] * h.vh.x),
            cold[CD_KSEI + 4] * (cold[CD_KSEV + 1] - cold[CD_BSE + 4] * h.vh.y),
            cold[CD_KSEI + 8] * (cold[CD_KSEV + 2] - cold[CD_BSE + 8] * h.vh.z)};
    h.au = {-cold[CD_KBTI + 0] * (cold[CD_BBT + 0] * h.uh.x), -cold[CD_KBTI + 4] * (cold[CD_BBT + 4] * h.uh.y),
            -cold[CD_KBTI + 8] * (cold[CD_BBT + 8] * h.uh.z)};
  } else {
    V3<T> bv = cold_matvec(cold + CD_BSE, h.vh);
    V3<T> t{cold[CD_KSEV] - bv.x, cold[CD_KSEV + 1] - bv.y, cold[CD_KSEV + 2] - bv.z};
    h.av = cold_matvec(cold + CD_KSEI, t);
    V3<T> bu = cold_matvec(cold + CD_BBT, h.uh);
    V3<T> t2 = cold_matvec(cold + CD_KBTI, bu);
    h.au = {-t2.x, -t2.y, -t2.z};
  }
}
template <typename T, int HS, bool DIAG>
__device__ __forceinline__ void build_hist_cold(const T* cold, T hc1, T hc2, const T (&cv)[12], const T (&pv)[12],
                                                T* dst) {
  T hv[HS];
#pragma unroll
  for (int k = 0; k < 12; ++k) hv[k] = hc1 * cv[k] + hc2 * pv[k];
  RodHist<T> h;
  h.qh = {hv[0], hv[1], hv[2]};
  h.wh = {hv[3], hv[4], hv[5]};
  h.vh = {hv[6], hv[7], hv[8]};
  h.uh = {hv[9], hv[10], hv[11]};
  hist_derive_cold<T, DIAG>(cold, h);
  hv[12] = h.av.x; hv[13] = h.av.y; hv[14] = h.av.z;
  hv[15] = h.au.x; hv[16] = h.au.y; hv[17] = h.au.z;
  if constexpr (HS > 18) {
    hv[18] = T(0);
    hv[19] = T(0);
  }
  store_vec<T, HS>(dst, hv);
}

// lane roles inside the wavefront
struct MsRole {
  int iv, col, s_i, len_i, lmax, comp, sbase, srem;
  bool idle;
};
__device__ __forceinline__ MsRole ms_role(int lane, int N) {
  MsRole R;
  R.idle = lane >= 7 + 17 * (MS_P - 1);
  if (lane < 7) { R.iv = 0; R.col = lane; }
  else if (!R.idle) { R.iv = 1 + (lane - 7) / 17; R.col = (lane - 7) % 17; }
  else { R.iv = 0; R.col = 0; }
  const int nseg = N - 1;
  R.sbase = nseg / MS_P;
  R.srem = nseg % MS_P;
  // interval i covers segments [s_i, s_i + len_i)
  R.s_i = R.iv * R.sbase + (R.iv < R.srem ? R.iv : R.srem);
  R.len_i = R.sbase + (R.iv < R.srem ? 1 : 0);
  R.lmax = R.sbase + (R.srem ? 1 : 0);
  // perturbed component in row order: interval 0 perturbs n,m (rows 7..12), others h..w (rows 3..18)
  R.comp = R.col == 0 ? -1 : (R.iv == 0 ? 6 + R.col : 2 + R.col);
  return R;
}
__device__ __forceinline__ int ms_interval_start(int i, int sbase, int srem) { return i * sbase + (i < srem ? i : srem); }

template <typename T>
struct MsLds {
  T* hist;  // [N][HS]
  T* Xs;    // [P][MS_YP]      start states (row order p h n m q w); the unknowns
  T* Es;    // [64][MS_YP]     per lane: end state (unperturbed lanes) or forward-difference column
  T* XB;    // [2][MS_YP][8]   condensed block X_g = [a_g | M_g], ping-pong between stages
  T* Tm;    // [6][8]          rows [rhs | T] of the 6x6 system for dG
  T* cold;  // [CD_SIZE]       per-step parameters (see ms_cold_fill)
  T* Ti;    // [6][6]          inverse of the condensed 6x6 matrix of the last full Newton update (chord check)
  T* c12;   // persistent kernel only: [N][12] leading slots (q w v u) of the newest state
  T* Bp;    // persistent kernel with the MLP on: p-column blocks [3][3][19] (+ scratch), see ms_newton
};
template <typename T, int HS>
__device__ __forceinline__ MsLds<T> ms_carve(T* smem, int N, bool persist, bool nn = false) {
  MsLds<T> L;
  L.hist = smem;
  L.Xs = L.hist + (size_t)N * HS;
  // everything accessed with 16-byte vectors sits at a multiple of 4 elements
  L.cold = L.Xs + ((MS_P * MS_YP + 3) & ~3);
  L.Ti = L.cold + ((CD_SIZE + 3) & ~3);
  L.XB = L.Ti + 40;
  L.Tm = L.XB + 2 * MS_YP * 8;
  L.Es = L.Tm + 48 + (nn ? MS_BO_PAD : 0);
  size_t alg = 2 * MS_YP * 8 + 48 + (nn ? MS_BO_PAD : 0) + ((WAVE * MS_YP + 3) & ~3);  // as in ms_lds_elems
  if (nn && alg < mj_scratch_elems<T>()) alg = mj_scratch_elems<T>();
  L.c12 = persist ? L.XB + ((alg + 3) & ~size_t(3)) : nullptr;
  L.Bp = (persist && nn) ? L.c12 + (((size_t)N * 12 + 3) & ~size_t(3)) : nullptr;
  return L;
}

template <typename T>
struct MsSolveArgs {
  T* out_rod;   // packed state of this rod to stream the accepted sweep into (may be null)
  T* tip;       // 3 values (may be null)
  V3<T> vlast, ulast;
  T tol, tolA, fd_eps;
  int maxit;
  // contraction constant |d_{k+1}| / |d_k|^2 seen where an earlier solve of this rod first got below the
  // tolerance (0: unknown).  In: used to decide which sweep streams the state out; out: refreshed.
  T kappa;
  // persistent several-wavefront kernel (kr_msw_impl.hpp): where a storing sweep also leaves the twelve leading slots
  // of every record, [N][12] in LDS (nullptr: nowhere)
  T* lead12 = nullptr;
  // ... and on a 3-slot ring streams only those twelve slots of an INTERIOR step's interior records to HBM (6 stores per
  // grid point instead of 14 from the three or four storing lanes of a wavefront): nobody reads the rest - the caller gets
  // the tips and the last three states, the kernel's own history needs q w v u (as kr_mso_impl.hpp's lean records)
  bool lean = false;
  bool quick_ok = true;  // option "residual_test": the residual test below may accept a storing sweep
  int prot = 0;          // persistent kernel with the MLP on: sweeps so far (which intervals the p-column lanes serve)
  // fp64 with the MLP on (option "nn_lowp_first"): the FIRST sweep of a step may evaluate the network with the fp32 base
  // chain when the step is going to need two Newton corrections anyway - its update only has to land within ~1e-7 of
  // where the exact one would, the second correction starts from a distance of kappa d1^2 >> that.  Decided from the
  // previous step of the same rod: its first update d1 left 4 kappa d1^2 above the tolerance.  Storing sweeps - the only
  // ones acceptance is measured on - always run the fp64 chain.
  bool lowp_allowed = false, lowp_first = false;
  // fp64 with the MLP on (option "nn_base_only_store"): BASE-ONLY STORING SWEEPS.  A storing sweep needs forward-difference
  // columns only if it is not accepted; evaluated at the unperturbed inputs alone the network costs half (a sweep 231 k
  // cycles instead of 378 k).  Such a sweep is judged like the MLP-off storing sweeps: residual test first (audited with
  // the MLP on: tools/quick_audit_nn.py, update / estimate <= 46, factor 256), else a CHORD update through the factors of
  // the last full sweep - its scratch lies on XB / Tm, so the columns in Es survive it -, accepted below half the
  // tolerance and otherwise TAKEN as the iteration's update (there are no fresh columns to do better with); after two
  // such sweeps in a row the next one is a full one again.
  bool bo_allowed = false;
};

// Scaled maximum norm of the residual of a sweep - the interface jumps E_g - Y_{g+1} (19 rows each) and the tip
// condition - from the end states of the unperturbed lanes in Es: one component per lane, 3 x 19 + 6 = 63 lanes.
template <typename T>
__device__ __forceinline__ float ms_residual_norm(const T* Es, const T* Xs, const T* cold, int lane) {
  static_assert(MS_P == 4, "lane map of the residual components");
  float rn = 0.f;
  if (lane < 3 * MS_YP) {
    const int g = lane / MS_YP, r = lane - MS_YP * g;
    const int l0 = g == 0 ? 0 : 7 + 17 * (g - 1);
    const T x = Xs[(g + 1) * MS_YP + r];
    rn = update_ratio(Es[l0 * MS_YP + r] - x, x);
  } else if (lane < 3 * MS_YP + 6) {
    const int k = lane - 3 * MS_YP;
    const T e = Es[(7 + 17 * 2) * MS_YP + 7 + k];
    rn = update_ratio(cold[CD_FTIP + k] - e, e);
  }
  return wave_max_nonneg(rn);
}

// Newton iteration on (G, Y_1..Y_{P-1}) for one rod = one wavefront.  On entry L.hist holds the
// history records and L.Xs the initial guess (visible to all lanes).  Returns the status; `it`
// = sweeps used.  On exit L.Xs holds the accepted unknowns.
template <typename T, bool DIAG, int SCHEME, int HS, bool PERSIST, bool NN>
__device__ __forceinline__ int ms_newton(const RodConst<T>& Pc, const MlpDev<T>& M, const MsLds<T>& L, const MsRole& R,
                                         int lane, const SweepCtx<T, HS>& C, MsSolveArgs<T>& S, int& it,
                                         MsStamps& stamps) {
  const int N = Pc.N;
#ifdef KR_MS_STAMPS
  unsigned long long tq;
  KR_STAMP(tq);
#endif
  T* Xs = L.Xs; T* Es = L.Es; T* XB = L.XB; T* Tm = L.Tm;
  const int iv = R.iv, col = R.col;
  const bool idle = R.idle;
  bool storing = false, flush = false;
  int status = KR_ST_MAXIT;
  it = 0;
  T dn_prev = T(-1);  // update norm of the previous iteration (contraction estimate)
  const T kappa_in = S.kappa;
  bool below = false;  // an update at or below the tolerance has been seen
  // kept from the last full Newton update of this solve, for the chord check of a storing sweep:
  // this lane's column pair of X_1 .. X_{P-1} (registers), inverse of the 6x6 matrix (L.Ti), the
  // forward-difference columns (Es)
  T Xreg[MS_P - 1][2];
#pragma unroll
  for (int g = 0; g < MS_P - 1; ++g) Xreg[g][0] = Xreg[g][1] = T(0);
  bool have_fac = false;
  const int kp = lane & 3;
  const int r = 3 + (lane >> 2);
  // update norm per unit of residual norm, measured at the last full Newton update of this solve (<= 0: unknown)
  float amp = -1.f;
  // ---- p columns (persistent kernel, MLP on) ----------------------------------------------------------------------
  // The physics does not read p, so the condensed Jacobian carries no columns for the start positions of the intervals
  // 1..3: dE_g/dp_g is the identity.  The network DOES read p, and what it adds, B_g = dE_g/dp_g - [I; 0] (19 x 3), was
  // what held the iteration at a contraction of 1e-3 per sweep (the slowest rods of BASELINE cfg3 needed a fourth
  // sweep for it).  Nine more trajectories do not fit a wavefront (58 + 9 lanes), so the six
  // spare lanes integrate the p columns of TWO intervals per sweep, in rotation; B_g lives in LDS across sweeps and
  // time steps (it only corrects the Jacobian: a block one or two sweeps old is accurate to a few per cent, which puts
  // the contraction at ~1e-5), their network evaluations ride on the free columns 6..14 of sample tile 0, and the
  // Newton update gets one step of defect correction, d = d0 + Jt^-1 (B dp0), through the factors just computed.
  // (fp64 only: with fp32's tolerance of 1e-5 the third sweep of a step is the accepted one either way, and the extra
  //  lanes and the second solve only cost - measured 0.55 -> 0.56 ms per step at cfg3)
  constexpr bool PCOL = NN && PERSIST && SCHEME == KR_EULER && sizeof(T) == 8;
  const bool pl = PCOL && idle;                     // lanes 58..63
  const int pslot = lane - (7 + 17 * (MS_P - 1));   // 0..5 on those lanes
  T* Bp = L.Bp;

  int bo_run = 0;  // base-only sweeps since the last full one
  while (true) {
    // ---- role of the spare lanes in this sweep ------------------------------
    int iv_l = iv, s_l = R.s_i, len_l = R.len_i, pc = 0, pg = 0;
    SweepCtx<T, HS> Cl = C;
    const bool bo = NN && S.bo_allowed && bo_run < 2 && storing && !flush && have_fac && dn_prev > T(0) && dn_prev <= T(1e-2);
    if constexpr (NN) Cl.role.base_only = bo;
    bo_run = bo ? bo_run + 1 : 0;
    if constexpr (PCOL) {
      const int rot = S.prot % 3;  // intervals served: (1, 2), (3, 1), (2, 3)
      const int ga = rot == 0 ? 1 : rot == 1 ? 3 : 2, gb = rot == 0 ? 2 : rot == 1 ? 1 : 3;
      const int gu = 6 - ga - gb;  // the interval nobody serves this sweep
      S.prot += 1;
      Cl.role.ptab = 1;
      Cl.role.lowp = S.lowp_allowed && S.lowp_first && it == 0 && !storing && !flush;
      if (pl) {
        pg = pslot < 3 ? ga : gb;
        pc = pslot < 3 ? pslot : pslot - 3;
        iv_l = pg;
        s_l = ms_interval_start(pg, R.sbase, R.srem);
        len_l = R.sbase + (pg < R.srem ? 1 : 0);
        Cl.role.iv = pg; Cl.role.col = 1; Cl.role.idle = false;
        Cl.role.xrow = 6 + 3 * (pg - 1) + pc;
      } else if (col == 0) {
        // (the unperturbed lanes zero the dx rows of sample tile 0 nobody owns: the unserved interval's three, and row 15)
        Cl.role.zrow = iv < 3 ? 6 + 3 * (gu - 1) + iv : 15;
      }
    }
    // ---- start state of this lane ------------------------------------------
    T yr[19];
#pragma unroll
    for (int r = 0; r < 19; ++r) yr[r] = Xs[iv_l * MS_YP + r];
    // forward-difference step of this lane's column (one LDS read at the lane's own component instead of a
    // 16-way select over the registers, which the compiler also re-derived after the sweep)
    const T hstep = col > 0 ? S.fd_eps * fmax(fabs(Xs[iv * MS_YP + (R.comp > 0 ? R.comp : 3)]), T(1)) : T(1);
#pragma unroll
    for (int r = 3; r < 19; ++r) yr[r] += r == R.comp ? hstep : T(0);
    T hp = T(1);
    if constexpr (PCOL) {
      if (pl) {
        hp = S.fd_eps * fmax(fabs(Xs[pg * MS_YP + pc]), T(1));
#pragma unroll
        for (int r = 0; r < 3; ++r) yr[r] += r == pc ? hp : T(0);
      }
    }
    RodState<T> y = rows_to_state(yr);
    const bool st = (storing || flush) && col == 0 && !idle;

    // ---- sweep over this lane's sub-interval --------------------------------
    T hv[HS];
    load_hist_vec<T, HS>(C.hbase + (size_t)s_l * HS, hv);
    // one grid point: evaluate, (on the accepted sweep, unperturbed lanes) stream the record out, advance.
    // STORE is a compile-time tag so that the other sweeps run a branch-free body the scheduler can
    // overlap across consecutive grid points.
    auto point = [&](auto store_tag, int j, bool live) __attribute__((always_inline)) {
      constexpr bool STORE = decltype(store_tag)::value;
      RodState<T> k1;
      V3<T> v, u;
      eval_point<T, DIAG, NN, HS, PERSIST>(Pc, M, Cl, y, hv, k1, v, u);
      if constexpr (STORE) {
        if (st && live) {
          T rec[KR_SLOTS];
          record_from(y, v, u, rec);
          if (S.out_rod) store_record(S.out_rod + (size_t)j * KR_SLOTS, rec);
          if constexpr (PERSIST) {
            T lead[12];
#pragma unroll
            for (int c = 0; c < 12; ++c) lead[c] = rec[c];
            store_vec<T, 12>(L.c12 + (size_t)j * 12, lead);
          }
        }
      }
      if constexpr (SCHEME == KR_EULER) {
        load_hist_vec<T, HS>(C.hbase + (size_t)(j + 1) * HS, hv);
        y = state_axpy(y, Pc.ds, k1);
      } else {
        T hn[HS], hm[HS];
        load_hist_vec<T, HS>(C.hbase + (size_t)(j + 1) * HS, hn);
#pragma unroll
        for (int c = 0; c < HS; ++c) hm[c] = T(0.5) * (hv[c] + hn[c]);
        RodState<T> k2, k3, k4;
        V3<T> v2, u2;
        RodState<T> ya = state_axpy(y, Pc.ds * T(0.5), k1);
        eval_point<T, DIAG, NN, HS, PERSIST>(Pc, M, C, ya, hm, k2, v2, u2);
        ya = state_axpy(y, Pc.ds * T(0.5), k2);
        eval_point<T, DIAG, NN, HS, PERSIST>(Pc, M, C, ya, hm, k3, v2, u2);
        ya = state_axpy(y, Pc.ds, k3);
        eval_point<T, DIAG, NN, HS, PERSIST>(Pc, M, C, ya, hn, k4, v2, u2);
        RodState<T> ksum = state_axpy(k1, T(2), k2);
        ksum = state_axpy(ksum, T(2), k3);
        ksum = state_axpy(ksum, T(1), k4);
        y = state_axpy(y, Pc.ds / T(6), ksum);
#pragma unroll
        for (int c = 0; c < HS; ++c) hv[c] = hn[c];
      }
    };
    if constexpr (NN) {
      // with the MLP on every lane must reach the wave-wide matrix-core call: lanes past the end of
      // their (shorter) interval keep running on the last grid point and simply do not commit
      for (int t = 0; t < R.lmax; ++t) {
        const bool live = t < len_l;
        const int j = live ? s_l + t : s_l + len_l - 1;
        const RodState<T> y_in = y;
        if (storing || flush) point(std::true_type{}, j, live);  // wave-uniform choice
        else point(std::false_type{}, j, true);
        if (!live) y = y_in;
      }
    } else {
      // every interval has sbase or sbase + 1 segments: the first sbase grid points need no predicate
      if (storing || flush) {
        for (int t = 0; t < R.sbase; ++t) point(std::true_type{}, R.s_i + t, true);
        if (R.len_i > R.sbase) point(std::true_type{}, R.s_i + R.sbase, true);
      } else {
        if constexpr (SCHEME == KR_EULER) {
#pragma unroll kMsUnroll
          for (int t = 0; t < R.sbase; ++t) point(std::false_type{}, R.s_i + t, true);
        } else {
          for (int t = 0; t < R.sbase; ++t) point(std::false_type{}, R.s_i + t, true);
        }
        if (R.len_i > R.sbase) point(std::false_type{}, R.s_i + R.sbase, true);
      }
    }
    if (st && iv == MS_P - 1) {
      T rec[KR_SLOTS];
      record_from(y, S.vlast, S.ulast, rec);
      if (S.out_rod) store_record(S.out_rod + (size_t)(N - 1) * KR_SLOTS, rec);
      if constexpr (PERSIST) {
        T lead[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) lead[c] = rec[c];
        store_vec<T, 12>(L.c12 + (size_t)(N - 1) * 12, lead);
      }
      if (S.tip) { S.tip[0] = y.p.x; S.tip[1] = y.p.y; S.tip[2] = y.p.z; }
    }
    if (flush) break;
    ++it;
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(stamps.sweep, tq);
#endif

    // ---- Newton update of this sweep ----------------------------------------------------
    // A storing sweep that follows a small update is first checked with a CHORD update - the residual of
    // this sweep through the Jacobian factors of the previous one (forward-difference columns still in Es,
    // X_g pairs in registers, T^-1 in LDS): it differs from the Newton update by a relative O(|d_prev|) <= 1 %
    // and costs a third of the full condensation.  If it is below half the tolerance the sweep is accepted;
    // otherwise the full update below is computed as usual.
#ifdef KR_MS_STAMPS
    unsigned long long ta = tq;
#endif
    // (with the MLP on the factors in Es do not survive a sweep - the evaluator's scratch lies over them - so there is no
    //  chord check; the chord branch is only entered right after a full update, for the p columns)
    const bool small_prev = storing && have_fac && dn_prev > T(0) && dn_prev <= T(1e-2);
    bool chord = (!NN && small_prev) || bo;
    T d[6];
    T updY[MS_P - 1];
    T* dYb = XB;  // [g][19] scratch for dY_1 .. dY_{P-2} (p rows below)
    float dnf;
    float res_full = -1.f;  // residual norm of this sweep (full Newton branch only)
    bool finite;
    T updP, updG, xsP, xsG, xsY[MS_P - 1];
    const bool plane = lane < 3 * (MS_P - 1);
    const int pi = plane ? lane / 3 : 0;     // term i = 0 .. P-2
    const int prow = plane ? lane - 3 * pi : 0;
    const bool glane = lane >= WAVE - 6;  // six otherwise idle lanes own the base wrench
    // Residual test, tried first on a storing sweep that follows a small update: the update this sweep would produce is
    // J^-1 times its residual, and `amp` is the ratio |update| / |residual| that the previous (full) iteration of this
    // solve measured.  The two residuals point in different directions, so the ratio is only indicative: audited over
    // 8 workloads x 1024 rods x 300 steps (tools/quick_audit.py, build with -DKR_QUICK_AUDIT) the chord update is at most
    // 39 x the estimate (median 12).  With a safety factor of 256 the sweep is accepted without any condensation
    // when the predicted update is below the tolerance - in the two-sweep regime the actual update is ~1e-12 against
    // a tolerance of 1e-8, so the factor costs nothing there.  Otherwise the chord check decides as before.
    // defect correction for the p columns (PCOL): first pass = the update of the condensed system, second pass = the same
    // factors applied to B dp of the first
    bool pcorr = false;
    T d1[6], updY1[MS_P - 1], updP1 = T(0);
#pragma unroll
    for (int k = 0; k < 6; ++k) d1[k] = T(0);
#pragma unroll
    for (int g = 0; g < MS_P - 1; ++g) updY1[g] = T(0);
    bool quick = false;
    float quick_est = 0.f;
#ifdef KR_QUICK_AUDIT
    if (S.quick_ok && small_prev && amp > 0.f) {  // (audit build: the estimate is formed with the MLP on as well - tools/quick_audit_nn.py)
#else
    if ((!NN || bo) && S.quick_ok && small_prev && amp > 0.f) {  // (with the MLP on: only on a base-only storing sweep)
#endif
      {
        T er[19];
        state_to_rows(y, er);
        if (col == 0 && !idle) {
#pragma unroll
          for (int q = 0; q < 19; ++q) Es[lane * MS_YP + q] = er[q];
        }
      }
      wave_sync();
      const float rn = ms_residual_norm<T>(Es, Xs, L.cold, lane);
      quick_est = amp * rn;
#ifndef KR_QUICK_AUDIT
      quick = T(256) * (T)quick_est <= S.tol;  // (NaN compares false)
#endif
    }
    while (true) {
    if (quick) {
#ifdef KR_MS_STAMPS
      stamps.qn += 1.0;
#endif
      dnf = quick_est;
      finite = true;
      updP = T(0); updG = T(0); xsP = T(0); xsG = T(0);
      break;
    }
    if (chord) {
      T* ach = XB;             // a_g, g = 1 .. P-1: [g][19]
      T* rt = XB + 4 * MS_YP;  // tip right-hand side [6]
      dYb = XB + MS_YP * 8;
      if (!pcorr) {
        T er[19];
        state_to_rows(y, er);
        if (col == 0 && !idle) {
#pragma unroll
          for (int q = 0; q < 19; ++q) Es[lane * MS_YP + q] = er[q];
        }
      }
      wave_sync();
      T areg[MS_P - 1];
      areg[0] = Es[0 * MS_YP + r] - Xs[1 * MS_YP + r];  // a_1 = c_0
      if (kp == 0) ach[1 * MS_YP + r] = areg[0];
      wave_sync();
#pragma unroll
      for (int g = 1; g < MS_P; ++g) {
        const int l0 = 7 + 17 * (g - 1);
        T av[4], xv[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          av[cc] = Es[(l0 + 1 + 4 * kp + cc) * MS_YP + r];
          xv[cc] = ach[g * MS_YP + 3 + 4 * kp + cc];
        }
        const T e0 = Es[l0 * MS_YP + r];
        const T ynext = g < MS_P - 1 ? Xs[(g + 1) * MS_YP + r] : T(0);
        T part = fma(av[0], xv[0], av[1] * xv[1]) + fma(av[2], xv[2], av[3] * xv[3]);
        part += quad_xor<0xB1>(part);
        part += quad_xor<0x4E>(part);
        if (g < MS_P - 1) {
          areg[g] = e0 - ynext + part;
          if (kp == 0) ach[(g + 1) * MS_YP + r] = areg[g];
        } else if (kp == 0 && r >= 7 && r < 13) {
          rt[r - 7] = L.cold[CD_FTIP + (r - 7)] - e0 - part;
        }
        wave_sync();
      }
      {
        T rtv[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) rtv[j] = rt[j];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          T acc = T(0);
#pragma unroll
          for (int j = 0; j < 6; ++j) acc = fma(L.Ti[i * 6 + j], rtv[j], acc);
          d[i] = acc;
        }
      }
      {
        const T dd0 = kp == 0 ? T(0) : kp == 1 ? d[1] : kp == 2 ? d[3] : d[5];  // (the a column is the new one)
        const T dd1 = kp == 0 ? d[0] : kp == 1 ? d[2] : kp == 2 ? d[4] : T(0);
#pragma unroll
        for (int g = 1; g < MS_P; ++g) {
          T sx = fma(Xreg[g - 1][1], dd1, Xreg[g - 1][0] * dd0);
          sx += quad_xor<0xB1>(sx);
          sx += quad_xor<0x4E>(sx);
          updY[g - 1] = areg[g - 1] + sx;
        }
      }
    } else {
      dYb = XB;
      // ---- hand the end states over ----------------------------------------------------
      // Es[lane] = end state E of the unperturbed lanes, forward-difference column
      // (E - E_unperturbed) / step of the others: column c of A_g = dE_g/dY_g sits at Es[l0_g + 1 + c].
      {
        T er[19];
        state_to_rows(y, er);
        const int l0own = iv == 0 ? 0 : 7 + 17 * (iv - 1);
        if (col == 0 && !idle) {
  #pragma unroll
          for (int r = 0; r < 19; ++r) Es[lane * MS_YP + r] = er[r];
        }
        wave_sync();
        res_full = ms_residual_norm<T>(Es, Xs, L.cold, lane);
        if (col > 0) {
          const T ih = fast_rcp(hstep);
          T e0[19];  // all loads first: the compiler cannot tell that they never alias the stores below
  #pragma unroll
          for (int r = 0; r < 19; ++r) e0[r] = Es[l0own * MS_YP + r];
          __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
          for (int r = 0; r < 19; ++r) Es[lane * MS_YP + r] = (er[r] - e0[r]) * ih;
        }
        if constexpr (PCOL) {
          if (pl) {  // column pc of B_pg = dE/dp - [I; 0]
            const T ihp = fast_rcp(hp);
            const int l0p = 7 + 17 * (pg - 1);
            T e0[19];
  #pragma unroll
            for (int r = 0; r < 19; ++r) e0[r] = Es[l0p * MS_YP + r];
            __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
            for (int r = 0; r < 19; ++r) Bp[((pg - 1) * 3 + pc) * 19 + r] = (er[r] - e0[r]) * ihp - (r == pc ? T(1) : T(0));
          }
        }
        wave_sync();
      }

  #ifdef KR_MS_STAMPS
      KR_STAMP_ADD(stamps.a1, ta);
  #endif
      // ---- condensation -------------------------------------------------------------------
      // dY_1 = c_0 + A_0 dG,  dY_{g+1} = c_g + A_g dY_g  with c_g = E_g - Y_{g+1}; written as
      // dY_g = X_g [1; dG], X_g = [a_g | M_g] (19 x 7).  The p rows never feed back (no equation reads
      // p), so the chain runs on rows 3..18 only: lane -> (row r, column pair kp) of X_g, 16 x 4 = 64
      // lanes, 2 x 16 fused multiply-adds per stage, X_g handed on through a ping-pong LDS tile.
      {
        const T c0 = Es[0 * MS_YP + r] - Xs[1 * MS_YP + r];
        const T da = Es[(2 * kp) * MS_YP + r];      // lanes 1..6 hold the columns of A_0 (lane 0: E_0, unused)
        const T db = Es[(2 * kp + 1) * MS_YP + r];  // (lane 7 is E_1: masked below)
        Xreg[0][0] = kp == 0 ? c0 : da;
        Xreg[0][1] = kp == 3 ? T(0) : db;
        store_pair(XB + r * 8 + 2 * kp, Xreg[0][0], Xreg[0][1]);
      }
      wave_sync();
  #pragma unroll
      for (int g = 1; g < MS_P; ++g) {
        const T* xcur = XB + ((g - 1) & 1) * (MS_YP * 8);
        T* xnext = XB + (g & 1) * (MS_YP * 8);
        const int l0 = 7 + 17 * (g - 1);  // unperturbed lane of interval g
        const T e0 = Es[l0 * MS_YP + r];
        T n0 = T(0), n1 = T(0), n2 = T(0), n3 = T(0);
        if (g < MS_P - 1) {
          const T cg = e0 - Xs[(g + 1) * MS_YP + r];
          n0 = kp == 0 ? cg : T(0);
        }
        {
          // issue all 32 LDS reads of the stage back to back, then the arithmetic (left alone, the
          // scheduler waits for every read before it issues the next one)
          T av[16], xa[16], xb[16];
  #pragma unroll
          for (int c = 0; c < 16; ++c) {
            av[c] = Es[(l0 + 1 + c) * MS_YP + r];
            load_pair(xcur + (3 + c) * 8 + 2 * kp, xa[c], xb[c]);
          }
          __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
          for (int c = 0; c < 16; c += 2) {
            n0 = fma(av[c], xa[c], n0);
            n1 = fma(av[c], xb[c], n1);
            n2 = fma(av[c + 1], xa[c + 1], n2);
            n3 = fma(av[c + 1], xb[c + 1], n3);
          }
          n0 += n2;
          n1 += n3;
        }
        if (g < MS_P - 1) {
          Xreg[g][0] = n0;
          Xreg[g][1] = n1;
          store_pair(xnext + r * 8 + 2 * kp, n0, n1);
        } else if (r >= 7 && r < 13) {
          // tip rows: [n;m](E_{P-1} + A_{P-1} dY_{P-1}) = [F_tip; M_tip]  ->  row [rhs | T] of T dG = rhs
          if (kp == 0) n0 = L.cold[CD_FTIP + (r - 7)] - e0 - n0;  // F_tip (3) and M_tip (3) are adjacent
          store_pair(Tm + (r - 7) * 8 + 2 * kp, n0, n1);
        }
        wave_sync();
      }

  #ifdef KR_MS_STAMPS
      KR_STAMP_ADD(stamps.a2, ta);
  #endif
      // ---- 6x6 solve, redundantly in every lane (registers only) -------------------------
      {
        T a6[6][7];
  #pragma unroll
        for (int i = 0; i < 6; ++i) {
          T row[8];
          load_hist_vec<T, 8>(Tm + i * 8, row);
  #pragma unroll
          for (int k = 0; k < 6; ++k) a6[i][k] = row[1 + k];
          a6[i][6] = lane < 6 ? (lane == i ? T(1) : T(0)) : row[0];  // lanes 0..5: unit vectors -> columns of T^-1
        }
        T x6[6];
        solve6(a6, x6);
        if (lane < 6) {
  #pragma unroll
          for (int i = 0; i < 6; ++i) L.Ti[i * 6 + lane] = x6[i];
        }
  #pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = lane_bcast<6>(x6[i]);
        have_fac = true;
      }

  #ifdef KR_MS_STAMPS
      KR_STAMP_ADD(stamps.a3, ta);
  #endif
      // ---- updates and convergence --------------------------------------------------------
      // rows 3..18 of dY_g: every lane of a quad ends up with the full dot product X_g[r] . [1; dG]
      {
        const T dd0 = kp == 0 ? T(1) : kp == 1 ? d[1] : kp == 2 ? d[3] : d[5];
        const T dd1 = kp == 0 ? d[0] : kp == 1 ? d[2] : kp == 2 ? d[4] : T(0);
  #pragma unroll
        for (int g = 1; g < MS_P; ++g) {
          T s = fma(Xreg[g - 1][1], dd1, Xreg[g - 1][0] * dd0);
          s += quad_xor<0xB1>(s);  // lanes ^1
          s += quad_xor<0x4E>(s);  // lanes ^2
          updY[g - 1] = s;
        }
      }
    }
    // the p rows: dY_{g}[p] = sum_{i<g} (c_i[p] + A_i[p,:] dY_i[3:]), one term per lane (i, row)
    T* sp = Tm;   // [P-1][3] partial sums (the solve has consumed Tm)
    if (kp == 0) {
#pragma unroll
      for (int g = 1; g < MS_P - 1; ++g) dYb[g * MS_YP + r] = updY[g - 1];
    }
    wave_sync();
    if (plane) {
      const int l0 = pi == 0 ? 0 : 7 + 17 * (pi - 1);
      T s = Es[l0 * MS_YP + prow] - Xs[(pi + 1) * MS_YP + prow];
      // uniform trip count (A_0 has 6 columns: the rest is masked) so that the reads can be batched
      T av[16], dv[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const bool use = pi > 0 || c < 6;
        av[c] = use ? Es[(l0 + 1 + c) * MS_YP + prow] : T(0);
        dv[c] = pi > 0 ? dYb[pi * MS_YP + 3 + c] : T(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      T s2 = T(0);
#pragma unroll
      for (int c = 0; c < 16; c += 2) {
        const T da = pi > 0 ? dv[c] : (c < 6 ? d[c < 6 ? c : 0] : T(0));
        const T db = pi > 0 ? dv[c + 1] : (c + 1 < 6 ? d[c + 1 < 6 ? c + 1 : 0] : T(0));
        s = fma(av[c], da, s);
        s2 = fma(av[c + 1], db, s2);
      }
      sp[pi * 3 + prow] = s + s2;
    }
    wave_sync();

    // scaled update norm over every unknown (base wrench and interior states).  The norm only steers the
    // iteration (stop test, contraction estimate), so it is formed in fp32 with the hardware reciprocal.
    dnf = 0.f;
    updP = T(0); updG = T(0); xsP = T(0); xsG = T(0);
    if constexpr (PCOL) {
      if (pcorr) {  // second pass: what was computed above is the correction - add the update of the first pass
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] += d1[k];
#pragma unroll
        for (int g = 0; g < MS_P - 1; ++g) updY[g] += updY1[g];
      }
    }
    if (plane) {  // this lane owns Y_{pi+1}[prow]
      xsP = Xs[(pi + 1) * MS_YP + prow];
#pragma unroll
      for (int i = 0; i < MS_P - 1; ++i) {
        const T t = sp[i * 3 + prow];
        updP += i <= pi ? t : T(0);
      }
      if constexpr (PCOL) updP += pcorr ? updP1 : T(0);
      dnf = update_ratio(updP, xsP);
    }
    if (glane) {
      const int k = lane - (WAVE - 6);
      xsG = Xs[0 * MS_YP + 7 + k];
      updG = k == 0 ? d[0] : k == 1 ? d[1] : k == 2 ? d[2] : k == 3 ? d[3] : k == 4 ? d[4] : d[5];
      dnf = fmaxf(dnf, update_ratio(updG, xsG));
    }
#pragma unroll
    for (int g = 1; g < MS_P; ++g) {
      xsY[g - 1] = T(0);
      if (((g - 1) & 3) == kp) {  // one lane of the quad owns Y_g[r]
        xsY[g - 1] = Xs[g * MS_YP + r];
        dnf = fmaxf(dnf, update_ratio(updY[g - 1], xsY[g - 1]));
      }
    }
    dnf = wave_max_nonneg(dnf);  // +inf if any update is not finite
    finite = dnf <= 3.0e38f;
    if constexpr (PCOL) {
      if (!pcorr && !chord && finite) {
        // ---- defect correction for the p columns ------------------------------------------------------------
        // The update above solves the system WITHOUT the blocks B_g.  With them, dY_{g+1} = c_g + A_g dY_g + B_g dY_g[p]:
        // the difference satisfies the same recursion with "residuals" c'_g = B_g dY_g[p] (c'_0 = 0) and a homogeneous
        // tip condition, which is exactly what the chord branch solves from the base end-state slots of Es - so those
        // slots are rewritten as Y_{g+1} + c'_g (tip rows: F_tip + c'_3) and the loop goes round once more.  One step
        // of this leaves an error of (1e-3)^2 of the update.
#pragma unroll
        for (int k = 0; k < 6; ++k) d1[k] = d[k];
#pragma unroll
        for (int g = 0; g < MS_P - 1; ++g) updY1[g] = updY[g];
        updP1 = updP;
        T* dPl = Bp + 3 * 3 * 19 + 1;  // [interval 1..3][3]: p update of the first pass
        if (plane) dPl[pi * 3 + prow] = updP;
        wave_sync();
        if (lane < 19) {
#pragma unroll
          for (int g = 0; g < MS_P; ++g) {
            const int l0 = g == 0 ? 0 : 7 + 17 * (g - 1);
            T cp = T(0);
            if (g > 0) {
#pragma unroll
              for (int c = 0; c < 3; ++c) cp = fma(Bp[((g - 1) * 3 + c) * 19 + lane], dPl[(g - 1) * 3 + c], cp);
            }
            T basev;
            if (g < MS_P - 1) basev = Xs[(g + 1) * MS_YP + lane];
            else basev = (lane >= 7 && lane < 13) ? L.cold[CD_FTIP + (lane - 7)] : T(0);
            Es[l0 * MS_YP + lane] = basev + cp;
          }
        }
        wave_sync();
        pcorr = true;
        chord = true;
        continue;
      }
      if (pcorr) break;  // (the second pass is a correction, not a chord check: nothing to be conclusive about)
    }
    if (chord && !bo && !(finite && (T)dnf <= T(0.5) * S.tol)) {
      chord = false;  // not conclusive: compute the Newton update proper
      continue;
    }
    break;  // (a base-only sweep has no columns of its own: its chord update is the update of this iteration)
    }
    const T dn = (T)dnf;
    if (res_full > 0.f && finite) amp = dnf / res_full;
    if constexpr (PCOL) {
      // the next step's first sweep may run the network's base chain in fp32 iff this step's first update left the
      // predicted error 4 kappa d1^2 above the tolerance (a second correction was needed whatever the first sweep's accuracy)
      if (it == 1) S.lowp_first = finite && (kappa_in > T(0) ? T(4) * kappa_in * dn * dn > S.tol : dn > T(1e-3));
    }
#if defined(KR_MS_STAMPS) && defined(KR_QUICK_AUDIT)
    if (quick_est > 0.f && finite && dnf > 0.f) { stamps.qa = fmax(stamps.qa, (double)(dnf / quick_est)); stamps.qn += 1.0; }
#endif

    if (finite && !below && dn <= S.tol) {
      below = true;
      // (the measured norm bottoms out at rounding level: bound it from below so that one lucky
      // step does not make the estimate wildly optimistic)
      if (dn_prev > T(0)) {
        const T floor_dn = T(64) * (sizeof(T) == 8 ? T(2.2e-16) : T(1.2e-7));
        const T k = fmax(dn, floor_dn) * fast_rcp(dn_prev * dn_prev);
        S.kappa = fmin(fmax(k, T(1e-4)), T(1));
      }
    }
    bool done = false;
    if (!finite) {
      done = true;
      status = KR_ST_NONFINITE;
      flush = !storing;  // nothing consistent stored yet: stream the current iterate once
    } else if (storing && dn <= (bo && !quick ? T(0.5) * S.tol : S.tol)) {  // (a chord update is conclusive below HALF the tolerance)
      done = true;  // the state streamed out by this sweep is the accepted one
      status = KR_ST_CONVERGED;
    } else {
      if (plane) Xs[(pi + 1) * MS_YP + prow] = xsP + updP;
      if (glane) Xs[0 * MS_YP + 7 + (lane - (WAVE - 6))] = xsG + updG;
#pragma unroll
      for (int g = 1; g < MS_P; ++g)
        if (((g - 1) & 3) == kp) Xs[g * MS_YP + r] = xsY[g - 1] + updY[g - 1];
      // stream the state out on the sweep that is expected to be accepted: Newton contracts
      // quadratically, |d_{k+1}| ~ kappa |d_k|^2 with kappa estimated from the last two updates
      if (predict_final<T>(dn, dn_prev, S.tol, S.tolA)) storing = true;
      // the estimate from inside one solve is pessimistic (the first update mostly closes the interface
      // jumps, which is a linear problem); the constant remembered from the previous time step is not
      if (kappa_in > T(0) && T(4) * kappa_in * dn * dn <= S.tol) storing = true;
      dn_prev = dn;
      if (it >= S.maxit) {
        done = true;
        status = KR_ST_MAXIT;
        flush = true;  // the unknowns moved after the last stored sweep
      }
    }
    wave_sync();
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(stamps.a4, ta);
    KR_STAMP_ADD(stamps.alg, tq);
    stamps.its += 1;
    if (it <= 4) stamps.dn[it - 1] = (double)dn;
#endif
    if (done && !flush) break;
    if (done && flush) storing = false;
  }
  return status;
}

// ---------------------------------------------------------------------------
// Last resort of a time step: damped single shooting.  When the iteration above fails from the predicted AND from
// the warm start (the reference's hybrd has a trust region where plain Newton has nothing; the untrained 512-wide
// default network of cosserat_ode_torch.py:60-88 needs it from its 10th step on, tests/golden/sim_more.npz), the
// wavefront solves the step the way the CPU oracle's `newton_shoot` does: Newton on the 6 base unknowns with a
// forward-difference Jacobian from lanes 1..6 (exact - unlike the condensed one it also sees the MLP's dependence
// on p), every update taken as G - lam d with lam halved until the residual norm has decreased.  Lane 0 streams
// every sweep; the sweep whose full Newton update is below the tolerance is the accepted one.  58 lanes idle along
// the N - 1 dependent grid points: this path is for robustness, not speed.  Leaves L.Xs consistent with the result.
// ---------------------------------------------------------------------------
template <typename T, bool DIAG, int SCHEME, int HS, bool PERSIST, bool NN>
__device__ __forceinline__ int ss_newton_damped(const RodConst<T>& Pc, const MlpDev<T>& M, const MsLds<T>& L,
                                                const MsRole& R, int lane, const SweepCtx<T, HS>& C, MsSolveArgs<T>& S,
                                                int& it) {
  const int N = Pc.N;
  const int col = lane < 7 ? lane : 0;  // lanes 7.. duplicate the unperturbed column and store nothing
  T G[6], Gold[6], d[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { G[k] = L.Xs[0 * MS_YP + 7 + k]; Gold[k] = G[k]; d[k] = T(0); }
  T nr_old = T(-1), lam = T(1);
  bool have_trial = false;
  int status = KR_ST_MAXIT;
  const int maxit = 8 * S.maxit;
  it = 0;
  T* Rx = L.Tm;  // [7][6] residuals of the seven columns
  while (true) {
    T hs[6];
    RodState<T> y;
    {
      const T* cold = L.cold;
      y.p = {cold[CD_P0], cold[CD_P0 + 1], cold[CD_P0 + 2]};
      y.h0 = cold[CD_H0]; y.h1 = cold[CD_H0 + 1]; y.h2 = cold[CD_H0 + 2]; y.h3 = cold[CD_H0 + 3];
      y.q = {cold[CD_Q0], cold[CD_Q0 + 1], cold[CD_Q0 + 2]};
      y.w = {cold[CD_W0], cold[CD_W0 + 1], cold[CD_W0 + 2]};
      T Gl[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        hs[k] = S.fd_eps * fmax(fabs(G[k]), T(1));
        Gl[k] = G[k] + (col == k + 1 ? hs[k] : T(0));
      }
      y.n = {Gl[0], Gl[1], Gl[2]};
      y.m = {Gl[3], Gl[4], Gl[5]};
    }
    T hv[HS];
    load_hist_vec<T, HS>(C.hbase, hv);
    int gnext = 1, jnext = ms_interval_start(1, R.sbase, R.srem);  // interval starts passed on the way (for L.Xs)
    for (int j = 0; j < N - 1; ++j) {
      RodState<T> k1;
      V3<T> v, u;
      eval_point<T, DIAG, NN, HS, PERSIST>(Pc, M, C, y, hv, k1, v, u);
      if (lane == 0) {
        T rec[KR_SLOTS];
        record_from(y, v, u, rec);
        if (S.out_rod) store_record(S.out_rod + (size_t)j * KR_SLOTS, rec);
        if constexpr (PERSIST) {
          T lead[12];
#pragma unroll
          for (int c = 0; c < 12; ++c) lead[c] = rec[c];
          store_vec<T, 12>(L.c12 + (size_t)j * 12, lead);
        }
        if (j == jnext && gnext < MS_P) {
          T yr[19];
          state_to_rows(y, yr);
#pragma unroll
          for (int q = 0; q < 19; ++q) L.Xs[gnext * MS_YP + q] = yr[q];
        }
      }
      if (j == jnext && gnext < MS_P) { ++gnext; jnext = ms_interval_start(gnext, R.sbase, R.srem); }
      if constexpr (SCHEME == KR_EULER) {
        load_hist_vec<T, HS>(C.hbase + (size_t)(j + 1) * HS, hv);
        y = state_axpy(y, Pc.ds, k1);
      } else {
        T hn[HS], hm[HS];
        load_hist_vec<T, HS>(C.hbase + (size_t)(j + 1) * HS, hn);
#pragma unroll
        for (int c = 0; c < HS; ++c) hm[c] = T(0.5) * (hv[c] + hn[c]);
        RodState<T> k2, k3, k4;
        V3<T> v2, u2;
        RodState<T> ya = state_axpy(y, Pc.ds * T(0.5), k1);
        eval_point<T, DIAG, NN, HS, PERSIST>(Pc, M, C, ya, hm, k2, v2, u2);
        ya = state_axpy(y, Pc.ds * T(0.5), k2);
        eval_point<T, DIAG, NN, HS, PERSIST>(Pc, M, C, ya, hm, k3, v2, u2);
        ya = state_axpy(y, Pc.ds, k3);
        eval_point<T, DIAG, NN, HS, PERSIST>(Pc, M, C, ya, hn, k4, v2, u2);
        RodState<T> ksum = state_axpy(k1, T(2), k2);
        ksum = state_axpy(ksum, T(2), k3);
        ksum = state_axpy(ksum, T(1), k4);
        y = state_axpy(y, Pc.ds / T(6), ksum);
#pragma unroll
        for (int c = 0; c < HS; ++c) hv[c] = hn[c];
      }
    }
    if (lane == 0) {
      T rec[KR_SLOTS];
      record_from(y, S.vlast, S.ulast, rec);
      if (S.out_rod) store_record(S.out_rod + (size_t)(N - 1) * KR_SLOTS, rec);
      if constexpr (PERSIST) {
        T lead[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) lead[c] = rec[c];
        store_vec<T, 12>(L.c12 + (size_t)(N - 1) * 12, lead);
      }
      if (S.tip) { S.tip[0] = y.p.x; S.tip[1] = y.p.y; S.tip[2] = y.p.z; }
    }
    ++it;
    if (lane < 7) {
      Rx[lane * 6 + 0] = L.cold[CD_FTIP + 0] - y.n.x; Rx[lane * 6 + 1] = L.cold[CD_FTIP + 1] - y.n.y;
      Rx[lane * 6 + 2] = L.cold[CD_FTIP + 2] - y.n.z; Rx[lane * 6 + 3] = L.cold[CD_MTIP + 0] - y.m.x;
      Rx[lane * 6 + 4] = L.cold[CD_MTIP + 1] - y.m.y; Rx[lane * 6 + 5] = L.cold[CD_MTIP + 2] - y.m.z;
    }
    wave_sync();
    T a[6][7];
    T nr = T(0);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      a[k][6] = Rx[k];
      nr = fma(a[k][6], a[k][6], nr);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const T ih = fast_rcp(hs[c]);
#pragma unroll
      for (int k = 0; k < 6; ++k) a[k][c] = (Rx[(c + 1) * 6 + k] - a[k][6]) * ih;
    }
    wave_sync();
    // backtracking (nr is a squared norm; a NaN fails the comparison and counts as "not decreased")
    const T keep = T(1) - T(1e-4) * lam;
    if (have_trial && !(nr <= nr_old * keep * keep) && lam > T(1.0 / 1024.0) && it < maxit) {
      lam *= T(0.5);
#pragma unroll
      for (int k = 0; k < 6; ++k) G[k] = Gold[k] - lam * d[k];
      continue;
    }
    T dn[6];
    solve6(a, dn);
    T dmax = T(0), gmax = T(1);
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      dmax = fmax(dmax, fabs(dn[k]));
      gmax = fmax(gmax, fabs(G[k]));
      finite = finite && isfinite(dn[k]);
    }
    if (!finite) { status = KR_ST_NONFINITE; break; }
    if (dmax <= S.tol * gmax) { status = KR_ST_CONVERGED; break; }  // this sweep's state is the accepted one
    if (it >= maxit) break;
#pragma unroll
    for (int k = 0; k < 6; ++k) { Gold[k] = G[k]; d[k] = dn[k]; G[k] = G[k] - dn[k]; }
    nr_old = nr;
    lam = T(1);
    have_trial = true;
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) L.Xs[0 * MS_YP + 7 + k] = G[k];
  }
  wave_sync();
  return status;
}

template <typename T, int HS>
__device__ __forceinline__ void ms_ctx_init(const T* cold, const T* hist, const T* tens4, SweepCtx<T, HS>& C) {
  C.hbase = hist;
  C.bufA = nullptr; C.bufB = nullptr; C.astride = 0;
  C.tile = nullptr; C.lane = 0;
  C.tf = {T(0), T(0), T(0)};
#pragma unroll
  for (int t = 0; t < 4; ++t) {  // cosserat_ode.py:195
    const T tt = tens4[t];
    C.tf.x += tt * cold[CD_TDIRS + t * 3 + 0];
    C.tf.y += tt * cold[CD_TDIRS + t * 3 + 1];
    C.tf.z += tt * cold[CD_TDIRS + t * 3 + 2];
  }
  C.fconst = {cold[CD_RHOAG] + C.tf.x, cold[CD_RHOAG + 1] + C.tf.y, cold[CD_RHOAG + 2] + C.tf.z};
}

// boundary conditions at the base (cosserat_ode.py:194): everything but n, m is prescribed
template <typename T>
__device__ __forceinline__ bool ms_base_bc(const T* cold, int r, T& g) {
  if (r >= 7 && r < 13) return false;
  // row order p h | n m | q w; the table stores p0 h0 q0 w0 contiguously
  g = cold[r < 7 ? r : r - 6];
  return true;
}

// ---------------------------------------------------------------------------
// Start-value predictor of one rod (= one wavefront): time levels of the unknowns (interval-start states,
// lane l keeps elements l and 64 + l of the MS_P x 19 vector) and what was learned about them.  The
// persistent kernel keeps it in registers across its time loop; the one-launch-per-step kernel carries it
// from launch to launch through an image in HBM (kr_simulate_batch owns the buffer).
// ---------------------------------------------------------------------------
static_assert(MS_YP == 19, "Xs is indexed as one flat vector by the predictor");
constexpr int MS_NE = MS_P * 19;
constexpr int MS_EPL = (MS_NE + WAVE - 1) / WAVE;
// NC: taps of the fitted recurrence.  3 with the MLP off (a constant plus one oscillation: what a rod driven by smooth
// tensions does, predicted to ~5e-6); 5 with the MLP on - the network makes the response nonlinear, its harmonics need two
// more taps (first guess 5e-3 -> 8e-4 on the fast rods of cfg3, 8e-4 -> 4e-5 on slow ones; measured with the oracle).
#ifndef KR_NN_TAPS
#define KR_NN_TAPS 5
#endif
template <typename T, int NC = 3>
struct MsPred {
  static_assert(NC == 3 || NC == 5 || NC == 7, "three, five or seven taps");
  static constexpr int nc = NC;
  T Hx[MS_EPL][MS_HLEV];  // Hx[q][k]: element lane + 64 q, k steps back
  // adaptive linear predictor: x(t+1) ~ a0 x(t) + a1 (x(t) - x(t-1)) + a2 (x(t) - 2 x(t-1) + x(t-2)) [+ a3, a4 times the
  // third and fourth backward difference] with the coefficients fitted per rod (a = (1, 1, 1) is quadratic extrapolation)
  double lpa[NC];
  bool lp_have;    // lpa was fitted on the previous step (so it can be tested on this one)
  bool lp_good;    // ... and predicted this step to better than 1e-3
  int lp_age;      // steps since lpa was fitted (a good fit is kept for a few steps)
  int avail;       // time levels behind the newest one that carry information
  int next_order;  // extrapolation order of the coming step (MS_ORDER_LP: the linear predictor)
  T kappa;         // contraction constant handed to ms_newton (MsSolveArgs::kappa)
};
// backward differences of the time levels: the basis of the fitted recurrence
template <typename T, int NC>
__device__ __forceinline__ void lp_basis(const T (&H)[MS_HLEV], double (&b)[NC]) {
  const double h0 = (double)H[0], h1 = (double)H[1], h2 = (double)H[2];
  b[0] = h0; b[1] = h0 - h1; b[2] = h0 - 2.0 * h1 + h2;
  if constexpr (NC >= 5) {
    const double h3 = (double)H[3], h4 = (double)H[4];
    b[3] = h0 - 3.0 * h1 + 3.0 * h2 - h3;
    b[4] = h0 - 4.0 * h1 + 6.0 * h2 - 4.0 * h3 + h4;
    if constexpr (NC == 7) {
      const double h5 = (double)H[5], h6 = (double)H[6];
      b[5] = h0 - 5.0 * h1 + 10.0 * h2 - 10.0 * h3 + 5.0 * h4 - h5;
      b[6] = h0 - 6.0 * h1 + 15.0 * h2 - 20.0 * h3 + 15.0 * h4 - 6.0 * h5 + h6;
    }
  }
}
template <int NC>
__device__ __forceinline__ double lp_eval(const double (&a)[NC], const double (&b)[NC]) {
  double s = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];  // (the order of the three-tap version)
  if constexpr (NC >= 5) s += a[3] * b[3] + a[4] * b[4];
  if constexpr (NC == 7) s += a[5] * b[5] + a[6] * b[6];
  return s;
}
// normal equations of the weighted fit: upper triangle row by row, then the right-hand side
template <int NC>
constexpr int lp_nsum() { return NC * (NC + 1) / 2 + NC; }
template <int NC>
__device__ __forceinline__ void lp_accumulate(double (&Sn)[NC * (NC + 1) / 2 + NC], const double (&b)[NC], double x, double w) {
  double wb[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) wb[i] = b[i] * w;
  const double xw = x * w;
  int k = 0;
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int j = i; j < NC; ++j) Sn[k++] += wb[i] * wb[j];
#pragma unroll
  for (int i = 0; i < NC; ++i) Sn[k++] += wb[i] * xw;
}
// (N + lam diag N) a = r + lam diag(N) 1: ridge towards polynomial extrapolation, relative per column; symmetric positive
// definite: elimination without pivoting.  false: no usable fit.
template <int NC>
__device__ __forceinline__ bool lp_solve(const double (&Sn)[NC * (NC + 1) / 2 + NC], double (&a)[NC]) {
  const double lam = 1e-12;
  double m[NC][NC + 1];
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int j = i; j < NC; ++j) { m[i][j] = Sn[k]; m[j][i] = Sn[k]; ++k; }
#pragma unroll
    for (int i = 0; i < NC; ++i) { m[i][NC] = Sn[k++] + lam * m[i][i]; m[i][i] *= (1 + lam); }
  }
  double inv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    inv[c] = fast_rcp(m[c][c]);
#pragma unroll
    for (int i = c + 1; i < NC; ++i) {
      const double f = m[i][c] * inv[c];
#pragma unroll
      for (int j = c + 1; j <= NC; ++j) m[i][j] -= f * m[c][j];
    }
  }
  bool ok = true;
  double bound = 4.0;
#pragma unroll
  for (int i = NC - 1; i >= 0; --i) {
    double r = m[i][NC];
#pragma unroll
    for (int j = i + 1; j < NC; ++j) r -= m[i][j] * a[j];
    a[i] = r * inv[i];
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) { ok = ok && isfinite(a[i]) && fabs(a[i]) < bound; bound *= 4.0; }
  return ok;
}

constexpr int MS_PRED_ROWS = MS_EPL * MS_HLEV + 8;  // doubles per lane of the HBM image
static_assert((size_t)MS_PRED_ROWS * WAVE == KR_PRED_IMG_DOUBLES, "kr_internal.hpp sizes the image buffer");

template <typename T, int NC>
__device__ __forceinline__ void ms_pred_init(MsPred<T, NC>& Q, int lane, const MsRole& R, const T* s0, const T* sp,
                                             bool has_prev, int predictor) {
#pragma unroll
  for (int q = 0; q < MS_EPL; ++q) {
    const int e = lane + q * WAVE;
    const int i = e < MS_NE ? e / 19 : 0, r = e < MS_NE ? e - i * 19 : 0;
    const size_t off = (size_t)ms_interval_start(i, R.sbase, R.srem) * KR_SLOTS + ms_slot_of_yrow(r);
    Q.Hx[q][0] = s0[off];
#pragma unroll
    for (int k = 1; k < MS_HLEV; ++k) Q.Hx[q][k] = sp[off];
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) Q.lpa[k] = 1.0;
  Q.lp_have = Q.lp_good = false;
  Q.lp_age = 0;
  Q.avail = has_prev ? 1 : 0;
  Q.next_order = Q.avail < predictor ? Q.avail : predictor;
  if (Q.next_order >= MS_HLEV) Q.next_order = MS_HLEV - 1;
  Q.kappa = T(0);
}
template <typename T, int NC>
__device__ __forceinline__ void ms_pred_save(const MsPred<T, NC>& Q, double* img, int lane) {
#pragma unroll
  for (int q = 0; q < MS_EPL; ++q)
#pragma unroll
    for (int k = 0; k < MS_HLEV; ++k) img[(q * MS_HLEV + k) * WAVE + lane] = (double)Q.Hx[q][k];
  double* u = img + MS_EPL * MS_HLEV * WAVE;
  u[0 * WAVE + lane] = Q.lpa[0]; u[1 * WAVE + lane] = Q.lpa[1]; u[2 * WAVE + lane] = Q.lpa[2];
  u[3 * WAVE + lane] = (double)Q.kappa;
  u[4 * WAVE + lane] = (double)Q.avail; u[5 * WAVE + lane] = (double)Q.next_order;
  u[6 * WAVE + lane] = (Q.lp_have ? 1.0 : 0.0) + 2.0 * (double)Q.lp_age;
  // (row 7: lp_good, and in lanes 1, 2 the fourth and fifth tap - everything here is uniform over the wavefront)
  double r7 = Q.lp_good ? 1.0 : 0.0;
  if constexpr (NC >= 5) r7 = lane == 1 ? Q.lpa[3] : lane == 2 ? Q.lpa[4] : r7;
  if constexpr (NC == 7) r7 = lane == 3 ? Q.lpa[5] : lane == 4 ? Q.lpa[6] : r7;
  u[7 * WAVE + lane] = r7;
}
template <typename T, int NC>
__device__ __forceinline__ void ms_pred_load(MsPred<T, NC>& Q, const double* img, int lane) {
#pragma unroll
  for (int q = 0; q < MS_EPL; ++q)
#pragma unroll
    for (int k = 0; k < MS_HLEV; ++k) Q.Hx[q][k] = (T)img[(q * MS_HLEV + k) * WAVE + lane];
  const double* u = img + MS_EPL * MS_HLEV * WAVE;
  Q.lpa[0] = u[0 * WAVE + lane]; Q.lpa[1] = u[1 * WAVE + lane]; Q.lpa[2] = u[2 * WAVE + lane];
  Q.kappa = (T)u[3 * WAVE + lane];
  // (uniform values: readfirstlane keeps them, and the control flow that depends on them, scalar)
  Q.avail = __builtin_amdgcn_readfirstlane((int)u[4 * WAVE + lane]);
  Q.next_order = __builtin_amdgcn_readfirstlane((int)u[5 * WAVE + lane]);
  const int hv = __builtin_amdgcn_readfirstlane((int)u[6 * WAVE + lane]);
  Q.lp_have = (hv & 1) != 0;
  Q.lp_age = hv >> 1;
  Q.lp_good = __builtin_amdgcn_readfirstlane((int)u[7 * WAVE + 0]) != 0;
  if constexpr (NC >= 5) { Q.lpa[3] = u[7 * WAVE + 1]; Q.lpa[4] = u[7 * WAVE + 2]; }
  if constexpr (NC == 7) { Q.lpa[5] = u[7 * WAVE + 3]; Q.lpa[6] = u[7 * WAVE + 4]; }
}

// writes the start values of the coming step into Xs (boundary rows of interval 0 from the cold table)
template <typename T, int NC>
__device__ __forceinline__ void ms_pred_guess(const MsPred<T, NC>& Q, int order, int lane, const T* cold, T* Xs) {
#pragma unroll
      for (int q = 0; q < MS_EPL; ++q) {
        const int e = lane + q * WAVE;
        if (e < MS_NE) {
          const int i = e / 19, r = e - i * 19;
          T g;
          if (order == MS_ORDER_LP) {
            double b[NC];
            lp_basis<T, NC>(Q.Hx[q], b);
            g = (T)lp_eval<NC>(Q.lpa, b);
          } else {
            g = extrapolate_n<T>(order, Q.Hx[q]);
          }
          if (i == 0) {
            T bc;
            if (ms_base_bc(cold, r, bc)) g = bc;
          }
          Xs[i * MS_YP + r] = g;
        }
      }
}

// after a step: which predictor would have predicted it best (-> Q.next_order), refit the linear
// predictor, shift the time levels.  `order` is what the step just solved started from.
template <typename T, int NC>
__device__ __forceinline__ void ms_pred_update(MsPred<T, NC>& Q, int order, int status, int predictor, int lane,
                                               const T* Xs, MsStamps& stamps) {
    // ---- extrapolation order of the next step: the one that would have predicted this step best ----
    // (smooth inputs climb to the highest order; after a jump in the controls the low orders win
    // until the jump has left the stencil; fresh random controls every step stay at order 0, which
    // is the reference's warm start, knode.py:89)
    {
#ifdef KR_MS_STAMPS
      unsigned long long tpu;
      KR_STAMP(tpu);
#endif
      // (while the fitted recurrence below is in use and predicts to better than 1e-3 the polynomial orders
      // are not evaluated at all: their errors would only be compared with a much smaller one)
      const bool poly_eval = !(order == MS_ORDER_LP && Q.lp_good);
      float err[MS_HLEV];
#pragma unroll
      for (int p = 0; p < MS_HLEV; ++p) err[p] = 0.f;
      if (poly_eval) {
#pragma unroll
        for (int q = 0; q < MS_EPL; ++q) {
          const int e = lane + q * WAVE;
          if (e < MS_NE) {
            const T x = Xs[e];
#pragma unroll
            for (int p = 0; p < MS_HLEV; ++p) err[p] = fmaxf(err[p], update_ratio(x - extrapolate_n<T>(p, Q.Hx[q]), x));
          }
        }
      }
      // The unknowns of a rod driven by smooth inputs follow, over a few steps, a linear recurrence (a constant
      // plus one dominant oscillation is reproduced exactly by three taps), which predicts far better than
      // any fixed polynomial when the motion is fast against the time step: weighted least squares over the
      // MS_P x 19 components for "this step from the previous three", tested on the step after.
      float err_lp = 0.f;
      if (Q.lp_have) {
#pragma unroll
        for (int q = 0; q < MS_EPL; ++q) {
          const int e = lane + q * WAVE;
          if (e < MS_NE) {
            const double x = (double)Xs[e];
            double b[NC];
            lp_basis<T, NC>(Q.Hx[q], b);
            err_lp = fmaxf(err_lp, update_ratio(x - lp_eval<NC>(Q.lpa, b), x));
          }
        }
      }
      const float em_lp = wave_max_nonneg(err_lp);
#ifdef KR_MS_STAMPS
      KR_STAMP_ADD(stamps.a1, tpu);  // errors of the predictors in use
#endif
      const bool lp_tested = Q.lp_have;
      // a fit that just predicted to better than 1e-3 is kept for up to four steps (its test on the following
      // steps stays out of sample); otherwise refit now
      const bool keep_fit = lp_tested && em_lp < 1.0e-3f && Q.lp_age < 3 && status == KR_ST_CONVERGED;
      Q.lp_age = keep_fit ? Q.lp_age + 1 : 0;
      Q.lp_have = keep_fit;
      if (!keep_fit && predictor >= MS_ORDER_LP && Q.avail >= NC - 1 && status == KR_ST_CONVERGED) {
        double Sn[lp_nsum<NC>()];
#pragma unroll
        for (int k = 0; k < lp_nsum<NC>(); ++k) Sn[k] = 0.0;
#pragma unroll
        for (int q = 0; q < MS_EPL; ++q) {
          const int e = lane + q * WAVE;
          if (e < MS_NE) {
            const double x = (double)Xs[e];
            double b[NC];
            lp_basis<T, NC>(Q.Hx[q], b);
            const double w = (double)__builtin_amdgcn_rcpf(fmaxf(fabsf((float)x), 1.0f));
            lp_accumulate<NC>(Sn, b, x, w);
          }
        }
#pragma unroll
        for (int k = 0; k < lp_nsum<NC>(); ++k) Sn[k] = wave_sum_f64(Sn[k]);
        double a[NC];
        if (lp_solve<NC>(Sn, a)) {
#pragma unroll
          for (int k = 0; k < NC; ++k) Q.lpa[k] = a[k];
          Q.lp_have = true;
        }
      }
#ifdef KR_MS_STAMPS
      KR_STAMP_ADD(stamps.a2, tpu);  // refit (every fourth step)
#endif
      const int pmax = Q.avail < predictor ? Q.avail : (predictor < MS_HLEV ? predictor : MS_HLEV - 1);  // orders the history supported
      float em[MS_HLEV];
#pragma unroll
      for (int p = 0; p < MS_HLEV; ++p) em[p] = poly_eval ? wave_max_nonneg(err[p]) : 3.0e38f;
      // best order, at most two above the one just used (chance hits on rough data - where the errors
      // grow with the order - do not add up to a high order; on smooth data the errors of neighbouring
      // orders can tie, every second one gains)
      int nxt = 0;
      float eb = em[0];
#pragma unroll
      for (int p = 1; p < MS_HLEV; ++p)
        if (p <= pmax && p <= order + 2 && em[p] < eb) { eb = em[p]; nxt = p; }
      if (nxt == Q.avail && Q.avail + 1 < MS_HLEV && Q.avail + 1 <= predictor && nxt >= order) nxt = Q.avail + 1;  // history still growing
      if (!poly_eval) nxt = pmax;  // not measured this step: what smooth data would have picked
      Q.next_order = nxt;
      // the fitted recurrence takes over when its previous fit predicted this step better than every polynomial
      Q.lp_good = lp_tested && Q.lp_have && em_lp < 1.0e-3f;
      if (lp_tested && Q.lp_have && (em_lp < eb || Q.lp_good)) Q.next_order = MS_ORDER_LP;
#ifdef KR_MS_STAMPS
      stamps.osum += (unsigned long long)order; stamps.olast = order;
      for (int p = 0; p < MS_HLEV; ++p) stamps.em[p] = em[p];
#endif
      if (status != KR_ST_CONVERGED) {  // start over from the plain warm start
        Q.next_order = 0;
        Q.avail = -1;  // becomes 0 below
      }
#ifdef KR_MS_STAMPS
      KR_STAMP_ADD(stamps.a3, tpu);  // choice of the next order
#endif
    }
#pragma unroll
    for (int q = 0; q < MS_EPL; ++q) {
      const int e = lane + q * WAVE;
#pragma unroll
      for (int k = MS_HLEV - 1; k > 0; --k) Q.Hx[q][k] = Q.Hx[q][k - 1];
      if (e < MS_NE) Q.Hx[q][0] = Xs[e];  // MS_YP == 19: Xs is the same flat vector
    }
  if (Q.avail < MS_HLEV - 1) ++Q.avail;
}

// ---------------------------------------------------------------------------
// one time step per launch (kr_step_batch, and kr_simulate_batch when the persistent form does not apply)
// ---------------------------------------------------------------------------
template <typename T, bool DIAG, int SCHEME, int HS, bool NN>
__global__ __launch_bounds__(WAVE * MS_WPB) void ms_step_kernel(const RodConst<T> Pc, const StepArgs<T> A,
                                                                const MlpDev<T> M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = Pc.N;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wv = threadIdx.x / WAVE;
  const int64_t rod = (int64_t)blockIdx.x * (blockDim.x / WAVE) + wv;  // 4, 2 or 1 rods per workgroup (ms_wpb)
  if (rod >= A.B) return;  // whole wavefront; there is no workgroup barrier in this kernel
  const size_t rod_elems = (size_t)N * KR_SLOTS;
  const MsLds<T> L = ms_carve<T, HS>(reinterpret_cast<T*>(smem_raw) + (size_t)wv * ms_lds_elems<T, HS>(N, false, NN), N, false);
  const MsRole R = ms_role(lane, N);
  ms_cold_fill<T>(Pc, L.cold, lane);
  wave_sync();

  // history terms (knode.py:74-75)
  for (int j = lane; j < N; j += WAVE) {
    const size_t off = rod * rod_elems + (size_t)j * KR_SLOTS;
    build_hist_point<T, HS>(Pc, A.hc1, A.hc2, A.cur + off, A.prev + off, L.hist + (size_t)j * HS);
  }
  SweepCtx<T, HS> C;
  ms_ctx_init<T, HS>(L.cold, L.hist, A.tens + rod * A.tens_stride, C);
  if constexpr (NN) {
    // exchange tile of the matrix-core MLP: used during sweeps only, so it lives on top of the
    // condensation buffers (XB .. Es), which are used between sweeps only
    C.tile = L.XB;
    C.lane = lane;
    C.role.iv = R.iv; C.role.col = R.col; C.role.idle = R.idle; C.role.jvp = true;
  }
  MsSolveArgs<T> S;
  {  // z of the last grid point is never touched by a sweep
    const T* cl = A.cur + rod * rod_elems + (size_t)(N - 1) * KR_SLOTS;
    S.vlast = {cl[SL_V], cl[SL_V + 1], cl[SL_V + 2]};
    S.ulast = {cl[SL_U], cl[SL_U + 1], cl[SL_U + 2]};
  }
  S.out_rod = A.next + rod * rod_elems;
  S.tip = A.tip ? A.tip + rod * A.tip_stride : nullptr;
  S.tol = A.tol; S.tolA = A.tolA; S.fd_eps = A.fd_eps; S.maxit = A.maxit;
  S.kappa = T(0);
  S.quick_ok = A.residual_test != 0;
  int it, status;
  MsStamps stamps;
  if (A.pred) {
    // kr_simulate_batch, one launch per step: the predictor of the persistent kernel, carried from launch to
    // launch through its image in HBM
    double* img = A.pred + (size_t)rod * MS_PRED_ROWS * WAVE;
    MsPred<T, NN ? KR_NN_TAPS : 3> Q;
    if (A.pred_reset) ms_pred_init<T>(Q, lane, R, A.cur + rod * rod_elems, A.prev + rod * rod_elems, A.pred_has_prev != 0, A.pred_limit);
    else ms_pred_load<T>(Q, img, lane);
    S.kappa = Q.kappa;
    int order = Q.next_order;
    while (true) {
      ms_pred_guess<T>(Q, order, lane, L.cold, L.Xs);
      wave_sync();
      if (order <= 0 && lane < 6) L.Xs[0 * MS_YP + 7 + lane] = A.G[rod * 6 + lane];  // caller's guess (knode.py:67,89)
      wave_sync();
      status = ms_newton<T, DIAG, SCHEME, HS, false, NN>(Pc, M, L, R, lane, C, S, it, stamps);
      if (status == KR_ST_CONVERGED || order == 0) break;
      order = 0;  // the predicted start did not converge: redo the step from the reference's warm start
    }
    if (status != KR_ST_CONVERGED) {  // plain Newton failed from the warm start too: damped single shooting from there
      ms_pred_guess<T>(Q, 0, lane, L.cold, L.Xs);
      wave_sync();
      if (lane < 6) L.Xs[0 * MS_YP + 7 + lane] = A.G[rod * 6 + lane];
      wave_sync();
      const int it_plain = it;  // `iters` reports the plain and the damped phase together (knode_rod.h)
      status = ss_newton_damped<T, DIAG, SCHEME, HS, false, NN>(Pc, M, L, R, lane, C, S, it);
      it += it_plain;
    }
    ms_pred_update<T>(Q, order, status, A.pred_limit, lane, L.Xs, stamps);
    Q.kappa = S.kappa;
    ms_pred_save<T>(Q, img, lane);
  } else {
    // initial guess: time extrapolation of the previous states (kr_step_batch)
    for (int e = lane; e < MS_P * 19; e += WAVE) {
      const int i = e / 19, r = e - i * 19;
      const int sj = ms_interval_start(i, R.sbase, R.srem);
      const size_t off = rod * rod_elems + (size_t)sj * KR_SLOTS + ms_slot_of_yrow(r);
      T g = extrapolate<T>(A.pred_order, A.cur[off], A.prev[off], A.prev2 ? A.prev2[off] : T(0));
      if (i == 0) {
        T bc;
        if (ms_base_bc(L.cold, r, bc)) g = bc;
        else if (A.pred_order <= 0) g = A.G[rod * 6 + (r - 7)];  // caller's guess unless extrapolated
      }
      L.Xs[i * MS_YP + r] = g;
    }
    wave_sync();
    status = ms_newton<T, DIAG, SCHEME, HS, false, NN>(Pc, M, L, R, lane, C, S, it, stamps);
    if (status != KR_ST_CONVERGED) {  // damped single shooting from the caller's guess
      if (lane < 6) L.Xs[0 * MS_YP + 7 + lane] = A.G[rod * 6 + lane];
      wave_sync();
      const int it_plain = it;  // `iters` reports the plain and the damped phase together (knode_rod.h)
      status = ss_newton_damped<T, DIAG, SCHEME, HS, false, NN>(Pc, M, L, R, lane, C, S, it);
      it += it_plain;
    }
  }
  if (lane < 6) A.G[rod * 6 + lane] = L.Xs[0 * MS_YP + 7 + lane];
  if (lane == 0) {
    if (A.status) A.status[rod * A.st_stride] = status;
    if (A.iters) A.iters[rod * A.st_stride] = it;
  }
}

// ---------------------------------------------------------------------------
// persistent form: one launch runs all T steps of kr_simulate_batch; a wavefront keeps its rod
// ---------------------------------------------------------------------------
constexpr int MS_NPL = 2;  // grid points per lane held in registers by the persistent kernel (N <= 128)

// OCC: workgroups the register allocation must leave room for on a CU (__launch_bounds__).  With OCC = 2 the fp32
// instantiation runs two wavefronts per SIMD - two rods - whose issue-bound sweeps and latency-bound algebra overlap:
// +25 % (N = 100) to +37 % (N = 64) at B >= 2048 (tools/occ_probe.py), -3 % at B = 1024 where every SIMD has one
// wavefront anyway (68 spilled registers).  In fp64 a rod needs 37.8 KB of LDS: four per CU, one per SIMD.
template <typename T, bool DIAG, int SCHEME, int HS, bool NN, int OCC = 1>
__global__ __launch_bounds__(WAVE * MS_WPB, OCC) void ms_sim_kernel(const RodConst<T> Pc, const SimArgs<T> A, const MlpDev<T> M) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = Pc.N;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wv = threadIdx.x / WAVE;
  const int64_t rod = (int64_t)blockIdx.x * MS_WPB + wv;
  if (rod >= A.B) return;  // whole wavefront; there is no workgroup barrier in this kernel
  // second launch of the two-launch form (kr_mso_impl.hpp): take over where the overlapped kernel stopped - a rod
  // it finished leaves at once
  const int64_t t0 = A.resume ? (int64_t)A.resume[rod] : 0;
  if (t0 >= A.T_steps) return;
  const bool resumed = t0 > 0;
  const size_t rod_elems = (size_t)N * KR_SLOTS;
  const MsLds<T> L = ms_carve<T, HS>(reinterpret_cast<T*>(smem_raw) + (size_t)wv * ms_lds_elems<T, HS>(N, true, NN), N, true, NN);
  const MsRole R = ms_role(lane, N);
  ms_cold_fill<T>(Pc, L.cold, lane);
  if constexpr (NN) {  // p-column blocks of the Jacobian (ms_newton): none known yet
    for (int e = lane; e < MS_BP_ELEMS; e += WAVE) L.Bp[e] = T(0);
  }
  wave_sync();

  // leading slots (q w v u) of the newest state live in LDS (c12), those of the state before it in
  // registers (regP), both indexed lane-per-grid-point: the BDF2 history of the next step never
  // touches HBM
  T regP[MS_NPL][12];
  const T* s0 = A.states + (A.ring ? t0 % 3 : t0) * A.slot_elems + rod * rod_elems;
  const T* sp = resumed ? A.states + (A.ring ? (t0 - 1) % 3 : t0 - 1) * A.slot_elems + rod * rod_elems
                        : (A.prev_init ? A.prev_init + rod * rod_elems : s0);
#pragma unroll
  for (int q = 0; q < MS_NPL; ++q) {
    const int j = lane + q * WAVE;
    if (j < N) {
      T cv[12];
      load_hist_vec<T, 12>(s0 + (size_t)j * KR_SLOTS, cv);
      store_vec<T, 12>(L.c12 + (size_t)j * 12, cv);
      load_hist_vec<T, 12>(sp + (size_t)j * KR_SLOTS, regP[q]);
    } else {
#pragma unroll
      for (int c = 0; c < 12; ++c) regP[q][c] = T(0);
    }
  }
  MsPred<T, NN ? KR_NN_TAPS : 3> Q;
  double* img = A.pred_io ? A.pred_io + (size_t)rod * MS_PRED_ROWS * WAVE : nullptr;
  if (img && A.pred_load && !resumed) ms_pred_load<T>(Q, img, lane);
  else ms_pred_init<T>(Q, lane, R, s0, sp, resumed || A.prev_init != nullptr, A.predictor);
  MsSolveArgs<T> S;
  {
    const T* cl = s0 + (size_t)(N - 1) * KR_SLOTS;
    S.vlast = {cl[SL_V], cl[SL_V + 1], cl[SL_V + 2]};
    S.ulast = {cl[SL_U], cl[SL_U + 1], cl[SL_U + 2]};
  }
  S.tol = A.tol; S.tolA = A.tolA; S.fd_eps = A.fd_eps; S.maxit = A.maxit;
  S.kappa = Q.kappa;
  S.quick_ok = A.residual_test != 0;
  S.lowp_allowed = NN && sizeof(T) == 8 && A.nn_lowp != 0 && M.f32_ok != 0;
  S.bo_allowed = NN && A.nn_base_only != 0 && (sizeof(T) == 8 ? M.n_layers == 3 && M.otiles[1] == 4 : M.f32_ok != 0);
  // from the straight rod / a fresh call the first steps take three sweeps and more; a call that resumes a steady trajectory
  // through the predictor image starts like the steps it continues (full-precision first sweep) until a step says otherwise
  S.lowp_first = !(img && A.pred_load && !resumed);
  T Gguess = lane < 6 ? A.G[rod * 6 + lane] : T(0);
  const T* ctl = A.ctl + rod * A.T_steps * 4;
  T tens[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) tens[k] = ctl[t0 * 4 + k];
  wave_sync();

  MsStamps stamps;
#ifdef KR_MS_STAMPS
  unsigned long long t_begin, tp;
  KR_STAMP(t_begin);
#endif
#ifdef KR_MS_STAMPS
  // clock trace (tools/clock_probe.py): the host asks for it by putting a magic number into the unused slot 15 of the
  // last rod's row and has then allocated 2 T more slots behind the B rows; rod 0 records, at the start of every
  // step, the shader-clock counter and the constant 100 MHz counter
  const bool clock_trace = A.dbg && rod == 0 && A.dbg[(A.B - 1) * 24 + 15] == 0xC10CULL;
#endif
  for (int64_t t = t0; t < A.T_steps; ++t) {
#ifdef KR_MS_STAMPS
    KR_STAMP(tp);
    if (clock_trace && lane == 0) {
      A.dbg[A.B * 24 + 2 * t] = tp;
      A.dbg[A.B * 24 + 2 * t + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // ---- history records from c12 (newest) and regP (the one before) ----------
#pragma unroll
    for (int q = 0; q < MS_NPL; ++q) {
      const int j = lane + q * WAVE;
      if (j < N) {
        T cv[12];
        load_hist_vec<T, 12>(L.c12 + (size_t)j * 12, cv);
        build_hist_cold<T, HS, DIAG>(L.cold, A.hc1, A.hc2, cv, regP[q], L.hist + (size_t)j * HS);
#pragma unroll
        for (int c = 0; c < 12; ++c) regP[q][c] = cv[c];
      }
    }
    SweepCtx<T, HS> C;
    ms_ctx_init<T, HS>(L.cold, L.hist, tens, C);
    if constexpr (NN) {  // exchange tile of the matrix-core MLP, on top of the condensation buffers
      C.tile = L.XB;
      C.lane = lane;
      C.role.iv = R.iv; C.role.col = R.col; C.role.idle = R.idle; C.role.jvp = true;
    }
    if (t + 1 < A.T_steps) {  // next step's tensions: issued now, consumed after this step's solve
#pragma unroll
      for (int k = 0; k < 4; ++k) tens[k] = ctl[(t + 1) * 4 + k];
    }
    const int64_t inx = A.ring ? (t + 1) % 3 : t + 1;
    S.out_rod = A.states + inx * A.slot_elems + rod * rod_elems;
    S.tip = A.tip ? A.tip + (rod * A.T_steps + t) * 3 : nullptr;
    // ---- initial guess and solve ---------------------------------------------------
    int order = Q.next_order;
    int status, it;
    while (true) {
      ms_pred_guess<T>(Q, order, lane, L.cold, L.Xs);
      wave_sync();
      if (order <= 0 && lane < 6) L.Xs[0 * MS_YP + 7 + lane] = Gguess;  // caller's guess (knode.py:67,89)
      wave_sync();
#ifdef KR_MS_STAMPS
      KR_STAMP_ADD(stamps.prep, tp);
#endif
      status = ms_newton<T, DIAG, SCHEME, HS, true, NN>(Pc, M, L, R, lane, C, S, it, stamps);
      if (status == KR_ST_CONVERGED || order == 0) break;
      order = 0;  // the extrapolated start did not converge: redo the step from the reference's warm start
#ifdef KR_MS_STAMPS
      stamps.retries += 1;
      KR_STAMP(tp);
#endif
    }
    if (status != KR_ST_CONVERGED) {  // plain Newton failed from the warm start too: damped single shooting from there
      if (lane < 6) L.Xs[0 * MS_YP + 7 + lane] = Gguess;
      wave_sync();
      const int it_plain = it;  // `iters` reports the plain and the damped phase together (knode_rod.h)
      status = ss_newton_damped<T, DIAG, SCHEME, HS, true, NN>(Pc, M, L, R, lane, C, S, it);
      it += it_plain;
    }
    if (lane == 0 && A.status) A.status[rod * A.T_steps + t] = status;
    ms_pred_update<T>(Q, order, status, A.predictor, lane, L.Xs, stamps);
    if (lane < 6) Gguess = L.Xs[0 * MS_YP + 7 + lane];
    wave_sync();
  }
  if (lane < 6) A.G[rod * 6 + lane] = Gguess;
  if (img) {
    Q.kappa = S.kappa;
    ms_pred_save<T>(Q, img, lane);
  }
#ifdef KR_MS_STAMPS
  if (lane == 0 && A.dbg) {
    unsigned long long te;
    KR_STAMP(te);
    unsigned long long* d = A.dbg + rod * 24;
    for (int p = 0; p < 8; ++p) d[16 + p] = (unsigned long long)__double_as_longlong((double)stamps.em[p]);
    d[8] = stamps.a1; d[9] = stamps.a2; d[10] = stamps.a3; d[11] = stamps.a4;
    d[12] = stamps.osum; d[13] = (unsigned long long)stamps.olast; d[14] = stamps.retries;
#ifdef KR_QUICK_AUDIT
    d[15] = (unsigned long long)__double_as_longlong(stamps.qa);  // worst (chord update) / (residual estimate)
#else
    d[15] = (unsigned long long)__double_as_longlong(stamps.qn);  // sweeps accepted by the residual test
#endif
    d[0] = te - t_begin; d[1] = stamps.sweep; d[2] = stamps.alg; d[3] = stamps.prep; d[4] = (unsigned long long)stamps.its;
    for (int k = 0; k < 3; ++k) d[5 + k] = (unsigned long long)__double_as_longlong(stamps.dn[k]);
  }
#endif
}

template <typename T, int HS>
static size_t ms_lds_bytes(int N, bool persist = false, bool nn = false, int wpb = MS_WPB) {
  return sizeof(T) * ms_lds_elems<T, HS>(N, persist, nn) * wpb;
}
// rods per workgroup of the per-step kernel: as many as the LDS history of N grid points allows (long rods
// get 2 or 1, e.g. fp64 N=400 needs 58 KB per rod); 0 = does not fit at all
template <typename T, int HS>
static int ms_wpb(int N, bool nn, size_t lds_limit) {
  for (int w = MS_WPB; w >= 1; w >>= 1)
    if (ms_lds_bytes<T, HS>(N, false, nn, w) <= lds_limit) return w;
  return 0;
}

template <typename T, bool DIAG, int SCHEME, bool NN>
static int launch_ms_inst(const RodConst<T>& P, const MlpDev<T>& M, const StepArgs<T>& a, size_t lds_limit,
                          hipStream_t s) {
  auto kern = ms_step_kernel<T, DIAG, SCHEME, hs_phys<T>(), NN>;
  const int wpb = ms_wpb<T, hs_phys<T>()>(P.N, NN, lds_limit);
  if (wpb <= 0) { set_error("multiple-shooting kernel: history of N grid points does not fit in LDS"); return KR_E_ARG; }
  const size_t smem = ms_lds_bytes<T, hs_phys<T>()>(P.N, false, NN, wpb);
  if (int rc_lds_ = dyn_lds(reinterpret_cast<const void*>(kern), smem)) return rc_lds_;
  hipLaunchKernelGGL(kern, dim3((unsigned)((a.B + wpb - 1) / wpb)), dim3(WAVE * wpb), smem, s, P, a, M);
  KR_HIP(hipGetLastError());
  return KR_OK;
}

// true if the multiple-shooting kernel can and should take this call
template <typename T>
static bool ms_eligible(kr_handle* h, int use_nn, const StepArgs<T>& a) {
  const RodConst<T>& P = consts<T>(h);
  if (a.mode != 0) return false;
  if (use_nn) {  // the MLP must be one the matrix-core evaluator serves (no per-lane activation buffers here)
    const MlpDev<T>& M = mlpdev<T>(h);
    if (M.n_layers <= 0 || !M.mfma_ok || h->params.nn_input_history) return false;
  }
  if (h->ms_mode == 0) return false;
  if (P.N - 1 < 2 * MS_P) return false;                 // too few segments to cut
  if (ms_wpb<T, hs_phys<T>()>(P.N, use_nn != 0, (size_t)h->lds_limit) <= 0) return false;
  if (h->ms_mode == 1) return true;                     // forced
  return a.B <= (int64_t)h->ms_batch_limit;             // auto (no limit by default: faster at every batch size)
}

template <typename T, bool NN>
static int launch_ms_nn(kr_handle* h, int scheme, const StepArgs<T>& a, hipStream_t s) {
  const RodConst<T>& P = consts<T>(h);
  const MlpDev<T>& M = mlpdev<T>(h);
  if (scheme == KR_EULER)
    return P.diag ? launch_ms_inst<T, true, KR_EULER, NN>(P, M, a, (size_t)h->lds_limit, s)
                  : launch_ms_inst<T, false, KR_EULER, NN>(P, M, a, (size_t)h->lds_limit, s);
  if (scheme == KR_RK4)
    return P.diag ? launch_ms_inst<T, true, KR_RK4, NN>(P, M, a, (size_t)h->lds_limit, s)
                  : launch_ms_inst<T, false, KR_RK4, NN>(P, M, a, (size_t)h->lds_limit, s);
  set_error("unknown scheme");
  return KR_E_ARG;
}
// The one-launch-per-step kernels WITH the MLP live in their own translation units (kr_msn_f32.hip / kr_msn_f64.hip,
// kr_msn_impl.hpp): they are the most register-starved kernels of the library, and hipcc 7.2 places ordinary VGPR
// spills inside the whole-wave-mode brackets it opens to reach its SGPR-spill registers there (LABBOOK.md section 4,
// tools/wwm_spill_scan.py).  Those units are compiled with SGPR spills going to memory instead of VGPR lanes, which
// removes the brackets altogether.
template <typename T>
int launch_ms_step_nn(kr_handle* h, int scheme, const StepArgs<T>& a, hipStream_t s);
template <typename T>
static int launch_ms(kr_handle* h, int scheme, int use_nn, const StepArgs<T>& a, hipStream_t s) {
  h->last_sim_path = 1;
  return use_nn ? launch_ms_step_nn<T>(h, scheme, a, s) : launch_ms_nn<T, false>(h, scheme, a, s);
}
template <typename T, bool DIAG, int SCHEME, bool NN, int OCC = 1>
static int launch_ms_sim_inst(const RodConst<T>& P, const MlpDev<T>& M, const SimArgs<T>& a, hipStream_t s) {
  auto kern = ms_sim_kernel<T, DIAG, SCHEME, hs_phys<T>(), NN, OCC>;
  const size_t smem = ms_lds_bytes<T, hs_phys<T>()>(P.N, true, NN);
  if (int rc_lds_ = dyn_lds(reinterpret_cast<const void*>(kern), smem)) return rc_lds_;
  hipLaunchKernelGGL(kern, dim3((unsigned)((a.B + MS_WPB - 1) / MS_WPB)), dim3(WAVE * MS_WPB), smem, s, P, a, M);
  KR_HIP(hipGetLastError());
  return KR_OK;
}

template <typename T>
static int launch_msw_sim(kr_handle* h, int W, const SimArgs<T>& a, hipStream_t s);  // kr_msw_impl.hpp

// returns 1 when the persistent form does not apply (caller falls back to one launch per step)
template <typename T>
int launch_sim_persistent(kr_handle* h, int scheme, int use_nn, const SimArgs<T>& a, hipStream_t s) {
  const RodConst<T>& P = consts<T>(h);
  if (h->ms_mode == 0 || h->persistent == 0) return 1;
  h->last_overlap = 0;
  if (const int W = step_waves_per_rod<T>(h, scheme, use_nn, a.B, 0)) {  // several wavefronts per rod
    if (h->msw_overlap && P.diag) {
      const int rc = launch_mswo_sim<T>(h, W, a, s);
      if (rc == KR_OK) h->last_overlap = 1;
      if (rc != 1) return rc;
    }
    return launch_msw_sim<T>(h, W, a, s);
  }
  h->last_waves_per_rod = 1;
  const MlpDev<T>& M = mlpdev<T>(h);
  if (use_nn) {
    if (const int W = nn_sim_waves_per_rod<T>(h, scheme, a.B)) {  // several wavefronts per rod, MLP on (kr_mswn_*.hip)
      const int rc = launch_msw_nn_sim<T>(h, W, a, s);
      if (rc != 1) return rc;
    }
    // MLP inside the sweeps: the matrix-core evaluator, Euler sweeps and diagonal material matrices only (one
    // more instantiation of the largest kernel per arithmetic type; everything else takes one launch per step)
    // (the persistent kernel carries the base + JVP evaluator only: a network it does not serve takes one launch per step)
    if (M.n_layers <= 0 || !M.mfma_ok || !M.jvp_ok || h->params.nn_input_history || scheme != KR_EULER || !P.diag) return 1;
    if (P.N - 1 < 2 * MS_P || P.N > MS_NPL * WAVE) return 1;
    if (ms_lds_bytes<T, hs_phys<T>()>(P.N, true, true) > (size_t)h->lds_limit) return 1;
    if (h->ms_mode != 1 && a.B > (int64_t)h->ms_batch_limit) return 1;
    return launch_ms_sim_inst<T, true, KR_EULER, true>(P, M, a, s);
  }
  if (P.N - 1 < 2 * MS_P || P.N > MS_NPL * WAVE) return 1;
  if (ms_lds_bytes<T, hs_phys<T>()>(P.N, true) > (size_t)h->lds_limit) return 1;
  if (h->ms_mode != 1 && a.B > (int64_t)h->ms_batch_limit) return 1;
  h->last_overlap = 0;
  if (scheme == KR_EULER && P.diag && h->overlap) {
    // two launches: the overlapped kernel (one sweep per step in the steady state), then this file's persistent
    // kernel for the rods that left steps behind (a rod that finished exits at once)
    SimArgs<T> a2 = a;
    int rc = ensure_resume(h, a.B);
    if (rc) return rc;
    a2.resume = static_cast<int32_t*>(h->resume_buf);
    rc = launch_mso_sim<T>(h, a2, s);
    if (rc == KR_OK) {
      h->last_overlap = 1;
      return launch_ms_sim_inst<T, true, KR_EULER, false>(P, M, a2, s);
    }
    if (rc != 1) return rc;
  }
  if (scheme == KR_EULER) {
    if constexpr (sizeof(T) == 4) {
      // fp32, more rods than SIMDs, and two workgroups fit the LDS of a CU: the two-wavefronts-per-SIMD instantiation
      if (P.diag && a.B > 1024 && 2 * ms_lds_bytes<T, hs_phys<T>()>(P.N, true) <= (size_t)h->lds_limit)
        return launch_ms_sim_inst<T, true, KR_EULER, false, 2>(P, M, a, s);
    }
    return P.diag ? launch_ms_sim_inst<T, true, KR_EULER, false>(P, M, a, s) : launch_ms_sim_inst<T, false, KR_EULER, false>(P, M, a, s);
  }
  if (scheme == KR_RK4)
    return P.diag ? launch_ms_sim_inst<T, true, KR_RK4, false>(P, M, a, s) : launch_ms_sim_inst<T, false, KR_RK4, false>(P, M, a, s);
  set_error("unknown scheme");
  return KR_E_ARG;
}
}  // namespace kr

#include "kr_msw_impl.hpp"  // several wavefronts per rod (uses everything above)

namespace kr {
// one-time host work of the first launch of the persistent Euler / MLP-off kernel, ahead of time (kr_simulate_prepare)
template <typename T>
int prepare_ms_sim(kr_handle* h) {
  const RodConst<T>& P = consts<T>(h);
  if (!P.diag) return 1;
  auto kern = ms_sim_kernel<T, true, KR_EULER, hs_phys<T>(), false, 1>;
  hipFuncAttributes fa;
  KR_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
  const size_t smem = ms_lds_bytes<T, hs_phys<T>()>(P.N, true);
  if (smem <= (size_t)h->lds_limit) if (int rc_lds_ = dyn_lds(reinterpret_cast<const void*>(kern), smem)) return rc_lds_;
  return KR_OK;
}
}  // namespace kr

#ifndef KR_MS_NO_INST  // (kr_mso_*.hip include this file for its device functions only)
namespace kr {
template int prepare_ms_sim<KR_SIM_T>(kr_handle*);
template int launch_sim_persistent<KR_SIM_T>(kr_handle*, int, int, const SimArgs<KR_SIM_T>&, hipStream_t);
template int step_waves_per_rod<KR_SIM_T>(kr_handle*, int, int, int64_t, int);

KR_INST(KR_SIM_T)

}  // namespace kr
#endif
