// kr_msw_impl.hpp - multiple shooting with SEVERAL wavefronts per rod: msw_step_kernel (one time step per launch)
// and msw_sim_kernel (all steps of kr_simulate_batch in one launch).
//
// kr_ms_impl.hpp gives a rod one wavefront and P = 4 sub-intervals; the dependent chain of a sweep is then
// (N - 1) / 4 grid points long and a batch keeps B of the chip's 1024 SIMDs busy.  Long rods in small batches
// (BASELINE cfg5: N = 400, B = 512) leave half the chip idle behind a 100-point chain.  Here a rod owns a workgroup
// of W wavefronts: wavefront 0 integrates sub-intervals 0..3 exactly as before (7 + 3 x 17 lanes), every further
// wavefront three more (3 x 17 lanes), P = 4 + 3 (W - 1) sub-intervals in all, chain length (N - 1) / P.
//
// The nonlinear system is the same (unknowns G, Y_1 .. Y_{P-1}; continuity E_g(Y_g) = Y_{g+1}; tip condition on the
// last interval), Newton with forward-difference columns, condensed onto the 6 base unknowns:
//   * wavefront 0 chains  X_{g+1} = [c_g | 0] + A_g X_g  (X_g = [a_g | M_g], 16 x 7: dY_g = X_g [1; dG]) over its
//     four intervals, as kr_ms_impl.hpp does;
//   * wavefront w >= 1 does not know dY at its first interval yet: it composes its three intervals into one affine
//     map  dY_out = a + B dY_in  (L_k = [a_k | B_k], 16 x 17:  L_0 = [c | A],  L_{k+1} = [c | 0] + A L_k) - all
//     wavefronts in parallel;
//   * then the boundary blocks are passed down the rod,  X^(w+1) = [a^w | 0] + B^w X^(w)  (one 16x16 by 16x7 product
//     per wavefront, serial, workgroup barriers in between); the last one yields the 6 x 6 system for dG;
//   * back-substitution: every wavefront forms dY at its first interval from X^(w) and dG, its inner ones from L_k.
// The p rows (no equation reads p) are accumulated afterwards from all intervals, like in the one-wavefront kernel.
// Stopping rule, storing sweep prediction, start-value predictor (one MsPred per wavefront over its own unknowns, the
// decisions reduced over the workgroup) and the residual test of a storing sweep are those of kr_ms_impl.hpp; the chord
// check is not used here.
//
// History records are the 12 raw BDF2 terms only (q_h w_h v_h u_h); av / au are re-derived per evaluation (6 FMAs), so
// that two N = 400 rods fit the LDS of a CU in fp64.  Euler sweeps, MLP off (what small batches and long rods run).
#pragma once
// (included by kr_ms_impl.hpp, after its definitions)

namespace kr {

constexpr int HS_LEAN = 12;
template <int W>
struct MswGeo {
  static constexpr int P = 4 + 3 * (W - 1);
};
constexpr int MSW_LT_LD = 20;  // row pitch of the 16 x 17 local-map tiles

struct MswRole {
  int w, iv, ivl, col, s_i, len_i, sbase, comp, l0own, g0, K;
  bool idle;
};
template <int W>
__device__ __forceinline__ MswRole msw_role(int wave, int lane, int N) {
  constexpr int P = MswGeo<W>::P;
  MswRole R;
  R.w = wave;
  if (wave == 0) {
    R.idle = lane >= 58;
    if (lane < 7) { R.ivl = 0; R.col = lane; }
    else if (!R.idle) { R.ivl = 1 + (lane - 7) / 17; R.col = (lane - 7) % 17; }
    else { R.ivl = 0; R.col = 0; }
    R.g0 = 0; R.K = 4;
    R.l0own = R.ivl == 0 ? 0 : 7 + 17 * (R.ivl - 1);
  } else {
    R.idle = lane >= 51;
    R.ivl = R.idle ? 0 : lane / 17;
    R.col = R.idle ? 0 : lane % 17;
    R.g0 = 4 + 3 * (wave - 1); R.K = 3;
    R.l0own = 17 * R.ivl;
  }
  R.iv = R.g0 + R.ivl;
  const int nseg = N - 1;
  R.sbase = nseg / P;
  const int srem = nseg % P;
  R.s_i = R.iv * R.sbase + (R.iv < srem ? R.iv : srem);
  R.len_i = R.sbase + (R.iv < srem ? 1 : 0);
  R.comp = R.col == 0 ? -1 : (R.iv == 0 ? 6 + R.col : 2 + R.col);
  return R;
}
__device__ __forceinline__ constexpr int msw_lpos(int cc) { return cc < 16 ? (cc & 3) * 4 + (cc >> 2) : 16; }  // column -> slot of an L tile row
__device__ __forceinline__ int msw_l0(int wave, int k) { return wave == 0 ? (k == 0 ? 0 : 7 + 17 * (k - 1)) : 17 * k; }
__device__ __forceinline__ int msw_start(int g, int N, int P) {
  const int nseg = N - 1, sbase = nseg / P, srem = nseg % P;
  return g * sbase + (g < srem ? g : srem);
}

// LDS of one rod (elements of T; every array starts at a multiple of 4 elements)
template <typename T, int W>
struct MswLds {
  T* hist;   // [N][HS_LEAN]
  T* Xs;     // [P][19] unknowns
  T* cold;   // [CD_SIZE]
  T* Tm;     // [6][8]
  T* Xbd;    // [W][19][8] boundary blocks X^(w) (rows 3..18), w = 1 .. W-1
  T* dY;     // [P][19] updates of all unknowns
  T* sp;     // [P][4] p-row terms of every interval
  T* red;    // [W][32] reduction scratch
  T* Es;     // [W] blocks of wblk elements: end states Es_w [64][19], then the local-map tiles Lt_w [2][19][MSW_LT_LD]
  size_t wblk;
  T* Bp;     // MLP on: [W][3 intervals][3 directions][19] p-column blocks (+1), see msw_newton; then dP [P][3] (+pad)
  __device__ __forceinline__ T* es(int wave) const { return Es + (size_t)wave * wblk; }
  __device__ __forceinline__ T* lt(int wave) const { return Es + (size_t)wave * wblk + ((64 * 19 + 3) & ~3); }
};
// Block of one wavefront.  With the MLP on it doubles, during a sweep, as the wavefront's scratch of the base + JVP
// evaluator (mlp_jvp.hpp: both are dead then), so it is at least that large.
template <typename T>
__host__ __device__ constexpr size_t msw_wave_block(bool nn) {
  size_t n = ((64 * 19 + 3) & ~3) + 2 * 19 * MSW_LT_LD;
  if (nn && n < mj_scratch_elems<T>()) n = mj_scratch_elems<T>();
  return (n + 3) & ~size_t(3);
}
// hist_lds = false: the history records live in global memory (the persistent kernel with the MLP on)
constexpr int MSW_BP_W = 3 * 3 * 19 + 1;  // p-column blocks of one wavefront
template <typename T, int W>
__host__ __device__ inline size_t msw_lds_elems(int N, bool nn = false, bool hist_lds = true) {
  constexpr int P = MswGeo<W>::P;
  auto r4 = [](size_t n) { return (n + 3) & ~size_t(3); };
  return (hist_lds ? r4((size_t)N * HS_LEAN) : 0) + r4(P * 19) + r4(CD_SIZE) + 48 + r4(W * 19 * 8) + r4(P * 19) + r4(P * 4) +
         W * 32 + (size_t)W * msw_wave_block<T>(nn) + (nn ? (size_t)W * MSW_BP_W + r4(P * 3) : 0);
}
template <typename T, int W>
__device__ __forceinline__ MswLds<T, W> msw_carve(T* smem, int N, bool nn = false, bool hist_lds = true) {
  constexpr int P = MswGeo<W>::P;
  auto r4 = [](size_t n) { return (n + 3) & ~size_t(3); };
  MswLds<T, W> L;
  L.hist = hist_lds ? smem : nullptr;
  L.Xs = smem + (hist_lds ? r4((size_t)N * HS_LEAN) : 0);
  L.cold = L.Xs + r4(P * 19);
  L.Tm = L.cold + r4(CD_SIZE);
  L.Xbd = L.Tm + 48;
  L.dY = L.Xbd + r4(W * 19 * 8);
  L.sp = L.dY + r4(P * 19);
  L.red = L.sp + r4(P * 4);
  L.Es = L.red + W * 32;
  L.wblk = msw_wave_block<T>(nn);
  L.Bp = nn ? L.Es + (size_t)W * L.wblk : nullptr;
  return L;
}

// the MLP inside a sweep (instantiations with NN; unused otherwise)
template <typename T>
struct MswNn {
  const MlpDev<T>* M = nullptr;
  T* tile = nullptr;   // scratch of this wavefront's evaluator calls: its Es / Lt block
  V3<T> tf{T(0), T(0), T(0)};  // tendon force sum, an input of the network (cosserat_ode.py:171-178)
  NnRole role;
};
// Padding slots of a record in the instantiations with the MLP on: a zero the optimiser cannot merge with the other
// zeros of the kernel.  As a shared constant, hipcc 7.2 kept (u.z, 0.0) in a register tuple across the whole kernel and
// spilled it inside a whole-wave-mode bracket it had opened for its SGPR-spill registers - the lanes that were idle
// where the tuple was built put garbage into the slot, and the next user under a full exec mask (a zero address
// offset) faulted.  tools/wwm_spill_scan.py looks for that pattern in the assembly.
template <typename T, bool NN>
__device__ __forceinline__ void msw_record_pad(T (&rec)[KR_SLOTS]) {
  if constexpr (NN) {
    T z;
    if constexpr (sizeof(T) == 8) asm volatile("v_mov_b64 %0, 0" : "=v"(z));
    else asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    rec[25] = z; rec[26] = z; rec[27] = z;
  }
}
template <typename T, int W>
__device__ __forceinline__ MswNn<T> msw_nn_ctx(const MlpDev<T>& M, const MswLds<T, W>& L, const MswRole& R, int lane, V3<T> tf) {
  MswNn<T> nn;
  nn.M = &M;
  nn.tile = L.es(R.w);
  nn.tf = tf;
  nn.role.iv = R.ivl; nn.role.col = R.col; nn.role.idle = R.idle; nn.role.jvp = true;
  // the dx rows nobody owns are zeroed by the lanes without a column: rows 6..15 of sample tile 0 on wavefront 0 (its
  // first interval has 6 columns), sample tile 3 on the others (three intervals)
  if (R.w == 0) nn.role.zrow = 6 + (R.idle ? 4 + (lane - 58) : R.ivl);
  else nn.role.zrow = 48 + (R.idle ? 3 + (lane - 51) : R.ivl);
  return nn;
}

// av, au of a grid point from its raw history (see RodHist)
template <typename T, bool DIAG>
__device__ __forceinline__ RodHist<T> hist_lean(const RodConst<T>& P, const T (&hv)[HS_LEAN]) {
  RodHist<T> h;
  h.qh = {hv[0], hv[1], hv[2]};
  h.wh = {hv[3], hv[4], hv[5]};
  h.vh = {hv[6], hv[7], hv[8]};
  h.uh = {hv[9], hv[10], hv[11]};
  if constexpr (DIAG) {
    h.av = {P.Ksei[0] * (P.Kse_vstar[0] - P.Bse[0] * h.vh.x), P.Ksei[4] * (P.Kse_vstar[1] - P.Bse[4] * h.vh.y),
            P.Ksei[8] * (P.Kse_vstar[2] - P.Bse[8] * h.vh.z)};
    h.au = {-P.Kbti[0] * (P.Bbt[0] * h.uh.x), -P.Kbti[4] * (P.Bbt[4] * h.uh.y), -P.Kbti[8] * (P.Bbt[8] * h.uh.z)};
  } else {
    hist_derive(P, h);
  }
  return h;
}

// Barrier between the wavefronts of a rod for hand-offs that go through LDS only (everything inside a Newton iteration:
// reductions, boundary blocks, updates).  __syncthreads() also waits for every global access in flight (s_waitcnt vmcnt(0)),
// and on gfx950 vmcnt counts STORES too - after a storing sweep that is the whole state record stream on its way to HBM.
__device__ __forceinline__ void msw_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// workgroup-wide reductions (W wavefronts, every one calls; scratch: W slots of T per call site)
template <int W>
__device__ __forceinline__ float msw_max(float v, float* red, int wave, int lane) {
  v = wave_max_nonneg(v);
  if (lane == 0) red[wave] = v;
  msw_lds_barrier();
  float m = red[0];
#pragma unroll
  for (int k = 1; k < W; ++k) m = fmaxf(m, red[k]);
  msw_lds_barrier();
  return m;
}
template <int W>
__device__ __forceinline__ double msw_sum(double v, double* red, int wave, int lane) {
  v = wave_sum_f64(v);
  if (lane == 0) red[wave] = v;
  msw_lds_barrier();
  double s = red[0];
#pragma unroll
  for (int k = 1; k < W; ++k) s += red[k];
  msw_lds_barrier();
  return s;
}

// K sums / K maxima at once (one pair of barriers; red holds W * 32 elements of T >= W * K doubles for K <= 9 ... 16)
template <int W, int K>
__device__ __forceinline__ void msw_sum_n(double (&v)[K], double* red, int wave, int lane) {
  static_assert(K <= 16, "scratch of the workgroup reductions");
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum_f64(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[wave * K + k] = v[k];
  }
  msw_lds_barrier();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double s = red[k];
#pragma unroll
    for (int w = 1; w < W; ++w) s += red[w * K + k];
    v[k] = s;
  }
  msw_lds_barrier();
}
template <int W, int K>
__device__ __forceinline__ void msw_max_n(float (&v)[K], float* red, int wave, int lane) {
  static_assert(K <= 16, "scratch of the workgroup reductions");
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_max_nonneg(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[wave * K + k] = v[k];
  }
  msw_lds_barrier();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float m = red[k];
#pragma unroll
    for (int w = 1; w < W; ++w) m = fmaxf(m, red[w * K + k]);
    v[k] = m;
  }
  msw_lds_barrier();
}

// two maxima at once (one pair of barriers)
template <int W>
__device__ __forceinline__ void msw_max2(float& a, float& b, float* red, int wave, int lane) {
  a = wave_max_nonneg(a);
  b = wave_max_nonneg(b);
  if (lane == 0) { red[2 * wave] = a; red[2 * wave + 1] = b; }
  msw_lds_barrier();
  a = red[0]; b = red[1];
#pragma unroll
  for (int k = 1; k < W; ++k) { a = fmaxf(a, red[2 * k]); b = fmaxf(b, red[2 * k + 1]); }
  msw_lds_barrier();
}

// Scaled maximum norm of this wavefront's part of the residual of a sweep (see ms_residual_norm): the interface
// jumps E_g - Y_{g+1} of its intervals (19 rows each) and, on the last wavefront, the tip condition - from the end
// states of the unperturbed lanes in Es.  Not yet reduced over the workgroup.
template <typename T, int W>
__device__ __forceinline__ float msw_residual_local(const T* Es, const T* Xs, const T* cold, const MswRole& R, int lane) {
  constexpr int P = MswGeo<W>::P;
  float rn = 0.f;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = lane + 64 * q;
    const int k = e / 19, r = e - 19 * k;
    const int g = R.g0 + k;
    if (k < R.K && g < P - 1) {
      const T x = Xs[(g + 1) * 19 + r];
      rn = fmaxf(rn, update_ratio(Es[msw_l0(R.w, k) * 19 + r] - x, x));
    }
  }
  if (R.w == W - 1 && lane >= 58) {  // (lanes the loop above never uses on the last wavefront: 2 x 19 = 38 entries)
    const int k = lane - 58;
    const T e = Es[msw_l0(R.w, 2) * 19 + 7 + k];
    rn = fmaxf(rn, update_ratio(cold[CD_FTIP + k] - e, e));
  }
  return rn;
}

// ---- predictor of one wavefront's unknowns (NE = K x 19 entries of Xs starting at interval g0) -------------------
template <typename T, int NC>
__device__ __forceinline__ void mswp_init(MsPred<T, NC>& Q, int lane, int ne, int g0, int N, int P, const T* s0, const T* sp,
                                          bool has_prev, int predictor) {
#pragma unroll
  for (int q = 0; q < MS_EPL; ++q) {
    const int e = lane + q * WAVE;
    const int i = e < ne ? e / 19 : 0, r = e < ne ? e - i * 19 : 0;
    const size_t off = (size_t)msw_start(g0 + i, N, P) * KR_SLOTS + ms_slot_of_yrow(r);
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
__device__ __forceinline__ void mswp_guess(const MsPred<T, NC>& Q, int order, int lane, int ne, bool first, const T* cold, T* Xl) {
#pragma unroll
  for (int q = 0; q < MS_EPL; ++q) {
    const int e = lane + q * WAVE;
    if (e < ne) {
      const int i = e / 19, r = e - i * 19;
      T g;
      if (order == MS_ORDER_LP) {
        double b[NC];
        lp_basis<T, NC>(Q.Hx[q], b);
        g = (T)lp_eval<NC>(Q.lpa, b);
      } else {
        g = extrapolate_n<T>(order, Q.Hx[q]);
      }
      if (first && i == 0) {
        T bc;
        if (ms_base_bc(cold, r, bc)) g = bc;
      }
      Xl[e] = g;
    }
  }
}
// ms_pred_update with its decisions reduced over the W wavefronts of the rod
template <typename T, int W, int NC>
__device__ __forceinline__ void mswp_update(MsPred<T, NC>& Q, int order, int status, int predictor, int lane, int wave, int ne,
                                            const T* Xl, T* redT) {
  float* redf = reinterpret_cast<float*>(redT);
  double* redd = reinterpret_cast<double*>(redT);  // (W * 32 elements of T >= W doubles)
  const bool poly_eval = !(order == MS_ORDER_LP && Q.lp_good);
  float err[MS_HLEV];
#pragma unroll
  for (int p = 0; p < MS_HLEV; ++p) err[p] = 0.f;
  if (poly_eval) {
#pragma unroll
    for (int q = 0; q < MS_EPL; ++q) {
      const int e = lane + q * WAVE;
      if (e < ne) {
        const T x = Xl[e];
#pragma unroll
        for (int p = 0; p < MS_HLEV; ++p) err[p] = fmaxf(err[p], update_ratio(x - extrapolate_n<T>(p, Q.Hx[q]), x));
      }
    }
  }
  float err_lp = 0.f;
  if (Q.lp_have) {
#pragma unroll
    for (int q = 0; q < MS_EPL; ++q) {
      const int e = lane + q * WAVE;
      if (e < ne) {
        const double x = (double)Xl[e];
        double b[NC];
        lp_basis<T, NC>(Q.Hx[q], b);
        err_lp = fmaxf(err_lp, update_ratio(x - lp_eval<NC>(Q.lpa, b), x));
      }
    }
  }
  const float em_lp = msw_max<W>(err_lp, redf, wave, lane);
  const bool lp_tested = Q.lp_have;
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
      if (e < ne) {
        const double x = (double)Xl[e];
        double b[NC];
        lp_basis<T, NC>(Q.Hx[q], b);
        const double w = (double)__builtin_amdgcn_rcpf(fmaxf(fabsf((float)x), 1.0f));
        lp_accumulate<NC>(Sn, b, x, w);
      }
    }
    {  // (the reduction scratch holds 16 doubles per wavefront: in chunks)
      constexpr int NS = lp_nsum<NC>();
      constexpr int C0 = NS < 16 ? NS : 16, C1 = NS - C0 < 16 ? (NS - C0 > 0 ? NS - C0 : 1) : 16, C2 = NS - C0 - C1 > 0 ? NS - C0 - C1 : 1;
      static_assert(NS <= 48, "three chunks");
      double S0[C0], S1[C1], S2[C2];
#pragma unroll
      for (int k = 0; k < C0; ++k) S0[k] = Sn[k];
      msw_sum_n<W, C0>(S0, redd, wave, lane);
#pragma unroll
      for (int k = 0; k < C0; ++k) Sn[k] = S0[k];
      if constexpr (NS > 16) {
#pragma unroll
        for (int k = 0; k < C1; ++k) S1[k] = Sn[C0 + k];
        msw_sum_n<W, C1>(S1, redd, wave, lane);
#pragma unroll
        for (int k = 0; k < C1; ++k) Sn[C0 + k] = S1[k];
      }
      if constexpr (NS > 32) {
#pragma unroll
        for (int k = 0; k < C2; ++k) S2[k] = Sn[C0 + C1 + k];
        msw_sum_n<W, C2>(S2, redd, wave, lane);
#pragma unroll
        for (int k = 0; k < C2; ++k) Sn[C0 + C1 + k] = S2[k];
      }
    }
    double a[NC];
    if (lp_solve<NC>(Sn, a)) {
#pragma unroll
      for (int k = 0; k < NC; ++k) Q.lpa[k] = a[k];
      Q.lp_have = true;
    }
  }
  const int pmax = Q.avail < predictor ? Q.avail : (predictor < MS_HLEV ? predictor : MS_HLEV - 1);
  float em[MS_HLEV];
#pragma unroll
  for (int p = 0; p < MS_HLEV; ++p) em[p] = err[p];
  if (poly_eval) msw_max_n<W, MS_HLEV>(em, redf, wave, lane);  // (wave-uniform condition: every wavefront takes it)
  else {
#pragma unroll
    for (int p = 0; p < MS_HLEV; ++p) em[p] = 3.0e38f;
  }
  int nxt = 0;
  float eb = em[0];
#pragma unroll
  for (int p = 1; p < MS_HLEV; ++p)
    if (p <= pmax && p <= order + 2 && em[p] < eb) { eb = em[p]; nxt = p; }
  if (nxt == Q.avail && Q.avail + 1 < MS_HLEV && Q.avail + 1 <= predictor && nxt >= order) nxt = Q.avail + 1;
  if (!poly_eval) nxt = pmax;
  Q.next_order = nxt;
  Q.lp_good = lp_tested && Q.lp_have && em_lp < 1.0e-3f;
  if (lp_tested && Q.lp_have && (em_lp < eb || Q.lp_good)) Q.next_order = MS_ORDER_LP;
  if (status != KR_ST_CONVERGED) {
    Q.next_order = 0;
    Q.avail = -1;
  }
#pragma unroll
  for (int q = 0; q < MS_EPL; ++q) {
    const int e = lane + q * WAVE;
#pragma unroll
    for (int k = MS_HLEV - 1; k > 0; --k) Q.Hx[q][k] = Q.Hx[q][k - 1];
    if (e < ne) Q.Hx[q][0] = Xl[e];
  }
  if (Q.avail < MS_HLEV - 1) ++Q.avail;
}

// ---- one full Newton update of the rod, distributed over its W wavefronts ----------------------------------------
// From the end states of a sweep (y: this lane's; hstep: its forward-difference step) to the update of every unknown
// this wavefront owns and the norms that steer the iteration - everything between a sweep and the decision what to do
// next.  Every wavefront of the workgroup calls it (it contains workgroup barriers).  Leaves the forward-difference
// columns in Es, the updates in L.dY / U, and returns in U.dnf / U.res_local the workgroup-wide scaled maximum norms
// of the update and of the sweep's residual.
template <typename T>
struct MswUpd {
  float dnf, res_local;
  T updP, xsP, updG, xsG, updR[4], xsR[4];
  int pg, pprow;
  bool plane, glane;
};
// SECOND: a second solve through the same forward-difference columns (they are still in Es) with the base end states
// replaced by what the caller passes in y - the defect correction for the p columns (msw_newton, MLP on).
template <typename T, int W, bool SECOND = false>
__device__ __forceinline__ void msw_condense(const MswLds<T, W>& L, const MswRole& R, int lane, const RodState<T>& y,
                                             T hstep, MswUpd<T>& U
#ifdef KR_MS_STAMPS
                                             , MsStamps& stamps, unsigned long long& ta
#endif
) {
  constexpr int P = MswGeo<W>::P;
  const int wave = R.w;
  T* Xs = L.Xs;
  T* Es = L.es(wave);
  T* Lt = L.lt(wave);
  float* redf = reinterpret_cast<float*>(L.red);
  const int col = R.col;
  const bool idle = R.idle;
  const int kp = lane & 3;
  const int r = 3 + (lane >> 2);
  float res_local = 0.f;
    // ---- end states and forward-difference columns of this wavefront's intervals -------------------------
    {
      T er[19];
      state_to_rows(y, er);
      if (col == 0 && !idle) {
#pragma unroll
        for (int q = 0; q < 19; ++q) Es[lane * 19 + q] = er[q];
      }
      wave_sync();
      res_local = msw_residual_local<T, W>(Es, Xs, L.cold, R, lane);
      if (col > 0 && !SECOND) {
        const T ih = fast_rcp(hstep);
        T e0[19];
#pragma unroll
        for (int q = 0; q < 19; ++q) e0[q] = Es[R.l0own * 19 + q];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 19; ++q) Es[lane * 19 + q] = (er[q] - e0[q]) * ih;
      }
      wave_sync();
    }
    // c_k = E_k - Y_{g0+k+1} (0 for the very last interval), row r
    auto cterm = [&](int k) -> T {
      const int g = R.g0 + k;
      const T e0 = Es[msw_l0(wave, k) * 19 + r];
      return g < P - 1 ? e0 - Xs[(g + 1) * 19 + r] : T(0);
    };
    // ---- local condensation ----------------------------------------------------------------------------
    T Xreg[3][2];   // wavefront 0: column pair of X_1 .. X_3
    T Lreg[2][5];   // wavefront >= 1: columns kp, kp+4, kp+8, kp+12 (and 16 for kp == 0) of L_0, L_1
    if (wave == 0) {
      T* XB = Lt;  // [2][19][8] fits in the L tiles
      {
        const T c0 = cterm(0);
        const T da = Es[(2 * kp) * 19 + r];
        const T db = Es[(2 * kp + 1) * 19 + r];
        Xreg[0][0] = kp == 0 ? c0 : da;
        Xreg[0][1] = kp == 3 ? T(0) : db;
        store_pair(XB + r * 8 + 2 * kp, Xreg[0][0], Xreg[0][1]);
      }
      wave_sync();
#pragma unroll
      for (int g = 1; g < 4; ++g) {
        const T* xcur = XB + ((g - 1) & 1) * (19 * 8);
        T* xnext = XB + (g & 1) * (19 * 8);
        const int l0 = 7 + 17 * (g - 1);
        T n0 = kp == 0 ? cterm(g) : T(0), n1 = T(0), n2 = T(0), n3 = T(0);
        {
          T av[16], xa[16], xb[16];
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            av[c] = Es[(l0 + 1 + c) * 19 + r];
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
        if (g < 3) {
          Xreg[g][0] = n0;
          Xreg[g][1] = n1;
          store_pair(xnext + r * 8 + 2 * kp, n0, n1);
        } else {
          store_pair(L.Xbd + (size_t)1 * 19 * 8 + r * 8 + 2 * kp, n0, n1);  // X^(1): dY at wavefront 1's first interval
        }
        wave_sync();
      }
    } else {
      // L_0 = [c | A] of the first interval; L_{k+1} = [c | 0] + A_{k+1} L_k.  Column cc of a tile row sits at
      // msw_lpos(cc): the four columns kp, kp + 4, kp + 8, kp + 12 of a lane are one 4-vector
      {
        T v4[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int cc = kp + 4 * m;  // column of L: 0 = a, 1..16 = B
          v4[m] = cc == 0 ? cterm(0) : Es[(msw_l0(wave, 0) + cc) * 19 + r];
          Lreg[0][m] = v4[m];
        }
        Lreg[0][4] = Es[(msw_l0(wave, 0) + 16) * 19 + r];
        store_vec<T, 4>(Lt + r * MSW_LT_LD + 4 * kp, v4);
        if (kp == 0) Lt[r * MSW_LT_LD + 16] = Lreg[0][4];
      }
      wave_sync();
#pragma unroll
      for (int k = 1; k < 3; ++k) {
        const T* lcur = Lt + ((k - 1) & 1) * (19 * MSW_LT_LD);
        T* lnext = Lt + (k & 1) * (19 * MSW_LT_LD);
        const int l0 = msw_l0(wave, k);
        T acc[5];
#pragma unroll
        for (int m = 0; m < 5; ++m) acc[m] = T(0);
        if (kp == 0) acc[0] = cterm(k);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          // all LDS reads of eight columns back to back, then the arithmetic
          T av[8], lv[8][4], l16[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const int cq = 8 * half + c;
            av[c] = Es[(l0 + 1 + cq) * 19 + r];
            load_hist_vec<T, 4>(lcur + (3 + cq) * MSW_LT_LD + 4 * kp, lv[c]);
            l16[c] = lcur[(3 + cq) * MSW_LT_LD + 16];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < 8; ++c) {
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = fma(av[c], lv[c][m], acc[m]);
            acc[4] = fma(av[c], l16[c], acc[4]);
          }
        }
        if (k < 2) {
#pragma unroll
          for (int m = 0; m < 5; ++m) Lreg[1][m] = acc[m];
        }
        {
          T v4[4] = {acc[0], acc[1], acc[2], acc[3]};
          store_vec<T, 4>(lnext + r * MSW_LT_LD + 4 * kp, v4);
          if (kp == 0) lnext[r * MSW_LT_LD + 16] = acc[4];
        }
        wave_sync();
      }
    }
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(stamps.a1, ta);
#endif
    msw_lds_barrier();
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(stamps.a2, ta);
#endif
    // ---- boundary blocks down the rod: X^(w+1) = [a^w | 0] + B^w X^(w);  the last wavefront forms T dG = rhs ------
#pragma unroll
    for (int w = 1; w < W; ++w) {
      if (wave == w) {
        const T* l2 = Lt;  // after two stages the final map sits in tile 0
        const T* xin = L.Xbd + (size_t)w * 19 * 8;
        T n0 = kp == 0 ? l2[r * MSW_LT_LD + 0] : T(0), n1 = T(0), n2 = T(0), n3 = T(0);
        {
          T bv[16], xa[16], xb[16];
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            bv[c] = l2[r * MSW_LT_LD + msw_lpos(1 + c)];
            load_pair(xin + (3 + c) * 8 + 2 * kp, xa[c], xb[c]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < 16; c += 2) {
            n0 = fma(bv[c], xa[c], n0);
            n1 = fma(bv[c], xb[c], n1);
            n2 = fma(bv[c + 1], xa[c + 1], n2);
            n3 = fma(bv[c + 1], xb[c + 1], n3);
          }
          n0 += n2;
          n1 += n3;
        }
        if (w < W - 1) {
          store_pair(L.Xbd + (size_t)(w + 1) * 19 * 8 + r * 8 + 2 * kp, n0, n1);
        } else if (r >= 7 && r < 13) {
          // tip rows: [n; m](E_last + dE_last) = [F_tip; M_tip]
          const T e0 = Es[msw_l0(wave, 2) * 19 + r];
          if (kp == 0) n0 = L.cold[CD_FTIP + (r - 7)] - e0 - n0;
          store_pair(L.Tm + (r - 7) * 8 + 2 * kp, n0, n1);
        }
      }
      msw_lds_barrier();
    }
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(stamps.a3, ta);
#endif
    // ---- 6 x 6 solve, redundantly in every lane -----------------------------------------------------------
    T d[6];
    {
      T a6[6][7];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        T row[8];
        load_hist_vec<T, 8>(L.Tm + i * 8, row);
#pragma unroll
        for (int k = 0; k < 6; ++k) a6[i][k] = row[1 + k];
        a6[i][6] = row[0];
      }
      solve6(a6, d);
    }
#ifdef KR_MS_STAMPS
    unsigned long long tb = ta;
    KR_STAMP_ADD(stamps.prep, tb);   // 6 x 6 solve
#endif
    // ---- updates of rows 3..18 of every unknown ---------------------------------------------------------------
    const T dd0 = kp == 0 ? T(1) : kp == 1 ? d[1] : kp == 2 ? d[3] : d[5];
    const T dd1 = kp == 0 ? d[0] : kp == 1 ? d[2] : kp == 2 ? d[4] : T(0);
    auto quad_sum = [](T s) {
      s += quad_xor<0xB1>(s);
      s += quad_xor<0x4E>(s);
      return s;
    };
    if (wave == 0) {
#pragma unroll
      for (int g = 1; g < 4; ++g) {
        const T s = quad_sum(fma(Xreg[g - 1][1], dd1, Xreg[g - 1][0] * dd0));
        if (kp == 0) L.dY[g * 19 + r] = s;
      }
      if (lane < 6) L.dY[0 * 19 + 7 + lane] = lane == 0 ? d[0] : lane == 1 ? d[1] : lane == 2 ? d[2] : lane == 3 ? d[3] : lane == 4 ? d[4] : d[5];
    } else {
      T xa, xb;
      load_pair(L.Xbd + (size_t)wave * 19 * 8 + r * 8 + 2 * kp, xa, xb);
      const T s = quad_sum(fma(xb, dd1, xa * dd0));  // dY at this wavefront's first interval, row r
      if (kp == 0) L.dY[R.g0 * 19 + r] = s;
      wave_sync();
      // inner intervals: dY_{g0+k+1} = L_k [1; dY_{g0}]
      T yin[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int cc = kp + 4 * m;
        yin[m] = cc == 0 ? T(1) : L.dY[R.g0 * 19 + 2 + cc];  // column cc multiplies dY_in[row 3 + cc - 1]
      }
      const T y16 = L.dY[R.g0 * 19 + 18];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        T part = T(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) part = fma(Lreg[k][m], yin[m], part);
        if (kp == 0) part = fma(Lreg[k][4], y16, part);
        part = quad_sum(part);
        if (kp == 0) L.dY[(R.g0 + k + 1) * 19 + r] = part;
      }
    }
    msw_lds_barrier();
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(stamps.total, tb);  // back-substitution + barrier
#endif
    // ---- p rows: dY_{g+1}[p] = sum_{i<=g} (c_i[p] + A_i[p, :] dY_i[3:]) ------------------------------------------
    if (lane < 3 * R.K) {
      const int k = lane / 3, prow = lane - 3 * k;
      const int g = R.g0 + k;
      const int l0 = msw_l0(wave, k);
      T s = g < P - 1 ? Es[l0 * 19 + prow] - Xs[(g + 1) * 19 + prow] : T(0);
      T s2 = T(0);
      // uniform trip count (A_0 has 6 columns: the rest is masked) so that the reads can be issued together
      T av[16], dv[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const bool use = g > 0 || c < 6;
        av[c] = use ? Es[(l0 + 1 + c) * 19 + prow] : T(0);
        dv[c] = use ? L.dY[g * 19 + (g > 0 ? 3 : 7) + c] : T(0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 16; c += 2) {
        s = fma(av[c], dv[c], s);
        s2 = fma(av[c + 1], dv[c + 1], s2);
      }
      L.sp[g * 4 + prow] = s + s2;
    }
    msw_lds_barrier();
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(stamps.osum, tb);   // p rows + barrier
#endif
    // ---- scaled update norm over this wavefront's unknowns, then over the rod -----------------------------------
    float dnf = 0.f;
    T updP = T(0), xsP = T(0);
    const bool plane = lane < 3 * R.K && R.g0 + lane / 3 + 1 < P;  // owns Y_{g0+k+1}[prow]
    int pg = 0, pprow = 0;
    if (plane) {
      pg = R.g0 + lane / 3 + 1;
      pprow = lane % 3;
      xsP = Xs[pg * 19 + pprow];
#pragma unroll
      for (int i = 0; i < P - 1; ++i) {
        const T t = L.sp[i * 4 + pprow];
        updP += i < pg ? t : T(0);
      }
      dnf = update_ratio(updP, xsP);
    }
    // rows 3..18: lanes 0..15 (+16 k) own (row 3 + (lane & 15)) of the local unknown k
    T updR[4], xsR[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      updR[k] = T(0); xsR[k] = T(0);
      const int g = wave == 0 ? 1 + k : R.g0 + k;       // unknown states owned by this wavefront
      const bool own = k < 3 && g < P && (lane >> 4) == k;
      if (own) {
        const int rr = 3 + (lane & 15);
        xsR[k] = Xs[g * 19 + rr];
        updR[k] = L.dY[g * 19 + rr];
        dnf = fmaxf(dnf, update_ratio(updR[k], xsR[k]));
      }
    }
    T updG = T(0), xsG = T(0);
    const bool glane = wave == 0 && lane >= WAVE - 6;
    if (glane) {
      const int k = lane - (WAVE - 6);
      xsG = Xs[0 * 19 + 7 + k];
      updG = L.dY[0 * 19 + 7 + k];
      dnf = fmaxf(dnf, update_ratio(updG, xsG));
    }
    msw_max2<W>(dnf, res_local, redf, wave, lane);
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(stamps.retries, tb);  // norms + reduction over the rod
#endif
  U.dnf = dnf; U.res_local = res_local;
  U.updP = updP; U.xsP = xsP; U.updG = updG; U.xsG = xsG;
#pragma unroll
  for (int k = 0; k < 4; ++k) { U.updR[k] = updR[k]; U.xsR[k] = xsR[k]; }
  U.pg = pg; U.pprow = pprow; U.plane = plane; U.glane = glane;
}
// adds the update msw_condense left in U to the unknowns this wavefront owns
template <typename T, int W>
__device__ __forceinline__ void msw_apply(const MswLds<T, W>& L, const MswRole& R, int lane, const MswUpd<T>& U) {
  constexpr int P = MswGeo<W>::P;
  T* Xs = L.Xs;
  if (U.plane) Xs[U.pg * 19 + U.pprow] = U.xsP + U.updP;
  if (U.glane) Xs[0 * 19 + 7 + (lane - (WAVE - 6))] = U.xsG + U.updG;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int g = R.w == 0 ? 1 + k : R.g0 + k;
    if (k < 3 && g < P && (lane >> 4) == k) Xs[g * 19 + 3 + (lane & 15)] = U.xsR[k] + U.updR[k];
  }
}

// ---- Newton iteration of one rod on W wavefronts ----------------------------------------------------------------
// hist: the history records [N][HS_LEAN] (LDS, or global memory in the persistent kernel with the MLP on)
template <typename T, bool DIAG, int W, bool NN = false, int EV = 0>
__device__ __forceinline__ int msw_newton(const RodConst<T>& Pc, const MswLds<T, W>& L, const MswRole& R, int lane,
                                          V3<T> fconst, MsSolveArgs<T>& S, int& it, MsStamps& stamps,
                                          const T* hist, const MswNn<T>& nn) {
  constexpr int P = MswGeo<W>::P;
  const int N = Pc.N;
  const int wave = R.w;
#ifdef KR_MS_STAMPS
  unsigned long long tq;
  KR_STAMP(tq);
#endif
  T* Xs = L.Xs;
  T* Es = L.es(wave);
  T* Lt = L.lt(wave);
  float* redf = reinterpret_cast<float*>(L.red);
  const int iv = R.iv, col = R.col;
  const bool idle = R.idle;
  const int kp = lane & 3;
  const int r = 3 + (lane >> 2);
  const bool last_wave = wave == W - 1;
  bool storing = false, flush = false;
  int status = KR_ST_MAXIT;
  it = 0;
  T dn_prev = T(-1);
  const T kappa_in = S.kappa;
  bool below = false;
  float amp = -1.f;  // |update| / |residual| of the last full iteration of this solve (residual test, see ms_newton)
  // p columns of the Jacobian (kr_ms_impl.hpp, ms_newton: the network reads the start position of an interval, the physics
  // does not).  The first wavefront has six spare lanes for its three interior intervals (two per sweep, in rotation), the
  // others thirteen for their three (all of them, every sweep).
  // (fp64 only, as there: measured with fp32 the extra lanes and the second solve cost 8 - 13 % and save no sweep)
  constexpr bool PCOL = NN && sizeof(T) == 8;
  const int nact = wave == 0 ? 58 : 51;              // lanes with a forward-difference role
  const int pslot = lane - nact;                     // >= 0 on the spare lanes
  const bool pl_any = PCOL && idle && pslot < (wave == 0 ? 6 : 9);
  T* Bw = PCOL ? L.Bp + (size_t)wave * MSW_BP_W : nullptr;   // [3][3][19]: local interval (0..2 after the first), direction, row
  T* dPl = PCOL ? L.Bp + (size_t)W * MSW_BP_W : nullptr;     // [P][3]: p update of every interval start

  while (true) {
    // ---- role of the spare lanes in this sweep --------------------------------------------------------
    int iv_l = iv, s_l = R.s_i, len_l = R.len_i, pk = 0, pc = 0;   // pk: index of the p lane's interval in Bw
    MswNn<T> nl = nn;
    if constexpr (PCOL) {
      nl.role.ptab = wave == 0 ? 1 : 2;
      int gu = 0;
      if (wave == 0) {
        const int rot = S.prot % 3;  // local intervals served: (1, 2), (3, 1), (2, 3)
        const int ga = rot == 0 ? 1 : rot == 1 ? 3 : 2, gb = rot == 0 ? 2 : rot == 1 ? 1 : 3;
        gu = 6 - ga - gb;
        if (pl_any) { const int gl = pslot < 3 ? ga : gb; pk = gl - 1; pc = pslot < 3 ? pslot : pslot - 3; iv_l = gl; }
      } else if (pl_any) {
        pk = pslot / 3; pc = pslot - 3 * pk; iv_l = R.g0 + pk;
      }
      S.prot += 1;
      if (pl_any) {
        s_l = msw_start(iv_l, N, P);
        len_l = R.sbase + (iv_l < (N - 1) % P ? 1 : 0);
        nl.role.iv = wave == 0 ? pk + 1 : pk; nl.role.col = 1; nl.role.idle = false;
        nl.role.xrow = wave == 0 ? 6 + 3 * pk + pc : 48 + 3 * pk + pc;
      } else if (wave == 0) {
        if (col == 0 && !idle) nl.role.zrow = R.ivl < 3 ? 6 + 3 * (gu - 1) + R.ivl : 15;
      } else {
        // rows 57..63 of sample tile 3: the three unperturbed lanes and the four lanes without any role
        if (col == 0 && !idle) nl.role.zrow = 57 + R.ivl;
        else if (idle) nl.role.zrow = 60 + (pslot - 9);
      }
    }
    // ---- start state of this lane, forward-difference step of its column -------------------------------
    T yr[19];
#pragma unroll
    for (int q = 0; q < 19; ++q) yr[q] = Xs[iv_l * 19 + q];
    const T hstep = col > 0 ? S.fd_eps * fmax(fabs(Xs[iv * 19 + (R.comp > 0 ? R.comp : 3)]), T(1)) : T(1);
#pragma unroll
    for (int q = 3; q < 19; ++q) yr[q] += q == R.comp ? hstep : T(0);
    T hp = T(1);
    if constexpr (PCOL) {
      if (pl_any) {
        hp = S.fd_eps * fmax(fabs(Xs[iv_l * 19 + pc]), T(1));
#pragma unroll
        for (int q = 0; q < 3; ++q) yr[q] += q == pc ? hp : T(0);
      }
    }
    RodState<T> y = rows_to_state(yr);
    const bool st = (storing || flush) && col == 0 && !idle;

    // ---- sweep over this lane's sub-interval (explicit Euler, cosserat_ode.py:198-201) ------------------
    T hv[HS_LEAN];
    load_hist_vec<T, HS_LEAN>(hist + (size_t)s_l * HS_LEAN, hv);
    auto point = [&](auto store_tag, int j, bool live) __attribute__((always_inline)) {
      constexpr bool STORE = decltype(store_tag)::value;
      RodState<T> k1;
      V3<T> v, u;
      ode_eval<T, DIAG>(Pc, y, hist_lean<T, DIAG>(Pc, hv), fconst, k1, v, u);
      if constexpr (NN)  // every wavefront evaluates the network for its own lanes (cosserat_ode.py:169-184)
        nn_correct<T, HS_LEAN, EV, true>(*nl.M, nullptr, nullptr, 0, nl.tile, lane, nl.role, y, hv, nl.tf, k1, v, u);
      if constexpr (STORE) {
        if (st && live) {
          T rec[KR_SLOTS];
          record_from(y, v, u, rec);
          msw_record_pad<T, NN>(rec);
          T lead[12];
#pragma unroll
          for (int c = 0; c < 12; ++c) lead[c] = rec[c];
          if (S.lean) store_vec<T, 12>(S.out_rod + (size_t)j * KR_SLOTS, lead);  // (wave-uniform choice)
          else store_record(S.out_rod + (size_t)j * KR_SLOTS, rec);
          if (S.lead12) store_vec<T, 12>(S.lead12 + (size_t)j * 12, lead);
        }
      }
      load_hist_vec<T, HS_LEAN>(hist + (size_t)(j + 1) * HS_LEAN, hv);
      y = state_axpy(y, Pc.ds, k1);
    };
    if constexpr (NN) {
      // with the MLP on every lane must reach the wavefront-wide matrix-core call: lanes past the end of their
      // (shorter) interval keep running on the last grid point and do not commit (kr_ms_impl.hpp, ms_newton)
      const int lmax = R.sbase + (R.g0 < (N - 1) % P ? 1 : 0);  // (the long intervals come first)
      for (int t = 0; t < lmax; ++t) {
        const bool live = t < len_l;
        const int j = live ? s_l + t : s_l + len_l - 1;
        const RodState<T> y_in = y;
        if (storing || flush) point(std::true_type{}, j, live);  // (wave-uniform choice)
        else point(std::false_type{}, j, true);
        if (!live) y = y_in;
      }
    } else if (storing || flush) {
      for (int t = 0; t < R.sbase; ++t) point(std::true_type{}, R.s_i + t, true);
      if (R.len_i > R.sbase) point(std::true_type{}, R.s_i + R.sbase, true);
    } else {
#pragma unroll kMsUnroll
      for (int t = 0; t < R.sbase; ++t) point(std::false_type{}, R.s_i + t, true);
      if (R.len_i > R.sbase) point(std::false_type{}, R.s_i + R.sbase, true);
    }
    if (st && iv == P - 1) {
      T rec[KR_SLOTS];
      record_from(y, S.vlast, S.ulast, rec);
      msw_record_pad<T, NN>(rec);
      store_record(S.out_rod + (size_t)(N - 1) * KR_SLOTS, rec);
      if (S.lead12) {
        T lead[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) lead[c] = rec[c];
        store_vec<T, 12>(S.lead12 + (size_t)(N - 1) * 12, lead);
      }
      if (S.tip) { S.tip[0] = y.p.x; S.tip[1] = y.p.y; S.tip[2] = y.p.z; }
    }
    if (flush) break;
    ++it;
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(stamps.sweep, tq);
    unsigned long long ta = tq;
#endif

    // ---- residual test of a storing sweep that follows a small update (kr_ms_impl.hpp, ms_newton) ------------
    float res_local = 0.f;
    if (!NN && S.quick_ok && storing && amp > 0.f && dn_prev > T(0) && dn_prev <= T(1e-2)) {  // (MLP on: approximate Jacobian, ratio not audited)
      {
        T er[19];
        state_to_rows(y, er);
        if (col == 0 && !idle) {
#pragma unroll
          for (int q = 0; q < 19; ++q) Es[lane * 19 + q] = er[q];
        }
      }
      wave_sync();
      const float rn = msw_max<W>(msw_residual_local<T, W>(Es, Xs, L.cold, R, lane), redf, wave, lane);
      const float est = amp * rn;
#ifdef KR_QUICK_AUDIT
      stamps.qn = (double)est;  // compared with the update below
#else
      if (T(256) * (T)est <= S.tol) {  // (NaN compares false)
        status = KR_ST_CONVERGED;
        if (!below && dn_prev > T(0)) {
          const T floor_dn = T(64) * (sizeof(T) == 8 ? T(2.2e-16) : T(1.2e-7));
          S.kappa = fmin(fmax(fmax((T)est, floor_dn) * fast_rcp(dn_prev * dn_prev), T(1e-4)), T(1));
        }
        break;
      }
#endif
    }
    MswUpd<T> U;
#ifdef KR_MS_STAMPS
    msw_condense<T, W>(L, R, lane, y, hstep, U, stamps, ta);
#else
    msw_condense<T, W>(L, R, lane, y, hstep, U);
#endif
    if constexpr (PCOL) {
      // ---- p columns: blocks from the spare lanes, then one step of defect correction (kr_ms_impl.hpp) ----------
      {
        T er[19];
        state_to_rows(y, er);
        if (pl_any) {  // column pc of B = dE/dp - [I; 0] of local interval (wave 0: pk + 1, others: pk)
          const T ihp = fast_rcp(hp);
          const int l0p = msw_l0(wave, wave == 0 ? pk + 1 : pk);
          T e0[19];
#pragma unroll
          for (int q = 0; q < 19; ++q) e0[q] = Es[l0p * 19 + q];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 19; ++q) Bw[(pk * 3 + pc) * 19 + q] = (er[q] - e0[q]) * ihp - (q == pc ? T(1) : T(0));
        }
      }
      if (U.plane) dPl[U.pg * 3 + U.pprow] = U.updP;
      msw_lds_barrier();
      if (U.dnf <= 3.0e38f) {   // (uniform over the workgroup)
        // "end states" of the second solve: Y_{g+1} + B_g dY_g[p] (tip rows of the last interval: F_tip + ...)
        RodState<T> ysub = y;
        if (col == 0 && !idle) {
          const int g = iv;
          const int kb = wave == 0 ? R.ivl - 1 : R.ivl;   // index of this interval's block (interval 0 has none)
          T rows[19];
#pragma unroll
          for (int q = 0; q < 19; ++q) {
            T cp = T(0);
            if (g > 0) {
#pragma unroll
              for (int c = 0; c < 3; ++c) cp = fma(Bw[(kb * 3 + c) * 19 + q], dPl[g * 3 + c], cp);
            }
            T basev;
            if (g < P - 1) basev = Xs[(g + 1) * 19 + q];
            else basev = (q >= 7 && q < 13) ? L.cold[CD_FTIP + (q - 7)] : T(0);
            rows[q] = basev + cp;
          }
          ysub = rows_to_state(rows);
        }
        MswUpd<T> U2;
#ifdef KR_MS_STAMPS
        msw_condense<T, W, true>(L, R, lane, ysub, hstep, U2, stamps, ta);
#else
        msw_condense<T, W, true>(L, R, lane, ysub, hstep, U2);
#endif
        U.updP += U2.updP; U.updG += U2.updG;
#pragma unroll
        for (int k = 0; k < 4; ++k) U.updR[k] += U2.updR[k];
        float nf = 0.f;
        if (U.plane) nf = update_ratio(U.updP, U.xsP);
        if (U.glane) nf = fmaxf(nf, update_ratio(U.updG, U.xsG));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int g = wave == 0 ? 1 + k : R.g0 + k;
          if (k < 3 && g < P && (lane >> 4) == k) nf = fmaxf(nf, update_ratio(U.updR[k], U.xsR[k]));
        }
        U.dnf = msw_max<W>(nf, redf, wave, lane);
      }
    }
    float dnf = U.dnf;
    res_local = U.res_local;
    const bool finite = dnf <= 3.0e38f;
    const T dn = (T)dnf;
    if (finite && res_local > 0.f) amp = dnf / res_local;
#if defined(KR_MS_STAMPS) && defined(KR_QUICK_AUDIT)
    if (stamps.qn > 0.0 && finite && dnf > 0.f) { stamps.qa = fmax(stamps.qa, (double)dnf / stamps.qn); stamps.qn = 0.0; }
#endif
    if (finite && !below && dn <= S.tol) {
      below = true;
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
      flush = !storing;
    } else if (storing && dn <= S.tol) {
      done = true;
      status = KR_ST_CONVERGED;
    } else {
      msw_apply<T, W>(L, R, lane, U);
      if (predict_final<T>(dn, dn_prev, S.tol, S.tolA)) storing = true;
      if (kappa_in > T(0) && T(4) * kappa_in * dn * dn <= S.tol) storing = true;
      dn_prev = dn;
      if (it >= S.maxit) {
        done = true;
        status = KR_ST_MAXIT;
        flush = true;
      }
    }
    msw_lds_barrier();
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(stamps.a4, ta);
    KR_STAMP_ADD(stamps.alg, tq);
    if (it <= 4) stamps.dn[it - 1] = (double)dn;
#endif
    if (done && !flush) break;
    if (done && flush) storing = false;
  }
  return status;
}

// ---- one time step per launch: one workgroup of W wavefronts per rod -----------------------------------------
// ---------------------------------------------------------------------------
// Last resort of a time step, as in kr_ms_impl.hpp (ss_newton_damped): damped single shooting on wavefront 0 - Newton
// on the 6 base unknowns with a forward-difference Jacobian from lanes 1..6, every update taken as G - lam d with lam
// halved until the residual norm has decreased, iteration cap 8 x maxit (the oracle's newton_shoot(damped=True)).
// Lane 0 streams every sweep; the sweep whose full Newton update is below the tolerance is the accepted one.  Leaves
// L.Xs (all P interval starts) consistent with the result, so the predictor of every wavefront sees the accepted
// unknowns.  Called by wavefront 0 only; the other wavefronts wait at the workgroup barrier that follows.  With it the
// status of a hard step no longer depends on which kernel the batch size selects.
// ---------------------------------------------------------------------------
template <typename T, bool DIAG, int W, bool NN = false, int EV = 0>
__device__ __forceinline__ int msw_ss_damped(const RodConst<T>& Pc, const MswLds<T, W>& L, int lane, V3<T> fconst,
                                             MsSolveArgs<T>& S, int& it, const T* hist, const MswNn<T>& nn) {
  constexpr int P = MswGeo<W>::P;
  const int N = Pc.N;
  const int col = lane < 7 ? lane : 0;  // lanes 7.. duplicate the unperturbed column and store nothing
  T G[6], Gold[6], d[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { G[k] = L.Xs[0 * 19 + 7 + k]; Gold[k] = G[k]; d[k] = T(0); }
  T nr_old = T(-1), lam = T(1);
  bool have_trial = false;
  int status = KR_ST_MAXIT;
  const int maxit = 8 * S.maxit;
  it = 0;
  // [7][6] residuals of the seven columns: Es of wavefront 0 is free between sweeps - with the MLP on it is the
  // evaluator's scratch (lanes 0..6 are the unperturbed lane and six columns of "interval 0" there too, the other
  // lanes copies of the unperturbed one: the roles of the multiple-shooting sweep fit as they are), so dY serves
  T* Rx = NN ? L.dY : L.Es;
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
    int gnext = 1, jnext = msw_start(1, N, P);  // interval starts passed on the way (for L.Xs)
    for (int j = 0; j < N - 1; ++j) {
      T hv[HS_LEAN];
      load_hist_vec<T, HS_LEAN>(hist + (size_t)j * HS_LEAN, hv);
      RodState<T> k1;
      V3<T> v, u;
      ode_eval<T, DIAG>(Pc, y, hist_lean<T, DIAG>(Pc, hv), fconst, k1, v, u);
      if constexpr (NN) nn_correct<T, HS_LEAN, EV, true>(*nn.M, nullptr, nullptr, 0, nn.tile, lane, nn.role, y, hv, nn.tf, k1, v, u);
      if (lane == 0) {
        T rec[KR_SLOTS];
        record_from(y, v, u, rec);
        msw_record_pad<T, NN>(rec);
        if (S.out_rod) store_record(S.out_rod + (size_t)j * KR_SLOTS, rec);
        if (S.lead12) {
          T lead[12];
#pragma unroll
          for (int c = 0; c < 12; ++c) lead[c] = rec[c];
          store_vec<T, 12>(S.lead12 + (size_t)j * 12, lead);
        }
        if (j == jnext && gnext < P) {
          T yr[19];
          state_to_rows(y, yr);
#pragma unroll
          for (int q = 0; q < 19; ++q) L.Xs[gnext * 19 + q] = yr[q];
        }
      }
      if (j == jnext && gnext < P) { ++gnext; jnext = msw_start(gnext, N, P); }
      y = state_axpy(y, Pc.ds, k1);
    }
    if (lane == 0) {
      T rec[KR_SLOTS];
      record_from(y, S.vlast, S.ulast, rec);
      msw_record_pad<T, NN>(rec);
      if (S.out_rod) store_record(S.out_rod + (size_t)(N - 1) * KR_SLOTS, rec);
      if (S.lead12) {
        T lead[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) lead[c] = rec[c];
        store_vec<T, 12>(S.lead12 + (size_t)(N - 1) * 12, lead);
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
    for (int k = 0; k < 6; ++k) L.Xs[0 * 19 + 7 + k] = G[k];
  }
  wave_sync();
  return status;
}

template <typename T, bool DIAG, int W>
__global__ __launch_bounds__(WAVE * W) void msw_step_kernel(const RodConst<T> Pc, const StepArgs<T> A) {
  constexpr int P = MswGeo<W>::P;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = Pc.N;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x / WAVE;
  const int64_t rod = blockIdx.x;
  const size_t rod_elems = (size_t)N * KR_SLOTS;
  const MswLds<T, W> L = msw_carve<T, W>(reinterpret_cast<T*>(smem_raw), N);
  const MswRole R = msw_role<W>(wave, lane, N);
  MsStamps stamps;
#ifdef KR_MS_STAMPS
  unsigned long long t_begin, tp;
  KR_STAMP(t_begin);
  tp = t_begin;
#endif
  if (wave == 0) ms_cold_fill<T>(Pc, L.cold, lane);
  // BDF2 history (knode.py:74-75), raw terms only
  for (int j = threadIdx.x; j < N; j += WAVE * W) {
    const size_t off = rod * rod_elems + (size_t)j * KR_SLOTS;
    T cv[12], pv[12], hv[12];
    load_hist_vec<T, 12>(A.cur + off, cv);
    load_hist_vec<T, 12>(A.prev + off, pv);
#pragma unroll
    for (int k = 0; k < 12; ++k) hv[k] = A.hc1 * cv[k] + A.hc2 * pv[k];
    store_vec<T, 12>(L.hist + (size_t)j * HS_LEAN, hv);
  }
  __syncthreads();
  V3<T> fconst;
  {
    V3<T> tf{T(0), T(0), T(0)};
    const T* tens4 = A.tens + rod * A.tens_stride;
#pragma unroll
    for (int t = 0; t < 4; ++t) {  // cosserat_ode.py:195
      const T tt = tens4[t];
      tf.x += tt * L.cold[CD_TDIRS + t * 3 + 0];
      tf.y += tt * L.cold[CD_TDIRS + t * 3 + 1];
      tf.z += tt * L.cold[CD_TDIRS + t * 3 + 2];
    }
    fconst = {L.cold[CD_RHOAG] + tf.x, L.cold[CD_RHOAG + 1] + tf.y, L.cold[CD_RHOAG + 2] + tf.z};
  }
  MsSolveArgs<T> S;
  {
    const T* cl = A.cur + rod * rod_elems + (size_t)(N - 1) * KR_SLOTS;
    S.vlast = {cl[SL_V], cl[SL_V + 1], cl[SL_V + 2]};
    S.ulast = {cl[SL_U], cl[SL_U + 1], cl[SL_U + 2]};
  }
  S.out_rod = A.next + rod * rod_elems;
  S.tip = A.tip ? A.tip + rod * A.tip_stride : nullptr;
  S.tol = A.tol; S.tolA = A.tolA; S.fd_eps = A.fd_eps; S.maxit = A.maxit; S.quick_ok = A.residual_test != 0;
  S.kappa = T(0);
  const int ne = R.K * 19;
  T* Xl = L.Xs + R.g0 * 19;
  int it, status;
  const MswNn<T> nn;  // (MLP off in this kernel)
#ifdef KR_MS_STAMPS
  KR_STAMP_ADD(stamps.prep, tp);
#endif
  if (A.pred) {
    double* img = A.pred + ((size_t)rod * W + wave) * MS_PRED_ROWS * WAVE;
    MsPred<T> Q;
    if (A.pred_reset) mswp_init<T>(Q, lane, ne, R.g0, N, P, A.cur + rod * rod_elems, A.prev + rod * rod_elems, A.pred_has_prev != 0, A.pred_limit);
    else ms_pred_load<T>(Q, img, lane);
    S.kappa = Q.kappa;
    int order = Q.next_order;
    while (true) {
      mswp_guess<T>(Q, order, lane, ne, wave == 0, L.cold, Xl);
      wave_sync();
      if (wave == 0 && order <= 0 && lane < 6) L.Xs[0 * 19 + 7 + lane] = A.G[rod * 6 + lane];  // caller's guess (knode.py:67,89)
      __syncthreads();
      status = msw_newton<T, DIAG, W>(Pc, L, R, lane, fconst, S, it, stamps, L.hist, nn);
      if (status == KR_ST_CONVERGED || order == 0) break;
      order = 0;  // the predicted start did not converge: redo the step from the reference's warm start
      __syncthreads();
    }
    if (status != KR_ST_CONVERGED) {  // (uniform over the workgroup) plain Newton failed from the warm start too
      __syncthreads();
      if (wave == 0) {
        if (lane < 6) L.Xs[0 * 19 + 7 + lane] = A.G[rod * 6 + lane];
        wave_sync();
        const int it_plain = it;  // `iters` reports the plain and the damped phase together (knode_rod.h)
        status = msw_ss_damped<T, DIAG, W>(Pc, L, lane, fconst, S, it, L.hist, nn);
        it += it_plain;
        if (lane == 0) { L.red[0] = (T)status; L.red[1] = (T)it; }
      }
      __syncthreads();
      status = (int)L.red[0];
      it = (int)L.red[1];
      __syncthreads();
    }
#ifdef KR_MS_STAMPS
    KR_STAMP(tp);
#endif
    mswp_update<T, W>(Q, order, status, A.pred_limit, lane, wave, ne, Xl, L.red);
    Q.kappa = S.kappa;
    ms_pred_save<T>(Q, img, lane);
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(stamps.osum, tp);
#endif
  } else {
    for (int e = lane; e < ne; e += WAVE) {
      const int i = e / 19, rr = e - i * 19;
      const int sj = msw_start(R.g0 + i, N, P);
      const size_t off = rod * rod_elems + (size_t)sj * KR_SLOTS + ms_slot_of_yrow(rr);
      T g = extrapolate<T>(A.pred_order, A.cur[off], A.prev[off], A.prev2 ? A.prev2[off] : T(0));
      if (wave == 0 && i == 0) {
        T bc;
        if (ms_base_bc(L.cold, rr, bc)) g = bc;
        else if (A.pred_order <= 0) g = A.G[rod * 6 + (rr - 7)];
      }
      Xl[e] = g;
    }
    __syncthreads();
    status = msw_newton<T, DIAG, W>(Pc, L, R, lane, fconst, S, it, stamps, L.hist, nn);
    if (status != KR_ST_CONVERGED) {  // damped single shooting from the caller's guess
      __syncthreads();
      if (wave == 0) {
        if (lane < 6) L.Xs[0 * 19 + 7 + lane] = A.G[rod * 6 + lane];
        wave_sync();
        const int it_plain = it;  // `iters` reports the plain and the damped phase together (knode_rod.h)
        status = msw_ss_damped<T, DIAG, W>(Pc, L, lane, fconst, S, it, L.hist, nn);
        it += it_plain;
        if (lane == 0) { L.red[0] = (T)status; L.red[1] = (T)it; }
      }
      __syncthreads();
      status = (int)L.red[0];
      it = (int)L.red[1];
      __syncthreads();
    }
  }
  if (wave == 0 && lane < 6) A.G[rod * 6 + lane] = L.Xs[0 * 19 + 7 + lane];
  if (wave == 0 && lane == 0) {
    if (A.status) A.status[rod * A.st_stride] = status;
    if (A.iters) A.iters[rod * A.st_stride] = it;
  }
#ifdef KR_MS_STAMPS
  // diagnostic build (tools/msw_stamps.py): the debug buffer is [16][B][T] int32, plane k = quantity k of wavefront W-1
  if (wave == W - 1 && lane == 0 && A.iters) {
    unsigned long long t_end;
    KR_STAMP(t_end);
    const size_t plane = (size_t)A.B * A.st_stride;
    int32_t* o = A.iters + rod * A.st_stride;
    o[1 * plane] = (int32_t)(t_end - t_begin);
    o[2 * plane] = (int32_t)stamps.prep;
    o[3 * plane] = (int32_t)stamps.sweep;
    o[4 * plane] = (int32_t)stamps.alg;
    o[5 * plane] = (int32_t)stamps.osum;
    o[6 * plane] = (int32_t)stamps.a1;
    o[7 * plane] = (int32_t)stamps.a2;
    o[8 * plane] = (int32_t)stamps.a3;
    o[9 * plane] = (int32_t)stamps.a4;
    for (int k = 0; k < 4; ++k) o[(10 + k) * plane] = __float_as_int((float)stamps.dn[k]);
    o[14 * plane] = __float_as_int((float)S.kappa);
    o[15 * plane] = __float_as_int((float)stamps.qa);  // KR_QUICK_AUDIT: (update) / (residual estimate) of this step
  }
#endif
}

// ---- all steps of kr_simulate_batch in one launch: the same solver, the leading slots of the two newest states kept in
// LDS (the history records of the next step never touch HBM), the predictors in registers ---------------------------
// MLP off: history records and the leading slots of the two newest states in LDS.  MLP on (NN): both in global memory
// (the history in a workspace [B][N][12] this workgroup writes and reads, the states where they are, in A.states) - a
// grid point then costs ~20 k cycles of network evaluation, against which an L2-resident 96-byte read is nothing, and
// without them four rods fit the LDS of a CU in fp64 too: 1024 rods run at two wavefronts per SIMD, each sweeping a
// chain half as long, and the waits of one wavefront's evaluator chain are filled by the other's.
// HM, where the time history lives: 0 everything in LDS (short rods, MLP off); 2 everything in global memory (MLP on);
// 1 the history records in LDS, the two newest states in A.states (MLP off, rods whose leading slots do not fit next to
// the records: N = 400, cfg5 - a rod then takes what the one-launch-per-step kernel takes, 76 KB in fp64 at W = 2).
template <typename T, int W>
__host__ __device__ inline size_t msw_sim_lds_elems(int N, bool nn = false, int hm = 0) {
  if (nn) hm = 2;
  return msw_lds_elems<T, W>(N, nn, hm != 2) + (hm == 0 ? (size_t)2 * N * 12 : 0);
}

#ifndef KR_MSW_TAPS
#define KR_MSW_TAPS 3  // taps of the fitted start-value recurrence, MLP off (a compile-time probe: 5 / 7)
#endif
template <typename T, bool DIAG, int W, bool NN = false, int OCC = 1, int HM = NN ? 2 : 0>
__global__ __launch_bounds__(WAVE * W, OCC) void msw_sim_kernel(const RodConst<T> Pc, const SimArgs<T> A, const MlpDev<T> M) {
  static_assert(HM == 2 || !NN, "the MLP-on instantiations keep their history in global memory");
  constexpr bool GH = HM == 2;   // history records in global memory
  constexpr bool GL = HM != 0;   // the two newest states read from A.states instead of an LDS copy of their leading slots
  constexpr int P = MswGeo<W>::P;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = Pc.N;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x / WAVE;
  const int64_t rod = blockIdx.x;
  const size_t rod_elems = (size_t)N * KR_SLOTS;
  T* smem = reinterpret_cast<T*>(smem_raw);
  const MswLds<T, W> L = msw_carve<T, W>(smem, N, NN, !GH);
  T* lead0 = GL ? nullptr : smem + msw_lds_elems<T, W>(N);  // [2][N][12]: newest state / the one before (roles alternate)
  const T* hist = GH ? A.hist_ws + (size_t)rod * N * HS_LEAN : L.hist;
  const int lsz = N * 12;
  const MswRole R = msw_role<W>(wave, lane, N);
  MsStamps stamps;
  if (wave == 0) ms_cold_fill<T>(Pc, L.cold, lane);
  if constexpr (NN) {  // p-column blocks of the Jacobian (msw_newton): none known yet
    for (int e = threadIdx.x; e < W * MSW_BP_W + ((P * 3 + 3) & ~3); e += WAVE * W) L.Bp[e] = T(0);
  }
  const T* s0 = A.states + rod * rod_elems;
  const T* sp = A.prev_init ? A.prev_init + rod * rod_elems : s0;
  if constexpr (!GL) {
    for (int j = threadIdx.x; j < N; j += WAVE * W) {
      T cv[12], pv[12];
      load_hist_vec<T, 12>(s0 + (size_t)j * KR_SLOTS, cv);
      load_hist_vec<T, 12>(sp + (size_t)j * KR_SLOTS, pv);
      store_vec<T, 12>(lead0 + (size_t)j * 12, cv);
      store_vec<T, 12>(lead0 + lsz + (size_t)j * 12, pv);
    }
  }
  const int ne = R.K * 19;
  T* Xl = L.Xs + R.g0 * 19;
  MsPred<T, NN ? KR_NN_TAPS : KR_MSW_TAPS> Q;
  double* img = A.pred_io ? A.pred_io + ((size_t)rod * W + wave) * MS_PRED_ROWS * WAVE : nullptr;
  if (img && A.pred_load) ms_pred_load<T>(Q, img, lane);
  else mswp_init<T>(Q, lane, ne, R.g0, N, P, s0, sp, A.prev_init != nullptr, A.predictor);
  MsSolveArgs<T> S;
  S.tol = A.tol; S.tolA = A.tolA; S.fd_eps = A.fd_eps; S.maxit = A.maxit; S.quick_ok = A.residual_test != 0;
  S.kappa = Q.kappa;
  T Gguess = (wave == 0 && lane < 6) ? A.G[rod * 6 + lane] : T(0);
  const T* ctl = A.ctl + rod * A.T_steps * 4;
  MswNn<T> nn;
#ifdef KR_MS_STAMPS
  unsigned long long t_launch;
  KR_STAMP(t_launch);
  long long sweeps_total = 0;
  int sweeps_hist[3] = {0, 0, 0};
  unsigned long long t_hist = 0, t_guess = 0, t_newton = 0, t_upd = 0, tph;
#endif
  __syncthreads();
  for (int64_t t = 0; t < A.T_steps; ++t) {
    T* prv = nullptr;
#ifdef KR_MS_STAMPS
    KR_STAMP(tph);
#endif
    if constexpr (GL) {
      // BDF2 history (knode.py:74-75) from the two newest states in A.states (this workgroup wrote them: visible after
      // the barrier that ended the step), into the rod's rows of the workspace or its LDS records
      const T* rc = A.states + (A.ring ? t % 3 : t) * A.slot_elems + rod * rod_elems;
      const T* rp = t > 0 ? A.states + (A.ring ? (t - 1) % 3 : t - 1) * A.slot_elems + rod * rod_elems : sp;
      T* hw = GH ? A.hist_ws + (size_t)rod * N * HS_LEAN : L.hist;
      // four grid points per thread at a time, all their loads in flight together: the records come from L2 (this
      // workgroup has just written them), and one round trip per grid point was most of this loop at N = 400
      for (int j0 = threadIdx.x; j0 < N; j0 += 4 * WAVE * W) {
        T cv[4][12], pv[4][12];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = j0 + q * WAVE * W;
          const int jc = j < N ? j : N - 1;  // (clamped: a valid record, not used)
          load_hist_vec<T, 12>(rc + (size_t)jc * KR_SLOTS, cv[q]);
          load_hist_vec<T, 12>(rp + (size_t)jc * KR_SLOTS, pv[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = j0 + q * WAVE * W;
          T hv[12];
#pragma unroll
          for (int k = 0; k < 12; ++k) hv[k] = A.hc1 * cv[q][k] + A.hc2 * pv[q][k];
          if (j < N) store_vec<T, 12>(hw + (size_t)j * HS_LEAN, hv);
        }
      }
      const T* cl = rc + (size_t)(N - 1) * KR_SLOTS;  // z of the last grid point is never touched by a sweep
      S.vlast = {cl[6], cl[7], cl[8]};
      S.ulast = {cl[9], cl[10], cl[11]};
    } else {
      const T* cur = lead0 + (int)(t & 1) * lsz;        // leading slots of the state at time level t
      prv = lead0 + (int)((t + 1) & 1) * lsz;           // ... of level t - 1; the storing sweep overwrites them with level t + 1
      // BDF2 history (knode.py:74-75), raw terms only
      for (int j = threadIdx.x; j < N; j += WAVE * W) {
        T cv[12], pv[12], hv[12];
        load_hist_vec<T, 12>(cur + (size_t)j * 12, cv);
        load_hist_vec<T, 12>(prv + (size_t)j * 12, pv);
#pragma unroll
        for (int k = 0; k < 12; ++k) hv[k] = A.hc1 * cv[k] + A.hc2 * pv[k];
        store_vec<T, 12>(L.hist + (size_t)j * HS_LEAN, hv);
      }
      {  // z of the last grid point is never touched by a sweep
        const T* cl = cur + (size_t)(N - 1) * 12;
        S.vlast = {cl[6], cl[7], cl[8]};
        S.ulast = {cl[9], cl[10], cl[11]};
      }
    }
    __syncthreads();
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(t_hist, tph);
#endif
    V3<T> fconst;
    {
      V3<T> tf{T(0), T(0), T(0)};
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // cosserat_ode.py:195
        const T tt = ctl[t * 4 + k];
        tf.x += tt * L.cold[CD_TDIRS + k * 3 + 0];
        tf.y += tt * L.cold[CD_TDIRS + k * 3 + 1];
        tf.z += tt * L.cold[CD_TDIRS + k * 3 + 2];
      }
      fconst = {L.cold[CD_RHOAG] + tf.x, L.cold[CD_RHOAG + 1] + tf.y, L.cold[CD_RHOAG + 2] + tf.z};
      if constexpr (NN) nn = msw_nn_ctx<T, W>(M, L, R, lane, tf);
    }
    const int64_t inx = A.ring ? (t + 1) % 3 : t + 1;
    S.out_rod = A.states + inx * A.slot_elems + rod * rod_elems;
    S.tip = A.tip ? A.tip + (rod * A.T_steps + t) * 3 : nullptr;
    S.lead12 = prv;
    S.lean = !NN && A.ring && t + 4 <= A.T_steps;  // (the last three states of a call stay complete)
    int order = Q.next_order;
    int status, it;
    while (true) {
      mswp_guess<T>(Q, order, lane, ne, wave == 0, L.cold, Xl);
      wave_sync();
      if (wave == 0 && order <= 0 && lane < 6) L.Xs[0 * 19 + 7 + lane] = Gguess;  // caller's guess (knode.py:67,89)
      __syncthreads();
#ifdef KR_MS_STAMPS
      KR_STAMP_ADD(t_guess, tph);
#endif
      status = msw_newton<T, DIAG, W, NN, OCC - 1>(Pc, L, R, lane, fconst, S, it, stamps, hist, nn);
#ifdef KR_MS_STAMPS
      KR_STAMP_ADD(t_newton, tph);
#endif
      if (status == KR_ST_CONVERGED || order == 0) break;
      order = 0;  // the predicted start did not converge: redo the step from the reference's warm start
      __syncthreads();
    }
    if (status != KR_ST_CONVERGED) {  // (uniform over the workgroup) plain Newton failed from the warm start too
      __syncthreads();
      if (wave == 0) {
        if (lane < 6) L.Xs[0 * 19 + 7 + lane] = Gguess;
        wave_sync();
        const int it_plain = it;  // `iters` reports the plain and the damped phase together (knode_rod.h)
        status = msw_ss_damped<T, DIAG, W, NN, OCC - 1>(Pc, L, lane, fconst, S, it, hist, nn);
        it += it_plain;
        if (lane == 0) { L.red[0] = (T)status; L.red[1] = (T)it; }
      }
      __syncthreads();
      status = (int)L.red[0];
      it = (int)L.red[1];
      __syncthreads();
    }
    if (wave == 0 && lane == 0 && A.status) A.status[rod * A.T_steps + t] = status;
#ifdef KR_MS_STAMPS
    sweeps_total += it;
    sweeps_hist[it <= 2 ? 0 : it == 3 ? 1 : 2] += 1;
#endif
    mswp_update<T, W>(Q, order, status, A.predictor, lane, wave, ne, Xl, L.red);
    if (wave == 0 && lane < 6) Gguess = L.Xs[0 * 19 + 7 + lane];
    __syncthreads();  // the storing lanes' leading slots (and Xs) before the next step reads them
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(t_upd, tph);
#endif
  }
#ifdef KR_MS_STAMPS
  // diagnostic build (tools/msw_sim_stamps.py): per rod {ticks of the launch, sweep ticks, algebra ticks, -, sweeps, steps with
  // <= 2 / 3 / >= 4 sweeps} of wavefront 0
  if (A.dbg && wave == 0 && lane == 0) {
    unsigned long long te;
    KR_STAMP(te);
    unsigned long long* d = A.dbg + rod * 24;
    d[0] = te - t_launch; d[1] = stamps.sweep; d[2] = stamps.alg; d[4] = (unsigned long long)sweeps_total;
    d[5] = (unsigned long long)sweeps_hist[0]; d[6] = (unsigned long long)sweeps_hist[1]; d[7] = (unsigned long long)sweeps_hist[2];
    d[8] = stamps.a1; d[9] = stamps.a2; d[10] = stamps.a3; d[11] = stamps.a4;  // condensation: local chains, barrier, boundary chain, rest
    d[12] = t_hist; d[13] = t_guess; d[14] = t_newton; d[15] = t_upd;           // step loop: history build, start values, Newton, predictor update
    d[16] = stamps.prep; d[17] = stamps.total; d[18] = stamps.osum; d[19] = stamps.retries;  // inside "rest": solve, back-substitution, p rows, norms
  }
#endif
  if (wave == 0 && lane < 6) A.G[rod * 6 + lane] = Gguess;
  if (img) {
    Q.kappa = S.kappa;
    ms_pred_save<T>(Q, img, lane);
  }
}

template <typename T, bool DIAG, int W, bool NN = false, int OCC = 1, int HM = NN ? 2 : 0>
static int launch_msw_sim_inst(const RodConst<T>& P, const MlpDev<T>& M, const SimArgs<T>& a, hipStream_t s) {
  auto kern = msw_sim_kernel<T, DIAG, W, NN, OCC, HM>;
  const size_t smem = sizeof(T) * msw_sim_lds_elems<T, W>(P.N, NN, HM);
  if (int rc_lds_ = dyn_lds(reinterpret_cast<const void*>(kern), smem)) return rc_lds_;
  hipLaunchKernelGGL(kern, dim3((unsigned)a.B), dim3(WAVE * W), smem, s, P, a, M);
  KR_HIP(hipGetLastError());
  return KR_OK;
}
// kr_simulate_batch with several wavefronts per rod: 0 launched (all steps in one launch), 1 does not apply
template <typename T>
static int launch_msw_sim(kr_handle* h, int W, const SimArgs<T>& a, hipStream_t s) {
  const RodConst<T>& P = consts<T>(h);
  const size_t bytes = sizeof(T) * (W == 2 ? msw_sim_lds_elems<T, 2>(P.N) : msw_sim_lds_elems<T, 4>(P.N));
  // (every rod resident at once: a second round of workgroups would wait for the first to finish all steps)
  if (bytes > (size_t)h->lds_limit || a.B > 256 * (int64_t)((size_t)h->lds_limit / bytes))
    return launch_msw_gh_sim<T>(h, W, a, s);  // long rods: the form that reads the two newest states from A.states
  h->last_waves_per_rod = W;
  const MlpDev<T>& M = mlpdev<T>(h);
  if (W == 2) return P.diag ? launch_msw_sim_inst<T, true, 2>(P, M, a, s) : launch_msw_sim_inst<T, false, 2>(P, M, a, s);
  return P.diag ? launch_msw_sim_inst<T, true, 4>(P, M, a, s) : launch_msw_sim_inst<T, false, 4>(P, M, a, s);
}

template <typename T, int W>
static size_t msw_lds_bytes(int N) { return sizeof(T) * msw_lds_elems<T, W>(N); }

// wavefronts per rod this call should use (0: not this kernel).  Auto: batches that leave SIMDs idle (one wavefront per
// SIMD at most, B W <= 1024); the option "waves_per_rod" forces 1 / 2 / 4.  Measured, fp64, us per step
// (tools/msw_timing.py; persistent = all steps of kr_simulate_batch in one launch):
//     N    B    persistent W=1   per step W=4   persistent W=2   persistent W=4
//    40  256        20.5            25.5            18.2             18.6
//    64  256        26.9             -              21.4             20.3
//   100  256        35.9            30.3            26.9             23.6
//   100  512        36.5             -              29.5              -
//   128  256     (47.7 per step)    32.6            31.1             25.6
//   400  256    (115 per step)      59.5             -              52.6     (persistent: the long-rod form, HM = 1)
template <typename T>
int step_waves_per_rod(kr_handle* h, int scheme, int use_nn, int64_t B, int mode) {
  const RodConst<T>& P = consts<T>(h);
  if (use_nn || scheme != KR_EULER || mode != 0 || h->ms_mode == 0) return 0;
  auto fits_bytes = [&](size_t bytes) {
    if (bytes > (size_t)h->lds_limit) return false;
    return B <= 256 * (int64_t)((size_t)h->lds_limit / bytes);
  };
  auto fits = [&](int W) {  // the one-launch-per-step kernel
    if (P.N - 1 < 2 * (4 + 3 * (W - 1)) || B * W > 1024) return false;
    return fits_bytes(W == 2 ? msw_lds_bytes<T, 2>(P.N) : msw_lds_bytes<T, 4>(P.N));
  };
  auto fits_sim = [&](int W) {  // ... and its persistent form
    return fits_bytes(sizeof(T) * (W == 2 ? msw_sim_lds_elems<T, 2>(P.N) : msw_sim_lds_elems<T, 4>(P.N)));
  };
  if (h->waves_per_rod == 1) return 0;
  if (h->waves_per_rod == 2) return fits(2) ? 2 : 0;
  if (h->waves_per_rod == 4) return fits(4) ? 4 : 0;
  if (P.N <= MS_NPL * WAVE) {
    // the persistent one-wavefront kernel serves these: several wavefronts only where their own persistent form fits
    // and the rod is long enough for the shorter chains to pay for the distributed condensation
    if (P.N >= 56 && fits(4) && fits_sim(4)) return 4;
    if (P.N >= 32 && fits(2) && fits_sim(2)) return 2;
    return 0;
  }
  if (fits(4)) return 4;
  if (fits(2)) return 2;
  return 0;
}
template <typename T, bool DIAG, int W>
static int launch_msw_inst(const RodConst<T>& P, const StepArgs<T>& a, hipStream_t s) {
  auto kern = msw_step_kernel<T, DIAG, W>;
  const size_t smem = msw_lds_bytes<T, W>(P.N);
  if (int rc_lds_ = dyn_lds(reinterpret_cast<const void*>(kern), smem)) return rc_lds_;
  hipLaunchKernelGGL(kern, dim3((unsigned)a.B), dim3(WAVE * W), smem, s, P, a);
  KR_HIP(hipGetLastError());
  return KR_OK;
}
template <typename T>
static int launch_msw(kr_handle* h, int W, const StepArgs<T>& a, hipStream_t s) {
  const RodConst<T>& P = consts<T>(h);
  h->last_sim_path = 1;
  h->last_waves_per_rod = W;
  if (W == 2) return P.diag ? launch_msw_inst<T, true, 2>(P, a, s) : launch_msw_inst<T, false, 2>(P, a, s);
  return P.diag ? launch_msw_inst<T, true, 4>(P, a, s) : launch_msw_inst<T, false, 4>(P, a, s);
}

}  // namespace kr
