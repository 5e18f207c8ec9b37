// kr_mswo_impl.hpp - the persistent several-wavefront kernel (kr_msw_impl.hpp: W = 2 or 4 wavefronts share a rod)
// with OVERLAPPED time steps, as kr_mso_impl.hpp does for one wavefront per rod (round 3).
//
// Small batches of medium rods (BASELINE cfg2: B = 256, N = 100) run four wavefronts per rod so that every SIMD of
// the chip has work; their steady state was two short sweeps, one distributed condensation and one residual test per
// time step.  Here every wavefront carries, beside the forward-difference lanes of its 4 (wavefront 0) or 3 intervals,
// one lane per interval that re-integrates the PREVIOUS step from its corrected unknowns, streams the state out and
// forms the history record of the coming step in place, one grid point ahead of the lanes that read it.  A wavefront's
// forward-difference lanes only read history records its own verifying lanes wrote, so the sweep needs no barrier.
// One sweep and one condensation per step remain.
//
// Acceptance of step t: the residual test of kr_msw_impl.hpp on the verifying lanes' end states (estimate from the
// update / residual ratio of step t's last condensation, factor 256).  If it does not accept - or the option
// "residual_test" is off - the kernel is not used / the sweep's work for step t + 1 is dropped, the history of step t
// is rebuilt (leading slots of state t are still in LDS, those of state t - 1 come back from HBM) and step t
// continues with plain sweeps; a step that converges neither from the predicted nor from the warm start goes to
// damped single shooting on wavefront 0 (msw_ss_damped), like in the kernel this one replaces.
#pragma once
// (included by kr_mso_impl.hpp: compiled in the kr_mso_*.hip translation units)

namespace kr {

template <typename T, int W>
__host__ __device__ inline size_t mswo_lds_elems(int N) {
  constexpr int P = MswGeo<W>::P;
  return msw_sim_lds_elems<T, W>(N) + ((P * 19 + 3) & ~3);  // + XsB: unknowns of the step under verification
}

#ifdef KR_MS_STAMPS
struct MswoStats { unsigned long long sweeps = 0, merged = 0, accepted = 0, rejects = 0, retries = 0, damped = 0, t_sweep = 0, t_alg = 0, t_pred = 0; };
#endif

template <typename T, bool DIAG, int W>
__global__ __launch_bounds__(WAVE * W) void mswo_sim_kernel(const RodConst<T> Pc, const SimArgs<T> A) {
  constexpr int P = MswGeo<W>::P;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = Pc.N;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x / WAVE;
  const int64_t rod = blockIdx.x;
  const size_t rod_elems = (size_t)N * KR_SLOTS;
  const int64_t T_steps = A.T_steps;
  T* smem = reinterpret_cast<T*>(smem_raw);
  const MswLds<T, W> L = msw_carve<T, W>(smem, N);
  T* const lead0 = smem + msw_lds_elems<T, W>(N);  // [2][N][12]: leading slots of the newest state / the one being written
  const int lsz = N * 12;
  T* const XsB = lead0 + 2 * (size_t)lsz;          // [P][19]
  T* const Xs = L.Xs;
  T* const Es = L.Es + (size_t)wave * ((64 * 19 + 3) & ~3);
  float* const redf = reinterpret_cast<float*>(L.red);
  const MswRole R = msw_role<W>(wave, lane, N);
  const int iv = R.iv, col = R.col;
  // verifying lanes: one per interval of this wavefront, behind its forward-difference lanes
  const int b0 = wave == 0 ? 58 : 51;
  const bool isA = !R.idle;
  const bool isB = lane >= b0 && lane < b0 + R.K;
  const int kB = isB ? lane - b0 : 0;
  const int gB = R.g0 + kB;
  const int srem = (N - 1) % P;
  const int s_l = isB ? msw_start(gB, N, P) : R.s_i;
  const int len_l = isB ? R.sbase + (gB < srem ? 1 : 0) : R.len_i;
  const int lmax = R.sbase + (srem ? 1 : 0);
  T* const EsB = Es + b0 * 19;
  MsStamps stamps;
  if (wave == 0) ms_cold_fill<T>(Pc, L.cold, lane);
  __syncthreads();

  auto state_ptr = [&](int64_t k) -> T* { return A.states + (A.ring ? k % 3 : k) * A.slot_elems + rod * rod_elems; };
  int cur_i = 0;  // which half of lead0 holds the leading slots of the newest ACCEPTED state
  // history records of step t and the leading slots of state t from the states in HBM (all threads of the workgroup)
  auto rebuild = [&](int64_t t) {
    const T* cs = state_ptr(t);
    const T* ps = t > 0 ? state_ptr(t - 1) : (A.prev_init ? A.prev_init + rod * rod_elems : cs);
    T* cur = lead0 + cur_i * lsz;
    for (int j = threadIdx.x; j < N; j += WAVE * W) {
      T cv[12], pv[12], hv[12];
      load_hist_vec<T, 12>(cs + (size_t)j * KR_SLOTS, cv);
      load_hist_vec<T, 12>(ps + (size_t)j * KR_SLOTS, pv);
      store_vec<T, 12>(cur + (size_t)j * 12, cv);
#pragma unroll
      for (int k = 0; k < 12; ++k) hv[k] = A.hc1 * cv[k] + A.hc2 * pv[k];
      store_vec<T, 12>(L.hist + (size_t)j * HS_LEAN, hv);
    }
    __syncthreads();
  };
  const T* ctl = A.ctl + rod * T_steps * 4;
  auto load_fc = [&](int64_t t) -> V3<T> {  // rhoA g + tendon force of step t (cosserat_ode.py:151,195)
    V3<T> tf{T(0), T(0), T(0)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const T tt = ctl[t * 4 + k];
      tf.x += tt * L.cold[CD_TDIRS + k * 3 + 0];
      tf.y += tt * L.cold[CD_TDIRS + k * 3 + 1];
      tf.z += tt * L.cold[CD_TDIRS + k * 3 + 2];
    }
    return {L.cold[CD_RHOAG] + tf.x, L.cold[CD_RHOAG + 1] + tf.y, L.cold[CD_RHOAG + 2] + tf.z};
  };

  const T* s0 = state_ptr(0);
  const T* sp0 = A.prev_init ? A.prev_init + rod * rod_elems : s0;
  const int ne = R.K * 19;
  T* const Xl = Xs + R.g0 * 19;
  T* const XlB = XsB + R.g0 * 19;
  MsPred<T> Q;
  double* img = A.pred_io ? A.pred_io + ((size_t)rod * W + wave) * MS_PRED_ROWS * WAVE : nullptr;
  if (img && A.pred_load) ms_pred_load<T>(Q, img, lane);
  else mswp_init<T>(Q, lane, ne, R.g0, N, P, s0, sp0, A.prev_init != nullptr, A.predictor);
  MsSolveArgs<T> S;  // (for msw_ss_damped)
  S.tol = A.tol; S.tolA = A.tolA; S.fd_eps = A.fd_eps; S.maxit = A.maxit; S.quick_ok = true;
  {  // z of the last grid point is never touched by a sweep (cosserat_ode.py:198-201)
    const T* cl = s0 + (size_t)(N - 1) * KR_SLOTS;
    S.vlast = {cl[SL_V], cl[SL_V + 1], cl[SL_V + 2]};
    S.ulast = {cl[SL_U], cl[SL_U + 1], cl[SL_U + 2]};
  }
  const T tol = A.tol, tolA = A.tolA, fd_eps = A.fd_eps;
  const int maxit = A.maxit;
  T kappa = Q.kappa;
  T Gguess = (wave == 0 && lane < 6) ? A.G[rod * 6 + lane] : T(0);
#ifdef KR_MS_STAMPS
  MswoStats st;
  unsigned long long tq;
  KR_STAMP(tq);
#endif

  rebuild(0);

  int64_t tA = 0;
  bool merged = false;
  int it = 0;
  int order = Q.next_order;
  bool retried = false;
  T dn_prev = T(-1);
  bool below = false;
  float amp = -1.f;
  T dnB = T(-1);
  float ampB = -1.f;
  bool belowB = false;
  int itB = 0, orderB = 0;
  bool pred_skip = false;
  V3<T> fcA = load_fc(0), fcB = fcA;
  V3<T> fcN = load_fc(T_steps > 1 ? 1 : 0);

  mswp_guess<T>(Q, order, lane, ne, wave == 0, L.cold, Xl);
  wave_sync();
  if (wave == 0 && order <= 0 && lane < 6) Xs[0 * 19 + 7 + lane] = Gguess;  // caller's guess (knode.py:67,89)
  __syncthreads();

  // a step finished by damped single shooting (wavefront 0): its state went to HBM and its leading slots into the free
  // half of lead0, like after a verifying sweep; what is left is the history of the next step
  auto finish_classic_step = [&](int64_t t) {
    const T* cur = lead0 + cur_i * lsz;
    const T* nxt = lead0 + (cur_i ^ 1) * lsz;
    for (int j = threadIdx.x; j < N; j += WAVE * W) {
      T cv[12], nv[12], hv[12];
      load_hist_vec<T, 12>(cur + (size_t)j * 12, cv);
      load_hist_vec<T, 12>(nxt + (size_t)j * 12, nv);
#pragma unroll
      for (int k = 0; k < 12; ++k) hv[k] = A.hc1 * nv[k] + A.hc2 * cv[k];
      store_vec<T, 12>(L.hist + (size_t)j * HS_LEAN, hv);
    }
    cur_i ^= 1;
    __syncthreads();
  };

  while (true) {
    const bool runA = tA < T_steps;
    const int64_t tB = tA - 1;
    // ---- start state of this lane -------------------------------------------------------------------------------
    T yr[19];
    {
      const T* src = isB ? XsB + gB * 19 : Xs + iv * 19;
#pragma unroll
      for (int q = 0; q < 19; ++q) yr[q] = src[q];
    }
    const T hstep = (isA && col > 0) ? fd_eps * fmax(fabs(Xs[iv * 19 + (R.comp > 0 ? R.comp : 3)]), T(1)) : T(1);
#pragma unroll
    for (int q = 3; q < 19; ++q) yr[q] += (isA && q == R.comp) ? hstep : T(0);
    RodState<T> y = rows_to_state(yr);
    const V3<T> fc = isB ? fcB : fcA;
    T hv[HS_LEAN];
    load_hist_vec<T, HS_LEAN>(L.hist + (size_t)s_l * HS_LEAN, hv);
    if (!merged) {
      auto fd_point = [&](int j, T dsl) __attribute__((always_inline)) {
        RodState<T> k1;
        V3<T> v, u;
        ode_eval<T, DIAG>(Pc, y, hist_lean<T, DIAG>(Pc, hv), fc, k1, v, u);
        load_hist_vec<T, HS_LEAN>(L.hist + (size_t)(j + 1) * HS_LEAN, hv);
        y = state_axpy(y, dsl, k1);
      };
#pragma unroll KR_MS_UNROLL
      for (int t = 0; t < R.sbase; ++t) fd_point(R.s_i + t, Pc.ds);
      if (srem) fd_point(R.s_i + (R.len_i > R.sbase ? R.sbase : R.sbase - 1), R.len_i > R.sbase ? Pc.ds : T(0));
    } else {
      const int lag = isA ? MSO_LAG : 0;
      const bool act = isB || (isA && runA);
      const int trips = lmax + (runA ? MSO_LAG : 0);
      T* const out_rod = state_ptr(tB + 1);
      const T* cur = lead0 + cur_i * lsz;       // leading slots of state tB
      T* nxt = lead0 + (cur_i ^ 1) * lsz;       // ... of state tB + 1, written by the verifying lanes
      auto point_of = [&](int k) -> int {
        const int kk = k - lag;
        return s_l + (kk < 0 ? 0 : (kk < len_l ? kk : len_l - 1));
      };
      auto trip = [&](int k, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int kk = k - lag;
        const bool live = FULL ? act : (act && kk >= 0 && kk < len_l);
        const int j = FULL ? s_l + kk : point_of(k);
        T old[12];
        load_hist_vec<T, 12>(cur + (size_t)j * 12, old);
        RodState<T> k1;
        V3<T> v, u;
        ode_eval<T, DIAG>(Pc, y, hist_lean<T, DIAG>(Pc, hv), fc, k1, v, u);
        if (isB && live) {
          // state tB + 1 at grid point j: to HBM; its leading slots into the free half of lead0; with those of state tB
          // the history record of step tB + 1 (knode.py:74-75), in place of the record of step tB this lane has just used
          T rec[KR_SLOTS];
          record_from(y, v, u, rec);
          store_record(out_rod + (size_t)j * KR_SLOTS, rec);
          T lead[12], hrec[HS_LEAN];
#pragma unroll
          for (int c = 0; c < 12; ++c) { lead[c] = rec[c]; hrec[c] = A.hc1 * rec[c] + A.hc2 * old[c]; }
          store_vec<T, 12>(nxt + (size_t)j * 12, lead);
          store_vec<T, HS_LEAN>(L.hist + (size_t)j * HS_LEAN, hrec);
        }
        load_hist_vec<T, HS_LEAN>(L.hist + (size_t)(FULL ? j + 1 : point_of(k + 1)) * HS_LEAN, hv);
        const T dsl = live ? Pc.ds : T(0);
        y = state_axpy(y, dsl, k1);
      };
      int k = 0;
      for (; k < MSO_LAG && k < trips; ++k) trip(k, std::false_type{});
#pragma unroll 2
      for (; k < R.sbase; ++k) trip(k, std::true_type{});
      for (; k < trips; ++k) trip(k, std::false_type{});
    }
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(st.t_sweep, tq);
    st.sweeps += 1;
    if (merged) st.merged += 1;
#endif

    // =============================================================================================================
    // verdict on the step under verification (all wavefronts: workgroup-wide maxima)
    // =============================================================================================================
    if (merged) {
      T* nxt = lead0 + (cur_i ^ 1) * lsz;
      if (isB) {
        T er[19];
        state_to_rows(y, er);
#pragma unroll
        for (int q = 0; q < 19; ++q) EsB[kB * 19 + q] = er[q];
        if (gB == P - 1) {  // the last grid point: y from the sweep, z untouched
          T rec[KR_SLOTS];
          record_from(y, S.vlast, S.ulast, rec);
          store_record(state_ptr(tB + 1) + (size_t)(N - 1) * KR_SLOTS, rec);
          T lead[12];
#pragma unroll
          for (int c = 0; c < 12; ++c) lead[c] = rec[c];
          store_vec<T, 12>(nxt + (size_t)(N - 1) * 12, lead);
          if (A.tip) {
            T* tp = A.tip + (rod * T_steps + tB) * 3;
            tp[0] = y.p.x; tp[1] = y.p.y; tp[2] = y.p.z;
          }
        }
      }
      wave_sync();
      float rn = 0.f;
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int e = lane + 64 * q2;
        const int k = e / 19, q = e - 19 * k;
        const int g = R.g0 + k;
        if (k < R.K && g < P - 1) {
          const T x = XsB[(g + 1) * 19 + q];
          rn = fmaxf(rn, update_ratio(EsB[k * 19 + q] - x, x));
        }
      }
      if (wave == W - 1 && lane >= 58) {  // tip condition (lanes the loop above never uses on the last wavefront)
        const int k = lane - 58;
        const T e = EsB[2 * 19 + 7 + k];
        rn = fmaxf(rn, update_ratio(L.cold[CD_FTIP + k] - e, e));
      }
      rn = msw_max<W>(rn, redf, wave, lane);
      const float est = ampB * rn;
      const bool accepted = ampB > 0.f && T(256) * (T)est <= tol;  // (NaN compares false)
      if (accepted) {
#ifdef KR_MS_STAMPS
        st.accepted += 1;
#endif
        if (!belowB && dnB > T(0)) {
          const T floor_dn = T(64) * (sizeof(T) == 8 ? T(2.2e-16) : T(1.2e-7));
          kappa = fmin(fmax(fmax((T)est, floor_dn) * fast_rcp(dnB * dnB), T(1e-4)), T(1));
        }
        if (wave == 0 && lane == 0 && A.status) A.status[rod * T_steps + tB] = KR_ST_CONVERGED;
        if (wave == 0 && lane < 6) Gguess = XsB[0 * 19 + 7 + lane];
        cur_i ^= 1;  // the verifying lanes' leading slots are those of the newest accepted state now
        pred_skip = false;
        merged = false;
        if (!runA) break;
      } else {
        // rejected: drop the work done for step tA, put the history of step tB back, continue it with plain sweeps
#ifdef KR_MS_STAMPS
        st.rejects += 1;
#endif
        __syncthreads();
        for (int e = threadIdx.x; e < P * 19; e += WAVE * W) Xs[e] = XsB[e];
        tA = tB;
        fcN = fcA;
        fcA = fcB;
        order = orderB;
        it = itB;
        dn_prev = dnB;
        amp = ampB;
        below = false;
        pred_skip = true;
        merged = false;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");  // the records streamed out above are re-read below
        rebuild(tA);
        continue;
      }
    }

    // =============================================================================================================
    // Newton update of step tA from the forward-difference sweep
    // =============================================================================================================
    ++it;
    MswUpd<T> U;
#ifdef KR_MS_STAMPS
    unsigned long long ta = tq;
    msw_condense<T, W>(L, R, lane, y, hstep, U, stamps, ta);
#else
    msw_condense<T, W>(L, R, lane, y, hstep, U);
#endif
    const float dnf = U.dnf;
    const bool finite = dnf <= 3.0e38f;
    const T dn = (T)dnf;
    if (finite && U.res_local > 0.f) amp = dnf / U.res_local;
    if (finite && !below && dn <= tol) {
      below = true;
      if (dn_prev > T(0)) {
        const T floor_dn = T(64) * (sizeof(T) == 8 ? T(2.2e-16) : T(1.2e-7));
        kappa = fmin(fmax(fmax(dn, floor_dn) * fast_rcp(dn_prev * dn_prev), T(1e-4)), T(1));
      }
    }
    bool next_final = false;
    if (finite) {
      msw_apply<T, W>(L, R, lane, U);
      next_final = predict_final<T>(dn, dn_prev, tol, tolA) || (kappa > T(0) && T(4) * kappa * dn * dn <= tol);
      dn_prev = dn;
    }
    __syncthreads();
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(st.t_alg, tq);
#endif
    const bool go_merged = finite && next_final && dn <= T(1e-2) && amp > 0.f;
    if (!finite || (it >= maxit && !go_merged)) {
      // no root from this start: once more from the reference's warm start (knode.py:89), then damped single shooting
      if (order > 0 && !retried) {
        retried = true;
        order = 0;
        mswp_guess<T>(Q, 0, lane, ne, wave == 0, L.cold, Xl);
        wave_sync();
        if (wave == 0 && lane < 6) Xs[0 * 19 + 7 + lane] = Gguess;
        __syncthreads();
        it = 0; dn_prev = T(-1); amp = -1.f; below = false;
#ifdef KR_MS_STAMPS
        st.retries += 1;
#endif
        continue;
      }
#ifdef KR_MS_STAMPS
      st.damped += 1;
#endif
      int status = KR_ST_MAXIT;
      S.out_rod = state_ptr(tA + 1);
      S.tip = A.tip ? A.tip + (rod * T_steps + tA) * 3 : nullptr;
      S.lead12 = lead0 + (cur_i ^ 1) * lsz;
      S.kappa = kappa;
      if (wave == 0) {
        if (lane < 6) Xs[0 * 19 + 7 + lane] = Gguess;
        wave_sync();
        int itd;
        status = msw_ss_damped<T, DIAG, W>(Pc, L, lane, fcA, S, itd);
        if (lane == 0) L.red[0] = (T)status;
      }
      __syncthreads();
      status = (int)L.red[0];
      __syncthreads();
      if (wave == 0 && lane == 0 && A.status) A.status[rod * T_steps + tA] = status;
      if (wave == 0 && lane < 6) Gguess = Xs[0 * 19 + 7 + lane];
      finish_classic_step(tA);
      if (!pred_skip) mswp_update<T, W>(Q, order, status, A.predictor, lane, wave, ne, Xl, L.red);
      pred_skip = false;
      tA += 1;
      if (tA >= T_steps) break;
      fcB = fcA; fcA = fcN; fcN = load_fc(tA + 1 < T_steps ? tA + 1 : tA);
      order = Q.next_order;
      mswp_guess<T>(Q, order, lane, ne, wave == 0, L.cold, Xl);
      wave_sync();
      if (wave == 0 && order <= 0 && lane < 6) Xs[0 * 19 + 7 + lane] = Gguess;
      __syncthreads();
      it = 0; retried = false; dn_prev = T(-1); amp = -1.f; below = false;
      continue;
    }
    if (go_merged) {
      // hand step tA to the verifying lanes and move the forward-difference lanes on to step tA + 1
      for (int e = threadIdx.x; e < P * 19; e += WAVE * W) XsB[e] = Xs[e];
      __syncthreads();
      dnB = dn; ampB = amp; belowB = below; itB = it; orderB = order;
      if (!pred_skip) mswp_update<T, W>(Q, order, KR_ST_CONVERGED, A.predictor, lane, wave, ne, XlB, L.red);
      tA += 1;
      fcB = fcA;
      if (tA < T_steps) {
        fcA = fcN;
        fcN = load_fc(tA + 1 < T_steps ? tA + 1 : tA);
        order = Q.next_order;
        mswp_guess<T>(Q, order, lane, ne, wave == 0, L.cold, Xl);
        wave_sync();
        if (wave == 0 && order <= 0 && lane < 6) Xs[0 * 19 + 7 + lane] = XsB[0 * 19 + 7 + lane];
      }
      __syncthreads();
      it = 0; retried = false; below = false; dn_prev = T(-1);
      merged = true;
#ifdef KR_MS_STAMPS
      KR_STAMP_ADD(st.t_pred, tq);
#endif
    }
  }

  if (wave == 0 && lane < 6) A.G[rod * 6 + lane] = Gguess;
  if (img) {
    Q.kappa = kappa;
    ms_pred_save<T>(Q, img, lane);
  }
#ifdef KR_MS_STAMPS
  if (wave == 0 && lane == 0 && A.dbg) {
    unsigned long long* dd = A.dbg + rod * 24;
    dd[1] = st.t_sweep; dd[2] = st.t_alg; dd[3] = st.t_pred; dd[4] = st.sweeps; dd[5] = st.merged; dd[6] = st.accepted;
    dd[8] = st.rejects; dd[9] = st.retries; dd[10] = st.damped;
  }
#endif
}

template <typename T, bool DIAG, int W>
static int launch_mswo_sim_inst(const RodConst<T>& P, const SimArgs<T>& a, hipStream_t s) {
  auto kern = mswo_sim_kernel<T, DIAG, W>;
  const size_t smem = sizeof(T) * mswo_lds_elems<T, W>(P.N);
  static thread_local size_t configured = 0;
  if (smem > 48 * 1024 && smem > configured) {
    KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    configured = smem;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)a.B), dim3(WAVE * W), smem, s, P, a);
  KR_HIP(hipGetLastError());
  return KR_OK;
}
// 0 launched, 1 does not apply (the caller runs the plain several-wavefront persistent kernel)
template <typename T>
int launch_mswo_sim(kr_handle* h, int W, const SimArgs<T>& a, hipStream_t s) {
  const RodConst<T>& P = consts<T>(h);
  if (!h->overlap || !h->residual_test || !a.residual_test) return 1;  // (acceptance here IS the residual test)
  const size_t bytes = sizeof(T) * (W == 2 ? mswo_lds_elems<T, 2>(P.N) : mswo_lds_elems<T, 4>(P.N));
  if (bytes > (size_t)h->lds_limit) return 1;
  const int64_t per_cu = (int64_t)((size_t)h->lds_limit / bytes);
  if (a.B > 256 * per_cu) return 1;
  if (W == 2) return P.diag ? launch_mswo_sim_inst<T, true, 2>(P, a, s) : launch_mswo_sim_inst<T, false, 2>(P, a, s);
  return P.diag ? launch_mswo_sim_inst<T, true, 4>(P, a, s) : launch_mswo_sim_inst<T, false, 4>(P, a, s);
}

}  // namespace kr
