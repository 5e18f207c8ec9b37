// kr_mswo_impl.hpp - several wavefronts per rod with OVERLAPPED time steps (round 5).
//
// kr_msw_impl.hpp's persistent kernel spends, in its steady state, exactly two sweeps on every time step (measured:
// tools/msw_sim_stamps.py, 2.000 for every rod of BASELINE cfg2 and cfg5): the forward-difference sweep that yields the
// Newton correction and a second one at the corrected unknowns whose only job is to measure the (negligible) residual
// and to stream the state out - 51 or 58 lanes of every wavefront recompute a Jacobian nobody uses.  kr_mso_impl.hpp
// removed that second sweep for one wavefront per rod; this file carries the same protocol to W wavefronts:
//     lanes 0 .. 57 (wavefront 0) / 0 .. 50 (others)   forward-difference sweep of step t + 1 from its predicted start
//     the next 4 / 3 lanes                             re-integrate step t from its corrected unknowns, one lane per
//                                                      sub-interval of the wavefront, one grid point ahead: they stream
//                                                      the state out and form the BDF2 history record of step t + 1 in
//                                                      place, which the forward-difference lanes pick up a trip later
// A sub-interval's verifying lane lives in the wavefront that owns the interval, so a sweep needs no barrier; every
// DECISION (residual test, stopping rule, hand-over, give-up) is formed from workgroup-wide reductions and therefore
// uniform over the W wavefronts, which meet at the barriers of the distributed condensation (msw_condense) exactly as
// in kr_msw_impl.hpp.  Acceptance of step t is the residual test on the verifying sweep (this kernel family has no
// chord update); a step it does not accept goes back to forward-difference sweeps at the same unknowns, whose Newton
// update is the measured quantity; a step that does not converge that way is solved by kr_msw_impl.hpp's full ladder
// (msw_newton from the warm start, damped single shooting) inside this kernel, after which the overlap resumes.
//
// Serves: Euler sweeps, diagonal material matrices, MLP off, rods whose history, leading slots and unknowns fit the
// LDS next to the condensation tiles (short rods in small batches: BASELINE cfg2).
#pragma once
#include "kr_mso_impl.hpp"
// KR_MSWO_LOOP: compile-time probes of the merged sweep's loop structure (cfg2, us per step): 0 = peeled (first trip, paired
// unpredicated trips, last trips) 21.7 - 22.0; 1 = one loop over the predicated trip 22.3; 2 = one loop, grid point clamped by
// min / max, one unsigned compare as predicate 21.8 - 21.9 (level with 0)
#ifndef KR_MSWO_LOOP
#define KR_MSWO_LOOP 0
#endif

namespace kr {

// GT ("global tiles"): rods whose three tiles of leading slots do not fit the LDS (N = 400 at two rods per CU) read the leading
// slots straight from the states in HBM / L2 - a state's record holds them in its first twelve slots, and on a 3-slot ring the
// three slots ARE the three tiles.  The reads of the next trip then go out at the head of a trip (the verifying lanes run TWO
// grid points ahead), under a grid point's worth of arithmetic.
template <typename T, int W, bool GT = false>
__host__ __device__ inline size_t mswo_lds_elems(int N) {
  constexpr int P = MswGeo<W>::P;
  return msw_lds_elems<T, W>(N) + (GT ? 0 : 3 * (((size_t)N * 12 + 3) & ~size_t(3))) + ((P * 19 + 3) & ~3);
}

// Scaled maximum norm of this wavefront's part of the residual of a VERIFYING sweep: interface jumps E_g - Y_{g+1} of its
// intervals and, on the last wavefront, the tip condition - end states in the Es slots of the verifying lanes (nact + k),
// unknowns XsB.  Not yet reduced over the workgroup.
template <typename T, int W>
__device__ __forceinline__ float mswo_residual_B(const T* Es, const T* XsB, const T* cold, const MswRole& R, int nact, int lane) {
  constexpr int P = MswGeo<W>::P;
  float rn = 0.f;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = lane + 64 * q;
    const int k = e / 19, r = e - 19 * k;
    const int g = R.g0 + k;
    if (k < R.K && g < P - 1) {
      const T x = XsB[(g + 1) * 19 + r];
      rn = fmaxf(rn, update_ratio(Es[(nact + k) * 19 + r] - x, x));
    }
  }
  if (R.w == W - 1 && lane >= 58) {
    const int k = lane - 58;
    const T e = Es[(nact + R.K - 1) * 19 + 7 + k];
    rn = fmaxf(rn, update_ratio(cold[CD_FTIP + k] - e, e));
  }
  return rn;
}

template <typename T, int W, bool GT = false>
__global__ __launch_bounds__(WAVE * W) void mswo_sim_kernel(const RodConst<T> Pc, const SimArgs<T> A) {
  constexpr bool DIAG = true;
  constexpr int LAGV = GT ? 2 : MSO_LAG;   // grid points the verifying lanes run ahead
  constexpr int LS = GT ? KR_SLOTS : 12;   // stride of the leading slots of consecutive grid points
  constexpr int P = MswGeo<W>::P;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = Pc.N;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x / WAVE;
  const int64_t rod = blockIdx.x;
  const size_t rod_elems = (size_t)N * KR_SLOTS;
  const int64_t T_steps = A.T_steps;
  if (T_steps <= 0) return;
  T* smem = reinterpret_cast<T*>(smem_raw);
  const MswLds<T, W> L = msw_carve<T, W>(smem, N);
  // leading slots (q w v u) of the three newest states, state s in tile s mod 3: a workgroup has a CU's LDS to itself here, so
  // a step can be rolled back without the states in HBM - and on a 3-slot ring the interior records of interior steps are
  // not stored at all (nobody reads them: the caller gets the tips and the last three states)
  const size_t lsz = ((size_t)N * 12 + 3) & ~size_t(3);
  T* const lead3 = smem + msw_lds_elems<T, W>(N);                   // [3][N][12] (not with GT)
  T* const XsB = lead3 + (GT ? 0 : 3 * lsz);                        // [P][19] unknowns of the step under verification
  auto tile = [&](int64_t st_) -> T* {
    if constexpr (GT) {  // the state itself (its leading slots come first in every record)
      return st_ >= 0 ? A.states + (A.ring ? st_ % 3 : st_) * A.slot_elems + rod * rod_elems
                      : (A.prev_init ? const_cast<T*>(A.prev_init) + rod * rod_elems : A.states + rod * rod_elems);
    } else {
      return lead3 + (size_t)((st_ + 3) % 3) * lsz;
    }
  };
  T* const Xs = L.Xs;
  T* const Es = L.es(wave);
  float* const redf = reinterpret_cast<float*>(L.red);
  const MswRole R = msw_role<W>(wave, lane, N);
  const int nact = wave == 0 ? 58 : 51;
  const int col = R.col;
  const bool isA = !R.idle;
  const bool isB = R.idle && lane - nact < R.K;
  const int ib = isB ? lane - nact : 0;
  const int srem = (N - 1) % P;
  const int gB = R.g0 + ib;
  const int s_l = isB ? msw_start(gB, N, P) : R.s_i;
  const int len_l = isB ? R.sbase + (gB < srem ? 1 : 0) : R.len_i;
  const int lmax = R.sbase + (R.g0 < srem ? 1 : 0);  // (the long intervals come first)
  const int ne = R.K * 19;
  T* const Xl = Xs + R.g0 * 19;
  T* const XlB = XsB + R.g0 * 19;
  if (wave == 0) ms_cold_fill<T>(Pc, L.cold, lane);
  __syncthreads();

  auto state_ptr = [&](int64_t k) -> T* { return A.states + (A.ring ? k % 3 : k) * A.slot_elems + rod * rod_elems; };
  // history records of step t (knode.py:74-75, raw terms) and the leading slots of state t from the states in HBM (this
  // workgroup wrote them: visible after the barrier in front)
  // leading slots of state t from HBM into its tile (the records of a state this workgroup has just stored in full: visible
  // after the barrier in front; with_prev: and those of the state before it - the start of a call)
  auto tile_from_hbm = [&](int64_t t, bool with_prev) {
    __syncthreads();
    if constexpr (GT) return;  // (the states are the tiles)
    const T* cur = state_ptr(t);
    const T* prv = t > 0 ? state_ptr(t - 1) : (A.prev_init ? A.prev_init + rod * rod_elems : cur);
    for (int j = threadIdx.x; j < N; j += WAVE * W) {
      T cv[12];
      load_hist_vec<T, 12>(cur + (size_t)j * KR_SLOTS, cv);
      store_vec<T, 12>(tile(t) + (size_t)j * 12, cv);
      if (with_prev) {
        T pv[12];
        load_hist_vec<T, 12>(prv + (size_t)j * KR_SLOTS, pv);
        store_vec<T, 12>(tile(t - 1) + (size_t)j * 12, pv);
      }
    }
    __syncthreads();
  };
  // history records of step t (knode.py:74-75, raw terms) from the tiles of the states t and t - 1
  auto rebuild = [&](int64_t t) {
    msw_lds_barrier();
    const T* cur = tile(t);
    const T* prv = tile(t - 1);
    for (int j = threadIdx.x; j < N; j += WAVE * W) {
      T cv[12], pv[12], hv[12];
      load_hist_vec<T, 12>(cur + (size_t)j * LS, cv);
      load_hist_vec<T, 12>(prv + (size_t)j * LS, pv);
#pragma unroll
      for (int k = 0; k < 12; ++k) hv[k] = A.hc1 * cv[k] + A.hc2 * pv[k];
      store_vec<T, 12>(L.hist + (size_t)j * HS_LEAN, hv);
    }
    msw_lds_barrier();
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
  MsPred<T, KR_MSW_TAPS> Q;
  double* img = A.pred_io ? A.pred_io + ((size_t)rod * W + wave) * MS_PRED_ROWS * WAVE : nullptr;
  if (img && A.pred_load) ms_pred_load<T>(Q, img, lane);
  else mswp_init<T>(Q, lane, ne, R.g0, N, P, s0, sp0, A.prev_init != nullptr, A.predictor);
  MsSolveArgs<T> S;  // (the plain ladder of kr_msw_impl.hpp, for steps the overlapped iteration gives up on)
  {
    const T* cl = s0 + (size_t)(N - 1) * KR_SLOTS;  // z of the last grid point is never touched by a sweep
    S.vlast = {cl[SL_V], cl[SL_V + 1], cl[SL_V + 2]};
    S.ulast = {cl[SL_U], cl[SL_U + 1], cl[SL_U + 2]};
  }
  S.tol = A.tol; S.tolA = A.tolA; S.fd_eps = A.fd_eps; S.maxit = A.maxit; S.quick_ok = A.residual_test != 0;
  S.lead12 = nullptr;
  const T tol = A.tol, tolA = A.tolA, fd_eps = A.fd_eps;
  const int maxit = A.maxit;
  T kappa = Q.kappa;
  T Gguess = (wave == 0 && lane < 6) ? A.G[rod * 6 + lane] : T(0);
  MsStamps stamps;
  const MswNn<T> nn;  // (MLP off)
#ifdef KR_MS_STAMPS
  unsigned long long o_sweeps = 0, o_merged = 0, o_accept = 0, o_reject = 0, o_plain = 0, o_stored = 0, o_retry = 0, o_chord = 0, t_begin, t_sw = 0, t_verd = 0, t_cond = 0, t_hand = 0, tq;
  KR_STAMP(t_begin);
  tq = t_begin;
#endif

  tile_from_hbm(0, true);

  // ---- state of the iteration (every variable below is uniform over the workgroup) --------------------------------
  int64_t tA = 0;        // step the forward-difference lanes work on (unknowns: Xs)
  bool merged = false;   // the coming sweep also re-integrates step tA - 1 from XsB and streams it out
  int it = 0;            // forward-difference sweeps spent on step tA
  int order = Q.next_order;
  bool retried = false;  // step tA has been restarted from the reference's warm start
  T dn_prev = T(-1);     // update norm of the previous iteration of step tA
  bool below = false;
  float amp = -1.f;      // update norm per unit of residual norm at the last condensation
  // of the step under verification (tA - 1)
  T dnB = T(-1);
  float ampB = -1.f;
  bool belowB = false;
  int itB = 0, orderB = 0;
  bool pred_skip = false;    // the predictor has already consumed step tA (a verifying sweep of it was rejected)
  bool stored_at_xs = false; // the state of step tA in HBM was streamed from the unknowns now in Xs (a rejected verifying sweep)
  bool reverify = false;     // the coming merged sweep verifies step tA - 1 for the second time (after a chord update)
  V3<T> fcA = load_fc(0), fcB = fcA;
  V3<T> fcN = load_fc(T_steps > 1 ? 1 : 0);

  auto start_guess = [&](int ord) {
    mswp_guess<T>(Q, ord, lane, ne, wave == 0, L.cold, Xl);
    wave_sync_lds();
    if (wave == 0 && ord <= 0 && lane < 6) Xs[0 * 19 + 7 + lane] = Gguess;  // caller's guess / the G just found (knode.py:67,89)
    msw_lds_barrier();
  };
  start_guess(order);

  // Step tA by the full ladder of kr_msw_impl.hpp's persistent kernel (warm start, plain Newton with storing sweeps, damped
  // single shooting); the history of step tA is in L.hist and c12.  Leaves everything ready for step tA + 1 in plain mode.
  auto plain_step = [&]() {
#ifdef KR_MS_STAMPS
    o_plain += 1;
#endif
    S.out_rod = state_ptr(tA + 1);
    S.tip = A.tip ? A.tip + (rod * T_steps + tA) * 3 : nullptr;
    S.kappa = kappa;
    rebuild(tA);  // (the plain ladder reads combined records: L.hist)
    start_guess(0);
    int its = 0;
    int status = msw_newton<T, DIAG, W>(Pc, L, R, lane, fcA, S, its, stamps, L.hist, nn);
    if (status != KR_ST_CONVERGED) {  // (uniform over the workgroup)
      __syncthreads();
      if (wave == 0) {
        if (lane < 6) Xs[0 * 19 + 7 + lane] = Gguess;
        wave_sync();
        int it2 = 0;
        status = msw_ss_damped<T, DIAG, W>(Pc, L, lane, fcA, S, it2, L.hist, nn);
        if (lane == 0) L.red[0] = (T)status;
      }
      __syncthreads();
      status = (int)L.red[0];
      __syncthreads();
    }
    kappa = S.kappa;
    if (wave == 0 && lane == 0 && A.status) A.status[rod * T_steps + tA] = status;
    if (!pred_skip) mswp_update<T, W>(Q, 0, status, A.predictor, lane, wave, ne, Xl, L.red);
    else if (status != KR_ST_CONVERGED) { Q.next_order = 0; Q.avail = -1; }
    pred_skip = false;
    if (wave == 0 && lane < 6) Gguess = Xs[0 * 19 + 7 + lane];
    tA += 1;
    fcB = fcA;
    tile_from_hbm(tA, false);  // (its leading barrier also drains the record stream of the storing sweep)
    if (tA < T_steps) {
      fcA = fcN;
      fcN = load_fc(tA + 1 < T_steps ? tA + 1 : tA);
      order = Q.next_order;
      start_guess(order);
    }
    it = 0; retried = false; below = false; dn_prev = T(-1); amp = -1.f;
    merged = false; stored_at_xs = false;
  };

  while (tA < T_steps || merged) {
    const bool runA = tA < T_steps;  // (false only for the sweep that verifies the last step)
    const int64_t tB = tA - 1;
    // ---- start state of this lane -------------------------------------------------------------------------------
    T yr[19];
    {
      const T* src = isB ? XsB + gB * 19 : Xs + R.iv * 19;
#pragma unroll
      for (int q = 0; q < 19; ++q) yr[q] = src[q];
    }
    const T hstep = (isA && col > 0) ? fd_eps * fmax(fabs(Xs[R.iv * 19 + (R.comp > 0 ? R.comp : 3)]), T(1)) : T(1);
#pragma unroll
    for (int q = 3; q < 19; ++q) yr[q] += (isA && q == R.comp) ? hstep : T(0);
    RodState<T> y = rows_to_state(yr);
    const V3<T> fc = isB ? fcB : fcA;
    // the verifying lanes run MSO_LAG grid points ahead of the forward-difference lanes that consume their records
    const int lag = (isA && merged) ? LAGV : 0;
    const bool act = isB ? merged : (isA && runA);
    const int trips = lmax + ((merged && runA) ? LAGV : 0);
    T* const out_rod = state_ptr(tB + 1);  // (used by the verifying lanes only)
    auto point_of = [&](int k) -> int {
      const int kk = k - lag;
      return s_l + (kk < 0 ? 0 : (kk < len_l ? kk : len_l - 1));
    };
    // History of this lane's step at grid point j from the tiles of leading slots (as kr_mso_impl.hpp: every lane forms its
    // own record, 30 instructions all lanes share; the verifying lanes only leave their twelve leading slots).  The
    // forward-difference lanes work on step tA (states tA, tA - 1), the verifying lanes on step tB = tA - 1 (states tB, tB - 1)
    // and write the tile of state tB + 1 = tA - with three tiles nothing anybody still reads.
    const T* const lead_n = tile(isB ? tB : tA);
    const T* const lead_o = tile(isB ? tB - 1 : tB);
    auto lead_load = [&](const T* p, int j, T (&v)[12]) __attribute__((always_inline)) {
      if constexpr (GT) load_hist_vec<T, 12>(p + (size_t)j * LS, v);
      else lds_load_vec<T, 12>(p + (size_t)j * 12, v);
    };
    auto hist_at = [&](int j, T (&hv)[HS_LEAN]) __attribute__((always_inline)) {
      T la[12], lb[12];
      lead_load(lead_n, j, la);
      lead_load(lead_o, j, lb);
#pragma unroll
      for (int c = 0; c < 12; ++c) hv[c] = A.hc1 * la[c] + A.hc2 * lb[c];
    };
    T hv[HS_LEAN];
    hist_at(point_of(0), hv);
    if (!merged) {
      // plain forward-difference sweep (start-up, rough inputs, after a rejection); every interval has sbase or sbase + 1
      // segments, so only the last grid point needs a predicate
      auto fd_point = [&](int j, T dsl) __attribute__((always_inline)) {
        RodState<T> k1;
        V3<T> v, u;
        ode_eval<T, DIAG>(Pc, y, hist_lean<T, DIAG>(Pc, hv), fc, k1, v, u);
        hist_at(j + 1, hv);  // (grid point N - 1 has leading slots too; its record is not used)
        y = state_axpy(y, dsl, k1);
      };
#pragma unroll 2
      for (int t = 0; t < R.sbase; ++t) fd_point(s_l + t, Pc.ds);
      if (lmax > R.sbase) fd_point(s_l + (len_l > R.sbase ? R.sbase : R.sbase - 1), len_l > R.sbase ? Pc.ds : T(0));
    } else {
      // one trip of the merged sweep.  FULL: every active lane is inside its interval - true for the trips
      // MSO_LAG .. sbase - 1
      const bool lean = A.ring && tB + 4 <= T_steps;  // (the last three states of a call stay complete)
      T* const tnew = tile(tB + 1);        // leading slots of the state this sweep produces (over those of state tB - 2)
      T la[12], lb[12];                    // GT: leading slots in flight for the trip after the one being worked on
      if constexpr (GT) {
        lead_load(lead_n, point_of(1), la);
        lead_load(lead_o, point_of(1), lb);
      }
      auto trip = [&](int k, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int kk = k - lag;
#if KR_MSWO_LOOP == 2
        // (a probe: one trip body for all trips - the grid point clamped into the lane's interval by one v_med3_i32, the
        //  predicate one unsigned compare - so that the compiler can pair ALL trips)
        const bool live = act && (unsigned)kk < (unsigned)len_l;
        const int j = max(s_l, min(s_l + kk, s_l + len_l - 1));
#else
        const bool live = FULL ? act : (act && kk >= 0 && kk < len_l);
        const int j = FULL ? s_l + kk : point_of(k);
#endif
        RodState<T> k1;
        V3<T> v, u;
        ode_eval<T, DIAG>(Pc, y, hist_lean<T, DIAG>(Pc, hv), fc, k1, v, u);
        if constexpr (GT) {
          // formed in front of the verifying lanes' block (behind it the wait for the loads would also wait for the block's
          // stores) and pinned there; not earlier either (the scheduler would pull the wait into the arithmetic above)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < 12; ++c) hv[c] = A.hc1 * la[c] + A.hc2 * lb[c];
          asm volatile("" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]), "+v"(hv[4]), "+v"(hv[5]));
          asm volatile("" : "+v"(hv[6]), "+v"(hv[7]), "+v"(hv[8]), "+v"(hv[9]), "+v"(hv[10]), "+v"(hv[11]));
        }
        if (isB && live) {
          // the accepted-to-be state of step tB at grid point j: its leading slots into the tile (interior records of
          // interior steps on a ring go nowhere else: three tiles roll a step back without the states in HBM)
          T rec[KR_SLOTS];
          record_from(y, v, u, rec);
          T lead[12];
#pragma unroll
          for (int c = 0; c < 12; ++c) lead[c] = rec[c];
          if constexpr (GT) {  // the record in HBM is the tile: its leading slots always, the rest where it must be complete
            if (lean) store_vec<T, 12>(out_rod + (size_t)j * KR_SLOTS, lead);
            else store_record(out_rod + (size_t)j * KR_SLOTS, rec);
          } else {
            // (no take-over kernel reads interval-start records here: lean steps store nothing at all in a trip)
            if (!lean) store_record(out_rod + (size_t)j * KR_SLOTS, rec);
            lds_store_vec<T, 12>(tnew + (size_t)j * 12, lead);
          }
        }
        if constexpr (GT) {
          // leading slots for the trip AFTER NEXT (the verifying lanes run two grid points ahead: a forward-difference lane's
          // were stored just above), from L2 under a whole trip of arithmetic; the next trip's record was formed above
          const int jn2 = FULL ? j + 2 : point_of(k + 2);
          lead_load(lead_n, jn2 < N ? jn2 : N - 1, la);
          lead_load(lead_o, jn2 < N ? jn2 : N - 1, lb);
        } else {
#if KR_MSWO_LOOP == 2
          hist_at(max(s_l, min(s_l + kk + 1, s_l + len_l - 1)), hv);
#else
          hist_at(FULL ? j + 1 : point_of(k + 1), hv);
#endif
        }
        const T dsl = live ? Pc.ds : T(0);  // (a lane outside its range evaluates finite data and adds nothing)
        y = state_axpy(y, dsl, k1);
      };
#if KR_MSWO_LOOP == 1 || KR_MSWO_LOOP == 2
      // (a probe: one general loop, trips paired by the compiler)
#pragma unroll 2
      for (int k = 0; k < trips; ++k) trip(k, std::false_type{});
#else
      int k = 0;
      for (; k < LAGV && k < trips; ++k) trip(k, std::false_type{});
#pragma unroll 2
      for (; k < R.sbase; ++k) trip(k, std::true_type{});
      for (; k < trips; ++k) trip(k, std::false_type{});
#endif
    }

#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(t_sw, tq);
    o_sweeps += 1;
    if (merged) o_merged += 1;
#endif
    // =============================================================================================================
    // verdict on the step under verification
    // =============================================================================================================
    if (merged) {
      if (isB) {
        T er[19];
        state_to_rows(y, er);
#pragma unroll
        for (int q = 0; q < 19; ++q) Es[lane * 19 + q] = er[q];
        if (gB == P - 1) {  // the last grid point: y from the sweep, z untouched
          T rec[KR_SLOTS];
          record_from(y, S.vlast, S.ulast, rec);
          T lead[12];
#pragma unroll
          for (int c = 0; c < 12; ++c) lead[c] = rec[c];
          const bool lean_v = A.ring && tB + 4 <= T_steps;  // (`lean` of the sweep)
          if (!lean_v) store_record(out_rod + (size_t)(N - 1) * KR_SLOTS, rec);
          else if constexpr (GT) store_vec<T, 12>(out_rod + (size_t)(N - 1) * KR_SLOTS, lead);
          if constexpr (!GT) lds_store_vec<T, 12>(tile(tB + 1) + (size_t)(N - 1) * 12, lead);
          if (A.tip) {
            T* tp = A.tip + (rod * T_steps + tB) * 3;
            tp[0] = y.p.x; tp[1] = y.p.y; tp[2] = y.p.z;
          }
        }
      }
      wave_sync_lds();
      const float rn = msw_max<W>(mswo_residual_B<T, W>(Es, XsB, L.cold, R, nact, lane), redf, wave, lane);
      const float est = ampB * rn;
      // residual test (kr_ms_impl.hpp: audited factor 256 on the measured update / residual ratio)
      bool accepted = A.residual_test != 0 && ampB > 0.f && T(256) * (T)est <= tol;
      float dnv = est;
      MswLds<T, W> LB = L;   // the solver's view with the unknowns of step tB in place of step tA's
      LB.Xs = XsB;
      MswUpd<T> U2;
      bool have_chord = false;
      if (!accepted && itB > 0) {
        // chord update through the factors of step tB's last condensation (its forward-difference columns are still in Es:
        // the condensation of step tA comes after this verdict): a second solve with the verifying lanes' end states as
        // base end states and the unknowns of step tB - the measured Newton-type update of the stored state
        RodState<T> yB = y;
        if (isA && col == 0) {
          T eb[19];
#pragma unroll
          for (int q = 0; q < 19; ++q) eb[q] = Es[(nact + R.ivl) * 19 + q];
          yB = rows_to_state(eb);
        }
#ifdef KR_MS_STAMPS
        unsigned long long tb_ = tq;
        msw_condense<T, W, true>(LB, R, lane, yB, T(1), U2, stamps, tb_);
        o_chord += 1;
#else
        msw_condense<T, W, true>(LB, R, lane, yB, T(1), U2);
#endif
        dnv = U2.dnf;
        have_chord = true;
        accepted = dnv <= 3.0e38f && (T)dnv <= T(0.5) * tol;
      }
#ifdef KR_MS_STAMPS
      if (accepted) o_accept += 1; else o_reject += 1;
#endif
      if (accepted) {
        if (!belowB && dnB > T(0)) {  // contraction constant where this step first got below the tolerance
          const T floor_dn = T(64) * (sizeof(T) == 8 ? T(2.2e-16) : T(1.2e-7));
          const T kq = fmax((T)dnv, floor_dn) * fast_rcp(dnB * dnB);
          kappa = fmin(fmax(kq, T(1e-4)), T(1));
        }
        if (wave == 0 && lane == 0 && A.status) A.status[rod * T_steps + tB] = KR_ST_CONVERGED;
        if (wave == 0 && lane < 6) Gguess = XsB[0 * 19 + 7 + lane];
        pred_skip = false;
        reverify = false;
        merged = false;
        if (!runA) break;  // that was the last step
      } else if (have_chord && dnv <= 3.0e38f && !reverify && itB + 2 <= maxit) {
        // First rejection of this step, and the chord update is a proper Newton-type correction (its factors are one
        // iteration old): take it and verify AGAIN in a merged sweep (kr_mso_impl.hpp does the same) - the verifying lanes
        // re-integrate step tB from the corrected unknowns over the state they have just left, the forward-difference lanes
        // repeat their sweep of step tA from the same start (their history was off by this correction).  A rejection then
        // costs one merged sweep instead of a roll-back to plain sweeps plus the restart of step tA.
        msw_apply<T, W>(LB, R, lane, U2);
        reverify = true;
        itB += 1;
        dnB = (T)dnv;
        msw_lds_barrier();
#ifdef KR_MS_STAMPS
        o_retry += 1;
#endif
        continue;
      } else {
        reverify = false;
        // Not accepted from its residual: the state in HBM was streamed from XsB, so a forward-difference sweep at XsB
        // measures the Newton update that belongs to it.  Drop the work done for step tA, put the history of step tB
        // back and carry on there.
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
        stored_at_xs = true;
        merged = false;
        msw_lds_barrier();
        if (it >= maxit) { plain_step(); }
        continue;
      }
    }

    // =============================================================================================================
    // Newton update of step tA from the forward-difference sweep (kr_msw_impl.hpp: msw_condense)
    // =============================================================================================================
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(t_verd, tq);
#endif
    ++it;
    MswUpd<T> U;
#ifdef KR_MS_STAMPS
    unsigned long long ta_ = tq;
    msw_condense<T, W>(L, R, lane, y, hstep, U, stamps, ta_);
#else
    msw_condense<T, W>(L, R, lane, y, hstep, U);
#endif
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(t_cond, tq);
#endif
    const float dnf = U.dnf;
    const bool finite = dnf <= 3.0e38f;
    const T dn = (T)dnf;
    if (finite && U.res_local > 0.f) amp = dnf / U.res_local;
    if (finite && !below && dn <= tol) {
      below = true;
      if (dn_prev > T(0)) {
        const T floor_dn = T(64) * (sizeof(T) == 8 ? T(2.2e-16) : T(1.2e-7));
        const T kq = fmax(dn, floor_dn) * fast_rcp(dn_prev * dn_prev);
        kappa = fmin(fmax(kq, T(1e-4)), T(1));
      }
    }
    if (finite && stored_at_xs && dn <= tol) {
      // the state a (rejected) verifying sweep streamed out belongs to these unknowns, and their Newton update is below
      // the tolerance: that state is the accepted one (the stopping rule of kr_msw_impl.hpp on a storing sweep)
      msw_lds_barrier();
#ifdef KR_MS_STAMPS
      o_stored += 1;
#endif
      if (wave == 0 && lane == 0 && A.status) A.status[rod * T_steps + tA] = KR_ST_CONVERGED;
      if (wave == 0 && lane < 6) Gguess = Xs[0 * 19 + 7 + lane];
      pred_skip = false;  // (the predictor consumed this step when it was first handed over)
      tA += 1;
      fcB = fcA;
      if (tA < T_steps) {
        fcA = fcN;
        fcN = load_fc(tA + 1 < T_steps ? tA + 1 : tA);
        order = Q.next_order;
        start_guess(order);
      }
      it = 0; retried = false; below = false; dn_prev = T(-1); amp = -1.f;
      stored_at_xs = false;
      continue;
    }
    stored_at_xs = false;
    bool next_final = false;
    if (finite) {
      msw_apply<T, W>(L, R, lane, U);
      // will the sweep at the updated unknowns be the accepted one?  (quadratic contraction, kr_ms_impl.hpp)
      next_final = predict_final<T>(dn, dn_prev, tol, tolA) || (kappa > T(0) && T(4) * kappa * dn * dn <= tol);
      dn_prev = dn;
    }
    msw_lds_barrier();
    if (!finite || (it >= maxit && !(next_final && dn <= T(1e-2)))) {
      // no root from this start: once more from the reference's warm start (knode.py:89), then the full ladder
      if (order > 0 && !retried && !pred_skip) {
        retried = true;
        order = 0;
        start_guess(0);
        it = 0; dn_prev = T(-1); amp = -1.f; below = false;
        continue;
      }
      // (plain sweeps leave the history of step tA alone: nothing to rebuild)
      plain_step();
      continue;
    }
    if (next_final && dn <= T(1e-2)) {
      // hand step tA to the verifying lanes and move the forward-difference lanes on to step tA + 1
      for (int e = threadIdx.x; e < P * 19; e += WAVE * W) XsB[e] = Xs[e];
      msw_lds_barrier();
      dnB = dn; ampB = amp; belowB = below; itB = it; orderB = order;
      if (!pred_skip) mswp_update<T, W>(Q, order, KR_ST_CONVERGED, A.predictor, lane, wave, ne, XlB, L.red);
      tA += 1;
      fcB = fcA;
      if (tA < T_steps) {
        fcA = fcN;
        fcN = load_fc(tA + 1 < T_steps ? tA + 1 : tA);
        order = Q.next_order;
        mswp_guess<T>(Q, order, lane, ne, wave == 0, L.cold, Xl);
        wave_sync_lds();
        if (wave == 0 && order <= 0 && lane < 6) Xs[0 * 19 + 7 + lane] = XsB[0 * 19 + 7 + lane];  // warm start: the G just found
        msw_lds_barrier();
      }
      it = 0; retried = false; below = false; dn_prev = T(-1);
      merged = true;
    }
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(t_hand, tq);
#endif
  }
#ifdef KR_MS_STAMPS
  if (A.dbg && wave == 0 && lane == 0) {
    unsigned long long te;
    KR_STAMP(te);
    unsigned long long* d = A.dbg + rod * 24;
    d[0] = te - t_begin; d[1] = t_sw; d[2] = t_cond; d[3] = t_verd; d[4] = o_sweeps; d[5] = o_merged; d[6] = o_accept; d[7] = o_reject;
    d[8] = o_plain; d[9] = o_stored; d[10] = t_hand; d[11] = o_chord;
  }
#endif

  if (wave == 0 && lane < 6) A.G[rod * 6 + lane] = Gguess;
  if (img) {
    Q.kappa = kappa;
    ms_pred_save<T>(Q, img, lane);
  }
}

// kr_simulate_batch with several wavefronts per rod and overlapped steps: 0 launched, 1 does not apply
template <typename T, int W, bool GT>
static int launch_mswo_inst(const RodConst<T>& P, const SimArgs<T>& a, size_t bytes, hipStream_t s) {
  auto kern = mswo_sim_kernel<T, W, GT>;
  if (int rc_lds_ = dyn_lds(reinterpret_cast<const void*>(kern), bytes)) return rc_lds_;
  hipLaunchKernelGGL(kern, dim3((unsigned)a.B), dim3(WAVE * W), bytes, s, P, a);
  KR_HIP(hipGetLastError());
  return KR_OK;
}
template <typename T>
int launch_mswo_sim(kr_handle* h, int W, const SimArgs<T>& a, hipStream_t s) {
  const RodConst<T>& P = consts<T>(h);
  if (!P.diag || (W != 2 && W != 4)) return 1;
  if (P.N - 1 < 2 * (4 + 3 * (W - 1))) return 1;
  // (every rod resident at once: a second round of workgroups would wait for the first to finish all steps)
  auto fits = [&](size_t bytes) { return bytes <= (size_t)h->lds_limit && a.B <= 256 * (int64_t)((size_t)h->lds_limit / bytes); };
  const size_t b_lds = sizeof(T) * (W == 2 ? mswo_lds_elems<T, 2, false>(P.N) : mswo_lds_elems<T, 4, false>(P.N));
  const size_t b_gt = sizeof(T) * (W == 2 ? mswo_lds_elems<T, 2, true>(P.N) : mswo_lds_elems<T, 4, true>(P.N));
  static const int force_gt = std::getenv("KR_MSWO_GT") ? std::atoi(std::getenv("KR_MSWO_GT")) : -1;  // (tests: 1 = tiles in HBM, 0 = never)
  const bool use_gt = force_gt == 1 || (force_gt != 0 && !fits(b_lds));
  if (use_gt ? !fits(b_gt) : !fits(b_lds)) return 1;
  if (use_gt && a.prev_init) {
    // The GT form reads the state before states[0] while step 0 is being verified, i.e. while the slot of state 1 is written:
    // a caller's prev_init inside that slot (knode_rod.h allows it to point into the ring) takes the plain form, which
    // consumes it before its first store.
    const T* s1 = a.states + a.slot_elems;
    if (a.prev_init >= s1 && a.prev_init < s1 + a.slot_elems) return 1;
  }
  h->last_waves_per_rod = W;
  if (use_gt) return W == 2 ? launch_mswo_inst<T, 2, true>(P, a, b_gt, s) : launch_mswo_inst<T, 4, true>(P, a, b_gt, s);
  return W == 2 ? launch_mswo_inst<T, 2, false>(P, a, b_lds, s) : launch_mswo_inst<T, 4, false>(P, a, b_lds, s);
}

}  // namespace kr
