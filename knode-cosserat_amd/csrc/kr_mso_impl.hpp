// kr_mso_impl.hpp - persistent multiple shooting with OVERLAPPED time steps (round 3).
//
// kr_ms_impl.hpp spends, in its steady state, two sweeps of (N-1)/4 dependent grid points on every time step: the
// forward-difference sweep at the predicted start x0(t) that yields the Newton correction, and a second sweep at the
// corrected unknowns x1(t) that measures the (by then negligible) residual and streams the accepted state out.  In the
// second sweep only the four unperturbed lanes do anything useful; the 54 forward-difference lanes recompute a
// Jacobian nobody uses once the residual test has accepted the sweep - which is what happens on 97 % of the steps.
//
// Here the verifying sweep of step t and the Jacobian sweep of step t + 1 are ONE sweep of the same wavefront:
//     lanes  0..57   forward-difference sweep of step t + 1 from its predicted start x0(t+1)   (as before)
//     lanes 58..61   re-integrate step t from x1(t), one lane per sub-interval, and stream the state out
// The start values of step t + 1 only need the UNKNOWNS of step t (known after its Newton update), and its BDF2
// history record at grid point j only needs the state of step t at j - so the verifying lanes run one grid point
// ahead and leave the twelve leading slots (q w v u) of every grid point they produce in an LDS tile; the
// forward-difference lanes pick them up at the end of the same trip and form the history record of their next grid point
// themselves (round 5: two raw tiles, time level t in tile t & 1 - a precombined record cost the four verifying lanes a
// divergent block of 30 fp64 instructions, 15 LDS stores and 6 LDS loads per trip, and a store from a lone wavefront is
// the most expensive instruction there is: LABBOOK, "what a store costs a lone wavefront").  Per time step that leaves
// one sweep and one condensation; the discrete equations, the Newton iteration, the stopping rule and the stored states
// are those of kr_ms_impl.hpp (cosserat_ode.py:188-213 inside knode.py:70-100).
//
// Acceptance of step t is still a measured quantity of a sweep at the stored unknowns: the residual test, else the
// chord update through the factors of step t's last condensation (which are only overwritten afterwards).  If
// neither accepts, the Jacobian work for step t + 1 is thrown away, the chord update is applied, the tiles are
// rebuilt from the two previous states in HBM (an accepted state's leading slots go there in one pass of the wavefront,
// the rest of a record that must be complete during its trip) and step t continues with plain forward-difference sweeps.
// Everything beyond that - a step that does not converge from the predicted or from the warm start - is left to
// kr_ms_impl.hpp's kernel, launched behind this one: a rod that gives up writes the step it stopped at to
// SimArgs::resume and the second kernel resumes there with the full ladder (warm start, damped single shooting).
//
// Serves: Euler sweeps, diagonal material matrices, MLP off, any N whose history fits the LDS (4 rods per CU).
#pragma once
#include "kr_ms_impl.hpp"

namespace kr {

constexpr int MSO_B0 = 7 + 17 * (MS_P - 1);  // 58: first of the four lanes that re-integrate the step under verification
#ifndef KR_MSO_LAG
#define KR_MSO_LAG 1
#endif
// grid points the verifying lanes run ahead of the lanes that consume their records.  2 (-DKR_MSO_LAG=2: a compile-time probe)
// lets the next trip's tile reads go out under this trip's arithmetic - measured 40.8 M against 42.1 M rod-steps/s (one more
// trip per sweep, and the pinned hand-over keeps the compiler from pairing trips)
constexpr int MSO_LAG = KR_MSO_LAG;
static_assert(MSO_B0 + MS_P <= WAVE, "the verifying lanes must fit beside the forward-difference lanes");

template <typename T, int HS>
__host__ __device__ inline size_t mso_lds_elems(int N) {
  return ms_lds_elems<T, HS>(N, true, false) + 80;  // + XsB: the unknowns of the step under verification
}

// Hand-offs between lanes of ONE wavefront through LDS need no hardware wait: the LDS unit serves a wavefront's
// accesses in order, and the compiler tracks the counters of the loads it consumes.  What is needed is that the
// compiler does not move LDS accesses across the hand-off - a wavefront-scope fence.  (wave_sync()'s workgroup-scope
// fence also waits for every outstanding global store, i.e. for the records a sweep has just streamed out.)
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// 16-byte vector loads / stores through pointers the CALLER knows to be LDS.  The tiles of leading slots are picked by the
// parity of a time level, and behind such a select the compiler no longer proves the address space: it fell back to
// flat_load / flat_store (24 + 12 per pair of trips) until the accesses said so themselves.
#define KR_LDS __attribute__((address_space(3)))
template <typename T, int NV>
__device__ __forceinline__ void lds_load_vec(const T* src, T (&v)[NV]) {
  using V = typename Vec16<T>::type;
  constexpr int n = Vec16<T>::n;
  static_assert(NV % n == 0, "vector load needs a multiple of 16 bytes");
  const KR_LDS V* s = (const KR_LDS V*)src;
#pragma unroll
  for (int c = 0; c < NV / n; ++c) {
    const V x = s[c];
#pragma unroll
    for (int e = 0; e < n; ++e) v[c * n + e] = x[e];
  }
}
template <typename T, int NV>
__device__ __forceinline__ void lds_store_vec(T* dst, const T (&v)[NV]) {
  using V = typename Vec16<T>::type;
  constexpr int n = Vec16<T>::n;
  static_assert(NV % n == 0, "vector store needs a multiple of 16 bytes");
  KR_LDS V* d = (KR_LDS V*)dst;
#pragma unroll
  for (int c = 0; c < NV / n; ++c) {
    V x;
#pragma unroll
    for (int e = 0; e < n; ++e) x[e] = v[c * n + e];
    d[c] = x;
  }
}

#ifdef KR_MS_STAMPS
struct MsoStats { unsigned long long total = 0, sweeps = 0, merged = 0, quick = 0, chord = 0, rejects = 0, retries = 0, rebuilds = 0, t_sweep = 0, t_alg = 0, t_pred = 0, t_verdict = 0, t_cond = 0, t_fin = 0, t_upd = 0, t_v1 = 0, t_v2 = 0, t_c1 = 0, t_c2 = 0, t_copy = 0; };
#endif

// OCC: workgroups per CU the register allocation leaves room for.  OCC = 2 (fp32, batches beyond one rod per SIMD)
// runs two rods on every SIMD, whose issue-bound sweeps and latency-bound algebra overlap (as kr_ms_impl.hpp's OCC).
template <typename T, bool DIAG, int HS, int OCC = 1>
__global__ __launch_bounds__(WAVE * MS_WPB, OCC) void mso_sim_kernel(const RodConst<T> Pc, const SimArgs<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = Pc.N;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wv = threadIdx.x / WAVE;
  const int64_t rod = (int64_t)blockIdx.x * MS_WPB + wv;
  if (rod >= A.B) return;  // whole wavefront; there is no workgroup barrier in this kernel
  const size_t rod_elems = (size_t)N * KR_SLOTS;
  const int64_t T_steps = A.T_steps;
  const MsLds<T> L = ms_carve<T, HS>(reinterpret_cast<T*>(smem_raw) + (size_t)wv * mso_lds_elems<T, HS>(N), N, true, false);
  T* const Xs = L.Xs;
  T* const Es = L.Es;
  T* const XB = L.XB;
  T* const Tm = L.Tm;
  T* const XsB = L.c12 + (size_t)N * 12;  // [P][19] unknowns of the step the verifying lanes re-integrate
  T* const EsB = Es + MSO_B0 * MS_YP;     // their end states: the Es slots of lanes 58..61
  const MsRole R = ms_role(lane, N);
  const int iv = R.iv, col = R.col;
  const bool isA = lane < MSO_B0;
  const bool isB = lane >= MSO_B0 && lane < MSO_B0 + MS_P;
  const int ib = isB ? lane - MSO_B0 : 0;
  const int s_l = isB ? ms_interval_start(ib, R.sbase, R.srem) : R.s_i;
  const int len_l = isB ? R.sbase + (ib < R.srem ? 1 : 0) : R.len_i;
  ms_cold_fill<T>(Pc, L.cold, lane);
  wave_sync();

  auto state_ptr = [&](int64_t k) -> T* { return A.states + (A.ring ? k % 3 : k) * A.slot_elems + rod * rod_elems; };
  // The BDF2 history (knode.py:74-75) is kept RAW: two tiles of leading slots (q w v u of every grid point), tile (t & 1)
  // for the state at time level t.  A lane forms the record it needs - hc1 * newest + hc2 * older, then av / au - itself,
  // right after the read: 30 instructions that every lane of the wavefront shares, where a precombined record cost the
  // four verifying lanes the same 30 instructions in a divergent block of their own plus 9 more 16-byte LDS stores and
  // 6 loads per trip (a wavefront-instruction costs the same with 4 lanes active as with 64, and an LDS store ~13 cycles).
  T* const lead0 = L.c12;                    // [N][12] states at even time levels
  const ptrdiff_t lead_d = L.hist - L.c12;  // states at odd time levels: the first 12 N elements of the record area
  // (offset arithmetic, not a select between two pointers: the compiler turned such selects into a table in scratch)
  auto lead_of = [&](int64_t t) -> T* { return lead0 + (ptrdiff_t)(t & 1) * lead_d; };
  // both tiles of step t's history from the states in HBM
  auto rebuild = [&](int64_t t) {
    const T* cur = state_ptr(t);
    const T* prv = t > 0 ? state_ptr(t - 1) : (A.prev_init ? A.prev_init + rod * rod_elems : cur);
    T* const lc = lead_of(t);
    T* const lp = lead_of(t - 1);
    for (int j = lane; j < N; j += WAVE) {
      T cv[12], pv[12];
      load_hist_vec<T, 12>(cur + (size_t)j * KR_SLOTS, cv);
      load_hist_vec<T, 12>(prv + (size_t)j * KR_SLOTS, pv);
      lds_store_vec<T, 12>(lc + (size_t)j * 12, cv);
      lds_store_vec<T, 12>(lp + (size_t)j * 12, pv);
    }
    wave_sync();
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
  MsPred<T> Q;
  double* img = A.pred_io ? A.pred_io + (size_t)rod * MS_PRED_ROWS * WAVE : nullptr;
  if (img && A.pred_load) ms_pred_load<T>(Q, img, lane);
  else ms_pred_init<T>(Q, lane, R, s0, sp0, A.prev_init != nullptr, A.predictor);
  V3<T> vlast, ulast;  // z of the last grid point is never touched by a sweep (cosserat_ode.py:198-201)
  {
    const T* cl = s0 + (size_t)(N - 1) * KR_SLOTS;
    vlast = {cl[SL_V], cl[SL_V + 1], cl[SL_V + 2]};
    ulast = {cl[SL_U], cl[SL_U + 1], cl[SL_U + 2]};
  }
  const T tol = A.tol, tolA = A.tolA, fd_eps = A.fd_eps;
  const int maxit = A.maxit;
  T kappa = Q.kappa;
  T Gguess = lane < 6 ? A.G[rod * 6 + lane] : T(0);
  MsStamps stamps;
#ifdef KR_MS_STAMPS
  MsoStats st;
  unsigned long long t_begin, tq;
  KR_STAMP(t_begin);
  tq = t_begin;
#endif

  rebuild(0);

  // ---- state of the iteration -----------------------------------------------------------------------------------
  int64_t tA = 0;        // step the forward-difference lanes work on (unknowns: Xs)
  bool merged = false;   // the coming sweep also re-integrates step tA - 1 from XsB and streams it out
  int it = 0;            // forward-difference sweeps spent on step tA
  int order = Q.next_order;
  bool retried = false;  // step tA has been restarted from the reference's warm start
  T dn_prev = T(-1);     // update norm of the previous iteration of step tA
  bool have_fac = false, below = false;
  float amp = -1.f;      // update norm per unit of residual norm at the last condensation
  // of the step under verification (tA - 1)
  T dnB = T(-1);
  float ampB = -1.f;
  bool belowB = false;
  int itB = 0, orderB = 0;
  bool pred_skip = false;  // the predictor has already consumed step tA (a verifying sweep of it was rejected)
  bool reverify = false;   // the coming merged sweep verifies step tA - 1 for the second time (after a chord update)
  int64_t resume_at = T_steps;
  T Xreg[MS_P - 1][2];
#pragma unroll
  for (int g = 0; g < MS_P - 1; ++g) Xreg[g][0] = Xreg[g][1] = T(0);
  const int kp = lane & 3;
  const int r = 3 + (lane >> 2);
  const bool plane = lane < 3 * (MS_P - 1);
  const int pi = plane ? lane / 3 : 0;  // term i = 0 .. P-2
  const int prow = plane ? lane - 3 * pi : 0;
  const bool glane = lane >= WAVE - 6;  // six lanes own the base wrench in the update step
  // (tensions are requested one step before they are needed: a load from HBM takes longer than what lies between
  // the hand-over of a step and the sweep that follows)
  V3<T> fcA = load_fc(0), fcB = fcA;
  V3<T> fcN = load_fc(T_steps > 1 ? 1 : 0);
  // av = hk_a + hk_b v_h, au = hk_c u_h (diagonal material matrices): kept in registers so that forming a history
  // record inside a sweep does not go back to the parameter table
  static_assert(DIAG, "the in-sweep history record assumes diagonal material matrices");
  T hk_a[3], hk_b[3], hk_c[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    hk_a[c] = L.cold[CD_KSEI + 4 * c] * L.cold[CD_KSEV + c];
    hk_b[c] = -L.cold[CD_KSEI + 4 * c] * L.cold[CD_BSE + 4 * c];
    hk_c[c] = -L.cold[CD_KBTI + 4 * c] * L.cold[CD_BBT + 4 * c];
  }

  ms_pred_guess<T>(Q, order, lane, L.cold, Xs);
  wave_sync_lds();
  if (order <= 0 && lane < 6) Xs[0 * MS_YP + 7 + lane] = Gguess;  // caller's guess (knode.py:67,89)
  wave_sync_lds();

  // p rows, scaled update norm and the per-lane pieces of an update of the unknowns X from base end states Eb(g)
  // (A: Es slot of the interval's unperturbed lane; B: EsB) - the tail both the chord and the Newton update share.
  // Returns the norm; the update itself is left in updP / updG / updY with the current values xsP / xsG / xsY.
  struct Upd { T updP, updG, xsP, xsG, xsY[MS_P - 1]; };

  while (true) {
    const bool runA = tA < T_steps;  // (false only for the sweep that verifies the last step)
    const int64_t tB = tA - 1;
    // ---- start state of this lane -------------------------------------------------------------------------------
    T yr[19];
    {
      const T* src = isB ? XsB + ib * MS_YP : Xs + iv * MS_YP;
#pragma unroll
      for (int q = 0; q < 19; ++q) yr[q] = src[q];
    }
    const T hstep = (isA && col > 0) ? fd_eps * fmax(fabs(Xs[iv * MS_YP + (R.comp > 0 ? R.comp : 3)]), T(1)) : T(1);
#pragma unroll
    for (int q = 3; q < 19; ++q) yr[q] += (isA && q == R.comp) ? hstep : T(0);
    RodState<T> y = rows_to_state(yr);
    const V3<T> fc = isB ? fcB : fcA;
    // the verifying lanes run MSO_LAG grid points ahead of the forward-difference lanes that consume their records
    const int lag = (isA && merged) ? MSO_LAG : 0;
    const bool act = isB ? merged : (isA && runA);
    const int trips = R.lmax + ((merged && runA) ? MSO_LAG : 0);
    T* const out_rod = state_ptr(tB + 1);  // (used by the verifying lanes only)
    // grid point this lane evaluates in trip k (clamped to its interval: lanes outside their range keep evaluating
    // their first / last point and do not advance)
    auto point_of = [&](int k) -> int {
      const int kk = k - lag;
      return s_l + (kk < 0 ? 0 : (kk < len_l ? kk : len_l - 1));
    };
    // History record of this lane's step at grid point j from the two tiles.  The forward-difference lanes work on step
    // tA (newest state: tile tA & 1), the verifying lanes on step tB = tA - 1 (newest: the other tile; at the grid points
    // they have not reached yet tile tA & 1 still holds the state at level tB - 1, behind them the state they are
    // producing).  The tile addresses differ per lane; the arithmetic is the same for all.
    const T* const lead_n = lead_of(isB ? tB : tA);      // this lane's newest state
    const T* const lead_o = lead_of(isB ? tB + 1 : tB);  // ... and the one before it (same parity as two levels on)
    auto hist_form = [&](const T (&la)[12], const T (&lb)[12]) __attribute__((always_inline)) -> RodHist<T> {
      T raw[12];
#pragma unroll
      for (int c = 0; c < 12; ++c) raw[c] = A.hc1 * la[c] + A.hc2 * lb[c];
      RodHist<T> h;
      h.qh = {raw[0], raw[1], raw[2]};
      h.wh = {raw[3], raw[4], raw[5]};
      h.vh = {raw[6], raw[7], raw[8]};
      h.uh = {raw[9], raw[10], raw[11]};
      h.av = {hk_b[0] * raw[6] + hk_a[0], hk_b[1] * raw[7] + hk_a[1], hk_b[2] * raw[8] + hk_a[2]};  // (Kse + c0 Bse)^-1 (Kse v* - Bse v_h)
      h.au = {hk_c[0] * raw[9], hk_c[1] * raw[10], hk_c[2] * raw[11]};                                       // -(Kbt + c0 Bbt)^-1 Bbt u_h
      return h;
    };
    auto hist_at = [&](int j) __attribute__((always_inline)) -> RodHist<T> {
      T la[12], lb[12];
      lds_load_vec<T, 12>(lead_n + (size_t)j * 12, la);
      lds_load_vec<T, 12>(lead_o + (size_t)j * 12, lb);
      return hist_form(la, lb);
    };
    RodHist<T> hst = hist_at(point_of(0));
    if (!merged) {
      // plain forward-difference sweep (start-up, rough inputs, after a rejection): the branch-free body of
      // kr_ms_impl.hpp, two grid points per trip so that the scheduler overlaps neighbours; every interval has
      // sbase or sbase + 1 segments, so only the last grid point needs a predicate
      auto fd_point = [&](int j, T dsl) __attribute__((always_inline)) {
        RodState<T> k1;
        V3<T> v, u;
        ode_eval<T, DIAG>(Pc, y, hst, fc, k1, v, u);
        hst = hist_at(j + 1);  // (grid point N - 1 has leading slots too; its record is not used)
        y = state_axpy(y, dsl, k1);
      };
#pragma unroll 2
      for (int t = 0; t < R.sbase; ++t) fd_point(R.s_i + t, Pc.ds);
      if (R.srem) fd_point(R.s_i + (R.len_i > R.sbase ? R.sbase : R.sbase - 1), R.len_i > R.sbase ? Pc.ds : T(0));
    } else {
      // one trip of the merged sweep.  FULL: every active lane is inside its interval (no predicates, no clamped
      // indices) - true for the trips MSO_LAG .. sbase - 1, i.e. all but the first and last few.
      const bool lean = A.ring && tB + 4 <= T_steps;  // (the last three states of a call stay complete)
      T* const lead_w = lead_of(tB + 1);               // the tile the verifying lanes write: state tB + 1 over state tB - 1
      auto trip = [&](int k, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int kk = k - lag;
        const bool live = FULL ? act : (act && kk >= 0 && kk < len_l);
        const int j = FULL ? s_l + kk : point_of(k);
        // MSO_LAG >= 2: what the next trip reads was written a trip ago - requested now, under the arithmetic of this one
        T la[12], lb[12];
        if constexpr (MSO_LAG >= 2) {
          const int jn = FULL ? j + 1 : point_of(k + 1);
          lds_load_vec<T, 12>(lead_n + (size_t)jn * 12, la);
          lds_load_vec<T, 12>(lead_o + (size_t)jn * 12, lb);
        }
        RodState<T> k1;
        V3<T> v, u;
        ode_eval<T, DIAG>(Pc, y, hst, fc, k1, v, u);
        // (formed in front of the verifying lanes' block: behind it the wait for the reads would also wait for its stores)
        if constexpr (MSO_LAG >= 2) {
          hst = hist_form(la, lb);
          // (pinned: the compiler otherwise sinks the arithmetic - and with it the wait - behind the block)
          asm volatile("" : "+v"(hst.qh.x), "+v"(hst.qh.y), "+v"(hst.qh.z), "+v"(hst.wh.x), "+v"(hst.wh.y), "+v"(hst.wh.z));
          asm volatile("" : "+v"(hst.vh.x), "+v"(hst.vh.y), "+v"(hst.vh.z), "+v"(hst.uh.x), "+v"(hst.uh.y), "+v"(hst.uh.z));
          asm volatile("" : "+v"(hst.av.x), "+v"(hst.av.y), "+v"(hst.av.z), "+v"(hst.au.x), "+v"(hst.au.y), "+v"(hst.au.z));
        }
        if (isB && live) {
          // the accepted-to-be state of step tB at grid point j: to HBM; its leading slots replace those of the state
          // two levels back in LDS (this lane has already read them: its record of grid point j was formed a trip ago)
          T rec[KR_SLOTS];
          record_from(y, v, u, rec);
          T lead[12];
#pragma unroll
          for (int c = 0; c < 12; ++c) lead[c] = rec[c];
          // On a 3-slot ring nobody reads the states of the call's interior steps - except this kernel when it rolls a
          // step back and the kernel launched behind it when it takes a rod over, and both need only the twelve leading
          // slots of every record plus the full records at the interval starts and at the last grid point (predictor,
          // z of the last point).  The leading slots are what this sweep leaves in its tile: once the step is accepted
          // the whole tile goes out in one pass of the wavefront (below, 2 x 6 stores from all lanes instead of 6 stores
          // from four lanes in every trip); interior records of interior steps store nothing here, all others only
          // their remaining sixteen slots.
          if constexpr (FULL) {
            if (!lean) {  // the rest of the record (p h n m); its leading slots follow with the tile
              T rest[KR_SLOTS - 12];
#pragma unroll
              for (int c = 0; c < KR_SLOTS - 12; ++c) rest[c] = rec[12 + c];
              store_vec<T, KR_SLOTS - 12>(out_rod + (size_t)j * KR_SLOTS + 12, rest);
            }
          } else {
            store_record(out_rod + (size_t)j * KR_SLOTS, rec);
          }
          lds_store_vec<T, 12>(lead_w + (size_t)j * 12, lead);
        }
        // history record of the next trip.  A forward-difference lane reads the leading slots a verifying lane wrote
        // MSO_LAG - 1 trips ago (for MSO_LAG = 1: above, in this trip); a verifying lane reads slots it has not
        // replaced yet.  (FULL: j + 1 <= N - 1 is a valid grid point even where it lies past the lane's interval.)
        if constexpr (MSO_LAG < 2) hst = hist_at(FULL ? j + 1 : point_of(k + 1));
        const T dsl = live ? Pc.ds : T(0);  // (a lane outside its range evaluates finite data and adds nothing)
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

    T er[19];
    state_to_rows(y, er);
    T d[6];
    T updY[MS_P - 1];
    T* dYb = XB;
    float dnf = 0.f;
    bool finite = true;
    Upd U;

    // Tail of an update: the p rows dY_g[p] = sum_{i<g} (c_i[p] + A_i[p,:] dY_i[3:]) (one term per lane), then the
    // scaled maximum norm over every unknown.  eb0: first Es slot (in lanes) of the base end states, ebs: slots
    // between consecutive intervals' base states; X: the unknowns the update belongs to.
    auto finish = [&](const T* X, auto base_slot) -> float {
      T* sp = Tm;  // [P-1][3] partial sums (the 6x6 solve has consumed Tm)
      if (kp == 0) {
#pragma unroll
        for (int g = 1; g < MS_P - 1; ++g) dYb[g * MS_YP + r] = updY[g - 1];
      }
      wave_sync_lds();
      if (plane) {
        const int l0 = pi == 0 ? 0 : 7 + 17 * (pi - 1);
        T s = Es[base_slot(pi) * MS_YP + prow] - X[(pi + 1) * MS_YP + prow];
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
      wave_sync_lds();
      float nf = 0.f;
      U.updP = T(0); U.updG = T(0); U.xsP = T(0); U.xsG = T(0);
      if (plane) {  // this lane owns Y_{pi+1}[prow]
        U.xsP = X[(pi + 1) * MS_YP + prow];
#pragma unroll
        for (int i = 0; i < MS_P - 1; ++i) {
          const T t = sp[i * 3 + prow];
          U.updP += i <= pi ? t : T(0);
        }
        nf = update_ratio(U.updP, U.xsP);
      }
      if (glane) {
        const int k = lane - (WAVE - 6);
        U.xsG = X[0 * MS_YP + 7 + k];
        U.updG = k == 0 ? d[0] : k == 1 ? d[1] : k == 2 ? d[2] : k == 3 ? d[3] : k == 4 ? d[4] : d[5];
        nf = fmaxf(nf, update_ratio(U.updG, U.xsG));
      }
#pragma unroll
      for (int g = 1; g < MS_P; ++g) {
        U.xsY[g - 1] = T(0);
        if (((g - 1) & 3) == kp) {  // one lane of the quad owns Y_g[r]
          U.xsY[g - 1] = X[g * MS_YP + r];
          nf = fmaxf(nf, update_ratio(updY[g - 1], U.xsY[g - 1]));
        }
      }
      return wave_max_nonneg(nf);  // +inf if any update is not finite
    };
    auto apply = [&](T* X) {
      if (plane) X[(pi + 1) * MS_YP + prow] = U.xsP + U.updP;
      if (glane) X[0 * MS_YP + 7 + (lane - (WAVE - 6))] = U.xsG + U.updG;
#pragma unroll
      for (int g = 1; g < MS_P; ++g)
        if (((g - 1) & 3) == kp) X[g * MS_YP + r] = U.xsY[g - 1] + updY[g - 1];
    };

    // =============================================================================================================
    // verdict on the step under verification
    // =============================================================================================================
    if (merged) {
      if (isB) {
#pragma unroll
        for (int q = 0; q < 19; ++q) EsB[ib * MS_YP + q] = er[q];
        if (ib == MS_P - 1) {  // the last grid point: y from the sweep, z untouched
          T rec[KR_SLOTS];
          record_from(y, vlast, ulast, rec);
          store_record(out_rod + (size_t)(N - 1) * KR_SLOTS, rec);
          T lead[12];
#pragma unroll
          for (int c = 0; c < 12; ++c) lead[c] = rec[c];
          lds_store_vec<T, 12>(lead_of(tB + 1) + (size_t)(N - 1) * 12, lead);
          if (A.tip) {
            T* tp = A.tip + (rod * T_steps + tB) * 3;
            tp[0] = y.p.x; tp[1] = y.p.y; tp[2] = y.p.z;
          }
        }
      }
      wave_sync_lds();
#ifdef KR_MS_STAMPS
      unsigned long long tv;
      { KR_STAMP(tv); st.t_v1 += tv - tq; }  // end states of the verifying lanes, record of the last grid point
#endif
      // residual of the sweep: interface jumps E_g - Y_{g+1} and the tip condition, one component per lane
      float rn = 0.f;
      if (lane < 3 * MS_YP) {
        const int g = lane / MS_YP, q = lane - MS_YP * g;
        const T x = XsB[(g + 1) * MS_YP + q];
        rn = update_ratio(EsB[g * MS_YP + q] - x, x);
      } else if (lane < 3 * MS_YP + 6) {
        const int k = lane - 3 * MS_YP;
        const T e = EsB[(MS_P - 1) * MS_YP + 7 + k];
        rn = update_ratio(L.cold[CD_FTIP + k] - e, e);
      }
      rn = wave_max_nonneg(rn);
#ifdef KR_MS_STAMPS
      KR_STAMP_ADD(st.t_v2, tv);  // residual norm
#endif
      const float est = ampB * rn;
      // residual test (kr_ms_impl.hpp: audited factor 256 on the measured update / residual ratio)
      bool accepted = A.residual_test != 0 && ampB > 0.f && T(256) * (T)est <= tol;
      float dnv = est;
#ifdef KR_MS_STAMPS
      if (accepted) st.quick += 1;
#endif
      if (!accepted) {
        // chord update through the factors of step tB's last condensation: forward-difference columns in Es,
        // X_g pairs in registers, T^-1 in LDS
        T* ach = XB;             // a_g, g = 1 .. P-1: [g][19]
        T* rt = XB + 4 * MS_YP;  // tip right-hand side [6]
        dYb = XB + MS_YP * 8;
        T areg[MS_P - 1];
        areg[0] = EsB[0 * MS_YP + r] - XsB[1 * MS_YP + r];  // a_1 = c_0
        if (kp == 0) ach[1 * MS_YP + r] = areg[0];
        wave_sync_lds();
#pragma unroll
        for (int g = 1; g < MS_P; ++g) {
          const int l0 = 7 + 17 * (g - 1);
          T av[4], xv[4];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            av[cc] = Es[(l0 + 1 + 4 * kp + cc) * MS_YP + r];
            xv[cc] = ach[g * MS_YP + 3 + 4 * kp + cc];
          }
          const T e0 = EsB[g * MS_YP + r];
          const T ynext = g < MS_P - 1 ? XsB[(g + 1) * MS_YP + r] : T(0);
          T part = fma(av[0], xv[0], av[1] * xv[1]) + fma(av[2], xv[2], av[3] * xv[3]);
          part += quad_xor<0xB1>(part);
          part += quad_xor<0x4E>(part);
          if (g < MS_P - 1) {
            areg[g] = e0 - ynext + part;
            if (kp == 0) ach[(g + 1) * MS_YP + r] = areg[g];
          } else if (kp == 0 && r >= 7 && r < 13) {
            rt[r - 7] = L.cold[CD_FTIP + (r - 7)] - e0 - part;
          }
          wave_sync_lds();
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
        dnv = finish(XsB, [](int g) { return MSO_B0 + g; });
        accepted = dnv <= 3.0e38f && (T)dnv <= T(0.5) * tol;
#ifdef KR_MS_STAMPS
        st.chord += 1;
#endif
      }
      if (accepted) {
        {  // leading slots of the accepted state: tile -> HBM, every grid point in one pass of the wavefront
          const T* const lt = lead_of(tB + 1);
          for (int j = lane; j < N; j += WAVE) {
            T lv[12];
            lds_load_vec<T, 12>(lt + (size_t)j * 12, lv);
            store_vec<T, 12>(out_rod + (size_t)j * KR_SLOTS, lv);
          }
        }
        if (!belowB && dnB > T(0)) {  // contraction constant where this step first got below the tolerance
          const T floor_dn = T(64) * (sizeof(T) == 8 ? T(2.2e-16) : T(1.2e-7));
          const T kq = fmax((T)dnv, floor_dn) * fast_rcp(dnB * dnB);
          kappa = fmin(fmax(kq, T(1e-4)), T(1));
        }
        if (lane == 0 && A.status) A.status[rod * T_steps + tB] = KR_ST_CONVERGED;
        if (lane < 6) Gguess = XsB[0 * MS_YP + 7 + lane];
        pred_skip = false;
        reverify = false;
        merged = false;
        if (!runA) break;  // that was the last step
      } else {
        // Rejected: the sweep stored a state that is not within the tolerance.  Take the chord update (when it is
        // finite), drop the work done for step tA, put the history of step tB back and carry on with plain
        // forward-difference sweeps at the updated unknowns.
#ifdef KR_MS_STAMPS
        st.rejects += 1;
#endif
        const bool ok = dnv <= 3.0e38f;
        if (ok) apply(XsB);
        wave_sync_lds();
        if (ok && !reverify && itB + 2 <= maxit) {
          // First rejection of this step, and the chord update is a proper Newton-type correction (its factors are one
          // iteration old): verify AGAIN in a merged sweep instead of falling back to two plain ones.  The history of
          // step tB comes back from HBM, the forward-difference lanes repeat their sweep of step tA from the same
          // start (what they computed above used a history that was off by this correction) - everything else is as
          // it was when the rejected sweep began, so a rejection costs one merged sweep instead of two plain sweeps
          // plus the restart of step tA.  In a short launch the slowest rod sets the time, and the slowest rod is the
          // one with rejections.
          reverify = true;
          itB += 1;
          dnB = (T)dnv;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");  // the records streamed out above are re-read below
          rebuild(tB);
#ifdef KR_MS_STAMPS
          st.rebuilds += 1;
          KR_STAMP_ADD(st.t_alg, tq);
#endif
          continue;
        }
        reverify = false;
        for (int e = lane; e < MS_NE; e += WAVE) Xs[e] = XsB[e];
        tA = tB;
        fcN = fcA;
        fcA = fcB;
        order = orderB;
        it = itB;
        dn_prev = ok ? (T)dnv : T(-1);
        have_fac = ok;
        amp = ampB;
        below = false;
        pred_skip = true;
        merged = false;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");  // the records streamed out above are re-read below
        rebuild(tA);
#ifdef KR_MS_STAMPS
        st.rebuilds += 1;
        KR_STAMP_ADD(st.t_alg, tq);
#endif
        if (it >= maxit) { resume_at = tA; break; }
        continue;
      }
    }

    // =============================================================================================================
    // Newton update of step tA from the forward-difference sweep (kr_ms_impl.hpp, "full" branch)
    // =============================================================================================================
#ifdef KR_MS_STAMPS
    { unsigned long long t_; KR_STAMP(t_); st.t_verdict += t_ - tq; }
    unsigned long long tq2;
    KR_STAMP(tq2);
#endif
    ++it;
    float res_full;
    {
      const int l0own = iv == 0 ? 0 : 7 + 17 * (iv - 1);
      if (isA && col == 0) {
#pragma unroll
        for (int q = 0; q < 19; ++q) Es[lane * MS_YP + q] = er[q];
      }
      wave_sync_lds();
      res_full = ms_residual_norm<T>(Es, Xs, L.cold, lane);
      if (isA && col > 0) {
        const T ih = fast_rcp(hstep);
        T e0[19];  // all loads first: the compiler cannot tell that they never alias the stores below
#pragma unroll
        for (int q = 0; q < 19; ++q) e0[q] = Es[l0own * MS_YP + q];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 19; ++q) Es[lane * MS_YP + q] = (er[q] - e0[q]) * ih;
      }
      wave_sync_lds();
    }
#ifdef KR_MS_STAMPS
    unsigned long long tc;
    { KR_STAMP(tc); st.t_c1 += tc - tq2; }  // end states, residual norm, forward-difference columns through Es
#endif
    dYb = XB;
    {
      const T c0 = Es[0 * MS_YP + r] - Xs[1 * MS_YP + r];
      const T da = Es[(2 * kp) * MS_YP + r];      // lanes 1..6 hold the columns of A_0 (lane 0: E_0, unused)
      const T db = Es[(2 * kp + 1) * MS_YP + r];  // (lane 7 is E_1: masked below)
      Xreg[0][0] = kp == 0 ? c0 : da;
      Xreg[0][1] = kp == 3 ? T(0) : db;
      store_pair(XB + r * 8 + 2 * kp, Xreg[0][0], Xreg[0][1]);
    }
    wave_sync_lds();
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
      wave_sync_lds();
    }
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(st.t_c2, tc);  // condensation chain
#endif
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
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(st.t_cond, tq2);
#endif
    dnf = finish(Xs, [](int g) { return g == 0 ? 0 : 7 + 17 * (g - 1); });
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(st.t_fin, tq2);
#endif
    finite = dnf <= 3.0e38f;
    const T dn = (T)dnf;
    if (res_full > 0.f && finite) amp = dnf / res_full;
    if (finite && !below && dn <= tol) {
      below = true;
      if (dn_prev > T(0)) {
        const T floor_dn = T(64) * (sizeof(T) == 8 ? T(2.2e-16) : T(1.2e-7));
        const T kq = fmax(dn, floor_dn) * fast_rcp(dn_prev * dn_prev);
        kappa = fmin(fmax(kq, T(1e-4)), T(1));
      }
    }
    bool next_final = false;
    if (finite) {
      apply(Xs);
      // will the sweep at the updated unknowns be the accepted one?  (quadratic contraction, kr_ms_impl.hpp)
      next_final = predict_final<T>(dn, dn_prev, tol, tolA) || (kappa > T(0) && T(4) * kappa * dn * dn <= tol);
      dn_prev = dn;
    }
    wave_sync_lds();
#ifdef KR_MS_STAMPS
    KR_STAMP_ADD(st.t_alg, tq);
#endif
    if (!finite || (it >= maxit && !(next_final && dn <= T(1e-2)))) {
      // no root from this start: once more from the reference's warm start (knode.py:89), then give the rod to the
      // kernel with the damped fallback
      if (order > 0 && !retried) {
        retried = true;
        order = 0;
        ms_pred_guess<T>(Q, 0, lane, L.cold, Xs);
        wave_sync_lds();
        if (lane < 6) Xs[0 * MS_YP + 7 + lane] = Gguess;
        wave_sync_lds();
        it = 0; dn_prev = T(-1); have_fac = false; amp = -1.f; below = false;
#ifdef KR_MS_STAMPS
        st.retries += 1;
#endif
        continue;
      }
      resume_at = tA;
      break;
    }
    if (next_final && dn <= T(1e-2)) {
      // hand step tA to the verifying lanes and move the forward-difference lanes on to step tA + 1
      for (int e = lane; e < MS_NE; e += WAVE) XsB[e] = Xs[e];
      wave_sync_lds();
#ifdef KR_MS_STAMPS
      { unsigned long long t_; KR_STAMP(t_); st.t_copy += t_ - tq; }
#endif
      dnB = dn; ampB = amp; belowB = below; itB = it; orderB = order;
      if (!pred_skip) ms_pred_update<T>(Q, order, KR_ST_CONVERGED, A.predictor, lane, XsB, stamps);
#ifdef KR_MS_STAMPS
      { unsigned long long t_; KR_STAMP(t_); st.t_upd += t_ - tq; }
#endif
      tA += 1;
      fcB = fcA;
      if (tA < T_steps) {
        fcA = fcN;
        fcN = load_fc(tA + 1 < T_steps ? tA + 1 : tA);
        order = Q.next_order;
        ms_pred_guess<T>(Q, order, lane, L.cold, Xs);
        wave_sync_lds();
        if (order <= 0 && lane < 6) Xs[0 * MS_YP + 7 + lane] = XsB[0 * MS_YP + 7 + lane];  // warm start: the G just found
        wave_sync_lds();
      }
      it = 0; retried = false; below = false; dn_prev = T(-1);
      merged = true;
#ifdef KR_MS_STAMPS
      KR_STAMP_ADD(st.t_pred, tq);
#endif
    }
  }

  if (lane < 6) A.G[rod * 6 + lane] = Gguess;
  if (lane == 0 && A.resume) A.resume[rod] = (int32_t)resume_at;
  // A rod handed over to the take-over kernel does not save: that kernel rebuilds the predictor from the stored states
  // when it resumes (t0 > 0) and, for a rod that gave up at step 0, must still find the image of the PREVIOUS call
  // (not one that already contains the rejected step); it saves its own image at the end.
  if (img && (resume_at >= T_steps || !A.resume)) {
    Q.kappa = kappa;
    ms_pred_save<T>(Q, img, lane);
  }
#ifdef KR_MS_STAMPS
  if (lane == 0 && A.dbg) {
    unsigned long long te;
    KR_STAMP(te);
    unsigned long long* dd = A.dbg + rod * 24;
    dd[0] = te - t_begin; dd[1] = st.t_sweep; dd[2] = st.t_alg; dd[3] = st.t_pred; dd[4] = st.sweeps;
    dd[5] = st.merged; dd[6] = st.quick; dd[7] = st.chord; dd[8] = st.rejects; dd[9] = st.retries; dd[10] = st.rebuilds;
    dd[11] = (unsigned long long)resume_at;
    dd[12] = st.t_verdict; dd[13] = st.t_cond; dd[14] = st.t_fin; dd[15] = st.t_upd;
    dd[16] = stamps.a1; dd[17] = stamps.a2; dd[18] = stamps.a3; dd[19] = st.t_v1; dd[20] = st.t_v2; dd[21] = st.t_c1; dd[22] = st.t_c2; dd[23] = st.t_copy;
  }
#endif
}

// Host-side one-time work of the first launch, done ahead of time (kr_simulate_prepare): resolves the kernel in the
// code object and sets its dynamic LDS limit.  Returns 1 when the kernel does not serve the handle's problem.
template <typename T>
int prepare_mso_sim(kr_handle* h, int64_t B) {
  const RodConst<T>& P = consts<T>(h);
  constexpr int HS = hs_phys<T>();
  if (!P.diag || P.N - 1 < 2 * MS_P) return 1;
  const size_t smem = sizeof(T) * mso_lds_elems<T, HS>(P.N) * MS_WPB;
  if (smem > (size_t)h->lds_limit) return 1;
  hipFuncAttributes fa;
  if (sizeof(T) == 4 && B > 1024 && 2 * smem <= (size_t)h->lds_limit) {
    auto k2 = mso_sim_kernel<T, true, HS, 2>;
    KR_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k2)));
    if (int rc_lds_ = dyn_lds(reinterpret_cast<const void*>(k2), smem)) return rc_lds_;
  } else {
    auto k1 = mso_sim_kernel<T, true, HS, 1>;
    KR_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k1)));
    if (int rc_lds_ = dyn_lds(reinterpret_cast<const void*>(k1), smem)) return rc_lds_;
  }
  return KR_OK;
}

template <typename T>
int launch_mso_sim(kr_handle* h, const SimArgs<T>& a, hipStream_t s) {
  const RodConst<T>& P = consts<T>(h);
  constexpr int HS = hs_phys<T>();
  if (!P.diag || P.N - 1 < 2 * MS_P) return 1;
  const size_t smem = sizeof(T) * mso_lds_elems<T, HS>(P.N) * MS_WPB;
  if (smem > (size_t)h->lds_limit) return 1;
  const dim3 grid((unsigned)((a.B + MS_WPB - 1) / MS_WPB)), block(WAVE * MS_WPB);
  if constexpr (sizeof(T) == 4) {
    // fp32, more rods than SIMDs, and two workgroups fit the LDS of a CU: two wavefronts per SIMD
    if (a.B > 1024 && 2 * smem <= (size_t)h->lds_limit) {
      auto kern2 = mso_sim_kernel<T, true, HS, 2>;
      if (int rc_lds_ = dyn_lds(reinterpret_cast<const void*>(kern2), smem)) return rc_lds_;
      hipLaunchKernelGGL(kern2, grid, block, smem, s, P, a);
      KR_HIP(hipGetLastError());
      return KR_OK;
    }
  }
  auto kern = mso_sim_kernel<T, true, HS, 1>;
  if (int rc_lds_ = dyn_lds(reinterpret_cast<const void*>(kern), smem)) return rc_lds_;
  hipLaunchKernelGGL(kern, grid, block, smem, s, P, a);
  KR_HIP(hipGetLastError());
  return KR_OK;
}

}  // namespace kr
