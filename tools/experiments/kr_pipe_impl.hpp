// kr_pipe_impl.hpp - the persistent multiple-shooting solver with consecutive time steps pipelined over two wavefronts.
//
// In the two-sweep regime of kr_ms_impl.hpp (smooth inputs, fitted predictor) a time step is: one sweep with the
// forward-difference columns, one Newton update - after which the unknowns are converged to ~1e-12 - and a second sweep
// that only re-integrates the rod from those unknowns to stream the state out, checked by the residual test.  With one
// wavefront per rod a batch of B <= 512 rods leaves at least half of the chip's 1024 SIMDs idle.  Here a rod owns two
// wavefronts with fixed roles:
//   SOLVER      (wavefront 0)  predictor, sweep with the forward-difference columns, condensation, Newton update -
//               ms_newton<PIPE> - then PUBLISHES the unknowns of step t and goes straight on to step t + 1;
//   INTEGRATOR  (wavefront 1)  integrates step t from the published unknowns (4 lanes, one per sub-interval), streams
//               the records to HBM, and writes - grid point by grid point - the BDF2 history records of step t + 1
//               into LDS, followed by a progress counter.  The solver's sweep of step t + 1 runs a grid point behind
//               it (pipe_wait).  At the end it applies the residual test and posts its verdict.
// A step then costs one sweep plus the solver's algebra instead of two sweeps plus algebra.  When the integrator
// rejects a step (the residual test fails: a few per cent of the steps) the solver throws away what it has done for
// step t + 1, takes one more full Newton iteration on step t - the history of step t is still intact: the
// integrator only overwrites it during step t + 1, which has not been published - and publishes again.
//
// LDS per rod: history records of two steps (ping-pong), leading slots (q w v u) of two states (ping-pong), the
// solver's condensation buffers, the published unknowns and a handful of counters: 62 KB at N = 100 in fp64, two rods
// per CU.  Both wavefronts spin on LDS counters with a bounded number of polls; running out of polls raises an abort
// flag that both check, and the step is reported as not converged.
// Euler sweeps, MLP off (what small batches of the BASELINE configurations run).
#pragma once
// (included by kr_ms_impl.hpp, after its definitions)

namespace kr {

constexpr int PIPE_PB = 1 << 12;  // progress counters: publication number * PIPE_PB + records written

template <typename T, int HS>
struct PipeLds {
  // (parity-indexed arrays are addressed as base + parity * size, not through an array of pointers: indexing a pointer
  // array at run time makes the pointers generic, and the compiler then emits FLAT loads / stores for LDS - which wait on
  // the vector-memory counter too, i.e. on the state records on their way to HBM: 2 k cycles per grid point)
  T* H0;         // [2][N][HS] history records of steps with even / odd index
  T* P0;         // [2][N][12] leading slots of the states with even / odd time level
  int hsz, psz;  // N * HS, N * 12
  MsLds<T> A;    // the solver's buffers (A.hist is set per step)
  T* Xpub;       // [MS_P][19] published unknowns
  T* EsB;        // [MS_P][19] end states of the integrator's sub-intervals
  T* ctr_raw;    // 16 counters (LdsCounter): [0] publication number, [1] published step, [2] status the solver wants reported (-1: decide by
                      // the residual test), [3] verdict = publication * 2 + ok, [4..7] progress per sub-interval,
                      // [8] abort, [9] amp (float bits)
};
template <typename T, int HS>
__host__ __device__ inline size_t pipe_lds_elems(int N) {
  const size_t alg = 2 * MS_YP * 8 + 48 + ((WAVE * MS_YP + 3) & ~3);
  size_t n = (size_t)2 * N * HS + (size_t)2 * N * 12 + ((MS_P * MS_YP + 3) & ~3) + ((CD_SIZE + 3) & ~3) + 40 + ((alg + 3) & ~size_t(3)) +
             2 * ((MS_P * MS_YP + 3) & ~3) + 16;
  return (n + 3) & ~size_t(3);
}
template <typename T, int HS>
__device__ __forceinline__ PipeLds<T, HS> pipe_carve(T* smem, int N) {
  PipeLds<T, HS> L;
  T* p = smem;
  L.hsz = N * HS; L.psz = N * 12;
  L.H0 = p; p += (size_t)2 * N * HS;
  L.P0 = p; p += (size_t)2 * N * 12;
  L.A.hist = L.H0;
  L.A.Xs = p; p += (MS_P * MS_YP + 3) & ~3;
  L.A.cold = p; p += (CD_SIZE + 3) & ~3;
  L.A.Ti = p; p += 40;
  L.A.XB = p;
  L.A.Tm = L.A.XB + 2 * MS_YP * 8;
  L.A.Es = L.A.Tm + 48;
  p += (2 * MS_YP * 8 + 48 + ((WAVE * MS_YP + 3) & ~3) + 3) & ~3;
  L.A.c12 = nullptr;
  L.Xpub = p; p += (MS_P * MS_YP + 3) & ~3;
  L.EsB = p; p += (MS_P * MS_YP + 3) & ~3;
  L.ctr_raw = p;
  return L;
}

// build_hist_cold with the (diagonal) material constants already in registers: inside the integrator's loop the
// compiler may not hoist the LDS reads of the cold table over the progress counter, and eight dependent LDS round trips
// per grid point cost as much as the physics
template <typename T>
struct HistConst {
  T ksei[3], ksev[3], bse[3], kbti[3], bbt[3];
};
template <typename T>
__device__ __forceinline__ HistConst<T> hist_const(const T* cold) {
  HistConst<T> c;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    c.ksei[k] = cold[CD_KSEI + 4 * k]; c.ksev[k] = cold[CD_KSEV + k]; c.bse[k] = cold[CD_BSE + 4 * k];
    c.kbti[k] = cold[CD_KBTI + 4 * k]; c.bbt[k] = cold[CD_BBT + 4 * k];
  }
  return c;
}
template <typename T, int HS, bool DIAG>
__device__ __forceinline__ void build_hist_pipe(const T* cold, const HistConst<T>& hc, T hc1, T hc2, const T (&cv)[12],
                                                const T (&pv)[12], T* dst) {
  if constexpr (DIAG) {
    T hv[HS];
#pragma unroll
    for (int k = 0; k < 12; ++k) hv[k] = hc1 * cv[k] + hc2 * pv[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      hv[12 + k] = hc.ksei[k] * (hc.ksev[k] - hc.bse[k] * hv[6 + k]);   // av, see hist_derive_cold
      hv[15 + k] = -hc.kbti[k] * (hc.bbt[k] * hv[9 + k]);               // au
    }
    if constexpr (HS > 18) { hv[18] = T(0); hv[19] = T(0); }
    store_vec<T, HS>(dst, hv);
  } else {
    build_hist_cold<T, HS, DIAG>(cold, hc1, hc2, cv, pv, dst);
  }
}

// Two rods per workgroup: four wavefronts, which the CU places on its four SIMDs - a rod's solver and integrator must
// not share one (measured with one rod per workgroup: both ran at half speed).
template <typename T, bool DIAG, int HS>
__global__ __launch_bounds__(4 * WAVE) void ms_pipe_kernel(const RodConst<T> Pc, const SimArgs<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = Pc.N;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wv = (threadIdx.x / WAVE) & 1;           // 0 solver, 1 integrator
  const int rib = threadIdx.x / (2 * WAVE);          // rod inside the workgroup
  const int tid = threadIdx.x & (2 * WAVE - 1);      // thread inside the rod's pair of wavefronts
  const int64_t rod_raw = (int64_t)blockIdx.x * 2 + rib;
  const bool valid = rod_raw < A.B;
  const int64_t rod = valid ? rod_raw : A.B - 1;
  const size_t rod_elems = (size_t)N * KR_SLOTS;
  const PipeLds<T, HS> L = pipe_carve<T, HS>(reinterpret_cast<T*>(smem_raw) + (size_t)rib * pipe_lds_elems<T, HS>(N), N);
  const MsRole R = ms_role(lane, N);
  if (wv == 0) ms_cold_fill<T>(Pc, L.A.cold, lane);
  LdsCounter* ctr = (LdsCounter*)L.ctr_raw;
  if (tid < 16) ctr[tid] = 0;
  __syncthreads();
  // history of step 0 and the leading slots of y_0 (and of the state before it, only needed for that history)
  const T* s0 = A.states + rod * rod_elems;
  const T* sp = A.prev_init ? A.prev_init + rod * rod_elems : s0;
  for (int j = tid; j < N; j += 2 * WAVE) {
    T cv[12], pv[12];
    load_hist_vec<T, 12>(s0 + (size_t)j * KR_SLOTS, cv);
    load_hist_vec<T, 12>(sp + (size_t)j * KR_SLOTS, pv);
    store_vec<T, 12>(L.P0 + (size_t)j * 12, cv);
    store_vec<T, 12>(L.P0 + L.psz + (size_t)j * 12, pv);
    build_hist_cold<T, HS, DIAG>(L.A.cold, A.hc1, A.hc2, cv, pv, L.H0 + (size_t)j * HS);
  }
  __syncthreads();
  if (!valid) return;  // (odd batch: the second rod of the last workgroup does not exist; no barrier follows)
  const T* ctl = A.ctl + rod * A.T_steps * 4;
  const MlpDev<T> Mnone{};

  if (wv == 0) {
    // =========================== SOLVER ===========================
    MsPred<T> Q;
    double* img = A.pred_io ? A.pred_io + (size_t)rod * MS_PRED_ROWS * WAVE : nullptr;
    if (img && A.pred_load) ms_pred_load<T>(Q, img, lane);
    else ms_pred_init<T>(Q, lane, R, s0, sp, A.prev_init != nullptr, A.predictor);
    MsSolveArgs<T> S;
    S.vlast = {T(0), T(0), T(0)}; S.ulast = S.vlast;  // (the solver never stores)
    S.out_rod = nullptr; S.tip = nullptr;
    S.tol = A.tol; S.tolA = A.tolA; S.fd_eps = A.fd_eps; S.maxit = A.maxit;
    S.kappa = Q.kappa;
    S.prog = ctr + 4; S.abort_flag = ctr + 8;
    T Gguess = lane < 6 ? A.G[rod * 6 + lane] : T(0);
    MsStamps stamps;
    int seq = 0;              // publications so far
    int redo_budget = 0;
    unsigned long long c_solve = 0, c_wait = 0, c_total0 = __builtin_amdgcn_s_memtime(), n_redo = 0, n_it = 0;
    auto publish = [&](int step, int report) __attribute__((always_inline)) {
      // (every lane has finished reading / writing Xs: ms_newton ends with a wave_sync)
      for (int e = lane; e < MS_P * MS_YP; e += WAVE) L.Xpub[e] = L.A.Xs[e];
      if (lane == 0) { ctr[1] = step; ctr[2] = report; ctr[9] = __float_as_int(S.amp); }
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wave_sync();
      ++seq;
      if (lane == 0) ctr[0] = seq;
    };
    auto wait_verdict = [&]() __attribute__((always_inline)) -> bool {  // of publication `seq`
      const unsigned long long w0 = __builtin_amdgcn_s_memtime();
      pipe_wait(ctr + 3, 2 * seq, ctr + 8);
      c_wait += __builtin_amdgcn_s_memtime() - w0;
      return (ctr[3] & 1) != 0 && ctr[3] >= 2 * seq;
    };
    // one full solve of step t from the start values in Xs; returns what to report (-1: leave it to the integrator)
    auto solve = [&](int64_t t, int prog_base, int& order, bool fresh) __attribute__((always_inline)) -> int {
      T tens[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) tens[k] = ctl[t * 4 + k];
      SweepCtx<T, HS> C;
      MsLds<T> LA = L.A;
      LA.hist = L.H0 + (int)(t & 1) * L.hsz;
      ms_ctx_init<T, HS>(LA.cold, LA.hist, tens, C);
      S.prog_base = prog_base;
      int status, it;
      const unsigned long long v0 = __builtin_amdgcn_s_memtime();
      while (true) {
        if (fresh) {
          ms_pred_guess<T>(Q, order, lane, L.A.cold, L.A.Xs);
          wave_sync();
          if (order <= 0 && lane < 6) L.A.Xs[0 * MS_YP + 7 + lane] = Gguess;  // caller's guess (knode.py:67,89)
          wave_sync();
        }
        status = ms_newton<T, DIAG, KR_EULER, HS, false, false, true>(Pc, Mnone, LA, R, lane, C, S, it, stamps);
        n_it += it;
        if (status == KR_ST_PIPE_READY || order == 0 || !fresh) break;
        order = 0;  // the predicted start did not converge: redo the step from the reference's warm start
      }
      c_solve += __builtin_amdgcn_s_memtime() - v0;
      return status == KR_ST_PIPE_READY ? -1 : status;
    };
    for (int64_t t = 0; t < A.T_steps && !ctr[8]; ++t) {
      int order = Q.next_order;
      // the history of step t is being written by the integrator (publication `seq`, step t - 1); for t = 0 it is complete
      int report = solve(t, t == 0 ? -PIPE_PB : seq * PIPE_PB, order, true);
      if (t > 0 && !wait_verdict() && !ctr[8]) {
        // step t - 1 was rejected: one more Newton iteration on it from the published unknowns (its history is intact),
        // publish again, wait; then this step from scratch
        redo_budget = 0;
        while (!ctr[8]) {
          ++n_redo;
          for (int e = lane; e < MS_P * MS_YP; e += WAVE) L.A.Xs[e] = L.Xpub[e];
          wave_sync();
          int o0 = 0;
          int rep = solve(t - 1, -PIPE_PB, o0, false);
          if (++redo_budget >= 8 && rep < 0) rep = KR_ST_MAXIT;
          publish((int)(t - 1), rep);
          if (wait_verdict()) break;
        }
        if (lane < 6) Gguess = L.Xpub[0 * MS_YP + 7 + lane];
        order = Q.next_order;
        report = solve(t, seq * PIPE_PB, order, true);
      }
      publish((int)t, report);
      ms_pred_update<T>(Q, order, report < 0 ? KR_ST_CONVERGED : report, A.predictor, lane, L.A.Xs, stamps);
      if (lane < 6) Gguess = L.A.Xs[0 * MS_YP + 7 + lane];
      wave_sync();
    }
    // the last step
    if (A.T_steps > 0 && !ctr[8] && !wait_verdict()) {
      const int64_t t = A.T_steps - 1;
      redo_budget = 0;
      while (!ctr[8]) {
        for (int e = lane; e < MS_P * MS_YP; e += WAVE) L.A.Xs[e] = L.Xpub[e];
        wave_sync();
        int o0 = 0;
        int rep = solve(t, -PIPE_PB, o0, false);
        if (++redo_budget >= 8 && rep < 0) rep = KR_ST_MAXIT;
        publish((int)t, rep);
        if (wait_verdict()) break;
      }
      if (lane < 6) Gguess = L.Xpub[0 * MS_YP + 7 + lane];
    }
    if (lane < 6) A.G[rod * 6 + lane] = Gguess;
    if (img) {
      Q.kappa = S.kappa;
      ms_pred_save<T>(Q, img, lane);
    }
    if (A.dbg && lane == 0) {  // diagnostics (kr_debug_buffer, [B][24]): cycles and counts of the solver
      unsigned long long* d = A.dbg + rod * 24;
      d[0] = __builtin_amdgcn_s_memtime() - c_total0; d[1] = c_solve; d[2] = c_wait; d[3] = n_redo; d[4] = n_it;
    }
  } else {
    // =========================== INTEGRATOR ===========================
    const int iv = lane < MS_P ? lane : 0;
    const bool act = lane < MS_P;
    const int s_i = ms_interval_start(iv, R.sbase, R.srem);
    const int len_i = R.sbase + (iv < R.srem ? 1 : 0);
    int expected = 1;
    const HistConst<T> hcst = hist_const<T>(L.A.cold);
    // The uniform parameters of the physics in VECTOR registers for this wavefront: the kernel as a whole is out of
    // scalar registers, and in this loop the compiler re-read them from the kernel-argument segment at every grid point
    // (five dependent s_load per point, ~1 k cycles).  The opaque asm keeps each value in a VGPR.
    RodConst<T> Pv;
    {
      auto pin = [](T x) { __asm__ volatile("" : "+v"(x)); return x; };
      Pv.c0 = pin(Pc.c0); Pv.c1 = Pv.c0; Pv.c2 = Pv.c0; Pv.ds = pin(Pc.ds); Pv.rhoA = pin(Pc.rhoA);
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const bool dg = !DIAG || k % 4 == 0;
        Pv.Ksei[k] = dg ? pin(Pc.Ksei[k]) : T(0); Pv.Kbti[k] = dg ? pin(Pc.Kbti[k]) : T(0);
        Pv.rhoJ[k] = dg ? pin(Pc.rhoJ[k]) : T(0); Pv.Bse[k] = T(0); Pv.Bbt[k] = T(0);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) Pv.C[k] = pin(Pc.C[k]);
      Pv.N = N; Pv.diag = Pc.diag;
    }
    unsigned long long c_int = 0, c_pubwait = 0;
    for (int64_t t = 0; t < A.T_steps; ++t) {
      const int64_t inx = A.ring ? (t + 1) % 3 : t + 1;
      T* out_rod = A.states + inx * A.slot_elems + rod * rod_elems;
      const T* Hc = L.H0 + (int)(t & 1) * L.hsz;
      T* Hn = L.H0 + (int)((t + 1) & 1) * L.hsz;
      const T* Pold = L.P0 + (int)(t & 1) * L.psz;
      T* Pnew = L.P0 + (int)((t + 1) & 1) * L.psz;
      T tens[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) tens[k] = ctl[t * 4 + k];
      SweepCtx<T, HS> C;
      ms_ctx_init<T, HS>(L.A.cold, Hc, tens, C);
      bool accepted = false;
      while (!accepted) {
        const unsigned long long p0 = __builtin_amdgcn_s_memtime();
        pipe_wait(ctr + 0, expected, ctr + 8);
        const unsigned long long p1 = __builtin_amdgcn_s_memtime();
        c_pubwait += p1 - p0;
        if (ctr[8]) break;
        const int report = ctr[2];
        const float amp = __int_as_float(ctr[9]);
        const int pbase = expected * PIPE_PB;
        RodState<T> y;
        {
          T yr[19];
#pragma unroll
          for (int q = 0; q < 19; ++q) yr[q] = L.Xpub[iv * MS_YP + q];
          y = rows_to_state(yr);
        }
        T hv[HS];
        load_hist_vec<T, HS>(Hc + (size_t)s_i * HS, hv);
        for (int tt = 0; tt < len_i; ++tt) {
          const int j = s_i + tt;
          RodState<T> k1;
          V3<T> v, u;
          eval_point<T, DIAG, false, HS>(Pv, Mnone, C, y, hv, k1, v, u);
          if (act) {
            T rec[KR_SLOTS];
            record_from(y, v, u, rec);
            store_record(out_rod + (size_t)j * KR_SLOTS, rec);
            T nv[12], ov[12];
#pragma unroll
            for (int c = 0; c < 12; ++c) nv[c] = rec[c];
            load_hist_vec<T, 12>(Pold + (size_t)j * 12, ov);
            store_vec<T, 12>(Pnew + (size_t)j * 12, nv);
            build_hist_pipe<T, HS, DIAG>(L.A.cold, hcst, A.hc1, A.hc2, nv, ov, Hn + (size_t)j * HS);
            // LDS executes a wavefront's accesses in order: the counter lands after the record.  (A release fence would
            // also wait for the state record on its way to HBM: 2 k cycles per grid point.)
            __asm__ volatile("" ::: "memory");
            ctr[4 + iv] = pbase + tt + 1;
          }
          load_hist_vec<T, HS>(Hc + (size_t)(j + 1) * HS, hv);
          y = state_axpy(y, Pv.ds, k1);
        }
        if (act) {
          T er[19];
          state_to_rows(y, er);
#pragma unroll
          for (int q = 0; q < 19; ++q) L.EsB[iv * MS_YP + q] = er[q];
          if (iv == MS_P - 1) {
            // last grid point: its z is never touched by a sweep (copied from the state before)
            T ov[12];
            load_hist_vec<T, 12>(Pold + (size_t)(N - 1) * 12, ov);
            const V3<T> vl{ov[6], ov[7], ov[8]}, ul{ov[9], ov[10], ov[11]};
            T rec[KR_SLOTS];
            record_from(y, vl, ul, rec);
            store_record(out_rod + (size_t)(N - 1) * KR_SLOTS, rec);
            T nv[12];
#pragma unroll
            for (int c = 0; c < 12; ++c) nv[c] = rec[c];
            store_vec<T, 12>(Pnew + (size_t)(N - 1) * 12, nv);
            build_hist_cold<T, HS, DIAG>(L.A.cold, A.hc1, A.hc2, nv, ov, Hn + (size_t)(N - 1) * HS);
            if (A.tip) {
              T* tp = A.tip + (rod * A.T_steps + t) * 3;
              tp[0] = y.p.x; tp[1] = y.p.y; tp[2] = y.p.z;
            }
          }
        }
        wave_sync();
        // residual of the sweep against the published unknowns (ms_residual_norm with the end states in EsB)
        float rn = 0.f;
        if (lane < 3 * MS_YP) {
          const int g = lane / MS_YP, r = lane - MS_YP * g;
          const T x = L.Xpub[(g + 1) * MS_YP + r];
          rn = update_ratio(L.EsB[g * MS_YP + r] - x, x);
        } else if (lane < 3 * MS_YP + 6) {
          const int k = lane - 3 * MS_YP;
          const T e = L.EsB[(MS_P - 1) * MS_YP + 7 + k];
          rn = update_ratio(L.A.cold[CD_FTIP + k] - e, e);
        }
        rn = wave_max_nonneg(rn);
        int verdict_ok;
        int status = KR_ST_CONVERGED;
        if (report >= 0) { verdict_ok = 1; status = report; }       // the solver gave up on this step: report its status
        else verdict_ok = (amp > 0.f && T(256) * (T)(amp * rn) <= A.tol) ? 1 : 0;
        if (verdict_ok && lane == 0 && A.status) A.status[rod * A.T_steps + t] = status;
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wave_sync();
        if (lane == 0) ctr[3] = 2 * expected + verdict_ok;
        ++expected;
        accepted = verdict_ok != 0;
        c_int += __builtin_amdgcn_s_memtime() - p1;
      }
      if (t == A.T_steps - 1 && A.dbg && lane == 0) { A.dbg[rod * 24 + 5] = c_int; A.dbg[rod * 24 + 6] = c_pubwait; }
      if (ctr[8]) {
        // a wait ran out of polls: report the remaining steps as not converged and leave
        for (int64_t tr = t; tr < A.T_steps; ++tr)
          if (lane == 0 && A.status) A.status[rod * A.T_steps + tr] = KR_ST_MAXIT;
        break;
      }
    }
  }
}

template <typename T, int HS>
static size_t pipe_lds_bytes(int N) { return 2 * sizeof(T) * pipe_lds_elems<T, HS>(N); }  // per workgroup = two rods

template <typename T, bool DIAG>
static int launch_pipe_inst(const RodConst<T>& P, const SimArgs<T>& a, hipStream_t s) {
  auto kern = ms_pipe_kernel<T, DIAG, hs_phys<T>()>;
  const size_t smem = pipe_lds_bytes<T, hs_phys<T>()>(P.N);
  static thread_local size_t configured = 0;
  if (smem > 48 * 1024 && smem > configured) {
    KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    configured = smem;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)((a.B + 1) / 2)), dim3(4 * WAVE), smem, s, P, a);
  KR_HIP(hipGetLastError());
  return KR_OK;
}

// the pipelined form applies (and pays: every rod gets two SIMDs) when B <= 512 and two rods fit the LDS of a CU
template <typename T>
static bool pipe_eligible(kr_handle* h, int scheme, int use_nn, const SimArgs<T>& a) {
  const RodConst<T>& P = consts<T>(h);
  if (h->pipeline == 0 || use_nn || scheme != KR_EULER) return false;
  if (P.N - 1 < 2 * MS_P) return false;
  const size_t bytes = pipe_lds_bytes<T, hs_phys<T>()>(P.N);
  if (bytes > (size_t)h->lds_limit) return false;
  if (h->pipeline == 2) return true;  // forced
  return a.B * 2 <= 1024;  // (one workgroup = two rods = four wavefronts per CU)
}

}  // namespace kr
