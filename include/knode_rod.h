/*
 * knode_rod.h - C ABI of libknode_rod.so, the MI355X (gfx950) backend for the
 * KNODE-Cosserat rod hot path.
 *
 * The reference (hsiehScalAR/KNODE-Cosserat) is pure Python and defines no FFI;
 * its "interface" for this path is the duck-typed surface of
 *   knode_cosserat/cosserat_ode.py        (class CosseratRod)
 *   knode_cosserat/cosserat_ode_torch.py  (class CosseratRodTorch)
 *   knode_cosserat/knode.py               (setup_robot, simulate)
 * Every entry point below names the reference function (file:line) it
 * replaces.  The Python shims in knode-cosserat_amd/ bind these symbols with
 * ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - every function returns 0 on success, <0 on error (KR_E_*);
 *    kr_last_error() gives the message of the calling thread's last error.
 *  - all array arguments are DEVICE pointers owned by the caller unless the
 *    name ends in _host; `stream` is a hipStream_t passed as void* (NULL = the
 *    null stream).  Calls are asynchronous on that stream.
 *  - one handle per (device, parameter set); a handle is not thread-safe,
 *    distinct handles are.  A handle may be used with several streams from
 *    ONE host thread: its calls share per-handle scratch (predictor images,
 *    resume steps of the two-launch simulate, history workspaces, loss
 *    partials), so a call on another stream than the handle's previous call
 *    is made to wait for everything that call's stream had queued (an event,
 *    no host synchronisation) - calls on one handle never overlap on the
 *    device; use one handle per stream for concurrency.  LIFETIME: the
 *    handle keeps the hipStream_t of its previous call to record that event
 *    on; a stream must therefore stay alive until the handle's NEXT call has
 *    returned (or the handle is destroyed) - destroy streams after the handle
 *    has moved on, not between two of its calls.  The current HIP
 *    device of the calling thread must be the handle's (KR_E_ARG otherwise).
 *  - dtype selects the arithmetic type of the call: KR_F32 or KR_F64.  All
 *    floating-point array arguments of a call have that element type.
 *
 * Rod state in HBM ("packed state", one record per grid point):
 *    state[b][j][KR_SLOTS]     b < B rods, j < N grid points
 *    slot  0.. 2  q   (y rows 13..15)     slot 12..14  p (y rows 0..2)
 *    slot  3.. 5  w   (y rows 16..18)     slot 15..18  h (y rows 3..6)
 *    slot  6.. 8  v   (z rows 0..2)       slot 19..21  n (y rows 7..9)
 *    slot  9..11  u   (z rows 3..5)       slot 22..24  m (y rows 10..12)
 *    slot 25..27  padding (kept zero)
 * The twelve leading slots are exactly what the BDF2 history terms of the next
 * step need (cosserat_ode.py:146-148 uses only q_t, w_t, v_t, u_t), so the
 * solver reads 96 contiguous bytes (fp64) per grid point and writes one
 * 16-byte-aligned record.  kr_state_pack / kr_state_unpack convert from/to
 * the reference's feature-major arrays y[19][N], z[6][N].
 */
#ifndef KNODE_ROD_H
#define KNODE_ROD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KR_SLOTS 28
#define KR_NY 19
#define KR_NZ 6
#define KR_MAX_LAYERS 8

enum { KR_F32 = 0, KR_F64 = 1 };
enum { KR_EULER = 0, KR_RK4 = 1 };
/* activation codes, cosserat_ode.py:99-108 */
enum { KR_ACT_NONE = 0, KR_ACT_TANH = 1, KR_ACT_SOFTPLUS = 2, KR_ACT_RELU = 3, KR_ACT_ELU = 4 };
/* per-rod solver status written by kr_step_batch */
enum { KR_ST_CONVERGED = 0, KR_ST_MAXIT = 1, KR_ST_NONFINITE = 2 };

enum {
  KR_OK = 0,
  KR_E_ARG = -1,     /* bad argument (null pointer, size, enum) */
  KR_E_HIP = -2,     /* HIP runtime error */
  KR_E_STATE = -3,   /* call order (e.g. NN requested but kr_set_mlp never called) */
  KR_E_UNSUPPORTED = -4
};

/* Independent rod parameters, cosserat_ode.py:15-47 (NumPy class) /
 * cosserat_ode_torch.py:14-45 (torch twin).  Matrices are row-major 3x3. */
typedef struct kr_params {
  double L;              /* rod length */
  int32_t N;             /* grid points; segments = N-1; ds = L/(N-1) */
  int32_t nn_input_history; /* 0: MLP input [y,z,tf] (28); 1: [y,yh,z,zh,tf] (53) */
  double E, r, rho;      /* Young's modulus, radius, density */
  double vstar[3];
  double g[3];
  double Bse[9];
  double Bbt[9];
  double C[3];
  double del_t;
  double F_tip[3];
  double M_tip[3];
  double tendon_dirs[12]; /* 4 x 3 */
  double p0[3];
  double h0[4];
  double q0[3];
  double w0[3];
} kr_params;

/* Dependent terms, cosserat_ode.py:58-78; returned so that the Python shim can
 * expose the attributes estimate_state.py and the drivers read. */
typedef struct kr_derived {
  double A, G, ds, c0, c1, c2, rhoA;
  double J[9], Kse[9], Kbt[9];
  double Kse_plus_c0_Bse_inv[9];
  double Kbt_plus_c0_Bbt_inv[9];
  double Kse_vstar[3];
  double rhoAg[3];
  double rhoJ[9];
} kr_derived;

typedef struct kr_handle kr_handle;

/* ---- library / handle ------------------------------------------------- */
const char* kr_last_error(void);
int kr_version(void);
/* fills the struct with the reference's class defaults (cosserat_ode.py:15-47) */
int kr_default_params(kr_params* out);
/* applies knode.setup_robot (knode.py:6-53); mod NULL or "" = no modifier.
 * Unknown mod -> KR_E_ARG, like the reference's exception. */
int kr_apply_preset(kr_params* inout, const char* mod);
/* The legacy ("original") parameter set of knode_cosserat_realworld/prepare.py:35-73
 * (setup_robot_original: del_t 0.005, L 0.4, E 209e9, r 0.0012, rho 8000, Bbt 5e-4 I) with its own
 * modifiers None|nsw|short|damping|diameter|youngs|dampstiff|lengthstiff. */
int kr_apply_preset_original(kr_params* inout, const char* mod);

int kr_create(const kr_params* p, int device, kr_handle** out);
int kr_destroy(kr_handle* h);
/* Solver options (integers):
 *   "ms_mode"        -1 auto (default) / 0 off / 1 forced: multiple-shooting form of the time-step
 *                    kernel (one rod per wavefront, 4 sub-intervals) for small batches
 *   "ms_batch_limit" auto mode uses it when B <= limit (default: no limit - it is the faster kernel at
 *                    every batch size measured; lower it to send large batches to the single-shooting kernel)
 *   "persistent"     1 (default) / 0: kr_simulate_batch runs all steps in one launch when the
 *                    multiple-shooting kernel applies
 *   "waves_per_rod"  0 auto (default) / 1 / 2 / 4: wavefronts that share one rod (7 or 13 sub-intervals instead of 4) -
 *                    for batches that would leave SIMDs idle (B waves_per_rod <= 1024).  With the MLP on: the persistent
 *                    form of kr_simulate_batch only (Euler sweeps, a network the matrix-core evaluator serves).
 *                    "last_waves_per_rod" / "last_sim_path" (read-only) tell what the last call ran.
 *   "mlp_grad_accumulate" 0 (default) / 1: see kr_adam_step
 *   "keep_predictor" 0 (default) / 1: kr_simulate_batch leaves the state of its start-value predictor behind
 *                    and the next call with the same batch size resumes from it - for a simulation that is
 *                    advanced by several calls, each continuing where the previous one stopped
 *                    (states[0] of a call = the last state of the one before).  Setting it to 0 drops
 *                    the stored state.
 *   "residual_test"  1 (default) / 0: the multiple-shooting kernels may accept a storing sweep from its RESIDUAL alone.
 *                    The documented stopping rule is on the Newton update (|dG|, |dY_i| <= tol max(1, |.|)); forming
 *                    that update costs a condensation.  With the option on, a storing sweep that follows an update
 *                    <= 1e-2 is first judged by amp x |residual|, amp = |update| / |residual| measured at the preceding
 *                    full iteration of the same solve, and accepted when 256 x that estimate is below the tolerance
 *                    (the two residuals point in different directions, so the ratio is only indicative: audited at
 *                    most 39 x off over 8 workloads x 1024 rods x 300 steps - DESIGN.md section 4; the factor 256 is
 *                    the margin).  status still reports "converged"; with 0 every accepted sweep carries a measured
 *                    update (Newton, or chord within 1 %) below the tolerance, at ~5 % of the throughput.
 *   "nn_base_only_store" 1 (default) / 0: MLP on, persistent one-wavefront kernel: a storing sweep that follows an update
 *                    <= 1e-2 evaluates the network at the unperturbed inputs only (no forward-difference columns: half an
 *                    evaluation) and is judged by the residual test above (audited with the MLP on: at most 46 x off) or,
 *                    failing that, by a CHORD update through the factors of the last full sweep - accepted below HALF
 *                    the tolerance, otherwise applied as the iteration's update; after two such sweeps the next one is
 *                    full.  0: every sweep carries its columns (~15 % slower; tools/bo_check.py compares the two).
 *   "nn_lowp_first"  1 (default) / 0: fp64 sweeps, MLP on: the FIRST sweep of a step whose predecessor needed two
 *                    corrections evaluates the network's base chain in fp32 (the accepted state never comes from it).
 *   "overlap"        1 (default) / 0: persistent form only - verify step t on spare lanes of the Jacobian sweep of
 *                    step t + 1 (kr_mso_impl.hpp; Euler sweeps, MLP off, diagonal material matrices)
 *   "msw_overlap"    1 (default) / 0: the same overlap where a rod owns 2 or 4 wavefronts (kr_mswo_impl.hpp; Euler
 *                    sweeps, MLP off, diagonal material matrices; the three tiles of leading slots in the LDS where
 *                    they fit beside the condensation tiles, else read from the states themselves: long rods)
 *   "predictor"      0..8: how kr_simulate_batch may form the initial guess of each step (default 8;
 *                    0 = the reference's warm start).  1..7: highest order of polynomial time
 *                    extrapolation; the persistent kernel picks, rod by rod and step by step, the
 *                    order <= this that would have predicted the step just solved best.  8 adds a
 *                    three-tap linear recurrence fitted per rod to the last steps, used whenever
 *                    it predicted the step just solved better than every polynomial.  One launch
 *                    per step uses min(this, 2).  The guess never changes the solution, only the
 *                    number of Newton sweeps. */
int kr_set_option(kr_handle* h, const char* name, int value);
/* reads an option back; additionally "last_sim_path": what the last kr_simulate_batch did -
 * 0 one single-shooting launch per step, 1 one multiple-shooting launch per step, 2 one
 * persistent launch for all steps */
int kr_get_option(kr_handle* h, const char* name, int* value);
/* Diagnostics (NULL switches them off).  While a buffer is set, kr_simulate_batch with one launch per step
 * writes the number of Newton sweeps of every rod and step into it as int32 [B][T]; in builds with
 * -DKR_MS_STAMPS (make dbg) the persistent kernel fills it as uint64 [B][24] with per-rod cycle counters
 * {total, sweep, algebra, prologue, iterations, ...} instead (tools/ms_stamps.py). */
int kr_debug_buffer(kr_handle* h, void* dev_ptr);
/* CosseratRod.compute_intermediate_terms, cosserat_ode.py:58-78 */
int kr_set_params(kr_handle* h, const kr_params* p);
int kr_get_derived(const kr_handle* h, kr_derived* out);
/* same derivation without a handle or a GPU (host arithmetic only) */
int kr_derive(const kr_params* p, kr_derived* out);

/* Residual MLP, cosserat_ode.py:90-112 / cosserat_ode_torch.py:60-62,131-134.
 * dims[n_layers+1]; W[k] is row-major [dims[k+1]][dims[k]] float32 (the dtype
 * the reference trains and stores), b[k] is [dims[k+1]]; acts[k] is applied
 * after layer k.  src_on_device: 0 = host pointers, 1 = device pointers.
 * The library keeps its own packed copies.  n_layers = 0 switches the MLP off. */
int kr_set_mlp(kr_handle* h, int n_layers, const int32_t* dims, const int32_t* acts,
               const float* const* W, const float* const* b, int src_on_device, void* stream);

/* CosseratRod.get_nn_output (cosserat_ode.py:90-112) on Q rows:
 * x[Q][dims[0]] -> out[Q][25], row-major, f32 or f64 arithmetic. */
int kr_mlp_eval_batch(kr_handle* h, int64_t Q, const void* x, void* out, int dtype, void* stream);

/* ---- batched per-segment derivative ----------------------------------- */
/* CosseratRodTorch.ODE_parallel (cosserat_ode_torch.py:217-322) and, row by
 * row, CosseratRod.ODE (cosserat_ode.py:114-186).
 * y[Q][19], yh[Q][19], zh[Q][6], tf[Q][3] -> dys[Q][19], z[Q][6], row-major.
 * use_nn != 0 adds the MLP correction (needs kr_set_mlp). */
int kr_ode_batch(kr_handle* h, int64_t Q, const void* y, const void* yh, const void* zh, const void* tf,
                 void* dys, void* z, int use_nn, int dtype, void* stream);

/* Derivatives of the PHYSICS of kr_ode_batch (use_nn = 0), what torch autograd propagates through
 * CosseratRodTorch.ODE_parallel (cosserat_ode_torch.py:264-306) and through the op-by-op graph of
 * CosseratRodTorch.getResidualEuler / ODE (:137-213, :325-367).  fp64 forward-mode differentiation on the
 * device whatever `dtype` (the element type of the arrays).
 *   kr_ode_vjp_batch: cotangents g_dys[Q][19], g_z[Q][6] -> J^T g with respect to y, yh, zh, tf
 *     (g_y[Q][19], g_yh[Q][19], g_zh[Q][6], g_tf[Q][3]; any of the four may be NULL).
 *   kr_ode_jacobian_batch: jac[Q][25][19] = d(dys, z) / dy of every row.
 * cut != 0 reproduces the reference's ODE graph where it differs from the function it evaluates: the quadratic part of
 * the rotation matrix (:159-162) and the quaternion-rate matrix (:185-189) are built with torch.tensor([...]), i.e.
 * as leaves - h is then seen through 2 / (h . h) only, and h_s does not see u.  cut = 0: the complete derivative
 * (ODE_parallel's graph). */
int kr_ode_vjp_batch(kr_handle* h, int64_t Q, const void* y, const void* yh, const void* zh, const void* tf,
                     const void* g_dys, const void* g_z, int cut, void* g_y, void* g_yh, void* g_zh, void* g_tf,
                     int dtype, void* stream);
int kr_ode_jacobian_batch(kr_handle* h, int64_t Q, const void* y, const void* yh, const void* zh, const void* tf,
                          int cut, void* jac, int dtype, void* stream);

/* ---- packed state helpers --------------------------------------------- */
/* knode.py:58-66: straight rod along +z, unit quaternion, v = e3 */
int kr_state_init_straight(kr_handle* h, int64_t B, void* state, int dtype, void* stream);
/* y_fm[B][19][N], z_fm[B][6][N] (reference layout) <-> state[B][N][KR_SLOTS] */
int kr_state_pack(kr_handle* h, int64_t B, const void* y_fm, const void* z_fm, void* state, int dtype, void* stream);
int kr_state_unpack(kr_handle* h, int64_t B, const void* state, void* y_fm, void* z_fm, int dtype, void* stream);
/* rows [y; z; yh; zh] of one knode.simulate trajectory entry (knode.py:96):
 * out[B][50][N] from state (step k) and the two states before it */
int kr_state_unpack50(kr_handle* h, int64_t B, const void* state, const void* state_m1, const void* state_m2,
                      void* out, int dtype, void* stream);
/* tip[B][3] = p of the last grid point */
int kr_state_tip(kr_handle* h, int64_t B, const void* state, void* tip, int dtype, void* stream);

/* ---- shooting residual ------------------------------------------------- */
/* CosseratRod.getResidualEuler / getResidualRK4 (cosserat_ode.py:188-255):
 * one sweep from the guessed base wrench G[B][6]; writes the swept state into
 * state_next (the reference mutates y, z in place) and r[B][6] =
 * [F_tip - n(L), M_tip - m(L)].  History terms come from state_cur, state_prev
 * (knode.py:74-75).  tensions[B][4].
 * hist_is_explicit != 0: state_cur holds the history terms yh, zh themselves
 * (packed like a state) and state_prev is ignored - the calling convention of
 * the reference method, which receives yh, zh as arguments. */
int kr_residual_batch(kr_handle* h, int64_t B, int scheme, const void* G, const void* state_prev,
                      const void* state_cur, void* state_next, const void* tensions, void* r, int use_nn,
                      int hist_is_explicit, int dtype, void* stream);
/* getResidualRK4 with the CALLER'S midpoint histories (cosserat_ode.py:215-255: stages 2 and 3 read
 * yh_int[:, j], zh_int[:, j], lines 225 and 233-234).  hist holds yh, zh and hist_mid holds yh_int, zh_int, both
 * packed like a state ([B][N][KR_SLOTS]; record j of hist_mid belongs to the midpoint between grid points j and
 * j + 1, record N-1 is not read).  hist_mid == NULL gives kr_residual_batch(..., hist_is_explicit = 1), i.e. the
 * linear interpolation knode.simulate hands over (knode.py:80-81).  scheme must be KR_RK4 when hist_mid is given
 * (the Euler sweep ignores the midpoints, cosserat_ode.py:188-213). */
int kr_residual_mid_batch(kr_handle* h, int64_t B, int scheme, const void* G, const void* hist, const void* hist_mid,
                          void* state_next, const void* tensions, void* r, int use_nn, int dtype, void* stream);

/* ---- one implicit time step -------------------------------------------- */
/* Body of the loop in knode.simulate (knode.py:70-100) for B rods: BDF2
 * history from (state_cur, state_prev), shooting solve for G (Newton with a
 * forward-difference Jacobian instead of MINPACK hybrd; same root), final
 * swept state into state_next.  G[B][6] is read as the initial guess and
 * overwritten with the solution.  tol: stop when |dG|_inf <= tol*max(1,|G|_inf)
 * (<=0 selects 1e-8 for f64 - the class of the reference's fsolve xtol=1.49e-8 - and 1e-5 for f32); maxit <= 0 selects 30.
 * status[B], iters[B] (int32) may be NULL.  A step that plain Newton does not solve within maxit sweeps (from the
 * predicted and from the caller's start) is redone by damped Newton with its OWN cap of 8 x maxit sweeps
 * (backtracking rejections count as sweeps); iters reports the sum of both phases.
 * Initial guess: predictor = 0 starts Newton from the caller's G (the
 * reference's warm start, knode.py:89); 1 / 2 extrapolate the unknowns linearly /
 * quadratically in time from state_cur, state_prev (and state_prev2, may be
 * NULL; it may alias state_next, it is read before anything is written);
 * -1 = the highest order the given states allow (prev == cur means "no history"). */
int kr_step_batch(kr_handle* h, int64_t B, int scheme, const void* state_prev, const void* state_cur,
                  void* state_next, void* G, const void* tensions, double tol, int maxit, int32_t* status,
                  int32_t* iters, int use_nn, const void* state_prev2, int predictor, int dtype, void* stream);

/* T steps of the above in one call.  ctl[B][T][4]; states[(T+1)][B][N][KR_SLOTS]
 * with states[0] the initial condition (e.g. kr_state_init_straight) - step t
 * writes states[t+1]; if ring != 0, `states` holds only 3 slots used
 * cyclically (slot (t+1)%3) for tip-only runs: on return they hold the complete
 * last three states of the call (states T, T-1, T-2); what passes through them
 * before that is scratch of the library (the persistent kernels write interior
 * states only as far as they may have to re-read them: the twelve leading slots).  tip[B][T][3] may be NULL.
 * status[B][T] may be NULL.  Replaces knode.simulate (knode.py:55-102).
 * state_prev_init: NULL = the reference's start (y_prev = y before the first
 * step, knode.py:65-66); otherwise the packed state one step before states[0],
 * which makes a second call continue a run exactly (it may point into the ring).
 * When the multiple-shooting kernel applies (see kr_set_option) all T steps run
 * in ONE launch: every wavefront keeps its rod's history in LDS / registers. */
/* Optional: the host-side one-time work of the first kr_simulate_batch call for batches of B rods (per-batch scratch
 * allocation, kernel lookup in the code object, LDS limits: ~0.4 ms) ahead of time, so that a latency-critical first
 * call does not carry it.  Launches nothing.  No counterpart in the reference. */
int kr_simulate_prepare(kr_handle* h, int64_t B, int dtype);
int kr_simulate_batch(kr_handle* h, int64_t B, int64_t T, int scheme, const void* ctl, void* states, int ring,
                      void* G, void* tip, double tol, int maxit, int32_t* status, int use_nn,
                      const void* state_prev_init, int dtype, void* stream);

/* ---- KNODE one-step-ahead training path -------------------------------- */
/* CosseratRodTorch.parallelGetNextSegmentEuler (cosserat_ode_torch.py:401-437)
 * and, with idx = 1..N-1, getNextSegmentEuler (:370-399), plus the four-term
 * loss of physics_train.py:252-259.  Defined in round 1 as three calls so that
 * torch autograd only sees dense tensors:
 *
 * kr_next_segment_physics: for every (s, k): j = idx[k]-1, row = s*K+k
 *    x[row][in_pad]   MLP input  [y_j, z_phys, tf] (or the 53-wide history form), zero padded
 *    base[row][25]    [y_j + ds*ys_phys, z_phys]    (prediction without the MLP)
 * Gs[S][25][N], yh[S][19][N], zh[S][6][N] feature-major (reference layout),
 * tensions[S][4], idx[K] int32 on device. f32 or f64. */
int kr_next_segment_physics(kr_handle* h, int64_t S, int K, const void* Gs, const void* yh, const void* zh,
                            const void* tensions, const int32_t* idx, void* x, int in_pad, void* base,
                            int dtype, void* stream);

/* MLP forward over Q rows on the matrix cores (fp32 MFMA: exact fp32, the
 * reference's training precision): out[Q][32] (columns 0..dims[n_layers]-1 valid -
 * 25 for the rod's MLP, at most 32 - the rest zero) = MLP(x[Q][in_pad]); keeps pre-activations and activations of
 * the hidden layers in `ws` (kr_mlp_ws_bytes) for the backward pass.  Weights
 * are the caller's device tensors (torch parameters), row-major like nn.Linear:
 * W[k][dims[k+1]][dims[k]], b[k][dims[k+1]]; acts[n_layers-1] must be NONE. */
size_t kr_mlp_ws_bytes(int n_layers, const int32_t* dims, int64_t Q);
int kr_mlp_forward(kr_handle* h, int64_t Q, int n_layers, const int32_t* dims, const int32_t* acts,
                   const float* const* W, const float* const* b, const float* x, int in_pad, float* out,
                   void* ws, void* stream);
/* dW[k], db[k] (overwritten) from dout[Q][32] = d loss / d out and the `ws` of
 * the matching forward call. */
int kr_mlp_backward(kr_handle* h, int64_t Q, int n_layers, const int32_t* dims, const int32_t* acts,
                    const float* const* W, const float* x, int in_pad, const float* dout, const void* ws,
                    float* const* dW, float* const* db, void* stream);

/* pred[row][25] = base + [ds*out[:19], out[19:]] (row = s*K + k); four-term
 * loss of physics_train.py:252-259 against target[S][25][N] at columns idx[k]
 * (z rows against idx[k]-1, :259): for every s the sum of four nn.MSELoss
 * (mean) terms - p rows, n/m/q/w rows, quaternion_to_euler(h rows), z rows -
 * summed over s and divided by `denom`.  Writes pred, *loss and
 * dout[row][32] = d loss / d out (columns 25..31 zero).  out is [rows][32]. */
int kr_loss_fwd_bwd(kr_handle* h, int64_t S, int K, const float* base, const float* out, const float* target,
                    const int32_t* idx, double denom, float* pred, float* loss, float* dout, void* stream);

/* torch.optim.Adam (amsgrad off) on one flat fp32 parameter vector - physics_train.py:199,289-296 - followed by
 * the reference's clamp (physics_train.py:299-304: params = max(params, lower), lower[i] = 0 for weight-matrix
 * entries and -inf for biases; NULL = no clamp) and by zeroing grads[0 .. n_zero) for the next epoch
 * (n_zero >= n covers trailing slots of the same buffer, e.g. the loss).  step = 1, 2, ... (bias correction).
 * With the option "mlp_grad_accumulate" = 1, kr_mlp_backward and kr_loss_rows_fwd_bwd add into dW / db / loss
 * instead of zeroing them first, so that this call is the only one that touches the buffers between epochs. */
int kr_adam_step(kr_handle* h, int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                 const float* lower, double lr, double beta1, double beta2, double eps, double weight_decay,
                 int64_t step, int64_t n_zero, void* stream);
/* kr_adam_step with the learning rate and torch.optim.lr_scheduler.ReduceLROnPlateau(mode "min", relative
 * threshold, cooldown 0) kept on the DEVICE - physics_train.py:206 creates the scheduler, :296-297 call
 * optimizer.step(); scheduler.step(total_loss) every epoch - so that an epoch needs no host round trip.
 * sched: 6 doubles in device memory, initialised by the caller to {lr, lr, +inf, 0, 0, 0}: [0]/[1] learning rate
 * of odd / even steps (step s uses sched[(s-1)&1], the rate of step s+1 goes to sched[s&1]), [2] best loss,
 * [3] bad epochs in a row, [4] loss of the step just taken, [5] reductions so far.  The loss is read from
 * grads[loss_index] (the trailing slot of the flat buffer, already all-reduced) before that slot is zeroed;
 * loss_log (nullable, device) receives it as a float.  Note: the fp32 step size is formed as lr * (1 / bias
 * correction) on the device; kr_adam_step forms lr / bias correction on the host - the same to rounding. */
int kr_adam_plateau_step(kr_handle* h, int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                         const float* lower, double* sched, double beta1, double beta2, double eps,
                         double weight_decay, int64_t step, int64_t n_zero, int64_t loss_index, double factor,
                         int patience, double threshold, double min_lr, float* loss_log, void* stream);

/* ONE EPOCH of the one-step-ahead training loop on one rank - the body of physics_train.py:283-304 / :388-401 for the
 * epoch's whole data set: prediction of every scored row from the MLP, the four-term loss, loss.backward() restricted to
 * the MLP, optimizer.step() (Adam), scheduler.step(total_loss) (ReduceLROnPlateau), the weight clamp - as 3 kernel launches
 * (forward + loss, backward, optimizer tail) and no host round trip.  It is kr_mlp_forward_loss + kr_mlp_backward +
 * kr_adam_plateau_step with the glue between them removed: the loss partials and the per-workgroup gradient slabs are
 * added up inside the optimizer launch (fixed order, no atomics: an epoch is reproducible bit for bit), and that launch
 * also writes every updated parameter into the MFMA fragment buffers of `ws`, so the next epoch starts without a packing
 * launch.
 *   params, grads, exp_avg, exp_avg_sq, lower: flat fp32 vectors in nn.Linear order W1 [out][in], b1, W2, b2, (W3, b3);
 *   grads has ONE trailing slot (the loss) and must be zero on entry of phase 0 / 1 (every update leaves it so);
 *   x [S*K][in_pad], base, target_rows [S*K][25], dout [S*K][32] (scratch), ws (kr_mlp_ws_bytes), sched, the Adam and
 *   plateau arguments, step, loss_log: as in kr_mlp_forward_loss and kr_adam_plateau_step.
 *   phase 0: the whole epoch.  phase 1: everything up to the update - grads then holds the summed gradients and the loss,
 *   ready for a data-parallel all-reduce - and phase 2: the update from grads as they stand (after the all-reduce).
 *   repack != 0: the caller changed `params` since the last call (a loaded checkpoint): fragments are packed afresh.
 *   They are also packed whenever ws, params or the network differ from the handle's previous call, and after any
 *   kr_mlp_forward[_loss] / kr_adam[_plateau]_step on the same ws / params.  "Differ" is judged by ADDRESS: a ws or
 *   params buffer that was freed and allocated again at the same address with other contents needs repack = 1 on its
 *   first call (a trainer object passes it on its first epoch).
 * KR_E_UNSUPPORTED for networks the fused training kernels do not serve (use the three separate calls). */
int kr_train_epoch(kr_handle* h, int64_t S, int K, int n_layers, const int32_t* dims, const int32_t* acts, float* params,
                   float* grads, float* exp_avg, float* exp_avg_sq, const float* lower, double* sched, const float* x,
                   int in_pad, const float* base, const float* target_rows, double denom, float* dout, void* ws,
                   double beta1, double beta2, double eps, double weight_decay, int64_t step, double factor,
                   int patience, double threshold, double min_lr, float* loss_log, int phase, int repack, void* stream);
/* n_epochs consecutive epochs of phase 0 queued by ONE call (the loop of physics_train.py:306-408 on one rank): epoch e runs
 * as kr_train_epoch(..., step + e, ..., loss_log + e, phase 0, repack for e = 0 only) - same launches, same results bit for
 * bit; what it saves is the caller's per-epoch host work (a Python / ctypes caller needs longer to issue an epoch than the
 * GPU to run it: 135 us against 128 at BASELINE cfg3).  loss_log must hold n_epochs floats (or be NULL). */
int kr_train_epochs(kr_handle* h, int64_t n_epochs, int64_t S, int K, int n_layers, const int32_t* dims, const int32_t* acts,
                    float* params, float* grads, float* exp_avg, float* exp_avg_sq, const float* lower, double* sched,
                    const float* x, int in_pad, const float* base, const float* target_rows, double denom, float* dout, void* ws,
                    double beta1, double beta2, double eps, double weight_decay, int64_t step, double factor, int patience,
                    double threshold, double min_lr, float* loss_log, int repack, void* stream);

/* The same loss against pre-gathered targets: the states a training set is scored against never
 * change between epochs, so kr_gather_targets extracts rows[S*K][25] once (y rows at column idx[k],
 * z rows at idx[k]-1) and kr_loss_rows_fwd_bwd reads them contiguously.  pred may be NULL. */
int kr_gather_targets(kr_handle* h, int64_t S, int K, const float* target, const int32_t* idx, float* rows,
                      void* stream);
int kr_loss_rows_fwd_bwd(kr_handle* h, int64_t S, int K, const float* base, const float* out,
                         const float* target_rows, double denom, float* pred, float* loss, float* dout,
                         void* stream);

/* kr_mlp_forward followed by kr_loss_rows_fwd_bwd (pred = NULL) over the Q = S*K rows of a training set in ONE
 * call - physics_train.py:250-259 / 345-352 from the MLP inputs to the loss and d loss / d out.  Where the fused
 * kernels serve the network the loss runs in the epilogue of the forward kernel (the MLP outputs never reach HBM);
 * otherwise the two kernels run back to back through `out` ([Q][32] scratch, always required).  Same meaning of
 * loss / dout / the option "mlp_grad_accumulate" as kr_loss_rows_fwd_bwd; ws as for kr_mlp_forward. */
int kr_mlp_forward_loss(kr_handle* h, int64_t S, int K, int n_layers, const int32_t* dims, const int32_t* acts,
                        const float* const* W, const float* const* b, const float* x, int in_pad, const float* base,
                        const float* target_rows, double denom, float* out, float* loss, float* dout, void* ws,
                        void* stream);

/* Full-state estimate from measured poses, knode_cosserat_realworld/estimate_state.py:158-242 (with compute_v_u
 * :48-95, compute_angular_velocities :97-123, compute_internal_forces_and_moments :126-156): data[T][7][N]
 * (positions, quaternions at the handle's N grid points), tensions[T][4] -> est[T][25][N], fp64, device pointers.
 * ws: kr_estimate_ws_bytes(T, N) bytes of device scratch.  Uses the handle's parameters (L, del_t, C, Bse, Bbt, ...). */
size_t kr_estimate_ws_bytes(int64_t T, int N);
int kr_estimate_state(kr_handle* h, int64_t T, const double* data, const double* tensions, double* est, void* ws,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KNODE_ROD_H */
