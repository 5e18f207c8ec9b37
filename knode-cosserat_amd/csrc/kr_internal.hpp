// kr_internal.hpp - host-side handle and launch declarations shared by the
// translation units of libknode_rod.so.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/knode_rod.h"
#include "rod_device.hpp"

namespace kr {

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what);
#define KR_HIP(expr)                                   \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) return kr::hip_fail(_e, #expr); \
  } while (0)

// Packed MLP on the device: per layer a transposed weight matrix
// Wt[in][out_pad] (out_pad = out rounded up to 16) and bias b[out_pad], in
// both precisions, so that for a fixed input unit the weights of 16 output
// units are contiguous and can be fetched with scalar loads.
template <typename T>
struct MlpDev {
  int n_layers;
  int dims[KR_MAX_LAYERS + 1];
  int out_pad[KR_MAX_LAYERS];
  int acts[KR_MAX_LAYERS];
  const T* Wt[KR_MAX_LAYERS];
  const T* b[KR_MAX_LAYERS];
  int max_dim;  // widest activation vector (incl. input and output)
  // matrix-core form (mlp_mfma.hpp): weights / biases pre-packed in MFMA fragment order
  const T* wfrag[KR_MAX_LAYERS];
  const T* bfrag[KR_MAX_LAYERS];
  int ksteps[KR_MAX_LAYERS];   // k-steps (of 4 inputs) per output tile
  int otiles[KR_MAX_LAYERS];   // 16-unit output tiles (hidden layers: a multiple of 4)
  int mfma_ok;                 // the network has a shape the matrix-core evaluator supports
  // Jacobian-vector-product chain of the multiple-shooting sweeps (mlp_jvp.hpp): bf16 A fragments
  // [tile][k-step of 32][lane][8], shared by both precisions
  const void* jfrag[KR_MAX_LAYERS];
  int jksteps[KR_MAX_LAYERS];
  // base chain of the same evaluator (v_mfma_f64_4x4x4_4b for both precisions): fp32 A fragments
  // [tile][k-group of 16 inputs][lane][4 k-steps], biases [tile][lane] in that instruction's D layout
  const float* wq[KR_MAX_LAYERS];
  const float* bq[KR_MAX_LAYERS];
  int kgroups[KR_MAX_LAYERS];
  int jvp_ok;                  // shape served by mlp_jvp.hpp (second hidden layer <= 64 (MJ_ACT_SLOTS - 1) units)
  // fp32 base chain of the same evaluator for in -> H1 <= 64 -> H2 <= 64 -> 25 (mlp_jvp_tile3f, v_mfma_f32_4x4x1_16B): lane l
  // owns row l of the weight matrix, [k-group of 4][lane][4] f32; biases per row [64]
  const float* w32[KR_MAX_LAYERS];
  const float* b32[KR_MAX_LAYERS];
  int f32_ok;
};

}  // namespace kr

struct kr_handle {
  int device = 0;
  kr_params params{};
  kr_derived derived{};
  kr::RodConst<float> cf{};
  kr::RodConst<double> cd{};
  // MLP
  kr::MlpDev<float> mlp_f{};
  kr::MlpDev<double> mlp_d{};
  struct kr_mlp_plan* mlp_plan = nullptr;  // packed buffers + gather plan of the current network shape (kr_api.hip)
  // lazily grown scratch (history fallback, MLP activation spill)
  void* ws = nullptr;
  void* pred_buf = nullptr;   // predictor images of the one-launch-per-step multiple-shooting path
  size_t pred_bytes = 0;
  int grad_accumulate = 0;    // kr_mlp_backward / kr_loss_rows_fwd_bwd add to dW, db, loss instead of zeroing them first
  int keep_predictor = 0;     // kr_simulate_batch resumes from / leaves behind the predictor image (option)
  int64_t pred_valid_B = 0;   // batch size the image in pred_buf was written for (0: none)
  int pred_valid_W = 1;       // ... and the wavefronts per rod of the kernel that wrote it
  int pred_valid_nn = 0;      // ... and whether the MLP was on (the image layout depends on the predictor's tap count)
  size_t ws_bytes = 0;
  int lds_limit = 160 * 1024;
  int ms_mode = -1;          // multiple-shooting step kernel: -1 auto (by batch size), 0 off, 1 forced
  // auto mode: multiple shooting when B <= this.  Measured (B = 2048 .. 8192, N = 100, fp64): 27.7 M rod-steps/s at
  // every batch size against 10.2 M for the 8-rods-per-wavefront kernel, so there is no limit by default
  int ms_batch_limit = 1 << 30;
  int predictor = 8;         // highest extrapolation order kr_simulate_batch may use (per-step launches: <= 2)
  int last_sim_path = 0;     // what the last kr_simulate_batch did: 0 one single-shooting launch per step,
                             // 1 one multiple-shooting launch per step, 2 one persistent launch for all steps
  int waves_per_rod = 0;     // per-step launches: wavefronts that share a rod (kr_msw_impl.hpp): 0 auto, 1, 2 or 4
  int last_waves_per_rod = 1;  // what the last step launch used
  void* loss_scratch = nullptr;  // per-workgroup loss partials of the fused forward + loss kernel
  // kr_train_epoch: the MFMA weight fragments in `frag_ws` mirror the flat parameter vector `frag_params` (the epoch's tail
  // kernel writes every updated parameter to both); another workspace, vector or network means packing afresh
  const void* frag_ws = nullptr;
  const void* frag_params = nullptr;
  uint64_t frag_net = 0;
  void* dbg = nullptr;       // diagnostic cycle-counter buffer (kr_debug_buffer)
  int fused_mlp = 1;         // training: fused MFMA forward/backward kernels (kr_mlp_fused.hip) when the shape allows
  int mfma_mlp = 1;          // evaluate the in-sweep MLP on the matrix cores when its shape allows
  int persistent = 1;        // kr_simulate_batch: run all steps in one launch when the multiple-shooting kernel applies
  int residual_test = 1;     // accept a storing sweep from its residual alone when the estimate is 256 x below the tolerance
  int overlap = 1;           // ... and overlap the verifying sweep of step t with the Jacobian sweep of step t + 1 (kr_mso_impl.hpp)
  int last_overlap = 0;      // the last kr_simulate_batch ran the overlapped kernel
  int msw_overlap = 1;       // several wavefronts per rod: the persistent kernel with overlapped steps (kr_mswo_impl.hpp)
  int nn_lowp_first = 1;     // fp64, MLP on, persistent one-wavefront kernel: first sweep of a three-sweep step on the fp32 base chain
  int nn_base_only_store = 1;  // fp64, MLP on, persistent one-wavefront kernel: storing sweeps without forward-difference columns
  void* resume_buf = nullptr;  // int32 per rod (SimArgs::resume)
  size_t resume_cap = 0;
  void* hist_ws = nullptr;     // history records [B][N][12] of the several-wavefront persistent kernel with the MLP on
  size_t hist_ws_cap = 0;
  // Stream ordering of the handle's scratch (ws, pred_buf, resume_buf, hist_ws, loss_scratch are shared by its calls):
  // a call on another stream than the previous one first waits for everything queued on that one (kr::order_stream)
  hipStream_t last_stream = nullptr;
  bool have_last_stream = false;
  hipEvent_t order_event = nullptr;
};

namespace kr {

int ensure_ws(kr_handle* h, size_t bytes);
// Makes work queued on `s` from here on run after everything the handle's previous calls queued on ANOTHER stream (an
// event on that stream's tail; nothing at all when the stream did not change).  Every entry point that launches calls it.
int order_stream(kr_handle* h, hipStream_t s);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) for a kernel that needs more than 48 KB of dynamic LDS, once per
// (host thread, device, kernel, size): the attribute belongs to the device's copy of the kernel, so the cache is keyed
// by the current device too.  The prepare_* functions go through the same cache, so the first timed launch finds it set.
int dyn_lds(const void* kern, size_t smem);

template <typename T>
inline const RodConst<T>& consts(kr_handle* h);
template <>
inline const RodConst<float>& consts<float>(kr_handle* h) { return h->cf; }
template <>
inline const RodConst<double>& consts<double>(kr_handle* h) { return h->cd; }
template <typename T>
inline const MlpDev<T>& mlpdev(kr_handle* h);
template <>
inline const MlpDev<float>& mlpdev<float>(kr_handle* h) { return h->mlp_f; }
template <>
inline const MlpDev<double>& mlpdev<double>(kr_handle* h) { return h->mlp_d; }

// kr_sim.hip
template <typename T>
int launch_init_straight(kr_handle* h, int64_t B, T* state, hipStream_t s);
template <typename T>
int launch_pack(kr_handle* h, int64_t B, const T* y_fm, const T* z_fm, T* state, hipStream_t s);
template <typename T>
int launch_unpack(kr_handle* h, int64_t B, const T* state, T* y_fm, T* z_fm, hipStream_t s);
template <typename T>
int launch_unpack50(kr_handle* h, int64_t B, const T* st, const T* m1, const T* m2, T* out, hipStream_t s);
template <typename T>
int launch_tip(kr_handle* h, int64_t B, const T* state, T* tip, hipStream_t s);

// doubles per rod of a predictor image (kr_ms_impl.hpp: MS_PRED_ROWS x 64 lanes)
constexpr size_t KR_PRED_IMG_DOUBLES = 24 * 64;

template <typename T>
struct StepArgs {
  int64_t B;
  const T* prev2;   // optional: state two steps back (quadratic predictor); may alias `next` (read first)
  const T* prev;
  const T* cur;
  T* next;
  T* G;             // [B][6] in/out (residual mode: in only)
  const T* tens;    // tensions of rod b at tens[b*tens_stride .. +4]
  int64_t tens_stride;
  T* r_out;         // residual mode: [B][6]
  T* tip;           // optional: tip[b*tip_stride .. +3]
  int64_t tip_stride;
  int32_t* status;  // optional: status[b*st_stride]
  int32_t* iters;
  int64_t st_stride;
  T* hist_ws;       // global history fallback [B][N][HS]
  T* act_ws;        // MLP activation spill (global) or nullptr
  T tol, tolA, fd_eps;
  T hc1, hc2;       // history = hc1*cur + hc2*prev (c1, c2 of BDF2, or 1, 0 for explicit history)
  int maxit;
  int mode;         // 0 = Newton step, 1 = single residual sweep
  int pred_order;   // initial guess: 0 = caller's G / current state, 1 = linear, 2 = quadratic extrapolation in time
  // multiple-shooting kernel inside kr_simulate_batch: image of the start-value predictor (MsPred, kr_ms_impl.hpp),
  // [B][MS_PRED_ROWS][64] doubles, read and rewritten by every launch; nullptr = extrapolate from the states
  double* pred = nullptr;
  int pred_reset = 0;     // first step of a simulate call: build the predictor from cur / prev
  int pred_has_prev = 0;  // ... prev is a real earlier state
  int pred_limit = 0;     // the handle's "predictor" option
  int residual_test = 1;  // option "residual_test": see kr_set_option (knode_rod.h)
  // residual mode, RK4 only: explicit midpoint histories [B][N][KR_SLOTS] (record j = between grid points j and j + 1);
  // nullptr = the linear interpolation knode.simulate forms (knode.py:80-81)
  const T* mid = nullptr;
};
template <typename T>
int launch_step(kr_handle* h, int scheme, int use_nn, const StepArgs<T>& a, hipStream_t s);
// wavefronts per rod the several-wavefront step kernel (kr_msw_impl.hpp) would use for this call; 0: not that kernel
template <typename T>
int step_waves_per_rod(kr_handle* h, int scheme, int use_nn, int64_t B, int mode);

// persistent multi-step form (kr_ms_impl.hpp)
template <typename T>
struct SimArgs {
  int64_t B, T_steps;
  T* states;            // [(T+1) or 3][B][N][KR_SLOTS]
  int64_t slot_elems;   // B*N*KR_SLOTS
  int ring;
  const T* prev_init;   // state before states[0] (nullable: y_prev = y, knode.py:65-66)
  const T* ctl;         // [B][T][4]
  T* G;                 // [B][6] in/out
  T* tip;               // [B][T][3] nullable
  int32_t* status;      // [B][T] nullable
  T tol, tolA, fd_eps, hc1, hc2;
  int maxit, predictor;
  unsigned long long* dbg;  // diagnostic builds (-DKR_MS_STAMPS): [B][24] cycle counters, else unused
  double* pred_io;          // predictor image [B][KR_PRED_IMG_DOUBLES]: saved at the end (nullable)
  int pred_load;            // ... and loaded at the start instead of building the predictor from the states
  // Two-launch form (kr_mso_impl.hpp): the overlapped kernel leaves, per rod, the first step it did NOT finish
  // (T_steps: all done); the one-wavefront persistent kernel launched behind it with the same arguments resumes
  // exactly there with its full fallback ladder.  nullptr: every rod starts at step 0.
  int32_t* resume = nullptr;
  int residual_test = 1;  // option "residual_test"
  T* hist_ws = nullptr;   // kr_msw_impl.hpp, MLP on: [B][N][12] history records (global memory instead of LDS)
  int nn_lowp = 1;        // option "nn_lowp_first" (MsSolveArgs::lowp_allowed)
  int nn_base_only = 1;   // option "nn_base_only_store" (MsSolveArgs::bo_allowed)
};

// returns 1 when the persistent form does not apply
template <typename T>
int launch_sim_persistent(kr_handle* h, int scheme, int use_nn, const SimArgs<T>& a, hipStream_t s);
// kr_mso_f32.hip / kr_mso_f64.hip: the overlapped persistent kernel (diagonal material matrices, Euler, MLP off);
// returns 1 when it does not serve the problem
template <typename T>
int launch_mso_sim(kr_handle* h, const SimArgs<T>& a, hipStream_t s);
int ensure_resume(kr_handle* h, int64_t B);
// kr_mswo_f32.hip / kr_mswo_f64.hip: the same overlap on W = 2 / 4 wavefronts per rod; 1: does not apply
template <typename T>
int launch_mswo_sim(kr_handle* h, int W, const SimArgs<T>& a, hipStream_t s);
template <typename T>
int prepare_mso_sim(kr_handle* h, int64_t B);
// kr_mswn_f32.hip / kr_mswn_f64.hip: several wavefronts per rod with the MLP on (persistent form; kr_msw_impl.hpp).
// nn_sim_waves_per_rod: wavefronts per rod kr_simulate_batch will use for this batch (0: the one-wavefront kernel);
// launch_msw_nn_sim returns 1 when it does not serve the problem
template <typename T>
int nn_sim_waves_per_rod(kr_handle* h, int scheme, int64_t B);
template <typename T>
int launch_msw_nn_sim(kr_handle* h, int W, const SimArgs<T>& a, hipStream_t s);
// ... and with the MLP off, for rods too long for the LDS form (N = 400); returns 1 when it does not apply either
template <typename T>
int launch_msw_gh_sim(kr_handle* h, int W, const SimArgs<T>& a, hipStream_t s);
int ensure_hist_ws(kr_handle* h, size_t bytes);
// kr_sim_f32.hip / kr_sim_f64.hip: the same for the one-wavefront persistent kernel that runs behind it
template <typename T>
int prepare_ms_sim(kr_handle* h);

// kr_mlp_fused.hip: fused fp32 MLP forward / backward for training
bool fused_mlp_supported(int n_layers, const int32_t* dims, const int32_t* acts, int in_pad);
size_t fused_ws_bytes(int n_layers, const int32_t* dims, int64_t Q);
// loss fused into the epilogue of the forward kernel (kr_mlp_forward_loss); all device pointers
struct FusedLoss {
  const float* base;         // [Q][25] parameter-free part of the prediction
  const float* target_rows;  // [Q][25]
  float* dout;               // [Q][32] gradient with respect to the MLP outputs
  float* loss;               // the four-term loss is ADDED here
  void* scratch;             // 4096 floats of the handle: per-workgroup partials
  float ds, inv_denom;
  int K;
};
// do_pack = false: the fragments in ws are current (kr_train_epoch keeps them so); n_partials != nullptr: the loss
// partials stay in fl->scratch, *n_partials of them, for the caller's next kernel
int fused_mlp_forward(int64_t Q, int n_layers, const int32_t* dims, const int32_t* acts, const float* const* W,
                      const float* const* b, const float* x, float* out, void* ws, hipStream_t s,
                      const FusedLoss* fl = nullptr, bool do_pack = true, int* n_partials = nullptr);
// leave != nullptr: no reduction launch - the slabs of partial gradients are described there (dW, db unused)
struct FusedSlabs {
  const float* slab;
  int nslab, P, nparams;
  int poff[6];
};
int fused_mlp_backward(int64_t Q, int n_layers, const int32_t* dims, const int32_t* acts, const float* const* W,
                       const float* x, const float* dout, void* ws, float* const* dW, float* const* db, hipStream_t s,
                       FusedSlabs* leave = nullptr);
// kr_train_epoch: flat parameter / gradient / moment vectors in nn.Linear order (W1, b1, W2, b2, ...), g with one
// trailing loss slot
struct FusedEpoch {
  int64_t Q;
  int K, n_layers;
  const int32_t* dims;
  const int32_t* acts;
  float *p, *g, *m, *v;
  const float* lower;
  double* sched;
  const float *x, *base, *target_rows;
  float* dout;
  void* ws;
  void* loss_scratch;
  float ds, inv_denom;
  double beta1, beta2, eps, weight_decay, factor, threshold, min_lr;
  int patience;
  int64_t step;
  float* loss_log;
  int phase;
  bool pack;
};
int fused_train_epoch(const FusedEpoch& E, hipStream_t s);
void fused_set_debug(void* dev_u64x48);  // diagnostic build: where the fused kernels add their phase stamps

// kr_ode.hip
template <typename T>
int launch_mlp_eval(kr_handle* h, int64_t Q, const T* x, T* out, hipStream_t s);
template <typename T>
int launch_ode_batch(kr_handle* h, int64_t Q, const T* y, const T* yh, const T* zh, const T* tf, T* dys, T* z,
                     int use_nn, hipStream_t s);

}  // namespace kr
