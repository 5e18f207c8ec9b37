// kr_api.hip - the extern "C" surface of libknode_rod.so (include/knode_rod.h):
// handle management, parameter derivation (reference cosserat_ode.py:58-78),
// presets (knode.py:6-53), MLP packing and the dtype dispatch of every call.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <utility>

#include "kr_internal.hpp"

namespace kr {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int hip_fail(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return KR_E_HIP;
}

int ensure_ws(kr_handle* h, size_t bytes) {
  if (bytes <= h->ws_bytes) return KR_OK;
  if (h->ws) {
    KR_HIP(hipDeviceSynchronize());
    KR_HIP(hipFree(h->ws));
    h->ws = nullptr;
    h->ws_bytes = 0;
  }
  size_t want = bytes + bytes / 4;
  KR_HIP(hipMalloc(&h->ws, want));
  h->ws_bytes = want;
  return KR_OK;
}

int order_stream(kr_handle* h, hipStream_t s) {
  int dev = -1;
  if (hipGetDevice(&dev) == hipSuccess && dev != h->device) {
    set_error("the current HIP device (" + std::to_string(dev) + ") is not the one the handle was created on (" +
              std::to_string(h->device) + "): call hipSetDevice first");
    return KR_E_ARG;
  }
  if (h->have_last_stream && h->last_stream != s) {
    if (!h->order_event) KR_HIP(hipEventCreateWithFlags(&h->order_event, hipEventDisableTiming));
    if (hipEventRecord(h->order_event, h->last_stream) == hipSuccess) {
      KR_HIP(hipStreamWaitEvent(s, h->order_event, 0));
    } else {  // the previous stream is gone (destroyed by its owner): whatever ran on it has to be complete the hard way
      (void)hipGetLastError();
      KR_HIP(hipDeviceSynchronize());
    }
  }
  h->last_stream = s;
  h->have_last_stream = true;
  return KR_OK;
}

int dyn_lds(const void* kern, size_t smem) {
  if (smem <= 48 * 1024) return KR_OK;
  int dev = 0;
  KR_HIP(hipGetDevice(&dev));
  static thread_local std::map<std::pair<const void*, int>, size_t> configured;
  size_t& cur = configured[{kern, dev}];
  if (smem > cur) {
    KR_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    cur = smem;
  }
  return KR_OK;
}

int ensure_resume(kr_handle* h, int64_t B) {
  const size_t bytes = (size_t)B * sizeof(int32_t);
  if (bytes <= h->resume_cap) return KR_OK;
  if (h->resume_buf) {
    KR_HIP(hipDeviceSynchronize());
    KR_HIP(hipFree(h->resume_buf));
    h->resume_buf = nullptr;
    h->resume_cap = 0;
  }
  KR_HIP(hipMalloc(&h->resume_buf, bytes + bytes / 4));
  h->resume_cap = bytes + bytes / 4;
  return KR_OK;
}

int ensure_hist_ws(kr_handle* h, size_t bytes) {
  if (bytes <= h->hist_ws_cap) return KR_OK;
  if (h->hist_ws) {
    KR_HIP(hipDeviceSynchronize());
    KR_HIP(hipFree(h->hist_ws));
    h->hist_ws = nullptr;
    h->hist_ws_cap = 0;
  }
  KR_HIP(hipMalloc(&h->hist_ws, bytes + bytes / 4));
  h->hist_ws_cap = bytes + bytes / 4;
  return KR_OK;
}

static void mat3_diag(double* m, double a, double b, double c) {
  for (int i = 0; i < 9; ++i) m[i] = 0;
  m[0] = a; m[4] = b; m[8] = c;
}
static bool inv3(const double* a, double* o) {
  const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  if (det == 0.0 || !std::isfinite(det)) return false;
  const double id = 1.0 / det;
  o[0] = c00 * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
  o[3] = c01 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
  o[6] = c02 * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
  return true;
}
static bool is_diag(const double* m) {
  return m[1] == 0 && m[2] == 0 && m[3] == 0 && m[5] == 0 && m[6] == 0 && m[7] == 0;
}

// cosserat_ode.py:58-78
static int derive(const kr_params& p, kr_derived& d) {
  if (p.N < 2) { set_error("N must be >= 2"); return KR_E_ARG; }
  if (!(p.del_t > 0) || !(p.L > 0) || !(p.r > 0)) { set_error("L, r, del_t must be positive"); return KR_E_ARG; }
  const double pi = 3.14159265358979323846;
  d.A = pi * p.r * p.r;
  d.G = p.E / (2 * (1 + 0.3));
  d.ds = p.L / (p.N - 1);
  const double r4 = p.r * p.r * p.r * p.r;
  mat3_diag(d.J, pi * r4 / 4, pi * r4 / 4, pi * r4 / 2);
  mat3_diag(d.Kse, d.G * d.A, d.G * d.A, p.E * d.A);
  mat3_diag(d.Kbt, p.E * d.J[0], p.E * d.J[4], d.G * d.J[8]);
  d.c0 = 1.5 / p.del_t;
  d.c1 = -2.0 / p.del_t;
  d.c2 = 0.5 / p.del_t;
  double t[9];
  for (int i = 0; i < 9; ++i) t[i] = d.Kse[i] + d.c0 * p.Bse[i];
  if (!inv3(t, d.Kse_plus_c0_Bse_inv)) { set_error("Kse + c0*Bse is singular"); return KR_E_ARG; }
  for (int i = 0; i < 9; ++i) t[i] = d.Kbt[i] + d.c0 * p.Bbt[i];
  if (!inv3(t, d.Kbt_plus_c0_Bbt_inv)) { set_error("Kbt + c0*Bbt is singular"); return KR_E_ARG; }
  for (int i = 0; i < 3; ++i)
    d.Kse_vstar[i] = d.Kse[3 * i] * p.vstar[0] + d.Kse[3 * i + 1] * p.vstar[1] + d.Kse[3 * i + 2] * p.vstar[2];
  d.rhoA = p.rho * d.A;
  for (int i = 0; i < 3; ++i) d.rhoAg[i] = d.rhoA * p.g[i];
  for (int i = 0; i < 9; ++i) d.rhoJ[i] = p.rho * d.J[i];
  return KR_OK;
}

template <typename T>
static void fill_consts(const kr_params& p, const kr_derived& d, RodConst<T>& c) {
  c.c0 = (T)d.c0; c.c1 = (T)d.c1; c.c2 = (T)d.c2; c.ds = (T)d.ds; c.rhoA = (T)d.rhoA;
  for (int i = 0; i < 9; ++i) {
    c.Ksei[i] = (T)d.Kse_plus_c0_Bse_inv[i];
    c.Kbti[i] = (T)d.Kbt_plus_c0_Bbt_inv[i];
    c.Bse[i] = (T)p.Bse[i];
    c.Bbt[i] = (T)p.Bbt[i];
    c.rhoJ[i] = (T)d.rhoJ[i];
  }
  for (int i = 0; i < 3; ++i) {
    c.Kse_vstar[i] = (T)d.Kse_vstar[i];
    c.rhoAg[i] = (T)d.rhoAg[i];
    c.C[i] = (T)p.C[i];
    c.Ftip[i] = (T)p.F_tip[i];
    c.Mtip[i] = (T)p.M_tip[i];
    c.p0[i] = (T)p.p0[i];
    c.q0[i] = (T)p.q0[i];
    c.w0[i] = (T)p.w0[i];
  }
  for (int i = 0; i < 4; ++i) c.h0[i] = (T)p.h0[i];
  for (int i = 0; i < 12; ++i) c.tdirs[i] = (T)p.tendon_dirs[i];
  c.N = p.N;
  c.diag = is_diag(d.Kse_plus_c0_Bse_inv) && is_diag(d.Kbt_plus_c0_Bbt_inv) && is_diag(p.Bse) && is_diag(p.Bbt) &&
           is_diag(d.rhoJ);
}

void free_mlp_plan(kr_handle* h);  // (below, beside kr_set_mlp)

}  // namespace kr

using namespace kr;

#define KR_CHECK_H(h)                      \
  if (!(h)) {                              \
    set_error("null handle");              \
    return KR_E_ARG;                       \
  }
#define KR_CHECK_PTR(p)                            \
  if (!(p)) {                                      \
    set_error("null pointer argument: " #p);       \
    return KR_E_ARG;                               \
  }
#define KR_CHECK_DTYPE(dt)                         \
  if ((dt) != KR_F32 && (dt) != KR_F64) {          \
    set_error("dtype must be KR_F32 or KR_F64");   \
    return KR_E_ARG;                               \
  }

template <typename T>
static StepArgs<T> make_args(const kr_handle* h, int64_t B, const void* prev, const void* cur, void* next, void* G, const void* tens,
                             int64_t tens_stride, double tol, int maxit) {
  StepArgs<T> a{};
  a.B = B;
  a.prev = (const T*)prev; a.cur = (const T*)cur; a.next = (T*)next;
  a.G = (T*)G;
  a.tens = (const T*)tens; a.tens_stride = tens_stride;
  const bool f64 = sizeof(T) == 8;
  if (!(tol > 0)) tol = f64 ? 1e-8 : 1e-5;  // fp64: the class of hybrd's xtol = 1.49e-8 the reference runs with
  a.tol = (T)tol;
  a.tolA = (T)std::sqrt(tol);
  a.fd_eps = f64 ? (T)1e-7 : (T)1e-3;
  a.maxit = maxit > 0 ? maxit : 30;
  a.mode = 0;
  a.prev2 = nullptr;
  a.pred_order = 0;
  a.hc1 = (T)h->derived.c1; a.hc2 = (T)h->derived.c2;
  a.st_stride = 1;
  a.tip_stride = 3;
  a.residual_test = h->residual_test;
  return a;
}

static int ensure_pred(kr_handle* h, int64_t B) {
  const size_t need = (size_t)B * KR_PRED_IMG_DOUBLES * sizeof(double);
  if (need > h->pred_bytes) {
    if (h->pred_buf) { KR_HIP(hipDeviceSynchronize()); KR_HIP(hipFree(h->pred_buf)); h->pred_buf = nullptr; h->pred_bytes = 0; }
    KR_HIP(hipMalloc(&h->pred_buf, need));
    h->pred_bytes = need;
    h->pred_valid_B = 0;
  }
  return KR_OK;
}

template <typename T>
static int simulate_impl(kr_handle* h, int64_t B, int64_t T_steps, int scheme, const void* ctl, void* states, int ring,
                         void* G, void* tip, double tol, int maxit, int32_t* status, int use_nn,
                         const void* prev_init, hipStream_t s) {
  const size_t slot = (size_t)B * h->params.N * KR_SLOTS;
  T* base = (T*)states;
  // (one predictor image per wavefront of a rod)
  const int wpr = use_nn ? nn_sim_waves_per_rod<T>(h, scheme, B) : step_waves_per_rod<T>(h, scheme, use_nn, B, 0);
  const int img_w = wpr ? wpr : 1;
  {
    // one launch for all steps when a multiple-shooting kernel with a persistent form applies
    auto a0 = make_args<T>(h, B, nullptr, nullptr, nullptr, G, ctl, 4, tol, maxit);
    SimArgs<T> sa{};
    sa.B = B; sa.T_steps = T_steps; sa.states = base; sa.slot_elems = (int64_t)slot; sa.ring = ring;
    sa.prev_init = (const T*)prev_init; sa.ctl = (const T*)ctl; sa.G = (T*)G; sa.tip = (T*)tip; sa.status = status;
    sa.tol = a0.tol; sa.tolA = a0.tolA; sa.fd_eps = a0.fd_eps; sa.hc1 = a0.hc1; sa.hc2 = a0.hc2;
    sa.maxit = a0.maxit; sa.predictor = h->predictor; sa.residual_test = h->residual_test; sa.nn_lowp = h->nn_lowp_first; sa.nn_base_only = h->nn_base_only_store;
    sa.dbg = static_cast<unsigned long long*>(h->dbg);
    sa.pred_io = nullptr; sa.pred_load = 0;
    if (h->keep_predictor && (size_t)B * img_w * KR_PRED_IMG_DOUBLES * sizeof(double) <= ((size_t)1 << 30)) {
      int rcp = ensure_pred(h, B * img_w);
      if (rcp) return rcp;
      sa.pred_io = static_cast<double*>(h->pred_buf);
      sa.pred_load = h->pred_valid_B == B && h->pred_valid_W == img_w && h->pred_valid_nn == (use_nn ? 1 : 0);
    }
    const int rc = launch_sim_persistent<T>(h, scheme, use_nn, sa, s);
    if (rc != 1) {
      h->last_sim_path = 2;
      if (rc == KR_OK && sa.pred_io) { h->pred_valid_B = B; h->pred_valid_W = img_w; h->pred_valid_nn = use_nn ? 1 : 0; }
      return rc;
    }
  }
  h->last_sim_path = 0;
  // one launch per step: the multiple-shooting kernel carries its start-value predictor from launch to launch
  // through an image in HBM (12 KB per rod; skipped for batches that would need more than 1 GB of it)
  double* pred = nullptr;
  if (h->predictor > 2 && h->ms_mode != 0 && (h->ms_mode == 1 || B <= (int64_t)h->ms_batch_limit)) {
    const size_t need = (size_t)B * img_w * KR_PRED_IMG_DOUBLES * sizeof(double);
    if (need <= ((size_t)1 << 30)) {
      int rcp = ensure_pred(h, B * img_w);
      if (rcp) return rcp;
      pred = static_cast<double*>(h->pred_buf);
    }
  }
  const bool resume = pred && h->keep_predictor && h->pred_valid_B == B && h->pred_valid_W == img_w &&
                      h->pred_valid_nn == (use_nn ? 1 : 0);
  for (int64_t t = 0; t < T_steps; ++t) {
    // knode.py:65-66,76-77: before the first step y_prev = y (unless the caller hands over the state before)
    const int64_t ic = ring ? t % 3 : t;
    const int64_t in = ring ? (t + 1) % 3 : t + 1;
    const T* pprev = t == 0 ? (prev_init ? (const T*)prev_init : base + ic * slot)
                            : base + (ring ? (t + 2) % 3 : t - 1) * slot;
    auto a = make_args<T>(h, B, pprev, base + ic * slot, base + in * slot, G, (const T*)ctl + t * 4,
                          T_steps * 4, tol, maxit);
    // time extrapolation of the initial guess from the states already computed (order <= h->predictor)
    int order = t == 0 ? (prev_init ? 1 : 0) : (t == 1 ? (prev_init ? 2 : 1) : 2);
    if (order > h->predictor) order = h->predictor;  // (orders 3..5: persistent kernel only)
    if (order == 2) a.prev2 = t == 1 ? (const T*)prev_init : base + (ring ? (t + 1) % 3 : t - 2) * slot;
    a.pred_order = order;
    a.pred = pred; a.pred_reset = t == 0 && !resume; a.pred_has_prev = prev_init != nullptr; a.pred_limit = h->predictor;
    if (tip) { a.tip = (T*)tip + t * 3; a.tip_stride = T_steps * 3; }
    if (status) { a.status = status + t; a.st_stride = T_steps; }
    if (h->dbg) { a.iters = static_cast<int32_t*>(h->dbg) + t; a.st_stride = T_steps; }  // diagnostics: sweeps per rod and step, [B][T] int32
    int rc = launch_step<T>(h, scheme, use_nn, a, s);
    if (rc) return rc;
  }
  // the image is current only if the multiple-shooting kernel took the steps (launch_step decides)
  if (pred) h->pred_valid_B = (T_steps > 0 && h->last_sim_path == 1) ? B : (T_steps > 0 ? 0 : h->pred_valid_B);
  if (pred && T_steps > 0) { h->pred_valid_W = img_w; h->pred_valid_nn = use_nn ? 1 : 0; }
  return KR_OK;
}

extern "C" {

const char* kr_last_error(void) { return g_err.c_str(); }
int kr_version(void) { return 100; }

int kr_default_params(kr_params* o) {
  KR_CHECK_PTR(o);
  std::memset(o, 0, sizeof(*o));
  o->L = 0.4; o->N = 10; o->E = 109e9; o->r = 0.0012; o->rho = 8000.0;
  o->vstar[2] = 1.0;
  o->g[2] = -9.81;
  o->Bbt[0] = o->Bbt[4] = o->Bbt[8] = 3e-2;
  o->C[0] = o->C[1] = o->C[2] = 1e-4;
  o->del_t = 0.005;
  const double pi = 3.14159265358979323846;
  const double th = pi / 4;  // pi / n_tendons
  for (int k = 0; k < 4; ++k) {
    o->tendon_dirs[3 * k + 0] = std::cos(th + k * pi / 2);
    o->tendon_dirs[3 * k + 1] = std::sin(th + k * pi / 2);
    o->tendon_dirs[3 * k + 2] = 0.0;
  }
  o->h0[0] = 1.0;
  return KR_OK;
}

int kr_apply_preset(kr_params* p, const char* mod) {
  KR_CHECK_PTR(p);
  p->del_t = 0.05; p->L = 0.635; p->r = 0.003175; p->rho = 1411.6751; p->E = 2.757903e9;
  double bbt = 3e-2;
  const std::string m = mod ? mod : "";
  if (m.empty() || m == "None") {
  } else if (m == "noair") {
    p->C[0] = p->C[1] = p->C[2] = 0;
  } else if (m == "nsw") {
    p->g[0] = p->g[1] = p->g[2] = 0;
  } else if (m == "short") {
    p->L = 0.4;
  } else if (m == "damping") {
    bbt = 0.2;
  } else if (m == "dampstiff") {
    bbt = 0.2; p->E = 10e9;
  } else if (m == "lengthstiff") {
    p->L = 0.4; p->E = 10e9;
  } else if (m == "youngs") {
    p->E = 10e9;
  } else {
    set_error("Unknown mod " + m);
    return KR_E_ARG;
  }
  for (int i = 0; i < 9; ++i) p->Bbt[i] = 0;
  p->Bbt[0] = p->Bbt[4] = p->Bbt[8] = bbt;
  return KR_OK;
}

int kr_apply_preset_original(kr_params* p, const char* mod) {
  KR_CHECK_PTR(p);
  p->del_t = 0.005; p->L = 0.4; p->E = 209e9; p->r = 0.0012; p->rho = 8000.0;
  double bbt = 5e-4;
  const std::string m = mod ? mod : "";
  if (m.empty() || m == "None") {
  } else if (m == "nsw") {
    p->g[0] = p->g[1] = p->g[2] = 0;
  } else if (m == "short") {
    p->L = 0.3;
  } else if (m == "damping") {
    bbt = 9e-4;
  } else if (m == "diameter") {
    p->r = 0.002;
  } else if (m == "youngs") {
    p->E = 109e9;
  } else if (m == "dampstiff") {
    bbt = 3e-2; p->E = 109e9;
  } else if (m == "lengthstiff") {
    p->L = 0.3; p->E = 109e9;
  } else {
    set_error("Unknown mod " + m);
    return KR_E_ARG;
  }
  for (int i = 0; i < 9; ++i) p->Bbt[i] = 0;
  p->Bbt[0] = p->Bbt[4] = p->Bbt[8] = bbt;
  return KR_OK;
}

int kr_set_params(kr_handle* h, const kr_params* p) {
  KR_CHECK_H(h);
  KR_CHECK_PTR(p);
  kr_derived d{};
  int rc = derive(*p, d);
  if (rc) return rc;
  h->params = *p;
  h->derived = d;
  fill_consts(*p, d, h->cf);
  fill_consts(*p, d, h->cd);
  return KR_OK;
}

int kr_create(const kr_params* p, int device, kr_handle** out) {
  KR_CHECK_PTR(p);
  KR_CHECK_PTR(out);
  int ndev = 0;
  KR_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) {
    set_error("no such HIP device");
    return KR_E_ARG;
  }
  KR_HIP(hipSetDevice(device));
  kr_handle* h = new kr_handle();
  h->device = device;
  int rc = kr_set_params(h, p);
  if (rc) {
    delete h;
    return rc;
  }
  int lds = 0;
  if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0)
    h->lds_limit = lds;
  if (const char* e = std::getenv("KR_MS_MODE")) h->ms_mode = std::atoi(e);
  if (const char* e = std::getenv("KR_WAVES_PER_ROD")) {
    const int w = std::atoi(e);
    if (w == 0 || w == 1 || w == 2 || w == 4) h->waves_per_rod = w;
  }
  if (const char* e = std::getenv("KR_PREDICTOR")) h->predictor = std::atoi(e);
  if (const char* e = std::getenv("KR_PERSISTENT")) h->persistent = std::atoi(e) ? 1 : 0;
  if (const char* e = std::getenv("KR_OVERLAP")) h->overlap = std::atoi(e) ? 1 : 0;
  if (const char* e = std::getenv("KR_MSW_OVERLAP")) h->msw_overlap = std::atoi(e) ? 1 : 0;
  if (const char* e = std::getenv("KR_RESIDUAL_TEST")) h->residual_test = std::atoi(e) ? 1 : 0;
  if (const char* e = std::getenv("KR_NN_LOWP_FIRST")) h->nn_lowp_first = std::atoi(e) ? 1 : 0;
  if (const char* e = std::getenv("KR_NN_BASE_ONLY_STORE")) h->nn_base_only_store = std::atoi(e) ? 1 : 0;
  if (const char* e = std::getenv("KR_MFMA_MLP")) h->mfma_mlp = std::atoi(e) ? 1 : 0;
  if (const char* e = std::getenv("KR_FUSED_MLP")) h->fused_mlp = std::atoi(e) ? 1 : 0;
  *out = h;
  return KR_OK;
}

int kr_set_option(kr_handle* h, const char* name, int value) {
  KR_CHECK_H(h);
  KR_CHECK_PTR(name);
  const std::string n = name;
  if (n == "ms_mode") {
    if (value < -1 || value > 1) { set_error("ms_mode must be -1, 0 or 1"); return KR_E_ARG; }
    h->ms_mode = value;
  } else if (n == "ms_batch_limit") {
    if (value < 0) { set_error("ms_batch_limit must be >= 0"); return KR_E_ARG; }
    h->ms_batch_limit = value;
  } else if (n == "persistent") {
    h->persistent = value ? 1 : 0;
  } else if (n == "waves_per_rod") {
    if (value != 0 && value != 1 && value != 2 && value != 4) { set_error("waves_per_rod must be 0 (auto), 1, 2 or 4"); return KR_E_ARG; }
    h->waves_per_rod = value;
  } else if (n == "mlp_grad_accumulate") {
    h->grad_accumulate = value ? 1 : 0;
  } else if (n == "keep_predictor") {
    h->keep_predictor = value ? 1 : 0;
    if (!value) h->pred_valid_B = 0;
  } else if (n == "fused_mlp") {
    h->fused_mlp = value ? 1 : 0;
  } else if (n == "mfma_mlp") {
    h->mfma_mlp = value ? 1 : 0;  // takes effect at the next kr_set_mlp
  } else if (n == "predictor") {
    if (value < 0 || value > 8) { set_error("predictor must be 0 .. 8"); return KR_E_ARG; }
    h->predictor = value;
  } else if (n == "overlap") {
    h->overlap = value ? 1 : 0;
  } else if (n == "msw_overlap") {
    h->msw_overlap = value ? 1 : 0;
  } else if (n == "residual_test") {
    h->residual_test = value ? 1 : 0;
  } else if (n == "nn_base_only_store") {
    h->nn_base_only_store = value ? 1 : 0;
  } else if (n == "nn_lowp_first") {
    h->nn_lowp_first = value ? 1 : 0;
  } else {
    set_error("unknown option " + n);
    return KR_E_ARG;
  }
  return KR_OK;
}

int kr_get_option(kr_handle* h, const char* name, int* value) {
  KR_CHECK_H(h);
  KR_CHECK_PTR(name);
  KR_CHECK_PTR(value);
  const std::string n = name;
  if (n == "ms_mode") *value = h->ms_mode;
  else if (n == "ms_batch_limit") *value = h->ms_batch_limit;
  else if (n == "persistent") *value = h->persistent;
  else if (n == "keep_predictor") *value = h->keep_predictor;
  else if (n == "mlp_grad_accumulate") *value = h->grad_accumulate;
  else if (n == "mfma_mlp") *value = h->mfma_mlp;
  else if (n == "fused_mlp") *value = h->fused_mlp;
  else if (n == "predictor") *value = h->predictor;
  else if (n == "last_sim_path") *value = h->last_sim_path;
  else if (n == "waves_per_rod") *value = h->waves_per_rod;
  else if (n == "last_waves_per_rod") *value = h->last_waves_per_rod;
  else if (n == "overlap") *value = h->overlap;
  else if (n == "msw_overlap") *value = h->msw_overlap;
  else if (n == "residual_test") *value = h->residual_test;
  else if (n == "nn_lowp_first") *value = h->nn_lowp_first;
  else if (n == "nn_base_only_store") *value = h->nn_base_only_store;
  else if (n == "last_overlap") *value = h->last_overlap;
  else {
    set_error("unknown option " + n);
    return KR_E_ARG;
  }
  return KR_OK;
}

int kr_debug_buffer(kr_handle* h, void* dev_ptr) {
  KR_CHECK_H(h);
  h->dbg = dev_ptr;
  return KR_OK;
}

int kr_destroy(kr_handle* h) {
  if (!h) return KR_OK;
  free_mlp_plan(h);
  if (h->ws) (void)hipFree(h->ws);
  if (h->pred_buf) (void)hipFree(h->pred_buf);
  if (h->resume_buf) (void)hipFree(h->resume_buf);
  if (h->hist_ws) (void)hipFree(h->hist_ws);
  if (h->loss_scratch) (void)hipFree(h->loss_scratch);
  if (h->order_event) (void)hipEventDestroy(h->order_event);
  delete h;
  return KR_OK;
}

int kr_get_derived(const kr_handle* h, kr_derived* out) {
  KR_CHECK_H(h);
  KR_CHECK_PTR(out);
  *out = h->derived;
  return KR_OK;
}

int kr_derive(const kr_params* p, kr_derived* out) {
  KR_CHECK_PTR(p);
  KR_CHECK_PTR(out);
  return derive(*p, *out);
}

int kr_mlp_eval_batch(kr_handle* h, int64_t Q, const void* x, void* out, int dtype, void* stream) {
  KR_CHECK_H(h);
  KR_CHECK_DTYPE(dtype);
  if (Q < 0) { set_error("Q < 0"); return KR_E_ARG; }
  if (Q == 0) return KR_OK;
  KR_CHECK_PTR(x); KR_CHECK_PTR(out);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc_order_ = order_stream(h, s)) return rc_order_;
  return dtype == KR_F32 ? launch_mlp_eval<float>(h, Q, (const float*)x, (float*)out, s)
                         : launch_mlp_eval<double>(h, Q, (const double*)x, (double*)out, s);
}

}  // extern "C"

// kr_set_mlp packs ON THE DEVICE: per network SHAPE the host builds, once, a gather plan - for every element of every
// packed buffer the (source array, offset) it comes from - and one launch of mlp_pack_kernel fills all of them from the
// caller's weights.  The same shape again (evaluate() with live weights every 50 epochs, physics_train.py:136-167) costs
// that one launch: no device-to-host copy, no hipMalloc, no host synchronisation.
namespace kr {
constexpr int PACK_MAX_JOBS = 112;  // 4 buffers per layer for KR_MAX_LAYERS layers + 9 per layer of the matrix-core forms (<= 3)
constexpr uint32_t PACK_ZERO = 0xFFFFFFFFu;
enum { PACK_F32 = 0, PACK_F64 = 1, PACK_BF16 = 2 };
struct PackJobs {
  int n;
  uint32_t start[PACK_MAX_JOBS + 1];  // first element of job j in the concatenated index table
  void* dst[PACK_MAX_JOBS];
  uint8_t type[PACK_MAX_JOBS];
  const float* src[2 * KR_MAX_LAYERS];  // W[0], b[0], W[1], b[1], ...
};
__global__ __launch_bounds__(256) void mlp_pack_kernel(const PackJobs J, const uint32_t* __restrict__ idx) {
  const uint32_t total = J.start[J.n];
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    int j = 0;
    while (i >= J.start[j + 1]) ++j;
    const uint32_t e = i - J.start[j], code = idx[i];
    const float v = code == PACK_ZERO ? 0.f : J.src[code >> 27][code & 0x07FFFFFFu];
    if (J.type[j] == PACK_F32) static_cast<float*>(J.dst[j])[e] = v;
    else if (J.type[j] == PACK_F64) static_cast<double*>(J.dst[j])[e] = (double)v;
    else {  // bf16, round to nearest even on the bits (what the host packing of earlier rounds did)
      const uint32_t u = __float_as_uint(v);
      static_cast<uint16_t*>(J.dst[j])[e] = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    }
  }
}
}  // namespace kr

struct kr_mlp_plan {
  uint64_t key = 0;
  kr::MlpDev<float> mf{};
  kr::MlpDev<double> md{};
  kr::PackJobs jobs{};
  uint32_t* idx = nullptr;    // device: source code of every destination element
  unsigned char* arena = nullptr;  // device: every packed buffer
  float* staging = nullptr;   // device copy of host-side sources (src_on_device = 0)
  size_t src_off[2 * KR_MAX_LAYERS + 1] = {};
};

namespace kr {
void free_mlp_plan(kr_handle* h) {
  kr_mlp_plan* P = h->mlp_plan;
  if (!P) return;
  if (P->idx) (void)hipFree(P->idx);
  if (P->arena) (void)hipFree(P->arena);
  if (P->staging) (void)hipFree(P->staging);
  delete P;
  h->mlp_plan = nullptr;
}

// Builds the plan of one shape.  The layouts are those of mlp_lane.hpp (Wt, b), mlp_mfma.hpp (wfrag, bfrag), mlp_jvp.hpp
// (wq, bq, jfrag, w32, b32); every buffer is described by the source code of each of its elements.
static int build_mlp_plan(kr_handle* h, int n_layers, const int32_t* dims, const int32_t* acts, kr_mlp_plan& P) {
  MlpDev<float>& mf = P.mf;
  MlpDev<double>& md = P.md;
  mf = MlpDev<float>{};
  md = MlpDev<double>{};
  mf.n_layers = md.n_layers = n_layers;
  struct Job { std::vector<uint32_t> code; int type; const void** t0; const void** t1; };
  std::vector<Job> jobs;
  auto add = [&](std::vector<uint32_t>&& code, int type, const void** t0, const void** t1 = nullptr) {
    jobs.push_back(Job{std::move(code), type, t0, t1});
  };
  int maxd = dims[0];
  for (int k = 0; k <= n_layers; ++k) mf.dims[k] = md.dims[k] = dims[k];
  P.src_off[0] = 0;
  for (int k = 0; k < n_layers; ++k) {
    P.src_off[2 * k + 1] = P.src_off[2 * k] + (size_t)dims[k] * dims[k + 1];
    P.src_off[2 * k + 2] = P.src_off[2 * k + 1] + (size_t)dims[k + 1];
    if ((size_t)dims[k] * dims[k + 1] >= (1u << 27)) { set_error("layer too large"); return KR_E_ARG; }
  }
  auto wc = [](int k, size_t off) { return ((uint32_t)(2 * k) << 27) | (uint32_t)off; };
  auto bc = [](int k, size_t off) { return ((uint32_t)(2 * k + 1) << 27) | (uint32_t)off; };
  for (int k = 0; k < n_layers; ++k) {
    const int in = dims[k], out = dims[k + 1];
    const int opad = (out + 15) / 16 * 16;
    if (opad > maxd) maxd = opad;
    mf.out_pad[k] = md.out_pad[k] = opad;
    mf.acts[k] = md.acts[k] = acts[k];
    std::vector<uint32_t> wt((size_t)in * opad, PACK_ZERO), bb(opad, PACK_ZERO);
    for (int o = 0; o < out; ++o) {
      bb[o] = bc(k, o);
      for (int i = 0; i < in; ++i) wt[(size_t)i * opad + o] = wc(k, (size_t)o * in + i);
    }
    add(std::vector<uint32_t>(wt), PACK_F32, (const void**)&mf.Wt[k]);
    add(std::vector<uint32_t>(bb), PACK_F32, (const void**)&mf.b[k]);
    add(std::move(wt), PACK_F64, (const void**)&md.Wt[k]);
    add(std::move(bb), PACK_F64, (const void**)&md.b[k]);
  }
  mf.max_dim = md.max_dim = maxd;
  // ---- matrix-core form (mlp_mfma.hpp) --------------------------------------------------------
  const bool shape_ok = (n_layers == 2 || n_layers == 3) && dims[0] == 28 && (n_layers == 2 || dims[1] <= 64) &&
                        (n_layers == 2 || acts[0] == acts[1]) && acts[n_layers - 1] == KR_ACT_NONE;
  mf.mfma_ok = md.mfma_ok = 0;
  if (shape_ok) {
    int prev_tiles = 0;
    for (int k = 0; k < n_layers; ++k) {
      const int in = dims[k], out = dims[k + 1];
      const bool last = (k == n_layers - 1);
      const int tiles = last ? 2 : ((out + 63) / 64) * 4;
      const int ks = k == 0 ? 7 : prev_tiles * 4;
      std::vector<uint32_t> wf_f((size_t)tiles * ks * 64, PACK_ZERO), bf_f((size_t)tiles * 4 * 64, PACK_ZERO);
      std::vector<uint32_t> wf_d(wf_f.size(), PACK_ZERO), bf_d(bf_f.size(), PACK_ZERO);
      for (int t = 0; t < tiles; ++t)
        for (int lane = 0; lane < 64; ++lane) {
          const int uo = 16 * t + (lane & 15), q = lane >> 4;
          for (int s2 = 0; s2 < ks; ++s2) {
            // input unit of k-step s2 for lane group q: natural order for the first layer and for f64;
            // for f32 the accumulator layout of the previous layer permutes it (mlp_mfma.hpp)
            const int ui_d = 4 * s2 + q;
            const int ui_f = k == 0 ? 4 * s2 + q : 16 * (s2 / 4) + 4 * q + (s2 % 4);
            const size_t o = ((size_t)t * ks + s2) * 64 + lane;
            if (uo < out && ui_d < in) wf_d[o] = wc(k, (size_t)uo * in + ui_d);
            if (uo < out && ui_f < in) wf_f[o] = wc(k, (size_t)uo * in + ui_f);
          }
          for (int r = 0; r < 4; ++r) {
            const int row_d = 16 * t + q + 4 * r, row_f = 16 * t + 4 * q + r;
            const size_t o = ((size_t)t * 4 + r) * 64 + lane;
            if (row_d < out) bf_d[o] = bc(k, row_d);
            if (row_f < out) bf_f[o] = bc(k, row_f);
          }
        }
      add(std::move(wf_f), PACK_F32, (const void**)&mf.wfrag[k]);
      add(std::move(bf_f), PACK_F32, (const void**)&mf.bfrag[k]);
      add(std::move(wf_d), PACK_F64, (const void**)&md.wfrag[k]);
      add(std::move(bf_d), PACK_F64, (const void**)&md.bfrag[k]);
      mf.ksteps[k] = md.ksteps[k] = ks;
      mf.otiles[k] = md.otiles[k] = tiles;
      {
        // base chain of mlp_jvp.hpp: fp32 A fragments of v_mfma_f64_4x4x4_4b, four k-steps per 16-byte element
        const int kg = k == 0 ? 2 : prev_tiles;  // k-groups of 16 inputs (first layer: 28 -> 32)
        std::vector<uint32_t> wq((size_t)tiles * kg * 64 * 4, PACK_ZERO), bq((size_t)tiles * 64, PACK_ZERO);
        for (int t = 0; t < tiles; ++t)
          for (int lane = 0; lane < 64; ++lane) {
            const int uo = 16 * t + (lane & 15);
            for (int g = 0; g < kg; ++g)
              for (int e = 0; e < 4; ++e) {
                const int ui = 4 * (4 * g + e) + (lane >> 4);
                if (uo < out && ui < in) wq[(((size_t)t * kg + g) * 64 + lane) * 4 + e] = wc(k, (size_t)uo * in + ui);
              }
            const int ub = 16 * t + 4 * ((lane >> 2) & 3) + (lane >> 4);  // unit of this lane in the D layout
            if (ub < out) bq[(size_t)t * 64 + lane] = bc(k, ub);
          }
        add(std::move(wq), PACK_F32, (const void**)&mf.wq[k], (const void**)&md.wq[k]);
        add(std::move(bq), PACK_F32, (const void**)&mf.bq[k], (const void**)&md.bq[k]);
        mf.kgroups[k] = md.kgroups[k] = kg;
      }
      {
        // bf16 fragments of the JVP chain (mlp_jvp.hpp): k-steps of 32; beyond the first layer the k order is the one
        // two adjacent 16x16x32 accumulator tiles present: slot (q, j) of k-step s = unit 32 s + 4 q + j (j < 4),
        // 32 s + 16 + 4 q + j - 4 (j >= 4)
        const int jks = k == 0 ? 1 : prev_tiles / 2;
        std::vector<uint32_t> jf((size_t)tiles * jks * 64 * 8, PACK_ZERO);
        for (int t = 0; t < tiles; ++t)
          for (int s2 = 0; s2 < jks; ++s2)
            for (int lane = 0; lane < 64; ++lane) {
              const int uo = 16 * t + (lane & 15), q = lane >> 4;
              for (int j = 0; j < 8; ++j) {
                const int ui = k == 0 ? 8 * q + j : 32 * s2 + (j < 4 ? 4 * q + j : 16 + 4 * q + (j - 4));
                if (uo < out && ui < in) jf[(((size_t)t * jks + s2) * 64 + lane) * 8 + j] = wc(k, (size_t)uo * in + ui);
              }
            }
        add(std::move(jf), PACK_BF16, (const void**)&mf.jfrag[k], (const void**)&md.jfrag[k]);
        mf.jksteps[k] = md.jksteps[k] = jks;
      }
      prev_tiles = tiles;
    }
    mf.f32_ok = md.f32_ok = 0;
    if (n_layers == 3 && dims[1] <= 64 && dims[2] <= 64) {
      // fp32 base chain (mlp_jvp_tile3f): row l of a layer's weight matrix on lane l; the output layer's 32 rows twice,
      // rows 32..63 with the second half of its k range
      const int groups[3] = {7, 16, 8};
      for (int k = 0; k < 3; ++k) {
        const int in = dims[k], out = dims[k + 1];
        std::vector<uint32_t> w32((size_t)groups[k] * 64 * 4, PACK_ZERO), b32(64, PACK_ZERO);
        for (int lane = 0; lane < 64; ++lane) {
          const int row = k == 2 ? (lane & 31) : lane;
          const int k0 = k == 2 ? 32 * (lane >> 5) : 0;
          if (row < out)
            for (int g = 0; g < groups[k]; ++g)
              for (int e = 0; e < 4; ++e) {
                const int ui = k0 + 4 * g + e;
                if (ui < in) w32[((size_t)g * 64 + lane) * 4 + e] = wc(k, (size_t)row * in + ui);
              }
          if (lane < out && (k < 2 || lane < 32)) b32[lane] = bc(k, lane);
        }
        add(std::move(w32), PACK_F32, (const void**)&mf.w32[k], (const void**)&md.w32[k]);
        add(std::move(b32), PACK_F32, (const void**)&mf.b32[k], (const void**)&md.b32[k]);
      }
      mf.f32_ok = md.f32_ok = 1;
    }
    mf.mfma_ok = md.mfma_ok = 1;  // (kr_set_mlp applies the option "mfma_mlp" on every call)
    mf.jvp_ok = md.jvp_ok = (n_layers == 2 || dims[2] <= 64 * 3) ? 1 : 0;  // MJ_ACT_SLOTS - 1 chunks of the second hidden layer
  }
  if ((int)jobs.size() > PACK_MAX_JOBS) { set_error("kr_set_mlp: too many packed buffers"); return KR_E_ARG; }
  // one arena for every buffer (256-byte aligned pieces), one index table
  const size_t esz[3] = {4, 8, 2};
  size_t arena_bytes = 0, total = 0;
  std::vector<size_t> off(jobs.size());
  for (size_t j = 0; j < jobs.size(); ++j) {
    off[j] = arena_bytes;
    arena_bytes += (jobs[j].code.size() * esz[jobs[j].type] + 255) / 256 * 256;
    total += jobs[j].code.size();
  }
  if (total >= 0xFFFFFFF0ull) { set_error("kr_set_mlp: network too large"); return KR_E_ARG; }
  KR_HIP(hipMalloc(&P.arena, arena_bytes));
  KR_HIP(hipMalloc(&P.idx, total * sizeof(uint32_t)));
  KR_HIP(hipMalloc(&P.staging, P.src_off[2 * n_layers] * sizeof(float)));
  std::vector<uint32_t> all(total);
  P.jobs.n = (int)jobs.size();
  size_t pos = 0;
  for (size_t j = 0; j < jobs.size(); ++j) {
    P.jobs.start[j] = (uint32_t)pos;
    P.jobs.dst[j] = P.arena + off[j];
    P.jobs.type[j] = (uint8_t)jobs[j].type;
    *jobs[j].t0 = P.arena + off[j];
    if (jobs[j].t1) *jobs[j].t1 = P.arena + off[j];
    std::memcpy(all.data() + pos, jobs[j].code.data(), jobs[j].code.size() * sizeof(uint32_t));
    pos += jobs[j].code.size();
  }
  P.jobs.start[jobs.size()] = (uint32_t)pos;
  KR_HIP(hipMemcpy(P.idx, all.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice));
  return KR_OK;
}
}  // namespace kr

extern "C" {

int kr_set_mlp(kr_handle* h, int n_layers, const int32_t* dims, const int32_t* acts, const float* const* W,
               const float* const* b, int src_on_device, void* stream) {
  KR_CHECK_H(h);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc_order_ = order_stream(h, s)) return rc_order_;
  h->mlp_f = MlpDev<float>{};
  h->mlp_d = MlpDev<double>{};
  if (n_layers == 0) return KR_OK;  // (the plan of the last shape stays cached)
  if (n_layers < 0 || n_layers > KR_MAX_LAYERS) {
    set_error("n_layers out of range");
    return KR_E_ARG;
  }
  KR_CHECK_PTR(dims); KR_CHECK_PTR(acts); KR_CHECK_PTR(W); KR_CHECK_PTR(b);
  const int want_in = h->params.nn_input_history ? 53 : 28;
  if (dims[0] != want_in || dims[n_layers] != 25) {
    set_error("MLP must map " + std::to_string(want_in) + " -> 25 (cosserat_ode_torch.py:60-62)");
    return KR_E_ARG;
  }
  uint64_t key = 1469598103934665603ull;  // FNV-1a over the shape
  auto mix = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
  mix((uint64_t)n_layers);
  for (int k = 0; k <= n_layers; ++k) {
    if (dims[k] <= 0) { set_error("bad layer width"); return KR_E_ARG; }
    mix((uint64_t)dims[k]);
  }
  for (int k = 0; k < n_layers; ++k) {
    if (acts[k] < KR_ACT_NONE || acts[k] > KR_ACT_ELU) { set_error("bad activation code"); return KR_E_ARG; }
    KR_CHECK_PTR(W[k]); KR_CHECK_PTR(b[k]);
    mix((uint64_t)acts[k] + 0x100);
  }
  if (!h->mlp_plan || h->mlp_plan->key != key) {
    // another shape: the old buffers may still be read by launches in flight
    KR_HIP(hipDeviceSynchronize());
    free_mlp_plan(h);
    h->mlp_plan = new kr_mlp_plan;
    if (int rc = build_mlp_plan(h, n_layers, dims, acts, *h->mlp_plan)) {
      free_mlp_plan(h);
      return rc;
    }
    h->mlp_plan->key = key;
  }
  kr_mlp_plan& P = *h->mlp_plan;
  for (int k = 0; k < n_layers; ++k) {
    if (src_on_device) {
      P.jobs.src[2 * k] = W[k];
      P.jobs.src[2 * k + 1] = b[k];
    } else {
      KR_HIP(hipMemcpyAsync(P.staging + P.src_off[2 * k], W[k], sizeof(float) * dims[k] * dims[k + 1], hipMemcpyHostToDevice, s));
      KR_HIP(hipMemcpyAsync(P.staging + P.src_off[2 * k + 1], b[k], sizeof(float) * dims[k + 1], hipMemcpyHostToDevice, s));
      P.jobs.src[2 * k] = P.staging + P.src_off[2 * k];
      P.jobs.src[2 * k + 1] = P.staging + P.src_off[2 * k + 1];
    }
  }
  // host sources are the caller's to free or overwrite as soon as this call returns: their copies must have left them
  if (!src_on_device) KR_HIP(hipStreamSynchronize(s));
  const uint32_t total = P.jobs.start[P.jobs.n];
  int grid = (int)((total + 255) / 256);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(kr::mlp_pack_kernel, dim3(grid), dim3(256), 0, s, P.jobs, P.idx);
  KR_HIP(hipGetLastError());
  h->mlp_f = P.mf;
  h->mlp_d = P.md;
  h->mlp_f.mfma_ok = h->mlp_d.mfma_ok = (P.mf.mfma_ok && h->mfma_mlp) ? 1 : 0;
  return KR_OK;
}

int kr_ode_batch(kr_handle* h, int64_t Q, const void* y, const void* yh, const void* zh, const void* tf, void* dys,
                 void* z, int use_nn, int dtype, void* stream) {
  KR_CHECK_H(h);
  KR_CHECK_DTYPE(dtype);
  if (Q < 0) { set_error("Q < 0"); return KR_E_ARG; }
  if (Q == 0) return KR_OK;
  KR_CHECK_PTR(y); KR_CHECK_PTR(yh); KR_CHECK_PTR(zh); KR_CHECK_PTR(tf); KR_CHECK_PTR(dys); KR_CHECK_PTR(z);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc_order_ = order_stream(h, s)) return rc_order_;
  if (dtype == KR_F32)
    return launch_ode_batch<float>(h, Q, (const float*)y, (const float*)yh, (const float*)zh, (const float*)tf,
                                   (float*)dys, (float*)z, use_nn, s);
  return launch_ode_batch<double>(h, Q, (const double*)y, (const double*)yh, (const double*)zh, (const double*)tf,
                                  (double*)dys, (double*)z, use_nn, s);
}

#define KR_BATCH_PROLOGUE(B)                       \
  KR_CHECK_H(h);                                   \
  KR_CHECK_DTYPE(dtype);                           \
  if ((B) < 0) { set_error("B < 0"); return KR_E_ARG; } \
  if ((B) == 0) return KR_OK;                      \
  hipStream_t s = static_cast<hipStream_t>(stream); \
  if (int rc_order_ = order_stream(h, s)) return rc_order_;

int kr_state_init_straight(kr_handle* h, int64_t B, void* state, int dtype, void* stream) {
  KR_BATCH_PROLOGUE(B);
  KR_CHECK_PTR(state);
  return dtype == KR_F32 ? launch_init_straight<float>(h, B, (float*)state, s)
                         : launch_init_straight<double>(h, B, (double*)state, s);
}
int kr_state_pack(kr_handle* h, int64_t B, const void* y_fm, const void* z_fm, void* state, int dtype, void* stream) {
  KR_BATCH_PROLOGUE(B);
  KR_CHECK_PTR(y_fm); KR_CHECK_PTR(z_fm); KR_CHECK_PTR(state);
  return dtype == KR_F32 ? launch_pack<float>(h, B, (const float*)y_fm, (const float*)z_fm, (float*)state, s)
                         : launch_pack<double>(h, B, (const double*)y_fm, (const double*)z_fm, (double*)state, s);
}
int kr_state_unpack(kr_handle* h, int64_t B, const void* state, void* y_fm, void* z_fm, int dtype, void* stream) {
  KR_BATCH_PROLOGUE(B);
  KR_CHECK_PTR(y_fm); KR_CHECK_PTR(z_fm); KR_CHECK_PTR(state);
  return dtype == KR_F32 ? launch_unpack<float>(h, B, (const float*)state, (float*)y_fm, (float*)z_fm, s)
                         : launch_unpack<double>(h, B, (const double*)state, (double*)y_fm, (double*)z_fm, s);
}
int kr_state_unpack50(kr_handle* h, int64_t B, const void* state, const void* m1, const void* m2, void* out, int dtype,
                      void* stream) {
  KR_BATCH_PROLOGUE(B);
  KR_CHECK_PTR(state); KR_CHECK_PTR(m1); KR_CHECK_PTR(m2); KR_CHECK_PTR(out);
  return dtype == KR_F32
             ? launch_unpack50<float>(h, B, (const float*)state, (const float*)m1, (const float*)m2, (float*)out, s)
             : launch_unpack50<double>(h, B, (const double*)state, (const double*)m1, (const double*)m2, (double*)out, s);
}
int kr_state_tip(kr_handle* h, int64_t B, const void* state, void* tip, int dtype, void* stream) {
  KR_BATCH_PROLOGUE(B);
  KR_CHECK_PTR(state); KR_CHECK_PTR(tip);
  return dtype == KR_F32 ? launch_tip<float>(h, B, (const float*)state, (float*)tip, s)
                         : launch_tip<double>(h, B, (const double*)state, (double*)tip, s);
}

int kr_residual_batch(kr_handle* h, int64_t B, int scheme, const void* G, const void* state_prev,
                      const void* state_cur, void* state_next, const void* tensions, void* r, int use_nn,
                      int hist_is_explicit, int dtype, void* stream) {
  KR_BATCH_PROLOGUE(B);
  KR_CHECK_PTR(G); KR_CHECK_PTR(state_cur); KR_CHECK_PTR(state_next);
  KR_CHECK_PTR(tensions); KR_CHECK_PTR(r);
  if (hist_is_explicit) state_prev = state_cur;
  KR_CHECK_PTR(state_prev);
  if (dtype == KR_F32) {
    auto a = make_args<float>(h, B, state_prev, state_cur, state_next, const_cast<void*>(G), tensions, 4, 0, 0);
    a.mode = 1; a.r_out = (float*)r;
    if (hist_is_explicit) { a.hc1 = 1.f; a.hc2 = 0.f; }
    return launch_step<float>(h, scheme, use_nn, a, s);
  }
  auto a = make_args<double>(h, B, state_prev, state_cur, state_next, const_cast<void*>(G), tensions, 4, 0, 0);
  a.mode = 1; a.r_out = (double*)r;
  if (hist_is_explicit) { a.hc1 = 1.0; a.hc2 = 0.0; }
  return launch_step<double>(h, scheme, use_nn, a, s);
}

int kr_residual_mid_batch(kr_handle* h, int64_t B, int scheme, const void* G, const void* hist, const void* hist_mid,
                          void* state_next, const void* tensions, void* r, int use_nn, int dtype, void* stream) {
  KR_BATCH_PROLOGUE(B);
  KR_CHECK_PTR(G); KR_CHECK_PTR(hist); KR_CHECK_PTR(state_next);
  KR_CHECK_PTR(tensions); KR_CHECK_PTR(r);
  if (hist_mid && scheme != KR_RK4) { set_error("hist_mid is read by the RK4 sweep only"); return KR_E_ARG; }
  if (dtype == KR_F32) {
    auto a = make_args<float>(h, B, hist, hist, state_next, const_cast<void*>(G), tensions, 4, 0, 0);
    a.mode = 1; a.r_out = (float*)r; a.hc1 = 1.f; a.hc2 = 0.f; a.mid = (const float*)hist_mid;
    return launch_step<float>(h, scheme, use_nn, a, s);
  }
  auto a = make_args<double>(h, B, hist, hist, state_next, const_cast<void*>(G), tensions, 4, 0, 0);
  a.mode = 1; a.r_out = (double*)r; a.hc1 = 1.0; a.hc2 = 0.0; a.mid = (const double*)hist_mid;
  return launch_step<double>(h, scheme, use_nn, a, s);
}

int kr_step_batch(kr_handle* h, int64_t B, int scheme, const void* state_prev, const void* state_cur,
                  void* state_next, void* G, const void* tensions, double tol, int maxit, int32_t* status,
                  int32_t* iters, int use_nn, const void* state_prev2, int predictor, int dtype, void* stream) {
  KR_BATCH_PROLOGUE(B);
  KR_CHECK_PTR(G); KR_CHECK_PTR(state_prev); KR_CHECK_PTR(state_cur); KR_CHECK_PTR(state_next);
  KR_CHECK_PTR(tensions);
  if (state_next == state_cur || state_next == state_prev) {
    set_error("state_next must not alias state_cur / state_prev");
    return KR_E_ARG;
  }
  // predictor: -1 = highest order the given history allows, else min(requested, available)
  int avail = state_prev2 ? 2 : (state_prev != state_cur ? 1 : 0);
  if (predictor >= 0 && predictor < avail) avail = predictor;
  if (dtype == KR_F32) {
    auto a = make_args<float>(h, B, state_prev, state_cur, state_next, G, tensions, 4, tol, maxit);
    a.status = status; a.iters = iters;
    a.prev2 = (const float*)state_prev2; a.pred_order = avail;
    return launch_step<float>(h, scheme, use_nn, a, s);
  }
  auto a = make_args<double>(h, B, state_prev, state_cur, state_next, G, tensions, 4, tol, maxit);
  a.status = status; a.iters = iters;
  a.prev2 = (const double*)state_prev2; a.pred_order = avail;
  return launch_step<double>(h, scheme, use_nn, a, s);
}

int kr_simulate_prepare(kr_handle* h, int64_t B, int dtype) {
  KR_CHECK_H(h);
  KR_CHECK_DTYPE(dtype);
  if (B <= 0) return KR_OK;
  int rc = ensure_resume(h, B);
  if (rc) return rc;
  if (dtype == KR_F32) { rc = prepare_mso_sim<float>(h, B); if (rc != 1 && rc) return rc; rc = prepare_ms_sim<float>(h); }
  else { rc = prepare_mso_sim<double>(h, B); if (rc != 1 && rc) return rc; rc = prepare_ms_sim<double>(h); }
  return rc == 1 ? KR_OK : rc;
}

int kr_simulate_batch(kr_handle* h, int64_t B, int64_t T, int scheme, const void* ctl, void* states, int ring, void* G,
                      void* tip, double tol, int maxit, int32_t* status, int use_nn, const void* state_prev_init,
                      int dtype, void* stream) {
  KR_BATCH_PROLOGUE(B);
  if (T < 0) { set_error("T < 0"); return KR_E_ARG; }
  if (T == 0) return KR_OK;
  KR_CHECK_PTR(ctl); KR_CHECK_PTR(states); KR_CHECK_PTR(G);
  return dtype == KR_F32
             ? simulate_impl<float>(h, B, T, scheme, ctl, states, ring, G, tip, tol, maxit, status, use_nn,
                                    state_prev_init, s)
             : simulate_impl<double>(h, B, T, scheme, ctl, states, ring, G, tip, tol, maxit, status, use_nn,
                                     state_prev_init, s);
}

}  // extern "C"
