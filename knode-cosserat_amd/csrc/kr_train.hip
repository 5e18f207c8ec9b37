// kr_train.hip - KNODE one-step-ahead training path.
//
//   kr_next_segment_physics  the physics half of CosseratRodTorch.parallelGetNextSegmentEuler
//                            (reference cosserat_ode_torch.py:401-437; with idx = 1..N-1 it is
//                            getNextSegmentEuler, :370-399): gathers the key columns of the
//                            teacher-forced next state, evaluates the rod derivative there and
//                            emits (a) the MLP input rows and (b) the prediction without the MLP.
//   kr_mlp_forward/backward  the residual MLP (cosserat_ode_torch.py:60-62,131-134) over Q rows
//                            as dense GEMMs on the matrix cores: v_mfma_f32_32x32x2_f32, i.e.
//                            exact fp32 like the reference's training arithmetic.
//   kr_loss_fwd_bwd          prediction assembly + the four-term loss of physics_train.py:252-259
//                            (incl. Utils/transformations.quaternion_to_euler) + its gradient with
//                            respect to the MLP output.
//
// The physics has no trainable parameter and the reference feeds ground-truth columns to every
// segment ("y and z are not updated here", cosserat_ode_torch.py:391), so d loss / d theta flows only
// through the MLP output: training = elementwise physics (forward only) + MLP forward/backward.
#include <cmath>

#include "kr_internal.hpp"
#include "kr_loss_device.hpp"

namespace kr {

// ---------------------------------------------------------------------------
// T1: physics rows
// ---------------------------------------------------------------------------
template <typename T, bool DIAG>
__global__ void next_segment_physics_kernel(const RodConst<T> P, int64_t S, int K, int hist, const T* __restrict__ Gs,
                                            const T* __restrict__ yh, const T* __restrict__ zh,
                                            const T* __restrict__ tens, const int32_t* __restrict__ idx,
                                            T* __restrict__ x, int in_pad, T* __restrict__ base) {
  const int N = P.N;
  const int64_t rows = S * K;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < rows;
       row += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = row / K;
    const int k = (int)(row - s * K);
    const int j = idx[k] - 1;  // cosserat_ode_torch.py:412 gathers column segment_idx-1
    T yr[19], yhr[19], zhr[6];
#pragma unroll
    for (int r = 0; r < 19; ++r) yr[r] = Gs[(s * 25 + r) * N + j];
#pragma unroll
    for (int r = 0; r < 19; ++r) yhr[r] = yh[(s * 19 + r) * N + j];
#pragma unroll
    for (int r = 0; r < 6; ++r) zhr[r] = zh[(s * 6 + r) * N + j];
    V3<T> tf{T(0), T(0), T(0)};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const T tt = tens[s * 4 + t];
      tf.x += tt * P.tdirs[t * 3 + 0];
      tf.y += tt * P.tdirs[t * 3 + 1];
      tf.z += tt * P.tdirs[t * 3 + 2];
    }
    const V3<T> fconst{P.rhoAg[0] + tf.x, P.rhoAg[1] + tf.y, P.rhoAg[2] + tf.z};
    RodState<T> y = rows_to_state(yr);
    RodHist<T> hst;
    hst.qh = {yhr[13], yhr[14], yhr[15]};
    hst.wh = {yhr[16], yhr[17], yhr[18]};
    hst.vh = {zhr[0], zhr[1], zhr[2]};
    hst.uh = {zhr[3], zhr[4], zhr[5]};
    hist_derive(P, hst);
    RodState<T> ks;
    V3<T> v, u;
    ode_eval<T, DIAG>(P, y, hst, fconst, ks, v, u);
    T ksr[19];
    state_to_rows(ks, ksr);
    T* b = base + row * 25;
#pragma unroll
    for (int r = 0; r < 19; ++r) b[r] = yr[r] + P.ds * ksr[r];
    b[19] = v.x; b[20] = v.y; b[21] = v.z; b[22] = u.x; b[23] = u.y; b[24] = u.z;
    // MLP input, cosserat_ode_torch.py:310-313
    T* xr = x + row * in_pad;
    int o = 0;
#pragma unroll
    for (int r = 0; r < 19; ++r) xr[o + r] = yr[r];
    o += 19;
    if (hist) {
#pragma unroll
      for (int r = 0; r < 19; ++r) xr[o + r] = yhr[r];
      o += 19;
    }
    xr[o + 0] = v.x; xr[o + 1] = v.y; xr[o + 2] = v.z; xr[o + 3] = u.x; xr[o + 4] = u.y; xr[o + 5] = u.z;
    o += 6;
    if (hist) {
#pragma unroll
      for (int r = 0; r < 6; ++r) xr[o + r] = zhr[r];
      o += 6;
    }
    xr[o + 0] = tf.x; xr[o + 1] = tf.y; xr[o + 2] = tf.z;
    o += 3;
    for (; o < in_pad; ++o) xr[o] = T(0);
  }
}

// ---------------------------------------------------------------------------
// T2/T3: fp32 GEMM on the matrix cores
// ---------------------------------------------------------------------------
// C[m][n] = sum_k A(m,k) B(k,n) with A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]
// (arbitrary strides cover every transpose the MLP needs).  Workgroup = 4 waves,
// tile 128 x 32, K in steps of 32 staged through LDS; each wave owns a 32 x 32
// tile and issues 16 v_mfma_f32_32x32x2_f32 per K step (A: lane l holds
// A[row l&31][k = l>>5], B: lane l holds B[k = l>>5][col l&31]; the two k lanes
// of an instruction take k = t and k = t+16 of the staged block so that every
// LDS read below is conflict free).
enum { EPI_STORE = 0, EPI_BIAS_ACT = 1, EPI_MUL_ACTGRAD = 2, EPI_ATOMIC = 3 };
constexpr int GM = 128, GN = 32, GK = 32;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  int M, N, K;
  const float* A; int64_t sam, sak;
  const float* B; int64_t sbk, sbn;
  float* C; int64_t ldc;        // C[m*ldc + n]
  const float* bias;            // EPI_BIAS_ACT: bias[n] (n < n_valid)
  float* C2;                    // EPI_BIAS_ACT: activations act(C) (same ldc); EPI_MUL_ACTGRAD: pre-activations
  int act;
  int n_valid;                  // columns >= n_valid are written as zero (padding)
  int k_chunk;                  // split-K: K range per blockIdx.z
};

template <int EPI, bool A_KCONTIG, bool B_NCONTIG>
__global__ __launch_bounds__(256) void gemm_f32_mfma(const GemmArgs g) {
  __shared__ float As[GM][GK + 1];
  __shared__ float Bs[GK][GN + 1];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int m0 = blockIdx.x * GM;
  const int n0 = blockIdx.y * GN;
  const int kbeg = blockIdx.z * g.k_chunk;
  const int kend = min(g.K, kbeg + g.k_chunk);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    // stage A tile (128 x 32): 16 elements per thread
#pragma unroll
    for (int e = 0; e < (GM * GK) / 256; ++e) {
      const int lin = e * 256 + tid;
      int mm, kk;
      if (A_KCONTIG) { mm = lin / GK; kk = lin % GK; }
      else { kk = lin / GM; mm = lin % GM; }
      const int m = m0 + mm, k = k0 + kk;
      As[mm][kk] = (m < g.M && k < kend) ? g.A[m * g.sam + k * g.sak] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < (GK * GN) / 256; ++e) {
      const int lin = e * 256 + tid;
      int kk, nn;
      if (B_NCONTIG) { kk = lin / GN; nn = lin % GN; }
      else { nn = lin / GK; kk = lin % GK; }
      const int k = k0 + kk, n = n0 + nn;
      Bs[kk][nn] = (k < kend && n < g.n_valid) ? g.B[k * g.sbk + n * g.sbn] : 0.f;
    }
    __syncthreads();
    const int r = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float a = As[wave * 32 + r][kh * 16 + t];
      const float b = Bs[kh * 16 + t][r];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int n = n0 + (lane & 31);
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int m = m0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    if (m >= g.M || n >= g.N) continue;
    float v = acc[reg];
    const int64_t o = (int64_t)m * g.ldc + n;
    if (EPI == EPI_STORE) {
      g.C[o] = n < g.n_valid ? v : 0.f;
    } else if (EPI == EPI_BIAS_ACT) {
      if (n < g.n_valid) {
        v += g.bias[n];
        g.C[o] = v;
        if (g.C2) g.C2[o] = activate<float>(g.act, v);
      } else {
        g.C[o] = 0.f;
        if (g.C2) g.C2[o] = 0.f;
      }
    } else if (EPI == EPI_MUL_ACTGRAD) {
      g.C[o] = n < g.n_valid ? v * activate_grad<float>(g.act, g.C2[o]) : 0.f;
    } else {
      if (n < g.n_valid) atomicAdd(&g.C[o], v);
    }
  }
}

template <int EPI>
static int launch_gemm(const GemmArgs& g, bool a_kcontig, bool b_ncontig, int splits, hipStream_t s) {
  dim3 grid((g.M + GM - 1) / GM, (g.N + GN - 1) / GN, splits);
  if (a_kcontig && b_ncontig) hipLaunchKernelGGL((gemm_f32_mfma<EPI, true, true>), grid, dim3(256), 0, s, g);
  else if (a_kcontig) hipLaunchKernelGGL((gemm_f32_mfma<EPI, true, false>), grid, dim3(256), 0, s, g);
  else if (b_ncontig) hipLaunchKernelGGL((gemm_f32_mfma<EPI, false, true>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_f32_mfma<EPI, false, false>), grid, dim3(256), 0, s, g);
  KR_HIP(hipGetLastError());
  return KR_OK;
}

// column sums of d[Q][ld] -> out[n] (bias gradients); out pre-zeroed
__global__ void colsum_kernel(int64_t Q, int n, int ld, const float* __restrict__ d, float* __restrict__ out) {
  const int c = blockIdx.y * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const int64_t chunk = (Q + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = blockIdx.x * chunk;
  const int64_t r1 = r0 + chunk < Q ? r0 + chunk : Q;
  float s = 0.f;
  for (int64_t r = r0; r < r1; ++r) s += d[r * ld + c];
  atomicAdd(&out[c], s);
}

static inline int pad32(int v) { return (v + 31) / 32 * 32; }

// workspace layout of one forward pass: for every layer k < n_layers-1 the
// pre-activations Z_k[Q][pad32(dims[k+1])] followed by the activations A_k of the
// same shape; then two scratch buffers of the widest hidden shape for the
// backward sweep.
struct MlpWs {
  float* Z[KR_MAX_LAYERS];
  float* Aact[KR_MAX_LAYERS];
  float* d0;
  float* d1;
  size_t bytes;
};
static MlpWs carve_ws(void* ws, int n_layers, const int32_t* dims, int64_t Q) {
  MlpWs w{};
  unsigned char* p = static_cast<unsigned char*>(ws);
  size_t off = 0;
  int maxh = 32;
  for (int k = 0; k + 1 < n_layers; ++k) {
    const size_t sz = ((size_t)Q * pad32(dims[k + 1]) * sizeof(float) + 255) & ~size_t(255);
    w.Z[k] = reinterpret_cast<float*>(p + off); off += sz;
    w.Aact[k] = reinterpret_cast<float*>(p + off); off += sz;
    if (pad32(dims[k + 1]) > maxh) maxh = pad32(dims[k + 1]);
  }
  const size_t sz = ((size_t)Q * maxh * sizeof(float) + 255) & ~size_t(255);
  w.d0 = reinterpret_cast<float*>(p + off); off += sz;
  w.d1 = reinterpret_cast<float*>(p + off); off += sz;
  w.bytes = off;
  return w;
}

// gathers the target values every (s, k) row is scored against: rows[s*K+k][r] = target[s][r][idx[k]] for the
// y rows, target[s][r][idx[k]-1] for the z rows (physics_train.py:252-259)
__global__ void gather_targets_kernel(int N, int64_t S, int K, const float* __restrict__ target,
                                      const int32_t* __restrict__ idx, float* __restrict__ rows) {
  const int64_t n = S * K * 25;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i % 25);
    const int64_t row = i / 25;
    const int64_t s = row / K;
    const int k = (int)(row - s * K);
    const int j = r < 19 ? idx[k] : idx[k] - 1;
    rows[i] = target[(s * 25 + r) * (int64_t)N + j];
  }
}

// ---------------------------------------------------------------------------
// T4: prediction + four-term loss + gradient w.r.t. the MLP output
// ---------------------------------------------------------------------------
// ROWS: `target` holds pre-gathered rows [S*K][25] (kr_gather_targets) instead of the full [S][25][N] states
template <bool ROWS>
__global__ __launch_bounds__(256) void loss_kernel(int N, float ds, int64_t S, int K, const float* __restrict__ base,
                                                   const float* __restrict__ out, int ld_out,
                                                   const float* __restrict__ target, const int32_t* __restrict__ idx,
                                                   float inv_denom, float* __restrict__ pred,
                                                   float* __restrict__ loss, float* __restrict__ dout, int ld_dout) {
  const int64_t rows = S * K;
  float part = 0.f;
  const LossWeights lw = loss_weights(inv_denom, K);
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < rows;
       row += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = row / K;
    const int k = (int)(row - s * K);
    float tgv[25];  // target values of this row: y rows at column idx[k], z rows at idx[k]-1 (physics_train.py:259)
    if constexpr (ROWS) {
#pragma unroll
      for (int r = 0; r < 25; ++r) tgv[r] = target[row * 25 + r];
    } else {
      const int jc = idx[k];
      const float* tg = target + s * 25 * (int64_t)N;
#pragma unroll
      for (int r = 0; r < 19; ++r) tgv[r] = tg[r * N + jc];
#pragma unroll
      for (int r = 19; r < 25; ++r) tgv[r] = tg[r * N + (jc - 1)];
    }
    float p[25], g[25];
#pragma unroll
    for (int r = 0; r < 19; ++r) p[r] = base[row * 25 + r] + ds * out[row * ld_out + r];
#pragma unroll
    for (int r = 19; r < 25; ++r) p[r] = base[row * 25 + r] + out[row * ld_out + r];
    if (pred) {
#pragma unroll
      for (int r = 0; r < 25; ++r) pred[row * 25 + r] = p[r];
    }
    part += loss_row(p, tgv, lw, g);
    // chain through pred = base + [ds*out[:19], out[19:]]
#pragma unroll
    for (int r = 0; r < 19; ++r) dout[row * ld_dout + r] = ds * g[r];
#pragma unroll
    for (int r = 19; r < 25; ++r) dout[row * ld_dout + r] = g[r];
    for (int r = 25; r < ld_dout; ++r) dout[row * ld_dout + r] = 0.f;
  }
  // block reduction of the loss
  __shared__ float red[256];
  red[threadIdx.x] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(loss, red[0]);
}


// torch.optim.Adam (no amsgrad) on one flat parameter vector, followed by the reference's weight clamp
// (physics_train.py:299-304: p = max(p, lower), lower = 0 for weight matrices, -inf for biases) and by the
// zeroing of the gradient buffer (and its trailing loss slot) for the next epoch: one launch per epoch
// instead of torch's three multi-tensor kernels, one clamp per layer and seven memsets.
__global__ void adam_kernel(int64_t n, int64_t n_zero, float* __restrict__ p, float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v, const float* __restrict__ lower,
                            float step_size, float b1, float b2, float inv_sqrt_bc2, float eps, float wd) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_zero; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n) {
      float pi = p[i];
      float gi = g[i];
      if (wd != 0.f) gi = fmaf(wd, pi, gi);
      const float mi = fmaf(b1, m[i], (1.f - b1) * gi);           // exp_avg.lerp_(grad, 1 - beta1)
      const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
      m[i] = mi;
      v[i] = vi;
      const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;         // (exp_avg_sq.sqrt() / sqrt(bias_correction2)).add_(eps)
      pi -= step_size * (mi / denom);                             // param.addcdiv_(exp_avg, denom, value=-lr / bias_correction1)
      if (lower) pi = fmaxf(pi, lower[i]);
      p[i] = pi;
    }
    g[i] = 0.f;
  }
}
// Adam + clamp + gradient zeroing as above, with the learning rate and torch's ReduceLROnPlateau(mode="min",
// threshold_mode="rel", cooldown=0) kept ON THE DEVICE, so that an epoch needs no host round trip
// (physics_train.py:206,296-297 call scheduler.step(total_loss) after every optimizer.step()).
// sched[0], sched[1]: learning rate of odd / even steps (step s reads sched[(s - 1) & 1] and writes the rate of step
// s + 1 to sched[s & 1]: no thread of this launch reads what another one writes), sched[2] best loss so far,
// sched[3] bad epochs in a row, sched[4] loss of this step, sched[5] number of reductions.
// The thread that owns the loss slot of the gradient buffer (index loss_idx) is the scheduler: it reads the loss
// before zeroing it.
__global__ void adam_plateau_kernel(int64_t n, int64_t n_zero, float* __restrict__ p, float* __restrict__ g,
                                    float* __restrict__ m, float* __restrict__ v, const float* __restrict__ lower,
                                    double* __restrict__ sched, int parity, float inv_bc1, float b1, float b2,
                                    float inv_sqrt_bc2, float eps, float wd, int64_t loss_idx, double factor,
                                    int patience, double threshold, double min_lr, float* __restrict__ loss_log) {
  const double lr = sched[parity];
  const float step_size = (float)(lr * (double)inv_bc1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_zero; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n) {
      float pi = p[i];
      float gi = g[i];
      if (wd != 0.f) gi = fmaf(wd, pi, gi);
      const float mi = fmaf(b1, m[i], (1.f - b1) * gi);
      const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
      m[i] = mi;
      v[i] = vi;
      const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
      pi -= step_size * (mi / denom);
      if (lower) pi = fmaxf(pi, lower[i]);
      p[i] = pi;
    }
    if (i == loss_idx) {
      const double cur = (double)g[i];
      double best = sched[2], bad = sched[3], red = sched[5];
      if (cur < best * (1.0 - threshold)) { best = cur; bad = 0.0; }  // (torch: best starts at +inf; a NaN loss is "not better")
      else bad += 1.0;
      double next = lr;
      if (bad > (double)patience) {
        const double cand = fmax(lr * factor, min_lr);
        if (lr - cand > 1e-8) { next = cand; red += 1.0; }  // ReduceLROnPlateau's eps
        bad = 0.0;
      }
      sched[parity ^ 1] = next;
      sched[2] = best; sched[3] = bad; sched[4] = cur; sched[5] = red;
      if (loss_log) *loss_log = (float)cur;
    }
    g[i] = 0.f;
  }
}
}  // namespace kr

using namespace kr;

#define KR_CHECK_H(h)            \
  if (!(h)) {                    \
    set_error("null handle");    \
    return KR_E_ARG;             \
  }
#define KR_CHECK_PTR(p)                        \
  if (!(p)) {                                  \
    set_error("null pointer argument: " #p);   \
    return KR_E_ARG;                           \
  }

static int check_mlp_shape(int n_layers, const int32_t* dims) {
  if (n_layers < 1 || n_layers > KR_MAX_LAYERS || !dims) {
    set_error("n_layers out of range");
    return KR_E_ARG;
  }
  for (int k = 0; k <= n_layers; ++k)
    if (dims[k] <= 0) {
      set_error("bad layer width");
      return KR_E_ARG;
    }
  if (dims[n_layers] > 32) {  // out / dout rows are [32]; the rod's MLP has 25 (cosserat_ode_torch.py:62)
    set_error("the last layer of the training MLP must have at most 32 outputs");
    return KR_E_ARG;
  }
  return KR_OK;
}

extern "C" {

int kr_next_segment_physics(kr_handle* h, int64_t S, int K, const void* Gs, const void* yh, const void* zh,
                            const void* tensions, const int32_t* idx, void* x, int in_pad, void* base, int dtype,
                            void* stream) {
  KR_CHECK_H(h);
  if (dtype != KR_F32 && dtype != KR_F64) { set_error("bad dtype"); return KR_E_ARG; }
  if (S < 0 || K < 0) { set_error("negative size"); return KR_E_ARG; }
  if (S == 0 || K == 0) return KR_OK;
  KR_CHECK_PTR(Gs); KR_CHECK_PTR(yh); KR_CHECK_PTR(zh); KR_CHECK_PTR(tensions); KR_CHECK_PTR(idx);
  KR_CHECK_PTR(x); KR_CHECK_PTR(base);
  const int hist = h->params.nn_input_history ? 1 : 0;
  const int need = hist ? 53 : 28;
  if (in_pad < need) { set_error("in_pad smaller than the MLP input width"); return KR_E_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc_order_ = order_stream(h, s)) return rc_order_;
  const int64_t rows = S * K;
  int grid = (int)((rows + 255) / 256);
  if (grid > 4096) grid = 4096;
#define KR_NS(T, D)                                                                                          \
  hipLaunchKernelGGL((next_segment_physics_kernel<T, D>), dim3(grid), dim3(256), 0, s, consts<T>(h), S, K, hist, \
                     (const T*)Gs, (const T*)yh, (const T*)zh, (const T*)tensions, idx, (T*)x, in_pad, (T*)base)
  if (dtype == KR_F32) {
    if (h->cf.diag) KR_NS(float, true); else KR_NS(float, false);
  } else {
    if (h->cd.diag) KR_NS(double, true); else KR_NS(double, false);
  }
#undef KR_NS
  KR_HIP(hipGetLastError());
  return KR_OK;
}

size_t kr_mlp_ws_bytes(int n_layers, const int32_t* dims, int64_t Q) {
  if (n_layers < 1 || n_layers > KR_MAX_LAYERS || !dims || Q <= 0) return 0;
  size_t n = carve_ws(nullptr, n_layers, dims, Q).bytes;
  if (n_layers == 2 || n_layers == 3) {
    const size_t f = fused_ws_bytes(n_layers, dims, Q);
    if (f > n) n = f;
  }
  return n;
}

int kr_mlp_forward(kr_handle* h, int64_t Q, int n_layers, const int32_t* dims, const int32_t* acts,
                   const float* const* W, const float* const* b, const float* x, int in_pad, float* out, void* ws,
                   void* stream) {
  KR_CHECK_H(h);
  int rc = check_mlp_shape(n_layers, dims);
  if (rc) return rc;
  if (Q < 0) { set_error("Q < 0"); return KR_E_ARG; }
  if (Q == 0) return KR_OK;
  KR_CHECK_PTR(acts); KR_CHECK_PTR(W); KR_CHECK_PTR(b); KR_CHECK_PTR(x); KR_CHECK_PTR(out);
  if (n_layers > 1) KR_CHECK_PTR(ws);
  if (in_pad < dims[0]) { set_error("in_pad < dims[0]"); return KR_E_ARG; }
  if (acts[n_layers - 1] != KR_ACT_NONE) {
    set_error("an activation after the last layer is not supported by the training path");
    return KR_E_UNSUPPORTED;
  }
  if (Q > (int64_t)1 << 30) { set_error("Q too large"); return KR_E_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc_order_ = order_stream(h, s)) return rc_order_;
  if (h->fused_mlp && fused_mlp_supported(n_layers, dims, acts, in_pad)) {
    KR_CHECK_PTR(ws);
    if (ws == h->frag_ws) h->frag_ws = nullptr;  // (this call packs its own fragments into the workspace: whatever layout kr_train_epoch left there is gone)
    return fused_mlp_forward(Q, n_layers, dims, acts, W, b, x, out, ws, s);
  }
  MlpWs w = carve_ws(ws, n_layers, dims, Q);
  const float* in = x;
  int64_t ld_in = in_pad;
  for (int k = 0; k < n_layers; ++k) {
    const bool last = (k == n_layers - 1);
    GemmArgs g{};
    g.M = (int)Q; g.K = dims[k];
    g.A = in; g.sam = ld_in; g.sak = 1;
    g.B = W[k]; g.sbk = 1; g.sbn = dims[k];  // B(k,n) = W[n][k]  (nn.Linear layout)
    g.bias = b[k];
    g.act = acts[k];
    g.n_valid = dims[k + 1];
    g.k_chunk = g.K;
    if (last) {
      g.N = 32; g.C = out; g.ldc = 32; g.C2 = nullptr;
    } else {
      g.N = pad32(dims[k + 1]); g.C = w.Z[k]; g.C2 = w.Aact[k]; g.ldc = g.N;
    }
    rc = launch_gemm<EPI_BIAS_ACT>(g, true, false, 1, s);
    if (rc) return rc;
    if (!last) { in = w.Aact[k]; ld_in = g.N; }
  }
  return KR_OK;
}

int kr_mlp_backward(kr_handle* h, int64_t Q, int n_layers, const int32_t* dims, const int32_t* acts,
                    const float* const* W, const float* x, int in_pad, const float* dout, const void* ws,
                    float* const* dW, float* const* db, void* stream) {
  KR_CHECK_H(h);
  int rc = check_mlp_shape(n_layers, dims);
  if (rc) return rc;
  if (Q < 0) { set_error("Q < 0"); return KR_E_ARG; }
  KR_CHECK_PTR(acts); KR_CHECK_PTR(W); KR_CHECK_PTR(dW); KR_CHECK_PTR(db);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc_order_ = order_stream(h, s)) return rc_order_;
  for (int k = 0; k < n_layers; ++k) {
    KR_CHECK_PTR(dW[k]); KR_CHECK_PTR(db[k]);
    if (!h->grad_accumulate) {  // option "mlp_grad_accumulate": the caller keeps the buffers zeroed (kr_adam_step does)
      KR_HIP(hipMemsetAsync(dW[k], 0, sizeof(float) * dims[k] * dims[k + 1], s));
      KR_HIP(hipMemsetAsync(db[k], 0, sizeof(float) * dims[k + 1], s));
    }
  }
  if (Q == 0) return KR_OK;
  KR_CHECK_PTR(x); KR_CHECK_PTR(dout);
  if (n_layers > 1) KR_CHECK_PTR(ws);
  if (acts[n_layers - 1] == KR_ACT_NONE && h->fused_mlp && fused_mlp_supported(n_layers, dims, acts, in_pad))
    return fused_mlp_backward(Q, n_layers, dims, acts, W, x, dout, const_cast<void*>(ws), dW, db, s);
  MlpWs w = carve_ws(const_cast<void*>(ws), n_layers, dims, Q);
  const float* dz = dout;  // d loss / d (pre-activation of layer k); the last layer has no activation
  int64_t ld_dz = 32;
  if (acts[n_layers - 1] != KR_ACT_NONE) {
    set_error("an activation after the last layer is not supported by the training path");
    return KR_E_UNSUPPORTED;
  }
  for (int k = n_layers - 1; k >= 0; --k) {
    const float* a_prev = k == 0 ? x : w.Aact[k - 1];
    const int64_t ld_prev = k == 0 ? in_pad : pad32(dims[k]);
    // dW[k][o][i] = sum_r dz[r][o] * a_prev[r][i]   (contraction over the Q rows: split-K + atomics)
    {
      GemmArgs g{};
      g.M = dims[k + 1]; g.N = dims[k]; g.K = (int)Q;
      g.A = dz; g.sam = 1; g.sak = ld_dz;
      g.B = a_prev; g.sbk = ld_prev; g.sbn = 1;
      g.C = dW[k]; g.ldc = dims[k];
      g.n_valid = dims[k];
      int splits = (int)((Q + 2047) / 2048);
      if (splits > 1024) splits = 1024;
      g.k_chunk = (int)(((Q + splits - 1) / splits + GK - 1) / GK * GK);
      splits = (int)((Q + g.k_chunk - 1) / g.k_chunk);
      rc = launch_gemm<EPI_ATOMIC>(g, false, true, splits, s);
      if (rc) return rc;
    }
    {
      int gx = (int)((Q + 4095) / 4096);
      if (gx > 512) gx = 512;
      dim3 grid(gx, (dims[k + 1] + 63) / 64);
      hipLaunchKernelGGL(colsum_kernel, grid, dim3(64), 0, s, Q, dims[k + 1], (int)ld_dz, dz, db[k]);
      KR_HIP(hipGetLastError());
    }
    if (k > 0) {
      // dz_{k-1}[r][i] = (sum_o dz[r][o] W[k][o][i]) * act'(Z_{k-1}[r][i])
      GemmArgs g{};
      g.M = (int)Q; g.N = pad32(dims[k]); g.K = dims[k + 1];
      g.A = dz; g.sam = ld_dz; g.sak = 1;
      g.B = W[k]; g.sbk = dims[k]; g.sbn = 1;
      float* dst = (dz == w.d0) ? w.d1 : w.d0;
      g.C = dst; g.ldc = g.N;
      g.C2 = w.Z[k - 1];
      g.act = acts[k - 1];
      g.n_valid = dims[k];
      g.k_chunk = g.K;
      // B is read with n contiguous only inside the valid columns; guard handles n >= N
      GemmArgs g2 = g;
      g2.N = g.N;
      rc = launch_gemm<EPI_MUL_ACTGRAD>(g2, true, true, 1, s);
      if (rc) return rc;
      dz = dst;
      ld_dz = g.N;
    }
  }
  return KR_OK;
}

int kr_loss_fwd_bwd(kr_handle* h, int64_t S, int K, const float* base, const float* out, const float* target,
                    const int32_t* idx, double denom, float* pred, float* loss, float* dout, void* stream) {
  KR_CHECK_H(h);
  if (S < 0 || K < 0) { set_error("negative size"); return KR_E_ARG; }
  KR_CHECK_PTR(loss);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc_order_ = order_stream(h, s)) return rc_order_;
  KR_HIP(hipMemsetAsync(loss, 0, sizeof(float), s));
  if (S == 0 || K == 0) return KR_OK;
  KR_CHECK_PTR(base); KR_CHECK_PTR(out); KR_CHECK_PTR(target); KR_CHECK_PTR(idx); KR_CHECK_PTR(pred); KR_CHECK_PTR(dout);
  if (!(denom > 0)) { set_error("denom must be positive"); return KR_E_ARG; }
  const int64_t rows = S * K;
  int grid = (int)((rows + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(loss_kernel<false>, dim3(grid), dim3(256), 0, s, h->params.N, (float)h->derived.ds, S, K, base, out,
                     32, target, idx, (float)(1.0 / denom), pred, loss, dout, 32);
  KR_HIP(hipGetLastError());
  return KR_OK;
}

int kr_gather_targets(kr_handle* h, int64_t S, int K, const float* target, const int32_t* idx, float* rows,
                      void* stream) {
  KR_CHECK_H(h);
  if (S < 0 || K < 0) { set_error("negative size"); return KR_E_ARG; }
  if (S == 0 || K == 0) return KR_OK;
  KR_CHECK_PTR(target); KR_CHECK_PTR(idx); KR_CHECK_PTR(rows);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc_order_ = order_stream(h, s)) return rc_order_;
  const int64_t n = S * K * 25;
  int grid = (int)((n + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(gather_targets_kernel, dim3(grid), dim3(256), 0, s, h->params.N, S, K, target, idx, rows);
  KR_HIP(hipGetLastError());
  return KR_OK;
}

int kr_loss_rows_fwd_bwd(kr_handle* h, int64_t S, int K, const float* base, const float* out, const float* target_rows,
                         double denom, float* pred, float* loss, float* dout, void* stream) {
  KR_CHECK_H(h);
  if (S < 0 || K < 0) { set_error("negative size"); return KR_E_ARG; }
  KR_CHECK_PTR(loss);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc_order_ = order_stream(h, s)) return rc_order_;
  if (!h->grad_accumulate) KR_HIP(hipMemsetAsync(loss, 0, sizeof(float), s));
  if (S == 0 || K == 0) return KR_OK;
  KR_CHECK_PTR(base); KR_CHECK_PTR(out); KR_CHECK_PTR(target_rows); KR_CHECK_PTR(dout);
  if (!(denom > 0)) { set_error("denom must be positive"); return KR_E_ARG; }
  const int64_t rows = S * K;
  int grid = (int)((rows + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(loss_kernel<true>, dim3(grid), dim3(256), 0, s, h->params.N, (float)h->derived.ds, S, K, base, out,
                     32, target_rows, (const int32_t*)nullptr, (float)(1.0 / denom), pred, loss, dout, 32);
  KR_HIP(hipGetLastError());
  return KR_OK;
}


int kr_mlp_forward_loss(kr_handle* h, int64_t S, int K, int n_layers, const int32_t* dims, const int32_t* acts,
                        const float* const* W, const float* const* b, const float* x, int in_pad, const float* base,
                        const float* target_rows, double denom, float* out, float* loss, float* dout, void* ws,
                        void* stream) {
  KR_CHECK_H(h);
  if (S < 0 || K < 0) { set_error("negative size"); return KR_E_ARG; }
  const int64_t Q = S * K;
  int rc = check_mlp_shape(n_layers, dims);
  if (rc) return rc;
  KR_CHECK_PTR(loss);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc_order_ = order_stream(h, s)) return rc_order_;
  if (Q > 0 && h->fused_mlp && dims[n_layers] == 25 && acts && fused_mlp_supported(n_layers, dims, acts, in_pad)) {
    KR_CHECK_PTR(W); KR_CHECK_PTR(b); KR_CHECK_PTR(x); KR_CHECK_PTR(base); KR_CHECK_PTR(target_rows); KR_CHECK_PTR(dout);
    KR_CHECK_PTR(ws);
    if (!(denom > 0)) { set_error("denom must be positive"); return KR_E_ARG; }
    if (Q > (int64_t)1 << 30) { set_error("Q too large"); return KR_E_ARG; }
    if (!h->grad_accumulate) KR_HIP(hipMemsetAsync(loss, 0, sizeof(float), s));
    if (!h->loss_scratch) {
      KR_HIP(hipMalloc(&h->loss_scratch, 4096 * sizeof(float)));  // one partial per workgroup of the forward kernel
    }
    FusedLoss fl{base, target_rows, dout, loss, h->loss_scratch, (float)h->derived.ds, (float)(1.0 / denom), K};
    if (ws == h->frag_ws) h->frag_ws = nullptr;  // (this call packs its own fragments into the workspace: whatever layout kr_train_epoch left there is gone)
    return fused_mlp_forward(Q, n_layers, dims, acts, W, b, x, out, ws, s, &fl);
  }
  KR_CHECK_PTR(out);
  rc = kr_mlp_forward(h, Q, n_layers, dims, acts, W, b, x, in_pad, out, ws, stream);
  if (rc) return rc;
  return kr_loss_rows_fwd_bwd(h, S, K, base, out, target_rows, denom, nullptr, loss, dout, stream);
}

int kr_train_epoch(kr_handle* h, int64_t S, int K, int n_layers, const int32_t* dims, const int32_t* acts, float* params,
                   float* grads, float* exp_avg, float* exp_avg_sq, const float* lower, double* sched, const float* x,
                   int in_pad, const float* base, const float* target_rows, double denom, float* dout, void* ws,
                   double beta1, double beta2, double eps, double weight_decay, int64_t step, double factor,
                   int patience, double threshold, double min_lr, float* loss_log, int phase, int repack, void* stream) {
  KR_CHECK_H(h);
  if (S < 0 || K < 0) { set_error("negative size"); return KR_E_ARG; }
  const int64_t Q = S * K;
  int rc = check_mlp_shape(n_layers, dims);
  if (rc) return rc;
  KR_CHECK_PTR(acts);
  if (phase < 0 || phase > 2) { set_error("kr_train_epoch: phase must be 0, 1 or 2"); return KR_E_ARG; }
  if (step < 1) { set_error("kr_train_epoch: step counts from 1"); return KR_E_ARG; }
  if (!(factor > 0.0 && factor < 1.0) || patience < 0) { set_error("kr_train_epoch: need 0 < factor < 1, patience >= 0"); return KR_E_ARG; }
  if (Q <= 0 || Q > (int64_t)1 << 30) { set_error("kr_train_epoch: need 0 < S * K <= 2^30"); return KR_E_ARG; }
  if (!h->fused_mlp || dims[n_layers] != 25 || !fused_mlp_supported(n_layers, dims, acts, in_pad)) {
    set_error("kr_train_epoch serves the networks of the fused training kernels (28 -> H -> 25 and 28 -> H1 -> H2 -> 25 with "
              "H1, H2 <= 64, one activation, none after the last layer); use kr_mlp_forward_loss / kr_mlp_backward / "
              "kr_adam_plateau_step for others");
    return KR_E_UNSUPPORTED;
  }
  KR_CHECK_PTR(params); KR_CHECK_PTR(grads); KR_CHECK_PTR(exp_avg); KR_CHECK_PTR(exp_avg_sq); KR_CHECK_PTR(sched);
  KR_CHECK_PTR(ws);
  if (phase != 2) { KR_CHECK_PTR(x); KR_CHECK_PTR(base); KR_CHECK_PTR(target_rows); KR_CHECK_PTR(dout); }
  if (phase != 2 && !(denom > 0)) { set_error("denom must be positive"); return KR_E_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc_order_ = order_stream(h, s)) return rc_order_;
  if (!h->loss_scratch) KR_HIP(hipMalloc(&h->loss_scratch, 4096 * sizeof(float)));
  uint64_t net = 1469598103934665603ull;  // FNV-1a over the shape
  for (int k = 0; k <= n_layers; ++k) net = (net ^ (uint64_t)dims[k]) * 1099511628211ull;
  const bool current = !repack && h->frag_ws == ws && h->frag_params == params && h->frag_net == net;
  FusedEpoch E{};
  E.Q = Q; E.K = K; E.n_layers = n_layers; E.dims = dims; E.acts = acts;
  E.p = params; E.g = grads; E.m = exp_avg; E.v = exp_avg_sq; E.lower = lower; E.sched = sched;
  E.x = x; E.base = base; E.target_rows = target_rows; E.dout = dout; E.ws = ws; E.loss_scratch = h->loss_scratch;
  E.ds = (float)h->derived.ds; E.inv_denom = phase != 2 ? (float)(1.0 / denom) : 0.f;
  E.beta1 = beta1; E.beta2 = beta2; E.eps = eps; E.weight_decay = weight_decay;
  E.factor = factor; E.threshold = threshold; E.min_lr = min_lr; E.patience = patience;
  E.step = step; E.loss_log = loss_log; E.phase = phase;
  E.pack = !current;
  if (phase == 2 && !current) {
    // the update scatters into fragments that must already hold every other slot (padding included)
    set_error("kr_train_epoch: phase 2 needs the fragments of a phase 1 call with the same ws / params");
    return KR_E_ARG;
  }
  fused_set_debug(h->dbg);
  rc = fused_train_epoch(E, s);
  if (rc) { h->frag_ws = nullptr; return rc; }
  h->frag_ws = ws; h->frag_params = params; h->frag_net = net;
  return KR_OK;
}

int kr_train_epochs(kr_handle* h, int64_t n_epochs, int64_t S, int K, int n_layers, const int32_t* dims, const int32_t* acts,
                    float* params, float* grads, float* exp_avg, float* exp_avg_sq, const float* lower, double* sched,
                    const float* x, int in_pad, const float* base, const float* target_rows, double denom, float* dout, void* ws,
                    double beta1, double beta2, double eps, double weight_decay, int64_t step, double factor, int patience,
                    double threshold, double min_lr, float* loss_log, int repack, void* stream) {
  if (n_epochs < 0) { set_error("kr_train_epochs: n_epochs < 0"); return KR_E_ARG; }
  for (int64_t e = 0; e < n_epochs; ++e) {
    const int rc = kr_train_epoch(h, S, K, n_layers, dims, acts, params, grads, exp_avg, exp_avg_sq, lower, sched, x, in_pad, base,
                                  target_rows, denom, dout, ws, beta1, beta2, eps, weight_decay, step + e, factor, patience,
                                  threshold, min_lr, loss_log ? loss_log + e : nullptr, 0, e == 0 ? repack : 0, stream);
    if (rc) return rc;
  }
  return KR_OK;
}

int kr_adam_step(kr_handle* h, int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                 const float* lower, double lr, double beta1, double beta2, double eps, double weight_decay,
                 int64_t step, int64_t n_zero, void* stream) {
  KR_CHECK_H(h);
  if (n < 0 || n_zero < n || step < 1) { set_error("kr_adam_step: need n >= 0, n_zero >= n, step >= 1"); return KR_E_ARG; }
  if (n_zero == 0) return KR_OK;
  KR_CHECK_PTR(grads);
  if (n) { KR_CHECK_PTR(params); KR_CHECK_PTR(exp_avg); KR_CHECK_PTR(exp_avg_sq); }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc_order_ = order_stream(h, s)) return rc_order_;
  if (params == h->frag_params) h->frag_ws = nullptr;  // (kr_train_epoch's fragment copies no longer match)
  const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
  int grid = (int)((n_zero + 255) / 256);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(kr::adam_kernel, dim3(grid), dim3(256), 0, s, n, n_zero, params, grads, exp_avg, exp_avg_sq, lower,
                     (float)(lr / bc1), (float)beta1, (float)beta2, (float)(1.0 / std::sqrt(bc2)), (float)eps,
                     (float)weight_decay);
  KR_HIP(hipGetLastError());
  return KR_OK;
}
int kr_adam_plateau_step(kr_handle* h, int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                         const float* lower, double* sched, double beta1, double beta2, double eps,
                         double weight_decay, int64_t step, int64_t n_zero, int64_t loss_index, double factor,
                         int patience, double threshold, double min_lr, float* loss_log, void* stream) {
  KR_CHECK_H(h);
  if (n < 0 || n_zero < n || step < 1) { set_error("kr_adam_plateau_step: need n >= 0, n_zero >= n, step >= 1"); return KR_E_ARG; }
  if (loss_index < 0 || loss_index >= n_zero) { set_error("kr_adam_plateau_step: loss_index must lie in [0, n_zero)"); return KR_E_ARG; }
  if (!(factor > 0.0 && factor < 1.0) || patience < 0) { set_error("kr_adam_plateau_step: need 0 < factor < 1, patience >= 0"); return KR_E_ARG; }
  KR_CHECK_PTR(grads); KR_CHECK_PTR(sched);
  if (n) { KR_CHECK_PTR(params); KR_CHECK_PTR(exp_avg); KR_CHECK_PTR(exp_avg_sq); }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc_order_ = order_stream(h, s)) return rc_order_;
  if (params == h->frag_params) h->frag_ws = nullptr;  // (kr_train_epoch's fragment copies no longer match)
  const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
  int grid = (int)((n_zero + 255) / 256);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(kr::adam_plateau_kernel, dim3(grid), dim3(256), 0, s, n, n_zero, params, grads, exp_avg, exp_avg_sq,
                     lower, sched, (int)((step - 1) & 1), (float)(1.0 / bc1), (float)beta1, (float)beta2,
                     (float)(1.0 / std::sqrt(bc2)), (float)eps, (float)weight_decay, loss_index, factor, patience,
                     threshold, min_lr, loss_log);
  KR_HIP(hipGetLastError());
  return KR_OK;
}
}  // extern "C"
