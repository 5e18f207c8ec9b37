// kr_vjp.hip - derivatives of the physics of one grid point, y -> (y_s, z): what autograd propagates through
// CosseratRodTorch.ODE_parallel (reference cosserat_ode_torch.py:264-306, uncut graph) and through the op-by-op graph of
// CosseratRodTorch.getResidualEuler / ODE (:137-213, :325-367 - a graph that is CUT in two places, see below).
//
//   kr_ode_vjp_batch       (g_ys, g_z) -> J^T g with respect to y, yh, zh and the tendon force     (backward of ODE_parallel)
//   kr_ode_jacobian_batch  the 25 x 19 Jacobian d(y_s, z) / dy of every row                        (adjoint sweep of a14)
//
// Method: forward-mode automatic differentiation on dual numbers, in fp64 whatever the dtype of the arrays (the torch graph
// this replaces ran in fp64 too).  One thread evaluates the point map once with the tangent set to ONE input direction
// (19 + 19 + 6 + 3 = 47 of them per row), so a row takes 47 threads; thread (row, i) then owns component i of J^T g, or
// column i of the Jacobian - no reduction across threads, no atomics.  47 evaluations of ~400 flops per row is nothing:
// no caller of the reference differentiates with respect to these inputs (its training loops feed data), and the adjoint
// sweep handles N - 1 rows per call.
//
// `cut` reproduces the reference's graph where it differs from the function it evaluates: CosseratRodTorch.ODE assembles
// the quadratic part of the rotation matrix (:159-162) and the quaternion-rate matrix Omega(u) (:185-189) with
// torch.tensor([...]), i.e. as new leaves - through R, h is seen only via the factor 2 / (h . h), and h_s sees h but
// not u.  cut = 0 differentiates everything (ODE_parallel builds both with torch.stack, so its graph is complete).
#include "kr_internal.hpp"

namespace kr {

struct Dual {
  double v, d;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, fma(a.v, b.d, a.d * b.v)}; }
__device__ __forceinline__ Dual operator*(double s, Dual a) { return {s * a.v, s * a.d}; }
__device__ __forceinline__ Dual operator+(Dual a, double s) { return {a.v + s, a.d}; }
__device__ __forceinline__ Dual dual_rcp(Dual a) {
  const double r = 1.0 / a.v;
  return {r, -a.d * r * r};
}
__device__ __forceinline__ Dual dual_abs(Dual a) { return a.v < 0.0 ? Dual{-a.v, -a.d} : a; }  // (torch: sign(0) = 0; d = 0 then anyway for q = 0 ... kept as the value's sign)
__device__ __forceinline__ Dual detach(Dual a) { return {a.v, 0.0}; }

struct D3 {
  Dual x, y, z;
};
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator*(double s, D3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ D3 dcross(D3 a, D3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ D3 dmatvec(const double (&A)[9], D3 x) {
  return {A[0] * x.x + A[1] * x.y + A[2] * x.z, A[3] * x.x + A[4] * x.y + A[5] * x.z, A[6] * x.x + A[7] * x.y + A[8] * x.z};
}
struct DMat {
  Dual m[9];
};
__device__ __forceinline__ D3 rot(const DMat& R, D3 x) {
  return {R.m[0] * x.x + R.m[1] * x.y + R.m[2] * x.z, R.m[3] * x.x + R.m[4] * x.y + R.m[5] * x.z,
          R.m[6] * x.x + R.m[7] * x.y + R.m[8] * x.z};
}
__device__ __forceinline__ D3 rot_t(const DMat& R, D3 x) {
  return {R.m[0] * x.x + R.m[3] * x.y + R.m[6] * x.z, R.m[1] * x.x + R.m[4] * x.y + R.m[7] * x.z,
          R.m[2] * x.x + R.m[5] * x.y + R.m[8] * x.z};
}

// the point map on dual numbers: in[47] = [y(19), yh(19), zh(6), tf(3)], out[25] = [y_s(19), z(6)]
__device__ __forceinline__ void point_map_dual(const RodConst<double>& P, const Dual (&in)[47], bool cut, Dual (&out)[25]) {
  const Dual a = in[3], b = in[4], c = in[5], e = in[6];
  const D3 n{in[7], in[8], in[9]}, m{in[10], in[11], in[12]}, q{in[13], in[14], in[15]}, w{in[16], in[17], in[18]};
  const D3 yh_q{in[19 + 13], in[19 + 14], in[19 + 15]}, yh_w{in[19 + 16], in[19 + 17], in[19 + 18]};
  const D3 vh{in[38], in[39], in[40]}, uh{in[41], in[42], in[43]};
  const D3 tf{in[44], in[45], in[46]};
  // R = I + (2 / h.h) quad(h); cut: the entries of quad(h) are leaves
  const Dual aq = cut ? detach(a) : a, bq = cut ? detach(b) : b, cq = cut ? detach(c) : c, eq = cut ? detach(e) : e;
  const Dual s = 2.0 * dual_rcp(a * a + b * b + c * c + e * e);
  DMat R;
  R.m[0] = s * (-(cq * cq) - eq * eq) + 1.0; R.m[1] = s * (bq * cq - eq * aq);        R.m[2] = s * (bq * eq + cq * aq);
  R.m[3] = s * (bq * cq + eq * aq);          R.m[4] = s * (-(bq * bq) - eq * eq) + 1.0; R.m[5] = s * (cq * eq - bq * aq);
  R.m[6] = s * (bq * eq - cq * aq);          R.m[7] = s * (cq * eq + bq * aq);        R.m[8] = s * (-(bq * bq) - cq * cq) + 1.0;
  const D3 ksv{{P.Kse_vstar[0], 0.0}, {P.Kse_vstar[1], 0.0}, {P.Kse_vstar[2], 0.0}};
  const D3 v = dmatvec(P.Ksei, rot_t(R, n) + ksv - dmatvec(P.Bse, vh));
  const D3 u = dmatvec(P.Kbti, rot_t(R, m) - dmatvec(P.Bbt, uh));
  const D3 qt = P.c0 * q + yh_q, wt = P.c0 * w + yh_w, vt = P.c0 * v + vh, ut = P.c0 * u + uh;
  const D3 drag{P.C[0] * (q.x * dual_abs(q.x)), P.C[1] * (q.y * dual_abs(q.y)), P.C[2] * (q.z * dual_abs(q.z))};
  const D3 rag{{P.rhoAg[0], 0.0}, {P.rhoAg[1], 0.0}, {P.rhoAg[2], 0.0}};
  const D3 load = rag - rot(R, drag) + tf;
  const D3 ps = rot(R, v);
  const D3 ns = P.rhoA * rot(R, dcross(w, q) + qt) - load;
  const D3 ms = rot(R, dcross(w, dmatvec(P.rhoJ, w)) + dmatvec(P.rhoJ, wt)) - dcross(ps, n);
  const D3 qs = vt - dcross(u, q) + dcross(w, v);
  const D3 ws = ut - dcross(u, w);
  // h_s = 0.5 Omega(u) h; cut: the entries of Omega(u) are leaves
  const Dual u0 = cut ? detach(u.x) : u.x, u1 = cut ? detach(u.y) : u.y, u2 = cut ? detach(u.z) : u.z;
  out[0] = ps.x; out[1] = ps.y; out[2] = ps.z;
  out[3] = 0.5 * (-(u0 * b) - u1 * c - u2 * e);
  out[4] = 0.5 * (u0 * a + u2 * c - u1 * e);
  out[5] = 0.5 * (u1 * a - u2 * b + u0 * e);
  out[6] = 0.5 * (u2 * a + u1 * b - u0 * c);
  out[7] = ns.x; out[8] = ns.y; out[9] = ns.z;
  out[10] = ms.x; out[11] = ms.y; out[12] = ms.z;
  out[13] = qs.x; out[14] = qs.y; out[15] = qs.z;
  out[16] = ws.x; out[17] = ws.y; out[18] = ws.z;
  out[19] = v.x; out[20] = v.y; out[21] = v.z;
  out[22] = u.x; out[23] = u.y; out[24] = u.z;
}

template <typename T>
struct VjpArgs {
  int64_t Q;
  const T *y, *yh, *zh, *tf;      // [Q][19], [Q][19], [Q][6], [Q][3]
  const T *g_ys, *g_z;            // VJP: cotangents [Q][19], [Q][6]
  T *o_y, *o_yh, *o_zh, *o_tf;    // VJP: J^T g per input block (any may be null)
  T* jac;                         // Jacobian mode: [Q][25][19]
  int cut, ndir;                  // ndir: 47 (VJP) or 19 (Jacobian: directions of y only)
};

template <typename T, bool JAC>
__global__ __launch_bounds__(256) void ode_vjp_kernel(const RodConst<double> P, const VjpArgs<T> A) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = t / A.ndir;
  const int dir = (int)(t - row * A.ndir);
  if (row >= A.Q) return;
  Dual in[47], out[25];
#pragma unroll
  for (int k = 0; k < 19; ++k) in[k] = {(double)A.y[row * 19 + k], 0.0};
#pragma unroll
  for (int k = 0; k < 19; ++k) in[19 + k] = {(double)A.yh[row * 19 + k], 0.0};
#pragma unroll
  for (int k = 0; k < 6; ++k) in[38 + k] = {(double)A.zh[row * 6 + k], 0.0};
#pragma unroll
  for (int k = 0; k < 3; ++k) in[44 + k] = {(double)A.tf[row * 3 + k], 0.0};
#pragma unroll
  for (int k = 0; k < 47; ++k) in[k].d = k == dir ? 1.0 : 0.0;
  point_map_dual(P, in, A.cut != 0, out);
  if constexpr (JAC) {
#pragma unroll
    for (int o = 0; o < 25; ++o) A.jac[(row * 25 + o) * 19 + dir] = (T)out[o].d;
  } else {
    double acc = 0.0;
#pragma unroll
    for (int o = 0; o < 19; ++o) acc = fma((double)A.g_ys[row * 19 + o], out[o].d, acc);
#pragma unroll
    for (int o = 0; o < 6; ++o) acc = fma((double)A.g_z[row * 6 + o], out[19 + o].d, acc);
    if (dir < 19) { if (A.o_y) A.o_y[row * 19 + dir] = (T)acc; }
    else if (dir < 38) { if (A.o_yh) A.o_yh[row * 19 + dir - 19] = (T)acc; }
    else if (dir < 44) { if (A.o_zh) A.o_zh[row * 6 + dir - 38] = (T)acc; }
    else if (A.o_tf) A.o_tf[row * 3 + dir - 44] = (T)acc;
  }
}

template <typename T, bool JAC>
static int launch_vjp(kr_handle* h, const VjpArgs<T>& A, hipStream_t s) {
  const int64_t threads = A.Q * A.ndir;
  const int64_t grid = (threads + 255) / 256;
  if (grid > 0x7fffffff) { set_error("Q too large"); return KR_E_ARG; }
  hipLaunchKernelGGL((ode_vjp_kernel<T, JAC>), dim3((unsigned)grid), dim3(256), 0, s, h->cd, A);
  KR_HIP(hipGetLastError());
  return KR_OK;
}

}  // namespace kr

using namespace kr;

extern "C" {

int kr_ode_vjp_batch(kr_handle* h, int64_t Q, const void* y, const void* yh, const void* zh, const void* tf,
                     const void* g_dys, const void* g_z, int cut, void* g_y, void* g_yh, void* g_zh, void* g_tf,
                     int dtype, void* stream) {
  if (!h) { set_error("null handle"); return KR_E_ARG; }
  if (dtype != KR_F32 && dtype != KR_F64) { set_error("dtype must be KR_F32 or KR_F64"); return KR_E_ARG; }
  if (Q < 0) { set_error("Q < 0"); return KR_E_ARG; }
  if (Q == 0) return KR_OK;
  if (!y || !yh || !zh || !tf || !g_dys || !g_z) { set_error("null pointer argument"); return KR_E_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = order_stream(h, s)) return rc;
  if (dtype == KR_F32) {
    VjpArgs<float> A{Q, (const float*)y, (const float*)yh, (const float*)zh, (const float*)tf, (const float*)g_dys,
                     (const float*)g_z, (float*)g_y, (float*)g_yh, (float*)g_zh, (float*)g_tf, nullptr, cut, 47};
    return launch_vjp<float, false>(h, A, s);
  }
  VjpArgs<double> A{Q, (const double*)y, (const double*)yh, (const double*)zh, (const double*)tf, (const double*)g_dys,
                    (const double*)g_z, (double*)g_y, (double*)g_yh, (double*)g_zh, (double*)g_tf, nullptr, cut, 47};
  return launch_vjp<double, false>(h, A, s);
}

int kr_ode_jacobian_batch(kr_handle* h, int64_t Q, const void* y, const void* yh, const void* zh, const void* tf, int cut,
                          void* jac, int dtype, void* stream) {
  if (!h) { set_error("null handle"); return KR_E_ARG; }
  if (dtype != KR_F32 && dtype != KR_F64) { set_error("dtype must be KR_F32 or KR_F64"); return KR_E_ARG; }
  if (Q < 0) { set_error("Q < 0"); return KR_E_ARG; }
  if (Q == 0) return KR_OK;
  if (!y || !yh || !zh || !tf || !jac) { set_error("null pointer argument"); return KR_E_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = order_stream(h, s)) return rc;
  if (dtype == KR_F32) {
    VjpArgs<float> A{Q, (const float*)y, (const float*)yh, (const float*)zh, (const float*)tf, nullptr, nullptr, nullptr,
                     nullptr, nullptr, nullptr, (float*)jac, cut, 19};
    return launch_vjp<float, true>(h, A, s);
  }
  VjpArgs<double> A{Q, (const double*)y, (const double*)yh, (const double*)zh, (const double*)tf, nullptr, nullptr, nullptr,
                    nullptr, nullptr, nullptr, (double*)jac, cut, 19};
  return launch_vjp<double, true>(h, A, s);
}

}  // extern "C"
