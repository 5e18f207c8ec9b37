// kr_estimate.hip - full-state estimate from measured poses on the device:
// knode_cosserat_realworld/estimate_state.py:158-242 with its helpers compute_v_u (:48-95),
// compute_angular_velocities (:97-123) and compute_internal_forces_and_moments (:126-156).
//
// The reference evaluates everything per (time step, grid point) in Python loops, with one scipy.linalg.logm per
// segment and step.  Here four small fp64 kernels:
//   1. est_rates_kernel   thread per (t, i): linear velocity by the first-order np.gradient stencil, angular velocity
//                         from consecutive quaternions
//   2. est_local_kernel   thread per (t, i): accelerations (second-order np.gradient), strains v, u from the measured
//                         curve (closed-form logarithm of the relative rotation), distributed force n_s, the part of
//                         m_s that does not need n
//   3. est_wrench_kernel  thread per t: internal force and moment integrated from the tip backwards with the
//                         reference's step L / N and its literal `i != 9` skip (for N != 10 the recursion wraps into
//                         the tip entry exactly as NumPy's negative index does), R^T n and R^T m
//   4. est_strain_kernel  thread per grid point: the two-term recurrence over time of the re-estimated strains
//                         (BDF2 history of the PREVIOUS estimate, :222-230)
// Quirks kept: base x / y and the base quaternion's vector part zeroed, v[:, 0] = (0, 0, 1), tip n and m left 0.
#include "kr_internal.hpp"

namespace kr {

struct EstConst {
  double dt, L, c1, c2, rhoA;
  double rhoAg[3], C[3], rhoJ[9], Ksei[9], Kbti[9], Bse[9], Bbt[9], Ksev[3], tdirs[12];
  int N;
};

__device__ __forceinline__ void rot_of(const double* h, double (&R)[9]) {  // cosserat_ode.py:133-137
  const double a = h[0], b = h[1], c = h[2], d = h[3];
  const double s = 2.0 / (a * a + b * b + c * c + d * d);
  R[0] = 1 + s * (-c * c - d * d); R[1] = s * (b * c - d * a); R[2] = s * (b * d + c * a);
  R[3] = s * (b * c + d * a); R[4] = 1 + s * (-b * b - d * d); R[5] = s * (c * d - b * a);
  R[6] = s * (b * d - c * a); R[7] = s * (c * d + b * a); R[8] = 1 + s * (-b * b - c * c);
}
__device__ __forceinline__ void mv3(const double (&A)[9], const double* x, double* y) {
  for (int r = 0; r < 3; ++r) y[r] = A[3 * r] * x[0] + A[3 * r + 1] * x[1] + A[3 * r + 2] * x[2];
}
__device__ __forceinline__ void mtv3(const double (&A)[9], const double* x, double* y) {  // A^T x
  for (int r = 0; r < 3; ++r) y[r] = A[r] * x[0] + A[3 + r] * x[1] + A[6 + r] * x[2];
}
__device__ __forceinline__ void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
// position / quaternion of (t, i) as the estimate sees them: base x, y forced to zero (estimate_state.py:171)
__device__ __forceinline__ void pos_of(const double* data, int n, int64_t t, int i, double* p) {
  for (int r = 0; r < 3; ++r) p[r] = data[(t * 7 + r) * n + i];
  if (i == 0) { p[0] = 0.0; p[1] = 0.0; }
}
__device__ __forceinline__ void quat_of(const double* data, int n, int64_t t, int i, double* h) {
  for (int r = 0; r < 4; ++r) h[r] = data[(t * 7 + 3 + r) * n + i];
}

// vel[T][3][N], ang[T][3][N]
__global__ void est_rates_kernel(const EstConst P, int64_t T, const double* __restrict__ data, double* __restrict__ vel,
                                 double* __restrict__ ang) {
  const int N = P.N;
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= T * N) return;
  const int64_t t = id / N;
  const int i = (int)(id - t * N);
  {  // np.gradient(..., edge_order=1) along time
    double a[3], b[3];
    const int64_t ta = t == 0 ? 0 : t - 1, tb = t == T - 1 ? T - 1 : t + 1;
    pos_of(data, N, ta, i, a);
    pos_of(data, N, tb, i, b);
    const double inv = T > 1 ? 1.0 / ((double)(tb - ta) * P.dt) : 0.0;
    for (int r = 0; r < 3; ++r) vel[(t * 3 + r) * N + i] = (b[r] - a[r]) * inv;
  }
  {  // estimate_state.py:97-123; step 0 copies step 1
    const int64_t t2 = t == 0 ? 1 : t;
    double w[3] = {0, 0, 0};
    if (T > 1) {
      double q1[4], q2[4];
      quat_of(data, N, t2 - 1, i, q1);
      quat_of(data, N, t2, i, q2);
      const double f = 2.0 / P.dt;
      w[0] = f * (q1[0] * q2[1] - q1[1] * q2[0] - q1[2] * q2[3] + q1[3] * q2[2]);
      w[1] = f * (q1[0] * q2[2] + q1[1] * q2[3] - q1[2] * q2[0] - q1[3] * q2[1]);
      w[2] = f * (q1[0] * q2[3] - q1[1] * q2[2] + q1[2] * q2[1] - q1[3] * q2[0]);
    }
    for (int r = 0; r < 3; ++r) ang[(t * 3 + r) * N + i] = w[r];
  }
}

// np.gradient(f, dt, axis=0, edge_order=2) of an array [T][3][N] at (t, r, i)
__device__ __forceinline__ double grad2(const double* f, int64_t T, int N, int64_t t, int r, int i, double dt) {
  auto at = [&](int64_t tt) { return f[(tt * 3 + r) * N + i]; };
  if (T < 3) return T == 2 ? (at(1) - at(0)) / dt : 0.0;
  if (t == 0) return (-3.0 * at(0) + 4.0 * at(1) - at(2)) / (2.0 * dt);
  if (t == T - 1) return (3.0 * at(T - 1) - 4.0 * at(T - 2) + at(T - 3)) / (2.0 * dt);
  return (at(t + 1) - at(t - 1)) / (2.0 * dt);
}

// per (t, i): raw strains v, u; p_s; n_s; the n-independent part of m_s.  loc[T][N][15] = v(3) u(3) ps(3) ns(3) msA(3)
__global__ void est_local_kernel(const EstConst P, int64_t T, const double* __restrict__ data,
                                 const double* __restrict__ tens, const double* __restrict__ vel,
                                 const double* __restrict__ ang, double* __restrict__ loc) {
  const int N = P.N;
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= T * N) return;
  const int64_t t = id / N;
  const int i = (int)(id - t * N);
  const double dsarc = P.L / (N - 1);  // np.linspace(0, L, N) is uniform
  double h[4], R[9];
  quat_of(data, N, t, i, h);
  rot_of(h, R);
  // p_s and R_s use the segment (j, j + 1) with j = min(i, N - 2): the last grid point copies its neighbour (:59-61, :72-75)
  const int j = i < N - 1 ? i : N - 2;
  double pa[3], pb[3], ps[3];
  pos_of(data, N, t, j, pa);
  pos_of(data, N, t, j + 1, pb);
  for (int r = 0; r < 3; ++r) ps[r] = (pb[r] - pa[r]) / dsarc;
  double Rs[9];
  {
    double ha[4], hb[4], Ra[9], Rb[9], Rrel[9];
    quat_of(data, N, t, j, ha);
    quat_of(data, N, t, j + 1, hb);
    rot_of(ha, Ra);
    rot_of(hb, Rb);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Rrel[3 * r + c] = Rb[3 * r] * Ra[3 * c] + Rb[3 * r + 1] * Ra[3 * c + 1] + Rb[3 * r + 2] * Ra[3 * c + 2];
    // logarithm of a rotation matrix (what scipy.linalg.logm returns for it)
    double tr = (Rrel[0] + Rrel[4] + Rrel[8] - 1.0) * 0.5;
    tr = fmin(fmax(tr, -1.0), 1.0);
    const double th = acos(tr);
    const double f = th < 1e-6 ? 1.0 + th * th / 6.0 : th / sin(th);
    double lg[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) lg[3 * r + c] = f * 0.5 * (Rrel[3 * r + c] - Rrel[3 * c + r]) / dsarc;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Rs[3 * r + c] = Ra[3 * r] * lg[c] + Ra[3 * r + 1] * lg[3 + c] + Ra[3 * r + 2] * lg[6 + c];
  }
  double v[3], u[3];
  mtv3(R, ps, v);
  {
    // uhat = R^T R_s; u = (uhat[2][1], uhat[0][2], uhat[1][0])
    auto uh = [&](int r, int c) { return R[r] * Rs[c] + R[3 + r] * Rs[3 + c] + R[6 + r] * Rs[6 + c]; };
    u[0] = uh(2, 1); u[1] = uh(0, 2); u[2] = uh(1, 0);
  }
  if (i == 0) { v[0] = 0.0; v[1] = 0.0; v[2] = 1.0; }
  double q[3], w[3], qt[3], wt[3];
  for (int r = 0; r < 3; ++r) {
    q[r] = vel[(t * 3 + r) * N + i];
    w[r] = ang[(t * 3 + r) * N + i];
    qt[r] = grad2(vel, T, N, t, r, i, P.dt);
    wt[r] = grad2(ang, T, N, t, r, i, P.dt);
  }
  double tf[3] = {0, 0, 0};
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 3; ++r) tf[r] += tens[t * 4 + k] * P.tdirs[3 * k + r];
  double dq[3], drag[3], wxq[3], tmp[3], ns[3], msA[3];
  for (int r = 0; r < 3; ++r) dq[r] = P.C[r] * q[r] * fabs(q[r]);
  mv3(R, dq, drag);
  cross3(w, q, wxq);
  for (int r = 0; r < 3; ++r) wxq[r] += qt[r];
  mv3(R, wxq, tmp);
  for (int r = 0; r < 3; ++r) ns[r] = P.rhoA * tmp[r] - (P.rhoAg[r] - drag[r] + tf[r]);
  double Jw[3], Jwt[3], wxJw[3];
  mv3(P.rhoJ, w, Jw);
  mv3(P.rhoJ, wt, Jwt);
  cross3(w, Jw, wxJw);
  for (int r = 0; r < 3; ++r) wxJw[r] += Jwt[r];
  mv3(R, wxJw, msA);
  double* o = loc + (t * N + i) * 15;
  for (int r = 0; r < 3; ++r) { o[r] = v[r]; o[3 + r] = u[r]; o[6 + r] = ps[r]; o[9 + r] = ns[r]; o[12 + r] = msA[r]; }
}

// per t: n and m from the tip backwards (:141-154), written to est rows 7..12 (tip column stays 0); the constant
// parts of the strain recurrence: rv = Ksei (R^T n + Kse v*), ru = Kbti R^T m  -> rc[T][N][6]
__global__ void est_wrench_kernel(const EstConst P, int64_t T, const double* __restrict__ data,
                                  const double* __restrict__ loc, double* __restrict__ nm, double* __restrict__ est,
                                  double* __restrict__ rc) {
  const int N = P.N;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const double step = P.L / N;
  double* nv = nm + t * 6 * N;        // n[3][N]
  double* mvv = nv + 3 * N;            // m[3][N]
  for (int k = 0; k < 6 * N; ++k) nv[k] = 0.0;
  // Python: n[:, N - i - 2] = n[:, N - i - 1] - ns * step; index -1 (i = N - 1) is the tip entry
  for (int i = 0; i < N; ++i) {
    if (i == 9) continue;
    const int src = N - i - 1, dst = (N - i - 2 + N) % N;
    const double* ns = loc + (t * N + src) * 15 + 9;
    for (int r = 0; r < 3; ++r) nv[r * N + dst] = nv[r * N + src] - ns[r] * step;
  }
  for (int i = 0; i < N; ++i) {
    if (i == 9) continue;
    const int src = N - i - 1, dst = (N - i - 2 + N) % N;
    const double* ps = loc + (t * N + src) * 15 + 6;
    const double* msA = loc + (t * N + src) * 15 + 12;
    const double nn[3] = {nv[src], nv[N + src], nv[2 * N + src]};
    double pxn[3];
    cross3(ps, nn, pxn);
    for (int r = 0; r < 3; ++r) mvv[r * N + dst] = mvv[r * N + src] - (msA[r] - pxn[r]) * step;
  }
  for (int i = 0; i < N; ++i) {
    double n3[3] = {0, 0, 0}, m3[3] = {0, 0, 0};
    if (i < N - 1) {
      for (int r = 0; r < 3; ++r) { n3[r] = nv[r * N + i]; m3[r] = mvv[r * N + i]; }
    }
    for (int r = 0; r < 3; ++r) {
      est[(t * 25 + 7 + r) * N + i] = n3[r];
      est[(t * 25 + 10 + r) * N + i] = m3[r];
    }
    double h[4], R[9], a[3], b[3];
    quat_of(data, N, t, i, h);
    rot_of(h, R);
    mtv3(R, n3, a);
    for (int r = 0; r < 3; ++r) a[r] += P.Ksev[r];
    mv3(P.Ksei, a, b);
    double c[3], d[3];
    mtv3(R, m3, c);
    mv3(P.Kbti, c, d);
    double* o = rc + (t * N + i) * 6;
    for (int r = 0; r < 3; ++r) { o[r] = b[r]; o[3 + r] = d[r]; }
  }
}

// per grid point: v'_t = rv_t - Ksei Bse (c1 v_t + c2 v'_{t-1}),  u'_t = ru_t - Kbti Bbt (c1 u_t + c2 u'_{t-1});
// at t = 0 the "previous" strains are the raw ones of t = 0 (:190-192).  Also fills the remaining rows of est.
__global__ void est_strain_kernel(const EstConst P, int64_t T, const double* __restrict__ data,
                                  const double* __restrict__ vel, const double* __restrict__ ang,
                                  const double* __restrict__ loc, const double* __restrict__ rc, double* __restrict__ est) {
  const int N = P.N;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double KB_v[9], KB_u[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double sv = 0, su = 0;
      for (int k = 0; k < 3; ++k) { sv += P.Ksei[3 * r + k] * P.Bse[3 * k + c]; su += P.Kbti[3 * r + k] * P.Bbt[3 * k + c]; }
      KB_v[3 * r + c] = sv; KB_u[3 * r + c] = su;
    }
  double vp[3], up[3];
  for (int64_t t = 0; t < T; ++t) {
    const double* l = loc + (t * N + i) * 15;
    if (t == 0) for (int r = 0; r < 3; ++r) { vp[r] = l[r]; up[r] = l[3 + r]; }
    double vh[3], uh[3], kv[3], ku[3];
    for (int r = 0; r < 3; ++r) { vh[r] = P.c1 * l[r] + P.c2 * vp[r]; uh[r] = P.c1 * l[3 + r] + P.c2 * up[r]; }
    mv3(KB_v, vh, kv);
    mv3(KB_u, uh, ku);
    const double* o = rc + (t * N + i) * 6;
    for (int r = 0; r < 3; ++r) { vp[r] = o[r] - kv[r]; up[r] = o[3 + r] - ku[r]; }
    double p[3], h[4];
    pos_of(data, N, t, i, p);
    quat_of(data, N, t, i, h);
    if (i == 0) { h[1] = 0.0; h[2] = 0.0; h[3] = 0.0; }  // :236, after everything else has used the measured one
    for (int r = 0; r < 3; ++r) {
      est[(t * 25 + r) * N + i] = p[r];
      est[(t * 25 + 13 + r) * N + i] = vel[(t * 3 + r) * N + i];
      est[(t * 25 + 16 + r) * N + i] = ang[(t * 3 + r) * N + i];
      est[(t * 25 + 19 + r) * N + i] = vp[r];
      est[(t * 25 + 22 + r) * N + i] = up[r];
    }
    for (int r = 0; r < 4; ++r) est[(t * 25 + 3 + r) * N + i] = h[r];
  }
}

}  // namespace kr

using namespace kr;

extern "C" size_t kr_estimate_ws_bytes(int64_t T, int N) {
  if (T <= 0 || N <= 0) return 0;
  return (size_t)T * N * (3 + 3 + 15 + 6 + 6) * sizeof(double);
}

extern "C" int kr_estimate_state(kr_handle* h, int64_t T, const double* data, const double* tensions, double* est,
                                 void* ws, void* stream) {
  if (!h) { set_error("null handle"); return KR_E_ARG; }
  if (T < 0) { set_error("T < 0"); return KR_E_ARG; }
  if (T == 0) return KR_OK;
  if (!data || !tensions || !est || !ws) { set_error("null pointer argument"); return KR_E_ARG; }
  const int N = h->params.N;
  if (N < 2) { set_error("N must be >= 2"); return KR_E_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  EstConst P{};
  P.dt = h->params.del_t; P.L = h->params.L; P.c1 = h->derived.c1; P.c2 = h->derived.c2; P.rhoA = h->derived.rhoA;
  P.N = N;
  for (int i = 0; i < 3; ++i) { P.rhoAg[i] = h->derived.rhoAg[i]; P.C[i] = h->params.C[i]; P.Ksev[i] = h->derived.Kse_vstar[i]; }
  for (int i = 0; i < 9; ++i) {
    P.rhoJ[i] = h->derived.rhoJ[i]; P.Ksei[i] = h->derived.Kse_plus_c0_Bse_inv[i]; P.Kbti[i] = h->derived.Kbt_plus_c0_Bbt_inv[i];
    P.Bse[i] = h->params.Bse[i]; P.Bbt[i] = h->params.Bbt[i];
  }
  for (int i = 0; i < 12; ++i) P.tdirs[i] = h->params.tendon_dirs[i];
  double* w = static_cast<double*>(ws);
  double* vel = w; w += (size_t)T * 3 * N;
  double* ang = w; w += (size_t)T * 3 * N;
  double* loc = w; w += (size_t)T * N * 15;
  double* nm = w; w += (size_t)T * N * 6;
  double* rc = w;
  const int64_t pts = T * N;
  const int g1 = (int)((pts + 255) / 256);
  hipLaunchKernelGGL(est_rates_kernel, dim3(g1), dim3(256), 0, s, P, T, data, vel, ang);
  hipLaunchKernelGGL(est_local_kernel, dim3(g1), dim3(256), 0, s, P, T, data, tensions, vel, ang, loc);
  hipLaunchKernelGGL(est_wrench_kernel, dim3((int)((T + 63) / 64)), dim3(64), 0, s, P, T, data, loc, nm, est, rc);
  hipLaunchKernelGGL(est_strain_kernel, dim3((N + 63) / 64), dim3(64), 0, s, P, T, data, vel, ang, loc, rc, est);
  KR_HIP(hipGetLastError());
  return KR_OK;
}
