/* cosserat_oracle_c.c - scalar C restatement of the rod integrator (TEST INFRASTRUCTURE, never linked into the
 * product: only tests/ and bench.py's cpu_baseline leg use it).
 *
 * Second, independent restatement beside oracle/cosserat_oracle.py, physics only (no MLP):
 *   ode()            cosserat_ode.py:114-166   per-segment spatial derivative
 *   residual_euler() cosserat_ode.py:188-213   explicit-Euler shooting sweep, mutating y, z; z[:, N-1] never written
 *   simulate()       knode.py:55-102           BDF2 time loop, straight initial rod, G warm-started
 * The shooting unknowns are found by Newton with a forward-difference Jacobian converged to 1e-12 (the
 * reference calls MINPACK hybrd with xtol 1.5e-8; same root - pinned against the reference's own trajectories
 * by tests/test_oracle_golden.py::test_c_oracle_*).  Plain dense 3x3 arithmetic, no shortcuts, so that it
 * reads like the NumPy code it restates.
 *   gcc -O2 -shared -fPIC -o oracle/lib/liboracle_c.so oracle/cosserat_oracle_c.c -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  double L, E, r, rho, del_t;
  int N;
  double vstar[3], g[3], Bse[9], Bbt[9], C[3], F_tip[3], M_tip[3], tendon_dirs[12], p0[3], h0[4], q0[3], w0[3];
} orc_params;

typedef struct {
  int N;
  double ds, c0, c1, c2, rhoA;
  double Kse_inv[9], Kbt_inv[9], Kse_vstar[3], rhoAg[3], rhoJ[9], Bse[9], Bbt[9], C[3];
  const orc_params* P;
} derived;

static void matvec(const double* A, const double* x, double* y) {
  for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}
static void matTvec(const double* A, const double* x, double* y) {
  for (int i = 0; i < 3; ++i) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}
static void cross(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static int inv3(const double* A, double* I) {
  const double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
  if (det == 0.0) return 1;
  const double d = 1.0 / det;
  I[0] = (A[4] * A[8] - A[5] * A[7]) * d; I[1] = (A[2] * A[7] - A[1] * A[8]) * d; I[2] = (A[1] * A[5] - A[2] * A[4]) * d;
  I[3] = (A[5] * A[6] - A[3] * A[8]) * d; I[4] = (A[0] * A[8] - A[2] * A[6]) * d; I[5] = (A[2] * A[3] - A[0] * A[5]) * d;
  I[6] = (A[3] * A[7] - A[4] * A[6]) * d; I[7] = (A[1] * A[6] - A[0] * A[7]) * d; I[8] = (A[0] * A[4] - A[1] * A[3]) * d;
  return 0;
}

/* compute_intermediate_terms, cosserat_ode.py:58-78 */
static int derive(const orc_params* P, derived* D) {
  const double pi = 3.14159265358979323846;
  memset(D, 0, sizeof(*D));
  D->P = P;
  D->N = P->N;
  const double r2 = P->r * P->r, A = pi * r2, G = P->E / (2 * (1 + 0.3)), Ixx = pi * r2 * r2 / 4;
  D->ds = P->L / (P->N - 1);
  D->c0 = 1.5 / P->del_t; D->c1 = -2.0 / P->del_t; D->c2 = 0.5 / P->del_t;
  double Kse[9] = {G * A, 0, 0, 0, G * A, 0, 0, 0, P->E * A};
  double Kbt[9] = {P->E * Ixx, 0, 0, 0, P->E * Ixx, 0, 0, 0, G * 2 * Ixx};
  double t[9];
  for (int i = 0; i < 9; ++i) t[i] = Kse[i] + D->c0 * P->Bse[i];
  if (inv3(t, D->Kse_inv)) return 1;
  for (int i = 0; i < 9; ++i) t[i] = Kbt[i] + D->c0 * P->Bbt[i];
  if (inv3(t, D->Kbt_inv)) return 1;
  matvec(Kse, P->vstar, D->Kse_vstar);
  D->rhoA = P->rho * A;
  for (int i = 0; i < 3; ++i) { D->rhoAg[i] = D->rhoA * P->g[i]; D->C[i] = P->C[i]; }
  D->rhoJ[0] = P->rho * Ixx; D->rhoJ[4] = P->rho * Ixx; D->rhoJ[8] = P->rho * 2 * Ixx;
  memcpy(D->Bse, P->Bse, sizeof(D->Bse));
  memcpy(D->Bbt, P->Bbt, sizeof(D->Bbt));
  return 0;
}

/* cosserat_ode.py:114-166.  y[19] = p h n m q w; yh[19], zh[6] history terms; ys[19], z[6] out */
static void ode(const derived* D, const double* y, const double* yh, const double* zh, const double* tf, double* ys, double* z) {
  const double *h = y + 3, *n = y + 7, *m = y + 10, *q = y + 13, *w = y + 16, *vh = zh, *uh = zh + 3;
  const double a = h[0], b = h[1], c = h[2], d = h[3], s = 2.0 / (a * a + b * b + c * c + d * d);
  const double R[9] = {1 + s * (-c * c - d * d), s * (b * c - d * a), s * (b * d + c * a),
                       s * (b * c + d * a), 1 + s * (-b * b - d * d), s * (c * d - b * a),
                       s * (b * d - c * a), s * (c * d + b * a), 1 + s * (-b * b - c * c)};
  double t1[3], t2[3], t3[3], v[3], u[3];
  matTvec(R, n, t1); matvec(D->Bse, vh, t2);
  for (int i = 0; i < 3; ++i) t3[i] = t1[i] + D->Kse_vstar[i] - t2[i];
  matvec(D->Kse_inv, t3, v);                      /* :140 */
  matTvec(R, m, t1); matvec(D->Bbt, uh, t2);
  for (int i = 0; i < 3; ++i) t3[i] = t1[i] - t2[i];
  matvec(D->Kbt_inv, t3, u);                      /* :141 */
  double qt[3], wt[3], vt[3], ut[3];
  for (int i = 0; i < 3; ++i) {                   /* :146-148 */
    qt[i] = D->c0 * q[i] + yh[13 + i]; wt[i] = D->c0 * w[i] + yh[16 + i];
    vt[i] = D->c0 * v[i] + vh[i]; ut[i] = D->c0 * u[i] + uh[i];
  }
  double drag[3], Rdrag[3], f[3];
  for (int i = 0; i < 3; ++i) drag[i] = D->C[i] * q[i] * fabs(q[i]);
  matvec(R, drag, Rdrag);
  for (int i = 0; i < 3; ++i) f[i] = D->rhoAg[i] - Rdrag[i] + tf[i];   /* :151 */
  double *ps = ys, *hs = ys + 3, *ns = ys + 7, *ms = ys + 10, *qs = ys + 13, *ws = ys + 16;
  matvec(R, v, ps);                                                    /* :154 */
  cross(w, q, t1);
  for (int i = 0; i < 3; ++i) t1[i] += qt[i];
  matvec(R, t1, t2);
  for (int i = 0; i < 3; ++i) ns[i] = D->rhoA * t2[i] - f[i];          /* :155 */
  double Jw[3], Jwt[3];
  matvec(D->rhoJ, w, Jw); matvec(D->rhoJ, wt, Jwt);
  cross(w, Jw, t1);
  for (int i = 0; i < 3; ++i) t1[i] += Jwt[i];
  matvec(R, t1, t2);
  cross(ps, n, t3);
  for (int i = 0; i < 3; ++i) ms[i] = t2[i] - t3[i];                   /* :156 */
  cross(u, q, t1); cross(w, v, t2);
  for (int i = 0; i < 3; ++i) qs[i] = vt[i] - t1[i] + t2[i];           /* :157 */
  cross(u, w, t1);
  for (int i = 0; i < 3; ++i) ws[i] = ut[i] - t1[i];                   /* :158 */
  hs[0] = 0.5 * (-u[0] * h[1] - u[1] * h[2] - u[2] * h[3]);            /* :161-165 */
  hs[1] = 0.5 * (u[0] * h[0] + u[2] * h[2] - u[1] * h[3]);
  hs[2] = 0.5 * (u[1] * h[0] - u[2] * h[1] + u[0] * h[3]);
  hs[3] = 0.5 * (u[2] * h[0] + u[1] * h[1] - u[0] * h[2]);
  for (int i = 0; i < 3; ++i) { z[i] = v[i]; z[3 + i] = u[i]; }
}

/* state arrays are point-major here: y[j*19 + r], z[j*6 + r] (the reference is feature-major; the wrapper transposes) */
static void residual_euler(const derived* D, const double* G, double* y, double* z, const double* yh, const double* zh,
                           const double* tf, double* res) {
  const orc_params* P = D->P;
  double* y0 = y;                                   /* cosserat_ode.py:194 */
  for (int i = 0; i < 3; ++i) y0[i] = P->p0[i];
  for (int i = 0; i < 4; ++i) y0[3 + i] = P->h0[i];
  for (int i = 0; i < 6; ++i) y0[7 + i] = G[i];
  for (int i = 0; i < 3; ++i) { y0[13 + i] = P->q0[i]; y0[16 + i] = P->w0[i]; }
  for (int j = 0; j < D->N - 1; ++j) {              /* :198-201 */
    double ys[19];
    ode(D, y + 19 * j, yh + 19 * j, zh + 6 * j, tf, ys, z + 6 * j);
    for (int r = 0; r < 19; ++r) y[19 * (j + 1) + r] = y[19 * j + r] + D->ds * ys[r];
  }
  const double* yl = y + 19 * (D->N - 1);           /* :204-207 */
  for (int i = 0; i < 3; ++i) { res[i] = P->F_tip[i] - yl[7 + i]; res[3 + i] = P->M_tip[i] - yl[10 + i]; }
}

static int solve6(double A[6][7], double* x) {
  for (int k = 0; k < 6; ++k) {
    int p = k;
    for (int i = k + 1; i < 6; ++i) if (fabs(A[i][k]) > fabs(A[p][k])) p = i;
    if (A[p][k] == 0.0) return 1;
    if (p != k) for (int c = 0; c < 7; ++c) { double t = A[k][c]; A[k][c] = A[p][c]; A[p][c] = t; }
    for (int i = k + 1; i < 6; ++i) {
      const double f = A[i][k] / A[k][k];
      for (int c = k; c < 7; ++c) A[i][c] -= f * A[k][c];
    }
  }
  for (int k = 5; k >= 0; --k) {
    double s = A[k][6];
    for (int c = k + 1; c < 6; ++c) s -= A[k][c] * x[c];
    x[k] = s / A[k][k];
  }
  return 0;
}

/* knode.py:55-102 with a Newton solve per step.  ctl[T][4]; tip_out[T][3] = tip after each solved step;
 * traj_out (nullable) [T+1][25][N] in the reference's row order, entry 0 = initial state.
 * returns the number of steps whose Newton iteration did not converge (or -1 for bad parameters). */
int orc_simulate(const orc_params* P, int T, const double* ctl, double* tip_out, double* traj_out) {
  derived D;
  if (P->N < 2 || derive(P, &D)) return -1;
  const int N = P->N;
  double* y = calloc((size_t)19 * N, sizeof(double));
  double* z = calloc((size_t)6 * N, sizeof(double));
  double* yp = malloc(sizeof(double) * 19 * N), *zp = malloc(sizeof(double) * 6 * N);
  double* yh = malloc(sizeof(double) * 19 * N), *zh = malloc(sizeof(double) * 6 * N);
  for (int j = 0; j < N; ++j) {                     /* knode.py:58-64 */
    y[19 * j + 2] = P->L * j / (N - 1);
    y[19 * j + 3] = 1.0;
    z[6 * j + 2] = 1.0;
  }
  memcpy(yp, y, sizeof(double) * 19 * N);
  memcpy(zp, z, sizeof(double) * 6 * N);
  double G[6] = {0, 0, 0, 0, 0, 0};
  int bad = 0;
  for (int t = 0; t <= T; ++t) {
    if (traj_out) {
      double* o = traj_out + (size_t)t * 25 * N;
      for (int j = 0; j < N; ++j) {
        for (int r = 0; r < 19; ++r) o[r * N + j] = y[19 * j + r];
        for (int r = 0; r < 6; ++r) o[(19 + r) * N + j] = z[6 * j + r];
      }
    }
    if (t == T) break;
    for (int i = 0; i < 19 * N; ++i) yh[i] = D.c1 * y[i] + D.c2 * yp[i];    /* knode.py:74-75 */
    for (int i = 0; i < 6 * N; ++i) zh[i] = D.c1 * z[i] + D.c2 * zp[i];
    memcpy(yp, y, sizeof(double) * 19 * N);
    memcpy(zp, z, sizeof(double) * 6 * N);
    double tf[3] = {0, 0, 0};                       /* cosserat_ode.py:195 */
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 3; ++i) tf[i] += ctl[4 * t + k] * P->tendon_dirs[3 * k + i];
    int ok = 0;
    for (int it = 0; it < 50 && !ok; ++it) {
      double r0[6], A[6][7], d[6];
      residual_euler(&D, G, y, z, yh, zh, tf, r0);
      for (int c = 0; c < 6; ++c) {
        const double e = 1e-7 * fmax(fabs(G[c]), 1.0);
        double Gp[6], rc[6];
        memcpy(Gp, G, sizeof(Gp));
        Gp[c] += e;
        residual_euler(&D, Gp, y, z, yh, zh, tf, rc);
        for (int i = 0; i < 6; ++i) A[i][c] = (rc[i] - r0[i]) / e;
      }
      for (int i = 0; i < 6; ++i) A[i][6] = r0[i];
      if (solve6(A, d)) break;
      double dm = 0, gm = 1;
      for (int i = 0; i < 6; ++i) { dm = fmax(dm, fabs(d[i])); gm = fmax(gm, fabs(G[i])); }
      if (!(dm <= 1e300)) break;
      if (dm <= 1e-12 * gm) { ok = 1; break; }
      for (int i = 0; i < 6; ++i) G[i] -= d[i];
    }
    double rf[6];
    residual_euler(&D, G, y, z, yh, zh, tf, rf);    /* the sweep at the accepted G is the stored state */
    if (!ok) ++bad;
    for (int i = 0; i < 3; ++i) tip_out[3 * t + i] = y[19 * (N - 1) + i];
  }
  free(y); free(z); free(yp); free(zp); free(yh); free(zh);
  return bad;
}

#ifdef ORC_SELFTEST
/* `make asan`: the same file as a standalone program under -fsanitize=address,undefined (SURVEY section 5 asks for a
 * sanitizer build of the CPU restatement).  setup_robot(None) parameters (knode.py:11-20), per-rod sine tensions, N = 10,
 * 20, 100, 400 with and without the trajectory buffer; exit code 1 on a non-finite tip or an unconverged step. */
#include <stdio.h>
int main(void) {
  const int Ns[4] = {10, 20, 100, 400}, Ts[4] = {40, 40, 20, 8};
  const double pi = 3.14159265358979323846;
  int fail = 0;
  for (int c = 0; c < 4; ++c) {
    orc_params P;
    memset(&P, 0, sizeof(P));
    P.L = 0.635; P.E = 2.757903e9; P.r = 0.003175; P.rho = 1411.6751; P.del_t = 0.05; P.N = Ns[c];
    P.vstar[2] = 1.0; P.g[2] = -9.81; P.h0[0] = 1.0;
    P.Bbt[0] = P.Bbt[4] = P.Bbt[8] = 3e-2;
    P.C[0] = P.C[1] = P.C[2] = 1e-4;
    for (int k = 0; k < 4; ++k) {
      const double th = pi / 4 + k * pi / 2;
      P.tendon_dirs[3 * k] = cos(th); P.tendon_dirs[3 * k + 1] = sin(th);
    }
    const int T = Ts[c];
    double* ctl = (double*)malloc(sizeof(double) * 4 * T);
    double* tip = (double*)malloc(sizeof(double) * 3 * T);
    double* traj = (double*)malloc(sizeof(double) * 25 * (size_t)Ns[c] * (T + 1));
    for (int t = 0; t < T; ++t)
      for (int k = 0; k < 4; ++k) ctl[4 * t + k] = 6.0 + sin(2 * pi * (t + 1) * P.del_t / 1.3 + 0.4 + k * pi / 2);
    for (int with_traj = 0; with_traj < 2; ++with_traj) {
      const int bad = orc_simulate(&P, T, ctl, tip, with_traj ? traj : NULL);
      int finite = 1;
      for (int i = 0; i < 3 * T; ++i) finite &= isfinite(tip[i]) != 0;
      printf("N=%d T=%d traj=%d unconverged=%d tip=(%.6f %.6f %.6f)\n", Ns[c], T, with_traj, bad, tip[3 * T - 3], tip[3 * T - 2], tip[3 * T - 1]);
      if (bad != 0 || !finite) fail = 1;
    }
    free(ctl); free(tip); free(traj);
  }
  return fail;
}
#endif
