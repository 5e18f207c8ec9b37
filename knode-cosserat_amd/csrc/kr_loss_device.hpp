// kr_loss_device.hpp - the four-term one-step-ahead loss of the KNODE training step for ONE row (window step, key
// point): prediction from the parameter-free part and the MLP output, loss terms, gradient with respect to the MLP
// output (physics_train.py:250-259 / 345-352, Utils/transformations.py:3-31).  Shared by loss_kernel (kr_train.hip)
// and the epilogue of the fused forward kernel (kr_mlp_fused.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace kr {

// quaternion_to_euler of Utils/transformations.py:3-31 and its Jacobian-transpose product.
// Short forms (round 5; the epilogue of the fused forward kernel spent 7.4 k of a row block's 28 k cycles here, most of it in
// the library's atan2f / asinf with their special-case ladders and in seven IEEE divisions per row):
//   1 / |q| by v_rsq_f32 + one Newton step; every other quotient by v_rcp_f32 (1 ulp);
//   atan2(a, b) = minimax polynomial of t = min(|a|, |b|) / max(|a|, |b|) (Abramowitz & Stegun 4.4.49, |error| <= 2e-8 on
//   [0, 1], evaluated as t + t (t^2 P(t^2)) so that small angles keep their RELATIVE accuracy) + the octant fix-ups;
//   asin(s) = s + s z P(z), z = s^2 for |s| <= 1/2, pi/2 - 2 (r + r z P(z)), z = (1 - |s|) / 2, r = sqrt(z) otherwise
//   (the Cephes asinf polynomial), select-free.
// Both are within 2.6e-7 relative of the exact functions over their whole range (checked in fp32 emulation against
// numpy), i.e. the same two ulp the library forms give; the reference's clamp of the asin argument is kept.
__device__ __forceinline__ float atan2_short(float a, float b) {
  const float aa = fabsf(a), ab = fabsf(b);
  const float mx = fmaxf(aa, ab), mn = fminf(aa, ab);
  const float t = mn * __builtin_amdgcn_rcpf(fmaxf(mx, 1e-37f));  // (0, 0) -> 0 like atan2
  const float z = t * t;
  float p = 0.0028662257f;
  p = fmaf(p, z, -0.0161657367f);
  p = fmaf(p, z, 0.0429096138f);
  p = fmaf(p, z, -0.0752896400f);
  p = fmaf(p, z, 0.1065626393f);
  p = fmaf(p, z, -0.1420889944f);
  p = fmaf(p, z, 0.1999355085f);
  p = fmaf(p, z, -0.3333314528f);
  float r = fmaf(t, p * z, t);
  r = aa > ab ? 1.57079632679489662f - r : r;
  r = b < 0.f ? 3.14159265358979324f - r : r;
  return copysignf(r, a);
}
__device__ __forceinline__ float asin_short(float s) {
  const float a = fabsf(s);
  const bool big = a > 0.5f;
  const float z = big ? 0.5f * (1.f - a) : a * a;
  const float r = big ? __builtin_amdgcn_sqrtf(z) : a;
  float p = 4.2163199048e-2f;
  p = fmaf(p, z, 2.4181311049e-2f);
  p = fmaf(p, z, 4.5470025998e-2f);
  p = fmaf(p, z, 7.4953002686e-2f);
  p = fmaf(p, z, 1.6666752422e-1f);
  const float v = fmaf(p * z, r, r);
  return copysignf(big ? 1.57079632679489662f - 2.f * v : v, s);
}
// 1 / sqrt(x): v_rsq_f32 (1 ulp) + one Newton step
__device__ __forceinline__ float rsqrt_nr(float x) {
  const float r = __builtin_amdgcn_rsqf(x);
  return r * fmaf(-0.5f * x * r, r, 1.5f);
}
// nq[0..3]: the normalised quaternion (w, x, y, z), nq[4]: 1 / |q| - what q2e_vjp_n needs again
__device__ __forceinline__ void q2e_n(const float q[4], float e[3], float (&nq)[5]) {
  const float inv = rsqrt_nr(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
  nq[0] = w; nq[1] = x; nq[2] = y; nq[3] = z; nq[4] = inv;
  e[0] = atan2_short(2.f * (w * y + x * z), 1.f - 2.f * (y * y + z * z));
  e[1] = asin_short(fminf(fmaxf(2.f * (w * z - x * y), -1.f), 1.f));
  e[2] = atan2_short(2.f * (w * x + y * z), 1.f - 2.f * (x * x + z * z));
}
__device__ __forceinline__ void q2e(const float q[4], float e[3]) {
  float nq[5];
  q2e_n(q, e, nq);
}
__device__ __forceinline__ void q2e_vjp_n(const float (&nq)[5], const float ge[3], float gq[4]) {
  const float w = nq[0], x = nq[1], y = nq[2], z = nq[3], inv = nq[4];
  float gn[4] = {0.f, 0.f, 0.f, 0.f};  // gradient w.r.t. the normalised quaternion (w, x, y, z)
  {  // roll = atan2(a, b)
    const float a = 2.f * (w * y + x * z), b = 1.f - 2.f * (y * y + z * z);
    const float s_ = ge[0] * __builtin_amdgcn_rcpf(a * a + b * b);
    const float ga = s_ * b, gb = -s_ * a;
    gn[0] += ga * 2.f * y; gn[1] += ga * 2.f * z; gn[2] += ga * 2.f * w - gb * 4.f * y; gn[3] += ga * 2.f * x - gb * 4.f * z;
  }
  {  // pitch = asin(clamp(s))
    const float s = 2.f * (w * z - x * y);
    if (s >= -1.f && s <= 1.f) {
      const float gs = ge[1] * __builtin_amdgcn_rsqf(1.f - s * s);
      gn[0] += gs * 2.f * z; gn[1] -= gs * 2.f * y; gn[2] -= gs * 2.f * x; gn[3] += gs * 2.f * w;
    }
  }
  {  // yaw = atan2(c, d)
    const float c = 2.f * (w * x + y * z), d = 1.f - 2.f * (x * x + z * z);
    const float s_ = ge[2] * __builtin_amdgcn_rcpf(c * c + d * d);
    const float gc = s_ * d, gd = -s_ * c;
    gn[0] += gc * 2.f * x; gn[1] += gc * 2.f * w - gd * 4.f * x; gn[2] += gc * 2.f * z; gn[3] += gc * 2.f * y - gd * 4.f * z;
  }
  const float dot = gn[0] * w + gn[1] * x + gn[2] * y + gn[3] * z;
  gq[0] = (gn[0] - w * dot) * inv;
  gq[1] = (gn[1] - x * dot) * inv;
  gq[2] = (gn[2] - y * dot) * inv;
  gq[3] = (gn[3] - z * dot) * inv;
}
__device__ __forceinline__ void q2e_vjp(const float q[4], const float ge[3], float gq[4]) {
  const float inv = rsqrt_nr(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float nq[5] = {q[0] * inv, q[1] * inv, q[2] * inv, q[3] * inv, inv};
  q2e_vjp_n(nq, ge, gq);
}

// weights of the four nn.MSELoss(mean) terms: each is a mean over (rows of the block) x K
struct LossWeights {
  float w_p, w_r, w_e, w_z;
};
__host__ __device__ inline LossWeights loss_weights(float inv_denom, int K) {
  return {inv_denom / (3.f * K), inv_denom / (12.f * K), inv_denom / (3.f * K), inv_denom / (6.f * K)};
}

// p[25]: prediction (in), tgv[25]: target values, g[25]: d loss / d p (out); returns the row's loss.
// ep, et: quaternion_to_euler of p[3:7] and tgv[3:7] (the caller may have had other lanes compute them).
// nq: the normalised predicted quaternion and 1 / |q| as q2e_n left them (the caller's lane has just computed them).
__device__ __forceinline__ float loss_row_angles(const float (&p)[25], const float (&tgv)[25], const float (&ep)[3],
                                                 const float (&et)[3], const float (&nq)[5], const LossWeights& w, float (&g)[25]) {
  float part = 0.f;
  // positions, physics_train.py:252-253
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float d = p[r] - tgv[r];
    part += w.w_p * d * d;
    g[r] = 2.f * w.w_p * d;
  }
  // n m q w, :254-255
#pragma unroll
  for (int r = 7; r < 19; ++r) {
    const float d = p[r] - tgv[r];
    part += w.w_r * d * d;
    g[r] = 2.f * w.w_r * d;
  }
  // Euler angles of the quaternion, :256-257
  {
    float ge[3], gq[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = ep[c] - et[c];
      part += w.w_e * d * d;
      ge[c] = 2.f * w.w_e * d;
    }
    q2e_vjp_n(nq, ge, gq);
    g[3] = gq[0]; g[4] = gq[1]; g[5] = gq[2]; g[6] = gq[3];
  }
  // z rows against the column before the key point, :258-259
#pragma unroll
  for (int r = 19; r < 25; ++r) {
    const float d = p[r] - tgv[r];
    part += w.w_z * d * d;
    g[r] = 2.f * w.w_z * d;
  }
  return part;
}
__device__ __forceinline__ float loss_row(const float (&p)[25], const float (&tgv)[25], const LossWeights& w, float (&g)[25]) {
  float qp[4] = {p[3], p[4], p[5], p[6]};
  float qt[4] = {tgv[3], tgv[4], tgv[5], tgv[6]};
  float ep[3], et[3], nq[5];
  q2e_n(qp, ep, nq);
  q2e(qt, et);
  return loss_row_angles(p, tgv, ep, et, nq, w, g);
}

}  // namespace kr
