// kr_loss_device.hpp - the four-term one-step-ahead loss of the KNODE training step for ONE row (window step, key
// point): prediction from the parameter-free part and the MLP output, loss terms, gradient with respect to the MLP
// output (physics_train.py:250-259 / 345-352, Utils/transformations.py:3-31).  Shared by loss_kernel (kr_train.hip)
// and the epilogue of the fused forward kernel (kr_mlp_fused.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace kr {

// quaternion_to_euler of Utils/transformations.py:3-31 and its Jacobian-transpose product
__device__ __forceinline__ void q2e(const float q[4], float e[3]) {
  const float inv = 1.f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
  e[0] = atan2f(2.f * (w * y + x * z), 1.f - 2.f * (y * y + z * z));
  e[1] = asinf(fminf(fmaxf(2.f * (w * z - x * y), -1.f), 1.f));
  e[2] = atan2f(2.f * (w * x + y * z), 1.f - 2.f * (x * x + z * z));
}
__device__ __forceinline__ void q2e_vjp(const float q[4], const float ge[3], float gq[4]) {
  const float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float inv = 1.f / nrm;
  const float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
  float gn[4] = {0.f, 0.f, 0.f, 0.f};  // gradient w.r.t. the normalised quaternion (w, x, y, z)
  {  // roll = atan2(a, b)
    const float a = 2.f * (w * y + x * z), b = 1.f - 2.f * (y * y + z * z);
    const float den = a * a + b * b;
    const float ga = ge[0] * b / den, gb = -ge[0] * a / den;
    gn[0] += ga * 2.f * y; gn[1] += ga * 2.f * z; gn[2] += ga * 2.f * w - gb * 4.f * y; gn[3] += ga * 2.f * x - gb * 4.f * z;
  }
  {  // pitch = asin(clamp(s))
    const float s = 2.f * (w * z - x * y);
    if (s >= -1.f && s <= 1.f) {
      const float gs = ge[1] / sqrtf(1.f - s * s);
      gn[0] += gs * 2.f * z; gn[1] -= gs * 2.f * y; gn[2] -= gs * 2.f * x; gn[3] += gs * 2.f * w;
    }
  }
  {  // yaw = atan2(c, d)
    const float c = 2.f * (w * x + y * z), d = 1.f - 2.f * (x * x + z * z);
    const float den = c * c + d * d;
    const float gc = ge[2] * d / den, gd = -ge[2] * c / den;
    gn[0] += gc * 2.f * x; gn[1] += gc * 2.f * w - gd * 4.f * x; gn[2] += gc * 2.f * z; gn[3] += gc * 2.f * y - gd * 4.f * z;
  }
  const float dot = gn[0] * w + gn[1] * x + gn[2] * y + gn[3] * z;
  gq[0] = (gn[0] - w * dot) * inv;
  gq[1] = (gn[1] - x * dot) * inv;
  gq[2] = (gn[2] - y * dot) * inv;
  gq[3] = (gn[3] - z * dot) * inv;
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
__device__ __forceinline__ float loss_row_angles(const float (&p)[25], const float (&tgv)[25], const float (&ep)[3],
                                                 const float (&et)[3], const LossWeights& w, float (&g)[25]) {
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
    float qp[4] = {p[3], p[4], p[5], p[6]};
    float ge[3], gq[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = ep[c] - et[c];
      part += w.w_e * d * d;
      ge[c] = 2.f * w.w_e * d;
    }
    q2e_vjp(qp, ge, gq);
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
  float ep[3], et[3];
  q2e(qp, ep);
  q2e(qt, et);
  return loss_row_angles(p, tgv, ep, et, w, g);
}

}  // namespace kr
