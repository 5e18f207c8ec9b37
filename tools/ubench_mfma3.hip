// Dev probe (GPU box): operand / result layout and issue cost of v_mfma_f32_4x4x1_16B_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void layout(float* out) {
  const int l = threadIdx.x;
  // A[b][i] = 100 b + 10 i + 1 ;  B[b][j] = 1000 b' encoded so that D tells who met whom
  f4 d = {0, 0, 0, 0};
  const float a = (float)(l + 1);          // lane id + 1 as A
  const float b = 1.0f;
  d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];           // D = A-lane that feeds (lane, reg r)
  f4 e = {0, 0, 0, 0};
  e = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(l + 1), e, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[256 + l * 4 + r] = e[r];     // E = B-lane that feeds (lane, reg r)
}
__global__ void cost(float* out, unsigned long long* cyc) {
  f4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.0001f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 256; ++it) {
    d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d1, 0, 0, 0);
    d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d2, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d3, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f4 dd = {0, 0, 0, 0};
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 1024; ++it) dd = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, dd, 0, 0, 0);
  unsigned long long t3 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = d0[0] + d1[1] + d2[2] + d3[3] + dd[0];
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; }
}
int main() {
  float* o; unsigned long long* c;
  hipMalloc(&o, 4096 * 4); hipMalloc(&c, 16);
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, o);
  float h[512]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  printf("D[lane][reg] = A lane + 1 that reached it (B = 1):\n");
  for (int l = 0; l < 64; l += 1) { if (l < 12 || l > 59) printf("  lane %2d: %3.0f %3.0f %3.0f %3.0f\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
  printf("E[lane][reg] = B lane + 1 that reached it (A = 1):\n");
  for (int l = 0; l < 64; l += 1) { if (l < 12 || l > 59) printf("  lane %2d: %3.0f %3.0f %3.0f %3.0f\n", l, h[256+l*4], h[256+l*4+1], h[256+l*4+2], h[256+l*4+3]); }
  hipLaunchKernelGGL(cost, dim3(1), dim3(64), 0, 0, o, c);
  unsigned long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
  printf("4 independent accumulators: %.1f cycles per MFMA; dependent chain: %.1f\n", hc[0] / 1024.0, hc[1] / 1024.0);
  return 0;
}
