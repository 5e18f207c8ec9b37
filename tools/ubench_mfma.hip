// Dev microbenchmark (GPU box): issue interval of the 16x16x4 MFMAs used by mlp_mfma.hpp.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k64(double* out, unsigned long long* cyc, int iters, double a, double b) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double av = a + threadIdx.x, bv = b;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
__global__ void k32(float* out, unsigned long long* cyc, int iters, float a, float b) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  float av = a + threadIdx.x, bv = b;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* o; float* of; unsigned long long* c; (void)hipMalloc(&o, 8 * 64); (void)hipMalloc(&of, 4 * 64); (void)hipMalloc(&c, 64);
  unsigned long long h; const int it = 4000;
#define RUN(K, N, O, name) hipLaunchKernelGGL((K<N>), dim3(1), dim3(64), 0, 0, O, c, it, 1.0, 0.5); (void)hipDeviceSynchronize(); \
  (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf("%s independent accumulators=%d: %.1f cycles per MFMA\n", name, N, (double)h / (it * N));
  RUN(k64, 1, o, "f64 16x16x4") RUN(k64, 4, o, "f64 16x16x4") RUN(k64, 16, o, "f64 16x16x4")
  RUN(k32, 1, of, "f32 16x16x4") RUN(k32, 4, of, "f32 16x16x4") RUN(k32, 16, of, "f32 16x16x4")
  return 0;
}
