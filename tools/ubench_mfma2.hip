// Dev microbenchmark (GPU box): issue intervals of the MFMA forms the in-sweep MLP could use, and the operand / result
// layout of v_mfma_f64_4x4x4_4b_f64 found by one-hot probes (A one-hot at lane la, B one-hot at lane lb -> lanes of D).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short bf8 __attribute__((ext_vector_type(8)));
template <int NACC, int KIND>
__global__ void kt(double* out, unsigned long long* cyc, int iters, double a, double b) {
  double av = a + threadIdx.x, bv = b;
  unsigned long long t0, t1;
  double s = 0;
  if constexpr (KIND == 0) {  // f64 16x16x4
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  } else if constexpr (KIND == 1) {  // f64 4x4x4 (4 blocks)
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
    t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int i = 0; i < NACC; ++i) s += acc[i];
  } else {  // bf16 16x16x32
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    bf8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (short)(0x3f80 + threadIdx.x); y[j] = 0x3f00; }
    t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[i], 0, 0, 0);
    t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  }
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void probe(double* out) {
  const int la = blockIdx.x, lb = blockIdx.y, l = threadIdx.x;
  const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[((size_t)la * 64 + lb) * 64 + l] = d;
}
int main() {
  double* o; unsigned long long* c; (void)hipMalloc(&o, 8 * 64); (void)hipMalloc(&c, 64);
  unsigned long long h; const int it = 4000;
#define RUN(N, KIND, name) hipLaunchKernelGGL((kt<N, KIND>), dim3(1), dim3(64), 0, 0, o, c, it, 1.0, 0.5); (void)hipDeviceSynchronize(); \
  (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf("%s independent accumulators=%d: %.1f cycles per MFMA\n", name, N, (double)h / (it * N));
  RUN(1, 0, "f64 16x16x4") RUN(4, 0, "f64 16x16x4") RUN(1, 1, "f64 4x4x4_4b") RUN(2, 1, "f64 4x4x4_4b") RUN(4, 1, "f64 4x4x4_4b") RUN(16, 1, "f64 4x4x4_4b")
  RUN(1, 2, "bf16 16x16x32") RUN(4, 2, "bf16 16x16x32") RUN(16, 2, "bf16 16x16x32")
  double* po; (void)hipMalloc(&po, 8ull * 64 * 64 * 64);
  hipLaunchKernelGGL(probe, dim3(64, 64), dim3(64), 0, 0, po);
  (void)hipDeviceSynchronize();
  std::vector<double> r(64 * 64 * 64); (void)hipMemcpy(r.data(), po, 8ull * 64 * 64 * 64, hipMemcpyDeviceToHost);
  // for every A lane: the B lanes it pairs with and where the product lands
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      for (int l = 0; l < 64; ++l)
        if (r[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf(" (B%d->D%d)", lb, l);
    printf("\n");
  }
  return 0;
}
