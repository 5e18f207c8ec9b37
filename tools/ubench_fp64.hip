// Dev microbenchmark (GPU box): cycles per v_fma_f64 / v_fma_f32 for one wave alone on a SIMD and for two
// waves sharing a SIMD, at several degrees of instruction-level parallelism.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_fp64.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int ILP>
__global__ void k(T* out, unsigned long long* cyc, int iters, T a, T b) {
  T x[ILP];
  for (int i = 0; i < ILP; ++i) x[i] = a + T(threadIdx.x + i);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < ILP; ++i) x[i] = fma(x[i], a, b);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  T s = 0;
  for (int i = 0; i < ILP; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <typename T, int ILP>
void run(const char* name, int threads) {
  T* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(T) * threads); hipMalloc(&cyc, 8 * 64);
  const int iters = 2000;
  hipLaunchKernelGGL((k<T, ILP>), dim3(1), dim3(threads), 0, 0, out, cyc, iters, T(0.999), T(0.001));
  hipDeviceSynchronize();
  unsigned long long h[16]; hipMemcpy(h, cyc, sizeof(unsigned long long) * (threads / 64), hipMemcpyDeviceToHost);
  unsigned long long mx = 0; for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
  printf("%s ILP=%d waves/block=%d (%.1f per SIMD): %.2f cycles per wave-instruction\n", name, ILP, threads / 64, threads / 256.0,
         (double)mx / (iters * 8.0 * ILP));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<double, 1>("f64", 64); run<double, 2>("f64", 64); run<double, 4>("f64", 64); run<double, 8>("f64", 64);
  run<double, 8>("f64", 256); run<double, 8>("f64", 512); run<double, 2>("f64", 512); run<double, 4>("f64", 512);
  run<float, 1>("f32", 64); run<float, 4>("f32", 64); run<float, 8>("f32", 64); run<float, 8>("f32", 512);
  return 0;
}
