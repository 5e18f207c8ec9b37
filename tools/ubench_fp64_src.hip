// Dev microbenchmark (GPU box): does the number of VGPR source operands change the issue rate of v_fma_f64 /
// v_mul_f64 / v_add_f64 for one wave alone on a SIMD?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_fp64_src.hip -o /tmp/ub2 && /tmp/ub2
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int ILP>
__global__ void k(double* out, unsigned long long* cyc, int iters, double a, double b, const double* src) {
  double x[ILP], y[ILP], z[ILP];
  for (int i = 0; i < ILP; ++i) { x[i] = a + threadIdx.x + i; y[i] = src[threadIdx.x + i]; z[i] = src[threadIdx.x + 64 + i]; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (MODE == 0) x[i] = fma(x[i], a, b);          // 1 VGPR source
        if (MODE == 1) x[i] = fma(x[i], y[i], b);       // 2 VGPR sources
        if (MODE == 2) x[i] = fma(x[i], y[i], z[i]);    // 3 VGPR sources
        if (MODE == 3) x[i] = x[i] * y[i];              // mul, 2 VGPR
        if (MODE == 4) x[i] = x[i] + y[i];              // add, 2 VGPR
        if (MODE == 5) x[i] = fma(y[i], z[i], x[i]);    // fmac form (accumulate into x)
      }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  double s = 0;
  for (int i = 0; i < ILP; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE, int ILP>
void run(const char* name, int threads) {
  double* out; unsigned long long* cyc; double* src;
  hipMalloc(&out, 8 * threads); hipMalloc(&cyc, 8 * 64); hipMalloc(&src, 8 * 1024);
  double hs[1024]; for (int i = 0; i < 1024; ++i) hs[i] = 0.999 + 1e-6 * i;
  hipMemcpy(src, hs, sizeof(hs), hipMemcpyHostToDevice);
  const int iters = 2000;
  hipLaunchKernelGGL((k<MODE, ILP>), dim3(1), dim3(threads), 0, 0, out, cyc, iters, 0.999, 0.001, src);
  hipDeviceSynchronize();
  unsigned long long h[16]; hipMemcpy(h, cyc, 8 * (threads / 64), hipMemcpyDeviceToHost);
  unsigned long long mx = 0; for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
  printf("%-28s ILP=%d waves/SIMD=%.1f: %.2f cycles per wave-instruction\n", name, ILP, threads / 256.0, (double)mx / (iters * 8.0 * ILP));
  hipFree(out); hipFree(cyc); hipFree(src);
}
int main() {
  run<0, 8>("fma 1 vgpr src", 64); run<1, 8>("fma 2 vgpr src", 64); run<2, 8>("fma 3 vgpr src", 64);
  run<3, 8>("mul 2 vgpr src", 64); run<4, 8>("add 2 vgpr src", 64); run<5, 8>("fmac 3 vgpr", 64);
  run<2, 4>("fma 3 vgpr src", 64); run<2, 2>("fma 3 vgpr src", 64); run<2, 1>("fma 3 vgpr src", 64);
  run<2, 8>("fma 3 vgpr src", 512); run<5, 8>("fmac 3 vgpr", 512);
  return 0;
}
