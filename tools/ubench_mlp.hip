// Dev microbenchmark (GPU box): cycles of one wave-level MLP evaluation on the matrix cores (mlp_mfma.hpp).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../knode-cosserat_amd/csrc/mlp_mfma.hpp"
namespace kr { void set_error(const std::string&) {} int hip_fail(hipError_t, const char*) { return -2; } int ensure_ws(kr_handle*, size_t) { return 0; } }
using namespace kr;
template <typename T>
__global__ __launch_bounds__(64) void k(MlpDev<T> M, T* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) T tile[64 * MM_TILE_LD];
  const int lane = threadIdx.x;
  T x[MM_IN];
  for (int c = 0; c < MM_IN; ++c) x[c] = T(0.01) * (lane + c);
  T o[25];
  for (int c = 0; c < 25; ++c) o[c] = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < iters; ++it) {
    T d[25];
    mlp_mfma_eval<T>(M, x, tile, lane, d);
    for (int c = 0; c < 25; ++c) o[c] += d[c];
    x[0] += d[0] * T(1e-9);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  T s = 0;
  for (int c = 0; c < 25; ++c) s += o[c];
  out[lane] = s;
  if (lane == 0) cyc[0] = t1 - t0;
}
template <typename T>
void run(const char* name, std::vector<int> dims) {
  const int L = (int)dims.size() - 1;
  MlpDev<T> M{};
  M.n_layers = L; M.mfma_ok = 1;
  int prev = 0;
  for (int kk = 0; kk < L; ++kk) {
    const bool last = kk == L - 1;
    const int tiles = last ? 2 : ((dims[kk + 1] + 63) / 64) * 4;
    const int ks = kk == 0 ? 7 : prev * 4;
    T *w, *b;
    (void)hipMalloc(&w, sizeof(T) * tiles * ks * 64); (void)hipMemset(w, 0, sizeof(T) * tiles * ks * 64);
    (void)hipMalloc(&b, sizeof(T) * tiles * 4 * 64); (void)hipMemset(b, 0, sizeof(T) * tiles * 4 * 64);
    M.wfrag[kk] = w; M.bfrag[kk] = b; M.ksteps[kk] = ks; M.otiles[kk] = tiles; M.acts[kk] = last ? KR_ACT_NONE : KR_ACT_ELU;
    M.dims[kk] = dims[kk]; prev = tiles;
  }
  T* out; unsigned long long* cyc; (void)hipMalloc(&out, sizeof(T) * 64); (void)hipMalloc(&cyc, 8);
  const int iters = 200;
  hipLaunchKernelGGL((k<T>), dim3(1), dim3(64), 0, 0, M, out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  int mf = 0; { int pt = 0; for (int kk = 0; kk < L; ++kk) { int tiles = kk == L - 1 ? 2 : ((dims[kk + 1] + 63) / 64) * 4; int ks = kk == 0 ? 7 : pt * 4; mf += tiles * ks * 4; pt = tiles; } }
  printf("%s: %.0f cycles per evaluation (%d MFMAs -> %.0f cycles of matrix-pipe time at %d each)\n", name, (double)h / iters, mf,
         mf * (sizeof(T) == 8 ? 64.0 : 32.0), sizeof(T) == 8 ? 64 : 32);
}
int main() {
  run<double>("f64 28-64-25", {28, 64, 25}); run<double>("f64 28-64-64-25", {28, 64, 64, 25}); run<double>("f64 28-512-25", {28, 512, 25});
  run<float>("f32 28-64-25", {28, 64, 25}); run<float>("f32 28-64-64-25", {28, 64, 64, 25}); run<float>("f32 28-512-25", {28, 512, 25});
  return 0;
}
