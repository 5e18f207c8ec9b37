// Dev microbenchmark (GPU box): cycles of one wave-level base + JVP evaluation of the residual MLP (mlp_jvp.hpp), with
// the chip loaded like the persistent kernel loads it (4 waves per workgroup, 256 workgroups), and with parts switched off.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#include "../knode-cosserat_amd/csrc/mlp_jvp.hpp"
namespace kr { void set_error(const std::string&) {} int hip_fail(hipError_t, const char*) { return -2; } int ensure_ws(kr_handle*, size_t) { return 0; } }
using namespace kr;
constexpr int WAVE = 64;
template <typename T>
__global__ __launch_bounds__(256) void k(MlpDev<T> M, T* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  T* scratch = reinterpret_cast<T*>(smem + (size_t)wv * ((mj_scratch_bytes<T>() + 15) & ~size_t(15)));
  const bool idle = lane >= 58;
  int iv = 0, col = 0;
  if (lane < 7) { iv = 0; col = lane; } else if (!idle) { iv = 1 + (lane - 7) / 17; col = (lane - 7) % 17; }
  T x[MM_IN];
  for (int c = 0; c < MM_IN; ++c) x[c] = T(0.01) * (iv + c) + (col > 0 ? T(1e-7) * col : T(0));
  T o[25];
  for (int c = 0; c < 25; ++c) o[c] = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < iters; ++it) {
    T d[25];
    mlp_jvp_eval<T>(M, x, scratch, lane, iv, col, idle, -1, d);
    for (int c = 0; c < 25; ++c) o[c] += d[c];
    x[0] += d[0] * T(1e-9);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  T s = 0;
  for (int c = 0; c < 25; ++c) s += o[c];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
#ifdef MJ_STAMPS
static void stamps(const char* name, int iters) {
  unsigned long long h[16];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(kr::mj_stamp_acc), sizeof(h));
  const char* nm[11] = {"weight loads", "base L1", "act 1", "base L2", "act 2", "base out", "jvp L1", "pack 1", "jvp L2", "pack 2", "jvp out"};
  double tot = 0;
  printf("  %s, per evaluation:", name);
  for (int k = 0; k < 11; ++k) { printf(" %s %.0f;", nm[k], (double)h[k] / iters); tot += (double)h[k] / iters; }
  printf(" (sum %.0f)\n", tot);
  unsigned long long z[16] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(kr::mj_stamp_acc), z, sizeof(z));
}
#endif
template <typename T>
void run(const char* name, std::vector<int> dims, int blocks, int act = KR_ACT_ELU) {
  const int L = (int)dims.size() - 1;
  MlpDev<T> M{};
  M.n_layers = L; M.mfma_ok = 1; M.jvp_ok = 1;
  int prev = 0;
  for (int kk = 0; kk < L; ++kk) {
    const bool last = kk == L - 1;
    const int tiles = last ? 2 : ((dims[kk + 1] + 63) / 64) * 4;
    const int kg = kk == 0 ? 2 : prev;
    const int jks = kk == 0 ? 1 : prev / 2;
    float *w, *b; void* j;
    // pseudo-random fragments (not a real network: the point is a data-dependent checksum that a re-ordering of the
    // evaluator must reproduce bit for bit)
    auto fill = [&](void* dst, size_t bytes, bool bf16) {
      std::vector<unsigned> hbuf(bytes / 4);
      unsigned st = 12345u + (unsigned)kk * 7919u + (unsigned)bytes;
      for (auto& v : hbuf) {
        st = st * 1664525u + 1013904223u;
        const float f = ((int)(st >> 8) % 2001 - 1000) * 1e-4f;  // |f| <= 0.1
        unsigned bits; memcpy(&bits, &f, 4);
        if (bf16) { st = st * 1664525u + 1013904223u; const float g = ((int)(st >> 8) % 2001 - 1000) * 1e-4f; unsigned b2; memcpy(&b2, &g, 4); v = (bits >> 16) | (b2 & 0xFFFF0000u); }
        else v = bits;
      }
      (void)hipMemcpy(dst, hbuf.data(), bytes, hipMemcpyHostToDevice);
    };
    (void)hipMalloc(&w, 16 * tiles * kg * 64); fill(w, 16 * tiles * kg * 64, false);
    (void)hipMalloc(&b, 4 * tiles * 64); fill(b, 4 * tiles * 64, false);
    (void)hipMalloc(&j, 16 * tiles * jks * 64); fill(j, 16 * tiles * jks * 64, true);
    M.wq[kk] = w; M.bq[kk] = b; M.kgroups[kk] = kg; M.otiles[kk] = tiles; M.acts[kk] = last ? KR_ACT_NONE : act;
    M.jfrag[kk] = j; M.jksteps[kk] = jks;
    M.dims[kk] = dims[kk]; prev = tiles;
  }
  T* out; unsigned long long* cyc; (void)hipMalloc(&out, sizeof(T) * 256 * blocks); (void)hipMalloc(&cyc, 8 * blocks);
  const int iters = 100;
  const size_t smem = 4 * ((mj_scratch_bytes<T>() + 15) & ~size_t(15));
  hipLaunchKernelGGL((k<T>), dim3(blocks), dim3(256), smem, 0, M, out, cyc, iters);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks); (void)hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += (double)v;
  std::vector<T> ho(256); (void)hipMemcpy(ho.data(), out, sizeof(T) * 256, hipMemcpyDeviceToHost);
  double cs = 0; for (int i = 0; i < 256; ++i) cs += (double)ho[i] * (1 + i % 7);
  printf("%s blocks=%d: %.0f cycles per evaluation, checksum %.17g\n", name, blocks, s / blocks / iters, cs);
#ifdef MJ_STAMPS
  stamps(name, iters);
#endif
}
int main() {
  for (int blocks : {1, 256}) {
    run<double>("f64 28-64-64-25", {28, 64, 64, 25}, blocks);
    run<float>("f32 28-64-64-25", {28, 64, 64, 25}, blocks);
  }
  run<double>("f64 28-64-64-25 RELU", {28, 64, 64, 25}, 256, KR_ACT_RELU);
  run<double>("f64 28-64-64-25 NONE", {28, 64, 64, 25}, 256, KR_ACT_NONE);
  run<double>("f64 28-64-64-25 TANH", {28, 64, 64, 25}, 256, KR_ACT_TANH);
  run<double>("f64 28-64-25", {28, 64, 25}, 256);
  run<double>("f64 28-512-25", {28, 512, 25}, 256);
  return 0;
}
