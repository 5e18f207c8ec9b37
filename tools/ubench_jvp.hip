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
  const char* nm[11] = {"entry+loads issued", "L1 products", "act 1", "sync+pack 1", "L2 products", "act 2", "sync+pack 2", "out products", "result stores", "-", "-"};
  double tot = 0;
  printf("  %s, per evaluation:", name);
  for (int k = 0; k < 11; ++k) { printf(" %s %.0f;", nm[k], (double)h[k] / iters); tot += (double)h[k] / iters; }
  printf(" (sum %.0f)\n", tot);
  unsigned long long z[16] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(kr::mj_stamp_acc), z, sizeof(z));
}
#endif
// A REAL random network 28 -> 64 -> 64 -> 25 packed into every fragment form (the formulas of kr_set_mlp, kr_api.hip), so
// that the fp32-base-chain evaluator (mlp_jvp_tile3f) can be compared with the fp64-base-chain one on the same inputs.
static uint16_t to_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }
template <typename T>
static MlpDev<T> real_net(bool f32_chain) {
  const int dims[4] = {28, 64, 64, 25};
  MlpDev<T> M{};
  M.n_layers = 3; M.mfma_ok = 1; M.jvp_ok = 1;
  unsigned st = 777u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((int)(st >> 8) % 2001 - 1000) * 1e-4f; };
  int prev_tiles = 0;
  const int groups[3] = {7, 16, 8};
  for (int k = 0; k < 3; ++k) {
    const int in = dims[k], out = dims[k + 1];
    std::vector<float> hw((size_t)in * out), hb(out);
    for (auto& v : hw) v = rnd() * (k == 0 ? 3.f : 1.5f);
    for (auto& v : hb) v = rnd();
    const bool last = k == 2;
    const int tiles = last ? 2 : 4;
    const int kg = k == 0 ? 2 : prev_tiles;
    std::vector<float> wq((size_t)tiles * kg * 64 * 4, 0.f), bq((size_t)tiles * 64, 0.f);
    for (int t = 0; t < tiles; ++t)
      for (int lane = 0; lane < 64; ++lane) {
        const int uo = 16 * t + (lane & 15);
        for (int g = 0; g < kg; ++g)
          for (int e = 0; e < 4; ++e) {
            const int ui = 4 * (4 * g + e) + (lane >> 4);
            if (uo < out && ui < in) wq[(((size_t)t * kg + g) * 64 + lane) * 4 + e] = hw[(size_t)uo * in + ui];
          }
        const int ub = 16 * t + 4 * ((lane >> 2) & 3) + (lane >> 4);
        if (ub < out) bq[(size_t)t * 64 + lane] = hb[ub];
      }
    const int jks = k == 0 ? 1 : prev_tiles / 2;
    std::vector<uint16_t> jf((size_t)tiles * jks * 64 * 8);
    for (int t = 0; t < tiles; ++t)
      for (int s2 = 0; s2 < jks; ++s2)
        for (int lane = 0; lane < 64; ++lane) {
          const int uo = 16 * t + (lane & 15), q = lane >> 4;
          for (int j = 0; j < 8; ++j) {
            const int ui = k == 0 ? 8 * q + j : 32 * s2 + (j < 4 ? 4 * q + j : 16 + 4 * q + (j - 4));
            jf[(((size_t)t * jks + s2) * 64 + lane) * 8 + j] = to_bf16((uo < out && ui < in) ? hw[(size_t)uo * in + ui] : 0.f);
          }
        }
    std::vector<float> w32((size_t)groups[k] * 64 * 4, 0.f), b32(64, 0.f);
    for (int lane = 0; lane < 64; ++lane) {
      const int row = k == 2 ? (lane & 31) : lane;
      const int k0 = k == 2 ? 32 * (lane >> 5) : 0;
      if (row < out)
        for (int g = 0; g < groups[k]; ++g)
          for (int e = 0; e < 4; ++e) {
            const int ui = k0 + 4 * g + e;
            if (ui < in) w32[((size_t)g * 64 + lane) * 4 + e] = hw[(size_t)row * in + ui];
          }
      if (lane < out && (k < 2 || lane < 32)) b32[lane] = hb[lane];
    }
    auto up = [](const void* src, size_t bytes) { void* d; (void)hipMalloc(&d, bytes); (void)hipMemcpy(d, src, bytes, hipMemcpyHostToDevice); return d; };
    M.wq[k] = (const float*)up(wq.data(), wq.size() * 4); M.bq[k] = (const float*)up(bq.data(), bq.size() * 4);
    M.jfrag[k] = up(jf.data(), jf.size() * 2);
    M.w32[k] = (const float*)up(w32.data(), w32.size() * 4); M.b32[k] = (const float*)up(b32.data(), b32.size() * 4);
    M.kgroups[k] = kg; M.otiles[k] = tiles; M.jksteps[k] = jks; M.acts[k] = last ? KR_ACT_NONE : KR_ACT_ELU; M.dims[k] = in;
    prev_tiles = tiles;
  }
  M.f32_ok = f32_chain ? 1 : 0;
  return M;
}
template <typename T>
__global__ __launch_bounds__(256) void k1(MlpDev<T> M, T* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  T* scratch = reinterpret_cast<T*>(smem + (size_t)wv * ((mj_scratch_bytes<T>() + 15) & ~size_t(15)));
  const bool idle = lane >= 58;
  int iv = 0, col = 0;
  if (lane < 7) { iv = 0; col = lane; } else if (!idle) { iv = 1 + (lane - 7) / 17; col = (lane - 7) % 17; }
  T x[MM_IN];
  for (int c = 0; c < MM_IN; ++c) x[c] = T(0.3) * ((iv * 7 + c * 3) % 11 - 5) + (col > 0 ? T(1e-3) * ((col * 5 + c) % 7 - 3) : T(0));
  T d[25];
  mlp_jvp_eval<T>(M, x, scratch, lane, iv, col, idle, -1, d);
  for (int c = 0; c < 25; ++c) out[(size_t)threadIdx.x * 25 + c] = d[c];
}
static void parity() {
  const size_t smem = 4 * ((mj_scratch_bytes<float>() + 15) & ~size_t(15));
  float* o[2]; std::vector<float> h[2];
  for (int v = 0; v < 2; ++v) {
    MlpDev<float> M = real_net<float>(v == 1);
    (void)hipMalloc(&o[v], 256 * 25 * 4);
    hipLaunchKernelGGL((k1<float>), dim3(1), dim3(256), smem, 0, M, o[v]);
    (void)hipDeviceSynchronize();
    h[v].resize(256 * 25); (void)hipMemcpy(h[v].data(), o[v], 256 * 25 * 4, hipMemcpyDeviceToHost);
  }
  double worst_base = 0, worst_col = 0, scale = 0;
  for (int l = 0; l < 58; ++l) {
    const bool base = l == 0 || (l >= 7 && (l - 7) % 17 == 0);
    for (int c = 0; c < 25; ++c) {
      const double a = h[0][l * 25 + c], b = h[1][l * 25 + c];
      scale = fmax(scale, fabs(a));
      (base ? worst_base : worst_col) = fmax(base ? worst_base : worst_col, fabs(a - b));
    }
  }
  printf("parity fp32 sweep: fp32 base chain vs fp64 base chain, real 28-64-64-25 network: max |diff| base lanes %.3g, perturbed lanes %.3g (values up to %.3g)\n",
         worst_base, worst_col, scale);
  for (int c = 0; c < 25; ++c) printf("  out %2d lane 0: %.6f | %.6f\n", c, h[0][c], h[1][c]);
  for (int l : {0, 1, 7, 8, 24, 41}) printf("  lane %2d: fp64-chain %.7f %.7f %.7f | fp32-chain %.7f %.7f %.7f\n", l, h[0][l*25], h[0][l*25+1], h[0][l*25+24], h[1][l*25], h[1][l*25+1], h[1][l*25+24]);
}
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
    if (L == 3) {  // fp32 base chain (timing only: fragments are unrelated pseudo-random numbers)
      const int groups[3] = {7, 16, 8};
      float *w32, *b32;
      (void)hipMalloc(&w32, 16 * groups[kk] * 64); fill(w32, 16 * groups[kk] * 64, false);
      (void)hipMalloc(&b32, 4 * 64); fill(b32, 4 * 64, false);
      M.w32[kk] = w32; M.b32[kk] = b32; M.f32_ok = (dims[1] <= 64 && dims[2] <= 64);
    }
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
  parity();
  for (int blocks : {1, 256}) {
    run<double>("f64 28-64-64-25", {28, 64, 64, 25}, blocks);
    run<float>("f32 28-64-64-25", {28, 64, 64, 25}, blocks);
  }
  run<float>("f32 28-64-64-25 RELU", {28, 64, 64, 25}, 256, KR_ACT_RELU);
  run<float>("f32 28-64-64-25 NONE", {28, 64, 64, 25}, 256, KR_ACT_NONE);
  run<double>("f64 28-64-64-25 RELU", {28, 64, 64, 25}, 256, KR_ACT_RELU);
  run<double>("f64 28-64-64-25 NONE", {28, 64, 64, 25}, 256, KR_ACT_NONE);
  run<double>("f64 28-64-64-25 TANH", {28, 64, 64, 25}, 256, KR_ACT_TANH);
  run<double>("f64 28-64-25", {28, 64, 25}, 256);
  run<double>("f64 28-512-25", {28, 512, 25}, 256);
  return 0;
}
