// kr_mlp_fused.hip - fused fp32 forward / backward of the residual MLP for the KNODE training step
// (reference: CosseratRodTorch.forward, cosserat_ode_torch.py:131-134, and loss.backward() of
// physics_train.py:290,394 restricted to the MLP parameters).
//
// One wavefront owns blocks of 16 FT (= 32) rows.  Everything is expressed in the transposed form
// [units x samples] so that an MFMA accumulator tile (v_mfma_f32_16x16x4_f32: lane l holds column
// sample l&15, rows 4*(l>>4)+r) is directly the B operand of the next product:
//
//   forward   Z1 = W1 X^T + b1, A1 = act(Z1);  Z2 = W2 A1 + b2, A2 = act(Z2);  OUT = W3 A2 + b3
//   backward  dZ2 = (W3^T dOUT) * act'(Z2);  dZ1 = (W2^T dZ2) * act'(Z1)          (chain in registers)
//             dW3 += dOUT A2^T, dW2 += dZ2 A1^T, dW1 += dZ1 X^T                   (contractions over the samples of
//             a block: both operands go through transposed LDS tiles T[unit][position], see "backward" below)
//
// Kernels (two wavefronts per SIMD, so that the matrix-pipe phases of one overlap the vector / LDS phases of the other):
//   mlp_fwd3_kernel        three layers (H1, H2 <= 64, <= 25 outputs): 8 wavefronts per workgroup, fragments in the LDS, the
//                          next block's x rows by LDS-direct loads; dumps A1 and A2 of every row block as raw register images
//   mlp_fwd2_kernel        two layers with H1 <= 512: all fragments in the LDS of the CU
//   mlp_fwd_fused_kernel   every other served shape: fragments from memory
//   mlp_bwd3_kernel        three layers, the whole backward pass of a row block (kr_train_epoch); mlp_bwd3a / 3b_kernel: the
//                          same in two launches with dZ2 through memory (kr_mlp_backward)
//   mlp_bwd2_kernel        two layers (any hidden width, streamed in chunks of 64 units): recomputes the hidden chunk
//   train_tail_kernel      slab and loss sums, Adam, clamp, plateau schedule, fragment update (kr_train_epoch);
//   reduce_slabs_kernel    the slab sum alone, with float atomics (kr_mlp_backward)
// Weight gradients accumulate in registers over all row blocks of a wave, are added up over the wavefronts of the workgroup
// through the LDS and leave it ONCE, as plain stores into the workgroup's slab [workgroup][all parameters].
// Splitting the three-layer backward in two and dumping activations instead of recomputing them is what lets every
// kernel fit 256 registers (2 waves per SIMD) without spilling; HBM is otherwise idle in these kernels.
// Weights are re-packed into MFMA fragment order on the device at every call (pack_all_kernel, one launch; they are
// torch parameters that change every optimizer step; ~30 KB).
// Shapes served: in(<=32) -> H1 -> out(<=32) with any H1 and in -> H1 -> H2 -> out with H1, H2 <= 64; one activation
// for all hidden layers.  Other networks use the generic GEMM path of kr_train.hip.
#include <cstdlib>
#include <type_traits>

#include "kr_internal.hpp"
#include "kr_loss_device.hpp"

namespace kr {

typedef float f4 __attribute__((ext_vector_type(4)));
#ifndef KR_FUSED_FT
#define KR_FUSED_FT 2
#endif
constexpr int FT = KR_FUSED_FT;  // sample tiles per row block (2 or 4): 16 FT rows.  2 halves the LDS tiles and the
                                 // register chunks of a wave, which buys the second wave per SIMD (the kernels alternate
                                 // matrix-pipe phases with vector / LDS phases: one wave leaves the pipe idle two thirds of the time)
static_assert(FT == 2 || FT == 4, "row blocks of 32 or 64 samples");
constexpr int FR = 16 * FT;  // rows per block
constexpr int F_MAXW = FT == 2 ? 2048 : 1024;  // backward wavefronts resident at once: LDS 20 / 40 KB each
typedef float fvT __attribute__((ext_vector_type(FT)));  // the FT samples a lane holds of one unit
constexpr int FPD = 3;       // prefetch distance of weight fragments, in k-steps
constexpr int F_LDX = 32;    // row length of the X / dOUT tiles
constexpr int F_LDH = 64;    // row length of the hidden tiles
constexpr int F_LDO = 36;    // pitch of the forward kernel's output tile

struct FChunk {
  f4 a[4][FT];  // [unit tile][sample tile]
};

// diagnostic build (make dbgf; tools/fused_stamps.py): cycle counts of the phases of a row block, summed over all
// wavefronts into the buffer given to kr_debug_buffer
#ifdef KR_FUSED_STAMPS
#define FSTAMP_DECL unsigned long long fst[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ft0
#define FSTAMP_T0 do { __builtin_amdgcn_sched_barrier(0); ft0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#define FSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); fst[k] += t_ - ft0; ft0 = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define FSTAMP_OUT(kern, wave) do { if (A.stamps && lane == 0) { for (int k_ = 0; k_ < 12; ++k_) A.stamps[((size_t)(kern) * 4096 + (wave)) * 12 + k_] += fst[k_]; } } while (0)
#else
#define FSTAMP_DECL
#define FSTAMP_T0 do { } while (0)
#define FSTAMP(k) do { } while (0)
#define FSTAMP_OUT(kern, wave) do { } while (0)
#endif

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// Every MFMA kernel of this file runs ONE wavefront per workgroup (dim3(64)), so all LDS hand-offs are between lanes of
// one wavefront, whose LDS instructions execute in order: a wavefront-scope fence (a compiler barrier) is enough.  The
// workgroup-scope fence this used to be also drains every global load in flight (s_waitcnt vmcnt(0)) - the prefetched
// weight fragments and row blocks - at each of the ~20 synchronisation points of a row block.
__device__ __forceinline__ void fsync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---- weight packing (device side, every call) ------------------------------------------------------
// forward fragments: wf[(t*ks + s)*64 + lane] = W[16t + (lane&15)][unit_in(s, lane>>4)], bias fragments
// bf[(t*4 + r)*64 + lane] = b[16t + 4*(lane>>4) + r]; transposed fragments for dA_prev = W^T dZ:
// wt[(ti*ks + s)*64 + lane] = W[unit_out(s, q)][16*ti + (lane&15)].  One launch packs every layer (the packing
// kernels are launch-latency bound: five of them cost 23 us of a 300 us epoch).
struct PackArgs {
  const float* W[3];
  const float* b[3];
  float* wf[3];
  float* bf[3];
  float* wt[3];
  int in[3], out[3], tiles[3], ks[3], in_tiles[3], kst[3], natural[3];
  int L;
};
__global__ void pack_all_kernel(const PackArgs P) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  for (int k = 0; k < P.L; ++k) {
    const float* __restrict__ W = P.W[k];
    const int in = P.in[k], out = P.out[k], ks = P.ks[k];
    const int n = P.tiles[k] * ks * 64;
    for (int i = tid; i < n; i += nth) {
      // element i = ((t * ks / 4 + s / 4) * 64 + lane) * 4 + s % 4: four consecutive k-steps of a lane are ONE 16-byte load
      const int e = i & 3, lane = (i >> 2) & 63, g = (i >> 8) % (ks / 4), t = (i >> 8) / (ks / 4);
      const int s = 4 * g + e;
      const int uo = 16 * t + (lane & 15), q = lane >> 4;
      const int ui = k == 0 ? 4 * s + q : 16 * (s / 4) + 4 * q + (s % 4);
      P.wf[k][i] = (uo < out && ui < in) ? W[(size_t)uo * in + ui] : 0.f;
    }
    const int nb = P.tiles[k] * 4 * 64;
    for (int i = tid; i < nb; i += nth) {
      const int lane = i & 63, r = (i >> 6) & 3, t = i >> 8;
      const int u = 16 * t + 4 * (lane >> 4) + r;
      P.bf[k][i] = u < out ? P.b[k][u] : 0.f;
    }
    if (k > 0) {
      const int kst = P.kst[k];
      const int nt = P.in_tiles[k] * kst * 64;
      for (int i = tid; i < nt; i += nth) {
        const int e = i & 3, lane = (i >> 2) & 63, g = (i >> 8) % (kst / 4), ti = (i >> 8) / (kst / 4);
        const int s = 4 * g + e;
        const int ui = 16 * ti + (lane & 15), q = lane >> 4;
        const int uo = P.natural[k] ? 4 * s + q : 16 * (s / 4) + 4 * q + (s % 4);
        P.wt[k][i] = (uo < out && ui < in) ? W[(size_t)uo * in + ui] : 0.f;
      }
    }
  }
}

// ---- register-level building blocks ------------------------------------------------------------------
// dst[o][s] += Wfrag[tile0+o][ks0+k] * bsrc(s, k), k < KS, weight fragments prefetched FPD k-steps ahead
template <int NO, int KS, typename BFn>
__device__ __forceinline__ void facc(f4 (&dst)[NO][FT], const float* __restrict__ w, int ksteps, int tile0, int ks0,
                                     int lane, BFn bsrc) {
  // fragments [tile][k-group of 4][lane][4]: one 16-byte load per four k-steps (the single-dword loads of round 3 were
  // 248 load instructions plus their address arithmetic per 32-row block); the next group is requested before the
  // products of the current one are issued
  static_assert(KS % 4 == 0, "k-steps come in groups of four");
  constexpr int KG = KS / 4;
  const f4* __restrict__ w4 = reinterpret_cast<const f4*>(w);
  const int kg = ksteps >> 2, g0 = ks0 >> 2;
  f4 a[KG][NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) a[0][o] = w4[((size_t)(tile0 + o) * kg + g0) * 64 + lane];
#pragma unroll
  for (int g = 0; g < KG; ++g) {
    if (g + 1 < KG) {
#pragma unroll
      for (int o = 0; o < NO; ++o) a[g + 1][o] = w4[((size_t)(tile0 + o) * kg + g0 + g + 1) * 64 + lane];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int s = 0; s < FT; ++s) dst[o][s] = mfma4(a[g][o][e], bsrc(s, 4 * g + e), dst[o][s]);
    // keep the software pipeline as written: left alone, the scheduler hoists every fragment load of the chain to its
    // head and the kernel no longer fits two waves per SIMD
    __builtin_amdgcn_sched_barrier(0);
  }
}
__device__ __forceinline__ void chunk_set_bias(FChunk& h, const float* __restrict__ bf, int tile0, int lane) {
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    f4 b;
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = bf[((size_t)(tile0 + o) * 4 + r) * 64 + lane];
#pragma unroll
    for (int s = 0; s < FT; ++s) h.a[o][s] = b;
  }
}
__device__ __forceinline__ void chunk_zero(FChunk& h) {
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int s = 0; s < FT; ++s) h.a[o][s] = f4{0.f, 0.f, 0.f, 0.f};
}
// z -> act(z) in place, g = act'(z)
template <int ACT>
__device__ __forceinline__ void chunk_act_grad(FChunk& z, FChunk& g) {
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int s = 0; s < FT; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = z.a[o][s][r];
        g.a[o][s][r] = activate_grad<float>(ACT, v);
        z.a[o][s][r] = activate<float>(ACT, v);
      }
}
// Activation of the training kernels.  ELU: max(x, 0) + (exp(min(x, 0)) - 1) on v_exp_f32, six instructions and no select
// (the general activate<float> also evaluates the short series that keeps expm1's RELATIVE accuracy for |x| < 0.25 and
// picks: fourteen instructions for each of the 64 values a lane holds per row block - a third of the forward kernel's
// vector work).  The absolute error of exp(x) - 1 is half an ulp of 1 (6e-8), which is what every other fp32 operation of
// the step commits on values of this size; the backward passes take act' = a + 1 = exp(x) from it, where nothing cancels.
template <int ACT>
__device__ __forceinline__ float act_train(float x) {
  if constexpr (ACT == KR_ACT_ELU) {
    const float em1 = __builtin_amdgcn_exp2f(fminf(x, 0.f) * 1.44269504088896341f) - 1.f;
    return fmaxf(x, 0.f) + em1;
  } else {
    return activate<float>(ACT, x);
  }
}
template <int ACT>
__device__ __forceinline__ void chunk_act_only(FChunk& z) {
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int s = 0; s < FT; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) z.a[o][s][r] = act_train<ACT>(z.a[o][s][r]);
}
// act'(z) expressed through a = act(z) (so only the activations need to be kept, in their LDS tile)
template <int ACT>
__device__ __forceinline__ float grad_from_act(float a) {
  if constexpr (ACT == KR_ACT_ELU) return a > 0.f ? 1.f : a + 1.f;       // exp(z) = a + 1 for z <= 0
  else if constexpr (ACT == KR_ACT_TANH) return 1.f - a * a;
  else if constexpr (ACT == KR_ACT_SOFTPLUS) return 1.f - __expf(-a);    // sigmoid(z) = 1 - exp(-softplus(z))
  else if constexpr (ACT == KR_ACT_RELU) return a > 0.f ? 1.f : 0.f;
  else return 1.f;
}
// d *= act'(z), with act(z) read back from its LDS tile [sample][unit] (row length ld, units tile0*16..)
template <int ACT>
__device__ __forceinline__ void chunk_mul_grad(FChunk& d, const float* tile, int ld, int lane) {
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int s = 0; s < FT; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        d.a[o][s][r] *= grad_from_act<ACT>(tile[(16 * s + (lane & 15)) * ld + 16 * o + 4 * (lane >> 4) + r]);
}
__device__ __forceinline__ void chunk_mul(FChunk& d, const FChunk& g) {
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int s = 0; s < FT; ++s) d.a[o][s] = d.a[o][s] * g.a[o][s];
}
// D layout -> LDS tile[sample][unit] (row length ld)
__device__ __forceinline__ void chunk_to_tile(float* tile, int ld, const FChunk& h, int lane) {
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int s = 0; s < FT; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[(16 * s + (lane & 15)) * ld + 16 * o + 4 * (lane >> 4) + r] = h.a[o][s][r];
}
// acc[o][i] += sum over the 64 samples of A[sample][16o + .] * B[sample][16i + .]  (both LDS tiles)
template <int NO, int NI>
__device__ __forceinline__ void wgrad(f4 (&acc)[NO][NI], const float* ta, int lda, const float* tb, int ldb, int lane) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int row = 4 * k + (lane >> 4);
    float a[NO], b[NI];
#pragma unroll
    for (int o = 0; o < NO; ++o) a[o] = ta[row * lda + 16 * o + (lane & 15)];
#pragma unroll
    for (int i = 0; i < NI; ++i) b[i] = tb[row * ldb + 16 * i + (lane & 15)];
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[o][i] = mfma4(a[o], b[i], acc[o][i]);
  }
}
// Weight gradients leave a wave ONCE, as plain stores into the wave's own slab of partial sums (nn.Linear layout);
// reduce_slabs_kernel adds the slabs up afterwards.  (Round 1 flushed with float atomics: ~1000 wavefronts adding
// 30 KB each into the same few rows serialise at the memory side - most of the backward kernel's time.)
template <int NO, int NI>
__device__ __forceinline__ void wgrad_flush(const f4 (&acc)[NO][NI], float* __restrict__ dW, int out, int in, int out0,
                                            int in0, int lane) {
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int uo = out0 + 16 * o + 4 * (lane >> 4) + r, ui = in0 + 16 * i + (lane & 15);
        if (uo < out && ui < in) dW[(size_t)uo * in + ui] = acc[o][i][r];
      }
}
// per-lane partial column sums of a D-layout chunk (sum over its sample tiles)
__device__ __forceinline__ void bias_partial(f4 (&p)[4], const FChunk& d) {
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int s = 0; s < FT; ++s) p[o] = p[o] + d.a[o][s];
}
__device__ __forceinline__ void bias_flush(const f4 (&p)[4], float* __restrict__ db, int out, int out0, int lane) {
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = p[o][r];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
      const int u = out0 + 16 * o + 4 * (lane >> 4) + r;
      if ((lane & 15) == 0 && u < out) db[u] = v;
    }
}

// B operands (natural k order) of the four sample tiles from an LDS tile [sample][32]
__device__ __forceinline__ void load_bops(float (&b)[FT][8], const float* tile, int lane) {
#pragma unroll
  for (int s = 0; s < FT; ++s)
#pragma unroll
    for (int k = 0; k < 8; ++k) b[s][k] = tile[(16 * s + (lane & 15)) * F_LDX + 4 * k + (lane >> 4)];
}

// rows [row0, row0 + FR) of a row-major [Q][32] array -> LDS tile (zero beyond Q); 64 / FR lanes share a row
template <int PITCH = F_LDX>
__device__ __forceinline__ void stage_rows(const float* __restrict__ g, int64_t row0, int64_t Q, float* tile, int lane) {
  constexpr int PARTS = 64 / FR, NV = F_LDX / 4 / PARTS;
  const int rl = lane % FR, part = lane / FR;
  const int64_t row = row0 + rl;
  const f4* src = reinterpret_cast<const f4*>(g + row * F_LDX) + part * NV;
  f4* dst = reinterpret_cast<f4*>(tile + rl * PITCH) + part * NV;
#pragma unroll
  for (int c = 0; c < NV; ++c) dst[c] = row < Q ? src[c] : f4{0.f, 0.f, 0.f, 0.f};
}

// raw register image of a chunk, [block][unit tile][sample tile][lane] x 16 bytes: coalesced both ways, and exactly
// the layout the consumer needs as B operand / for its T tile
__device__ __forceinline__ void chunk_dump(float* __restrict__ base, int64_t rb, const FChunk& h, int lane) {
#ifdef KR_ABL_NODUMP  // (timing ablation of a diagnostic build: the backward pass then reads stale images)
  return;
#endif
  f4* dst = reinterpret_cast<f4*>(base) + (size_t)rb * (4 * FT * 64) + lane;
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int sidx = 0; sidx < FT; ++sidx) dst[(o * FT + sidx) * 64] = h.a[o][sidx];
}
__device__ __forceinline__ void chunk_undump(FChunk& h, const float* __restrict__ base, int64_t rb, int lane) {
  const f4* src = reinterpret_cast<const f4*>(base) + (size_t)rb * (4 * FT * 64) + lane;
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int sidx = 0; sidx < FT; ++sidx) h.a[o][sidx] = src[(o * FT + sidx) * 64];
}

struct FusedArgs {
  int64_t Q;
  int L;                 // 2 or 3 layers
  int in, h1, h2;        // true widths (h2 unused for L == 2)
  int nout;              // width of the last layer (<= 32; 25 for the rod's MLP)
  int c1;                // hidden chunks of layer 1 (h1 padded to 64*c1)
  const float* x;        // [Q][32]
  const float* dout;     // [Q][32] (backward)
  float* out;            // [Q][32] (forward)
  const float* wf[3];    // forward fragments per layer
  const float* bfr[3];
  const float* wt[3];    // transposed fragments (index = layer whose input gradient they produce; [0] unused)
  int ks[3];             // forward k-steps per tile
  int kst[3];            // k-steps of the transposed products
  float* dW[3];
  float* db[3];
  float* slab;           // [streams][P] partial sums of every parameter gradient, P = all parameters of the network
  int P, nslab, nparams;  // P: slab pitch (parameters rounded up to 64)
  int poff[6];           // offsets of dW[0], db[0], dW[1], db[1], dW[2], db[2] inside a slab
  float* a1d;            // three-layer networks: hidden activations A1, A2 of every row block as register images, written
  float* a2d;            // by the forward kernel and read by the backward passes (HBM is idle here, the matrix pipe is not)
  float* dz2;            // three-layer backward: dZ2 of every row block between the two passes, [blocks][16][64] x 16 bytes
  // forward with the loss fused into its epilogue (kr_mlp_forward_loss): parameter-free part of the prediction and
  // target values per row [Q][25], gradient out [Q][32]; lbase == nullptr: plain forward
  const float* lbase;
  const float* ltarget;
  float* ldout;
  float* lpart;          // one loss partial per workgroup (summed by loss_partials_kernel: one atomic per workgroup on
                         // a single address serialises at the memory side - 4096 of them cost 16 us; a device-scope
                         // fence per workgroup for an in-kernel arrival count writes the L2 back each time - 180 us)
  float lds;             // arc-length step: pred = base + [ds out[:19], out[19:]]
  LossWeights lw;
  unsigned long long* stamps;  // diagnostic build only
};
static unsigned long long* g_fused_stamps = nullptr;
void fused_set_debug(void* p) { g_fused_stamps = static_cast<unsigned long long*>(p); }

// ---- forward -------------------------------------------------------------------------------------------
// Three-layer networks (H1, H2 <= 64) keep their forward fragments in the LDS, one copy per workgroup of FW3 wavefronts
// (mlp_fwd3_kernel): 8 + 16 + 8 KB of products' A operands and the three bias vectors in compact form.  Fetched from
// memory (L2) inside the products they queue behind the row block's own loads in the wavefront's in-order return path:
// stamps (tools/fused_stamps.py) showed a wavefront spending 35 k cycles on a row block whose 256 products take 8 k.
constexpr int FW3 = 8;
constexpr int FW3_W0 = 0, FW3_W1 = 4 * 8 * 64, FW3_W2 = FW3_W1 + 4 * 16 * 64, FW3_B = FW3_W2 + 2 * 16 * 64;  // floats
constexpr int FW3_FRAG = FW3_B + 64 + 64 + 32;
constexpr int FW3_WAVE = 2 * FR * F_LDX + 2 * FR * 25;  // per wavefront: two x tiles (the output tile lies over the current one), base + target rows

// LDS-direct load of 16 bytes per lane: lane l's bytes land at LDS byte address lds_byte + 16 l (wave-uniform base in M0)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_byte)
               : "memory");
}

// the row blocks first, first + stride, ... of one wavefront; WLDS: fragments at `wl` in the LDS (three layers)
template <int ACT, bool WLDS>
__device__ __forceinline__ void fwd_rows(const FusedArgs& A, float* tx, float* tbt, const float* wl, int lane, int64_t first,
                                         int64_t stride, int part) {
  const int64_t nblk = (A.Q + FR - 1) / FR;
  const bool with_loss = A.lbase != nullptr;
  float loss_part = 0.f;
  FSTAMP_DECL;
#ifdef KR_FUSED_STAMPS
  const unsigned long long fkernel0 = __builtin_amdgcn_s_memtime();
#endif
  FSTAMP_T0;
  // WLDS: the x rows of the NEXT block travel while this one is worked on, by LDS-direct loads (no registers, and nothing
  // the compiler would have to wait for at the loop head: with ordinary loads carried around the loop it drains the
  // whole queue there - s_waitcnt vmcnt(0) - and that queue ends with the gradient rows just stored).  The x tile is
  // row-major with pitch 32 and an XOR swizzle of its 16-byte pieces, applied on the SOURCE side (an LDS-direct load
  // writes lane-linear); two tiles per wavefront; the output tile of a block (pitch OP) lies over its x tile.
  constexpr int OP = WLDS ? 25 : F_LDO;
  float* xcur = tx;
  auto x_request = [&](int64_t rbn, float* buf) {
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)buf);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int P = 64 * j + lane, row = P >> 3, kp = P & 7;
      int64_t grow = rbn * FR + row;
      grow = grow < A.Q ? grow : A.Q - 1;  // rows past the end repeat the last one (their results are never used)
      glds16(A.x + grow * F_LDX + 4 * (kp ^ (row & 7)), base + 1024 * j);
    }
  };
  if constexpr (WLDS) {
    if (first < nblk) x_request(first, tx);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  bool first_block = true;
  for (int64_t rb = first; rb < nblk; rb += stride) {
    if constexpr (WLDS) {
      // x of this block was requested a block ago; younger in the queue: this wavefront's 16 dump stores and 4 row
      // stores of the last block (and its base / target loads, long consumed) - they may stay in flight
      if (!first_block) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      first_block = false;
      FSTAMP(10);
      if (rb + stride < nblk) x_request(rb + stride, xcur == tx ? tx + FR * F_LDX : tx);
      FSTAMP(11);
    } else {
      stage_rows<F_LDO>(A.x, rb * FR, A.Q, tx, lane);  // (pitch 36: the operand reads below hit 2 banks, not 16)
    }
    // fused loss: the block's base and target rows (FR x 25 contiguous floats of each array; a row block starts at a
    // multiple of 3200 bytes) are requested now, as 16-byte pieces into registers, and go to LDS only when the
    // epilogue needs them - their latency hides behind the layers
    constexpr int LNV = (FR * 25 / 4 + 63) / 64;
    f4 lvb[LNV], lvt[LNV];
    if (with_loss) {
      const int64_t e0 = rb * FR * 25, eN = A.Q * 25;
      const f4* sb = reinterpret_cast<const f4*>(A.lbase + e0);
      const f4* st = reinterpret_cast<const f4*>(A.ltarget + e0);
#pragma unroll
      for (int q = 0; q < LNV; ++q) {
        const int i = lane + 64 * q;
        lvb[q] = f4{0.f, 0.f, 0.f, 0.f};
        lvt[q] = f4{1.f, 0.f, 0.f, 0.f};
        if (i < FR * 25 / 4) {
          if (e0 + 4 * i + 3 < eN) {
            lvb[q] = sb[i];
            lvt[q] = st[i];
          } else {
            for (int c = 0; c < 4; ++c)
              if (e0 + 4 * i + c < eN) { lvb[q][c] = A.lbase[e0 + 4 * i + c]; lvt[q][c] = A.ltarget[e0 + 4 * i + c]; }
          }
        }
      }
    }
    fsync();
    FSTAMP(0);  // x rows landed in the tile (the wait for the block's loads is here)
    float bin[FT][8];
#pragma unroll
    for (int s = 0; s < FT; ++s)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        bin[s][k] = WLDS ? xcur[(16 * s + (lane & 15)) * F_LDX + 4 * (k ^ (lane & 7)) + (lane >> 4)]
                         : tx[(16 * s + (lane & 15)) * F_LDO + 4 * k + (lane >> 4)];
    f4 oacc[2][FT];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      f4 b;
      if constexpr (WLDS) {
        b = *reinterpret_cast<const f4*>(wl + FW3_B + 128 + 16 * o + 4 * (lane >> 4));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r] = A.bfr[A.L - 1][((size_t)o * 4 + r) * 64 + lane];
      }
#pragma unroll
      for (int s = 0; s < FT; ++s) oacc[o][s] = b;
    }
    if constexpr (WLDS) {
      FChunk h1, h2;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const f4 b = *reinterpret_cast<const f4*>(wl + FW3_B + 16 * o + 4 * (lane >> 4));
#pragma unroll
        for (int s = 0; s < FT; ++s) h1.a[o][s] = b;
      }
      FSTAMP(1);
      facc<4, 8>(h1.a, wl + FW3_W0, 8, 0, 0, lane, [&](int s, int k) { return bin[s][k]; });
      FSTAMP(2);
      chunk_act_only<ACT>(h1);
      chunk_dump(A.a1d, rb, h1, lane);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const f4 b = *reinterpret_cast<const f4*>(wl + FW3_B + 64 + 16 * o + 4 * (lane >> 4));
#pragma unroll
        for (int s = 0; s < FT; ++s) h2.a[o][s] = b;
      }
      FSTAMP(3);
      facc<4, 16>(h2.a, wl + FW3_W1, 16, 0, 0, lane, [&](int s, int k) { return h1.a[k >> 2][s][k & 3]; });
      FSTAMP(4);
      chunk_act_only<ACT>(h2);
      chunk_dump(A.a2d, rb, h2, lane);
      FSTAMP(5);
      facc<2, 16>(oacc, wl + FW3_W2, 16, 0, 0, lane, [&](int s, int k) { return h2.a[k >> 2][s][k & 3]; });
      FSTAMP(6);
    } else if (A.L == 2) {
      for (int c = 0; c < A.c1; ++c) {
        FChunk h;
        chunk_set_bias(h, A.bfr[0], 4 * c, lane);
        facc<4, 8>(h.a, A.wf[0], 8, 4 * c, 0, lane, [&](int s, int k) { return bin[s][k]; });
        chunk_act_only<ACT>(h);
        facc<2, 16>(oacc, A.wf[1], A.ks[1], 0, 16 * c, lane, [&](int s, int k) { return h.a[k >> 2][s][k & 3]; });
      }
    } else {
      FChunk h1, h2;
      chunk_set_bias(h1, A.bfr[0], 0, lane);
      FSTAMP(1);
      facc<4, 8>(h1.a, A.wf[0], 8, 0, 0, lane, [&](int s, int k) { return bin[s][k]; });
      FSTAMP(2);
      chunk_act_only<ACT>(h1);
      chunk_dump(A.a1d, rb, h1, lane);
      chunk_set_bias(h2, A.bfr[1], 0, lane);
      FSTAMP(3);
      facc<4, 16>(h2.a, A.wf[1], A.ks[1], 0, 0, lane, [&](int s, int k) { return h1.a[k >> 2][s][k & 3]; });
      FSTAMP(4);
      chunk_act_only<ACT>(h2);
      chunk_dump(A.a2d, rb, h2, lane);
      FSTAMP(5);
      facc<2, 16>(oacc, A.wf[2], A.ks[2], 0, 0, lane, [&](int s, int k) { return h2.a[k >> 2][s][k & 3]; });
      FSTAMP(6);
    }
    fsync();
    // outputs -> row-major tile (pitch F_LDO: one lane reads a whole row below; 36 keeps the rows 16-byte aligned and
    // spreads them over 8 banks instead of 1)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int s = 0; s < FT; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int u = 16 * o + 4 * (lane >> 4) + r;
          if constexpr (WLDS) {
            if (u < 25) xcur[(16 * s + (lane & 15)) * OP + u] = oacc[o][s][r];
          } else {
            tx[(16 * s + (lane & 15)) * F_LDO + u] = u < A.nout ? oacc[o][s][r] : 0.f;
          }
        }
    if (with_loss) {
#pragma unroll
      for (int q = 0; q < LNV; ++q) {
        const int i = lane + 64 * q;
        if (i < FR * 25 / 4) {
          reinterpret_cast<f4*>(tbt)[i] = lvb[q];
          reinterpret_cast<f4*>(tbt + FR * 25)[i] = lvt[q];
        }
      }
    }
    fsync();
    FSTAMP(7);  // outputs, base and target rows in their tiles
    if (with_loss) {
      // one row per lane (lanes 0 .. FR-1): prediction, four-term loss, gradient with respect to the outputs (written
      // over them).  The upper half of the wave is idle here, so it takes one of the two quaternion_to_euler calls
      // of a row: lane FR + r converts the target quaternion of row r while lane r converts the predicted one.
      // (Giving the upper lane the 21 plain components as well was measured: the two halves then run different code, one
      // after the other - 10.7 k cycles per block instead of 7.4 k.)
      static_assert(FR == 32, "lane <-> row map of the loss epilogue");
      const int rl = lane & (FR - 1);
      const bool valid = rb * FR + rl < A.Q;
      float e[3] = {0.f, 0.f, 0.f}, nq[5];
      {
        float q[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          q[c] = lane < FR ? tbt[rl * 25 + 3 + c] + A.lds * xcur[rl * OP + 3 + c] : tbt[FR * 25 + rl * 25 + 3 + c];
        if (!valid) { q[0] = 1.f; q[1] = q[2] = q[3] = 0.f; }
        q2e_n(q, e, nq);  // (the lower lanes keep the normalised quaternion for the Jacobian-transpose product below)
      }
      float eo[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) eo[c] = __shfl_xor(e[c], FR, 64);
      if (lane < FR && valid) {
        float p[25], tgv[25], g[25];
#pragma unroll
        for (int r = 0; r < 25; ++r) {
          p[r] = tbt[lane * 25 + r] + (r < 19 ? A.lds : 1.f) * xcur[lane * OP + r];
          tgv[r] = tbt[FR * 25 + lane * 25 + r];
        }
        loss_part += loss_row_angles(p, tgv, e, eo, nq, A.lw, g);
#pragma unroll
        for (int r = 0; r < 25; ++r) xcur[lane * OP + r] = (r < 19 ? A.lds : 1.f) * g[r];
      }
      fsync();
    }
    FSTAMP(8);  // loss epilogue
    {
      constexpr int PARTS = 64 / FR, NV = F_LDX / 4 / PARTS;  // lanes per row, 16-byte pieces per lane
      const int rl = lane % FR, part = lane / FR;
      const int64_t row = rb * FR + rl;
      if (row < A.Q) {
        f4* dst = reinterpret_cast<f4*>((with_loss ? A.ldout : A.out) + row * F_LDX) + part * NV;
        if constexpr (WLDS) {  // pitch 25: scalar reads, columns 25 .. 31 are zero
#pragma unroll
          for (int c = 0; c < NV; ++c) {
            f4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int u = 4 * (part * NV + c) + e;
              v[e] = u < 25 ? xcur[rl * OP + (u < 25 ? u : 0)] : 0.f;
            }
            dst[c] = v;
          }
        } else {
          const f4* src = reinterpret_cast<const f4*>(tx + rl * F_LDO) + part * NV;
#pragma unroll
          for (int c = 0; c < NV; ++c) dst[c] = src[c];
        }
      }
    }
    fsync();
    FSTAMP(9);  // gradient rows stored
    if constexpr (WLDS) xcur = xcur == tx ? tx + FR * F_LDX : tx;
  }

  FSTAMP_OUT(0, part);
  if (with_loss) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) loss_part += __shfl_xor(loss_part, m, 64);
    if (lane == 0) A.lpart[part] = loss_part;
  }
}

extern __shared__ __attribute__((aligned(16))) float wg_lds[];
// Epilogue of a two-layer forward row block: outputs (accumulator layout) -> tile [row][25], then - with the loss fused in -
// prediction, four-term loss and gradient with respect to the outputs per row (lane r owns row r, lane FR + r converts the
// target quaternion of row r; the base rows and then the target rows take the place of the outputs in the tile), and the
// row store.  `to`: FR x 25 floats of LDS owned by the calling wavefront for the duration of the call.
constexpr int LNV2 = (FR * 25 / 4 + 63) / 64;
__device__ __forceinline__ void fwd2_outputs_to_tile(const FusedArgs& A, float* to, const f4 (&oacc)[2][FT], int lane) {
  const int c = lane & 15, g = lane >> 4;
  // outputs -> tile [row][25]
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int s = 0; s < FT; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int u = 16 * o + 4 * g + r;
        if (u < 25 && u < A.nout) to[(16 * s + c) * 25 + u] = oacc[o][s][r];
        else if (u < 25) to[(16 * s + c) * 25 + u] = 0.f;
      }
  fsync();
}
// ... the rest of it, from the outputs in the tile
__device__ __forceinline__ void fwd2_epilogue_rest(const FusedArgs& A, float* to, const f4 (&lvb)[LNV2], const f4 (&lvt)[LNV2],
                                                   int64_t rb, int lane, float& loss_part) {
  constexpr int LNV = LNV2;
  const bool with_loss = A.lbase != nullptr;
  if (with_loss) {
    // as in fwd_rows: lane r owns row r, lane FR + r converts the target quaternion of row r; the base rows and then
    // the target rows take the place of the outputs in the tile
    static_assert(FR == 32, "lane <-> row map of the loss epilogue");
    const int rl = lane & (FR - 1);
    const bool valid = rb * FR + rl < A.Q, lower = lane < FR;
    float p[25], tgv[25], gr[25];
#pragma unroll
    for (int r = 0; r < 25; ++r) p[r] = to[rl * 25 + r];
    fsync();
#pragma unroll
    for (int q = 0; q < LNV; ++q)
      if (lane + 64 * q < FR * 25 / 4) reinterpret_cast<f4*>(to)[lane + 64 * q] = lvb[q];
    fsync();
#pragma unroll
    for (int r = 0; r < 25; ++r) p[r] = to[rl * 25 + r] + (r < 19 ? A.lds : 1.f) * p[r];
    fsync();
#pragma unroll
    for (int q = 0; q < LNV; ++q)
      if (lane + 64 * q < FR * 25 / 4) reinterpret_cast<f4*>(to)[lane + 64 * q] = lvt[q];
    fsync();
#pragma unroll
    for (int r = 0; r < 25; ++r) tgv[r] = to[rl * 25 + r];
    float e[3] = {0.f, 0.f, 0.f}, nq[5];
    {
      float q[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) q[cc] = lower ? p[3 + cc] : tgv[3 + cc];
      if (!valid) { q[0] = 1.f; q[1] = q[2] = q[3] = 0.f; }
      q2e_n(q, e, nq);
    }
    float eo[3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) eo[cc] = __shfl_xor(e[cc], FR, 64);
    fsync();
    if (lower && valid) {
      loss_part += loss_row_angles(p, tgv, e, eo, nq, A.lw, gr);
#pragma unroll
      for (int r = 0; r < 25; ++r) to[lane * 25 + r] = (r < 19 ? A.lds : 1.f) * gr[r];
    }
    fsync();
  }
  {
    constexpr int PARTS = 64 / FR, NV = F_LDX / 4 / PARTS;
    const int rl = lane % FR, part = lane / FR;
    const int64_t row = rb * FR + rl;
    if (row < A.Q) {
      f4* dst = reinterpret_cast<f4*>((with_loss ? A.ldout : A.out) + row * F_LDX) + part * NV;
#pragma unroll
      for (int cc = 0; cc < NV; ++cc) {
        f4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int u = 4 * (part * NV + cc) + e;
          v[e] = u < 25 ? to[rl * 25 + (u < 25 ? u : 0)] : 0.f;
        }
        dst[cc] = v;
      }
    }
  }
  fsync();
}
__device__ __forceinline__ void fwd2_epilogue(const FusedArgs& A, float* to, const f4 (&lvb)[LNV2], const f4 (&lvt)[LNV2],
                                              const f4 (&oacc)[2][FT], int64_t rb, int lane, float& loss_part) {
  fwd2_outputs_to_tile(A, to, oacc, lane);
  fwd2_epilogue_rest(A, to, lvb, lvt, rb, lane, loss_part);
}

// Two-layer networks in -> H1 -> out with H1 <= 512: ALL forward fragments in the LDS, one copy per CU (workgroups of FW2
// wavefronts, one per CU): stamps of the kernel below on 28 -> 512 -> 25 showed, per hidden chunk, 1.1 k cycles waiting for
// the bias fragments, 3.3 k for the 2.0 k of layer-1 products and 4.4 k for the 2.0 k of layer-2 products - fragments
// from memory, each behind the previous in the wavefront's in-order return path.  With 128 KB of fragments there is no LDS
// left for x / base / target tiles: the B operands of layer 1 are read from memory in operand layout (16 dwords per
// lane), and base and target rows pass through the output tile one after the other (3.2 KB per wavefront).
constexpr int FW2 = 8;
constexpr int FW2_OUT = FR * 25;
template <int ACT>
__global__ __launch_bounds__(64 * FW2) void mlp_fwd2_kernel(const FusedArgs A) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c1 = A.c1;
  float* const wl0 = wg_lds;
  float* const wl1 = wl0 + c1 * 2048;
  float* const bl1 = wl1 + c1 * 2048;
  float* const bl2 = bl1 + 64 * c1;
  float* const to = bl2 + 32 + wv * FW2_OUT;
  {
    const int t = threadIdx.x;
    const f4* s0 = reinterpret_cast<const f4*>(A.wf[0]);
    const f4* s1 = reinterpret_cast<const f4*>(A.wf[1]);
    f4* d0 = reinterpret_cast<f4*>(wl0);
    f4* d1 = reinterpret_cast<f4*>(wl1);
    for (int i = t; i < c1 * 512; i += 64 * FW2) { d0[i] = s0[i]; d1[i] = s1[i]; }
    for (int u = t; u < 64 * c1; u += 64 * FW2) bl1[u] = A.bfr[0][((u >> 4) * 4 + (u & 3)) * 64 + 16 * ((u & 15) >> 2)];
    if (t < 32) bl2[t] = A.bfr[1][((t >> 4) * 4 + (t & 3)) * 64 + 16 * ((t & 15) >> 2)];
  }
  __syncthreads();
  const int64_t nblk = (A.Q + FR - 1) / FR;
  const bool with_loss = A.lbase != nullptr;
  const int gid = blockIdx.x * FW2 + wv;
  const int c = lane & 15, g = lane >> 4;
  float loss_part = 0.f;
  for (int64_t rb = gid; rb < nblk; rb += (int64_t)gridDim.x * FW2) {
    float bin[FT][8];
#pragma unroll
    for (int s = 0; s < FT; ++s) {
      const int64_t row = rb * FR + 16 * s + c;
      const float* xr = A.x + (row < A.Q ? row : A.Q - 1) * F_LDX + g;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = xr[4 * k];
        bin[s][k] = row < A.Q ? v : 0.f;
      }
    }
    constexpr int LNV = (FR * 25 / 4 + 63) / 64;
    f4 lvb[LNV], lvt[LNV];
    if (with_loss) {
      const int64_t e0 = rb * FR * 25, eN = A.Q * 25;
      const f4* sb = reinterpret_cast<const f4*>(A.lbase + e0);
      const f4* st = reinterpret_cast<const f4*>(A.ltarget + e0);
#pragma unroll
      for (int q = 0; q < LNV; ++q) {
        const int i = lane + 64 * q;
        lvb[q] = f4{0.f, 0.f, 0.f, 0.f};
        lvt[q] = f4{1.f, 0.f, 0.f, 0.f};
        if (i < FR * 25 / 4) {
          if (e0 + 4 * i + 3 < eN) {
            lvb[q] = sb[i];
            lvt[q] = st[i];
          } else {
            for (int cc = 0; cc < 4; ++cc)
              if (e0 + 4 * i + cc < eN) { lvb[q][cc] = A.lbase[e0 + 4 * i + cc]; lvt[q][cc] = A.ltarget[e0 + 4 * i + cc]; }
          }
        }
      }
    }
    f4 oacc[2][FT];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const f4 b = *reinterpret_cast<const f4*>(bl2 + 16 * o + 4 * g);
#pragma unroll
      for (int s = 0; s < FT; ++s) oacc[o][s] = b;
    }
    for (int ch = 0; ch < c1; ++ch) {
      FChunk h;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const f4 b = *reinterpret_cast<const f4*>(bl1 + 64 * ch + 16 * o + 4 * g);
#pragma unroll
        for (int s = 0; s < FT; ++s) h.a[o][s] = b;
      }
      facc<4, 8>(h.a, wl0, 8, 4 * ch, 0, lane, [&](int s, int k) { return bin[s][k]; });
      chunk_act_only<ACT>(h);
      facc<2, 16>(oacc, wl1, 16 * c1, 0, 16 * ch, lane, [&](int s, int k) { return h.a[k >> 2][s][k & 3]; });
    }
    fwd2_epilogue(A, to, lvb, lvt, oacc, rb, lane, loss_part);
  }
  if (with_loss) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) loss_part += __shfl_xor(loss_part, m, 64);
    if (lane == 0) A.lpart[gid] = loss_part;
  }
}

template <int ACT>
__global__ __launch_bounds__(64, 2) void mlp_fwd_fused_kernel(const FusedArgs A) {
  __shared__ __attribute__((aligned(16))) float tx[FR * F_LDO];
  __shared__ __attribute__((aligned(16))) float tbt[2 * FR * 25];  // fused loss: base rows, target rows of the block
  fwd_rows<ACT, false>(A, tx, tbt, nullptr, threadIdx.x, blockIdx.x, gridDim.x, blockIdx.x);
}

template <int ACT>
__global__ __launch_bounds__(64 * FW3, 1) void mlp_fwd3_kernel(const FusedArgs A) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* const wl = wg_lds;
  {
    const int t = threadIdx.x;
    const f4* s0 = reinterpret_cast<const f4*>(A.wf[0]);
    const f4* s1 = reinterpret_cast<const f4*>(A.wf[1]);
    const f4* s2 = reinterpret_cast<const f4*>(A.wf[2]);
    f4* d = reinterpret_cast<f4*>(wl);
    for (int i = t; i < FW3_W1 / 4; i += 64 * FW3) d[FW3_W0 / 4 + i] = s0[i];
    for (int i = t; i < (FW3_W2 - FW3_W1) / 4; i += 64 * FW3) d[FW3_W1 / 4 + i] = s1[i];
    for (int i = t; i < (FW3_B - FW3_W2) / 4; i += 64 * FW3) d[FW3_W2 / 4 + i] = s2[i];
    if (t < 160) {  // compact biases out of the bias fragments: b[u] sits at fragment slot ((u / 16) 4 + u % 4) 64 + 16 ((u % 16) / 4)
      const int k = t < 64 ? 0 : t < 128 ? 1 : 2, u = t - 64 * k;
      wl[FW3_B + t] = A.bfr[k][((u >> 4) * 4 + (u & 3)) * 64 + 16 * ((u & 15) >> 2)];
    }
  }
  __syncthreads();
  float* const tx = wg_lds + FW3_FRAG + wv * FW3_WAVE;
  float* const tbt = tx + 2 * FR * F_LDX;
  const int gid = blockIdx.x * FW3 + wv;
  fwd_rows<ACT, true>(A, tx, tbt, wl, lane, gid, (int64_t)gridDim.x * FW3, gid);
}

// sum of the per-workgroup loss partials of a fused forward + loss launch, added to *loss
__global__ __launch_bounds__(256) void loss_partials_kernel(const float* __restrict__ part, int n, float* __restrict__ loss) {
  __shared__ float red[256];
  float v = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) v += part[i];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(loss, red[0]);
}

// ---- backward ------------------------------------------------------------------------------------------
// LDS tiles of the backward kernel are TRANSPOSED: T[unit][position], 64 positions per row, where position
// p = 4 c + s stands for the sample that sits in column c of sample tile s of an accumulator (row 16 s + c of the
// block).  Two things follow:
//   * an accumulator chunk goes to its tile with one 16-byte store per (unit tile, register): the four sample tiles of
//     a lane are four consecutive positions;
//   * the weight-gradient contractions  dW += dZ^T A  take k-step kk as the positions {16 q + kk}: lane (unit i, q)
//     reads the sixteen k-values of a tile as 4 x ds_read_b128 instead of 16 x ds_read_b32 - which sample is which k
//     does not matter as long as both operands agree.
// Rows are XOR-swizzled in 16-byte groups (group ^ sigma(unit & 15), sigma = swap of the two 2-bit halves) so that the
// stores, the operand reads and the activation read-backs are all bank-conflict free.  40 KB per wave (X^T and dOUT^T
// share one tile: X^T is staged a second time for dW1), one wave per SIMD.
constexpr int TP = 16 * FT;   // positions per row
constexpr int TNG = TP / 4;   // 16-byte groups per row
__device__ __forceinline__ int tsig(int i) { return ((i & 3) << 2) | ((i >> 2) & 3); }
__device__ __forceinline__ int tgrp(int u, int grp) { return u * TP + ((grp ^ (tsig(u & 15) & (TNG - 1))) << 2); }
// address of position p of row u
__device__ __forceinline__ int tpos(int u, int p) { return tgrp(u, p >> 2) + (p & 3); }

// accumulator chunk (4 unit tiles) -> T[unit][position]
__device__ __forceinline__ void chunk_to_T(float* tile, const FChunk& h, int lane) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      fvT v;
#pragma unroll
      for (int sidx = 0; sidx < FT; ++sidx) v[sidx] = h.a[o][sidx][r];
      *reinterpret_cast<fvT*>(tile + tpos(16 * o + 4 * g + r, FT * c)) = v;
    }
}
// d *= act'(z) with act(z) read back from its tile
template <int ACT>
__device__ __forceinline__ void chunk_mul_grad_T(FChunk& d, const float* tile, int lane) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const fvT a = *reinterpret_cast<const fvT*>(tile + tpos(16 * o + 4 * g + r, FT * c));
#pragma unroll
      for (int sidx = 0; sidx < FT; ++sidx) d.a[o][sidx][r] *= grad_from_act<ACT>(a[sidx]);
    }
}
// rows [row0, row0 + 64) of a row-major [Q][32] array -> T[32][position] (zero beyond Q); lane = row of the block
constexpr int RNV = F_LDX / 4 / (64 / FR);  // 16-byte pieces of a row per lane (64 / FR lanes share a row)
__device__ __forceinline__ void rows_request(f4 (&v)[RNV], const float* __restrict__ gsrc, int64_t row0, int64_t Q, int lane) {
  const int rl = lane % FR, part = lane / FR;
  const int64_t row = row0 + rl;
  const f4* src = reinterpret_cast<const f4*>(gsrc + row * F_LDX) + part * RNV;
#pragma unroll
  for (int k = 0; k < RNV; ++k) v[k] = row < Q ? src[k] : f4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void rows_to_T(const f4 (&v)[RNV], float* tile, int lane) {
  const int rl = lane % FR, part = lane / FR;
  const int p = FT * (rl & 15) + (rl >> 4);  // position of this row: column rl & 15 of sample tile rl >> 4
#pragma unroll
  for (int k = 0; k < RNV; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[tpos(4 * (part * RNV + k) + e, p)] = v[k][e];
}
__device__ __forceinline__ void stage_rows_T(const float* __restrict__ gsrc, int64_t row0, int64_t Q, float* tile, int lane) {
  f4 v[RNV];
  rows_request(v, gsrc, row0, Q, lane);
  rows_to_T(v, tile, lane);
}
// B operands (natural k order) of the four sample tiles from T[32][position]: input 4 k + (lane >> 4)
__device__ __forceinline__ void load_bops_T(float (&b)[FT][8], const float* tile, int lane) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const fvT v = *reinterpret_cast<const fvT*>(tile + tpos(4 * k + g, FT * c));
#pragma unroll
    for (int sidx = 0; sidx < FT; ++sidx) b[sidx][k] = v[sidx];
  }
}
// sum over all positions of row `row` of a T tile (bias gradients: d b = sum over the samples of dZ)
__device__ __forceinline__ float row_sum_T(const float* tile, int row) {
  f4 sacc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int grp = 0; grp < TNG; ++grp) sacc = sacc + *reinterpret_cast<const f4*>(tile + row * TP + 4 * grp);
  return (sacc[0] + sacc[1]) + (sacc[2] + sacc[3]);
}
// acc[o][i] += sum over the 64 positions of A[16 o + .][p] * B[16 i + .][p]   (both T tiles)
template <int NO, int NI>
__device__ __forceinline__ void wgrad_T(f4 (&acc)[NO][NI], const float* ta, const float* tb, int lane) {
  const int u = lane & 15, q = lane >> 4;
  // one 16-byte group (4 k-steps) of every operand tile at a time: NO + NI live operand registers x 4, and the
  // scheduler is kept from hoisting all sixteen k-steps' reads (the kernel sits at the 256 architectural registers)
#pragma unroll
  for (int j = 0; j < FT; ++j) {  // lane (unit u, q) owns positions q TP / 4 .. : FT groups of 4 k-steps
    f4 av[NO], bv[NI];
#pragma unroll
    for (int o = 0; o < NO; ++o) av[o] = *reinterpret_cast<const f4*>(ta + tgrp(16 * o + u, FT * q + j));
#pragma unroll
    for (int i = 0; i < NI; ++i) bv[i] = *reinterpret_cast<const f4*>(tb + tgrp(16 * i + u, FT * q + j));
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[o][i] = mfma4(av[o][e], bv[i][e], acc[o][i]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Three-layer networks (H1, H2 <= 64) take two passes over the rows so that neither kernel needs more than the 256
// architectural registers of a wave (one kernel for everything spilled 0.5 KB per lane):
//   pass A   X -> A1 -> A2;  dW3 += dOUT^T A2;  dZ2 = (W3^T dOUT) * act'(Z2);  dW2 += dZ2^T A1;  db2, db3;
//            dZ2 leaves as a raw register dump [block][tile][sample tile][lane] x 16 bytes (coalesced both ways)
//   pass B   X -> A1 (layer 1 again: 10 % more matrix work);  dZ1 = (W2^T dZ2) * act'(Z1);  dW1 += dZ1^T X;  db1
// The two passes run WPB wavefronts per workgroup.  The wavefronts of a workgroup work on their own row blocks with their
// own tiles and meet only once, after the last block: their weight-gradient accumulators are added up through the LDS
// (two rounds of a tree) and leave the workgroup as ONE slab - a quarter of the slab bytes written here and read by the
// reduction (2048 slabs x 30 KB = 63 MB each way per epoch before).
constexpr int WPB = 4;
constexpr int B3A_LDS = (64 + 64) * TP;       // floats per wavefront: [dOUT^T, later A1], A2 -> dZ2
constexpr int B3A_WT = 4 * 8 * 64;            // W3^T fragments (floats), once per workgroup behind the tiles
constexpr int B3B_LDS = (32 + 64) * TP;       // X^T, A1 -> dZ1
constexpr int B3B_WT = 4 * 16 * 64;           // W2^T fragments (floats), once per workgroup behind the wavefronts' tiles

// acc (NA x NI and NB x NI accumulator tiles, two per-lane scalars) <- sum over the WPB wavefronts of the workgroup; the
// result is valid in wavefront 0.  R: the workgroup's LDS (free once every wavefront is past its last row block).
template <int W = WPB, int NA, int NB, int NI, int NI2>
__device__ __forceinline__ void wg_tree_sum(f4 (&a)[NA][NI], f4 (&b)[NB][NI2], float& s0, float& s1, float* lds, int w,
                                            int lane) {
  static_assert(W >= 2 && W <= 16, "up to four rounds");
  constexpr int NF = NA * NI + NB * NI2 + 1;
  f4* R = reinterpret_cast<f4*>(lds);
  auto put = [&](f4* dst) {
#pragma unroll
    for (int o = 0; o < NA; ++o)
#pragma unroll
      for (int i = 0; i < NI; ++i) dst[(o * NI + i) * 64 + lane] = a[o][i];
#pragma unroll
    for (int o = 0; o < NB; ++o)
#pragma unroll
      for (int i = 0; i < NI2; ++i) dst[(NA * NI + o * NI2 + i) * 64 + lane] = b[o][i];
    dst[(NF - 1) * 64 + lane] = f4{s0, s1, 0.f, 0.f};
  };
  auto add = [&](const f4* src) {
#pragma unroll
    for (int o = 0; o < NA; ++o)
#pragma unroll
      for (int i = 0; i < NI; ++i) a[o][i] = a[o][i] + src[(o * NI + i) * 64 + lane];
#pragma unroll
    for (int o = 0; o < NB; ++o)
#pragma unroll
      for (int i = 0; i < NI2; ++i) b[o][i] = b[o][i] + src[(NA * NI + o * NI2 + i) * 64 + lane];
    const f4 t = src[(NF - 1) * 64 + lane];
    s0 += t[0];
    s1 += t[1];
  };
  // round with stride st: wavefront (2 k + 1) st hands its sums to wavefront 2 k st through region k
#pragma unroll
  for (int st = 1; st < W; st *= 2) {
    __syncthreads();
    if ((w & (2 * st - 1)) == st) put(R + (w / (2 * st)) * NF * 64);
    __syncthreads();
    if ((w & (2 * st - 1)) == 0 && w + st < W) add(R + (w / (2 * st)) * NF * 64);
  }
}

// ---- three-layer backward in ONE pass (kr_train_epoch) -------------------------------------------------------------------
// Both passes below on the same row block back to back: dZ2 stays in registers (it is the B operand of W2^T dZ2 as it
// stands), A1 stays in its tile - no dZ2 images through memory (99 MB per cfg3 epoch), one launch less.  The three
// accumulator sets (128 registers) fit since the fragments moved to the LDS; 8 wavefronts per workgroup, one per CU:
// 8 x 16 KB of tiles (dOUT^T / A1 / dZ1 share one, A2 / dZ2 / X^T the other) + W3^T + W2^T = 152 KB.
constexpr int BW3 = 8;
constexpr int BW3_LDS = (64 + 64) * TP;
template <int ACT>
__global__ __launch_bounds__(64 * BW3) void mlp_bwd3_kernel(const FusedArgs A) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* const tu = wg_lds + wv * BW3_LDS;  // dOUT^T, then A1, then dZ1
  float* const tv = tu + 64 * TP;           // A2, then dZ2, then X^T
  float* const wl3 = wg_lds + BW3 * BW3_LDS;  // W3^T fragments: 4 x 8 x 64
  float* const wl2 = wl3 + B3A_WT;            // W2^T fragments: 4 x 16 x 64
  {
    const f4* s3 = reinterpret_cast<const f4*>(A.wt[2]);
    const f4* s2 = reinterpret_cast<const f4*>(A.wt[1]);
    f4* d3 = reinterpret_cast<f4*>(wl3);
    f4* d2 = reinterpret_cast<f4*>(wl2);
    for (int i = threadIdx.x; i < B3A_WT / 4; i += 64 * BW3) d3[i] = s3[i];
    for (int i = threadIdx.x; i < B3B_WT / 4; i += 64 * BW3) d2[i] = s2[i];
  }
  __syncthreads();
  const int64_t nblk = (A.Q + FR - 1) / FR;
  const int64_t wave0 = (int64_t)blockIdx.x * BW3 + wv, nwaves = (int64_t)gridDim.x * BW3;
  f4 aW1[4][2], aW2[4][4], aW3[2][4];
#pragma unroll
  for (int o = 0; o < 4; ++o) {
#pragma unroll
    for (int i = 0; i < 2; ++i) aW1[o][i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) aW2[o][i] = f4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int i = 0; i < 4; ++i) aW3[o][i] = f4{0.f, 0.f, 0.f, 0.f};
  float pb1 = 0.f, pb2 = 0.f, pbo = 0.f;
  for (int64_t rb = wave0; rb < nblk; rb += nwaves) {
    stage_rows_T(A.dout, rb * FR, A.Q, tu, lane);
    FChunk d2;
    {
      FChunk h1;
      {
        FChunk h;
        chunk_undump(h, A.a2d, rb, lane);
        chunk_undump(h1, A.a1d, rb, lane);
        chunk_to_T(tv, h, lane);  // A2
      }
      fsync();
      if (lane < 32) pbo += row_sum_T(tu, lane);
      wgrad_T<2, 4>(aW3, tu, tv, lane);  // dW3 += dOUT^T A2
      chunk_zero(d2);
      float bd[FT][8];
      load_bops_T(bd, tu, lane);
      fsync();
      chunk_to_T(tu, h1, lane);  // A1 over dOUT^T
      facc<4, 8>(d2.a, wl3, 8, 0, 0, lane, [&](int s, int k) { return bd[s][k]; });
    }
    chunk_mul_grad_T<ACT>(d2, tv, lane);
    fsync();
    chunk_to_T(tv, d2, lane);  // dZ2 (A2 is consumed)
    fsync();
    pb2 += row_sum_T(tv, lane);
    wgrad_T<4, 4>(aW2, tv, tu, lane);  // dW2 += dZ2^T A1
    fsync();
    // ---- what used to be the second pass
    stage_rows_T(A.x, rb * FR, A.Q, tv, lane);  // X^T over dZ2 (its first 32 rows)
    FChunk d1;
    chunk_zero(d1);
    facc<4, 16>(d1.a, wl2, 16, 0, 0, lane, [&](int s, int k) { return d2.a[k >> 2][s][k & 3]; });
    chunk_mul_grad_T<ACT>(d1, tu, lane);
    fsync();
    chunk_to_T(tu, d1, lane);  // dZ1 (A1 is consumed)
    fsync();
    pb1 += row_sum_T(tu, lane);
    wgrad_T<4, 2>(aW1, tu, tv, lane);  // dW1 += dZ1^T X
    fsync();
  }
  wg_tree_sum<BW3>(aW2, aW3, pb2, pbo, wg_lds, wv, lane);
  {
    f4 none[1][1] = {{f4{0.f, 0.f, 0.f, 0.f}}};
    float unused = 0.f;
    wg_tree_sum<BW3>(aW1, none, pb1, unused, wg_lds, wv, lane);
  }
  if (wv != 0) return;
  float* slab = A.slab + (size_t)blockIdx.x * A.P;
  wgrad_flush<2, 4>(aW3, slab + A.poff[4], A.nout, A.h2, 0, 0, lane);
  wgrad_flush<4, 4>(aW2, slab + A.poff[2], A.h2, A.h1, 0, 0, lane);
  wgrad_flush<4, 2>(aW1, slab + A.poff[0], A.h1, A.in, 0, 0, lane);
  if (lane < A.h1) slab[A.poff[1] + lane] = pb1;
  if (lane < A.h2) slab[A.poff[3] + lane] = pb2;
  if (lane < A.nout) slab[A.poff[5] + lane] = pbo;
}

template <int ACT>
__global__ __launch_bounds__(64 * WPB, 2) void mlp_bwd3a_kernel(const FusedArgs A) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // dOUT^T is dead once its B operands are in registers, A1 is needed last: they share a tile (A1 waits in registers until
  // then), which makes room for the W3^T fragments - one copy per workgroup, as in the other kernels
  float* const ts = wg_lds + wv * B3A_LDS;  // dOUT^T
  float* const tu = ts;                     // A1 (after dOUT^T)
  float* const tv = tu + 64 * TP;           // A2 -> dZ2
  float* const wl = wg_lds + WPB * B3A_LDS;
  {
    const f4* src = reinterpret_cast<const f4*>(A.wt[2]);
    f4* dst = reinterpret_cast<f4*>(wl);
    for (int i = threadIdx.x; i < B3A_WT / 4; i += 64 * WPB) dst[i] = src[i];
  }
  __syncthreads();
  const int64_t nblk = (A.Q + FR - 1) / FR;
  const int64_t wave0 = (int64_t)blockIdx.x * WPB + wv, nwaves = (int64_t)gridDim.x * WPB;
  f4 aW2[4][4], aW3[2][4];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int i = 0; i < 4; ++i) aW2[o][i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int i = 0; i < 4; ++i) aW3[o][i] = f4{0.f, 0.f, 0.f, 0.f};
  float pb2 = 0.f, pbo = 0.f;  // bias gradients of unit `lane`: row sums of the dZ2 / dOUT^T tiles
  FSTAMP_DECL;
  FSTAMP_T0;
  for (int64_t rb = wave0; rb < nblk; rb += nwaves) {
    stage_rows_T(A.dout, rb * FR, A.Q, ts, lane);
    FSTAMP(0);
    FChunk h1;
    {
      FChunk h;
      chunk_undump(h, A.a2d, rb, lane);
      chunk_undump(h1, A.a1d, rb, lane);
      chunk_to_T(tv, h, lane);  // A2
    }
    fsync();
    FSTAMP(1);
    if (lane < 32) pbo += row_sum_T(ts, lane);
    wgrad_T<2, 4>(aW3, ts, tv, lane);  // dW3 += dOUT^T A2
    FSTAMP(2);
    FChunk d2;
    chunk_zero(d2);
    {
      float bd[FT][8];
      load_bops_T(bd, ts, lane);
      fsync();
      chunk_to_T(tu, h1, lane);  // A1 over dOUT^T
      facc<4, 8>(d2.a, wl, 8, 0, 0, lane, [&](int s, int k) { return bd[s][k]; });
    }
    FSTAMP(3);
    chunk_mul_grad_T<ACT>(d2, tv, lane);
    chunk_dump(A.dz2, rb, d2, lane);
    fsync();
    chunk_to_T(tv, d2, lane);  // dZ2 (A2 is consumed)
    fsync();
    FSTAMP(4);
    pb2 += row_sum_T(tv, lane);
    wgrad_T<4, 4>(aW2, tv, tu, lane);  // dW2 += dZ2^T A1
    fsync();
    FSTAMP(5);
  }
  FSTAMP_OUT(1, wave0);
  wg_tree_sum(aW2, aW3, pb2, pbo, wg_lds, wv, lane);
  if (wv != 0) return;
  float* slab = A.slab + (size_t)blockIdx.x * A.P;
  wgrad_flush<2, 4>(aW3, slab + A.poff[4], A.nout, A.h2, 0, 0, lane);
  wgrad_flush<4, 4>(aW2, slab + A.poff[2], A.h2, A.h1, 0, 0, lane);
  if (lane < A.h2) slab[A.poff[3] + lane] = pb2;
  if (lane < A.nout) slab[A.poff[5] + lane] = pbo;
}

template <int ACT>
__global__ __launch_bounds__(64 * WPB, 2) void mlp_bwd3b_kernel(const FusedArgs A) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* const ts = wg_lds + wv * B3B_LDS;  // X^T
  float* const tu = ts + 32 * TP;           // A1 -> dZ1
  // W2^T fragments (4 tiles x 16 k-steps x 64 lanes: 16 KB) live in the LDS, one copy per workgroup: fetched from memory
  // inside the products they sat in the wavefront's in-order load queue behind the row block's own loads
  float* const wl = wg_lds + WPB * B3B_LDS;
  {
    const f4* src = reinterpret_cast<const f4*>(A.wt[1]);
    f4* dst = reinterpret_cast<f4*>(wl);
    for (int i = threadIdx.x; i < B3B_WT / 4; i += 64 * WPB) dst[i] = src[i];
  }
  __syncthreads();
  const int64_t nblk = (A.Q + FR - 1) / FR;
  const int64_t wave0 = (int64_t)blockIdx.x * WPB + wv, nwaves = (int64_t)gridDim.x * WPB;
  f4 aW1[4][2];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int i = 0; i < 2; ++i) aW1[o][i] = f4{0.f, 0.f, 0.f, 0.f};
  float pb1 = 0.f;
  FSTAMP_DECL;
  FSTAMP_T0;
  for (int64_t rb = wave0; rb < nblk; rb += nwaves) {
    stage_rows_T(A.x, rb * FR, A.Q, ts, lane);
    FSTAMP(0);
    FChunk d2;
    chunk_undump(d2, A.dz2, rb, lane);
    {
      FChunk h1;
      chunk_undump(h1, A.a1d, rb, lane);
      chunk_to_T(tu, h1, lane);  // A1
    }
    FSTAMP(1);
    FChunk d1;
    chunk_zero(d1);
    facc<4, 16>(d1.a, wl, 16, 0, 0, lane, [&](int s, int k) { return d2.a[k >> 2][s][k & 3]; });
    fsync();
    FSTAMP(2);
    chunk_mul_grad_T<ACT>(d1, tu, lane);
    fsync();
    chunk_to_T(tu, d1, lane);  // dZ1 (A1 is consumed)
    fsync();
    FSTAMP(3);
    pb1 += row_sum_T(tu, lane);
    wgrad_T<4, 2>(aW1, tu, ts, lane);  // dW1 += dZ1^T X
    fsync();
    FSTAMP(4);
  }
  FSTAMP_OUT(2, wave0);
  {
    f4 none[1][1] = {{f4{0.f, 0.f, 0.f, 0.f}}};
    float unused = 0.f;
    wg_tree_sum(aW1, none, pb1, unused, wg_lds, wv, lane);
  }
  if (wv != 0) return;
  float* slab = A.slab + (size_t)blockIdx.x * A.P;
  wgrad_flush<4, 2>(aW1, slab + A.poff[0], A.h1, A.in, 0, 0, lane);
  if (lane < A.h1) slab[A.poff[1] + lane] = pb1;
}

// Two-layer networks in -> H1 -> out with any H1: wave w handles hidden chunk w % c1 (64 units) over the row blocks
// w / c1, w / c1 + G / c1, ...
// Workgroups of WPB wavefronts that all work on the SAME hidden chunk: the chunk's W1 fragments, its W2^T fragments and its
// bias fragments (8 + 8 + 4 KB) sit in the LDS once per workgroup (fetched from memory inside the products they queued
// behind the row block's loads in the wavefront's in-order return path - the same finding as in the three-layer kernels),
// dZ1 takes the tile of A1 (dead by then), and the four wavefronts leave ONE slab.
constexpr int B2_LDS = (32 + 64) * TP;                    // floats per wavefront: X^T / dOUT^T, A1 -> dZ1
constexpr int B2_FRAG = 4 * 8 * 64 + 4 * 8 * 64 + 4 * 4 * 64;  // W1 chunk, W2^T chunk, bias chunk
template <int ACT>
__global__ __launch_bounds__(64 * WPB, 2) void mlp_bwd2_kernel(const FusedArgs A) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* const ts = wg_lds + wv * B2_LDS;  // X^T, then dOUT^T, then X^T again
  float* const tu = ts + 32 * TP;          // A1
  float* const tv = tu;                    // dZ1 (A1 is consumed when it is written)
  float* const wl0 = wg_lds + WPB * B2_LDS;
  float* const wl1 = wl0 + 4 * 8 * 64;
  float* const bl = wl1 + 4 * 8 * 64;
  const int64_t nblk = (A.Q + FR - 1) / FR;
  const int nchunk = A.c1;
  const int chunk = blockIdx.x % nchunk;
  const int64_t group = blockIdx.x / nchunk, ngroups = gridDim.x / nchunk;
  {
    const f4* s0 = reinterpret_cast<const f4*>(A.wf[0] + (size_t)4 * chunk * 8 * 64);
    const f4* s1 = reinterpret_cast<const f4*>(A.wt[1] + (size_t)4 * chunk * 8 * 64);
    const f4* s2 = reinterpret_cast<const f4*>(A.bfr[0] + (size_t)4 * chunk * 4 * 64);
    f4* d0 = reinterpret_cast<f4*>(wl0);
    f4* d1 = reinterpret_cast<f4*>(wl1);
    f4* d2 = reinterpret_cast<f4*>(bl);
    for (int i = threadIdx.x; i < 512; i += 64 * WPB) { d0[i] = s0[i]; d1[i] = s1[i]; }
    for (int i = threadIdx.x; i < 256; i += 64 * WPB) d2[i] = s2[i];
  }
  __syncthreads();
  f4 aW1[4][2], aWo[2][4];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int i = 0; i < 2; ++i) aW1[o][i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int i = 0; i < 4; ++i) aWo[o][i] = f4{0.f, 0.f, 0.f, 0.f};
  float pb1 = 0.f, pbo = 0.f;
  // the rows of X and dOUT of a block are requested while the previous block finishes (registers: the fragments in the
  // LDS freed 45 of them); X is staged twice from the same registers
  f4 xr[RNV], dr[RNV];
  {
    const int64_t rb = group * WPB + wv;
    rows_request(xr, A.x, rb * FR, rb < nblk ? A.Q : 0, lane);
    rows_request(dr, A.dout, rb * FR, rb < nblk ? A.Q : 0, lane);
  }
  for (int64_t rb = group * WPB + wv; rb < nblk; rb += ngroups * WPB) {
    rows_to_T(xr, ts, lane);
    fsync();
    {
      FChunk h1;
      float bin[FT][8];
      load_bops_T(bin, ts, lane);
      chunk_set_bias(h1, bl, 0, lane);
      facc<4, 8>(h1.a, wl0, 8, 0, 0, lane, [&](int s, int k) { return bin[s][k]; });
      fsync();
      rows_to_T(dr, ts, lane);
      chunk_act_only<ACT>(h1);
      chunk_to_T(tu, h1, lane);  // A1 chunk
    }
    fsync();
    if (chunk == 0 && lane < 32) pbo += row_sum_T(ts, lane);
    wgrad_T<2, 4>(aWo, ts, tu, lane);  // dW2[:, chunk] += dOUT^T A1
    // dZ1 = (W2^T[chunk] dOUT) * act'(Z1)
    FChunk d1;
    chunk_zero(d1);
    {
      float bd[FT][8];
      load_bops_T(bd, ts, lane);
      facc<4, 8>(d1.a, wl1, 8, 0, 0, lane, [&](int s, int k) { return bd[s][k]; });
    }
    chunk_mul_grad_T<ACT>(d1, tu, lane);
    fsync();
    rows_to_T(xr, ts, lane);
    {
      const int64_t rn = rb + ngroups * WPB;
      rows_request(xr, A.x, rn * FR, rn < nblk ? A.Q : 0, lane);
      rows_request(dr, A.dout, rn * FR, rn < nblk ? A.Q : 0, lane);
    }
    chunk_to_T(tv, d1, lane);
    fsync();
    pb1 += row_sum_T(tv, lane);
    wgrad_T<4, 2>(aW1, tv, ts, lane);  // dW1[chunk] += dZ1^T X
    fsync();
  }
  wg_tree_sum(aW1, aWo, pb1, pbo, wg_lds, wv, lane);
  if (wv != 0) return;
  float* slab = A.slab + (size_t)group * A.P;  // one slab per group of row-block streams; its chunks write disjoint parts
  wgrad_flush<2, 4>(aWo, slab + A.poff[2], A.nout, A.h1, 0, 64 * chunk, lane);
  wgrad_flush<4, 2>(aW1, slab + A.poff[0], A.h1, A.in, 64 * chunk, 0, lane);
  if (64 * chunk + lane < A.h1) slab[A.poff[1] + 64 * chunk + lane] = pb1;
  if (chunk == 0 && lane < A.nout) slab[A.poff[3] + lane] = pbo;
}

// dW[k] += sum over the slabs.  Thread (parameter i, group g) adds up the slabs g, g + RG, g + 2 RG, ... (consecutive
// threads read consecutive floats of a slab) and contributes with ONE float atomic: RG adders per address, spread
// over the whole parameter range - the uncontended regime of the memory-side atomics.
constexpr int RG = 32;
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const FusedArgs A) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  if (i >= A.nparams) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int w = g;
  for (; w + 3 * RG < A.nslab; w += 4 * RG) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += A.slab[(size_t)(w + e * RG) * A.P + i];
  }
  for (; w < A.nslab; w += RG) acc[0] += A.slab[(size_t)w * A.P + i];
  const float sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  int seg = 5;
#pragma unroll
  for (int k = 4; k >= 0; --k)
    if (i < A.poff[k + 1]) seg = k;
  float* dst = (seg & 1) ? A.db[seg >> 1] : A.dW[seg >> 1];
  atomicAdd(&dst[i - A.poff[seg]], sum);
}

// ---- tail of an epoch: slab sum + loss sum + Adam + clamp + plateau schedule + fragment update, ONE launch ----------
// (physics_train.py:289-304: optimizer.step(), scheduler.step(total_loss), the weight clamp.)  A workgroup owns 64
// consecutive parameters and does everything for them - no workgroup waits for another, no atomics, and the order of
// every sum is fixed, so an epoch is reproducible bit for bit:
//   g[i] += sum over the slabs (16 slab groups x 16-byte loads, added up through the LDS);
//   the workgroup that holds the loss slot g[nparams] adds the forward kernel's per-workgroup loss partials to it;
//   update != 0: torch.optim.Adam + clamp on p[i] (adam_plateau_kernel's arithmetic), g[i] = 0, and the new value goes
//   straight to its places in the MFMA fragment buffers the next epoch's kernels read (pack_all_kernel's layout: one
//   forward fragment slot, one transposed slot for layers > 0, sixteen bias slots) - the separate packing launch of
//   every epoch is gone; the thread of the loss slot steps ReduceLROnPlateau.
// update == 0 stops after the sums: the gradients and the loss are complete in g for a data-parallel all-reduce, and a
// second launch with nslab = nlpart = 0 does the rest.
struct TailArgs {
  const float* slab;
  int nslab, P, nparams;
  float *p, *g, *m, *v;
  const float* lower;
  double* sched;
  int parity;
  float inv_bc1, b1, b2, inv_sqrt_bc2, eps, wd;
  double factor, threshold, min_lr;
  int patience;
  float* loss_log;
  const float* lpart;
  int nlpart;
  int update;
  int L;
  int poff[7];
  int in[3], out[3], ks[3], kst[3], natural[3];
  float* wf[3];
  float* bf[3];
  float* wt[3];
};
constexpr int TAIL_T = 1024, TAIL_G = TAIL_T / 16;  // threads of a workgroup, slab groups (16 lanes x 16 bytes = 64 parameters each)
__global__ __launch_bounds__(TAIL_T) void train_tail_kernel(const TailArgs T) {
  __shared__ __attribute__((aligned(16))) float red[TAIL_G][64];
  __shared__ float lred[TAIL_T];
  const int tid = threadIdx.x, c0 = blockIdx.x * 64;
  // the optimizer state of this workgroup's parameters is requested first, so that it travels with the slabs
  float g_in = 0.f, p_in = 0.f, m_in = 0.f, v_in = 0.f, lo_in = 0.f;
  if (tid < 64 && c0 + tid <= T.nparams) {
    g_in = T.g[c0 + tid];
    if (T.update && c0 + tid < T.nparams) {
      p_in = T.p[c0 + tid];
      m_in = T.m[c0 + tid];
      v_in = T.v[c0 + tid];
      if (T.lower) lo_in = T.lower[c0 + tid];
    }
  }
  {
    const int l16 = tid & 15, sg = tid >> 4;
    f4 a0 = f4{0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (c0 + 4 * l16 < T.P) {
      const float* src = T.slab + c0 + 4 * l16;
      int w = sg;
      for (; w + 3 * TAIL_G < T.nslab; w += 4 * TAIL_G) {
        const f4 v0 = *reinterpret_cast<const f4*>(src + (size_t)w * T.P);
        const f4 v1 = *reinterpret_cast<const f4*>(src + (size_t)(w + TAIL_G) * T.P);
        const f4 v2 = *reinterpret_cast<const f4*>(src + (size_t)(w + 2 * TAIL_G) * T.P);
        const f4 v3 = *reinterpret_cast<const f4*>(src + (size_t)(w + 3 * TAIL_G) * T.P);
        a0 = a0 + v0; a1 = a1 + v1; a2 = a2 + v2; a3 = a3 + v3;
      }
      for (; w < T.nslab; w += TAIL_G) a0 = a0 + *reinterpret_cast<const f4*>(src + (size_t)w * T.P);
    }
    *reinterpret_cast<f4*>(&red[sg][4 * l16]) = (a0 + a1) + (a2 + a3);
  }
  const bool loss_blk = T.nparams >= c0 && T.nparams < c0 + 64;  // (uniform over the workgroup)
  float lsum = 0.f;
  if (loss_blk && T.nlpart > 0) {
    for (int j = tid; j < T.nlpart; j += TAIL_T) lsum += T.lpart[j];
    lred[tid] = lsum;
  }
  __syncthreads();
  if (loss_blk && T.nlpart > 0) {
    for (int o = TAIL_T / 2; o > 0; o >>= 1) {
      if (tid < o) lred[tid] += lred[tid + o];
      __syncthreads();
    }
    lsum = lred[0];
  }
  if (T.nslab > 0) {  // 64 x TAIL_G partial sums -> 64 x 4 (every thread of the first four wavefronts adds sixteen)
    float part = 0.f;
    if (tid < 256) {
#pragma unroll
      for (int k = 0; k < TAIL_G / 4; ++k) part += red[(tid >> 6) * (TAIL_G / 4) + k][tid & 63];
    }
    __syncthreads();
    if (tid < 256) red[tid >> 6][tid & 63] = part;
    __syncthreads();
  }
  if (tid >= 64) return;
  const int i = c0 + tid;
  if (i > T.nparams) return;
  if (i == T.nparams) {  // the loss slot
    const float cur_f = g_in + lsum;
    if (!T.update) {
      T.g[i] = cur_f;
      return;
    }
    const double lr = T.sched[T.parity];
    const double cur = (double)cur_f;
    double best = T.sched[2], bad = T.sched[3], nred = T.sched[5];
    if (cur < best * (1.0 - T.threshold)) { best = cur; bad = 0.0; }
    else bad += 1.0;
    double next = lr;
    if (bad > (double)T.patience) {
      const double cand = fmax(lr * T.factor, T.min_lr);
      if (lr - cand > 1e-8) { next = cand; nred += 1.0; }
      bad = 0.0;
    }
    T.sched[T.parity ^ 1] = next;
    T.sched[2] = best; T.sched[3] = bad; T.sched[4] = cur; T.sched[5] = nred;
    if (T.loss_log) *T.loss_log = cur_f;
    T.g[i] = 0.f;
    return;
  }
  float gi = g_in;
  if (T.nslab > 0) gi += (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  if (!T.update) {
    T.g[i] = gi;
    return;
  }
  float pi = p_in;
  {
    const float step_size = (float)(T.sched[T.parity] * (double)T.inv_bc1);
    if (T.wd != 0.f) gi = fmaf(T.wd, pi, gi);
    const float mi = fmaf(T.b1, m_in, (1.f - T.b1) * gi);
    const float vi = fmaf(T.b2, v_in, (1.f - T.b2) * gi * gi);
    T.m[i] = mi;
    T.v[i] = vi;
    const float denom = sqrtf(vi) * T.inv_sqrt_bc2 + T.eps;
    pi -= step_size * (mi / denom);
    if (T.lower) pi = fmaxf(pi, lo_in);
    T.p[i] = pi;
    T.g[i] = 0.f;
  }
  int seg = 0;
#pragma unroll
  for (int k = 1; k < 6; ++k)
    if (i >= T.poff[k]) seg = k;
  const int k = seg >> 1, r = i - T.poff[seg];
  if (seg & 1) {  // bias b_k[u]: bf[(t*4 + r4)*64 + lane] = b[16 t + 4 (lane >> 4) + r4] for the sixteen lanes of a quad
    const int u = r;
    float* dst = T.bf[k] + ((size_t)(u >> 4) * 4 + (u & 3)) * 64 + 16 * ((u & 15) >> 2);
#pragma unroll
    for (int c = 0; c < 16; ++c) dst[c] = pi;
  } else {
    const int in = T.in[k], uo = r / in, ui = r - uo * in;
    {
      const int sfw = k == 0 ? ui >> 2 : 4 * (ui >> 4) + (ui & 3);
      const int q = k == 0 ? ui & 3 : (ui & 15) >> 2;
      const int lane = 16 * q + (uo & 15);
      T.wf[k][(((size_t)(uo >> 4) * (T.ks[k] >> 2) + (sfw >> 2)) * 64 + lane) * 4 + (sfw & 3)] = pi;
    }
    if (k > 0) {
      const int st = T.natural[k] ? uo >> 2 : 4 * (uo >> 4) + (uo & 3);
      const int q = T.natural[k] ? uo & 3 : (uo & 15) >> 2;
      const int lane = 16 * q + (ui & 15);
      T.wt[k][(((size_t)(ui >> 4) * (T.kst[k] >> 2) + (st >> 2)) * 64 + lane) * 4 + (st & 3)] = pi;
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------
static bool merged_bwd3() {
  static const bool on = [] { const char* e = getenv("KR_BWD3_MERGED"); return !e || atoi(e) != 0; }();
  return on;
}
bool fused_mlp_supported(int n_layers, const int32_t* dims, const int32_t* acts, int in_pad) {
  if (n_layers != 2 && n_layers != 3) return false;
  if (in_pad != 32 || dims[0] > 32 || dims[n_layers] > 32 || dims[n_layers] < 1) return false;
  if (acts[n_layers - 1] != KR_ACT_NONE) return false;
  if (n_layers == 3 && (dims[1] > 64 || dims[2] > 64 || acts[0] != acts[1])) return false;
  return true;
}

// workspace for the packed fragments (floats)
static size_t fused_frag_floats(int n_layers, const int32_t* dims) {
  size_t n = 0;
  int prev_tiles = 2;
  for (int k = 0; k < n_layers; ++k) {
    const bool last = k == n_layers - 1;
    const int tiles = last ? 2 : ((dims[k + 1] + 63) / 64) * 4;
    const int ks = k == 0 ? 8 : prev_tiles * 4;
    n += (size_t)tiles * ks * 64 + (size_t)tiles * 4 * 64;        // forward + bias
    if (k > 0) n += (size_t)prev_tiles * (last ? 8 : tiles * 4) * 64;  // transposed
    prev_tiles = tiles;
  }
  return n;
}
size_t fused_ws_bytes(int n_layers, const int32_t* dims, int64_t Q) {
  size_t n = ((fused_frag_floats(n_layers, dims) + 63) & ~size_t(63)) * sizeof(float) + 256;
  if (n_layers == 3) n += 3 * (size_t)((Q + FR - 1) / FR) * 4 * FT * 64 * 16;  // A1, A2 (forward -> backward), dZ2 (pass A -> B)
  size_t P = 0;
  for (int k = 0; k < n_layers; ++k) P += (size_t)dims[k] * dims[k + 1] + dims[k + 1];
  n += (size_t)F_MAXW * ((P + 63) & ~size_t(63)) * sizeof(float);  // slabs of partial weight gradients
  return n;
}

// lays the fragment buffers out in `ws`; do_pack launches the packing kernels (forward call), otherwise
// the fragments packed by the matching forward call are reused (backward call)
static int fused_pack(FusedArgs& A, int n_layers, const int32_t* dims, const float* const* W, const float* const* b,
                      float* ws, bool do_pack, hipStream_t s, PackArgs* layout = nullptr) {
  A.L = n_layers;
  A.stamps = g_fused_stamps;
  A.in = dims[0];
  A.h1 = dims[1];
  A.h2 = n_layers == 3 ? dims[2] : 0;
  A.nout = dims[n_layers];
  A.c1 = (dims[1] + 63) / 64;
  float* p = ws;
  int prev_tiles = 2;
  PackArgs PA{};
  PA.L = n_layers;
  for (int k = 0; k < n_layers; ++k) {
    const bool last = k == n_layers - 1;
    const int tiles = last ? 2 : ((dims[k + 1] + 63) / 64) * 4;
    const int ks = k == 0 ? 8 : prev_tiles * 4;
    float* wf = p; p += (size_t)tiles * ks * 64;
    float* bf = p; p += (size_t)tiles * 4 * 64;
    A.wf[k] = wf; A.bfr[k] = bf; A.ks[k] = ks;
    PA.W[k] = W[k]; PA.b[k] = b ? b[k] : nullptr; PA.wf[k] = wf; PA.bf[k] = bf;
    PA.in[k] = dims[k]; PA.out[k] = dims[k + 1]; PA.tiles[k] = tiles; PA.ks[k] = ks;
    if (k > 0) {
      // dA_{k-1}[in unit][sample] = sum_out W_k[out][in] dZ_k[out][sample]: k-steps run over the outputs of layer k
      const int kst = last ? 8 : tiles * 4;
      float* wt = p; p += (size_t)prev_tiles * kst * 64;
      A.wt[k] = wt; A.kst[k] = kst;
      PA.wt[k] = wt; PA.in_tiles[k] = prev_tiles; PA.kst[k] = kst; PA.natural[k] = last ? 1 : 0;
    }
    prev_tiles = tiles;
  }
  if (do_pack) hipLaunchKernelGGL(pack_all_kernel, dim3(64), dim3(256), 0, s, PA);
  KR_HIP(hipGetLastError());
  if (layout) *layout = PA;
  return KR_OK;
}

template <typename K>
static void launch_by_act(int act, K&& fn) {
  switch (act) {
    case KR_ACT_TANH: fn(std::integral_constant<int, KR_ACT_TANH>{}); break;
    case KR_ACT_SOFTPLUS: fn(std::integral_constant<int, KR_ACT_SOFTPLUS>{}); break;
    case KR_ACT_RELU: fn(std::integral_constant<int, KR_ACT_RELU>{}); break;
    case KR_ACT_ELU: fn(std::integral_constant<int, KR_ACT_ELU>{}); break;
    default: fn(std::integral_constant<int, KR_ACT_NONE>{}); break;
  }
}

int fused_mlp_forward(int64_t Q, int n_layers, const int32_t* dims, const int32_t* acts, const float* const* W,
                      const float* const* b, const float* x, float* out, void* ws, hipStream_t s,
                      const FusedLoss* fl, bool do_pack, int* n_partials) {
  FusedArgs A{};
  A.Q = Q; A.x = x; A.out = out;
  if (fl) {
    A.lbase = fl->base; A.ltarget = fl->target_rows; A.ldout = fl->dout; A.lds = fl->ds;
    A.lw = loss_weights(fl->inv_denom, fl->K);
    A.lpart = static_cast<float*>(fl->scratch);
  }
  int rc = fused_pack(A, n_layers, dims, W, b, static_cast<float*>(ws), do_pack, s);
  if (rc) return rc;
  const int64_t nblk = (Q + FR - 1) / FR;
  const int grid = (int)(nblk < 4096 ? nblk : 4096);
  if (n_layers == 3) {  // hidden activations stay in the workspace for the backward passes (same layout there)
    float* wsf = static_cast<float*>(ws) + ((fused_frag_floats(n_layers, dims) + 63) & ~size_t(63));
    A.a1d = wsf;
    A.a2d = wsf + (size_t)nblk * 4 * FT * 64 * 4;
  }
  int nparts = grid;
  int lrc = KR_OK;
  if (n_layers == 3 && dims[3] <= 25) {
    const int wgs = (int)((nblk + FW3 - 1) / FW3 < 256 ? (nblk + FW3 - 1) / FW3 : 256);  // one workgroup per CU
    nparts = wgs * FW3;
    launch_by_act(acts[0], [&](auto act) {
      constexpr int a = decltype(act)::value;
      const size_t lb = sizeof(float) * (FW3_FRAG + FW3 * FW3_WAVE);
      if ((lrc = dyn_lds(reinterpret_cast<const void*>(&mlp_fwd3_kernel<a>), lb))) return;
      hipLaunchKernelGGL((mlp_fwd3_kernel<a>), dim3(wgs), dim3(64 * FW3), lb, s, A);
    });
  } else if (n_layers == 2 && A.c1 <= 8 && dims[2] <= 25) {
    const int wgs = (int)((nblk + FW2 - 1) / FW2 < 256 ? (nblk + FW2 - 1) / FW2 : 256);
    nparts = wgs * FW2;
    launch_by_act(acts[0], [&](auto act) {
      constexpr int a = decltype(act)::value;
      const size_t lb = sizeof(float) * ((size_t)A.c1 * (4096 + 64) + 32 + FW2 * FW2_OUT);
      if ((lrc = dyn_lds(reinterpret_cast<const void*>(&mlp_fwd2_kernel<a>), lb))) return;
      hipLaunchKernelGGL((mlp_fwd2_kernel<a>), dim3(wgs), dim3(64 * FW2), lb, s, A);
    });
  } else {
    launch_by_act(acts[0], [&](auto act) {
      hipLaunchKernelGGL((mlp_fwd_fused_kernel<decltype(act)::value>), dim3(grid), dim3(64), 0, s, A);
    });
  }
  if (lrc) return lrc;
  KR_HIP(hipGetLastError());
  if (fl && n_partials) {
    *n_partials = nparts;  // the caller's next kernel adds them up (train_tail_kernel)
  } else if (fl) {
    hipLaunchKernelGGL(loss_partials_kernel, dim3(1), dim3(256), 0, s, A.lpart, nparts, fl->loss);
    KR_HIP(hipGetLastError());
  }
  return KR_OK;
}

int fused_mlp_backward(int64_t Q, int n_layers, const int32_t* dims, const int32_t* acts, const float* const* W,
                       const float* x, const float* dout, void* ws, float* const* dW, float* const* db,
                       hipStream_t s, FusedSlabs* leave) {
  FusedArgs A{};
  A.Q = Q; A.x = x; A.dout = dout;
  if (!leave)
    for (int k = 0; k < n_layers; ++k) { A.dW[k] = dW[k]; A.db[k] = db[k]; }
  int rc = fused_pack(A, n_layers, dims, W, nullptr, static_cast<float*>(ws), false, s);
  if (rc) return rc;
  const int64_t nblk = (Q + FR - 1) / FR;
  const int nchunk = n_layers == 2 ? A.c1 : 1;
  int64_t waves = nblk * nchunk;
  if (waves > F_MAXW) waves = F_MAXW / nchunk * nchunk;  // as many workgroups as the LDS of the chip holds
  if (waves < nchunk) waves = nchunk;
  const int grid = (int)waves;
  // workspace behind the packed fragments: [A1, A2, dZ2 images of a three-layer network] [gradient slabs]
  float* wsf = static_cast<float*>(ws) + ((fused_frag_floats(n_layers, dims) + 63) & ~size_t(63));
  if (n_layers == 3) {
    const size_t img = (size_t)nblk * 4 * FT * 64 * 4;
    A.a1d = wsf; A.a2d = wsf + img; A.dz2 = wsf + 2 * img;
    wsf += 3 * img;
  }
  int P = 0;
  for (int k = 0; k < n_layers; ++k) {
    A.poff[2 * k] = P; P += dims[k] * dims[k + 1];
    A.poff[2 * k + 1] = P; P += dims[k + 1];
  }
  for (int k = 2 * n_layers; k < 6; ++k) A.poff[k] = P;
  A.nparams = P;
  A.P = (P + 63) & ~63;
  A.slab = wsf;
  if (n_layers == 3) {
    const int waves3 = (int)(nblk < F_MAXW ? nblk : F_MAXW);
    const int grid3 = (waves3 + WPB - 1) / WPB;  // wavefronts beyond the row blocks only take part in the sum
    A.nslab = grid3;
    int lrc = KR_OK;
    if (merged_bwd3()) {  // both passes in one launch, one workgroup per CU (kr_train_epoch and, since round 5, kr_mlp_backward)
      const int wgs = (int)((nblk + BW3 - 1) / BW3 < 256 ? (nblk + BW3 - 1) / BW3 : 256);
      A.nslab = wgs;
      launch_by_act(acts[0], [&](auto act) {
        constexpr int a = decltype(act)::value;
        const size_t lm = sizeof(float) * (BW3 * BW3_LDS + B3A_WT + B3B_WT);
        if ((lrc = dyn_lds(reinterpret_cast<const void*>(&mlp_bwd3_kernel<a>), lm))) return;
        hipLaunchKernelGGL((mlp_bwd3_kernel<a>), dim3(wgs), dim3(64 * BW3), lm, s, A);
      });
    } else
    launch_by_act(acts[0], [&](auto act) {
      constexpr int a = decltype(act)::value;
      const size_t la = sizeof(float) * (WPB * B3A_LDS + B3A_WT), lb = sizeof(float) * (WPB * B3B_LDS + B3B_WT);
      if ((lrc = dyn_lds(reinterpret_cast<const void*>(&mlp_bwd3a_kernel<a>), la))) return;
      if ((lrc = dyn_lds(reinterpret_cast<const void*>(&mlp_bwd3b_kernel<a>), lb))) return;
      hipLaunchKernelGGL((mlp_bwd3a_kernel<a>), dim3(grid3), dim3(64 * WPB), la, s, A);
      hipLaunchKernelGGL((mlp_bwd3b_kernel<a>), dim3(grid3), dim3(64 * WPB), lb, s, A);
    });
    if (lrc) return lrc;
  } else {
    // groups of WPB row-block streams x hidden chunks: as many workgroups as the LDS of the chip holds
    int64_t ngroups = (nblk + WPB - 1) / WPB;
    const int64_t cap = F_MAXW / WPB / nchunk;
    if (ngroups > cap) ngroups = cap;
    if (ngroups < 1) ngroups = 1;
    A.nslab = (int)ngroups;
    int lrc = KR_OK;
    launch_by_act(acts[0], [&](auto act) {
      constexpr int a = decltype(act)::value;
      const size_t l2 = sizeof(float) * (WPB * B2_LDS + B2_FRAG);
      if ((lrc = dyn_lds(reinterpret_cast<const void*>(&mlp_bwd2_kernel<a>), l2))) return;
      hipLaunchKernelGGL((mlp_bwd2_kernel<a>), dim3((unsigned)(ngroups * nchunk)), dim3(64 * WPB), l2, s, A);
    });
    if (lrc) return lrc;
  }
  if (leave) {  // the slabs stay where they are: train_tail_kernel sums them
    leave->slab = A.slab; leave->nslab = A.nslab; leave->P = A.P; leave->nparams = A.nparams;
    for (int k = 0; k < 6; ++k) leave->poff[k] = A.poff[k];
    KR_HIP(hipGetLastError());
    return KR_OK;
  }
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((P + 255) / 256, RG), dim3(256), 0, s, A);
  KR_HIP(hipGetLastError());
  return KR_OK;
}

// One epoch on one rank (kr_train_epoch).  phase 0: forward + loss, backward, tail with the update; 1: the same with the
// tail stopping after the sums (gradients and loss complete in E.g); 2: only the update, from E.g as it stands.
int fused_train_epoch(const FusedEpoch& E, hipStream_t s) {
  const int n = E.n_layers;
  const float* W[3];
  const float* b[3];
  size_t off = 0;
  for (int k = 0; k < n; ++k) {
    W[k] = E.p + off; off += (size_t)E.dims[k] * E.dims[k + 1];
    b[k] = E.p + off; off += (size_t)E.dims[k + 1];
  }
  const int nparams = (int)off;
  FusedSlabs sl{};
  int nlpart = 0;
  int rc;
  if (E.phase != 2) {
    FusedLoss fl{E.base, E.target_rows, E.dout, E.g + nparams, E.loss_scratch, E.ds, E.inv_denom, E.K};
    rc = fused_mlp_forward(E.Q, n, E.dims, E.acts, W, b, E.x, nullptr, E.ws, s, &fl, E.pack, &nlpart);
    if (rc) return rc;
    rc = fused_mlp_backward(E.Q, n, E.dims, E.acts, W, E.x, E.dout, E.ws, nullptr, nullptr, s, &sl);
    if (rc) return rc;
  }
  FusedArgs A{};
  PackArgs PA{};
  rc = fused_pack(A, n, E.dims, W, b, static_cast<float*>(E.ws), false, s, &PA);
  if (rc) return rc;
  TailArgs T{};
  T.slab = sl.slab; T.nslab = sl.nslab; T.nparams = nparams;
  T.P = sl.nslab ? sl.P : (nparams + 63) & ~63;
  T.p = E.p; T.g = E.g; T.m = E.m; T.v = E.v; T.lower = E.lower;
  T.sched = E.sched; T.parity = (int)((E.step - 1) & 1);
  const double bc1 = 1.0 - std::pow(E.beta1, (double)E.step), bc2 = 1.0 - std::pow(E.beta2, (double)E.step);
  T.inv_bc1 = (float)(1.0 / bc1); T.b1 = (float)E.beta1; T.b2 = (float)E.beta2;
  T.inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2)); T.eps = (float)E.eps; T.wd = (float)E.weight_decay;
  T.factor = E.factor; T.threshold = E.threshold; T.min_lr = E.min_lr; T.patience = E.patience;
  T.loss_log = E.loss_log;
  T.lpart = static_cast<const float*>(E.loss_scratch); T.nlpart = nlpart;
  T.update = E.phase != 1;
  T.L = n;
  int P = 0;
  for (int k = 0; k < n; ++k) {
    T.poff[2 * k] = P; P += E.dims[k] * E.dims[k + 1];
    T.poff[2 * k + 1] = P; P += E.dims[k + 1];
  }
  for (int k = 2 * n; k < 7; ++k) T.poff[k] = P;
  for (int k = 0; k < n; ++k) {
    T.in[k] = PA.in[k]; T.out[k] = PA.out[k]; T.ks[k] = PA.ks[k]; T.kst[k] = PA.kst[k]; T.natural[k] = PA.natural[k];
    T.wf[k] = PA.wf[k]; T.bf[k] = PA.bf[k]; T.wt[k] = PA.wt[k];
  }
  const int grid = (nparams + 1 + 63) / 64;
  hipLaunchKernelGGL(train_tail_kernel, dim3(grid), dim3(TAIL_T), 0, s, T);
  KR_HIP(hipGetLastError());
  return KR_OK;
}

}  // namespace kr
