// mlp_mfma.hpp - the residual MLP of a whole wavefront on the matrix cores.
//
// Inside a shooting sweep every lane owns a different input vector (one per rod x column), so one
// evaluation of the network (reference cosserat_ode.py:90-112) for the wave is a dense contraction
// with a batch of 64 samples: exactly what MFMA is for.  v_mfma_f64_16x16x4_f64 /
// v_mfma_f32_16x16x4_f32 keep the arithmetic type of the sweep (the reference evaluates the MLP in
// fp64 inside simulate, fp32 inside training).
//
// Formulation: out^T[units x samples] = W[units x in] * in^T[in x samples].
//   A operand (lane l):  W[16*To + (l&15)][k]            - weights, pre-packed in fragment order
//   B operand (lane l):  in[sample 16*s + (l&15)][k]      - k = 4*step + (l>>4) in the order below
//   D (lane l, reg r):   out[unit 16*To + row(l,r)][sample 16*s + (l&15)]
//                        row(l,r) = (l>>4) + 4r for f64,  4*(l>>4) + r for f32
// so register r of an accumulator tile IS the B operand of k-step (To, r) of the next layer: the
// whole chain stays in VGPRs, no LDS transpose between layers.  For f64 that k order is the natural
// one; for f32 it is a permutation, which the host bakes into the packed weights of the next layer.
// Only the inputs (each lane's 28-vector) and the 25 outputs cross lanes, through one LDS tile.
//
// Hidden layers are processed in chunks of 64 units (4 tiles x 4 sample tiles = 16 accumulators)
// that are consumed by the next layer immediately, so wide single-hidden-layer networks (the
// reference default is 28 -> 512 -> 25) never materialise their hidden activations.
// Supported: n_layers 2 (in -> H -> 25) or 3 (in -> H1 <= 64 -> H2 -> 25), 28 inputs, one activation
// for all hidden layers; everything
// else uses the per-lane evaluator of mlp_lane.hpp.
#pragma once
#include "kr_internal.hpp"

namespace kr {

template <typename T>
struct MfmaOp;
template <>
struct MfmaOp<double> {
  typedef double acc __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc run(double a, double b, acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
};
template <>
struct MfmaOp<float> {
  typedef float acc __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc run(float a, float b, acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

constexpr int MM_IN = 28;       // MLP input width served by this path
constexpr int MM_KS1 = MM_IN / 4;  // k-steps of the first layer
constexpr int MM_OUT_T = 2;     // output tiles (25 -> 32)
constexpr int MM_TILE_LD = 29;  // row pitch of the LDS exchange tile [64][28 in / 25 out]: odd, so that the per-lane rows
                                // spread over all banks, and small enough for the persistent kernel's LDS budget
constexpr int MM_SH = 2;        // sample tiles processed together (2 x 16 samples): bounds the register footprint

// what the evaluator needs of MlpDev, passed in scalar registers
template <typename T>
struct MfmaNet {
  const T* w[3];
  const T* b[3];
  int ks[3], ot[3], act[3];
  int L;
};
template <typename T>
__device__ __forceinline__ MfmaNet<T> mfma_net(const MlpDev<T>& M) {
  MfmaNet<T> n;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    n.w[k] = M.wfrag[k]; n.b[k] = M.bfrag[k];
    n.ks[k] = M.ksteps[k]; n.ot[k] = M.otiles[k]; n.act[k] = M.acts[k];
  }
  n.L = M.n_layers;
  return n;
}

// One chunk of a hidden layer: 4 unit tiles x MM_SH sample tiles.
template <typename T>
struct HChunk {
  typename MfmaOp<T>::acc a[4][MM_SH];
};

template <typename T>
__device__ __forceinline__ void chunk_bias(HChunk<T>& h, const T* __restrict__ bfrag, int tile0, int lane) {
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    typename MfmaOp<T>::acc b;
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = bfrag[((size_t)(tile0 + o) * 4 + r) * 64 + lane];
#pragma unroll
    for (int s = 0; s < MM_SH; ++s) h.a[o][s] = b;
  }
}
template <typename T, int ACT>
__device__ __forceinline__ void chunk_act(HChunk<T>& h) {
  // 8 values at a time: enough independent chains to cover the fp64 latency, few live temporaries
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    constexpr int NV = MM_SH * 4;
    T v[NV];
#pragma unroll
    for (int s = 0; s < MM_SH; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[s * 4 + r] = h.a[o][s][r];
    activate_block<T, ACT, NV>(v);
#pragma unroll
    for (int s = 0; s < MM_SH; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) h.a[o][s][r] = v[s * 4 + r];
  }
}
constexpr int MM_PD = 3;  // weight fragments are requested this many k-steps ahead of their MFMAs

// Generic accumulation  dst[o][s] += W[tile0+o][ks] * bsrc(s, ks)  over KS k-steps with NO output tiles,
// software-pipelined: the A fragments (weights, global memory / L2) of k-step ks+MM_PD are loaded
// before the MFMAs of k-step ks are issued, so the matrix pipe never waits for an L2 round trip.
template <typename T, int NO, int KS, typename BFn>
__device__ __forceinline__ void mfma_accumulate(typename MfmaOp<T>::acc (&dst)[NO][MM_SH], const T* __restrict__ w,
                                                int ksteps, int tile0, int ks0, int lane, BFn bsrc) {
  T a[KS][NO];
#pragma unroll
  for (int ks = 0; ks < (MM_PD < KS ? MM_PD : KS); ++ks)
#pragma unroll
    for (int o = 0; o < NO; ++o) a[ks][o] = w[((size_t)(tile0 + o) * ksteps + ks0 + ks) * 64 + lane];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + MM_PD < KS) {
#pragma unroll
      for (int o = 0; o < NO; ++o) a[ks + MM_PD][o] = w[((size_t)(tile0 + o) * ksteps + ks0 + ks + MM_PD) * 64 + lane];
    }
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int s = 0; s < MM_SH; ++s) dst[o][s] = MfmaOp<T>::run(a[ks][o], bsrc(s, ks), dst[o][s]);
  }
}

// first layer: chunk (tiles tile0..tile0+3) += W0 * inputs
template <typename T>
__device__ __forceinline__ void chunk_from_inputs(HChunk<T>& h, const T* __restrict__ w0, int tile0,
                                                  const T (&bin)[MM_SH][MM_KS1], int lane) {
  mfma_accumulate<T, 4, MM_KS1>(h.a, w0, MM_KS1, tile0, 0, lane, [&](int s, int ks) { return bin[s][ks]; });
}
// dst chunk (tiles tile0.. of a layer with `ksteps` k-steps per tile) += W * src chunk, whose 16 k-steps
// start at k-step ks0 of that layer's input; register r of src tile o1 is k-step 4*o1 + r
template <typename T>
__device__ __forceinline__ void chunk_from_chunk(HChunk<T>& dst, const T* __restrict__ wfrag, int ksteps, int tile0,
                                                 const HChunk<T>& src, int ks0, int lane) {
  mfma_accumulate<T, 4, 16>(dst.a, wfrag, ksteps, tile0, ks0, lane,
                            [&](int s, int ks) { return src.a[ks >> 2][s][ks & 3]; });
}
// output layer accumulators += Wout * src chunk (k-steps ks0..ks0+15)
template <typename T>
__device__ __forceinline__ void out_from_chunk(typename MfmaOp<T>::acc (&oacc)[MM_OUT_T][MM_SH], const T* __restrict__ wo,
                                               int ks_out, const HChunk<T>& src, int ks0, int lane) {
  mfma_accumulate<T, MM_OUT_T, 16>(oacc, wo, ks_out, 0, ks0, lane,
                                   [&](int s, int ks) { return src.a[ks >> 2][s][ks & 3]; });
}

// ---- fp32 only: activations scheduled under the MFMAs that consume them ---------------------------------
// The source chunk still holds pre-activations; unit tile o is activated right before the four k-steps that
// read it, so that the activation of tile o+1 (vector ALU) can run under the MFMAs of tile o (matrix pipe)
// instead of with the matrix pipe idle: +10 % on the MLP-on simulation.  (The same interleaving in fp64 needs
// more live registers than a wave has and spills - 2.9 -> 5.4 ms per step - so fp64 keeps the block form; the
// fp64 code path is deliberately left textually unchanged, its allocation is at the limit.)
template <typename T, int ACT>
__device__ __forceinline__ void tile_act(HChunk<T>& h, int o) {
  constexpr int NV = MM_SH * 4;
  T v[NV];
#pragma unroll
  for (int s = 0; s < MM_SH; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[s * 4 + r] = h.a[o][s][r];
  activate_block<T, ACT, NV>(v);
#pragma unroll
  for (int s = 0; s < MM_SH; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) h.a[o][s][r] = v[s * 4 + r];
}
template <typename T, int NO, int ACT>
__device__ __forceinline__ void accumulate_lazy(typename MfmaOp<T>::acc (&dst)[NO][MM_SH], const T* __restrict__ w,
                                                int ksteps, int tile0, int ks0, int lane, HChunk<T>& src) {
  constexpr int KS = 16;
  T a[KS][NO];
#pragma unroll
  for (int ks = 0; ks < MM_PD; ++ks)
#pragma unroll
    for (int o = 0; o < NO; ++o) a[ks][o] = w[((size_t)(tile0 + o) * ksteps + ks0 + ks) * 64 + lane];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + MM_PD < KS) {
#pragma unroll
      for (int o = 0; o < NO; ++o) a[ks + MM_PD][o] = w[((size_t)(tile0 + o) * ksteps + ks0 + ks + MM_PD) * 64 + lane];
    }
    if ((ks & 3) == 0) tile_act<T, ACT>(src, ks >> 2);
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int s = 0; s < MM_SH; ++s) dst[o][s] = MfmaOp<T>::run(a[ks][o], src.a[ks >> 2][s][ks & 3], dst[o][s]);
  }
}

__device__ __forceinline__ void mm_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// Evaluates the network for the 64 samples of the wave, in place on the LDS tile [64][MM_TILE_LD]:
// on entry row b holds the 28 inputs of lane b, on exit its 25 outputs.  One copy per precision in
// the whole library (not inlined into the sweep kernels).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ const T* uni(const T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const T*>(((unsigned long long)hi << 32) | lo);
}

// ACT: activation of every hidden layer (compile time: one activation's code per instantiation).
// Arguments of a non-inlined device function arrive in VGPRs; everything that steers control flow
// or addresses weights is wave-uniform and is moved back to scalar registers first.
template <typename T, int ACT>
__device__ __attribute__((noinline)) void mlp_mfma_tile(MfmaNet<T> netv, T* tile, int lane) {
  using Acc = typename MfmaOp<T>::acc;
  MfmaNet<T> net;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    net.w[k] = uni(netv.w[k]); net.b[k] = uni(netv.b[k]);
    net.ks[k] = uni(netv.ks[k]); net.ot[k] = uni(netv.ot[k]); net.act[k] = ACT;
  }
  net.L = uni(netv.L);
  {
    const unsigned long long tv = reinterpret_cast<unsigned long long>(tile);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)tv);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(tv >> 32));
    tile = reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
  }
  const int L = net.L;
  const T* wo = net.w[L == 2 ? 1 : 2];
  const T* bo = net.b[L == 2 ? 1 : 2];
  const int ks_out = net.ks[L == 2 ? 1 : 2];
#pragma unroll 1
  for (int sh = 0; sh < 4 / MM_SH; ++sh) {
    // B operands of the first layer for these sample tiles
    T bin[MM_SH][MM_KS1];
#pragma unroll
    for (int s = 0; s < MM_SH; ++s)
#pragma unroll
      for (int ks = 0; ks < MM_KS1; ++ks)
        bin[s][ks] = tile[(16 * (sh * MM_SH + s) + (lane & 15)) * MM_TILE_LD + 4 * ks + (lane >> 4)];
    Acc oacc[MM_OUT_T][MM_SH];
#pragma unroll
    for (int o2 = 0; o2 < MM_OUT_T; ++o2) {
      Acc b;
#pragma unroll
      for (int r = 0; r < 4; ++r) b[r] = bo[((size_t)o2 * 4 + r) * 64 + lane];
#pragma unroll
      for (int s = 0; s < MM_SH; ++s) oacc[o2][s] = b;
    }
    if (L == 2) {
      // in -> H -> 25: hidden chunks are consumed by the output layer as soon as they exist
      const int chunks = net.ot[0] / 4;
#pragma unroll 1
      for (int c = 0; c < chunks; ++c) {
        HChunk<T> h;
        chunk_bias<T>(h, net.b[0], 4 * c, lane);
        chunk_from_inputs<T>(h, net.w[0], 4 * c, bin, lane);
        if constexpr (sizeof(T) == 4) {
          accumulate_lazy<T, MM_OUT_T, ACT>(oacc, wo, ks_out, 0, 16 * c, lane, h);
        } else {
          chunk_act<T, ACT>(h);
          out_from_chunk<T>(oacc, wo, ks_out, h, 16 * c, lane);
        }
      }
    } else {
      // in -> H1 (one chunk) -> H2 -> 25
      HChunk<T> h1;
      chunk_bias<T>(h1, net.b[0], 0, lane);
      chunk_from_inputs<T>(h1, net.w[0], 0, bin, lane);
      const int chunks2 = net.ot[1] / 4;
      if constexpr (sizeof(T) == 4) {
        {  // first chunk of the second layer activates h1 tile by tile; later chunks find it activated
          HChunk<T> h2;
          chunk_bias<T>(h2, net.b[1], 0, lane);
          accumulate_lazy<T, 4, ACT>(h2.a, net.w[1], net.ks[1], 0, 0, lane, h1);
          accumulate_lazy<T, MM_OUT_T, ACT>(oacc, wo, ks_out, 0, 0, lane, h2);
        }
#pragma unroll 1
        for (int c = 1; c < chunks2; ++c) {
          HChunk<T> h2;
          chunk_bias<T>(h2, net.b[1], 4 * c, lane);
          chunk_from_chunk<T>(h2, net.w[1], net.ks[1], 4 * c, h1, 0, lane);
          accumulate_lazy<T, MM_OUT_T, ACT>(oacc, wo, ks_out, 0, 16 * c, lane, h2);
        }
      } else {
        chunk_act<T, ACT>(h1);
#pragma unroll 1
        for (int c = 0; c < chunks2; ++c) {
          HChunk<T> h2;
          chunk_bias<T>(h2, net.b[1], 4 * c, lane);
          chunk_from_chunk<T>(h2, net.w[1], net.ks[1], 4 * c, h1, 0, lane);
          chunk_act<T, ACT>(h2);
          out_from_chunk<T>(oacc, wo, ks_out, h2, 16 * c, lane);
        }
      }
    }
    // D layout -> tile[sample][unit]; these rows' inputs are already in registers
    mm_wave_sync();
#pragma unroll
    for (int o2 = 0; o2 < MM_OUT_T; ++o2)
#pragma unroll
      for (int s = 0; s < MM_SH; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = sizeof(T) == 8 ? (lane >> 4) + 4 * r : 4 * (lane >> 4) + r;
          // (units 25..31 of the padded output tiles have no column in the tile)
          if (16 * o2 + row < 25) tile[(16 * (sh * MM_SH + s) + (lane & 15)) * MM_TILE_LD + 16 * o2 + row] = oacc[o2][s][r];
        }
  }
  mm_wave_sync();
}

// per-lane wrapper: x in, 25 outputs back
template <typename T>
__device__ __forceinline__ void mlp_mfma_eval(const MlpDev<T>& M, const T (&x)[MM_IN], T* tile, int lane,
                                              T (&out)[25]) {
#pragma unroll
  for (int c = 0; c < MM_IN; ++c) tile[lane * MM_TILE_LD + c] = x[c];
  mm_wave_sync();
  const MfmaNet<T> net = mfma_net<T>(M);
  switch (M.acts[0]) {  // wave-uniform
    case KR_ACT_TANH: mlp_mfma_tile<T, KR_ACT_TANH>(net, tile, lane); break;
    case KR_ACT_SOFTPLUS: mlp_mfma_tile<T, KR_ACT_SOFTPLUS>(net, tile, lane); break;
    case KR_ACT_RELU: mlp_mfma_tile<T, KR_ACT_RELU>(net, tile, lane); break;
    case KR_ACT_ELU: mlp_mfma_tile<T, KR_ACT_ELU>(net, tile, lane); break;
    default: mlp_mfma_tile<T, KR_ACT_NONE>(net, tile, lane); break;
  }
#pragma unroll
  for (int c = 0; c < 25; ++c) out[c] = tile[lane * MM_TILE_LD + c];
  mm_wave_sync();
}

}  // namespace kr
