// mlp_jvp.hpp - the residual MLP inside a multiple-shooting sweep: one exact evaluation per sub-interval,
// Jacobian-vector products for the forward-difference columns.
//
// In a sweep of the multiple-shooting kernels (kr_ms_impl.hpp) the 58 active lanes of a rod's wavefront are
// 4 unperturbed trajectories (one per sub-interval) and 54 trajectories that differ from "their" unperturbed one
// by a forward-difference perturbation of relative size 1e-7 (fp64) / 1e-3 (fp32).  Evaluating the network
// (reference cosserat_ode.py:90-112, 169-184) in full precision for all 64 lanes - what mlp_mfma.hpp does and what
// single shooting still uses - spends 93 % of the matrix-core time on inputs that are equal to 7 digits.  Here:
//
//   * BASE: the 4 unperturbed inputs go through the network in the arithmetic type of the sweep (fp64 in
//     `simulate`, like the reference) on v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, as ONE tile of 16 sample
//     columns (4 used).  These outputs are what the stored trajectory and the Newton residual are made of.
//   * JVP: for a perturbed lane  NN(x_b + dx) = NN(x_b) + J(x_b) dx + O(dx^2)  with  J dx = W_L diag(act'_{L-1}) ...
//     diag(act'_1) W_1 dx  evaluated on v_mfma_f32_16x16x32_bf16 (bf16 operands, fp32 accumulation): 16 perturbed
//     lanes of sub-interval i are exactly one 16-column sample tile, whose act' factors - those of the interval's
//     base point - depend on the accumulator ROW only, so they are applied in registers.  The truncation O(dx^2) is
//     what the forward difference itself commits; the bf16 rounding (2^-9 relative) only touches the Jacobian of
//     the Newton iteration, never the residual, so the root is unchanged and the contraction stays ~1e-3 per sweep
//     better than any stopping tolerance needs.  Newton also no longer differentiates across activation kinks:
//     act' is the base point's one-sided derivative (FD through an ELU/ReLU kink mixes both sides).
//
// Register chaining: a D tile of 16x16x32 (lane l: column l&15, rows 4(l>>4)+r) holds, for two adjacent unit tiles,
// the eight k-values a lane needs as B operand of the next layer (k-slot (q, j) of k-step s = unit 32s + 4q + j for
// j < 4, 32s + 16 + 4q + j - 4 otherwise); the host packs the next layer's bf16 weights in that k order, so no LDS
// transpose is needed between layers.  The base chain keeps the layouts of mlp_mfma.hpp.
#pragma once
#include <type_traits>
#include "mlp_mfma.hpp"

namespace kr {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <typename T>
struct MjVec;  // 16-byte vector of T
template <>
struct MjVec<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int n = 4; };
template <>
struct MjVec<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int n = 2; };

constexpr int MJ_XB_LD = 28;                 // pitch of the base input / output rows [4][28] (T): 16-byte aligned rows
constexpr int MJ_ACT_SLOTS = 4;              // act' tables the scratch holds (one per 64-unit hidden chunk of a 3-layer net)
constexpr int MJ_ACTP_BYTES = MJ_ACT_SLOTS * 4 * 64 * 4;  // each [4 intervals][64 units] f32
constexpr int MJ_DX_LD = 32;                 // bf16 per perturbed sample row
constexpr int MJ_DOUT_LD = 28;               // f32 per output row: 16-byte aligned rows, 25 used
template <typename T>
__host__ __device__ constexpr size_t mj_xb_bytes() { return (size_t)4 * MJ_XB_LD * sizeof(T); }  // 896 / 448: multiples of 16
template <typename T>
__host__ __device__ constexpr size_t mj_scratch_bytes() {
  return mj_xb_bytes<T>() + MJ_ACTP_BYTES + (size_t)64 * MJ_DOUT_LD * 4;  // (dx [64][32] bf16 aliases dout)
}
template <typename T>
__host__ __device__ constexpr size_t mj_scratch_elems() { return (mj_scratch_bytes<T>() + sizeof(T) - 1) / sizeof(T); }

__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {  // v_cvt_pk_bf16_f32: round to nearest even
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}

// d act / d pre from the ACTIVATED value a (what the base chain holds in its accumulators)
template <int ACT>
__device__ __forceinline__ float act_grad_from_value(float a) {
  if constexpr (ACT == KR_ACT_ELU) return a > 0.f ? 1.f : a + 1.f;          // e^x = elu(x) + 1 for x <= 0
  else if constexpr (ACT == KR_ACT_TANH) return 1.f - a * a;
  else if constexpr (ACT == KR_ACT_SOFTPLUS) return 1.f - __expf(-a);       // sigmoid(x) = 1 - e^{-softplus(x)}
  else if constexpr (ACT == KR_ACT_RELU) return a > 0.f ? 1.f : 0.f;
  else return 1.f;
}

// what the evaluator needs of MlpDev (scalar registers)
struct JvpNet {
  const float* wq[3];       // base chain: A fragments of v_mfma_f64_4x4x4_4b, fp32 (the weights ARE fp32 numbers), 4 k-steps per
  const float* bq[3];       //   16-byte element: [tile][k-group][lane][4]; biases [tile][lane] in the D layout
  const bf16x8* j[3];       // JVP chain: bf16 A fragments [tile][k-step of 32][lane]
  int kg[3], ot[3], jks[3]; // k-groups (of 16 inputs) per tile, 16-unit tiles, JVP k-steps
  int L;
  int ptab;                 // columns 6..14 of sample tile 0 are p columns of the intervals 1..3 (jvp_scale_pack)
};
template <typename T>
__device__ __forceinline__ JvpNet jvp_net(const MlpDev<T>& M) {
  JvpNet n;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    n.wq[k] = M.wq[k]; n.bq[k] = M.bq[k]; n.j[k] = reinterpret_cast<const bf16x8*>(M.jfrag[k]);
    n.kg[k] = M.kgroups[k]; n.ot[k] = M.otiles[k]; n.jks[k] = M.jksteps[k];
  }
  n.L = M.n_layers;
  n.ptab = 0;
  return n;
}

#ifdef MJ_STAMPS  // tools/ubench_jvp.hip: cycles per phase of one evaluation (wave 0 of block 0)
__device__ unsigned long long mj_stamp_acc[16];
#define MJ_STAMP(k)                                                                       \
  do {                                                                                    \
    __builtin_amdgcn_s_waitcnt(0xC07F); /* (vmcnt / lgkmcnt 0: charge a phase its own waits) */ \
    const unsigned long long mj_now = __builtin_amdgcn_s_memtime();                       \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                   \
    if (blockIdx.x == 0 && threadIdx.x == 0) mj_stamp_acc[k] += mj_now - mj_t;            \
    mj_t = mj_now;                                                                        \
  } while (0)
#define MJ_STAMP_BEGIN() unsigned long long mj_t = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
#else
#define MJ_STAMP(k) do {} while (0)
#define MJ_STAMP_BEGIN() do {} while (0)
#endif
#ifdef MJ_STAMPS
#define MJ_T3(k) do { __builtin_amdgcn_sched_barrier(0); mj3[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MJ_T3(k) do {} while (0)
#endif
#ifdef MJ_NOCVT  // timing experiment only: no f32 -> f64 conversion of the weight fragments (garbage values)
typedef float mj_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double mj_nocvt(float a, float b) { mj_f2 v = {a, b}; return __builtin_bit_cast(double, v); }
#define MJ_W(v, e) mj_nocvt((v)[(e) & 2], (v)[((e) & 2) + 1])
#else
#define MJ_W(v, e) ((double)(v)[e])
#endif
#ifndef MJ_VARIANT
#define MJ_VARIANT 0
#endif
#define MJ_LDS __attribute__((address_space(3)))
#define MJ_GLB __attribute__((address_space(1)))
// Every LDS hand-off of the evaluator is between lanes of ONE wavefront (its scratch is the wavefront's own), and the LDS
// unit serves a wavefront's instructions in order: a wavefront-scope fence (a compiler barrier) is all it takes.  The
// workgroup-scope fence of mm_wave_sync also waits for every global load in flight (vmcnt(0)) - here the weight fragments
// requested one layer ahead, i.e. a full L2 round trip at each of the evaluator's six synchronisation points.
__device__ __forceinline__ void mj_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
typedef MJ_GLB const float* gfp;
typedef MJ_GLB const f32x4* gf4p;
typedef MJ_GLB const bf16x8* gbfp;

// v_mfma_f64_4x4x4_4b_f64 (measured layout, tools/ubench_mfma2.hip): 4 independent 4x4x4 products; lane l holds
//   A_b[i][k]:  k = l >> 4, b = (l >> 2) & 3, i = l & 3        B_b[k][j]:  k = l >> 4, b = (l >> 2) & 3, j = l & 3
//   D_b[i][j]:  i = l >> 4, b = (l >> 2) & 3, j = l & 3
// Used as out^T[16 units x 4 samples] += W[16 x 4] in^T[4 x 4]: block b carries units 4 b .. 4 b + 3 of the tile,
// every block the same inputs; 17 cycles per instruction against 64 for the 16x16x4 form, whose other 12 sample
// columns would be padding here.  The result (unit 16 t + 4 b + i on lane (i, b, j)) becomes the next layer's B
// operand (input 4 s + k on lane (k, *, j)) by a ds_swizzle that copies block s & 3 of accumulator s >> 2 to all blocks.
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
template <int BSRC>
__device__ __forceinline__ double bcast_block(double v) {
  constexpr int pat = 0x13 | ((BSRC << 2) << 5);  // bit-mask mode: lane' = (lane & 0b10011) | (BSRC << 2)
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_ds_swizzle((int)(bits & 0xFFFFFFFFll), pat);
  const int hi = __builtin_amdgcn_ds_swizzle((int)(bits >> 32), pat);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// acc[o] += W(tile0 + o, k-steps 4 kg0 .. 4 kg0 + 4 NKG) * b(ks) for NO tiles; w4: fragments requested earlier
template <int NO, int NKG, typename BFn>
__device__ __forceinline__ void base_run(double (&acc)[NO], const f32x4 (&w4)[NKG][NO], BFn bsrc) {
#pragma unroll
  for (int g = 0; g < NKG; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double bv = bsrc(4 * g + e);
#pragma unroll
      for (int o = 0; o < NO; ++o) acc[o] = mfma4((double)w4[g][o][e], bv, acc[o]);
    }
}
template <int NO, int NKG>
__device__ __forceinline__ void base_load(f32x4 (&w4)[NKG][NO], gf4p w, int kgroups, int tile0, int kg0, int lane) {
#pragma unroll
  for (int g = 0; g < NKG; ++g)
#pragma unroll
    for (int o = 0; o < NO; ++o) w4[g][o] = w[((size_t)(tile0 + o) * kgroups + kg0 + g) * 64 + lane];
}
// B operand of k-step ks (0..15) from the four accumulators of a 64-unit chunk
__device__ __forceinline__ double chunk_operand(const double (&h)[4], int ks) {
  const double v = h[ks >> 2];
  switch (ks & 3) {
    case 0: return bcast_block<0>(v);
    case 1: return bcast_block<1>(v);
    case 2: return bcast_block<2>(v);
    default: return bcast_block<3>(v);
  }
}
// the same for all 16 k-steps at once: 32 swizzles issued back to back, one wait (pays where registers allow)
__device__ __forceinline__ void chunk_operands(const double (&h)[4], double (&bv)[16]) {
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    bv[4 * o + 0] = bcast_block<0>(h[o]);
    bv[4 * o + 1] = bcast_block<1>(h[o]);
    bv[4 * o + 2] = bcast_block<2>(h[o]);
    bv[4 * o + 3] = bcast_block<3>(h[o]);
  }
}
// activation of a chunk (4 values per lane, all of them real) and its act' to the table [interval j][unit]
template <int ACT, bool TABLE = true>
__device__ __forceinline__ void chunk_activate(double (&h)[4], MJ_LDS float* actp, int lane) {
  // (fp64 also for fp32 sweeps: converting to float for a v_exp_f32 activation was measured SLOWER, 16.8 k against
  //  14.9 k cycles per evaluation in tools/ubench_jvp.hip)
  activate_block<double, ACT, 4>(h);
  if constexpr (TABLE) {
    const int unit0 = 4 * ((lane >> 2) & 3) + (lane >> 4);
#pragma unroll
    for (int o = 0; o < 4; ++o) actp[(lane & 3) * 64 + 16 * o + unit0] = act_grad_from_value<ACT>((float)h[o]);
  }
}

// JVP: dst[o][s] += A(o, ks) * b[s][ks]  over KS k-steps of 32, NO unit tiles, 4 sample tiles
template <int NO, int KS>
__device__ __forceinline__ void jvp_load(bf16x8 (&a)[KS][NO], gbfp w, int ksteps, int tile0, int ks0, int lane) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int o = 0; o < NO; ++o) a[ks][o] = w[((size_t)(tile0 + o) * ksteps + ks0 + ks) * 64 + lane];
}
template <int NO, int KS>
__device__ __forceinline__ void jvp_accumulate(f32x4 (&dst)[NO][4], const bf16x8 (&a)[KS][NO], const bf16x8 (&b)[4][KS]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int s = 0; s < 4; ++s) dst[o][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks][o], b[s][ks], dst[o][s], 0, 0, 0);
}
// D tiles of a 64-unit hidden chunk (4 unit tiles x 4 sample tiles) -> scaled by act' of the tile's interval ->
// bf16 B operands of the next layer (2 k-steps of 32 units)
// Sample tile 0 holds the 6 columns of interval 0; its columns 6..14 may carry three extra samples for each of the
// intervals 1..3 (the p columns of kr_ms_impl.hpp: perturbations of an interval's start position, which only the network
// reads) - a lane's column is fixed, so it simply takes the act' table of the interval its column belongs to.
__device__ __forceinline__ void jvp_scale_pack(const f32x4 (&dh)[4][4], const MJ_LDS float* actp, int lane, bf16x8 (&b)[4][2],
                                               int ptab) {
  const int q = lane >> 4;
  const int c0 = lane & 15;
  // ptab 1: the one-wavefront layout and the first wavefront of a rod (tile 0: interval 0 with 6 columns, then p columns of
  // the intervals 1..3 in columns 6..14); ptab 2: the other wavefronts of a rod (three intervals with 16 columns each in the
  // tiles 0..2, their p columns in the columns 0..8 of tile 3)
  const int tbl0 = (ptab == 1 && c0 >= 6 && c0 < 15) ? 1 + (c0 - 6) / 3 : 0;
  const int tbl3 = (ptab == 2 && c0 < 9) ? c0 / 3 : 3;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    f32x4 g[4];
    const int tbl = s == 0 ? tbl0 : s == 3 ? tbl3 : s;
#pragma unroll
    for (int o = 0; o < 4; ++o) g[o] = *reinterpret_cast<const MJ_LDS f32x4*>(actp + tbl * 64 + 16 * o + 4 * q);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      u32x4 p;
      const f32x4 lo = dh[2 * ks][s] * g[2 * ks], hi = dh[2 * ks + 1][s] * g[2 * ks + 1];
      p[0] = pack_bf2(lo[0], lo[1]); p[1] = pack_bf2(lo[2], lo[3]);
      p[2] = pack_bf2(hi[0], hi[1]); p[3] = pack_bf2(hi[2], hi[3]);
      b[s][ks] = __builtin_bit_cast(bf16x8, p);
    }
  }
}

__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// Base + JVP evaluation for one wavefront on the LDS scratch:
//   xb   [4][MJ_XB_LD] T       in: row i = input of the base lane of interval i;  out: its 25 outputs
//   actp [slots][4][64] f32    scratch: act' of the hidden chunks at the 4 base points
//   dx   [64][32] bf16         in: row 16 i + c = dx of perturbed column c of interval i (unused rows zero)
//   dout [64][MJ_DOUT_LD] f32  out: J dx of the same rows (aliases dx)
// The base chain runs in fp64 for both sweep precisions (it is the faster matrix-core form for 4 samples), then the
// JVP chain; pointers arrive as generic ones (the function is not inlined: one copy per activation in the library)
// and are put back into their address spaces first - flat loads would serialise LDS and global traffic.
// VAR: a second copy of the function for the kernels built for two wavefronts per SIMD (the register limit of a kernel
// reaches its callees only if all callers of a function agree on it)
// The network descriptor arrives as SCALAR ARGUMENTS (29 dwords: the AMDGPU calling convention passes the first 32 in
// VGPRs): as one by-value struct it went through private memory - 26 scratch stores in the caller, as many scratch loads
// and a full memory round trip at the head of every evaluation.
// not_tail_called: a call site the optimiser marks `tail` (possible now that no argument lives in the caller's frame)
// switches off LLVM's "no callee-saved registers" treatment of this internal function, and the callee then saves and
// restores ~170 VGPRs through scratch memory on every evaluation.
template <typename T, int ACT, int VAR = 0>
__device__ __attribute__((noinline, not_tail_called)) void mlp_jvp_tile(const float* wq0, const float* wq1, const float* wq2, const float* bq0,
                                                       const float* bq1, const float* bq2, const bf16x8* j0, const bf16x8* j1,
                                                       const bf16x8* j2, int kg1, int kg2, int ot0, int ot1, int jks1, int jks2,
                                                       int Lv, int ptabv, T* scratch_generic, int lane) {
  gf4p wq[3] = {(gf4p)uni64((unsigned long long)wq0), (gf4p)uni64((unsigned long long)wq1), (gf4p)uni64((unsigned long long)wq2)};
  gfp bq[3] = {(gfp)uni64((unsigned long long)bq0), (gfp)uni64((unsigned long long)bq1), (gfp)uni64((unsigned long long)bq2)};
  gbfp jq[3] = {(gbfp)uni64((unsigned long long)j0), (gbfp)uni64((unsigned long long)j1), (gbfp)uni64((unsigned long long)j2)};
  const int kgs[3] = {2, uni(kg1), uni(kg2)};
  const int ots[3] = {uni(ot0), uni(ot1), MM_OUT_T};
  const int jkss[3] = {1, uni(jks1), uni(jks2)};
  const int L = uni(Lv);
  const int ptab = uni(ptabv);
  MJ_LDS unsigned char* sbase = (MJ_LDS unsigned char*)(unsigned)__builtin_amdgcn_readfirstlane(
      (int)(unsigned long long)(MJ_LDS unsigned char*)scratch_generic);
  MJ_LDS T* xb = (MJ_LDS T*)sbase;
  MJ_LDS float* actp = (MJ_LDS float*)(sbase + mj_xb_bytes<T>());
  MJ_LDS unsigned char* dreg = sbase + mj_xb_bytes<T>() + MJ_ACTP_BYTES;
  const MJ_LDS bf16x8* dx = (const MJ_LDS bf16x8*)dreg;
  MJ_LDS float* dout = (MJ_LDS float*)dreg;
  const int lo_ = L == 2 ? 1 : 2;
  const int q = lane >> 4, c = lane & 15, j4 = lane & 3;
  const int dunit = 4 * ((lane >> 2) & 3) + (lane >> 4);  // unit of this lane inside a D tile

  double obase[MM_OUT_T][2];  // two partial sums per output tile (4 independent accumulators for the matrix pipe)
#pragma unroll
  for (int o2 = 0; o2 < MM_OUT_T; ++o2) { obase[o2][0] = (double)bq[lo_][o2 * 64 + lane]; obase[o2][1] = 0.0; }
  f32x4 ojvp[MM_OUT_T][4];
#pragma unroll
  for (int o2 = 0; o2 < MM_OUT_T; ++o2)
#pragma unroll
    for (int s = 0; s < 4; ++s) ojvp[o2][s] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto zero_dh = [](f32x4 (&dh)[4][4]) {
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int s = 0; s < 4; ++s) dh[o][s] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto out_layer = [&](const double (&h)[4], const f32x4 (&wo4)[4][MM_OUT_T], auto batched) {
    // output tiles from a 64-unit chunk; even / odd k-groups into separate partial sums
    double a4[4] = {obase[0][0], obase[1][0], obase[0][1], obase[1][1]};
    double bvs[16];
    if constexpr (decltype(batched)::value) chunk_operands(h, bvs);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double bv;
        if constexpr (decltype(batched)::value) bv = bvs[4 * g + e];
        else bv = chunk_operand(h, 4 * g + e);
#pragma unroll
        for (int o2 = 0; o2 < MM_OUT_T; ++o2) a4[2 * (g & 1) + o2] = mfma4((double)wo4[g][o2][e], bv, a4[2 * (g & 1) + o2]);
      }
    obase[0][0] = a4[0]; obase[1][0] = a4[1]; obase[0][1] = a4[2]; obase[1][1] = a4[3];
  };

  // base inputs as B operand: input 4 ks + (l >> 4) of sample l & 3 (the same in every block); k-step 7 is padding
  double bin[8];
#pragma unroll
  for (int ks = 0; ks < MM_KS1; ++ks) bin[ks] = (double)xb[j4 * MJ_XB_LD + 4 * ks + q];
  bin[7] = 0.0;

  if (L == 2) {
    // in -> H -> 25, hidden layer in chunks of 64 units that the output layer consumes at once
    bf16x8 bdx[4][1];
#pragma unroll
    for (int s = 0; s < 4; ++s) bdx[s][0] = dx[(16 * s + c) * (MJ_DX_LD / 8) + q];
    const int chunks = ots[0] / 4;
#pragma unroll 1
    for (int ch = 0; ch < chunks; ++ch) {
      f32x4 w1[2][4], wo4[4][MM_OUT_T];
      base_load<4, 2>(w1, wq[0], 2, 4 * ch, 0, lane);
      base_load<MM_OUT_T, 4>(wo4, wq[1], kgs[1], 0, 4 * ch, lane);
      bf16x8 a1[1][4], ao[2][MM_OUT_T];
      jvp_load<4, 1>(a1, jq[0], 1, 4 * ch, 0, lane);
      jvp_load<MM_OUT_T, 2>(ao, jq[1], jkss[1], 0, 2 * ch, lane);
      double h[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) h[o] = (double)bq[0][(4 * ch + o) * 64 + lane];
      base_run<4, 2>(h, w1, [&](int ks) { return bin[ks]; });
      chunk_activate<ACT>(h, actp, lane);
      out_layer(h, wo4, std::true_type{});
      f32x4 dh[4][4];
      zero_dh(dh);
      jvp_accumulate<4, 1>(dh, a1, bdx);
      mj_wave_sync();
      bf16x8 b1[4][2];
      jvp_scale_pack(dh, actp, lane, b1, ptab);
      mj_wave_sync();  // (actp is rewritten by the next chunk)
      jvp_accumulate<MM_OUT_T, 2>(ojvp, ao, b1);
    }
  } else {
    // in -> H1 (<= 64: one chunk) -> H2 (<= 64 (MJ_ACT_SLOTS - 1)) -> 25: the whole base chain, then the whole JVP chain
    const int chunks2 = ots[1] / 4;
    MJ_STAMP_BEGIN();
    f32x4 w1[2][4], w2[4][4], wo4[4][MM_OUT_T];
    base_load<4, 2>(w1, wq[0], 2, 0, 0, lane);
    base_load<4, 4>(w2, wq[1], 4, 0, 0, lane);
    base_load<MM_OUT_T, 4>(wo4, wq[2], kgs[2], 0, 0, lane);
    bf16x8 a1[1][4], a2[2][4], ao[2][MM_OUT_T];
    jvp_load<4, 1>(a1, jq[0], 1, 0, 0, lane);
    jvp_load<4, 2>(a2, jq[1], 2, 0, 0, lane);
    jvp_load<MM_OUT_T, 2>(ao, jq[2], jkss[2], 0, 0, lane);
    MJ_STAMP(0);  // weight loads (waited for)
    {
      double h1[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) h1[o] = (double)bq[0][o * 64 + lane];
      base_run<4, 2>(h1, w1, [&](int ks) { return bin[ks]; });
      MJ_STAMP(1);  // base layer 1
      chunk_activate<ACT>(h1, actp, lane);
      MJ_STAMP(2);  // activation 1
#pragma unroll 1
      for (int ch = 0; ch < chunks2; ++ch) {
        if (ch > 0) {
          base_load<4, 4>(w2, wq[1], 4, 4 * ch, 0, lane);
          base_load<MM_OUT_T, 4>(wo4, wq[2], kgs[2], 0, 4 * ch, lane);
        }
        double h2[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) h2[o] = (double)bq[1][(4 * ch + o) * 64 + lane];
        base_run<4, 4>(h2, w2, [&](int ks) { return chunk_operand(h1, ks); });
        MJ_STAMP(3);  // base layer 2
        chunk_activate<ACT>(h2, actp + (1 + ch) * 256, lane);
        MJ_STAMP(4);  // activation 2
        out_layer(h2, wo4, std::false_type{});
        MJ_STAMP(5);  // base output layer
      }
    }
    mj_wave_sync();
    bf16x8 b1[4][2];
    {
      bf16x8 bdx[4][1];
#pragma unroll
      for (int s = 0; s < 4; ++s) bdx[s][0] = dx[(16 * s + c) * (MJ_DX_LD / 8) + q];
      f32x4 dh[4][4];
      zero_dh(dh);
      jvp_accumulate<4, 1>(dh, a1, bdx);
      MJ_STAMP(6);  // JVP layer 1
      jvp_scale_pack(dh, actp, lane, b1, ptab);
      MJ_STAMP(7);  // scale + pack 1
    }
#pragma unroll 1
    for (int ch = 0; ch < chunks2; ++ch) {
      if (ch > 0) {
        jvp_load<4, 2>(a2, jq[1], 2, 4 * ch, 0, lane);
        jvp_load<MM_OUT_T, 2>(ao, jq[2], jkss[2], 0, 2 * ch, lane);
      }
      f32x4 dh[4][4];
      zero_dh(dh);
      jvp_accumulate<4, 2>(dh, a2, b1);
      MJ_STAMP(8);  // JVP layer 2
      bf16x8 b2[4][2];
      jvp_scale_pack(dh, actp + (1 + ch) * 256, lane, b2, ptab);
      MJ_STAMP(9);  // scale + pack 2
      jvp_accumulate<MM_OUT_T, 2>(ojvp, ao, b2);
      MJ_STAMP(10);  // JVP output layer
    }
  }
  // results: base outputs over the input rows, J dx rows over the dx rows (every lane has read its operands; the
  // sync orders the other lanes' reads before these writes)
  mj_wave_sync();
#pragma unroll
  for (int o2 = 0; o2 < MM_OUT_T; ++o2)
    if (16 * o2 + dunit < 25) xb[j4 * MJ_XB_LD + 16 * o2 + dunit] = (T)(obase[o2][0] + obase[o2][1]);
#pragma unroll
  for (int o2 = 0; o2 < MM_OUT_T; ++o2)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (16 * o2 + 4 * q < MJ_DOUT_LD)  // units 16 o2 + 4 q .. + 3 as one 16-byte store (28 .. 31 do not exist)
        *reinterpret_cast<MJ_LDS f32x4*>(dout + (16 * s + c) * MJ_DOUT_LD + 16 * o2 + 4 * q) = ojvp[o2][s];
  mj_wave_sync();
}

// mlp_jvp_tile3: the evaluator for networks in -> H1 <= 64 -> H2 <= 64 -> 25 (every layer ONE chunk: BASELINE cfg3's
// 28 -> 64 -> 64 -> 25) as its own function, so that its register allocation is its own (a function's footprint is the
// maximum over its paths, and the caller spills around the call what the callee may clobber).
// The network descriptor arrives as SCALAR ARGUMENTS (29 dwords: the AMDGPU calling convention passes the first 32 in
// VGPRs): as one by-value struct it went through private memory - 26 scratch stores in the caller, as many scratch loads
// and a full memory round trip at the head of every evaluation.
// not_tail_called: a call site the optimiser marks `tail` (possible now that no argument lives in the caller's frame)
// switches off LLVM's "no callee-saved registers" treatment of this internal function, and the callee then saves and
// restores ~170 VGPRs through scratch memory on every evaluation.
// BO ("base only"): no JVP - the four unperturbed inputs alone (a storing sweep that is accepted by the residual test needs
// no forward-difference columns: ms_newton, "base-only storing sweeps"); the J dx rows are not written.
template <typename T, int ACT, int VAR = 0, bool BO = false>
__device__ __attribute__((noinline, not_tail_called)) void mlp_jvp_tile3(const float* wq0, const float* wq1, const float* wq2, const float* bq0,
                                                       const float* bq1, const float* bq2, const bf16x8* j0, const bf16x8* j1,
                                                       const bf16x8* j2, int kg1, int kg2, int ot0, int ot1, int jks1, int jks2,
                                                       int Lv, int ptabv, T* scratch_generic, int lane) {
#ifdef MJ_STAMPS
  unsigned long long mj3[12];
#endif
  MJ_T3(0);
  gf4p wq[3] = {(gf4p)uni64((unsigned long long)wq0), (gf4p)uni64((unsigned long long)wq1), (gf4p)uni64((unsigned long long)wq2)};
  gfp bq[3] = {(gfp)uni64((unsigned long long)bq0), (gfp)uni64((unsigned long long)bq1), (gfp)uni64((unsigned long long)bq2)};
  gbfp jq[3] = {(gbfp)uni64((unsigned long long)j0), (gbfp)uni64((unsigned long long)j1), (gbfp)uni64((unsigned long long)j2)};
  const int kgs[3] = {2, uni(kg1), uni(kg2)};
  const int ots[3] = {uni(ot0), uni(ot1), MM_OUT_T};
  const int jkss[3] = {1, uni(jks1), uni(jks2)};
  const int L = uni(Lv);
  const int ptab = uni(ptabv);
  MJ_LDS unsigned char* sbase = (MJ_LDS unsigned char*)(unsigned)__builtin_amdgcn_readfirstlane(
      (int)(unsigned long long)(MJ_LDS unsigned char*)scratch_generic);
  MJ_LDS T* xb = (MJ_LDS T*)sbase;
  MJ_LDS float* actp = (MJ_LDS float*)(sbase + mj_xb_bytes<T>());
  MJ_LDS unsigned char* dreg = sbase + mj_xb_bytes<T>() + MJ_ACTP_BYTES;
  const MJ_LDS bf16x8* dx = (const MJ_LDS bf16x8*)dreg;
  MJ_LDS float* dout = (MJ_LDS float*)dreg;
  const int lo_ = L == 2 ? 1 : 2;
  const int q = lane >> 4, c = lane & 15, j4 = lane & 3;
  const int dunit = 4 * ((lane >> 2) & 3) + (lane >> 4);  // unit of this lane inside a D tile

  double obase[MM_OUT_T][2];  // two partial sums per output tile (4 independent accumulators for the matrix pipe)
#pragma unroll
  for (int o2 = 0; o2 < MM_OUT_T; ++o2) { obase[o2][0] = (double)bq[lo_][o2 * 64 + lane]; obase[o2][1] = 0.0; }
  f32x4 ojvp[MM_OUT_T][4];
#pragma unroll
  for (int o2 = 0; o2 < MM_OUT_T; ++o2)
#pragma unroll
    for (int s = 0; s < 4; ++s) ojvp[o2][s] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto zero_dh = [](f32x4 (&dh)[4][4]) {
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int s = 0; s < 4; ++s) dh[o][s] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto out_layer = [&](const double (&h)[4], const f32x4 (&wo4)[4][MM_OUT_T], auto batched) {
    // output tiles from a 64-unit chunk; even / odd k-groups into separate partial sums
    double a4[4] = {obase[0][0], obase[1][0], obase[0][1], obase[1][1]};
    double bvs[16];
    if constexpr (decltype(batched)::value) chunk_operands(h, bvs);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double bv;
        if constexpr (decltype(batched)::value) bv = bvs[4 * g + e];
        else bv = chunk_operand(h, 4 * g + e);
#pragma unroll
        for (int o2 = 0; o2 < MM_OUT_T; ++o2) a4[2 * (g & 1) + o2] = mfma4(MJ_W(wo4[g][o2], e), bv, a4[2 * (g & 1) + o2]);
      }
    obase[0][0] = a4[0]; obase[1][0] = a4[1]; obase[0][1] = a4[2]; obase[1][1] = a4[3];
  };

  // base inputs as B operand: input 4 ks + (l >> 4) of sample l & 3 (the same in every block); k-step 7 is padding
  double bin[8];
#pragma unroll
  for (int ks = 0; ks < MM_KS1; ++ks) bin[ks] = (double)xb[j4 * MJ_XB_LD + 4 * ks + q];
  bin[7] = 0.0;

  {
    // in -> H1 (<= 64) -> H2 (<= 64) -> 25, everything one chunk (BASELINE cfg3's 28 -> 64 -> 64 -> 25): straight-line
    // code.  (1) The JVP products of a layer are issued BETWEEN the base products of the same layer - they only depend
    // on the previous layer's act' table.  (2) Weight fragments are requested ONE LAYER AHEAD instead of all at the top
    // (sched_barrier keeps the scheduler from hoisting them back): the evaluator's register footprint is what the
    // calling sweep kernel has to spill around every call, and what the compiler shuffles through AGPRs in here.
    f32x4 w1[2][4], w2[4][4];
    bf16x8 a1[1][4];
    base_load<4, 2>(w1, wq[0], 2, 0, 0, lane);
    bf16x8 bdx[4];
    if constexpr (!BO) {
      jvp_load<4, 1>(a1, jq[0], 1, 0, 0, lane);
#pragma unroll
      for (int s = 0; s < 4; ++s) bdx[s] = dx[(16 * s + c) * (MJ_DX_LD / 8) + q];
    }
    double h1[4], h2[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) { h1[o] = (double)bq[0][o * 64 + lane]; h2[o] = (double)bq[1][o * 64 + lane]; }
    base_load<4, 4>(w2, wq[1], 4, 0, 0, lane);
    // Hand-off of a hidden layer to the next product through LDS (2.3 KB behind the dx rows, free until the results are
    // stored): the activated values are written in the ORDER THE NEXT LAYER READS THEM - lane (k, *, j) needs input
    // 4 ks + k of sample j for the 16 k-steps ks, sixteen consecutive doubles of row (j, k) - so the sixteen B operands
    // of a layer are 8 ds_read_b128 behind ONE wait, where the in-register route took 32 ds_swizzle and a wait per
    // k-step.  Rows are 18 doubles apart (bank spread).  The act' table goes through LDS anyway.
    constexpr int HOP_LD = 18;
    // (BO: right behind the input rows - the whole scratch of a base-only evaluation is 112 + 288 elements, which the caller
    //  places on XB / Tm so that Es survives: kr_ms_impl.hpp, MS_BO_PAD)
    MJ_LDS double* hop = BO ? (MJ_LDS double*)(sbase + mj_xb_bytes<T>()) : (MJ_LDS double*)(dreg + 64 * MJ_DX_LD * 2);
    const int hop_w = ((lane & 3) * 4 + (lane >> 4)) * HOP_LD + ((lane >> 2) & 3);  // + 4 o: unit 16 o + 4 b + i = 4 (4 o + b) + i
    const MJ_LDS double* hop_r = hop + ((lane & 3) * 4 + (lane >> 4)) * HOP_LD;
    auto hand_off = [&](const double (&h)[4]) {
#pragma unroll
      for (int o = 0; o < 4; ++o) hop[hop_w + 4 * o] = h[o];
    };
    auto operands = [&](double (&bv)[16]) {
      typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const d2 v = *reinterpret_cast<const MJ_LDS d2*>(hop_r + 2 * e);
        bv[2 * e] = v[0]; bv[2 * e + 1] = v[1];
      }
    };
    f32x4 dh[4][4];
    zero_dh(dh);
    MJ_T3(1);
    // layer 1, base: 32 products
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int o = 0; o < 4; ++o) h1[o] = mfma4(MJ_W(w1[g][o], e), bin[4 * g + e], h1[o]);
    __builtin_amdgcn_sched_barrier(0);
    MJ_T3(2);
    // layer 1, JVP (16 products) under the activation of the base chain: the matrix pipe works while the vector ALU does
    bf16x8 a2[2][4];
    if constexpr (!BO) {
      jvp_load<4, 2>(a2, jq[1], 2, 0, 0, lane);
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int s = 0; s < 4; ++s) dh[o][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[0][o], bdx[s], dh[o][s], 0, 0, 0);
    }
    chunk_activate<ACT, !BO>(h1, actp, lane);
    hand_off(h1);
    MJ_T3(3);
    mj_wave_sync();
    double bv[16];
    operands(bv);
    bf16x8 b1[4][2];
    if constexpr (!BO) jvp_scale_pack(dh, actp, lane, b1, ptab);
    MJ_T3(4);
    // layer 2, base: 64 products; the first k-step of the JVP (16 products) between them
    zero_dh(dh);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int o = 0; o < 4; ++o) h2[o] = mfma4(MJ_W(w2[g][o], e), bv[4 * g + e], h2[o]);
      if constexpr (!BO) {
#pragma unroll
        for (int s = 0; s < 4; ++s) dh[g][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[0][g], b1[s][0], dh[g][s], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    MJ_T3(5);
    // ... its second k-step (16 products) under the second activation
    f32x4 wo4[4][MM_OUT_T];
    bf16x8 ao[2][MM_OUT_T];
    base_load<MM_OUT_T, 4>(wo4, wq[2], kgs[2], 0, 0, lane);
    if constexpr (!BO) {
      jvp_load<MM_OUT_T, 2>(ao, jq[2], jkss[2], 0, 0, lane);
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int s = 0; s < 4; ++s) dh[o][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[1][o], b1[s][1], dh[o][s], 0, 0, 0);
    }
    chunk_activate<ACT, !BO>(h2, actp + 256, lane);
    hand_off(h2);
    MJ_T3(6);
    mj_wave_sync();
    operands(bv);
    bf16x8 b2[4][2];
    if constexpr (!BO) jvp_scale_pack(dh, actp + 256, lane, b2, ptab);
    MJ_T3(7);
    // output layer: even / odd k-groups into separate partial sums; 4 JVP products behind each k-group
    {
      double a4[4] = {obase[0][0], obase[1][0], obase[0][1], obase[1][1]};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int o2 = 0; o2 < MM_OUT_T; ++o2) a4[2 * (g & 1) + o2] = mfma4(MJ_W(wo4[g][o2], e), bv[4 * g + e], a4[2 * (g & 1) + o2]);
        if constexpr (!BO) {
          const int ks = g >> 1, o2 = g & 1;
#pragma unroll
          for (int s = 0; s < 4; ++s) ojvp[o2][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ao[ks][o2], b2[s][ks], ojvp[o2][s], 0, 0, 0);
        }
      }
      obase[0][0] = a4[0]; obase[1][0] = a4[1]; obase[0][1] = a4[2]; obase[1][1] = a4[3];
    }
    MJ_T3(8);
  }
  // results: base outputs over the input rows, J dx rows over the dx rows (every lane has read its operands; the
  // sync orders the other lanes' reads before these writes)
  mj_wave_sync();
#pragma unroll
  for (int o2 = 0; o2 < MM_OUT_T; ++o2)
    if (16 * o2 + dunit < 25) xb[j4 * MJ_XB_LD + 16 * o2 + dunit] = (T)(obase[o2][0] + obase[o2][1]);
  if constexpr (!BO) {
#pragma unroll
    for (int o2 = 0; o2 < MM_OUT_T; ++o2)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (16 * o2 + 4 * q < MJ_DOUT_LD)  // units 16 o2 + 4 q .. + 3 as one 16-byte store (28 .. 31 do not exist)
          *reinterpret_cast<MJ_LDS f32x4*>(dout + (16 * s + c) * MJ_DOUT_LD + 16 * o2 + 4 * q) = ojvp[o2][s];
  }
  mj_wave_sync();
  MJ_T3(9);
#ifdef MJ_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int k = 0; k < 9; ++k) mj_stamp_acc[k] += mj3[k + 1] - mj3[k];
#endif
}


// mlp_jvp_tile3f: the same one-chunk three-layer evaluator for fp32 sweeps with the BASE CHAIN IN FP32 on
// v_mfma_f32_4x4x1_16B_f32 (the fp64 chain above spends 128 v_cvt_f64_f32 - ~1.5 k cycles - on widening fp32 weights, and
// an fp32 sweep cannot use the extra digits).  Layout (measured, tools/ubench_mfma3.hip): 16 blocks of D[4x4] += A[4x1] B[1x4];
// lane l = (block b = l >> 2, index l & 3): A holds row i = l & 3, B column j = l & 3, D register r is D[b][r][j].  Used as
// out^T[64 units x 4 samples]: block b carries units 4 b .. 4 b + 3, so lane l owns ROW l of the weight matrix (A operand
// of k-step k = W[l][k]: fragments [k-group][lane][4], one 16-byte load per 4 k-steps), the B operand of k-step k is
// input k of sample j, the same in every block (16-byte broadcast reads of row j of an LDS tile), and a lane ends up with
// units 4 b .. 4 b + 3 of sample j - contiguous, so activations, act' and the hand-off to the next layer are ONE 16-byte
// LDS write each.  8 cycles per product (4 independent accumulators: one per k mod 4), 124 products.  The output layer
// (25 -> 32 rows) uses the upper 8 blocks for the second half of its k range; the halves meet through ds_bpermute.
template <int ACT, int VAR = 0, bool BO = false>
__device__ __attribute__((noinline, not_tail_called)) void mlp_jvp_tile3f(const float* w0, const float* w1p, const float* w2p,
                                                                          const float* b0, const float* b1p, const float* b2p,
                                                                          const bf16x8* j0, const bf16x8* j1, const bf16x8* j2,
                                                                          int ptabv, float* scratch_generic, int lane) {
  using T = float;
#ifdef MJ_STAMPS
  unsigned long long mj3[12];
#endif
  MJ_T3(0);
  gf4p wf[3] = {(gf4p)uni64((unsigned long long)w0), (gf4p)uni64((unsigned long long)w1p), (gf4p)uni64((unsigned long long)w2p)};
  gf4p bf[3] = {(gf4p)uni64((unsigned long long)b0), (gf4p)uni64((unsigned long long)b1p), (gf4p)uni64((unsigned long long)b2p)};
  gbfp jq[3] = {(gbfp)uni64((unsigned long long)j0), (gbfp)uni64((unsigned long long)j1), (gbfp)uni64((unsigned long long)j2)};
  const int ptab = uni(ptabv);
  MJ_LDS unsigned char* sbase = (MJ_LDS unsigned char*)(unsigned)__builtin_amdgcn_readfirstlane(
      (int)(unsigned long long)(MJ_LDS unsigned char*)scratch_generic);
  MJ_LDS T* xb = (MJ_LDS T*)sbase;
  MJ_LDS float* actp = (MJ_LDS float*)(sbase + mj_xb_bytes<T>());
  MJ_LDS unsigned char* dreg = sbase + mj_xb_bytes<T>() + MJ_ACTP_BYTES;
  const MJ_LDS bf16x8* dx = (const MJ_LDS bf16x8*)dreg;
  MJ_LDS float* dout = (MJ_LDS float*)dreg;
  constexpr int HOP_LD = 68;  // floats per sample row of the hand-off tile (bank spread of the four rows)
  // (BO - base only, see mlp_jvp_tile3: the hand-off rows right behind the input rows, 112 + 272 elements in all)
  MJ_LDS float* hop = BO ? (MJ_LDS float*)(sbase + mj_xb_bytes<T>()) : (MJ_LDS float*)(dreg + 64 * MJ_DX_LD * 2);
  const int q = lane >> 4, c = lane & 15, j4 = lane & 3, blk = lane >> 2;
  auto mfma1 = [](float a, float b, f32x4 acc) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0); };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  f32x4 W1[7];
#pragma unroll
  for (int g = 0; g < 7; ++g) W1[g] = wf[0][g * 64 + lane];
  bf16x8 a1[1][4];
  if constexpr (!BO) jvp_load<4, 1>(a1, jq[0], 1, 0, 0, lane);
  f32x4 acc[4] = {bf[0][blk], zero4, zero4, zero4};
  f32x4 xin[16];
#pragma unroll
  for (int g = 0; g < 7; ++g) xin[g] = *reinterpret_cast<const MJ_LDS f32x4*>(xb + j4 * MJ_XB_LD + 4 * g);
  bf16x8 bdx[4];
  if constexpr (!BO) {
#pragma unroll
    for (int s = 0; s < 4; ++s) bdx[s] = dx[(16 * s + c) * (MJ_DX_LD / 8) + q];
  }
  f32x4 W2[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) W2[g] = wf[1][g * 64 + lane];
  const f32x4 bias2 = bf[1][blk];
  f32x4 dh[4][4];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int s = 0; s < 4; ++s) dh[o][s] = zero4;
  f32x4 ojvp[MM_OUT_T][4];
#pragma unroll
  for (int o2 = 0; o2 < MM_OUT_T; ++o2)
#pragma unroll
    for (int s = 0; s < 4; ++s) ojvp[o2][s] = zero4;
  // activation of the lane's four units, act' table and hand-off row: 16-byte writes
  auto activate_hand_off = [&](f32x4& h, MJ_LDS float* tab) {
    float v[4] = {h[0], h[1], h[2], h[3]};
    activate_block<float, ACT, 4>(v);
    f32x4 a = {v[0], v[1], v[2], v[3]};
    f32x4 gr = {act_grad_from_value<ACT>(v[0]), act_grad_from_value<ACT>(v[1]), act_grad_from_value<ACT>(v[2]),
                act_grad_from_value<ACT>(v[3])};
    if constexpr (!BO) *reinterpret_cast<MJ_LDS f32x4*>(tab + j4 * 64 + 4 * blk) = gr;
    *reinterpret_cast<MJ_LDS f32x4*>(hop + j4 * HOP_LD + 4 * blk) = a;
    h = a;
  };
  MJ_T3(1);
  // layer 1, base: 28 products
#pragma unroll
  for (int g = 0; g < 7; ++g) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = mfma1(W1[g][e], xin[g][e], acc[e]);
#if MJ_VARIANT & 1
    if (!BO && g < 4) {
#pragma unroll
      for (int s = 0; s < 4; ++s) dh[g][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[0][g], bdx[s], dh[g][s], 0, 0, 0);
    }
#endif
  }
  __builtin_amdgcn_sched_barrier(0);
  MJ_T3(2);
  // layer 1, JVP (16 products) under the activation
  bf16x8 a2[2][4];
  if constexpr (!BO) {
    jvp_load<4, 2>(a2, jq[1], 2, 0, 0, lane);
#if !(MJ_VARIANT & 1)
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int s = 0; s < 4; ++s) dh[o][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[0][o], bdx[s], dh[o][s], 0, 0, 0);
#endif
  }
  f32x4 h = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  activate_hand_off(h, actp);
  MJ_T3(3);
  mj_wave_sync();
#pragma unroll
  for (int g = 0; g < 16; ++g) xin[g] = *reinterpret_cast<const MJ_LDS f32x4*>(hop + j4 * HOP_LD + 4 * g);
  bf16x8 b1[4][2];
  if constexpr (!BO) jvp_scale_pack(dh, actp, lane, b1, ptab);
  MJ_T3(4);
  // layer 2, base: 64 products; the first k-step of the JVP (16 products) between them
  acc[0] = bias2; acc[1] = zero4; acc[2] = zero4; acc[3] = zero4;
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int s = 0; s < 4; ++s) dh[o][s] = zero4;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = mfma1(W2[g][e], xin[g][e], acc[e]);
    if (!BO && (g & 3) == 3) {
      const int o = g >> 2;
#pragma unroll
      for (int s = 0; s < 4; ++s) dh[o][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[0][o], b1[s][0], dh[o][s], 0, 0, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  MJ_T3(5);
  // ... its second k-step (16 products) under the second activation
  f32x4 Wo[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) Wo[g] = wf[2][g * 64 + lane];
  bf16x8 ao[2][MM_OUT_T];
  const f32x4 bias3 = bf[2][blk];
  if constexpr (!BO) {
    jvp_load<MM_OUT_T, 2>(ao, jq[2], 2, 0, 0, lane);
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int s = 0; s < 4; ++s) dh[o][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[1][o], b1[s][1], dh[o][s], 0, 0, 0);
  }
  h = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  activate_hand_off(h, actp + 256);
  MJ_T3(6);
  mj_wave_sync();
  // output layer: lanes of the blocks 0..7 take k = 0..31, those of 8..15 k = 32..63 (rows 32..63 of the fragment array)
  const int khalf = lane >> 5;
#pragma unroll
  for (int g = 0; g < 8; ++g) xin[g] = *reinterpret_cast<const MJ_LDS f32x4*>(hop + j4 * HOP_LD + 32 * khalf + 4 * g);
  bf16x8 b2[4][2];
  if constexpr (!BO) jvp_scale_pack(dh, actp + 256, lane, b2, ptab);
  MJ_T3(7);
  acc[0] = bias3; acc[1] = zero4; acc[2] = zero4; acc[3] = zero4;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = mfma1(Wo[g][e], xin[g][e], acc[e]);
    if (!BO && (g & 1)) {
      const int ks = g >> 2, o2 = (g >> 1) & 1;
#pragma unroll
      for (int s = 0; s < 4; ++s) ojvp[o2][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ao[ks][o2], b2[s][ks], ojvp[o2][s], 0, 0, 0);
    }
  }
  h = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  MJ_T3(8);
  // the two halves of the k range meet: lane l < 32 adds what lane l + 32 summed
  {
    // (component by component: written as a loop over h[r] with __builtin_bit_cast, hipcc 7.2 emitted ONE ds_bpermute and
    //  added its result to all four components)
    const int partner = (lane ^ 32) * 4;
    const float h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3];
    const float o0 = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(h0)));
    const float o1 = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(h1)));
    const float o2 = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(h2)));
    const float o3 = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(h3)));
    h = f32x4{h0 + o0, h1 + o1, h2 + o2, h3 + o3};
  }
  mj_wave_sync();
  if (lane < 28) *reinterpret_cast<MJ_LDS f32x4*>(xb + j4 * MJ_XB_LD + 4 * blk) = h;  // units 4 blk .. 4 blk + 3 < 28 (25 used)
  if constexpr (!BO) {
#pragma unroll
    for (int o2 = 0; o2 < MM_OUT_T; ++o2)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (16 * o2 + 4 * q < MJ_DOUT_LD)
          *reinterpret_cast<MJ_LDS f32x4*>(dout + (16 * s + c) * MJ_DOUT_LD + 16 * o2 + 4 * q) = ojvp[o2][s];
  }
  mj_wave_sync();
  MJ_T3(9);
#ifdef MJ_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int k = 0; k < 9; ++k) mj_stamp_acc[k] += mj3[k + 1] - mj3[k];
#endif
}

// Per-lane wrapper.  iv / col: the lane's role (sub-interval of the wavefront 0..3, 0 = unperturbed); idle lanes pass
// col = 0 and a copy of their interval's state.  zrow: the dx row a lane without a column zeroes (the rows no column
// owns: every sample tile has 16, an interval 6 or 16 columns; -1 = lane layout 7 + 3 x 17 of kr_ms_impl.hpp).
// xrow: the dx / J dx row of a lane with a column when it is not 16 iv + col - 1 (the p columns in sample tile 0).
// x in, NN(x) (base lanes) or NN(x_base) + J dx (the others) out.
// fp64 sweep, fp32 BASE CHAIN (mlp_jvp_tile3f) - for a sweep whose result only feeds a Newton update that is far from the
// tolerance anyway (the first sweep of a step that will need two corrections: kr_ms_impl.hpp, ms_newton).  The evaluator
// sees the wavefront's scratch in its fp32 layout; the fp64 base rows the dx rows are formed from (x - x_base must be taken
// in fp64: the perturbations are 1e-7 relative) sit in the act' area, which the evaluator only writes later.
template <int VAR = 0>
__device__ __forceinline__ void mlp_jvp_eval_lowp(const MlpDev<double>& M, const double (&x)[MM_IN], double* scratch, int lane,
                                                  int iv, int col, bool idle, int zrow, double (&out)[25], int xrow, int ptab) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  unsigned char* sb = reinterpret_cast<unsigned char*>(scratch);
  float* xbf = reinterpret_cast<float*>(sb);
  double* xb64 = reinterpret_cast<double*>(sb + mj_xb_bytes<float>());
  unsigned char* dreg = sb + mj_xb_bytes<float>() + MJ_ACTP_BYTES;
  const bool base = col == 0 && !idle;
  if (base) {
    d2* d = reinterpret_cast<d2*>(xb64 + iv * MJ_XB_LD);
    f32x4* f = reinterpret_cast<f32x4*>(xbf + iv * MJ_XB_LD);
#pragma unroll
    for (int k = 0; k < MM_IN / 2; ++k) d[k] = d2{x[2 * k], x[2 * k + 1]};
#pragma unroll
    for (int k = 0; k < MM_IN / 4; ++k) f[k] = f32x4{(float)x[4 * k], (float)x[4 * k + 1], (float)x[4 * k + 2], (float)x[4 * k + 3]};
  }
  mj_wave_sync();
  {
    const bool has = col > 0;
    const int row = has ? (xrow >= 0 ? xrow : 16 * iv + col - 1) : (zrow >= 0 ? zrow : 6 + (idle ? 4 + (lane - 58) : iv));
    float d[MM_IN];
    const d2* b = reinterpret_cast<const d2*>(xb64 + iv * MJ_XB_LD);
#pragma unroll
    for (int k = 0; k < MM_IN / 2; ++k) {
      const d2 v = b[k];
      d[2 * k] = has ? (float)(x[2 * k] - v[0]) : 0.f;
      d[2 * k + 1] = has ? (float)(x[2 * k + 1] - v[1]) : 0.f;
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4* dst = reinterpret_cast<u32x4*>(dreg + (size_t)row * (MJ_DX_LD * 2));
    unsigned zpad;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zpad));
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      u32x4 p;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = 8 * v + 2 * e;
        p[e] = k + 1 < MM_IN ? pack_bf2(d[k < MM_IN ? k : 0], d[k + 1 < MM_IN ? k + 1 : 0]) : zpad;
      }
      dst[v] = p;
    }
  }
  mj_wave_sync();
#define MJ_F_ARGS                                                                                                    \
  M.w32[0], M.w32[1], M.w32[2], M.b32[0], M.b32[1], M.b32[2], reinterpret_cast<const bf16x8*>(M.jfrag[0]),             \
      reinterpret_cast<const bf16x8*>(M.jfrag[1]), reinterpret_cast<const bf16x8*>(M.jfrag[2]), ptab, xbf, lane
  switch (M.acts[0]) {
    case KR_ACT_TANH: mlp_jvp_tile3f<KR_ACT_TANH, VAR>(MJ_F_ARGS); break;
    case KR_ACT_SOFTPLUS: mlp_jvp_tile3f<KR_ACT_SOFTPLUS, VAR>(MJ_F_ARGS); break;
    case KR_ACT_RELU: mlp_jvp_tile3f<KR_ACT_RELU, VAR>(MJ_F_ARGS); break;
    case KR_ACT_ELU: mlp_jvp_tile3f<KR_ACT_ELU, VAR>(MJ_F_ARGS); break;
    default: mlp_jvp_tile3f<KR_ACT_NONE, VAR>(MJ_F_ARGS); break;
  }
#undef MJ_F_ARGS
  {
    const f32x4* b = reinterpret_cast<const f32x4*>(xbf + iv * MJ_XB_LD);
    const int row = col > 0 ? (xrow >= 0 ? xrow : 16 * iv + col - 1) : (zrow >= 0 ? zrow : 6 + (idle ? 4 + (lane - 58) : iv));
    const bool has = col > 0;
    const f32x4* dr = reinterpret_cast<const f32x4*>(dreg + (size_t)row * (MJ_DOUT_LD * 4));
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const f32x4 v = b[k], w = dr[k];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * k + e < 25) out[4 * k + e] = (double)v[e] + (has ? (double)w[e] : 0.0);  // (the sum in fp64: J dx is ~1e-7 of the base value)
    }
  }
  mj_wave_sync();
}

// Base-only evaluation (fp64 sweeps, one-chunk three-layer networks): the network at the four unperturbed inputs; EVERY lane
// gets the output of its interval's unperturbed lane (the perturbed trajectories of such a sweep are not used - they
// only have to stay finite).  No dx rows, no JVP products, no J dx read-back: about half an evaluation.
template <int VAR = 0>
__device__ __forceinline__ void mlp_jvp_eval_base(const MlpDev<double>& M, const double (&x)[MM_IN], double* scratch, int lane,
                                                  int iv, int col, bool idle, double (&out)[25], int ptab) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  double* xb = scratch;
  if (col == 0 && !idle) {
    d2* d = reinterpret_cast<d2*>(xb + iv * MJ_XB_LD);
#pragma unroll
    for (int k = 0; k < MM_IN / 2; ++k) d[k] = d2{x[2 * k], x[2 * k + 1]};
  }
  mj_wave_sync();
#define MJ_B_ARGS                                                                                                           \
  M.wq[0], M.wq[1], M.wq[2], M.bq[0], M.bq[1], M.bq[2], reinterpret_cast<const bf16x8*>(M.jfrag[0]),                         \
      reinterpret_cast<const bf16x8*>(M.jfrag[1]), reinterpret_cast<const bf16x8*>(M.jfrag[2]), M.kgroups[1], M.kgroups[2],  \
      M.otiles[0], M.otiles[1], M.jksteps[1], M.jksteps[2], M.n_layers, ptab, scratch, lane
  switch (M.acts[0]) {
    case KR_ACT_TANH: mlp_jvp_tile3<double, KR_ACT_TANH, VAR, true>(MJ_B_ARGS); break;
    case KR_ACT_SOFTPLUS: mlp_jvp_tile3<double, KR_ACT_SOFTPLUS, VAR, true>(MJ_B_ARGS); break;
    case KR_ACT_RELU: mlp_jvp_tile3<double, KR_ACT_RELU, VAR, true>(MJ_B_ARGS); break;
    case KR_ACT_ELU: mlp_jvp_tile3<double, KR_ACT_ELU, VAR, true>(MJ_B_ARGS); break;
    default: mlp_jvp_tile3<double, KR_ACT_NONE, VAR, true>(MJ_B_ARGS); break;
  }
#undef MJ_B_ARGS
  {
    const d2* b = reinterpret_cast<const d2*>(xb + iv * MJ_XB_LD);
#pragma unroll
    for (int k = 0; k < 13; ++k) {
      const d2 v = b[k];
      out[2 * k] = v[0];
      if (2 * k + 1 < 25) out[2 * k + 1] = v[1];
    }
  }
  mj_wave_sync();
}

// the same for fp32 sweeps (fp32 base chain, mlp_jvp_tile3f)
template <int VAR = 0>
__device__ __forceinline__ void mlp_jvp_eval_base(const MlpDev<float>& M, const float (&x)[MM_IN], float* scratch, int lane,
                                                  int iv, int col, bool idle, float (&out)[25], int ptab) {
  float* xb = scratch;
  if (col == 0 && !idle) {
    f32x4* d = reinterpret_cast<f32x4*>(xb + iv * MJ_XB_LD);
#pragma unroll
    for (int k = 0; k < MM_IN / 4; ++k) d[k] = f32x4{x[4 * k], x[4 * k + 1], x[4 * k + 2], x[4 * k + 3]};
  }
  mj_wave_sync();
#define MJ_B_ARGS                                                                                                    \
  M.w32[0], M.w32[1], M.w32[2], M.b32[0], M.b32[1], M.b32[2], reinterpret_cast<const bf16x8*>(M.jfrag[0]),             \
      reinterpret_cast<const bf16x8*>(M.jfrag[1]), reinterpret_cast<const bf16x8*>(M.jfrag[2]), ptab, scratch, lane
  switch (M.acts[0]) {
    case KR_ACT_TANH: mlp_jvp_tile3f<KR_ACT_TANH, VAR, true>(MJ_B_ARGS); break;
    case KR_ACT_SOFTPLUS: mlp_jvp_tile3f<KR_ACT_SOFTPLUS, VAR, true>(MJ_B_ARGS); break;
    case KR_ACT_RELU: mlp_jvp_tile3f<KR_ACT_RELU, VAR, true>(MJ_B_ARGS); break;
    case KR_ACT_ELU: mlp_jvp_tile3f<KR_ACT_ELU, VAR, true>(MJ_B_ARGS); break;
    default: mlp_jvp_tile3f<KR_ACT_NONE, VAR, true>(MJ_B_ARGS); break;
  }
#undef MJ_B_ARGS
  {
    const f32x4* b = reinterpret_cast<const f32x4*>(xb + iv * MJ_XB_LD);
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const f32x4 v = b[k];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * k + e < 25) out[4 * k + e] = v[e];
    }
  }
  mj_wave_sync();
}

template <typename T, int VAR = 0>
__device__ __forceinline__ void mlp_jvp_eval(const MlpDev<T>& M, const T (&x)[MM_IN], T* scratch, int lane, int iv, int col,
                                             bool idle, int zrow, T (&out)[25], int xrow = -1, int ptab = 0, bool lowp = false,
                                             bool base_only = false) {
  using V = typename MjVec<T>::type;
  constexpr int n = MjVec<T>::n;
  if constexpr (std::is_same<T, float>::value) {
    if (__builtin_amdgcn_readfirstlane((int)base_only) && M.f32_ok) {  // wave-uniform
      mlp_jvp_eval_base<VAR>(M, x, scratch, lane, iv, col, idle, out, ptab);
      return;
    }
  }
  if constexpr (std::is_same<T, double>::value) {
    if (__builtin_amdgcn_readfirstlane((int)base_only) && M.n_layers == 3 && M.otiles[1] == 4) {  // wave-uniform
      mlp_jvp_eval_base<VAR>(M, x, scratch, lane, iv, col, idle, out, ptab);
      return;
    }
    if (__builtin_amdgcn_readfirstlane((int)lowp) && M.f32_ok) {  // wave-uniform
      mlp_jvp_eval_lowp<VAR>(M, x, scratch, lane, iv, col, idle, zrow, out, xrow, ptab);
      return;
    }
  }
  T* xb = scratch;
  unsigned char* dreg = reinterpret_cast<unsigned char*>(scratch) + mj_xb_bytes<T>() + MJ_ACTP_BYTES;
  const bool base = col == 0 && !idle;
  if (base) {
    V* d = reinterpret_cast<V*>(xb + iv * MJ_XB_LD);
#pragma unroll
    for (int k = 0; k < MM_IN / n; ++k) {
      V v;
#pragma unroll
      for (int e = 0; e < n; ++e) v[e] = x[k * n + e];
      d[k] = v;
    }
  }
  mj_wave_sync();
  // dx row of this lane; the lanes without a column (base and idle ones) zero the rows no column owns.  Written without
  // a branch on the lane's role: the wrapper sits in the middle of the register-starved sweep kernels, and spill code
  // inside divergent control flow there has produced wrong reloads (hipcc 7.2, fp64, kr_msw_impl.hpp with the MLP on).
  {
    const bool has = col > 0;
    const int row = has ? (xrow >= 0 ? xrow : 16 * iv + col - 1) : (zrow >= 0 ? zrow : 6 + (idle ? 4 + (lane - 58) : iv));
    float d[MM_IN];
    const V* b = reinterpret_cast<const V*>(xb + iv * MJ_XB_LD);
#pragma unroll
    for (int k = 0; k < MM_IN / n; ++k) {
      const V v = b[k];
#pragma unroll
      for (int e = 0; e < n; ++e) d[k * n + e] = has ? (float)(x[k * n + e] - v[e]) : 0.f;
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4* dst = reinterpret_cast<u32x4*>(dreg + (size_t)row * (MJ_DX_LD * 2));
    // the padding columns 28..31: a zero the optimiser cannot share with other zeros of the kernel (as a common
    // constant it was kept in a register tuple across the whole kernel, spilled, and that spill is what went wrong)
    unsigned zpad;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zpad));
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      u32x4 p;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = 8 * v + 2 * e;
        p[e] = k + 1 < MM_IN ? pack_bf2(d[k < MM_IN ? k : 0], d[k + 1 < MM_IN ? k + 1 : 0]) : zpad;
      }
      dst[v] = p;
    }
  }
  mj_wave_sync();
#define MJ_TILE_ARGS                                                                                                        \
  M.wq[0], M.wq[1], M.wq[2], M.bq[0], M.bq[1], M.bq[2], reinterpret_cast<const bf16x8*>(M.jfrag[0]),                         \
      reinterpret_cast<const bf16x8*>(M.jfrag[1]), reinterpret_cast<const bf16x8*>(M.jfrag[2]), M.kgroups[1], M.kgroups[2],  \
      M.otiles[0], M.otiles[1], M.jksteps[1], M.jksteps[2], M.n_layers, ptab, scratch, lane
  if constexpr (std::is_same<T, float>::value) {
    if (M.f32_ok) {  // wave-uniform: one chunk per layer, fp32 base chain
#define MJ_F_ARGS                                                                                                    \
  M.w32[0], M.w32[1], M.w32[2], M.b32[0], M.b32[1], M.b32[2], reinterpret_cast<const bf16x8*>(M.jfrag[0]),             \
      reinterpret_cast<const bf16x8*>(M.jfrag[1]), reinterpret_cast<const bf16x8*>(M.jfrag[2]), ptab, scratch, lane
      switch (M.acts[0]) {
        case KR_ACT_TANH: mlp_jvp_tile3f<KR_ACT_TANH, VAR>(MJ_F_ARGS); break;
        case KR_ACT_SOFTPLUS: mlp_jvp_tile3f<KR_ACT_SOFTPLUS, VAR>(MJ_F_ARGS); break;
        case KR_ACT_RELU: mlp_jvp_tile3f<KR_ACT_RELU, VAR>(MJ_F_ARGS); break;
        case KR_ACT_ELU: mlp_jvp_tile3f<KR_ACT_ELU, VAR>(MJ_F_ARGS); break;
        default: mlp_jvp_tile3f<KR_ACT_NONE, VAR>(MJ_F_ARGS); break;
      }
#undef MJ_F_ARGS
      goto mj_evaluated;
    }
  }
  if (M.n_layers == 3 && M.otiles[1] == 4) {  // wave-uniform: one chunk per layer
    switch (M.acts[0]) {
      case KR_ACT_TANH: mlp_jvp_tile3<T, KR_ACT_TANH, VAR>(MJ_TILE_ARGS); break;
      case KR_ACT_SOFTPLUS: mlp_jvp_tile3<T, KR_ACT_SOFTPLUS, VAR>(MJ_TILE_ARGS); break;
      case KR_ACT_RELU: mlp_jvp_tile3<T, KR_ACT_RELU, VAR>(MJ_TILE_ARGS); break;
      case KR_ACT_ELU: mlp_jvp_tile3<T, KR_ACT_ELU, VAR>(MJ_TILE_ARGS); break;
      default: mlp_jvp_tile3<T, KR_ACT_NONE, VAR>(MJ_TILE_ARGS); break;
    }
  } else
  switch (M.acts[0]) {  // wave-uniform
    case KR_ACT_TANH: mlp_jvp_tile<T, KR_ACT_TANH, VAR>(MJ_TILE_ARGS); break;
    case KR_ACT_SOFTPLUS: mlp_jvp_tile<T, KR_ACT_SOFTPLUS, VAR>(MJ_TILE_ARGS); break;
    case KR_ACT_RELU: mlp_jvp_tile<T, KR_ACT_RELU, VAR>(MJ_TILE_ARGS); break;
    case KR_ACT_ELU: mlp_jvp_tile<T, KR_ACT_ELU, VAR>(MJ_TILE_ARGS); break;
    default: mlp_jvp_tile<T, KR_ACT_NONE, VAR>(MJ_TILE_ARGS); break;
  }
#undef MJ_TILE_ARGS
mj_evaluated:
  {
    const V* b = reinterpret_cast<const V*>(xb + iv * MJ_XB_LD);
    constexpr int NB = (25 + n - 1) / n;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const V v = b[k];
#pragma unroll
      for (int e = 0; e < n; ++e)
        if (k * n + e < 25) out[k * n + e] = v[e];
    }
    {  // (no branch on the role here either: a lane without a column adds its zero row)
      const int row = col > 0 ? (xrow >= 0 ? xrow : 16 * iv + col - 1) : (zrow >= 0 ? zrow : 6 + (idle ? 4 + (lane - 58) : iv));
      const bool has = col > 0;  // (a select, not a product: the rows of a missing interval may hold NaN)
      const f32x4* dr = reinterpret_cast<const f32x4*>(dreg + (size_t)row * (MJ_DOUT_LD * 4));
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const f32x4 v = dr[k];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * k + e < 25) out[4 * k + e] += has ? (T)v[e] : T(0);
      }
    }
  }
  mj_wave_sync();
}

}  // namespace kr
