// cst_common.h — shared device/host helpers for the gfx950 kernels (wave64, MFMA 32x32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include "../../include/cst.h"

#define CST_WAVE 64

// ---------------------------------------------------------------------------------------
// host side: error reporting + launch bookkeeping (profiling table)
// ---------------------------------------------------------------------------------------
void cst_set_error(const char* fmt, ...);
struct CstProfScope {
  int cls; hipStream_t s; int slot;
  CstProfScope(int cls, hipStream_t s, double flops, double bytes);
  ~CstProfScope();
  void tag(const char* fmt, ...);  // free-form description of this launch (shape, kernel family) for cst_prof_dump; no-op when off
};
int cst_check_launch(const char* what);
// attention.hip: the matrix-core kernel of cst_dec_cross_attn / cst_dec_ln_q_cross_attn (bf16, head dim 64, beam <= 32; K / V head-major
// [bsz][H][S][64]); Wg != NULL: q is the residual stream and the LayerNorm-folded query projection runs inside the kernel
int cst_fa_dec_cross(const void* q, const void* kx, const void* vx, const uint8_t* kpm, void* out, const int32_t* step, int64_t max_len,
                     int64_t bsz, int64_t beam, int64_t H, int64_t S, float scale, const void* Wg, const float* sg, const float* sb, float eps,
                     int64_t K, int64_t ldx, hipStream_t s);
bool cst_prof_is_on();  // the hipEvent profiling table of bench.py's roofline step is recording

#define CST_REQUIRE(cond, ...)                         \
  do {                                                 \
    if (!(cond)) {                                     \
      cst_set_error(__VA_ARGS__);                      \
      return CST_ERR_BAD_ARG;                          \
    }                                                  \
  } while (0)

static inline int64_t cst_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t cst_dtype_size(int dt) { return dt == CST_BF16 ? 2 : 4; }

// ---------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------
typedef __bf16 bf16_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned int)b) << 16);
}
// float -> bf16, round-to-nearest-even.  A plain cast: hipcc lowers it to the gfx950 hardware converter
// (v_cvt_pk_bf16_f32), branch-free — a hand-written bit-twiddling version with a NaN branch cost ~8 VALU + an exec-mask
// branch per element and dominated the attention inner loop.
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  return __builtin_bit_cast(unsigned short, static_cast<__bf16>(f));
}

template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int VEC = 4;  // elements per 16-byte vector
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct DT<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) {
    return bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(p));
  }
  __device__ static __forceinline__ void st(bf16_t* p, float v) {
    *reinterpret_cast<unsigned short*>(p) = f32_to_bf16_bits(v);
  }
};

// 8 consecutive elements <-> 8 floats (vectorised: 16 B for bf16, 2 x 16 B for f32)
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
  v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(r[i] << 16);
    v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = static_cast<__bf16>(v[i]);
  *reinterpret_cast<bf16x8*>(p) = r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------------------
// Dropout masks are COUNTER-BASED: the keep/drop decision of element `idx` of a dropout site is a pure function of
// (key, idx), so backward kernels and fused epilogues regenerate the forward mask instead of storing it (the reference's
// modules/fairseq_dropout.py stores nothing either: autograd keeps the mask tensor; here no mask tensor exists).
//   x = (lo32(idx/2) ^ key) + hi32(idx/2) * 0x9E3779B1;  x *= 0x9E3779B1;  x = (x ^ x >> 15) + key2;  x *= 0x85EBCA77;  x ^= x >> 13
//   (key2 = key * 0x2C1B3C6D + 0x297A2D39: the key enters twice, so two sites are not index-permutations of one sequence);
//   one 32-bit word serves the element pair (idx & ~1, idx | 1):
//   keep  = 16-bit half (idx & 1) of x >= thr16,  thr16 = round(p * 65536);   kept values are scaled by 1 / (1 - p).
// 7 integer VALU ops per pair: the masks are regenerated inside VALU-bound kernels (attention softmax, GEMM epilogues).
// `key` = host-side mix of (seed + update number, site ordinal): see chimera-st_amd/rng.py (same function in numpy).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t cst_drop_bits(uint32_t key, uint64_t pair) {
  uint32_t x = ((uint32_t)pair ^ key) + (uint32_t)(pair >> 32) * 0x9E3779B1U;
  x *= 0x9E3779B1U;
  x = (x ^ (x >> 15)) + (key * 0x2C1B3C6DU + 0x297A2D39U);
  x *= 0x85EBCA77U;
  return x ^ (x >> 13);
}
// 32-bit pair index variant (tensors below 2^33 elements: attention probabilities, activations)
__device__ __forceinline__ uint32_t cst_drop_bits32(uint32_t key, uint32_t key2, uint32_t pair) {
  uint32_t x = (pair ^ key) * 0x9E3779B1U;
  x = (x ^ (x >> 15)) + key2;
  x *= 0x85EBCA77U;
  return x ^ (x >> 13);
}
__device__ __forceinline__ uint32_t cst_drop_key2(uint32_t key) { return key * 0x2C1B3C6DU + 0x297A2D39U; }
// Attention-probability dropout: the mask is generated ON THE MATRIX CORES.  Hashing a mask word per score pair cost more VALU issue
// than a 32 x 64 score tile's 16 MFMAs (270 of 390 VALU instructions per tile in round 2; a separable hash of round 3 still 4
// operations per score, 9 in the dK/dV kernel where a lane owns a key), in kernels whose matrix pipe idles two thirds of the time.
//   every probability ROW rho = (b*H + h)*Tq + q and every KEY k carry a 32-byte signature of pseudo-random int8:
//     sigB(rho)[4w .. 4w+3] = bytes of  cst_drop_bits32(key, key2, 8 rho + w),   sigA(k)[4w .. 4w+3] = bytes of cst_drop_bits32(key ^ 0x68E31DA4, key2, 8 k + w)
//   D(rho, k) = sum_i sigA(k)[i] * sigB(rho)[i]     (int32; ONE v_mfma_i32_32x32x32_i8 yields it for a whole 32 x 32 score block,
//                                                     in the accumulator layout the scores themselves have)
//   keep(rho, k)  <=>  int16((D << 2) & 0xffff) >= thr16 - 32768        (the low 14 bits of D: sigma(D) = 31 k, so D mod 2^14 is uniform
//                                                     to 1e-30, while D mod 2^16 carries a 2.5 % ripple — measured in the numpy twin)
// For a fixed key-signature table the decisions of two rows are independent (independent sigB), and those of two keys of one row are
// a pair of dot products of one random vector with two fixed ones — the classic pairwise-independent family.  Numpy twin:
// rng.keep_mask_attn_numpy (max |column correlation| 0.07, rows 0.14 on 6000 x 1500 = the iid level; drop rate 0.1000).
// Cost per score: fast forward 2 packed operations (+ 2 MFMAs per 64-key tile), dQ 3, dK/dV 4; signatures are hashed once per lane
// (the operand a lane owns) and once per tile and thread (the staged operand: 2 words each).
__device__ __forceinline__ uint32_t cst_asig_row(uint32_t key, uint32_t key2, uint32_t rho, int w) { return cst_drop_bits32(key, key2, 8u * rho + (uint32_t)w); }
__device__ __forceinline__ uint32_t cst_asig_key(uint32_t key, uint32_t key2, uint32_t k, int w) { return cst_drop_bits32(key ^ 0x68E31DA4U, key2, 8u * k + (uint32_t)w); }
typedef int cst_i32x4 __attribute__((ext_vector_type(4)));
typedef int cst_i32x16 __attribute__((ext_vector_type(16)));
// the 16 signature bytes a lane supplies to the i8 MFMA: words 4 hi .. 4 hi + 3 (hi = lane >> 5 = the k half)
__device__ __forceinline__ cst_i32x4 cst_asig_row_frag(uint32_t key, uint32_t key2, uint32_t rho, int hi) {
  cst_i32x4 f;
#pragma unroll
  for (int w = 0; w < 4; ++w) f[w] = (int)cst_asig_row(key, key2, rho, 4 * hi + w);
  return f;
}
__device__ __forceinline__ cst_i32x4 cst_asig_key_frag(uint32_t key, uint32_t key2, uint32_t k, int hi) {
  cst_i32x4 f;
#pragma unroll
  for (int w = 0; w < 4; ++w) f[w] = (int)cst_asig_key(key, key2, k, 4 * hi + w);
  return f;
}
// D of a 32 x 32 block: rows = the operand passed first
__device__ __forceinline__ cst_i32x16 cst_asig_block(const cst_i32x4& rows, const cst_i32x4& cols) {
  cst_i32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0;
  return __builtin_amdgcn_mfma_i32_32x32x32_i8(rows, cols, z, 0, 0, 0);
}
__device__ __forceinline__ bool cst_adrop_keep(int D, int thr_s) { return (int)(short)((unsigned)D << 2) >= thr_s; }
// multiply 8 consecutive elements starting at the (even) element index idx0 by their dropout factors
__device__ __forceinline__ void cst_drop8(float (&v)[8], uint32_t key, uint64_t idx0, uint32_t thr16, float scale) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t b = cst_drop_bits(key, (idx0 >> 1) + j);
    v[2 * j] *= (b & 0xffffU) >= thr16 ? scale : 0.0f;
    v[2 * j + 1] *= (b >> 16) >= thr16 ? scale : 0.0f;
  }
}
__device__ __forceinline__ float cst_drop1(uint32_t key, uint64_t idx, uint32_t thr16, float scale) {
  const uint32_t b = cst_drop_bits(key, idx >> 1);
  return (((idx & 1) ? (b >> 16) : (b & 0xffffU)) >= thr16) ? scale : 0.0f;
}
__host__ __device__ inline uint32_t cst_drop_thr16(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }

// exact GELU (erf form, modules/gelu.py:25) and its derivative.  erf via Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, below
// fp32 rounding of the products it feeds): 1 + erf(u) = 2 - P(t) E for u >= 0 and P(t) E for u < 0, with t = 1/(1 + p|u|),
// E = exp(-u^2) = exp(-x^2/2) — the same exponential the derivative's density term needs.  ~14 VALU ops instead of libm erff's
// ~50: the activation epilogue of the 768 -> 3072 GEMMs is VALU-bound (147 M elements per call), not MFMA-bound.
__device__ __forceinline__ float gelu_tail_f(float x, float& E) {  // returns P(t) * E = 1 - erf(|x| / sqrt 2)
  const float u = fabsf(x) * 0.70710678118654752f;
  const float t = __frcp_rn(fmaf(0.3275911f, u, 1.0f));
  E = __expf(-0.5f * x * x);
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  return pl * t * E;
}
__device__ __forceinline__ float gelu_f(float x) {
  float E;
  const float q = 0.5f * x * gelu_tail_f(x, E);  // 0.5 x (1 - erf|u|)
  return x >= 0.0f ? x - q : q;
}
__device__ __forceinline__ float dgelu_f(float x) {
  float E;
  const float q = 0.5f * gelu_tail_f(x, E);
  const float cdf = x >= 0.0f ? 1.0f - q : q;
  return fmaf(x * 0.39894228040143268f, E, cdf);
}
// bf16 storage: odd minimax polynomials on the clamped argument (|x| <= 4), 11-13 VALU ops and no transcendental.
//   gelu:  x * (0.5 + x P13(x^2))   max abs error 3.3e-4   |   gelu':  0.5 + x Q15(x^2)   max abs error 3.1e-4
// (a bf16 result carries 2^-9 ~ 2e-3 relative rounding; fp32 storage keeps the 3e-7 erf forms above).  The activation
// epilogues are VALU-bound — 4.9 G GELU and 4.9 G GELU' evaluations per B=32 update (conv0, conv 1-6, 12 x fc1).
__device__ __forceinline__ float gelu_poly_f(float x) {
  const float xc = fminf(fmaxf(x, -4.0f), 4.0f), t = xc * xc;
  float p = fmaf(2.6867663649454698e-08f, t, -1.824730247790285e-06f);
  p = fmaf(p, t, 5.28495256730821e-05f);
  p = fmaf(p, t, -0.0008661365136504173f);
  p = fmaf(p, t, 0.009053179994225502f);
  p = fmaf(p, t, -0.06526926904916763f);
  p = fmaf(p, t, 0.39845922589302063f);
  return x * fmaf(p, xc, 0.5f);
}
__device__ __forceinline__ float dgelu_poly_f(float x) {
  const float xc = fminf(fmaxf(x, -4.0f), 4.0f), t = xc * xc;
  float p = fmaf(-1.5577683143419563e-08f, t, 1.1633505891950335e-06f);
  p = fmaf(p, t, -3.7250658351695165e-05f);
  p = fmaf(p, t, 0.0006728997686877847f);
  p = fmaf(p, t, -0.0075911665335297585f);
  p = fmaf(p, t, 0.0555923730134964f);
  p = fmaf(p, t, -0.26155415177345276f);
  p = fmaf(p, t, 0.7965189218521118f);
  return fmaf(p, xc, 0.5f);
}
template <typename T> __device__ __forceinline__ float gelu_t(float x) { return sizeof(T) == 2 ? gelu_poly_f(x) : gelu_f(x); }
template <typename T> __device__ __forceinline__ float dgelu_t(float x) { return sizeof(T) == 2 ? dgelu_poly_f(x) : dgelu_f(x); }
template <typename T> __device__ __forceinline__ float act_t(float x, int act) {
  return act == CST_ACT_RELU ? fmaxf(x, 0.0f) : (act == CST_ACT_GELU ? gelu_t<T>(x) : x);
}
template <typename T> __device__ __forceinline__ float dact_t(float z, int act) {
  return act == CST_ACT_RELU ? (z > 0.0f ? 1.0f : 0.0f) : (act == CST_ACT_GELU ? dgelu_t<T>(z) : 1.0f);
}
__device__ __forceinline__ float act_f(float x, int act) {
  return act == CST_ACT_RELU ? fmaxf(x, 0.0f) : (act == CST_ACT_GELU ? gelu_f(x) : x);
}
__device__ __forceinline__ float dact_f(float z, int act) {
  return act == CST_ACT_RELU ? (z > 0.0f ? 1.0f : 0.0f) : (act == CST_ACT_GELU ? dgelu_f(z) : 1.0f);
}

// ---------------------------------------------------------------------------------------
// MFMA 32x32 step over 16 reduction indices.  A fragment: lane l holds Aop[row = l&31][k0 + 8*(l>>5) .. +8);
// B fragment: lane l holds Bop[k0 + 8*(l>>5) .. +8)[col = l&31].  C/D: col = l&31,
// row = (r&3) + 8*(r>>2) + 4*(l>>5) for r in [0,16).
// bf16: one v_mfma_f32_32x32x16_bf16.  f32: eight v_mfma_f32_32x32x2_f32 (slot j of both half-waves per step;
// the pairing of k indices differs from the bf16 instruction but the sum over all 16 is the same).
// ---------------------------------------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  bf16x8 v;
};
template <> struct Frag<float> {
  float v[8];
};
__device__ __forceinline__ void mma16(f32x16& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}
// fragment from 8 contiguous elements in LDS / registers
__device__ __forceinline__ void frag_load_contig(Frag<bf16_t>& f, const bf16_t* p) {
  f.v = *reinterpret_cast<const bf16x8*>(p);
}
__device__ __forceinline__ void frag_load_contig(Frag<float>& f, const float* p) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
  f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
  f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}
// fragment from 8 elements `stride` apart (operand whose reduction index is NOT contiguous in LDS)
__device__ __forceinline__ void frag_load_strided(Frag<bf16_t>& f, const bf16_t* p, int stride) {
  u16x8 t;
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = *reinterpret_cast<const unsigned short*>(p + j * stride);
  f.v = __builtin_bit_cast(bf16x8, t);
}
__device__ __forceinline__ void frag_load_strided(Frag<float>& f, const float* p, int stride) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = p[j * stride];
}
// Fragment whose reduction index is the SLOW index of an LDS tile laid out [k][mn] (mn contiguous, row stride `ld`):
// lane l receives tile[k][mn0 + (l&31)] for the 8 k values {ka..ka+3, kb..kb+3} (ka/kb already include this half-wave's
// offset).  bf16: two ds_read_b64_tr_b16 (gfx950 LDS transpose read; measured semantics, tools/probes/trprobe.hip:
// within a 16-lane group source lane s supplies the 8-byte chunk (row s>>2, chunk s&3) of a 4 x 16 block and lane i
// receives column i of that block, 4 consecutive rows).  f32: eight ds_read_b32 (half-wave = 32 consecutive addresses).
typedef short v4s_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void frag_load_tr(Frag<bf16_t>& f, const bf16_t* tile, int ld, int mn0, int ka, int kb, int lane) {
  const int s = lane & 15;
  const int col = mn0 + 16 * ((lane >> 4) & 1) + 4 * (s & 3);
  const bf16_t* pa = tile + (ka + (s >> 2)) * ld + col;
  const bf16_t* pb = tile + (kb + (s >> 2)) * ld + col;
  const v4s_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)pa);
  const v4s_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)pb);
  u16x8 t;
  t[0] = (unsigned short)x[0]; t[1] = (unsigned short)x[1]; t[2] = (unsigned short)x[2]; t[3] = (unsigned short)x[3];
  t[4] = (unsigned short)y[0]; t[5] = (unsigned short)y[1]; t[6] = (unsigned short)y[2]; t[7] = (unsigned short)y[3];
  f.v = __builtin_bit_cast(bf16x8, t);
}
__device__ __forceinline__ void frag_load_tr(Frag<float>& f, const float* tile, int ld, int mn0, int ka, int kb, int lane) {
  const float* p = tile + mn0 + (lane & 31);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f.v[j] = p[(ka + j) * ld];
    f.v[4 + j] = p[(kb + j) * ld];
  }
}
// ---- fragment loads from the UNPADDED, swizzled LDS images that global_load_lds produces (gemm_glds_kernel) ----
// k-major image: rows of 128 bytes, 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7); k = element index in the row
__device__ __forceinline__ void frag_load_kswz(Frag<bf16_t>& f, const bf16_t* tile, int r, int k) {
  const int c = (k >> 3) ^ ((r >> 1) & 7);
  f.v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(tile) + r * 128 + c * 16);
}
__device__ __forceinline__ void frag_load_kswz(Frag<float>& f, const float* tile, int r, int k) {
  const int sw = (r >> 1) & 7, c0 = (k >> 2) ^ sw, c1 = ((k >> 2) + 1) ^ sw;
  const char* row = reinterpret_cast<const char*>(tile) + r * 128;
  const f32x4 a = *reinterpret_cast<const f32x4*>(row + c0 * 16);
  const f32x4 b = *reinterpret_cast<const f32x4*>(row + c1 * 16);
  f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
  f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}
// mn-major image [k][W]: bf16 rows swizzled chunk ^= 4*(k&3) (pairs of 16-B chunks stay adjacent for the transpose read);
// lane l receives image[k0 + {0..3, 4..7}][mn0 + (l&31)], k0 already includes the half-wave offset.
__device__ __forceinline__ void frag_load_tr_swz(Frag<bf16_t>& f, const bf16_t* tile, int W, int mn0, int k0, int lane) {
  const int s = lane & 15;
  const int col = mn0 + 16 * ((lane >> 4) & 1) + 4 * (s & 3);  // element column this lane's 8-byte source chunk starts at
  const int ka = k0 + (s >> 2), kb = ka + 4;
  const int ca = ((col >> 3) ^ (4 * (ka & 3))) * 8 + (col & 7), cb = ((col >> 3) ^ (4 * (kb & 3))) * 8 + (col & 7);
  const v4s_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(tile + ka * W + ca));
  const v4s_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(tile + kb * W + cb));
  u16x8 t;
  t[0] = (unsigned short)x[0]; t[1] = (unsigned short)x[1]; t[2] = (unsigned short)x[2]; t[3] = (unsigned short)x[3];
  t[4] = (unsigned short)y[0]; t[5] = (unsigned short)y[1]; t[6] = (unsigned short)y[2]; t[7] = (unsigned short)y[3];
  f.v = __builtin_bit_cast(bf16x8, t);
}
__device__ __forceinline__ void frag_load_tr_swz(Frag<float>& f, const float* tile, int W, int mn0, int k0, int lane) {
  const float* p = tile + mn0 + (lane & 31);  // f32 rows are not swizzled: 32 consecutive lanes -> 32 consecutive banks
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = p[(k0 + j) * W];
}
// fragment from 8 fp32 register values (e.g. softmax probabilities)
__device__ __forceinline__ void frag_from_f32(Frag<bf16_t>& f, const float (&x)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = static_cast<__bf16>(x[j]);
}
__device__ __forceinline__ void frag_from_f32(Frag<float>& f, const float (&x)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = x[j];
}
// the accumulator row owned by register r of lane `lane`
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
