// attention.hip — flash-style fused attention for gfx950: scores never reach HBM.
//
// Replaces F.multi_head_attention_forward / the in-tree bmm-softmax-bmm branch of
// fairseq/modules/multihead_attention.py:155-187,326-361 (softmax in fp32, key padding -> -inf).
//
// Forward (per wave: 32 query rows, KV tiles of 64 keys staged through LDS, shared by the 4 waves):
//   S^T[key][q] = K Q^T      (A = K fragment from LDS, B = Q fragment held in registers)
//   online softmax per query: every lane owns ONE query column (q = lane&31) -> row max / sum are
//   lane-local + one exchange with lane^32 (the "swapped QK^T" idiom).
//   O^T[d][q] += V^T P^T     (A = V^T gathered from the [key][d] LDS image, B = P straight from the
//   S accumulator registers: the MFMA k-slot <-> key permutation of the C layout is applied to the
//   V gather instead of shuffling P).
// Backward = two kernels that both recompute P from (Q,K,lse):
//   dQ kernel : grid over query blocks, loops KV tiles  -> dQ
//   dKV kernel: grid over key blocks,   loops Q tiles   -> dK, dV   (no atomics, deterministic)
// plus a row-wise delta = sum_d dO*O pre-pass.
// dtype: bf16 (v_mfma_f32_32x32x16_bf16, P rounded to bf16 like the reference's fp16/bf16 path) or
// f32 (v_mfma_f32_32x32x2_f32, exact-fp32 products) from one source via Frag<T>.
#include "cst_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int NW = 4;          // waves per workgroup
constexpr int QB = 32;         // rows per wave
constexpr int KT = 64;         // keys (fwd/dQ) or queries (dKV) per LDS tile

struct AttnParams {
  int64_t B, H, Tq, Tk;
  const void *Q, *K, *V; void* O;
  int64_t q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st;
  float* lse;
  const uint8_t* kpm; int64_t kpm_stride;
  int causal;
  float scale;
  const void* dO; int64_t do_sb, do_sh, do_st;
  void *dQ, *dK, *dV;
  int64_t dq_sb, dq_sh, dq_st, dk_sb, dk_sh, dk_st, dv_sb, dv_sh, dv_st;
  float* delta;
  uint32_t drop_thr, drop_key; float drop_scale;  // attention-probability dropout (drop_thr == 0: none)
  const int32_t* kv_len;  // optional [B]: keys >= kv_len[b] are all padding -> their tiles are skipped (they contribute exact zeros)
  uint8_t* q_flags;       // optional [B][H][ceil(Tq/64)] (backward): 1 = the 64-query tile has a non-zero dO row
  // PACKED (variable-length) self-attention: seq_off != nullptr -> Q/K/V/O and all gradients are [rows, H*D] with the rows of sequence b
  // at [seq_off[b], seq_off[b+1]); its keys are the first kv_len[b] rows (the rest of the segment are query-only rows: padding frames
  // kept for their outputs).  Tq / Tk then hold the LONGEST segment: grid size and the strides of lse / delta / q_flags / the dropout
  // index space — so a packed call regenerates exactly the masks of the padded call it replaces.
  const int32_t* seq_off;
  const unsigned long long* kpm_bits; int64_t kpm_bits_stride;  // optional: key_padding_mask as one 64-bit word per 64-key tile (bit k: key 64 j + k masked)
};

// per-sequence view of the problem: row offset, number of query rows, number of keys
struct SeqView { int64_t row0; int tq, tk; };
__device__ __forceinline__ SeqView seq_view(const AttnParams& p, int64_t b) {
  SeqView v;
  if (p.seq_off) {
    v.row0 = p.seq_off[b];
    v.tq = p.seq_off[b + 1] - p.seq_off[b];
    v.tk = p.kv_len ? (p.kv_len[b] < v.tq ? p.kv_len[b] : v.tq) : v.tq;
    if (v.tk < 0) v.tk = 0;
  } else {
    v.row0 = 0;
    v.tq = (int)p.Tq;
    v.tk = (int)p.Tk;
  }
  return v;
}

// last query (exclusive, multiple of 64) whose tile carries a non-zero upstream gradient; wave 0 scans the flags
__device__ __forceinline__ int live_query_end(const AttnParams& p, int64_t b, int64_t h, int* sh, int tid, int tq) {
  const int nqt = (int)((p.Tq + 63) / 64);
  if (!p.q_flags) return tq;
  if (tid < 64) {
    const uint8_t* f = p.q_flags + (b * p.H + h) * nqt;
    int last = -1;
    for (int base = 0; base < nqt; base += 64) {
      const int idx = base + tid;
      const unsigned long long m = __ballot(idx < nqt && f[idx < nqt ? idx : 0] != 0);
      if (m) last = base + 63 - __clzll(m);
    }
    if (tid == 0) *sh = (last + 1) * 64;
  }
  __syncthreads();
  const int e = *sh;
  return e < tq ? e : tq;
}

// the row (within a 32-row MFMA tile) that k-slot j of half-wave `hi` holds for the 16-row step t
__device__ __forceinline__ int slot_row(int t, int j, int hi) { return 16 * t + (j & 3) + 8 * (j >> 2) + 4 * hi; }

// Register-staged tile prefetch: a [KT][D] tile (row stride `st` in global) is fetched into NV 16-byte vectors per thread
// (issued BEFORE the MFMA work of the current tile, T14 in the CDNA guide) and written to an LDS stage afterwards.
// Rows beyond `nrows` are clamped to the last valid row: they only ever meet probabilities / gradients that are exactly
// zero (masked), so no zero-fill pass is needed, but they must be finite.
template <typename T, int D>
struct TileRegs {
  static constexpr int VEC = DT<T>::VEC, VPR = D / VEC, NV = (KT * VPR + NW * 64 - 1) / (NW * 64);
  u32x4 r[NV];
  __device__ __forceinline__ void fetch(const T* g, int64_t st, int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + NW * 64 * i;
      int row = v / VPR;
      row = row < nrows ? row : nrows - 1;
      if (v < KT * VPR) r[i] = *reinterpret_cast<const u32x4*>(g + (int64_t)row * st + (v % VPR) * VEC);
    }
  }
  __device__ __forceinline__ void stage(T* lds, int tid) const {
    constexpr int LD = D + VEC;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + NW * 64 * i;
      if (v < KT * VPR) *reinterpret_cast<u32x4*>(lds + (v / VPR) * LD + (v % VPR) * VEC) = r[i];
    }
  }
};

// registers: D/16 fragments of one row (row = lane&31) of a [T][D] matrix in global memory
template <typename T, int D>
__device__ __forceinline__ void load_row_frags(Frag<T> (&f)[D / 16], const T* g, int64_t st, int row, int nrows, int lane) {
  const int lk = 8 * (lane >> 5);
#pragma unroll
  for (int kk = 0; kk < D / 16; ++kk) {
    if (row < nrows) frag_load_contig(f[kk], g + (int64_t)row * st + kk * 16 + lk);
    else {
      float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      frag_from_f32(f[kk], z);
    }
  }
}

// store acc (rows = d = d0 + acc_row(r), col = this lane's token) as 4-element runs along d
template <typename T>
__device__ __forceinline__ void store_dcol(T* g, const f32x16& acc, int d0, int lane, float mul) {
  const int hi = lane >> 5;
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    T* p = g + d0 + 8 * gq + 4 * hi;
#pragma unroll
    for (int e = 0; e < 4; ++e) DT<T>::st(p + e, acc[gq * 4 + e] * mul);
  }
}

// ---------------------------------------------------------------------------------------------
template <typename T, int D, bool DROP>
__global__ __launch_bounds__(NW * 64, (sizeof(T) == 2 ? 3 : 2)) void attn_fwd_kernel(AttnParams p) {
  constexpr int VEC = DT<T>::VEC, LD = D + VEC, TILE = KT * LD;
  // dynamic LDS: 2 stages x {K tile, V tile} + 2 x KT mask bytes; stage addresses are always smem + stage * 2*TILE
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  unsigned long long* sMask = reinterpret_cast<unsigned long long*>(smem + 4 * TILE);  // per stage: bit k = key k masked
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t b = blockIdx.z, h = blockIdx.y;
  const int q_blk0 = blockIdx.x * (NW * QB);
  const int q = q_blk0 + wave * QB + (lane & 31);
  const SeqView sv = seq_view(p, b);
  if (q_blk0 >= sv.tq) return;  // packed: this sequence is shorter than the longest one (workgroup-uniform)
  const T* Qg = (const T*)p.Q + (p.seq_off ? sv.row0 * p.q_st : b * p.q_sb) + h * p.q_sh;
  const T* Kg = (const T*)p.K + (p.seq_off ? sv.row0 * p.k_st : b * p.k_sb) + h * p.k_sh;
  const T* Vg = (const T*)p.V + (p.seq_off ? sv.row0 * p.v_st : b * p.v_sb) + h * p.v_sh;
  const int cshift = sv.tk - sv.tq;  // causal: key j visible to query i iff j <= i + cshift
  const float c2 = p.scale * 1.4426950408889634f;  // softmax in the exp2 domain: p = 2^(s*c2 - m)
  // attention dropout (modules/multihead_attention.py:359): element (b,h,q,k) of the probability tensor has index
  // ((b*H + h)*Tq + q) * Tkp + k with Tkp = Tk rounded up to even, so one mask word serves keys (2j, 2j+1) of a query
  const uint32_t dkey2 = cst_drop_key2(p.drop_key);
  const int thr_s = (int)p.drop_thr - 32768;  // the mask halves are compared as signed 16-bit numbers (cst_common.h)
  const uint32_t rho = (uint32_t)((b * p.H + h) * p.Tq + q);
  cst_i32x4 bsig = {0, 0, 0, 0};  // this lane's query signature (cst_common.h: the mask comes out of an i8 MFMA per 32 x 32 block)
  if (DROP) bsig = cst_asig_row_frag(p.drop_key, dkey2, rho, hi);

  Frag<T> fq[D / 16];
  load_row_frags<T, D>(fq, Qg, p.q_st, q, sv.tq, lane);

  f32x16 o[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.0f;
  float m_run = -1.0e30f, l_run = 0.0f;

  int64_t kend = sv.tk;
  if (!p.seq_off && p.kv_len && p.kv_len[b] < kend) kend = p.kv_len[b] > 0 ? p.kv_len[b] : 0;  // trailing padding: P = exp2(-inf) = 0, alpha = 1 — skipping is exact
  if (p.causal) {
    const int64_t last = (int64_t)q_blk0 + NW * QB - 1 + cshift + 1;
    if (last < kend) kend = last;
  }
  TileRegs<T, D> rk, rv;
  uint8_t rmask = 0;
  auto prefetch = [&](int64_t j0) {
    const int nk = (int)((sv.tk - j0 < KT) ? (sv.tk - j0) : KT);
    rk.fetch(Kg + j0 * p.k_st, p.k_st, nk, tid);
    rv.fetch(Vg + j0 * p.v_st, p.v_st, nk, tid);
    if (tid < KT) rmask = (tid >= nk) ? 1 : (p.kpm ? p.kpm[b * p.kpm_stride + j0 + tid] : 0);
  };
  auto stage = [&](int st) {
    rk.stage(smem + st * 2 * TILE, tid);
    rv.stage(smem + st * 2 * TILE + TILE, tid);
    if (wave == 0) {  // KT == 64: wave 0 holds one mask byte per key
      const unsigned long long bits = __ballot(rmask != 0);
      if (lane == 0) sMask[st] = bits;
    }
  };
  if (kend > 0) { prefetch(0); stage(0); }
  __syncthreads();
  int cur = 0;
  const int qlo = q_blk0 + wave * QB;
  for (int64_t j0 = 0; j0 < kend; j0 += KT) {
    const bool more = j0 + KT < kend;
    if (more) prefetch(j0 + KT);  // global loads in flight under this tile's MFMAs
    const T* sK = smem + cur * 2 * TILE;
    const T* sV = sK + TILE;
    const unsigned long long mb = sMask[cur];

    f32x16 s[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[ks][r] = 0.0f;
#pragma unroll
      for (int kk = 0; kk < D / 16; ++kk) {
        Frag<T> fk;
        frag_load_contig(fk, sK + (ks * 32 + (lane & 31)) * LD + kk * 16 + 8 * hi);
        mma16(s[ks], fk, fq[kk]);
      }
    }
    // masking only on tiles that contain a padded key or cross the causal diagonal (wave-uniform test)
    if (mb != 0ull || (p.causal && ((int)j0 + KT - 1 > qlo + cshift))) {
      const unsigned long long mbl = mb >> (4 * hi);
      const int lim = p.causal ? q + cshift - (int)j0 : 0x7fffffff;  // keys with tile index > lim are in the future
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kb = ks * 32 + (r & 3) + 8 * (r >> 2);  // + 4*hi = key index inside the tile
          const bool masked = ((mbl >> kb) & 1ull) || (kb + 4 * hi > lim);
          s[ks][r] = masked ? -INFINITY : s[ks][r];
        }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[ks][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * c2;  // c2 > 0: max commutes with the scale; -inf stays -inf
    // m_run starts at a large negative FINITE value: a tile whose keys are all masked gives m_new = m_run, alpha = 1 and
    // e = exp2(-inf) = 0 without a special case (a -inf running max would need one to avoid inf - inf)
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    {
      float ls0 = 0.0f, ls1 = 0.0f;  // two partial sums: the adds pair up into v_pk_add_f32
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float e0 = __builtin_amdgcn_exp2f(fmaf(s[ks][r], c2, -m_new));
          const float e1 = __builtin_amdgcn_exp2f(fmaf(s[ks][r + 1], c2, -m_new));
          s[ks][r] = e0;
          s[ks][r + 1] = e1;
          ls0 += e0;
          ls1 += e1;
        }
      l_run = l_run * alpha + (ls0 + ls1);
      m_run = m_new;
      if (DROP) {  // dropped probabilities leave the PV product; the row sum (normaliser) keeps them
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const cst_i32x16 dm = cst_asig_block(cst_asig_key_frag(p.drop_key, dkey2, (uint32_t)((int)j0 + ks * 32 + (lane & 31)), hi), bsig);
#pragma unroll
          for (int r = 0; r < 16; ++r) s[ks][r] = cst_adrop_keep(dm[r], thr_s) ? s[ks][r] : 0.0f;
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    // O^T += V^T P^T
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        Frag<T> fp;
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = s[ks][8 * t + j];
        frag_from_f32(fp, pv);
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
          Frag<T> fv;
          frag_load_tr(fv, sV, LD, dt * 32, ks * 32 + 16 * t + 4 * hi, ks * 32 + 16 * t + 4 * hi + 8, lane);
          mma16(o[dt], fv, fp);
        }
      }
    if (more) stage(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (q < sv.tq) {
    const float inv = l_tot > 0.0f ? p.drop_scale / l_tot : 0.0f;
    T* Og = (T*)p.O + (p.seq_off ? sv.row0 * p.o_st : b * p.o_sb) + h * p.o_sh + (int64_t)q * p.o_st;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) store_dcol<T>(Og, o[dt], dt * 32, lane, inv);
    if (hi == 0) p.lse[(b * p.H + h) * p.Tq + q] = l_tot > 0.0f ? m_run * 0.6931471805599453f + __logf(l_tot) : -INFINITY;
  }
}

// delta[b,h,q] = sum_d dO*O; one wave per (b, h, 64-query tile).  It also records whether the tile has any non-zero dO row: in a
// length-sorted, right-padded batch the gradient of padded frames is exactly zero wherever nothing downstream reads them (every
// wav2vec2 layer of the s2t model beyond the subsampler's receptive field), and a tile of zero dO rows contributes exact zeros
// to dQ, dK and dV (dP = dO V^T = 0, delta = 0 -> dS = P (0 - 0) = 0): the backward kernels stop at the last live tile.
template <typename T, int D>
__global__ __launch_bounds__(64) void attn_delta_kernel(AttnParams p) {
  // D / 8 lanes per query row (16 bytes each: a load instruction fetches whole 128-byte head rows of 64 / (D / 8) queries), the dot
  // product is finished by a butterfly over those lanes
  constexpr int LPR = D / 8, RPI = 64 / LPR;
  const int64_t b = blockIdx.z, h = blockIdx.y;
  const int qt = blockIdx.x, lane = threadIdx.x;
  const int rl = lane / LPR, c8 = (lane % LPR) * 8;
  bool nz = false;
  const SeqView sv = seq_view(p, b);
  const T* obase = (const T*)p.O + (p.seq_off ? sv.row0 * p.o_st : b * p.o_sb) + h * p.o_sh + c8;
  const T* gbase = (const T*)p.dO + (p.seq_off ? sv.row0 * p.do_st : b * p.do_sb) + h * p.do_sh + c8;
#pragma unroll
  for (int it = 0; it < 64 / RPI; ++it) {
    const int64_t q = (int64_t)qt * 64 + it * RPI + rl;
    float acc = 0.0f;
    if (q < sv.tq) {
      float a[8], c[8];
      load8(obase + q * p.o_st, a);
      load8(gbase + q * p.do_st, c);
#pragma unroll
      for (int e = 0; e < 8; ++e) { acc += a[e] * c[e]; nz = nz || c[e] != 0.0f; }
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (q < sv.tq && (lane % LPR) == 0) p.delta[(b * p.H + h) * p.Tq + q] = acc;
  }
  const unsigned long long m = __ballot(nz);
  if (p.q_flags && lane == 0) p.q_flags[(b * p.H + h) * ((p.Tq + 63) / 64) + qt] = m != 0ull ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// dQ: per wave 32 queries (lane-local query column), loop over KV tiles.
template <typename T, int D, bool DROP>
__global__ __launch_bounds__(NW * 64, (sizeof(T) == 2 ? 3 : 2)) void attn_bwd_dq_kernel(AttnParams p) {
  constexpr int VEC = DT<T>::VEC, LD = D + VEC, TILE = KT * LD;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  unsigned long long* sMask = reinterpret_cast<unsigned long long*>(smem + 4 * TILE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t b = blockIdx.z, h = blockIdx.y;
  const int q_blk0 = blockIdx.x * (NW * QB);
  const int q = q_blk0 + wave * QB + (lane & 31);
  const SeqView sv = seq_view(p, b);
  if (q_blk0 >= sv.tq) return;  // packed: beyond this sequence (workgroup-uniform, before any barrier)
  const T* Qg = (const T*)p.Q + (p.seq_off ? sv.row0 * p.q_st : b * p.q_sb) + h * p.q_sh;
  const T* Kg = (const T*)p.K + (p.seq_off ? sv.row0 * p.k_st : b * p.k_sb) + h * p.k_sh;
  const T* Vg = (const T*)p.V + (p.seq_off ? sv.row0 * p.v_st : b * p.v_sb) + h * p.v_sh;
  const T* dOg = (const T*)p.dO + (p.seq_off ? sv.row0 * p.do_st : b * p.do_sb) + h * p.do_sh;
  T* const dQb = (T*)p.dQ + (p.seq_off ? sv.row0 * p.dq_st : b * p.dq_sb) + h * p.dq_sh;
  const int cshift = sv.tk - sv.tq;
  const float c2 = p.scale * 1.4426950408889634f;
  const uint32_t dkey2 = cst_drop_key2(p.drop_key);
  const int thr_s = (int)p.drop_thr - 32768;  // the mask halves are compared as signed 16-bit numbers (cst_common.h)
  const uint32_t rho = (uint32_t)((b * p.H + h) * p.Tq + q);
  cst_i32x4 bsig = {0, 0, 0, 0};
  if (DROP) bsig = cst_asig_row_frag(p.drop_key, dkey2, rho, hi);

  {
    int* sh_qend = reinterpret_cast<int*>(sMask + 2);
    const int qend = live_query_end(p, b, h, sh_qend, tid, sv.tq);
    if (q_blk0 >= qend) {  // every query of this workgroup has a zero upstream gradient: dQ = 0 (workgroup-uniform exit)
      if (q < sv.tq) {
        T* g = dQb + (int64_t)q * p.dq_st;
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.0f;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) store_dcol<T>(g, z, dt * 32, lane, 1.0f);
      }
      return;
    }
  }
  Frag<T> fq[D / 16], fdo[D / 16];
  load_row_frags<T, D>(fq, Qg, p.q_st, q, sv.tq, lane);
  load_row_frags<T, D>(fdo, dOg, p.do_st, q, sv.tq, lane);
  // log2 domain.  Rows beyond Tq and fully masked rows (lse = -inf) get +inf: exp2(s - inf) = 0 without a per-element select
  float lse = INFINITY, dlt = 0.0f;
  if (q < sv.tq) {
    const float l = p.lse[(b * p.H + h) * p.Tq + q];
    lse = l == -INFINITY ? INFINITY : l * 1.4426950408889634f;
    dlt = p.delta[(b * p.H + h) * p.Tq + q];
  }
  f32x16 dq[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.0f;

  int64_t kend = sv.tk;
  if (!p.seq_off && p.kv_len && p.kv_len[b] < kend) kend = p.kv_len[b] > 0 ? p.kv_len[b] : 0;
  if (p.causal) {
    const int64_t last = (int64_t)q_blk0 + NW * QB + cshift;
    if (last < kend) kend = last;
  }
  TileRegs<T, D> rk, rv;
  uint8_t rmask = 0;
  auto prefetch = [&](int64_t j0) {
    const int nk = (int)((sv.tk - j0 < KT) ? (sv.tk - j0) : KT);
    rk.fetch(Kg + j0 * p.k_st, p.k_st, nk, tid);
    rv.fetch(Vg + j0 * p.v_st, p.v_st, nk, tid);
    if (tid < KT) rmask = (tid >= nk) ? 1 : (p.kpm ? p.kpm[b * p.kpm_stride + j0 + tid] : 0);
  };
  auto stage = [&](int st) {
    rk.stage(smem + st * 2 * TILE, tid);
    rv.stage(smem + st * 2 * TILE + TILE, tid);
    if (wave == 0) {
      const unsigned long long bits = __ballot(rmask != 0);
      if (lane == 0) sMask[st] = bits;
    }
  };
  if (kend > 0) { prefetch(0); stage(0); }
  __syncthreads();
  int cur = 0;
  for (int64_t j0 = 0; j0 < kend; j0 += KT) {
    const bool more = j0 + KT < kend;
    if (more) prefetch(j0 + KT);
    const T* sK = smem + cur * 2 * TILE;
    const T* sV = sK + TILE;
    const unsigned long long mb = sMask[cur];
    const bool need_mask = mb != 0ull || (p.causal && ((int)j0 + KT - 1 > q_blk0 + wave * QB + cshift));
    const unsigned long long mbl = mb >> (4 * hi);
    const int lim = p.causal ? q + cshift - (int)j0 : 0x7fffffff;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
      for (int kk = 0; kk < D / 16; ++kk) {
        Frag<T> fk, fv;
        frag_load_contig(fk, sK + (ks * 32 + (lane & 31)) * LD + kk * 16 + 8 * hi);
        frag_load_contig(fv, sV + (ks * 32 + (lane & 31)) * LD + kk * 16 + 8 * hi);
        mma16(s, fk, fq[kk]);     // S^T[key][q]
        mma16(dp, fv, fdo[kk]);   // dP^T[key][q] = V dO^T
      }
      if (DROP) {  // dP = (dO V^T) * keep / (1 - p): the same mask as the forward pass
        const cst_i32x16 dm = cst_asig_block(cst_asig_key_frag(p.drop_key, dkey2, (uint32_t)((int)j0 + ks * 32 + (lane & 31)), hi), bsig);
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = cst_adrop_keep(dm[r], thr_s) ? dp[r] * p.drop_scale : 0.0f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pr = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lse));
        if (need_mask) {
          const int kb = ks * 32 + (r & 3) + 8 * (r >> 2);
          pr = (((mbl >> kb) & 1ull) || (kb + 4 * hi > lim)) ? 0.0f : pr;
        }
        s[r] = pr * (dp[r] - dlt);  // dS^T (scale folded in at the end)
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        Frag<T> fds;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = s[8 * t + j];
        frag_from_f32(fds, x);
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
          Frag<T> fkt;
          frag_load_tr(fkt, sK, LD, dt * 32, ks * 32 + 16 * t + 4 * hi, ks * 32 + 16 * t + 4 * hi + 8, lane);
          mma16(dq[dt], fkt, fds);  // dQ^T[d][q] += K^T dS^T
        }
      }
    }
    if (more) stage(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  if (q < sv.tq) {
    T* g = dQb + (int64_t)q * p.dq_st;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) store_dcol<T>(g, dq[dt], dt * 32, lane, p.scale);
  }
}

// dK,dV: per wave 32 keys (lane-local key column), loop over Q tiles of 64 queries.
template <typename T, int D, bool DROP>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dkv_kernel(AttnParams p) {
  constexpr int VEC = DT<T>::VEC, LD = D + VEC, TILE = KT * LD;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  float* sStat = reinterpret_cast<float*>(smem + 4 * TILE);  // [2 stages][lse KT | delta KT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t b = blockIdx.z, h = blockIdx.y;
  const int k_blk0 = blockIdx.x * (NW * QB);
  const int key = k_blk0 + wave * QB + (lane & 31);
  const SeqView sv = seq_view(p, b);
  const int krows = p.seq_off ? sv.tq : sv.tk;  // rows of dK / dV this sequence owns (packed: the query-only rows get zeros)
  if (k_blk0 >= krows) return;  // packed: beyond this sequence (workgroup-uniform, before any barrier)
  const T* Qg = (const T*)p.Q + (p.seq_off ? sv.row0 * p.q_st : b * p.q_sb) + h * p.q_sh;
  const T* Kg = (const T*)p.K + (p.seq_off ? sv.row0 * p.k_st : b * p.k_sb) + h * p.k_sh;
  const T* Vg = (const T*)p.V + (p.seq_off ? sv.row0 * p.v_st : b * p.v_sb) + h * p.v_sh;
  const T* dOg = (const T*)p.dO + (p.seq_off ? sv.row0 * p.do_st : b * p.do_sb) + h * p.do_sh;
  T* const dKb = (T*)p.dK + (p.seq_off ? sv.row0 * p.dk_st : b * p.dk_sb) + h * p.dk_sh;
  T* const dVb = (T*)p.dV + (p.seq_off ? sv.row0 * p.dv_st : b * p.dv_sb) + h * p.dv_sh;
  const int cshift = sv.tk - sv.tq;
  const float c2 = p.scale * 1.4426950408889634f;
  // here a lane owns one key and its registers run over queries, so every element needs its own mask word (the word of the
  // key pair (key & ~1, key | 1) in that query's row); this lane's half of the word is `dsh`
  const uint32_t dkey2 = cst_drop_key2(p.drop_key);
  const int thr_s = (int)p.drop_thr - 32768;  // the mask halves are compared as signed 16-bit numbers (cst_common.h)
  const uint32_t rho0 = (uint32_t)((b * p.H + h) * p.Tq);
  cst_i32x4 ksig = {0, 0, 0, 0};  // this lane's key signature
  if (DROP) ksig = cst_asig_key_frag(p.drop_key, dkey2, (uint32_t)key, hi);

  if (p.kv_len && k_blk0 >= p.kv_len[b]) {  // every key of this block is padding: dK = dV = 0 (workgroup-uniform exit)
    if (key < krows) {
      T* gk = dKb + (int64_t)key * p.dk_st;
      T* gv = dVb + (int64_t)key * p.dv_st;
      f32x16 z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.0f;
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt) {
        store_dcol<T>(gk, z, dt * 32, lane, 1.0f);
        store_dcol<T>(gv, z, dt * 32, lane, 1.0f);
      }
    }
    return;
  }
  Frag<T> fk[D / 16], fv[D / 16];
  load_row_frags<T, D>(fk, Kg, p.k_st, key, sv.tk, lane);
  load_row_frags<T, D>(fv, Vg, p.v_st, key, sv.tk, lane);
  const bool key_masked = key >= sv.tk || (p.kpm && p.kpm[b * p.kpm_stride + key]);

  f32x16 dk[D / 32], dv[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.0f; dv[dt][r] = 0.0f; }

  int64_t qstart = 0;
  if (p.causal) {
    qstart = (int64_t)k_blk0 - cshift;  // first query that can see the first key of this block
    if (qstart < 0) qstart = 0;
    qstart = (qstart / KT) * KT;
  }
  TileRegs<T, D> rq, rdo;
  float rl = -INFINITY, rd = 0.0f;
  auto prefetch = [&](int64_t i0) {
    const int nq = (int)((sv.tq - i0 < KT) ? (sv.tq - i0) : KT);
    rq.fetch(Qg + i0 * p.q_st, p.q_st, nq, tid);
    rdo.fetch(dOg + i0 * p.do_st, p.do_st, nq, tid);
    if (tid < KT) {
      // log2 domain; out-of-range and fully masked rows (lse = -inf) are staged as +inf: exp2(s - inf) = 0, no per-element select
      const float l = tid < nq ? p.lse[(b * p.H + h) * p.Tq + i0 + tid] : -INFINITY;
      rl = l == -INFINITY ? INFINITY : l * 1.4426950408889634f;
      rd = tid < nq ? p.delta[(b * p.H + h) * p.Tq + i0 + tid] : 0.0f;
    }
  };
  auto stage = [&](int st) {
    rq.stage(smem + st * 2 * TILE, tid);
    rdo.stage(smem + st * 2 * TILE + TILE, tid);
    if (tid < KT) { sStat[st * 2 * KT + tid] = rl; sStat[st * 2 * KT + KT + tid] = rd; }
  };
  const int64_t qlim = live_query_end(p, b, h, reinterpret_cast<int*>(sStat + 4 * KT), tid, sv.tq);  // queries beyond it have dO = 0
  if (qstart < qlim) { prefetch(qstart); stage(0); }
  __syncthreads();
  int cur = 0;
  for (int64_t i0 = qstart; i0 < qlim; i0 += KT) {
    const bool more = i0 + KT < qlim;
    if (more) prefetch(i0 + KT);
    const T* sQ = smem + cur * 2 * TILE;
    const T* sdO = sQ + TILE;
    const float* sL = sStat + cur * 2 * KT;
    const float* sD = sL + KT;
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
      for (int kk = 0; kk < D / 16; ++kk) {
        Frag<T> fqr, fdor;
        frag_load_contig(fqr, sQ + (qs * 32 + (lane & 31)) * LD + kk * 16 + 8 * hi);
        frag_load_contig(fdor, sdO + (qs * 32 + (lane & 31)) * LD + kk * 16 + 8 * hi);
        mma16(s, fqr, fk[kk]);     // S[q][key]
        mma16(dp, fdor, fv[kk]);   // dP[q][key] = dO V^T
      }
      f32x16 pr;
      const bool diag = p.causal && (k_blk0 + wave * QB + QB - 1 > (int)i0 + qs * 32 + cshift);  // wave-uniform: tile touches the future
      // dropout: rows of the mask block = the 32 queries of this step (their signatures are hashed here, one row per lane)
      uint32_t keepbits = 0;  // bit r: element r of this tile column is kept
      if (DROP) {
        const cst_i32x16 dm = cst_asig_block(cst_asig_row_frag(p.drop_key, dkey2, rho0 + (uint32_t)((int)i0 + qs * 32 + (lane & 31)), hi), ksig);
#pragma unroll
        for (int r = 0; r < 16; ++r) keepbits |= (cst_adrop_keep(dm[r], thr_s) ? 1u : 0u) << r;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qr = qs * 32 + acc_row(r, lane);
        const float l = sL[qr];
        float e = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -l));  // l = +inf for dead / out-of-range rows -> 0
        // (a masked key's lane computes unused values: they only reach that key's own dK / dV column, zeroed at the store)
        if (diag) e = (key > (int)i0 + qr + cshift) ? 0.0f : e;
        float dpr = dp[r];
        if (DROP) {
          const bool keep = (keepbits >> r) & 1u;
          dpr = keep ? dpr * p.drop_scale : 0.0f;
          pr[r] = keep ? e * p.drop_scale : 0.0f;   // dV uses the dropped, rescaled probabilities
        } else {
          pr[r] = e;
        }
        s[r] = e * (dpr - sD[qr]);  // dS[q][key]
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        Frag<T> fp, fds;
        float x[8], y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[j] = pr[8 * t + j]; y[j] = s[8 * t + j]; }
        frag_from_f32(fp, x);
        frag_from_f32(fds, y);
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
          Frag<T> fdot, fqt;
          frag_load_tr(fdot, sdO, LD, dt * 32, qs * 32 + 16 * t + 4 * hi, qs * 32 + 16 * t + 4 * hi + 8, lane);
          frag_load_tr(fqt, sQ, LD, dt * 32, qs * 32 + 16 * t + 4 * hi, qs * 32 + 16 * t + 4 * hi + 8, lane);
          mma16(dv[dt], fdot, fp);   // dV^T[d][key] += dO^T P
          mma16(dk[dt], fqt, fds);   // dK^T[d][key] += Q^T dS
        }
      }
    }
    if (more) stage(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  if (key_masked) {
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.0f; dv[dt][r] = 0.0f; }
  }
  if (key < krows) {
    T* gk = dKb + (int64_t)key * p.dk_st;
    T* gv = dVb + (int64_t)key * p.dv_st;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) {
      store_dcol<T>(gk, dk[dt], dt * 32, lane, p.scale);
      store_dcol<T>(gv, dv[dt], dt * 32, lane, 1.0f);
    }
  }
}

#include "attention_fast.inc"

int fill_params(const cst_attn_desc* d, AttnParams& p, bool bwd) {
  CST_REQUIRE(d, "cst_attn: null descriptor");
  CST_REQUIRE(d->dtype == CST_F32 || d->dtype == CST_BF16, "cst_attn: bad dtype %d", d->dtype);
  CST_REQUIRE(d->D == 32 || d->D == 64, "cst_attn: head dim %lld not in {32,64}", (long long)d->D);
  CST_REQUIRE(d->B > 0 && d->H > 0 && d->Tq > 0 && d->Tk > 0, "cst_attn: empty problem");
  CST_REQUIRE(d->Q && d->K && d->V && d->O && d->lse, "cst_attn: null tensor");
  CST_REQUIRE(d->H < 65536 && d->B < 65536, "cst_attn: B/H too large for the grid");
  const int vec = d->dtype == CST_BF16 ? 8 : 4;
  const int64_t strides[] = {d->q_sb, d->q_sh, d->q_st, d->k_sb, d->k_sh, d->k_st, d->v_sb, d->v_sh, d->v_st};
  for (int64_t s : strides) CST_REQUIRE(s % vec == 0, "cst_attn: strides must be multiples of %d elements", vec);
  CST_REQUIRE(((uintptr_t)d->Q % 16 == 0) && ((uintptr_t)d->K % 16 == 0) && ((uintptr_t)d->V % 16 == 0), "cst_attn: Q/K/V must be 16-byte aligned");
  p.B = d->B; p.H = d->H; p.Tq = d->Tq; p.Tk = d->Tk;
  p.Q = d->Q; p.K = d->K; p.V = d->V; p.O = d->O;
  p.q_sb = d->q_sb; p.q_sh = d->q_sh; p.q_st = d->q_st;
  p.k_sb = d->k_sb; p.k_sh = d->k_sh; p.k_st = d->k_st;
  p.v_sb = d->v_sb; p.v_sh = d->v_sh; p.v_st = d->v_st;
  p.o_sb = d->o_sb; p.o_sh = d->o_sh; p.o_st = d->o_st;
  p.lse = d->lse; p.kpm = d->key_padding_mask; p.kpm_stride = d->kpm_stride;
  p.causal = d->causal; p.scale = d->scale;
  CST_REQUIRE(d->drop_p >= 0.0f && d->drop_p < 1.0f, "cst_attn: drop_p must be in [0, 1)");
  CST_REQUIRE(d->drop_p == 0.0f || (double)d->B * d->H * d->Tq < 4294967296.0, "cst_attn: dropout row space exceeds 2^32 rows");
  p.drop_thr = d->drop_p > 0.0f ? cst_drop_thr16(d->drop_p) : 0u;
  p.drop_key = d->drop_key;
  p.drop_scale = d->drop_p > 0.0f ? 1.0f / (1.0f - d->drop_p) : 1.0f;
  p.dO = d->dO; p.do_sb = d->do_sb; p.do_sh = d->do_sh; p.do_st = d->do_st;
  p.dQ = d->dQ; p.dK = d->dK; p.dV = d->dV;
  p.dq_sb = d->dq_sb; p.dq_sh = d->dq_sh; p.dq_st = d->dq_st;
  p.dk_sb = d->dk_sb; p.dk_sh = d->dk_sh; p.dk_st = d->dk_st;
  p.dv_sb = d->dv_sb; p.dv_sh = d->dv_sh; p.dv_st = d->dv_st;
  p.delta = d->delta;
  p.kv_len = d->kv_len;
  p.seq_off = d->seq_offsets;
  p.kpm_bits = (const unsigned long long*)d->kpm_bits;
  p.kpm_bits_stride = (d->Tk + 63) / 64;
  if (p.seq_off) {
    CST_REQUIRE(d->Tq == d->Tk && d->key_padding_mask == nullptr, "cst_attn: packed sequences are self-attention (Tq == Tk = longest segment) without a key padding mask (kv_len gives the keys per sequence)");
    CST_REQUIRE(d->q_st == d->k_st && d->q_st == d->v_st, "cst_attn: packed Q/K/V must share one row stride");
  }
  p.q_flags = bwd ? d->q_flags : nullptr;
  if (bwd) {
    CST_REQUIRE(d->dO && d->dQ && d->dK && d->dV && d->delta, "cst_attn_bwd: null gradient tensor");
    CST_REQUIRE(d->do_sb % vec == 0 && d->do_sh % vec == 0 && d->do_st % vec == 0 && d->o_st % vec == 0 && d->o_sb % vec == 0 && d->o_sh % vec == 0,
                "cst_attn_bwd: dO/O strides must be multiples of %d", vec);
    CST_REQUIRE(((uintptr_t)d->dO % 16 == 0) && ((uintptr_t)d->O % 16 == 0), "cst_attn_bwd: dO/O must be 16-byte aligned");
  }
  return CST_OK;
}

template <typename T, int D>
constexpr size_t attn_lds_bytes() {
  return 4 * (size_t)KT * (D + DT<T>::VEC) * sizeof(T) + 4 * KT * sizeof(float) + 16;  // 2 stages x 2 tiles + masks / (lse, delta) + 1 int
}
void attn_set_lds(const void* fn) {  // tiles of the f32 / D=64 variants exceed the 64 KiB default (gfx950: 160 KiB per CU)
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
}

double attn_flops(const cst_attn_desc* d, double gemms) {
  double pairs = (double)d->Tq * (double)d->Tk;
  if (d->causal) pairs *= 0.5;
  return gemms * 2.0 * pairs * (double)d->D * (double)d->B * (double)d->H;
}

// the DMA-staged kernels cover the hot configuration: bf16, head dim 64, no causal mask, key masks as packed tile words (or none)
bool attn_fast_ok(const cst_attn_desc* d) {
  const bool off = getenv("CST_ATTN_GENERIC") != nullptr;  // test / A-B hook (read per call: tools toggle it in-process): the generic kernels for everything
  return !off && d->dtype == CST_BF16 && d->D == 64 && !d->causal && (d->key_padding_mask == nullptr || d->kpm_bits != nullptr) &&
         d->Tk <= 65536 && (int64_t)d->Tk * d->k_st < (1ll << 29) && (int64_t)d->Tq * d->q_st < (1ll << 29) && (int64_t)d->Tk * d->v_st < (1ll << 29) &&
         d->o_st % 8 == 0 && d->o_sh % 8 == 0 && d->o_sb % 8 == 0 && (uintptr_t)d->O % 16 == 0;
}
bool attn_fast_bwd_ok(const cst_attn_desc* d) {
  const int64_t st[] = {d->dq_sb, d->dq_sh, d->dq_st, d->dk_sb, d->dk_sh, d->dk_st, d->dv_sb, d->dv_sh, d->dv_st};
  for (int64_t x : st) if (x % 8) return false;
  return attn_fast_ok(d) && d->bwd_ws != nullptr && (uintptr_t)d->dQ % 16 == 0 && (uintptr_t)d->dK % 16 == 0 && (uintptr_t)d->dV % 16 == 0 &&
         (uintptr_t)d->bwd_ws % 16 == 0 && (int64_t)d->Tq * d->do_st < (1ll << 29);
}

}  // namespace

extern "C" int cst_attn_fwd(const cst_attn_desc* d, cst_stream stream) {
  AttnParams p;
  int rc = fill_params(d, p, false);
  if (rc != CST_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (attn_fast_ok(d)) {
    CstProfScope prof(CST_K_ATTN_FWD, s, attn_flops(d, 2.0), 0.0);
    const unsigned nblk = fa_grid(d->Tq, d->B, d->H);
    const size_t lds = FA_NSLOT * FA_SLOT + 8 * (size_t)cst_ceil_div(d->Tk, 64) + (p.drop_thr ? 4096 : 0);
    if (p.drop_thr) hipLaunchKernelGGL((fa_fwd_kernel<true>), dim3(nblk), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((fa_fwd_kernel<false>), dim3(nblk), dim3(256), lds, s, p);
    return cst_check_launch("cst_attn_fwd");
  }
  dim3 grid((unsigned)cst_ceil_div(d->Tq, NW * QB), (unsigned)d->H, (unsigned)d->B);
  CstProfScope prof(CST_K_ATTN_FWD, s, attn_flops(d, 2.0), 0.0);
#define CST_FWD1(T, DD, DR) do { const size_t lds = attn_lds_bytes<T, DD>(); attn_set_lds(reinterpret_cast<const void*>(&attn_fwd_kernel<T, DD, DR>)); \
    hipLaunchKernelGGL((attn_fwd_kernel<T, DD, DR>), grid, dim3(NW * 64), lds, s, p); } while (0)
#define CST_FWD(T, DD) do { if (p.drop_thr) CST_FWD1(T, DD, true); else CST_FWD1(T, DD, false); } while (0)
  if (d->dtype == CST_BF16) { if (d->D == 64) CST_FWD(bf16_t, 64); else CST_FWD(bf16_t, 32); }
  else { if (d->D == 64) CST_FWD(float, 64); else CST_FWD(float, 32); }
#undef CST_FWD
#undef CST_FWD1
  return cst_check_launch("cst_attn_fwd");
}

// (internal, cst_common.h) the matrix-core kernel behind cst_dec_cross_attn / cst_dec_ln_q_cross_attn for bf16 / head dim 64 / beam <= 32
// (attention_fast.inc).  Wg != NULL: q is the residual stream x [bsz * beam, K] (row stride ldx) and the LayerNorm-folded query
// projection (Wg [H * 64, K], sg, sb, eps: cst_dec_ln_linear's operands) runs inside the kernel.
int cst_fa_dec_cross(const void* q, const void* kx, const void* vx, const uint8_t* kpm, void* out, const int32_t* step, int64_t max_len,
                     int64_t bsz, int64_t beam, int64_t H, int64_t S, float scale, const void* Wg, const float* sg, const float* sb, float eps,
                     int64_t K, int64_t ldx, hipStream_t s) {
  if (Wg) {
    FdProj pj = {(const bf16_t*)Wg, sg, sb, eps, (int)K, (long long)ldx};
    return fa_dec_cross_launch(q, kx, vx, kpm, out, step, max_len, bsz, beam, H, S, scale, &pj, s);
  }
  return fa_dec_cross_launch(q, kx, vx, kpm, out, step, max_len, bsz, beam, H, S, scale, nullptr, s);
}

extern "C" int64_t cst_attn_bwd_workspace(const cst_attn_desc* d) {
  if (!d) return 0;
  // per (batch, head) and query row padded to whole 64-query tiles: -lse2 and -delta / scale (fp32) + the row's 32-byte dropout
  // signature (written by the DMA-staged dQ kernel for the dK / dV kernel behind it: attention_fast.inc)
  return (int64_t)d->B * d->H * cst_ceil_div(d->Tq, 64) * 64 * (int64_t)(2 * sizeof(float) + 32);
}

extern "C" int cst_attn_bwd(const cst_attn_desc* d, cst_stream stream) {
  AttnParams p;
  int rc = fill_params(d, p, true);
  if (rc != CST_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ATTN_BWD, s, attn_flops(d, 5.0), 0.0);
  if (attn_fast_bwd_ok(d)) {
    const int64_t Tq64 = cst_ceil_div(d->Tq, 64) * 64;
    float* ws = (float*)d->bwd_ws;
    // (no pre-pass launch: the dQ kernel computes delta / -lse2 / the live-tile flags of its own 128 queries and leaves them in `ws` /
    //  q_flags for the dK / dV kernel behind it; CST_ATTN_DELTA_PASS=1 keeps the separate launch in front for A/B runs — the dQ kernel
    //  then rewrites the same values)
    static const bool delta_pass = getenv("CST_ATTN_DELTA_PASS") != nullptr;
    if (delta_pass) hipLaunchKernelGGL(fa_delta_kernel, dim3((unsigned)(Tq64 / 64), (unsigned)d->H, (unsigned)d->B), dim3(64), 0, s, p, ws, Tq64);
    const unsigned nq = fa_grid(d->Tq, d->B, d->H), nk = fa_grid(d->Tk, d->B, d->H);
    const size_t lds_q = FA_NSLOT * FA_SLOT + 8 * (size_t)cst_ceil_div(d->Tk, 64) + (p.drop_thr ? 4096 : 0) + 16, lds_k = FA_NSLOT * (FA_SLOT + FA_STATS);
    if (p.drop_thr) {
      hipLaunchKernelGGL((fa_dq_kernel<true>), dim3(nq), dim3(256), lds_q, s, p, ws, Tq64);
      hipLaunchKernelGGL((fa_dkv_kernel<true>), dim3(nk), dim3(256), lds_k, s, p, (const float*)ws, Tq64);
    } else {
      hipLaunchKernelGGL((fa_dq_kernel<false>), dim3(nq), dim3(256), lds_q, s, p, ws, Tq64);
      hipLaunchKernelGGL((fa_dkv_kernel<false>), dim3(nk), dim3(256), lds_k, s, p, (const float*)ws, Tq64);
    }
    return cst_check_launch("cst_attn_bwd");
  }
  dim3 gq((unsigned)cst_ceil_div(d->Tq, NW * QB), (unsigned)d->H, (unsigned)d->B);
  dim3 gk((unsigned)cst_ceil_div(d->Tk, NW * QB), (unsigned)d->H, (unsigned)d->B);
#define CST_BWD1(T, DD, DR)                                                                                \
  do {                                                                                                     \
    hipLaunchKernelGGL((attn_delta_kernel<T, DD>), dim3((unsigned)cst_ceil_div(d->Tq, 64), (unsigned)d->H, (unsigned)d->B), dim3(64), 0, s, p); \
    const size_t lds = attn_lds_bytes<T, DD>();                                                              \
    attn_set_lds(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<T, DD, DR>));                             \
    attn_set_lds(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<T, DD, DR>));                            \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DD, DR>), gq, dim3(NW * 64), lds, s, p);                      \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DD, DR>), gk, dim3(NW * 64), lds, s, p);                     \
  } while (0)
#define CST_BWD(T, DD) do { if (p.drop_thr) CST_BWD1(T, DD, true); else CST_BWD1(T, DD, false); } while (0)
  if (d->dtype == CST_BF16) { if (d->D == 64) CST_BWD(bf16_t, 64); else CST_BWD(bf16_t, 32); }
  else { if (d->D == 64) CST_BWD(float, 64); else CST_BWD(float, 32); }
#undef CST_BWD
#undef CST_BWD1
  return cst_check_launch("cst_attn_bwd");
}
