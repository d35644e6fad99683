// attn_issue_probe.hip — how much of a flash-attention tile's softmax VALU work hides under its MFMAs on gfx950, per schedule.
// One synthetic "tile" per loop iteration: 32 queries (lane = query) x 64 keys, head dim 64: 8 MFMA (K Q^T) + softmax on 32 scores per
// lane + 8 MFMA (V^T P^T); K / V fragments come from a static LDS image (ds_read_b128 / ds_read_b64_tr_b16) as in the real kernel,
// there is no global traffic and no barrier.  Variants:
//   MODE 0  phased:    QK(i) ; softmax(i) ; PV(i)                                   (the round-2 kernel's order)
//   MODE 1  pipelined: one basic block per tile holding QK(i+1), exp(i) and PV(i-1), interleaved by sched_group_barrier; the running
//           max is an integer, enters through the accumulator's C operand (a persistent 16-register copy of -m) and is raised
//           lazily (rare branch at the top of the block, decided from the max taken at the end of the previous block)
//   CINIT   (MODE 0 only) same C-operand trick in the phased order
//   XV      extra integer VALU operations per score pair (stands for the dropout mask work)
// Grid = 256 CUs x W workgroups of 4 waves -> W waves per SIMD.  Output: shader cycles per tile per wave and per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 attn_issue_probe.hip -o attn_issue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16_t;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;
typedef short v4s_t __attribute__((ext_vector_type(4)));
constexpr int LD = 72, TILE = 64 * LD;

__device__ __forceinline__ bf16x8 ld_contig(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 ld_tr(const bf16_t* tile, int mn0, int ka, int kb, int lane) {
  const int s = lane & 15;
  const int col = mn0 + 16 * ((lane >> 4) & 1) + 4 * (s & 3);
  const v4s_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(tile + (ka + (s >> 2)) * LD + col));
  const v4s_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(tile + (kb + (s >> 2)) * LD + col));
  u16x8 t;
  t[0] = x[0]; t[1] = x[1]; t[2] = x[2]; t[3] = x[3]; t[4] = y[0]; t[5] = y[1]; t[6] = y[2]; t[7] = y[3];
  return __builtin_bit_cast(bf16x8, t);
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// scores of one 64-key tile; `c0` is the C operand of the first MFMA of each chain (zeros, or -m in every register)
__device__ __forceinline__ void qk(f32x16 (&s)[2], const bf16_t* sK, const bf16x8 (&fq)[4], const f32x16& c0, int lane) {
  const int hi = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    s[ks] = MFMA(ld_contig(sK + (ks * 32 + (lane & 31)) * LD + 8 * hi), fq[0], c0);
#pragma unroll
    for (int kk = 1; kk < 4; ++kk) s[ks] = MFMA(ld_contig(sK + (ks * 32 + (lane & 31)) * LD + kk * 16 + 8 * hi), fq[kk], s[ks]);
  }
}
__device__ __forceinline__ float tile_max(const f32x16 (&s)[2]) {
  float mt = -INFINITY;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[ks][r]);
  return mt;
}
template <int XV>
__device__ __forceinline__ void drop(f32x16 (&s)[2], unsigned key) {
  if (XV > 0) {  // stand-in for the dropout mask: XV dependent integer operations per score pair-word, then two selects
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        unsigned x = key + ks * 16 + r;
#pragma unroll
        for (int j = 0; j < XV; ++j) x = (j & 1) ? (x ^ (x >> 13)) : __umul24(x, 0x9E3779U) + key;
        s[ks][r] = (x & 0xffffU) >= 6554U ? s[ks][r] : 0.0f;
        s[ks][r + 1] = (x >> 16) >= 6554U ? s[ks][r + 1] : 0.0f;
      }
  }
}
__device__ __forceinline__ void to_frags(const f32x16 (&s)[2], bf16x8 (&p)[4]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) p[ks * 2 + t][j] = static_cast<__bf16>(s[ks][8 * t + j]);
}
// the round-2 softmax: eager running max, fma + exp2 per score, O rescaled every tile
template <int XV>
__device__ __forceinline__ void softmax_eager(f32x16 (&s)[2], bf16x8 (&p)[4], float& m_run, float& l_run, f32x16 (&o)[2], float c2, unsigned key) {
  float mt = tile_max(s);
  mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * c2;
  const float m_new = fmaxf(m_run, mt);
  const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
  float ls = 0.0f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[ks][r] = __builtin_amdgcn_exp2f(fmaf(s[ks][r], c2, -m_new)); ls += s[ks][r]; }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
  l_run = l_run * alpha + ls;
  m_run = m_new;
  drop<XV>(s, key);
  to_frags(s, p);
}
// lazy integer max: `s` already holds score - m.  Raise m (rare) when a row's tile max exceeds the threshold.
__device__ __forceinline__ void raise_max(f32x16 (&s)[2], float mt, f32x16& negm, float& l_run, f32x16 (&o)[2]) {
  const float mi = ceilf(fmaxf(mt, 0.0f));
  const float alpha = __builtin_amdgcn_exp2f(-mi);
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[ks][r] -= mi;
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] -= mi;
  l_run *= alpha;
}
template <int XV>
__device__ __forceinline__ void exp_sum(f32x16 (&s)[2], bf16x8 (&p)[4], float& l_run, unsigned key) {
  float l0 = 0.0f, l1 = 0.0f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      s[ks][r] = __builtin_amdgcn_exp2f(s[ks][r]);
      s[ks][r + 1] = __builtin_amdgcn_exp2f(s[ks][r + 1]);
      l0 += s[ks][r];
      l1 += s[ks][r + 1];
    }
  l_run += l0 + l1;
  drop<XV>(s, key);
  to_frags(s, p);
}
__device__ __forceinline__ void pv(f32x16 (&o)[2], const bf16_t* sV, const bf16x8 (&p)[4], int lane) {
  const int hi = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        o[dt] = MFMA(ld_tr(sV, dt * 32, ks * 32 + 16 * t + 4 * hi, ks * 32 + 16 * t + 4 * hi + 8, lane), p[ks * 2 + t], o[dt]);
}

template <int NV, int NT>
__device__ __forceinline__ void interleave16() {
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
    __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);  // NV VALU
    if (NT > 0) __builtin_amdgcn_sched_group_barrier(0x400, NT, 0);  // NT transcendental
  }
}

template <int MODE, bool CINIT, int XV, int NV, int NT, int WPS>
__global__ __launch_bounds__(256, WPS) void probe(float* out, long long* cyc, int iters, float c2) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[4 * TILE];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4 * TILE; i += 256) smem[i] = static_cast<__bf16>(0.001f * (float)((i * 7919) % 257 - 128));
  __syncthreads();
  bf16x8 fq[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int j = 0; j < 8; ++j) fq[kk][j] = static_cast<__bf16>(0.01f * (float)((lane * 31 + kk * 8 + j) % 17 - 8));
  f32x16 o[2], negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.0f; o[1][r] = 0.0f; negm[r] = 0.0f; }
  float m_run = -1.0e30f, l_run = 0.0f;
  const unsigned key = 0x1234567u + tid * 2654435761u;
  const long long t0 = __builtin_readcyclecounter();
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
      const bf16_t* sK = smem + (it & 1) * 2 * TILE;
      f32x16 s[2];
      bf16x8 p[4];
      qk(s, sK, fq, negm, lane);
      if (CINIT) {
        float mt = tile_max(s);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        if (__builtin_expect(__any(mt > 8.0f), 0)) raise_max(s, mt, negm, l_run, o);
        exp_sum<XV>(s, p, l_run, key + it);
      } else {
        softmax_eager<XV>(s, p, m_run, l_run, o, c2, key + it);
      }
      pv(o, sK + TILE, p, lane);
    }
  } else {
    f32x16 sA[2], sB[2];
    bf16x8 pA[4], pB[4];
    qk(sA, smem, fq, negm, lane);
    exp_sum<XV>(sA, pA, l_run, key);
    qk(sB, smem + 2 * TILE, fq, negm, lane);
    float mt = tile_max(sB);
    // steady state, unrolled by two so that every register set is statically named
    for (int it = 0; it < iters; it += 2) {
      // step a: [rare: raise max for sB] ; QK -> sA ; exp(sB) -> pB ; PV(pA) ; max(sA)
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      if (__builtin_expect(__any(mt > 8.0f), 0)) raise_max(sB, mt, negm, l_run, o);
      qk(sA, smem, fq, negm, lane);
      exp_sum<XV>(sB, pB, l_run, key + it);
      pv(o, smem + TILE, pA, lane);
      interleave16<NV, NT>();
      mt = tile_max(sA);
      // step b
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      if (__builtin_expect(__any(mt > 8.0f), 0)) raise_max(sA, mt, negm, l_run, o);
      qk(sB, smem + 2 * TILE, fq, negm, lane);
      exp_sum<XV>(sA, pA, l_run, key + it + 1);
      pv(o, smem + 3 * TILE, pB, lane);
      interleave16<NV, NT>();
      mt = tile_max(sB);
    }
    float sink = mt;
#pragma unroll
    for (int r = 0; r < 16; ++r) sink += sA[0][r] + sB[1][r] + (float)pA[0][r & 7] + (float)pB[3][r & 7];
    l_run += sink * 1e-30f;
  }
  const long long t1 = __builtin_readcyclecounter();
  float acc = l_run + m_run + negm[3];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += o[dt][r];
  out[blockIdx.x * 256 + tid] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}


// ---- unpadded 128-byte-row images written by LDS-DMA (lane-linear 1 KiB pieces; the swizzle sits on the SOURCE address) ----
// K image: 16-byte chunk c of key row r is stored at chunk position c ^ ((r >> 1) & 7)  (ds_read_b128, one row per lane)
// V image: chunk c of key row r at position c ^ (4 * ((r >> 1) & 1))                    (ds_read_b64_tr_b16, 4 rows x 64 B per half-wave)
__device__ __forceinline__ bf16x8 ld_kswz(const bf16_t* tile, int r, int c) {
  return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(tile) + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
}
__device__ __forceinline__ bf16x8 ld_tr_vswz(const bf16_t* tile, int mn0, int ka, int kb, int lane) {
  const int s = lane & 15;
  const int col = mn0 + 16 * ((lane >> 4) & 1) + 4 * (s & 3);  // element column of this lane's 8-byte source chunk
  const int ra = ka + (s >> 2), rb = kb + (s >> 2);
  const char* base = reinterpret_cast<const char*>(tile);
  const char* pa = base + ra * 128 + ((((col >> 3) ^ (4 * ((ra >> 1) & 1))) << 4) | ((col & 7) << 1));
  const char* pb = base + rb * 128 + ((((col >> 3) ^ (4 * ((rb >> 1) & 1))) << 4) | ((col & 7) << 1));
  const v4s_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)pa);
  const v4s_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)pb);
  u16x8 t;
  t[0] = x[0]; t[1] = x[1]; t[2] = x[2]; t[3] = x[3]; t[4] = y[0]; t[5] = y[1]; t[6] = y[2]; t[7] = y[3];
  return __builtin_bit_cast(bf16x8, t);
}
template <bool SWZ>
__device__ __forceinline__ void qk_t(f32x16 (&s)[2], const bf16_t* sK, const bf16x8 (&fq)[4], const f32x16& c0, int lane) {
  const int hi = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bf16x8 a = SWZ ? ld_kswz(sK, ks * 32 + (lane & 31), kk * 2 + hi) : ld_contig(sK + (ks * 32 + (lane & 31)) * LD + kk * 16 + 8 * hi);
      s[ks] = MFMA(a, fq[kk], kk == 0 ? c0 : s[ks]);
    }
  }
}
template <bool SWZ>
__device__ __forceinline__ void pv_t(f32x16 (&o)[2], const bf16_t* sV, const bf16x8 (&p)[4], int lane) {
  const int hi = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int ka = ks * 32 + 16 * t + 4 * hi;
        const bf16x8 a = SWZ ? ld_tr_vswz(sV, dt * 32, ka, ka + 8, lane) : ld_tr(sV, dt * 32, ka, ka + 8, lane);
        o[dt] = MFMA(a, p[ks * 2 + t], o[dt]);
      }
}
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
constexpr int GROW = 2304;  // elements per token row of the packed q|k|v projection (12 heads x 64 x 3)

// STG 1: register-staged double buffer + __syncthreads per tile (round 2); STG 2: LDS-DMA double buffer; STG 3: LDS-DMA 3-slot ring
template <int STG, bool CINIT, int XV, int WPS>
__global__ __launch_bounds__(256, WPS) void probe_stg(float* out, long long* cyc, int ntiles, float c2, const bf16_t* kv, int nbh) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x % nbh;
  const bf16_t* Kg = kv + (size_t)(bh / 12) * 1536 * GROW + 768 + (bh % 12) * 64;
  const bf16_t* Vg = Kg + 768;
  bf16x8 fq[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int j = 0; j < 8; ++j) fq[kk][j] = static_cast<__bf16>(0.01f * (float)((lane * 31 + kk * 8 + j) % 17 - 8));
  f32x16 o[2], negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.0f; o[1][r] = 0.0f; negm[r] = 0.0f; }
  float m_run = -1.0e30f, l_run = 0.0f;
  const unsigned key = 0x1234567u + tid * 2654435761u;
  constexpr int TB = STG == 1 ? TILE : 64 * 64;  // elements per tile image
  // ---- staging state ----
  u32x4 rk[2], rv[2];
  auto fetch = [&](int j) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + 256 * i, row = v >> 3, c = v & 7;
      rk[i] = *reinterpret_cast<const u32x4*>(Kg + (size_t)(j * 64 + row) * GROW + c * 8);
      rv[i] = *reinterpret_cast<const u32x4*>(Vg + (size_t)(j * 64 + row) * GROW + c * 8);
    }
  };
  auto stage = [&](int st) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + 256 * i, row = v >> 3, c = v & 7;
      *reinterpret_cast<u32x4*>(smem + st * 2 * TB + row * LD + c * 8) = rk[i];
      *reinterpret_cast<u32x4*>(smem + st * 2 * TB + TB + row * LD + c * 8) = rv[i];
    }
  };
  // LDS-DMA: wave w moves pieces 2w, 2w+1 (8 rows of 128 B each) of the K and of the V tile.  Issued through inline asm: hipcc
  // orders LDS reads behind every LDS-DMA it knows of with s_waitcnt vmcnt(0) (seen in the .s in front of the V transpose reads),
  // which would expose the whole DMA latency per tile; the waits are counted by hand instead.
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  auto mk_rsrc = [](const void* p) {
    const unsigned long long a = (unsigned long long)p;
    i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    r[2] = 0x7fffffff;
    r[3] = 0x00020000;
    return r;
  };
  const i32x4_t rsK = mk_rsrc(Kg), rsV = mk_rsrc(Vg);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  unsigned vk[2], vv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * (2 * wv + i) + (lane >> 3), pos = lane & 7;
    vk[i] = (unsigned)(row * GROW * 2 + ((pos ^ ((row >> 1) & 7)) << 4));
    vv[i] = (unsigned)(row * GROW * 2 + ((pos ^ (4 * ((row >> 1) & 1))) << 4));
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
  auto dma1 = [&](const i32x4_t& rs, unsigned lds, unsigned voff, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
  };
  auto dma = [&](int j, int st) {
    const unsigned kb = lds0 + st * 2 * TB * 2 + (2 * wv) * 1024;
    const int soff = j * 64 * GROW * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      dma1(rsK, kb + i * 1024, vk[i], soff);
      dma1(rsV, kb + TB * 2 + i * 1024, vv[i], soff);
    }
  };
  constexpr bool SWZ = STG != 1;
  if (STG == 1) { fetch(0); stage(0); __syncthreads(); }
  if (STG == 2) { dma(0, 0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
  if (STG == 3) { dma(0, 0); dma(1, 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
  const long long t0 = __builtin_readcyclecounter();
  int cur = 0;
  for (int it = 0; it < ntiles; ++it) {
    const int nxt = (it + 1) % 24;  // 24 tiles of 64 keys, walked round and round
    if (STG == 1) fetch(nxt);
    if (STG == 2) dma(nxt, cur ^ 1);
    if (STG == 3) dma((it + 2) % 24, (cur + 2) % 3);
    const bf16_t* sK = smem + cur * 2 * TB;
    f32x16 s[2];
    bf16x8 p[4];
    qk_t<SWZ>(s, sK, fq, negm, lane);
    if (CINIT) {
      float mt = tile_max(s);
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      if (__builtin_expect(__any(mt > 8.0f), 0)) raise_max(s, mt, negm, l_run, o);
      exp_sum<XV>(s, p, l_run, key + it);
    } else {
      softmax_eager<XV>(s, p, m_run, l_run, o, c2, key + it);
    }
    pv_t<SWZ>(o, sK + TB, p, lane);
    if (STG == 1) { stage(cur ^ 1); __syncthreads(); cur ^= 1; }
    if (STG == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); cur ^= 1; }
    if (STG == 3) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); __builtin_amdgcn_s_barrier(); cur = (cur + 1) % 3; }
  }
  const long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float acc = l_run + m_run + negm[3];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += o[dt][r];
  out[blockIdx.x * 256 + tid] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int STG, bool CINIT, int XV, int WPS>
void run_stg(const char* name, float* out, long long* cyc, int ntiles, const bf16_t* kv, int rounds = 1) {
  const size_t lds = STG == 1 ? 4 * TILE * 2 : (STG == 2 ? 4 * 8192 : 6 * 8192);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_stg<STG, CINIT, XV, WPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  int occ = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe_stg<STG, CINIT, XV, WPS>, 256, lds);
  for (int W = (rounds > 1 ? occ : 1); W <= occ && W <= 4; ++W) {
    const int grid = 256 * W * rounds;
    hipLaunchKernelGGL((probe_stg<STG, CINIT, XV, WPS>), dim3(grid), dim3(256), lds, 0, out, cyc, 48, 0.18f, kv, 96);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe_stg<STG, CINIT, XV, WPS>), dim3(grid), dim3(256), lds, 0, out, cyc, ntiles, 0.18f, kv, 96);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    double avg = 0;
    for (int i = 0; i < grid; ++i) avg += (double)cyc[i];
    avg /= grid;
    const double tf = 4.0 * 32 * 64 * 64 * (double)ntiles * grid * 4 / (ms * 1e-3) / 1e12;
    printf("%-40s W=%d/%d %8.1f cyc/tile/wave %8.1f cyc/tile/SIMD  MFMA busy %5.1f%%  %7.1f TF/s  %.3f ms\n", name, W, occ, avg / ntiles, avg / ntiles / W,
           100.0 * 512.0 * W / (avg / ntiles), tf, ms);
  }
}

template <int MODE, bool CINIT, int XV, int NV, int NT, int WPS>
void run(const char* name, float* out, long long* cyc, int iters) {
  int occ = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe<MODE, CINIT, XV, NV, NT, WPS>, 256, 0);
  for (int W = 1; W <= occ && W <= 4; ++W) {
    const int grid = 256 * W;
    hipLaunchKernelGGL((probe<MODE, CINIT, XV, NV, NT, WPS>), dim3(grid), dim3(256), 0, 0, out, cyc, 64, 0.18f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, CINIT, XV, NV, NT, WPS>), dim3(grid), dim3(256), 0, 0, out, cyc, iters, 0.18f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    double avg = 0;
    for (int i = 0; i < grid; ++i) avg += (double)cyc[i];
    avg /= grid;
    const double tf = 4.0 * 32 * 64 * 64 * (double)iters * grid * 4 / (ms * 1e-3) / 1e12;
    printf("%-40s W=%d/%d %8.1f cyc/tile/wave %8.1f cyc/tile/SIMD  MFMA busy %5.1f%%  %7.1f TF/s  %.3f ms\n", name, W, occ, avg / iters, avg / iters / W,
           100.0 * 512.0 * W / (avg / iters), tf, ms);
  }
}

int main(int argc, char** argv) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 8192 * 256 * 4);
  (void)hipMallocManaged(&cyc, 8192 * 8);
  const int iters = 2048;
  const int which = argc > 1 ? atoi(argv[1]) : 3;
  if (which & 1) {
    run<0, false, 0, 0, 0, 3>("phased eager fma+exp", out, cyc, iters);
    run<0, true, 0, 0, 0, 3>("phased C-init lazy max", out, cyc, iters);
    run<0, false, 8, 0, 0, 3>("phased eager, +8 int/pair", out, cyc, iters);
    run<0, true, 8, 0, 0, 3>("phased C-init, +8 int/pair", out, cyc, iters);
    run<1, true, 0, 4, 2, 2>("pipelined NV4 NT2", out, cyc, iters);
    run<1, true, 0, 6, 0, 2>("pipelined NV6 (trans as VALU)", out, cyc, iters);
    run<1, true, 4, 8, 2, 2>("pipelined NV8 NT2, +4 int/pair", out, cyc, iters);
  }
  if (which & 2) {
    // K / V of 8 batches x 1536 tokens x 2304 columns (packed q|k|v rows), values in [-1, 1)
    const size_t n = (size_t)8 * 1536 * GROW;
    std::vector<unsigned short> h(n);
    unsigned x = 12345u;
    for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; float f = (float)((x >> 8) & 0xffff) / 32768.0f - 1.0f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
    bf16_t* kv; (void)hipMalloc(&kv, n * 2);
    (void)hipMemcpy(kv, h.data(), n * 2, hipMemcpyHostToDevice);
    run_stg<1, false, 0, 3>("reg-staged + syncthreads, eager", out, cyc, iters, kv);
    run_stg<1, true, 0, 3>("reg-staged + syncthreads, C-init", out, cyc, iters, kv);
    run_stg<2, false, 0, 3>("LDS-DMA 2 slots, eager", out, cyc, iters, kv);
    run_stg<2, true, 0, 3>("LDS-DMA 2 slots, C-init", out, cyc, iters, kv);
    run_stg<3, true, 0, 3>("LDS-DMA 3 slots, C-init", out, cyc, iters, kv);
    run_stg<3, true, 0, 4>("LDS-DMA 3 slots, C-init (<=128 VGPR)", out, cyc, iters, kv);
    run_stg<2, true, 0, 4>("LDS-DMA 2 slots, C-init (<=128 VGPR)", out, cyc, iters, kv);
    run_stg<1, false, 8, 3>("reg-staged, eager, +8 int/pair", out, cyc, iters, kv);
    run_stg<3, true, 4, 3>("LDS-DMA 3 slots, C-init, +4 int/pair", out, cyc, iters, kv);
  }
  if (which & 4) {  // short workgroups in several rounds, as the real launch has them (24 tiles, 6 rounds of 768 workgroups)
    const size_t n = (size_t)8 * 1536 * GROW;
    bf16_t* kv; (void)hipMalloc(&kv, n * 2);
    (void)hipMemset(kv, 0x3c, n * 2);
    run_stg<2, true, 0, 3>("DMA 2 slots C-init, 24 tiles x 6 rounds", out, cyc, 24, kv, 6);
    run_stg<2, true, 0, 3>("DMA 2 slots C-init, 24 tiles x 1 round", out, cyc, 24, kv, 1);
    run_stg<2, true, 0, 3>("DMA 2 slots C-init, 96 tiles x 6 rounds", out, cyc, 96, kv, 6);
    run_stg<2, true, 0, 3>("DMA 2 slots C-init, 2048 tiles x 1 round", out, cyc, 2048, kv, 1);
  }
  return 0;
}
