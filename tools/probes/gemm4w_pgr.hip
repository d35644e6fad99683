// PROBE (not built into libcst_hip.so): the workgroup shape of the vendor's large bf16 GEMM — 256 x 256 x 64 macro-tile, FOUR waves
// (one per SIMD), each a 128 x 128 wave tile whose 256 accumulator registers live in the AGPR half of the register file — written
// from scratch as a register-staged kernel: global -> VGPR (16 x 16 B per thread and K tile, issued one K tile ahead) -> ds_write_b128
// into a swizzled k-major LDS image (two stages) -> ds_read_b128 fragments (prefetched one k16 step ahead) -> v_mfma_f32_32x32x16_bf16.
// Per MFMA it reads 0.5 fragments from LDS where the 8-wave kernel of the library (128 x 64 wave tiles) reads 0.75, and it has no
// LDS-DMA instruction (100-185 issue cycles each inside a loaded phase, MI355X_MICROARCH.md) in its loop.
// Question: how far does the plain structure get before any hand scheduling?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/gemm4w_pgr tools/probes/gemm4w_pgr.hip && /tmp/gemm4w_pgr
// C[M, N] (bf16) = A[M, K] B[N, K]^T, both operands k-major bf16; M, N multiples of 256, K a multiple of 64.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16_t;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;

constexpr int BM = 256, BN = 256, BK = 64, NT = 256;
constexpr int IMG = 256 * 128;          // one operand image: 256 rows x 128 B
constexpr int STAGE = 2 * IMG;          // A | B
constexpr int LDS_BYTES = 4 * 128 * 272;  // 136 KiB: two K-tile stages (128 KiB); the epilogue images of the four waves need 136

__device__ __forceinline__ unsigned pack2(float a, float b) {
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fffu + ((ua >> 16) & 1u);
  ub += 0x7fffu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xffff0000u);
}

template <int PF>
__global__ __launch_bounds__(NT) void gemm4w_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                    int M, int N, int K, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane & 31, hi = lane >> 5;
  // XCD-aware grouped tile order (the map of gemm.hip): each XCD's L2 sees a compact patch of tiles
  const int ntiles = tiles_m * tiles_n;
  int id = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = id % 8, loc = id / 8;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * tiles_n;
  const int grp = id / per_group, rem = id % per_group;
  const int gm0 = grp * GROUP_M;
  const int gsz = (tiles_m - gm0 < GROUP_M) ? (tiles_m - gm0) : GROUP_M;
  const int tm = gm0 + rem % gsz, tn = rem / gsz;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  // ---- staging map: vector v = tid + 256 i (i < 8) of an operand tile = (row v / 8, 16-byte chunk v % 8); LDS chunk = chunk ^ swizzle(row)
  const int srow = tid >> 3, sch = tid & 7;                 // rows srow + 32 i
  const bf16_t* pa = A + (m0 + srow) * (int64_t)K + sch * 8;
  const bf16_t* pb = B + (n0 + srow) * (int64_t)K + sch * 8;
  const int64_t rstep = 32 * (int64_t)K;                    // 32 rows down
  unsigned soff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = srow + 32 * i;
    soff[i] = (unsigned)(r * 128 + ((sch ^ ((r >> 1) & 7)) << 4));
  }
  u32x4 ra[8], rb[8];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 8; ++i) ra[i] = *reinterpret_cast<const u32x4*>(pa + i * rstep + (int64_t)kt * BK);
#pragma unroll
    for (int i = 0; i < 8; ++i) rb[i] = *reinterpret_cast<const u32x4*>(pb + i * rstep + (int64_t)kt * BK);
  };
  auto lstore = [&](int st) {
    char* base = smem + st * STAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(base + soff[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(base + IMG + soff[i]) = rb[i];
  };

  // ---- fragment addresses: row (wm * 128 + i * 32 + lrow) of the A image, chunk (2 kk + hi) ^ swizzle(row); the swizzle of row
  //      i * 32 + lrow does not depend on i
  const int swr = (lrow >> 1) & 7;
  const unsigned fa0 = (unsigned)((wm * 128 + lrow) * 128), fb0 = (unsigned)(IMG + (wn * 128 + lrow) * 128);
  unsigned fch[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fch[kk] = (unsigned)(((2 * kk + hi) ^ swr) << 4);

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nkt = K / BK;
  gload(0);
  lstore(0);
  if (nkt > 1) gload(1);
  __syncthreads();

  bf16x8 fa[2][4], fb[2][4];
  auto fread = [&](int st, int kk, int buf) {
    const char* base = smem + st * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[buf][i] = *reinterpret_cast<const bf16x8*>(base + fa0 + i * 4096 + fch[kk]);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[buf][j] = *reinterpret_cast<const bf16x8*>(base + fb0 + j * 4096 + fch[kk]);
  };

  if (PF == 5 || PF == 6) {
    constexpr bool EXACT = PF == 6;
    // LDS-DMA instead of register staging: the next K tile goes HBM/L2 -> LDS by 16 buffer_load ... lds per wave (1 KiB each, the lane
    // -> (row, chunk) map carries the swizzle), issued behind every second MFMA of the first two k16 steps; nothing passes through
    // VGPRs or the ds_write path.  Stage st ^ 1 was last read a K tile ago (barrier in between); its DMAs are waited for (vmcnt(0))
    // in front of the barrier that ends this tile, >= 32 MFMAs after the last issue.
    const int row0 = 8 * wave + (lane >> 3);
    const unsigned voff = (unsigned)row0 * (unsigned)K * 2u + (unsigned)(((lane & 7) ^ ((row0 >> 1) & 7)) << 4);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + m0 * (int64_t)K), (short)0, (int)0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(B + n0 * (int64_t)K), (short)0, (int)0x7fffffff, 0x00020000);
    const unsigned rstep32 = 32u * (unsigned)K * 2u;
    auto dma = [&](int d, int kt, int stg) {  // d < 8: A group wave + 4 d; else B group wave + 4 (d - 8)
      const int i = d & 7;
      char* dst = smem + stg * STAGE + (d < 8 ? 0 : IMG) + (wave + 4 * i) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d < 8 ? rsA : rsB, (__attribute__((address_space(3))) void*)dst, 16, (int)voff,
                                               (int)((unsigned)i * rstep32 + (unsigned)kt * 128u), 0, 0);
    };
    __syncthreads();  // (the prologue's register-staged tile 0 is in stage 0; tile 1 arrives by DMA now)
    if (nkt > 1) {
#pragma unroll
      for (int d = 0; d < 16; ++d) dma(d, 1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (EXACT) {
      // the same work with the issue order pinned by sched_barrier(0) behind every MFMA pair: { 2 MFMA ; one fragment read of the next
      // k16 step ; one DMA (first two steps) }
      for (int kt = 0; kt < nkt; ++kt) {
        const int st = kt & 1;
        const char* rbase = smem + st * STAGE;
        fread(st, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
          for (int pr = 0; pr < 8; ++pr) {
            const int i = pr >> 1, j0 = (pr & 1) * 2;
            acc[i][j0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk & 1][j0], fa[kk & 1][i], acc[i][j0], 0, 0, 0);
            acc[i][j0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk & 1][j0 + 1], fa[kk & 1][i], acc[i][j0 + 1], 0, 0, 0);
            if (kk < 3) {
              const int nb = (kk + 1) & 1;
              if (pr < 4) fa[nb][pr] = *reinterpret_cast<const bf16x8*>(rbase + fa0 + pr * 4096 + fch[kk + 1]);
              else fb[nb][pr - 4] = *reinterpret_cast<const bf16x8*>(rbase + fb0 + (pr - 4) * 4096 + fch[kk + 1]);
            }
            if (kk < 2 && kt > 0) dma(kk * 8 + pr, kt + 1 < nkt ? kt + 1 : nkt - 1, st ^ 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    } else
    for (int kt = 0; kt < nkt; ++kt) {
      const int st = kt & 1;
      const int ktn = kt + 2 < nkt ? kt + 2 : nkt - 1;
      fread(st, 0, 0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk < 3) fread(st, kk + 1, (kk + 1) & 1);
        if (kk < 2 && kt > 0) {  // (tile 1 was fetched in the prologue: the first pass of the loop has nothing to stage)
#pragma unroll
          for (int u = 0; u < 8; ++u) dma(kk * 8 + u, kt + 1 < nkt ? kt + 1 : nkt - 1, st ^ 1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk & 1][j], fa[kk & 1][i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (m < 8 && kk < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if ((m & 1) && kk < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
      (void)ktn;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else if (PF == 4) {
    // hand-placed issue order: every k16 step = 16 MFMAs with ONE filler behind each — the 8 fragment reads of the next step, then a
    // quarter of the staged tile's ds_writes (4), then a quarter of the next tile's global loads (4).  No branches inside the K tile
    // (the last tiles re-fetch a clamped tile into the stage nobody reads), so the whole tile is one scheduling region.
    for (int kt = 0; kt < nkt; ++kt) {
      const int st = kt & 1;
      const int ktn = kt + 2 < nkt ? kt + 2 : nkt - 1;
      char* wbase = smem + (st ^ 1) * STAGE;
      fread(st, 0, 0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk < 3) fread(st, kk + 1, (kk + 1) & 1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {  // quarter kk of the staging: A vectors 2kk, 2kk+1 and B vectors 2kk, 2kk+1
          const int i = 2 * kk + u;
          *reinterpret_cast<u32x4*>(wbase + soff[i]) = ra[i];
          *reinterpret_cast<u32x4*>(wbase + IMG + soff[i]) = rb[i];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = 2 * kk + u;
          ra[i] = *reinterpret_cast<const u32x4*>(pa + i * rstep + (int64_t)ktn * BK);
          rb[i] = *reinterpret_cast<const u32x4*>(pb + i * rstep + (int64_t)ktn * BK);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk & 1][j], fa[kk & 1][i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // one MFMA
          if (m < 8 && kk < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one ds_read
          if (m >= 8 && m < 12) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); // one ds_write
          if (m >= 12) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);          // one global load
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else
  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt & 1;
    fread(st, 0, 0);
    if (PF == 1 && kt + 1 < nkt) {  // the tile fetched during the previous K tile goes to the other stage, the one after it is requested
      lstore(st ^ 1);
      if (kt + 2 < nkt) gload(kt + 2);
    }
    if (PF == 3 && kt + 2 < nkt) gload(kt + 2);  // (timing only: loads without the LDS stores; PF == 2: neither — wrong results)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) fread(st, kk + 1, (kk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);  // (without it hipcc sinks the reads below this step's MFMAs and re-uses one fragment set: no prefetch)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk & 1][j], fa[kk & 1][i], acc[i][j], 0, 0, 0);  // swapped: lane -> row, regs -> 4 cols
      __builtin_amdgcn_sched_barrier(0);
    }
    if (PF == 0 && kt + 1 < nkt) lstore(st ^ 1);
    __syncthreads();
    if (PF == 0 && kt + 2 < nkt) gload(kt + 2);
  }

  // ---- epilogue: swapped layout: acc[i][j][r]: output row = wm*128 + i*32 + lrow, column = wn*128 + j*32 + 8*(r/4) + 4*hi + r%4.
  //      Each wave turns its 128 x 128 block into a bf16 image in LDS (row stride 272 B) and stores it row-contiguously, 16 B per lane
  //      (a first version stored the 8-byte register quads straight from the accumulators — 64 rows per store instruction: 260 TF/s
  //      on a K = 768 shape).
  __syncthreads();
  {
    constexpr int ERS = 272;
    char* img = smem + wave * (128 * ERS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u32x2 w = {pack2(acc[i][j][4 * q], acc[i][j][4 * q + 1]), pack2(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3])};
          *reinterpret_cast<u32x2*>(img + (i * 32 + lrow) * ERS + (j * 32 + 8 * q + 4 * hi) * 2) = w;
        }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave reads back only what it wrote itself
    bf16_t* cbase = C + (m0 + wm * 128) * (int64_t)N + n0 + wn * 128;
#pragma unroll 8
    for (int t = 0; t < 32; ++t) {
      const int row = t * 4 + (lane >> 4), ch = lane & 15;
      const u32x4 v = *reinterpret_cast<const u32x4*>(img + row * ERS + ch * 16);
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(cbase + row * (int64_t)N + ch * 8));
    }
  }
}

// ---- the pinned-order LDS-DMA loop alone, for 256 x (64 TN) tiles: TN = 4 is the kernel above (PF = 6), TN = 3 makes N = 768 exactly
//      four tile columns (wave tile 128 x 96, 14 DMAs per wave and K tile, 12 MFMAs per k16 step) ----
template <int TN>
__global__ __launch_bounds__(NT) void gemm4w_dma_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                        int M, int N, int K, int tiles_m, int tiles_n) {
  constexpr int BNt = 64 * TN, IMGB = BNt * 128, STG = IMG + IMGB, GB = BNt / 32, G = 8 + GB, NSLOT = 2 * TN, NFR = 4 + TN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane & 31, hi = lane >> 5;
  const int ntiles = tiles_m * tiles_n;
  int id = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = id % 8, loc = id / 8;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * tiles_n;
  const int grp = id / per_group, rem = id % per_group;
  const int gm0 = grp * GROUP_M;
  const int gsz = (tiles_m - gm0 < GROUP_M) ? (tiles_m - gm0) : GROUP_M;
  const int tm = gm0 + rem % gsz, tn = rem / gsz;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BNt;

  const int row0 = 8 * wave + (lane >> 3);
  const unsigned voff = (unsigned)row0 * (unsigned)K * 2u + (unsigned)(((lane & 7) ^ ((row0 >> 1) & 7)) << 4);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + m0 * (int64_t)K), (short)0, (int)0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(B + n0 * (int64_t)K), (short)0, (int)0x7fffffff, 0x00020000);
  const unsigned rstep32 = 32u * (unsigned)K * 2u;
  auto dma = [&](int d, int kt, int stg) {
    const int i = d < 8 ? d : d - 8;
    char* dst = smem + stg * STG + (d < 8 ? 0 : IMG) + (wave + 4 * i) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(d < 8 ? rsA : rsB, (__attribute__((address_space(3))) void*)dst, 16, (int)voff,
                                             (int)((unsigned)i * rstep32 + (unsigned)kt * 128u), 0, 0);
  };
  const int swr = (lrow >> 1) & 7;
  const unsigned fa0 = (unsigned)((wm * 128 + lrow) * 128), fb0 = (unsigned)(IMG + (wn * 32 * TN + lrow) * 128);
  unsigned fch[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fch[kk] = (unsigned)(((2 * kk + hi) ^ swr) << 4);
  f32x16 acc[4][TN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int nkt = K / BK;
#pragma unroll
  for (int d = 0; d < G; ++d) dma(d, 0, 0);
  bf16x8 fa[2][4], fb[2][TN];
  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* rbase = smem + st * STG;
    auto read_one = [&](int f, int kk, int nb) {
      if (f < 4) fa[nb][f] = *reinterpret_cast<const bf16x8*>(rbase + fa0 + f * 4096 + fch[kk]);
      else fb[nb][f - 4] = *reinterpret_cast<const bf16x8*>(rbase + fb0 + (f - 4) * 4096 + fch[kk]);
    };
#pragma unroll
    for (int f = 0; f < NFR; ++f) read_one(f, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    const bool more = kt + 1 < nkt;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int sl = 0; sl < NSLOT; ++sl) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int m = 2 * sl + u, i = m / TN, j = m % TN;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk & 1][j], fa[kk & 1][i], acc[i][j], 0, 0, 0);
        }
        if (kk < 3) {
          if (sl < NFR) read_one(sl, kk + 1, (kk + 1) & 1);
          if (sl == 0 && NFR > NSLOT) {
#pragma unroll
            for (int f = NSLOT; f < NFR; ++f) read_one(f, kk + 1, (kk + 1) & 1);
          }
        }
        if (kk * NSLOT + sl < G && more) dma(kk * NSLOT + sl, kt + 1, st ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();
  {
    constexpr int ERS = 64 * TN + 16;  // bf16 image row stride in bytes (32 TN columns x 2 B + 16)
    char* img = smem + wave * (128 * ERS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u32x2 w = {pack2(acc[i][j][4 * q], acc[i][j][4 * q + 1]), pack2(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3])};
          *reinterpret_cast<u32x2*>(img + (i * 32 + lrow) * ERS + (j * 32 + 8 * q + 4 * hi) * 2) = w;
        }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    bf16_t* cbase = C + (m0 + wm * 128) * (int64_t)N + n0 + wn * 32 * TN;
    constexpr int CPR = 4 * TN;  // 16-byte chunks per row of the wave tile
    for (int v = lane; v < 128 * CPR; v += 64) {
      const int row = v / CPR, ch = v % CPR;
      const u32x4 x = *reinterpret_cast<const u32x4*>(img + row * ERS + ch * 16);
      __builtin_nontemporal_store(x, reinterpret_cast<u32x4*>(cbase + row * (int64_t)N + ch * 8));
    }
  }
}

template <int TN>
static double run_dma(const bf16_t* dA, const bf16_t* dB, bf16_t* dC, int M, int N, int K, int iters) {
  constexpr int lds = 2 * (IMG + 64 * TN * 128) > 4 * 128 * (64 * TN + 16) ? 2 * (IMG + 64 * TN * 128) : 4 * 128 * (64 * TN + 16);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_dma_kernel<TN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int tm = M / BM, tn = N / (64 * TN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm4w_dma_kernel<TN>, dim3(tm * tn), dim3(NT), lds, 0, dA, dB, dC, M, N, K, tm, tn);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm4w_dma_kernel<TN>, dim3(tm * tn), dim3(NT), lds, 0, dA, dB, dC, M, N, K, tm, tn);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

template <int PF>
static double run(const bf16_t* dA, const bf16_t* dB, bf16_t* dC, int M, int N, int K, int iters) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel<PF>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  const int tm = M / BM, tn = N / BN;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm4w_kernel<PF>, dim3(tm * tn), dim3(NT), LDS_BYTES, 0, dA, dB, dC, M, N, K, tm, tn);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm4w_kernel<PF>, dim3(tm * tn), dim3(NT), LDS_BYTES, 0, dA, dB, dC, M, N, K, tm, tn);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main() {
  const int shapes[][3] = {{8192, 8192, 8192}, {47872, 768, 3072}, {31744, 768, 3072}, {31744, 768, 2304}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    uint32_t x = 12345u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f; };
    for (auto& v : hA) v = f2bf(rnd());
    for (auto& v : hB) v = f2bf(rnd());
    bf16_t *dA, *dB, *dC;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, hC.size() * 2);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    for (int pf = 0; pf < 7; ++pf) {
      if (pf == 3) continue;
      hipMemset(dC, 0, hC.size() * 2);
      const double ms = pf == 0 ? run<0>(dA, dB, dC, M, N, K, 20) : pf == 1 ? run<1>(dA, dB, dC, M, N, K, 20) : pf == 2 ? run<2>(dA, dB, dC, M, N, K, 20) : pf == 4 ? run<4>(dA, dB, dC, M, N, K, 20) : pf == 5 ? run<5>(dA, dB, dC, M, N, K, 20) : run<6>(dA, dB, dC, M, N, K, 20);
      hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
      double worst = 0.0;
      for (int t = 0; t < 64; ++t) {  // sampled elements against a double-precision dot product
        x = x * 1664525u + 1013904223u; const int r = (int)((x >> 4) % (unsigned)M);
        x = x * 1664525u + 1013904223u; const int c = (int)((x >> 4) % (unsigned)N);
        double ref = 0.0;
        for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)r * K + k]) * (double)bf2f(hB[(size_t)c * K + k]);
        const double got = bf2f(hC[(size_t)r * N + c]);
        const double err = fabs(got - ref) / (fabs(ref) + 1.0);
        worst = (err > worst || err != err) ? err : worst;  // (NaN sticks)
      }
      printf("%6d x %5d x %5d  ds_write %s: %.3f ms  %.0f TF/s  (worst sampled rel. error %.1e)\n", M, N, K,
             pf == 1 ? "at the head of the next K tile, loads a tile ahead" : pf == 0 ? "after the K tile's MFMAs, loads behind the barrier " : pf == 2 ? "NONE and no loads (timing only)                   " : pf == 4 ? "one filler behind every MFMA (sched_group_barrier)" : pf == 5 ? "NONE: LDS-DMA behind every 2nd MFMA of steps 0, 1 " : "NONE: LDS-DMA, order pinned per MFMA pair         ", ms, 2.0 * M * N * K / ms / 1e9, worst);
    }
    for (int tn = 4; tn >= 3; --tn) {
      if (N % (64 * tn)) continue;
      hipMemset(dC, 0, hC.size() * 2);
      const double ms = tn == 4 ? run_dma<4>(dA, dB, dC, M, N, K, 20) : run_dma<3>(dA, dB, dC, M, N, K, 20);
      hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
      double worst = 0.0;
      for (int t = 0; t < 64; ++t) {
        x = x * 1664525u + 1013904223u; const int r = (int)((x >> 4) % (unsigned)M);
        x = x * 1664525u + 1013904223u; const int c = (int)((x >> 4) % (unsigned)N);
        double ref = 0.0;
        for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)r * K + k]) * (double)bf2f(hB[(size_t)c * K + k]);
        const double err = fabs((double)bf2f(hC[(size_t)r * N + c]) - ref) / (fabs(ref) + 1.0);
        worst = (err > worst || err != err) ? err : worst;
      }
      printf("%6d x %5d x %5d  pinned LDS-DMA loop alone, 256 x %d tiles (%d tiles = %.2f rounds of 256): %.3f ms  %.0f TF/s  (worst sampled rel. error %.1e)\n",
             M, N, K, 64 * tn, (M / 256) * (N / (64 * tn)), (M / 256) * (N / (64 * tn)) / 256.0, ms, 2.0 * M * N * K / ms / 1e9, worst);
    }
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  return 0;
}
