// gemm4w.hip — the long-K, N = 768-family bf16 GEMM of libcst_hip: 256 x 192 x 64 tile, FOUR waves (one per SIMD), wave tile 128 x 96,
// 192 accumulator registers per lane in the AGPR half of the register file, operands HBM/L2 -> LDS by buffer_load ... lds,
// v_mfma_f32_32x32x16_bf16 with swapped operands, and the issue order of a K tile pinned by hand:
//     { 2 MFMA ; one fragment read of the next k16 step ; one DMA of the next K tile }  x 6 per k16 step
// A lone wave per SIMD has no partner to cover its LDS-DMA issues (~60 cycles each among MFMAs) or its fragment reads, so each sits
// in the shadow of an MFMA pair; hipcc's own order of the same statements runs 20 % slower (DESIGN 5.1, round 4;
// tools/probes/gemm4w_pgr.hip is the stand-alone form with its measurements).  Per MFMA the wave reads 0.58 fragments from LDS (the
// 8-wave kernel of gemm8p.hip: 0.75), and N = 768 is exactly four tile columns: 31 760 rows are 500 tiles = 1.95 rounds of 256 CUs.
//
// Contract: what the step's long-K N = 768 launches need — fc2 forward (bias + dropout + residual), the dX GEMMs of fc1 / q|k|v
// (residual, live-tile stamps).  bf16, both operands k-major, unbatched, unsplit, N % 192 == 0, K % 64 == 0, no activation /
// pre-activation output / act'.  RESULTS ARE THE BITS OF gemm8p's fast path: the bias row seeds the accumulators, K tiles and k16
// steps in ascending order, ONE bf16 rounding of the accumulator (the pre-activation, as the reference's bf16 F.linear output),
// dropout on that value, residual added in fp32, one more rounding.
// LDS (144 KiB): THREE stages of the A image (32 KiB each) and two of the B image (24 KiB): inside a training update A — the
// activations, 195 MB at 31 760 x 3072 — comes from HBM, and with two stages a lone wave has one K tile (~0.8 us) of prefetch distance:
// the kernel was 7-8 % faster than the 8-wave one in a timing loop (operand resident in the 256 MB Infinity Cache) and 8-11 % slower
// inside the update; with A two K tiles ahead it is 3-8 % faster there too.  k-major images [rows][128 B], 16-byte chunk c of row r
// stored at c ^ ((r >> 1) & 7) through the DMA's per-lane SOURCE address; ds_read_b128 fragments apply the same permutation.
// Rows of A beyond M come back as zeros through the buffer descriptor's range check; their outputs are not stored.
#include "gemm_common.h"
#include <cstdlib>

namespace {
using namespace cstg;
using T = bf16_t;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;

constexpr int BM = 256, TN = 3, BN = 64 * TN, BK = 64, NT = 256;
static_assert(BM <= cstg::GEMM_MAX_BM && cstg::GEMM_MAX_BM % BM == 0, "conv0.hip (cst_conv_row_limits) sizes live frames from GEMM_MAX_BM");
constexpr int IMGA = BM * 128, IMGB = BN * 128;
constexpr int NSA = 3, NSB = 2;            // A (the activation rows: HBM-cold inside an update) rides three stages, B (the weight) two
constexpr int BOFF = NSA * IMGA;           // LDS: A stages | B stages
constexpr int GA = BM / 32, GB = BN / 32, G = GA + GB;  // 1-KiB DMA pieces per wave and K tile: 8 + 6
constexpr int NSLOT = 2 * TN, NFR = 4 + TN;             // MFMA pairs per k16 step, fragments per k16 step
constexpr int ERS = 2 * (BN / 2) + 16;                  // epilogue image (bf16, one wave's 128 x 96 block): row stride in bytes
constexpr int LDS_BYTES = NSA * IMGA + NSB * IMGB > 4 * 128 * ERS ? NSA * IMGA + NSB * IMGB : 4 * 128 * ERS;  // 144 KiB

__global__ __launch_bounds__(NT) void gemm4w_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane & 31, hi = lane >> 5;
  // XCD-aware grouped tile order (the map of gemm.hip)
  const int ntiles = p.tiles_m * p.tiles_n;
  int id = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = id % 8, loc = id / 8;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int GROUP_M = p.group_m;
  const int per_group = GROUP_M * p.tiles_n;
  const int grp = id / per_group, rem = id % per_group;
  const int gm0 = grp * GROUP_M;
  const int gsz = (p.tiles_m - gm0 < GROUP_M) ? (p.tiles_m - gm0) : GROUP_M;
  const int tm = gm0 + rem % gsz, tn = rem / gsz;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int mrem = (int)(p.M - m0 < BM ? p.M - m0 : BM);

  int nkt = (int)(p.K / BK);
  if (p.m_live) {  // every 64-row block of this tile's A rows stamped dead: no K loop, the epilogue runs on the seeded accumulators
    typedef const __attribute__((address_space(4))) uint32_t* cptr_t;
    cptr_t ml = (cptr_t)p.m_live;
    const int t0 = (int)(m0 >> 6), tnn = (int)((p.M + 63) >> 6);
    bool live = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) live |= (t0 + i < tnn) && ml[t0 + i < tnn ? t0 + i : t0] == p.m_epoch;
    if (!live) nkt = 0;
  }

  const int row0 = 8 * wave + (lane >> 3);
  const unsigned lda2 = (unsigned)(p.lda * 2), ldb2 = (unsigned)(p.ldb * 2);
  const unsigned chs = (unsigned)(((lane & 7) ^ ((row0 >> 1) & 7)) << 4);
  const unsigned voffa = (unsigned)row0 * lda2 + chs, voffb = (unsigned)row0 * ldb2 + chs;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.A + m0 * p.lda), (short)0,
                                                                         (int)((unsigned)(mrem - 1) * lda2 + (unsigned)(p.K * 2)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.B + n0 * p.ldb), (short)0,
                                                                         (int)((unsigned)(BN - 1) * ldb2 + (unsigned)(p.K * 2)), 0x00020000);
  auto dma_a = [&](int i, int kt, int stg) {  // A rows 8 (wave + 4 i) .. of K tile kt -> A stage stg
    char* dst = smem + stg * IMGA + (wave + 4 * i) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, (int)voffa,
                                             (int)((unsigned)(32 * i) * lda2 + (unsigned)kt * 128u), 0, 0);
  };
  auto dma_b = [&](int i, int kt, int stg) {
    char* dst = smem + BOFF + stg * IMGB + (wave + 4 * i) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)dst, 16, (int)voffb,
                                             (int)((unsigned)(32 * i) * ldb2 + (unsigned)kt * 128u), 0, 0);
  };
  const int swr = (lrow >> 1) & 7;
  const unsigned fa0 = (unsigned)((wm * 128 + lrow) * 128), fb0 = (unsigned)(BOFF + (wn * 32 * TN + lrow) * 128);
  unsigned fch[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fch[kk] = (unsigned)(((2 * kk + hi) ^ swr) << 4);

  // acc[i][j][r] (swapped operands: D = B-frag x A-frag): output row m0 + wm*128 + i*32 + lrow, column n0 + wn*96 + j*32 + 8*(r/4) + 4*hi + r%4
  f32x16 acc[4][TN];
  if (p.bias_mode == CST_BIAS_COL) {  // the bias row seeds the accumulators (gemm8p: bias_in_acc)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u32x2 raw = *reinterpret_cast<const u32x2*>((const T*)p.bias + n0 + wn * 32 * TN + j * 32 + 8 * q + 4 * hi);
        const float b0 = __uint_as_float(raw[0] << 16), b1 = __uint_as_float(raw[0] & 0xffff0000u);
        const float b2 = __uint_as_float(raw[1] << 16), b3 = __uint_as_float(raw[1] & 0xffff0000u);
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i][j][4 * q] = b0; acc[i][j][4 * q + 1] = b1; acc[i][j][4 * q + 2] = b2; acc[i][j][4 * q + 3] = b3; }
      }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  }

  // Ring: during K tile kt the wave issues B(kt + 1) (slots 0..5) and then A(kt + 2) (slots 6..13).  At the head of tile kt the newest
  // eight pieces in flight are A(kt + 1): vmcnt(8) retires B(kt) and A(kt) and leaves them travelling — two K tiles of flight time
  // for the cold operand.  WAR: A stage (kt + 2) % 3 and B stage (kt + 1) % 2 were last read during tile kt - 1, behind the barrier.
  if (nkt > 0) {
#pragma unroll
    for (int i = 0; i < GA; ++i) dma_a(i, 0, 0);
#pragma unroll
    for (int i = 0; i < GB; ++i) dma_b(i, 0, 0);
    if (nkt > 1) {
#pragma unroll
      for (int i = 0; i < GA; ++i) dma_a(i, 1, 1);
    }
  }
  bf16x8 fa[2][4], fb[2][TN];
  int sa = 0;  // A stage of tile kt
  for (int kt = 0; kt < nkt; ++kt) {
    const int sb = kt & 1;
    if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");  // (A(kt + 1) may still be on its way)
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* abase = smem + sa * IMGA;
    const char* bbase = smem + sb * IMGB;
    auto read_one = [&](int f, int kk, int nb) {
      if (f < 4) fa[nb][f] = *reinterpret_cast<const bf16x8*>(abase + fa0 + f * 4096 + fch[kk]);
      else fb[nb][f - 4] = *reinterpret_cast<const bf16x8*>(bbase + fb0 + (f - 4) * 4096 + fch[kk]);
    };
#pragma unroll
    for (int f = 0; f < NFR; ++f) read_one(f, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    const bool more_b = kt + 1 < nkt, more_a = kt + 2 < nkt;
    const int sa2 = sa == 0 ? 2 : sa - 1;  // (kt + 2) % 3
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
          if (sl == 0) read_one(NSLOT, kk + 1, (kk + 1) & 1);  // seven fragments, six slots
        }
        {
          const int d = kk * NSLOT + sl;
          if (d < GB) { if (more_b) dma_b(d, kt + 1, sb ^ 1); }
          else if (d < G) { if (more_a) dma_a(d - GB, kt + 2, sa2); }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    sa = sa == 2 ? 0 : sa + 1;
  }
  static_assert(NFR == NSLOT + 1, "the slot plan above places exactly one extra fragment read");

  // ---- epilogue: z = bf16(acc * alpha) into this wave's image; read back row-contiguously: dropout, residual, one more rounding.
  //      Global loads and global stores never share a loop (with a store pending, waiting for a load costs a full vmcnt(0), i.e. a
  //      store round trip): the wave's 24 residual vectors are requested in ONE burst before the image is written, combined into
  //      the image in place, and the stores run in a loop of their own. ----
  constexpr int CPR = BN / 2 / 8;         // 16-byte pieces per row of the wave's 128 x 96 block: 12
  constexpr int NPC = 128 * CPR / 64;     // pieces per lane: 24
  const int64_t rbase_m = m0 + wm * 128, cbase_n = n0 + wn * (BN / 2);
  const T* resid = (const T*)p.resid;
  u32x4 exq[NPC];
  if (resid) {
#pragma unroll
    for (int t = 0; t < NPC; ++t) {
      const int v = t * 64 + lane, rl = v / CPR, cl = (v % CPR) * 8;
      int64_t row = rbase_m + rl;
      row = row < p.M ? row : p.M - 1;
      exq[t] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(resid + row * p.ld_resid + cbase_n + cl));
    }
  }
  __syncthreads();
  {
    char* img = smem + wave * (128 * ERS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
          bf16x4_t pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) pk[e] = static_cast<__bf16>(acc[i][j][4 * q + e] * p.alpha);
          *reinterpret_cast<u32x2*>(img + (i * 32 + lrow) * ERS + (j * 32 + 8 * q + 4 * hi) * 2) = __builtin_bit_cast(u32x2, pk);
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave reads back only what it wrote itself
    if (p.drop_thr || resid) {
#pragma unroll
      for (int t = 0; t < NPC; ++t) {
        const int v = t * 64 + lane, rl = v / CPR, cl = (v % CPR) * 8;
        const int64_t row = rbase_m + rl, col = cbase_n + cl;
        const u32x4 zraw = *reinterpret_cast<const u32x4*>(img + rl * ERS + cl * 2);
        float x[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x[2 * e] = __uint_as_float(zraw[e] << 16);
          x[2 * e + 1] = __uint_as_float(zraw[e] & 0xffff0000u);
        }
        if (p.drop_thr) cst_drop8(x, p.drop_key, (uint64_t)((row + p.drop_row0) * p.N + col), p.drop_thr, p.drop_scale);
        if (resid) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x[2 * e] += __uint_as_float(exq[t][e] << 16);
            x[2 * e + 1] += __uint_as_float(exq[t][e] & 0xffff0000u);
          }
        }
        bf16x8 ob;
#pragma unroll
        for (int e = 0; e < 8; ++e) ob[e] = static_cast<__bf16>(x[e]);
        *reinterpret_cast<u32x4*>(img + rl * ERS + cl * 2) = __builtin_bit_cast(u32x4, ob);  // (this lane's own piece: no hazard)
      }
    }
#pragma unroll 8
    for (int t = 0; t < NPC; ++t) {
      const int v = t * 64 + lane, rl = v / CPR, cl = (v % CPR) * 8;
      const int64_t row = rbase_m + rl;
      if (row >= p.M) continue;
      const u32x4 o = *reinterpret_cast<const u32x4*>(img + rl * ERS + cl * 2);
      __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>((T*)p.C + row * p.ldc + cbase_n + cl));
    }
  }
}
}  // namespace

bool cst_gemm4w_supported(const cstg::GemmParams& p, bool ak, bool bk, int64_t nbatch) {
  if (!ak || !bk || nbatch != 1 || p.splits != 1 || p.c_f32 || !p.vec_epi) return false;
  if (p.N % BN || p.K % BK || p.K < 1536 || p.M < 256) return false;
  if (p.a_seg || p.b_seg || p.act || p.aux_out || p.dact || p.m_len) return false;
  if (p.bias_mode != CST_BIAS_NONE && (p.bias_mode != CST_BIAS_COL || p.alpha != 1.0f)) return false;
  if (p.lda * 2 * 256 + p.K * 2 >= (1ll << 31) || p.ldb * 2 * 256 >= (1ll << 31)) return false;  // 32-bit offsets inside a tile's descriptor
  // One tile per workgroup, so the launch costs whole rounds of 256 CUs; the persistent 8-wave kernel (half-height tail items, claimed
  // work) degrades more gently.  Measured with the residual epilogue at K = 1536 / 2304 / 3072 (tools/bench_gemm4w_sweep.py): this
  // kernel wins by 5-9 % at 1.0, 1.47-2.0 and 2.7-3.0 rounds, loses by 20-40 % just above a whole round and from ~3.4 rounds on.
  const int64_t tiles = cst_ceil_div(p.M, BM) * (p.N / BN);
  const int64_t r = tiles % 256;
  // (third round: only when it is at least two thirds full — 752 tiles, the all-30 s batch: 88.5-89.4 ms per update against 89.0-91.0)
  static const int64_t max_tiles = [] { const char* e = getenv("CST_GEMM_4W_MAXTILES"); return e ? (int64_t)atoll(e) : (int64_t)768; }();
  return tiles >= 256 && tiles <= max_tiles && (r == 0 || r >= (tiles > 512 ? 170 : 115));
}

int cst_gemm4w_launch(cstg::GemmParams p, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  p.tiles_m = (int)cst_ceil_div(p.M, BM);
  p.tiles_n = (int)(p.N / BN);
  static const int group_m = [] { const char* e = getenv("CST_GEMM_4W_GROUP_M"); const int g = e ? atoi(e) : 8; return g > 0 ? g : 8; }();
  p.group_m = group_m;  // m tiles per group of the tile walk (an XCD's concurrent workgroups cover group_m x tiles_n tiles)
  hipLaunchKernelGGL(gemm4w_kernel, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(NT), LDS_BYTES, s, p);
  return cst_check_launch("cst_gemm (4-wave)");
}
