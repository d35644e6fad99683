// gemm.hip — MFMA GEMM for gfx950 (wave64, v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32).
//
// One kernel family serves every dense contraction of the Chimera hot path (see include/cst.h):
// Linear fwd (A k-major, B k-major), dX (A k-major, B mn-major), dW (A mn-major, B mn-major),
// the wav2vec2 conv stack / subsampler as implicit GEMM (lda < K: overlapping channels-last rows)
// and the grouped pos-conv (segmented K addressing + two batch levels).
//
// Tile: 128 x 128 x BK (BK = 64 bf16 / 32 f32 -> always 8 16-byte vectors per k-major row),
// 256 threads = 4 waves in 2x2, each wave 64x64 = 2x2 MFMA 32x32 accumulators (64 acc VGPRs).
// Global -> registers -> LDS staging, double-buffered LDS, one barrier per K tile; the next
// tile's global loads are issued before the current tile's MFMAs (latency hidden under MFMA).
// LDS images: k-major tiles [128][BK + VEC] (16-B row pad -> conflict-free ds_read_b128),
// mn-major tiles [BK][128 + VEC] read with per-element strided gathers (half-wave = 32
// consecutive mn -> conflict-free).
// Workgroup -> tile map is XCD-aware (bijective remap so each XCD's L2 sees a contiguous
// band of tiles sharing A rows).
#include "gemm_common.h"
#include <vector>
#include <stdlib.h>
#include <string>
#include <type_traits>

namespace {
using namespace cstg;

// tile configurations: Cfg<BM, BN, WM, WN>: WM x WN waves, each (BM/WM) x (BN/WN) = (TM*32) x (TN*32)
template <int BM_, int BN_, int WM_, int WN_, int KV_ = 8>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, NT = 64 * WM_ * WN_, TM = BM_ / WM_ / 32, TN = BN_ / WN_ / 32;
  static constexpr int KV = KV_;  // 16-byte vectors per k-major tile row: BK = KV * VEC (64 or 32 bf16)
  static_assert(BM_ <= GEMM_MAX_BM && GEMM_MAX_BM % BM_ == 0, "conv0.hip (cst_conv_row_limits) sizes live frames from GEMM_MAX_BM");
};
using CfgSmall = Cfg<128, 128, 2, 2>;   // 4 waves, 2 blocks/CU: small / skinny problems, grouped conv
// 16 waves (wave tile 64 x 64), 1 block/CU, 4 waves/SIMD: half the L2->LDS bytes per FLOP of CfgSmall.  Measured
// alternatives on MI355X at M=48k, K/N in {768, 3072} (tools/bench_kernels.py): 8 waves x (128 x 64) 660-690 TF/s (VGPR-capped,
// 2 waves/SIMD); 256 x 128 x 64 8 waves with a 3-stage DMA ring 590; 256 x 128 x 32 4 waves 2 blocks/CU 590-660; this one 670-760.
// decode-time problems (M = batch x beam <= 256 rows, k-major operands): 64 x 64 tiles, 4 waves of one 32 x 32 MFMA tile each, on a
// 4-stage DMA ring (3 K tiles = 48 KiB in flight per workgroup, 2 workgroups per CU).  The step of an incremental decoder is
// bound by how many bytes each CU keeps in flight, not by MFMA: with 128 x 128 tiles a 160 x 1024 x 1024 projection is 16
// workgroups behind a 2-stage ring (13 us + a 5 us split-K reduce, or 33 us unsplit); this configuration runs it as 48.
using CfgSkinny = Cfg<64, 64, 2, 2>;
// narrow problems (wav2vec2 pos_conv: per (utterance, group) 1499 x 48 x 6144 forward / dX, 48 x 6144 x 1499 dW): a 128 x 128 tile
// spends 62 % of its MFMAs on columns (rows) that do not exist; these keep 75 %
using CfgNarrowN = Cfg<128, 64, 2, 2>;
using CfgNarrowM = Cfg<64, 128, 2, 2>;
using CfgLarge = Cfg<256, 256, 4, 4>;   // 8 waves (wave tile 128 x 64), 1 block/CU: half the L2->LDS bytes per FLOP

// CS: the column sums of A as a by-product (cst_gemm_desc.colsum); its own instantiation — eight more live VGPRs do not fit the
// 16-wave configuration (128 VGPRs: 213 spills inside the K loop), so only the 4-wave configurations are built with it and the
// launcher runs the two-launch column sum for the others.
template <typename T, bool A_KMAJOR, bool B_KMAJOR, bool SEG, typename C, bool CS = false>
__global__ __launch_bounds__(C::NT) void gemm_kernel(GemmParams p) {
  constexpr int BM = C::BM, BN = C::BN, NTHREADS = C::NT, TM = C::TM, TN = C::TN;
  constexpr int VEC = DT<T>::VEC;
  constexpr int KV = C::KV, BK = KV * VEC;
  constexpr int LDK = BK + VEC;    // k-major LDS row stride (elements)
  // mn-major LDS row stride: bf16 rows are padded to (16 mod 64) dwords -> conflict-free ds_read_b64_tr_b16
  constexpr int LDMA = sizeof(T) == 2 ? BM + 32 : BM + VEC;
  constexpr int LDMB = sizeof(T) == 2 ? BN + 32 : BN + VEC;
  constexpr int A_ELEMS = A_KMAJOR ? BM * LDK : BK * LDMA;
  constexpr int B_ELEMS = B_KMAJOR ? BN * LDK : BK * LDMB;
  constexpr int MVA = BM / VEC, MVB = BN / VEC;          // vectors per mn-major row
  constexpr int NVA = BM * KV / NTHREADS, NVB = BN * KV / NTHREADS;  // 16-byte vectors per thread per K tile
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  // NOTE: stage addresses are formed as smem + cur * STAGE (never by selecting between two pointers): a pointer select
  // loses the LDS address space and hipcc then emits flat_load/flat_store, whose vmcnt waits also drain the global prefetch.
  constexpr int STAGE = A_ELEMS + B_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WN, wn = wave % C::WN;

  // ---- XCD-aware tile id (bijective) ----
  const int ntiles = p.tiles_m * p.tiles_n;
  int tm, tn, z;
  if (p.split_order) {
    // split-K, unbatched (the dW GEMMs: 12 x 3 tiles x 7 splits = 252 workgroups, one round): the work items (split, m tile,
    // n tile) are laid out split-major / n-fastest and each XCD takes one CONTIGUOUS run of them, so the ~32 workgroups that share
    // an XCD's 4 MiB L2 work on one or two K slices and neighbouring tiles: an A slice is fetched once for its 3 n tiles and a B
    // slice once for the ~11 m tiles of the run.  With the plain (x, z) grid order the tiles of one split were dealt round-robin
    // over all 8 XCDs and every operand slice was fetched by several L2s: 924 MB at the fabric for 370 MB of operands, at 5.9 TB/s.
    const int total = ntiles * p.splits, L = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.z;
    const int q = total / 8, r = total % 8, xcd = L % 8, loc = L / 8;
    const int item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    z = item / ntiles;
    const int t = item % ntiles;
    tm = t / p.tiles_n;
    tn = t % p.tiles_n;
  } else {
    int id = blockIdx.x;
    {
      const int q = ntiles / 8, r = ntiles % 8, xcd = id % 8, loc = id / 8;
      id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    // grouped order inside each XCD's contiguous id range: GROUP_M row-tiles x all column-tiles, walked column-major, so the
    // ~64 tiles an XCD runs concurrently form a compact 8 x 8 patch (8 A panels + 8 B panels ~ 3 MB at K=768: fits the 4 MB L2)
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * p.tiles_n;
    const int grp = id / per_group, rem = id % per_group;
    const int gm0 = grp * GROUP_M;
    const int gsz = (p.tiles_m - gm0 < GROUP_M) ? (p.tiles_m - gm0) : GROUP_M;
    tm = gm0 + rem % gsz;
    tn = rem / gsz;
    z = blockIdx.z;
  }
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int split = z % p.splits;
  const int64_t bidx = z / p.splits;
  const int64_t b0 = bidx / p.batch1, b1 = bidx % p.batch1;
  const T* A = (const T*)p.A + b0 * p.sa0 + b1 * p.sa1;
  const T* B = (const T*)p.B + b0 * p.sb0 + b1 * p.sb1;
  const int64_t cofs = b0 * p.sc0 + b1 * p.sc1;
  const int64_t bofs = b0 * p.sbias0 + b1 * p.sbias1;

  int ktiles = (int)((p.K + BK - 1) / BK);
  const int kfull = (int)(p.K / BK);  // tiles [0, kfull) need no K bound check
  if (p.k_len) {  // this batch's A is zero from row k_len[b0] on: the K loop (and the split-K partition) ends there
    typedef const __attribute__((address_space(4))) int32_t* cptr_t;
    const int kl = ((cptr_t)p.k_len)[b0];
    const int live = kl <= 0 ? 0 : (kl + BK - 1) / BK;
    ktiles = live < ktiles ? live : ktiles;
  }
  const int per = (ktiles + p.splits - 1) / p.splits;
  const int kt0 = split * per;
  const int kt1 = (kt0 + per < ktiles) ? kt0 + per : ktiles;

  // ---- per-thread staging state: 4 x 16-byte vectors of A and of B per K tile.  Rows / mn positions beyond the
  //      matrix are CLAMPED to a valid address (their products only reach accumulator rows/cols that are never stored),
  //      so the steady-state loop is branch-free pointer bumps; only the K tail tile is predicated. ----
  u32x4 ra[NVA], rb[NVB];
  const T* pa[NVA];
  const T* pb[NVB];
  int ka[NVA], kb[NVB];          // k index this vector covers in tile kt0 (k-major: of its first element; mn-major: its k row)
  int aq[NVA], ar[NVA], bq[NVB], br[NVB];  // SEG only: running (segment, offset) of the k index
#pragma unroll
  for (int i = 0; i < NVA; ++i) {
    const int v = tid + NTHREADS * i;
    if (A_KMAJOR) {
      int64_t row = m0 + v / KV;
      row = row < p.M ? row : p.M - 1;
      ka[i] = kt0 * BK + (v % KV) * VEC;
      if (SEG && p.a_seg) { aq[i] = ka[i] / (int)p.a_seg; ar[i] = ka[i] % (int)p.a_seg; pa[i] = A + row * p.lda; }
      else { aq[i] = 0; ar[i] = 0; pa[i] = A + row * p.lda + ka[i]; }
    } else {
      int64_t m = m0 + (int64_t)(v % MVA) * VEC;
      m = m < p.M ? m : p.M - VEC;
      ka[i] = kt0 * BK + v / MVA;
      aq[i] = 0; ar[i] = 0;
      pa[i] = A + (int64_t)ka[i] * p.lda + segaddr(m, p.a_seg, p.a_seg_stride);
    }
  }
#pragma unroll
  for (int i = 0; i < NVB; ++i) {
    const int v = tid + NTHREADS * i;
    if (B_KMAJOR) {
      int64_t row = n0 + v / KV;
      row = row < p.N ? row : p.N - 1;
      kb[i] = kt0 * BK + (v % KV) * VEC;
      if (SEG && p.b_seg) { bq[i] = kb[i] / (int)p.b_seg; br[i] = kb[i] % (int)p.b_seg; pb[i] = B + row * p.ldb; }
      else { bq[i] = 0; br[i] = 0; pb[i] = B + row * p.ldb + kb[i]; }
    } else {
      int64_t n = n0 + (int64_t)(v % MVB) * VEC;
      n = n < p.N ? n : p.N - VEC;
      kb[i] = kt0 * BK + v / MVB;
      bq[i] = 0; br[i] = 0;
      pb[i] = B + (int64_t)kb[i] * p.ldb + segaddr(n, p.b_seg, p.b_seg_stride);
    }
  }
  const int64_t step_a = A_KMAJOR ? (int64_t)BK : (int64_t)BK * p.lda;
  const int64_t step_b = B_KMAJOR ? (int64_t)BK : (int64_t)BK * p.ldb;

  // TAIL = true: the tile may cross K -> vectors at k >= K are replaced by zeros (K % VEC == 0 for k-major operands)
  auto load_tile = [&](auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const u32x4 zero = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const T* qa = pa[i];
      if (SEG && A_KMAJOR && p.a_seg) qa = pa[i] + (int64_t)aq[i] * p.a_seg_stride + ar[i];
      if (TAIL) {
        const bool oka = ka[i] < p.K;
        const u32x4 va = *reinterpret_cast<const u32x4*>(oka ? qa : A);
        ra[i] = oka ? va : zero;
      } else {
        ra[i] = *reinterpret_cast<const u32x4*>(qa);
      }
      ka[i] += BK;
      if (SEG && A_KMAJOR && p.a_seg) { ar[i] += BK; while (ar[i] >= (int)p.a_seg) { ar[i] -= (int)p.a_seg; ++aq[i]; } }
      else pa[i] += step_a;
    }
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const T* qb = pb[i];
      if (SEG && B_KMAJOR && p.b_seg) qb = pb[i] + (int64_t)bq[i] * p.b_seg_stride + br[i];
      if (TAIL) {
        const bool okb = kb[i] < p.K;
        const u32x4 vb = *reinterpret_cast<const u32x4*>(okb ? qb : B);
        rb[i] = okb ? vb : zero;
      } else {
        rb[i] = *reinterpret_cast<const u32x4*>(qb);
      }
      kb[i] += BK;
      if (SEG && B_KMAJOR && p.b_seg) { br[i] += BK; while (br[i] >= (int)p.b_seg) { br[i] -= (int)p.b_seg; ++bq[i]; } }
      else pb[i] += step_b;
    }
  };
  auto load_any = [&](int kt) {
    if (kt < kfull) load_tile(std::false_type{});
    else load_tile(std::true_type{});
  };
  // column sums of A (cst_gemm_desc.colsum: the bias gradient next to dW = dY^T X): the workgroups of output-tile column 0 add up
  // the A vectors they stage anyway.  Thread t always stages the same VEC columns (NTHREADS % MVA == 0), so VEC running sums per
  // thread cover its k rows of every K tile; the threads sharing a column group are added through LDS after the K loop.
  static_assert(!CS || (!A_KMAJOR && NTHREADS % MVA == 0), "a thread must stage the same columns of an mn-major A in every vector");
  const bool do_cs = CS && tn == 0;
  float cs[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) cs[e] = 0.0f;
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int v = tid + NTHREADS * i;
      T* da = smem + buf * STAGE + (A_KMAJOR ? (v / KV) * LDK + (v % KV) * VEC : (v / MVA) * LDMA + (v % MVA) * VEC);
      *reinterpret_cast<u32x4*>(da) = ra[i];
      if (CS && do_cs) {
        if (sizeof(T) == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            cs[2 * e] += __uint_as_float(ra[i][e] << 16);
            cs[2 * e + 1] += __uint_as_float(ra[i][e] & 0xffff0000u);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) cs[e] += __uint_as_float(ra[i][e]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const int v = tid + NTHREADS * i;
      T* db = smem + buf * STAGE + A_ELEMS + (B_KMAJOR ? (v / KV) * LDK + (v % KV) * VEC : (v / MVB) * LDMB + (v % MVB) * VEC);
      *reinterpret_cast<u32x4*>(db) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // ---- live K tiles (p.k_live: the weight-gradient GEMMs reduce over tokens, and dY is exactly zero at padded frames): wave 0
  //      compacts the indices of the live tiles into LDS; the K loop walks that list and the staging pointers jump over the dead
  //      tiles (each would have added 0 * b to every accumulator).  With split-K the LIVE tiles — not the K range — are dealt out
  //      evenly: batches are sorted by length, so equal K ranges would leave the first split (the longest utterances) with
  //      nearly all of its tiles and the launch no faster.  (The fp32 summation order across splits therefore differs from the
  //      unstamped launch; with one split the result is bit-identical to it.) ----
  constexpr int KLIST_MAX = 1000;  // (4 000 B of static LDS: the launch attribute leaves 4 096; the positional convolution's weight gradient walks 812 K tiles)
  __shared__ int klist[KLIST_MAX];
  __shared__ int klist_n;
  const bool kskip = !SEG && BK == 64 && p.k_live != nullptr && per <= KLIST_MAX;
  int nk = kt1 > kt0 ? kt1 - kt0 : 0;
  if (kskip) {
    if (wave == 0) {
      int total = 0;
      for (int base = 0; base < ktiles; base += 64) {
        const int kt = base + lane;
        total += __builtin_popcountll(__builtin_amdgcn_ballot_w64(kt < ktiles && p.k_live[kt] == p.k_epoch));
      }
      const int lo = (int)(((int64_t)total * split) / p.splits), hi = (int)(((int64_t)total * (split + 1)) / p.splits);
      int seen = 0, n = 0;
      for (int base = 0; base < ktiles && seen < hi; base += 64) {
        const int kt = base + lane;
        const bool live = kt < ktiles && p.k_live[kt] == p.k_epoch;
        const uint64_t m = __builtin_amdgcn_ballot_w64(live);
        const int idx = seen + __builtin_popcountll(m & ((1ull << lane) - 1ull));  // rank of this lane's tile among the live ones
        if (live && idx >= lo && idx < hi) klist[idx - lo] = kt;
        seen += __builtin_popcountll(m);
      }
      n = hi - lo;  // <= ceil(total / splits) <= per <= KLIST_MAX
      if (lane == 0) klist_n = n;
    }
    __syncthreads();
    nk = klist_n;
  }
  if (p.m_live) {  // an output tile all of whose A rows are stamped dead: accumulators stay exactly 0, only the epilogue runs
    typedef const __attribute__((address_space(4))) uint32_t* cptr_t;  // scalar loads (the stamps were written by an earlier kernel)
    cptr_t ml = (cptr_t)p.m_live;
    const int t0 = (int)(m0 >> 6), t1 = (int)((((m0 + BM < p.M) ? m0 + BM : p.M) + 63) >> 6);
    bool live = false;
    for (int t = t0; t < t1; ++t) live |= ml[t] == p.m_epoch;
    if (!live) nk = 0;
  }
  if (p.m_len) {  // this batch's rows from m_len[b0] on are all zero and unread: tiles behind it keep zero accumulators
    typedef const __attribute__((address_space(4))) int32_t* cptr_t;
    if (m0 >= (int64_t)((cptr_t)p.m_len)[b0]) nk = 0;
  }
  int pos = kt0;  // the K tile the staging pointers stand at
  auto fetch = [&](int kt) {
    const int d = kt - pos;
    if (d) {
#pragma unroll
      for (int i = 0; i < NVA; ++i) { pa[i] += (int64_t)d * step_a; ka[i] += d * BK; }
#pragma unroll
      for (int i = 0; i < NVB; ++i) { pb[i] += (int64_t)d * step_b; kb[i] += d * BK; }
    }
    load_any(kt);
    pos = kt + 1;
  };
  auto tile_at = [&](int it) { return kskip ? klist[it] : kt0 + it; };

  if (nk > 0) {
    fetch(tile_at(0));
    store_tile(0);
  }
  __syncthreads();
  int cur = 0;
  const int lrow = lane & 31, lk = 8 * (lane >> 5);
  constexpr int WROWS = TM * 32, WCOLS = TN * 32;
  for (int it = 0; it < nk; ++it) {
    const bool more = it + 1 < nk;
    if (more) fetch(tile_at(it + 1));
    const T* sa = smem + cur * STAGE;
    const T* sb = sa + A_ELEMS;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      Frag<T> fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (A_KMAJOR) frag_load_contig(fa[i], sa + (wm * WROWS + i * 32 + lrow) * LDK + kk + lk);
        else frag_load_tr(fa[i], sa, LDMA, wm * WROWS + i * 32, kk + lk, kk + lk + 4, lane);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (B_KMAJOR) frag_load_contig(fb[j], sb + (wn * WCOLS + j * 32 + lrow) * LDK + kk + lk);
        else frag_load_tr(fb[j], sb, LDMB, wn * WCOLS + j * 32, kk + lk, kk + lk + 4, lane);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma16(acc[i][j], fa[i], fb[j]);
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  if (CS && do_cs) {  // (block-uniform) column sums: threads with equal tid % MVA hold partial sums of the same VEC columns
    float* red = reinterpret_cast<float*>(smem_raw);  // [NTHREADS / MVA][BM]: the operand stages are free (barrier at the loop's end)
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[(tid / MVA) * BM + (tid % MVA) * VEC + e] = cs[e];
    __syncthreads();
    if (tid < BM && m0 + tid < p.M) {
      float tot = 0.0f;
      for (int r = 0; r < NTHREADS / MVA; ++r) tot += red[r * BM + tid];  // fixed order
      if (p.splits > 1) p.colsum_ws[((int64_t)bidx * p.splits + split) * p.M + m0 + tid] = tot;
      else DT<T>::st((T*)p.colsum + bidx * p.M + m0 + tid, tot);
    }
  }

  // ---- epilogue: accumulators -> LDS (fp32, 64 rows x (BN+4) per pass) -> row-contiguous 8-column vectors -> 16-byte
  //      global stores.  Each pass covers tile rows [64 p, 64 p + 64); a wave contributes the 32-row MFMA tiles inside it.
  constexpr int LDC = BN + 4;
  constexpr int VPR = BN / 8;                       // 8-column vectors per tile row
  constexpr int NPASS = BM / 64;
  constexpr int ITERS = 64 * VPR / NTHREADS;        // vectors per thread per pass
  float* stage = reinterpret_cast<float*>(smem_raw);
  float* wsp = p.splits > 1 ? p.ws + ((int64_t)z) * p.M * p.N : nullptr;
  const bool ws_vec = (p.N % 4) == 0;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    __syncthreads();  // operand tiles (pass 0) / previous pass's staging fully consumed
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int trow = wm * WROWS + i * 32;  // tile-relative first row of this MFMA tile
      if (trow / 64 == ps) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            stage[(trow - 64 * ps + acc_row(r, lane)) * LDC + wn * WCOLS + j * 32 + lrow] = acc[i][j][r];
      }
    }
    __syncthreads();
#pragma unroll 2
    for (int it = 0; it < ITERS; ++it) {
      const int v = tid + NTHREADS * it;
      const int rl = v / VPR, cl = (v % VPR) * 8;
      const int64_t row = m0 + 64 * ps + rl, col = n0 + cl;
      if (row >= p.M || col >= p.N) continue;
      float x[8];
      {
        const f32x4 a = *reinterpret_cast<const f32x4*>(stage + rl * LDC + cl);
        const f32x4 b = *reinterpret_cast<const f32x4*>(stage + rl * LDC + cl + 4);
        x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
      }
      const bool full = col + 8 <= p.N;
      if (wsp) {  // split-K partial slab (raw fp32 sums; epilogue applied by the reduce kernel)
        if (full && ws_vec) store8(wsp + row * p.N + col, x);
        else
          for (int e = 0; e < 8 && col + e < p.N; ++e) wsp[row * p.N + col + e] = x[e];
      } else if (full && p.vec_epi) {
        epilogue_store8<T>(p, cofs, bofs, row, col, x);
      } else {
        for (int e = 0; e < 8 && col + e < p.N; ++e) epilogue_store<T>(p, cofs, bofs, row, col + e, x[e]);
      }
    }
  }
}

// =====================================================================================================================
// gemm_glds_kernel — same contract as gemm_kernel, operand tiles fetched with global_load_lds (16 B per lane straight into
// LDS: no staging VGPRs, no ds_write pass), NS-stage ring with counted vmcnt, one raw s_barrier per K tile.
// LDS images are UNPADDED (the DMA writes wave-uniform-base + lane*16) and XOR-swizzled through the per-lane SOURCE
// address; fragment reads apply the same permutation (cdna guide rule 21):
//   k-major tile  [rows][128 B]:  16-B chunk c of row r lives at chunk  c ^ ((r >> 1) & 7)   -> ds_read_b128 conflict-free
//   mn-major tile [BK][W elems] (bf16): chunk c of k-row k lives at     c ^ (4 * (k & 3))     -> ds_read_b64_tr_b16 conflict-free
// Used for every non-segmented problem; the K tail tile (K % BK != 0) is staged synchronously with zero fill.
// =====================================================================================================================
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T, bool A_KMAJOR, bool B_KMAJOR, typename C, int NS>
__global__ __launch_bounds__(C::NT) void gemm_glds_kernel(GemmParams p) {
  constexpr int BM = C::BM, BN = C::BN, NTHREADS = C::NT, TM = C::TM, TN = C::TN, NWAVES = C::NT / 64;
  constexpr int ES = sizeof(T), VEC = DT<T>::VEC, BK = 128 / ES;
  constexpr int A_BYTES = A_KMAJOR ? BM * 128 : BK * BM * ES;
  constexpr int B_BYTES = B_KMAJOR ? BN * 128 : BK * BN * ES;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int GA = A_BYTES / 1024 / NWAVES, GB = B_BYTES / 1024 / NWAVES;  // 1-KiB DMA groups per wave per tile
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int ntiles = p.tiles_m * p.tiles_n;
  int id = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = id % 8, loc = id / 8;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  constexpr int GROUP_M = 8;
  int tm, tn;
  {
    const int per_group = GROUP_M * p.tiles_n;
    const int grp = id / per_group, rem = id % per_group;
    const int gm0 = grp * GROUP_M;
    const int gsz = (p.tiles_m - gm0 < GROUP_M) ? (p.tiles_m - gm0) : GROUP_M;
    tm = gm0 + rem % gsz;
    tn = rem / gsz;
  }
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int z = blockIdx.z;
  const int split = z % p.splits;
  const int64_t bidx = z / p.splits;
  const int64_t b0 = bidx / p.batch1, b1 = bidx % p.batch1;
  const T* A = (const T*)p.A + b0 * p.sa0 + b1 * p.sa1;
  const T* B = (const T*)p.B + b0 * p.sb0 + b1 * p.sb1;
  const int64_t cofs = b0 * p.sc0 + b1 * p.sc1;
  const int64_t bofs = b0 * p.sbias0 + b1 * p.sbias1;
  const int ktiles = (int)((p.K + BK - 1) / BK);
  const int kfull = (int)(p.K / BK);
  const int per = (ktiles + p.splits - 1) / p.splits;
  const int kt0 = split * per;
  const int kt1 = (kt0 + per < ktiles) ? kt0 + per : ktiles;
  int nt = kt1 - kt0;
  if (p.m_len) {  // this batch's rows from m_len[b0] on are all zero and unread: tiles behind it keep zero accumulators
    typedef const __attribute__((address_space(4))) int32_t* cptr_t;
    if (m0 >= (int64_t)((cptr_t)p.m_len)[b0]) nt = 0;
  }

  // ---- per-lane DMA sources.  Group g of an operand tile = LDS bytes [g*1024, g*1024 + 1024); this wave owns groups
  //      wave + NWAVES*i.  Lane l fills bytes [l*16, l*16+16) of the group: which (row, chunk) that is, and therefore which
  //      global address feeds it, follows from the swizzle. ----
  const T* pa[GA];
  const T* pb[GB];
  int ka[GA], kb[GB];  // k index covered (k-major: first element of the chunk; mn-major: the k row), for the tail predicate
#pragma unroll
  for (int i = 0; i < GA; ++i) {
    const int g = wave + NWAVES * i;
    if (A_KMAJOR) {
      const int r = 8 * g + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
      int64_t row = m0 + r;
      row = row < p.M ? row : p.M - 1;
      ka[i] = kt0 * BK + c * VEC;
      pa[i] = A + row * p.lda + ka[i];
    } else {
      constexpr int RB = BM * ES, RPG = 1024 / RB > 0 ? 1024 / RB : 1, CPR = RB / 16;
      const int kl = g * RPG + (lane * 16) / RB, pc = ((lane * 16) % RB) / 16;
      const int c = ES == 2 ? (pc ^ (4 * (kl & 3))) : pc;
      int64_t m = m0 + (int64_t)c * VEC;
      m = m < p.M ? m : p.M - VEC;
      ka[i] = kt0 * BK + kl;
      pa[i] = A + (int64_t)ka[i] * p.lda + segaddr(m, p.a_seg, p.a_seg_stride);
      (void)CPR;
    }
  }
#pragma unroll
  for (int i = 0; i < GB; ++i) {
    const int g = wave + NWAVES * i;
    if (B_KMAJOR) {
      const int r = 8 * g + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
      int64_t row = n0 + r;
      row = row < p.N ? row : p.N - 1;
      kb[i] = kt0 * BK + c * VEC;
      pb[i] = B + row * p.ldb + kb[i];
    } else {
      constexpr int RB = BN * ES, RPG = 1024 / RB > 0 ? 1024 / RB : 1;
      const int kl = g * RPG + (lane * 16) / RB, pc = ((lane * 16) % RB) / 16;
      const int c = ES == 2 ? (pc ^ (4 * (kl & 3))) : pc;
      int64_t n = n0 + (int64_t)c * VEC;
      n = n < p.N ? n : p.N - VEC;
      kb[i] = kt0 * BK + kl;
      pb[i] = B + (int64_t)kb[i] * p.ldb + segaddr(n, p.b_seg, p.b_seg_stride);
    }
  }
  const int64_t step_a = A_KMAJOR ? (int64_t)BK : (int64_t)BK * p.lda;
  const int64_t step_b = B_KMAJOR ? (int64_t)BK : (int64_t)BK * p.ldb;

  auto issue = [&](int kt, int stage) {
    char* sbase = smem_raw + stage * STAGE_BYTES;
    if (kt < kfull) {
#pragma unroll
      for (int i = 0; i < GA; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa[i],
                                         (__attribute__((address_space(3))) void*)(sbase + (wave + NWAVES * i) * 1024), 16, 0, 0);
        pa[i] += step_a;
        ka[i] += BK;
      }
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb[i],
                                         (__attribute__((address_space(3))) void*)(sbase + A_BYTES + (wave + NWAVES * i) * 1024), 16, 0, 0);
        pb[i] += step_b;
        kb[i] += BK;
      }
    } else {  // K tail: synchronous, zero-filled (runs at most once per workgroup)
      const u32x4 zero = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < GA; ++i) {
        const bool ok = ka[i] < p.K;
        const u32x4 v = *reinterpret_cast<const u32x4*>(ok ? pa[i] : A);
        *reinterpret_cast<u32x4*>(sbase + (wave + NWAVES * i) * 1024 + lane * 16) = ok ? v : zero;
        pa[i] += step_a;
        ka[i] += BK;
      }
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        const bool ok = kb[i] < p.K;
        const u32x4 v = *reinterpret_cast<const u32x4*>(ok ? pb[i] : B);
        *reinterpret_cast<u32x4*>(sbase + A_BYTES + (wave + NWAVES * i) * 1024 + lane * 16) = ok ? v : zero;
        pb[i] += step_b;
        kb[i] += BK;
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int lrow = lane & 31, hi = lane >> 5;
  constexpr int WROWS = TM * 32, WCOLS = TN * 32;
  constexpr int G = GA + GB;  // DMA instructions per thread per tile

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nt) issue(kt0 + s, s);

  // PIN — four waves, 128-row wave tiles (one wave per SIMD, accumulators in AGPRs), two stages: the issue order of a K tile is
  // pinned by sched_barrier(0) behind every MFMA pair — { 2 MFMA ; one fragment read of the next k16 step ; one DMA of the next K tile
  // } — instead of { all DMAs ; per k16 step: all reads, all MFMAs }: a lone wave has no partner to cover its LDS-DMA issues (~60
  // cycles each among MFMAs) or its fragment reads, so they have to sit one per MFMA shadow (tools/probes/gemm4w_pgr.hip: 1 246 ->
  // 1 339 TF/s at 8192^3 for the pinned order; hipcc's own order of the same statements: 1 088).
  constexpr bool PIN = NTHREADS == 256 && TM == 4 && NS == 2 && A_KMAJOR && B_KMAJOR && sizeof(T) == 2;
  if constexpr (PIN) {
    auto issue_one = [&](int d, char* sbase) {
      if (d < GA) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa[d],
                                         (__attribute__((address_space(3))) void*)(sbase + (wave + NWAVES * d) * 1024), 16, 0, 0);
        pa[d] += step_a;
        ka[d] += BK;
      } else {
        const int e = d - GA;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb[e],
                                         (__attribute__((address_space(3))) void*)(sbase + A_BYTES + (wave + NWAVES * e) * 1024), 16, 0, 0);
        pb[e] += step_b;
        kb[e] += BK;
      }
    };
    constexpr int NSLOT = TM * TN / 2, NFR = TM + TN;
    for (int it = 0; it < nt; ++it) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      const bool more = it + 1 < nt;
      const bool dma_ok = more && (kt0 + it + 1) < kfull;
      if (more && !dma_ok) issue(kt0 + it + 1, (it + 1) & 1);  // the K tail tile: synchronous, zero-filled (at most once)
      const char* sbase = smem_raw + (it & 1) * STAGE_BYTES;
      char* nbase = smem_raw + ((it + 1) & 1) * STAGE_BYTES;
      const T* sa = reinterpret_cast<const T*>(sbase);
      const T* sb = reinterpret_cast<const T*>(sbase + A_BYTES);
      Frag<T> fa[2][TM], fb[2][TN];
      auto read_one = [&](int f, int kk, int buf) {  // fragment f of k16 step kk: A rows first, then B rows
        if (f < TM) frag_load_kswz(fa[buf][f], sa, wm * WROWS + f * 32 + lrow, kk * 16 + 8 * hi);
        else frag_load_kswz(fb[buf][f - TM], sb, wn * WCOLS + (f - TM) * 32 + lrow, kk * 16 + 8 * hi);
      };
#pragma unroll
      for (int f = 0; f < NFR; ++f) read_one(f, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int m = 2 * sl + u, i = m / TN, j = m % TN;
            mma16(acc[i][j], fb[kk & 1][j], fa[kk & 1][i]);  // SWAPPED (D = B-frag x A-frag): lane -> output row, register quad -> 4 columns
          }
          if (kk < 3) {
            if (sl < NFR) read_one(sl, kk + 1, (kk + 1) & 1);
            if (sl == 0 && NFR > NSLOT) {  // (TN = 3: seven fragments, six slots)
#pragma unroll
              for (int f = NSLOT; f < NFR; ++f) read_one(f, kk + 1, (kk + 1) & 1);
            }
          }
          if (kk * NSLOT + sl < G && dma_ok) issue_one(kk * NSLOT + sl, nbase);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else
  for (int it = 0; it < nt; ++it) {
    // tile `it` must have landed; up to NS-2 later tiles may stay in flight (steady state), fewer near the end
    const int issued = (it + NS - 1 < nt) ? it + NS - 1 : nt;
    if (NS > 2 && issued - it - 1 == NS - 2) wait_vmcnt<(NS - 2) * G>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // raw barrier: every wave's part of tile `it` is in LDS, and nobody still reads the stage
                                   // that the next issue overwrites
    if (it + NS - 1 < nt) issue(kt0 + it + NS - 1, (it + NS - 1) % NS);
    const char* sbase = smem_raw + (it % NS) * STAGE_BYTES;
    const T* sa = reinterpret_cast<const T*>(sbase);
    const T* sb = reinterpret_cast<const T*>(sbase + A_BYTES);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      Frag<T> fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (A_KMAJOR) {
          const int r = wm * WROWS + i * 32 + lrow;
          frag_load_kswz(fa[i], sa, r, kk + 8 * hi);
        } else {
          frag_load_tr_swz(fa[i], sa, BM, wm * WROWS + i * 32, kk + 8 * hi, lane);
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (B_KMAJOR) {
          const int r = wn * WCOLS + j * 32 + lrow;
          frag_load_kswz(fb[j], sb, r, kk + 8 * hi);
        } else {
          frag_load_tr_swz(fb[j], sb, BN, wn * WCOLS + j * 32, kk + 8 * hi, lane);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma16(acc[i][j], fa[i], fb[j]);
    }
  }

  float* wsp = p.splits > 1 ? p.ws + ((int64_t)z) * p.M * p.N : nullptr;
  const bool ws_vec = (p.N % 4) == 0;
  if constexpr (PIN) {
    // ---- epilogue of the four-wave instantiations: every wave turns its own 128-row block into an fp32 image of 32 rows at a time in
    //      ITS quarter of the LDS (16-byte writes from the swapped accumulator layout), reads it back row-contiguously and runs the
    //      common element epilogue — one pass per 32-row block, no workgroup barrier between them (the generic epilogue below stages 64
    //      rows of the whole tile per pass through one image: four passes, two barriers each, 4-byte LDS writes) ----
    constexpr int LDW = WCOLS + 4, VPRW = WCOLS / 8, RITER = 32 * VPRW / 64;
    static_assert(4 * 32 * LDW * 4 <= NS * STAGE_BYTES && (32 * VPRW) % 64 == 0, "the four waves' images live in the stage buffers");
    __syncthreads();  // every wave is done reading the stages
    float* img = reinterpret_cast<float*>(smem_raw) + wave * (32 * LDW);
#pragma unroll
    for (int i = 0; i < TM; ++i) {  // one 32-row block of the wave tile per pass
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          *reinterpret_cast<f32x4*>(img + lrow * LDW + j * 32 + 8 * q + 4 * hi) = v;
        }
#pragma unroll 2
      for (int tv = 0; tv < RITER; ++tv) {
        const int v = tv * 64 + lane;
        const int rl = v / VPRW, cl = (v % VPRW) * 8;
        const int64_t row = m0 + wm * WROWS + i * 32 + rl, col = n0 + wn * WCOLS + cl;
        if (row >= p.M || col >= p.N) continue;
        float x[8];
        {
          const f32x4 a = *reinterpret_cast<const f32x4*>(img + rl * LDW + cl);
          const f32x4 b = *reinterpret_cast<const f32x4*>(img + rl * LDW + cl + 4);
          x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
        }
        const bool full = col + 8 <= p.N;
        if (wsp) {
          if (full && ws_vec) store8(wsp + row * p.N + col, x);
          else
            for (int e = 0; e < 8 && col + e < p.N; ++e) wsp[row * p.N + col + e] = x[e];
        } else if (full && p.vec_epi) {
          epilogue_store8<T>(p, cofs, bofs, row, col, x);
        } else {
          for (int e = 0; e < 8 && col + e < p.N; ++e) epilogue_store<T>(p, cofs, bofs, row, col + e, x[e]);
        }
      }
    }
    return;
  }
  // ---- epilogue (identical to gemm_kernel) ----
  constexpr int LDC = BN + 4;
  constexpr int VPR = BN / 8;
  constexpr int NPASS = BM / 64;
  constexpr int ITERS = 64 * VPR / NTHREADS;
  float* stage = reinterpret_cast<float*>(smem_raw);
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int trow = wm * WROWS + i * 32;
      if (trow / 64 == ps) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            stage[(trow - 64 * ps + acc_row(r, lane)) * LDC + wn * WCOLS + j * 32 + lrow] = acc[i][j][r];
      }
    }
    __syncthreads();
#pragma unroll 2
    for (int itv = 0; itv < ITERS; ++itv) {
      const int v = tid + NTHREADS * itv;
      const int rl = v / VPR, cl = (v % VPR) * 8;
      const int64_t row = m0 + 64 * ps + rl, col = n0 + cl;
      if (row >= p.M || col >= p.N) continue;
      float x[8];
      {
        const f32x4 a = *reinterpret_cast<const f32x4*>(stage + rl * LDC + cl);
        const f32x4 b = *reinterpret_cast<const f32x4*>(stage + rl * LDC + cl + 4);
        x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
      }
      const bool full = col + 8 <= p.N;
      if (wsp) {
        if (full && ws_vec) store8(wsp + row * p.N + col, x);
        else
          for (int e = 0; e < 8 && col + e < p.N; ++e) wsp[row * p.N + col + e] = x[e];
      } else if (full && p.vec_epi) {
        epilogue_store8<T>(p, cofs, bofs, row, col, x);
      } else {
        for (int e = 0; e < 8 && col + e < p.N; ++e) epilogue_store<T>(p, cofs, bofs, row, col + e, x[e]);
      }
    }
  }
}

template <typename T>
__global__ void splitk_reduce_kernel(GemmParams p) {
  const int64_t total = p.M * p.N;
  const int64_t bidx = blockIdx.y;
  const int64_t b0 = bidx / p.batch1, b1 = bidx % p.batch1;
  const int64_t cofs = b0 * p.sc0 + b1 * p.sc1;
  const int64_t bofs = b0 * p.sbias0 + b1 * p.sbias1;
  const float* ws = p.ws + bidx * p.splits * total;
  if (p.colsum) {  // the slices' column sums of A (cst_gemm_desc.colsum), added in split order
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < p.M; m += (int64_t)gridDim.x * blockDim.x) {
      float v = 0.0f;
      for (int s = 0; s < p.splits; ++s) v += p.colsum_ws[(bidx * p.splits + s) * p.M + m];
      DT<T>::st((T*)p.colsum + bidx * p.M + m, v);
    }
  }
  if (p.vec_epi && (p.N % 8) == 0) {
    // 8 consecutive columns per thread: 16-byte loads, four splits' loads in flight, ONE row/column division per 8 elements, the vector
    // epilogue.  The slabs are added in split order, as in the scalar loop below (same bits).
    const int64_t total8 = total >> 3;
    for (int64_t i8 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i8 < total8; i8 += (int64_t)gridDim.x * blockDim.x) {
      const float* src = ws + i8 * 8;
      float v[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      int sp = 0;
      for (; sp + 4 <= p.splits; sp += 4) {
        f32x4 a[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a[u][0] = *reinterpret_cast<const f32x4*>(src + (int64_t)(sp + u) * total);
          a[u][1] = *reinterpret_cast<const f32x4*>(src + (int64_t)(sp + u) * total + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] += a[u][0][e]; v[4 + e] += a[u][1][e]; }
      }
      for (; sp < p.splits; ++sp) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + (int64_t)sp * total), a1 = *reinterpret_cast<const f32x4*>(src + (int64_t)sp * total + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += a0[e]; v[4 + e] += a1[e]; }
      }
      const int64_t i = i8 * 8, row = i / p.N;
      epilogue_store8<T>(p, cofs, bofs, row, i - row * p.N, v);
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.0f;
    for (int s = 0; s < p.splits; ++s) v += ws[s * total + i];
    epilogue_store<T>(p, cofs, bofs, i / p.N, i % p.N, v);
  }
}

template <typename T, bool AK, bool BK_, bool SEG, typename C>
int launch(GemmParams p, int64_t M, int64_t N, int64_t nbatch, hipStream_t s) {
  constexpr int VEC = DT<T>::VEC;
  constexpr int BKc = C::KV * VEC, LDK = BKc + VEC;
  constexpr int LDMA = sizeof(T) == 2 ? C::BM + 32 : C::BM + VEC, LDMB = sizeof(T) == 2 ? C::BN + 32 : C::BN + VEC;
  constexpr int A_ELEMS = AK ? C::BM * LDK : BKc * LDMA;
  constexpr int B_ELEMS = BK_ ? C::BN * LDK : BKc * LDMB;
  size_t lds = 2 * (size_t)(A_ELEMS + B_ELEMS) * sizeof(T);
  const size_t stage_bytes = (size_t)64 * (C::BN + 4) * sizeof(float);
  if (lds < stage_bytes) lds = stage_bytes;
  static bool attr_set = false;  // LDS > 64 KiB needs the opt-in attribute (160 KiB/CU on gfx950)
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, AK, BK_, SEG, C>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);  // the kernel also has 2 KiB of static LDS (live K-tile list)
    attr_set = true;
  }
  p.tiles_m = (int)cst_ceil_div(M, C::BM);
  p.tiles_n = (int)cst_ceil_div(N, C::BN);
  static const bool no_split_order = getenv("CST_GEMM_NO_SPLIT_ORDER") != nullptr;
  p.split_order = (p.splits > 1 && nbatch == 1 && !no_split_order) ? 1 : 0;
  dim3 grid(p.tiles_m * p.tiles_n, 1, (unsigned)(nbatch * p.splits));
  if constexpr (!AK && !BK_ && !SEG && C::NT <= 256) {
    if (p.colsum) {
      static bool attr_cs = false;
      if (!attr_cs) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, AK, BK_, SEG, C, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        attr_cs = true;
      }
      hipLaunchKernelGGL((gemm_kernel<T, AK, BK_, SEG, C, true>), grid, dim3(C::NT), lds, s, p);
      return cst_check_launch("cst_gemm");
    }
  }
  if (p.colsum) { cst_set_error("cst_gemm: internal: column sums requested from a configuration that does not build them"); return CST_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL((gemm_kernel<T, AK, BK_, SEG, C>), grid, dim3(C::NT), lds, s, p);
  return cst_check_launch("cst_gemm");
}

template <typename T, bool AK, bool BK_, typename C, int NS>
int launch_glds(GemmParams p, int64_t M, int64_t N, int64_t nbatch, hipStream_t s) {
  constexpr int ES = sizeof(T), BKc = 128 / ES;
  constexpr int A_BYTES = AK ? C::BM * 128 : BKc * C::BM * ES, B_BYTES = BK_ ? C::BN * 128 : BKc * C::BN * ES;
  size_t lds = (size_t)NS * (A_BYTES + B_BYTES);
  const size_t stage_bytes = (size_t)64 * (C::BN + 4) * sizeof(float);
  if (lds < stage_bytes) lds = stage_bytes;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<T, AK, BK_, C, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  p.tiles_m = (int)cst_ceil_div(M, C::BM);
  p.tiles_n = (int)cst_ceil_div(N, C::BN);
  dim3 grid(p.tiles_m * p.tiles_n, 1, (unsigned)(nbatch * p.splits));
  hipLaunchKernelGGL((gemm_glds_kernel<T, AK, BK_, C, NS>), grid, dim3(C::NT), lds, s, p);
  return cst_check_launch("cst_gemm");
}

// Large tiles when they still give every CU a workgroup (256 CUs, 1 large block each); else the small configuration.
bool use_large(const cst_gemm_desc* d, int splits) {
  if ((d->a_kmajor && d->a_seg) || (d->b_kmajor && d->b_seg)) return false;
  const int64_t tiles = cst_ceil_div(d->M, 256) * cst_ceil_div(d->N, 256) * d->batch0 * d->batch1 * splits;
  return d->M >= 256 && d->N >= 256 && tiles >= 200;
}

bool getenv_no_glds() {
  static const bool v = getenv("CST_GEMM_NO_GLDS") != nullptr;
  return v;
}

int choose_splits(const cst_gemm_desc* d) {
  if (d->split_k > 1) return d->split_k;
  if (d->split_k == 0 || d->split_k == 1) return 1;
  const int bk = d->dtype == CST_BF16 ? 64 : 32;
  const int64_t ktiles = cst_ceil_div(d->K, bk);
  const int64_t nb = d->batch0 * d->batch1;
  // prefer the large configuration: splits so that ~256-512 large workgroups exist
  if (d->M >= 256 && d->N >= 256) {
    const int64_t tl = cst_ceil_div(d->M, 256) * cst_ceil_div(d->N, 256) * nb;
    if (tl >= 200) return 1;
    // one workgroup per CU: the largest split count that still fits ONE round of 256 workgroups (288 would run as a full
    // round plus a 12 %-occupied second one) — and leaves every workgroup at least 16 K tiles: below that the 256 x 256 tile's
    // prologue and its 256 KB of partial sums per workgroup outweigh it (512 x 2048 x 7901: 15 slices of 8 K tiles 45.6 us,
    // 128 x 128 tiles in 4 slices 39.2)
    int64_t s = 256 / tl;
    if (s > ktiles / 16) s = ktiles / 16;
    if (s > 32) s = 32;
    if (s >= 1 && tl * s >= 200) return (int)s;
  }
  const int64_t tiles = cst_ceil_div(d->M, 128) * cst_ceil_div(d->N, 128) * nb;
  if (tiles >= 256 || ktiles < 16) return 1;
  // 128 x 128 tiles, two workgroups fit a CU: a split count is priced as (K share of one workgroup) x (workgroups on the busiest
  // CU) — one round of <= 256 workgroups at 1/s1, or <= 512 (two per CU, each at half rate) at 2/s2.  The second only when it is
  // clearly cheaper: it doubles the partial sums the reduce kernel reads.  (Measured, tools/bench_gemm_splitk.py: 1536 x 512 x 7901
  // s=4 37.8 us against 56.4 us for the former 512-workgroup target (s=11); 512 x 512 x 12000 s=12 26.1 against 33.7 (s=23).)
  const int64_t cap = ktiles / 8 < 32 ? ktiles / 8 : 32;
  int64_t s1 = 256 / tiles, s2 = 512 / tiles;
  if (s1 > cap) s1 = cap;
  if (s2 > cap) s2 = cap;
  if (s1 < 1) s1 = 1;
  if (s2 < 1) s2 = 1;
  const int64_t s = (2.0 / (double)s2 < 0.9 / (double)s1) ? s2 : s1;
  return (int)s;
}

}  // namespace

// cst_gemm_desc.colsum is a by-product of the 4-wave register-staged configurations (both operands mn-major: the dW layout); every
// other launch gets it from the two-launch column sum over A, run by cst_gemm itself behind the GEMM (unbatched problems only).
static bool colsum_fused(const cst_gemm_desc* d, int splits) {
  return d->colsum && !d->a_kmajor && !d->b_kmajor && !d->a_seg && !d->b_seg && !use_large(d, splits);
}

extern "C" int64_t cst_colsum_workspace(int64_t rows, int64_t cols);
extern "C" int cst_colsum_typed(const void* x, int64_t ldx, void* out, void* workspace, int64_t rows, int64_t cols, int dtype, int out_dtype, cst_stream stream);
extern "C" int cst_colsum_typed_live(const void* x, int64_t ldx, void* out, void* workspace, int64_t rows, int64_t cols, int dtype, int out_dtype,
                                     const uint32_t* row_live, uint32_t epoch, cst_stream stream);

extern "C" int cst_gemm_splits(const cst_gemm_desc* d) { return d ? choose_splits(d) : 0; }

extern "C" int cst_gemm_colsum_is_fused(const cst_gemm_desc* d) { return d && colsum_fused(d, choose_splits(d)) ? 1 : 0; }

extern "C" int64_t cst_gemm_workspace(const cst_gemm_desc* d) {
  const int s = choose_splits(d);
  const int64_t nb = d->batch0 * d->batch1;
  int64_t bytes = s > 1 ? (int64_t)s * d->M * d->N * nb * (int64_t)sizeof(float) : 0;
  if (d->colsum) bytes += colsum_fused(d, s) ? (s > 1 ? (int64_t)s * d->M * nb * (int64_t)sizeof(float) : 0) : cst_colsum_workspace(d->K, d->M);
  return bytes;
}

static int gemm_one(const cst_gemm_desc* d, cst_stream stream, int64_t drop_row0) {
  CST_REQUIRE(d && d->A && d->B && d->C, "cst_gemm: null operand");
  CST_REQUIRE(d->dtype == CST_F32 || d->dtype == CST_BF16, "cst_gemm: bad dtype %d", d->dtype);
  CST_REQUIRE(d->c_dtype == d->dtype || d->c_dtype == CST_F32, "cst_gemm: c_dtype must be dtype or f32");
  CST_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "cst_gemm: empty problem M=%lld N=%lld K=%lld", (long long)d->M,
              (long long)d->N, (long long)d->K);
  const int vec = d->dtype == CST_BF16 ? 8 : 4;
  if (d->a_kmajor) CST_REQUIRE(d->K % vec == 0 && d->lda % vec == 0, "cst_gemm: k-major A needs K,lda %% %d == 0 (K=%lld lda=%lld)", vec, (long long)d->K, (long long)d->lda);
  else CST_REQUIRE(d->M % vec == 0 && d->lda % vec == 0, "cst_gemm: mn-major A needs M,lda %% %d == 0 (M=%lld lda=%lld)", vec, (long long)d->M, (long long)d->lda);
  if (d->b_kmajor) CST_REQUIRE(d->K % vec == 0 && d->ldb % vec == 0, "cst_gemm: k-major B needs K,ldb %% %d == 0 (K=%lld ldb=%lld)", vec, (long long)d->K, (long long)d->ldb);
  else CST_REQUIRE(d->N % vec == 0 && d->ldb % vec == 0, "cst_gemm: mn-major B needs N,ldb %% %d == 0 (N=%lld ldb=%lld)", vec, (long long)d->N, (long long)d->ldb);
  CST_REQUIRE(d->a_seg % vec == 0 && d->b_seg % vec == 0 && d->a_seg_stride % vec == 0 && d->b_seg_stride % vec == 0,
              "cst_gemm: segment sizes must be multiples of %d", vec);
  CST_REQUIRE(((uintptr_t)d->A % 16 == 0) && ((uintptr_t)d->B % 16 == 0), "cst_gemm: A/B must be 16-byte aligned");
  CST_REQUIRE(d->sa0 % vec == 0 && d->sa1 % vec == 0 && d->sb0 % vec == 0 && d->sb1 % vec == 0, "cst_gemm: batch strides of A/B must be multiples of %d", vec);
  CST_REQUIRE(d->batch0 >= 1 && d->batch1 >= 1, "cst_gemm: batch counts must be >= 1");
  CST_REQUIRE(!d->dact || d->aux_in, "cst_gemm: dact needs aux_in");

  GemmParams p;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.A = d->A; p.lda = d->lda; p.a_seg = d->a_seg; p.a_seg_stride = d->a_seg_stride;
  p.B = d->B; p.ldb = d->ldb; p.b_seg = d->b_seg; p.b_seg_stride = d->b_seg_stride;
  p.C = d->C; p.ldc = d->ldc;
  p.bias = d->bias; p.bias_mode = d->bias ? d->bias_mode : CST_BIAS_NONE; p.sbias0 = d->sbias0; p.sbias1 = d->sbias1;
  p.act = d->act;
  p.aux_out = d->aux_out; p.ld_aux_out = d->ld_aux_out;
  p.dact = d->dact; p.aux_in = d->aux_in; p.ld_aux_in = d->ld_aux_in;
  p.resid = d->resid; p.ld_resid = d->ld_resid;
  p.alpha = d->alpha;
  CST_REQUIRE(d->drop_p >= 0.0f && d->drop_p < 1.0f, "cst_gemm: drop_p must be in [0, 1)");
  CST_REQUIRE(d->drop_p == 0.0f || d->batch0 * d->batch1 == 1, "cst_gemm: fused dropout needs an unbatched problem");
  p.drop_thr = d->drop_p > 0.0f ? cst_drop_thr16(d->drop_p) : 0u;
  p.drop_key = d->drop_key;
  p.drop_scale = d->drop_p > 0.0f ? 1.0f / (1.0f - d->drop_p) : 1.0f;
  p.drop_row0 = drop_row0;
  p.batch1 = d->batch1;
  p.sa0 = d->sa0; p.sa1 = d->sa1; p.sb0 = d->sb0; p.sb1 = d->sb1; p.sc0 = d->sc0; p.sc1 = d->sc1;
  p.c_f32 = (d->c_dtype == CST_F32 && d->dtype != CST_F32) ? 1 : 0;
  if (d->dtype == CST_F32) p.c_f32 = 0;  // T == float already stores fp32
  {
    auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q % 16) == 0; };
    bool ok = d->ldc % 8 == 0 && d->sc0 % 8 == 0 && d->sc1 % 8 == 0 && al16(d->C) && al16(d->bias) && al16(d->aux_out) && al16(d->aux_in) && al16(d->resid);
    if (d->bias && d->bias_mode == CST_BIAS_COL) ok = ok && d->sbias0 % 8 == 0 && d->sbias1 % 8 == 0;
    if (d->aux_out) ok = ok && d->ld_aux_out % 8 == 0;
    if (d->aux_in) ok = ok && d->ld_aux_in % 8 == 0;
    if (d->resid) ok = ok && d->ld_resid % 8 == 0;
    if (d->drop_p > 0.0f) ok = ok && (d->N % 2 == 0);  // pair-aligned mask words
    p.vec_epi = ok ? 1 : 0;
  }
  p.tiles_m = p.tiles_n = 0;  // set per configuration in launch()
  p.split_order = 0;
  p.sched = nullptr;  // set by the persistent kernel's launcher
  p.gperm_full = 0;  // (likewise)
  p.mrot = 0;
  {
    static const bool no_klive = getenv("CST_GEMM_NO_KLIVE") != nullptr;  // A/B switch: visit the all-zero K blocks too
    p.k_live = no_klive ? nullptr : d->k_live;
    p.k_epoch = d->k_epoch;
    p.m_live = no_klive ? nullptr : d->m_live;
    p.m_epoch = d->m_epoch;
    p.k_len = no_klive ? nullptr : d->k_len;
    p.m_len = no_klive ? nullptr : d->m_len;
  }
  {
    // m tiles per group of the XCD-local tile walk: 4 (a 4 x 8 patch of concurrent tiles per XCD) measured 2-8 % faster than 8 on the
    // N = 768 / 2304 / 3072 Linear shapes at M = 47 968 (same-box sweep over 2 / 4 / 8 / 16 / 32, tools/bench_gemm4w.py)
    static const int gm = getenv("CST_GEMM8P_GROUP_M") ? atoi(getenv("CST_GEMM8P_GROUP_M")) : 4;
    p.group_m = gm > 0 ? gm : 4;
  }
  p.splits = choose_splits(d);
  p.ws = nullptr;
  CST_REQUIRE(!d->colsum || (!d->a_kmajor && !d->a_seg), "cst_gemm: colsum needs an mn-major, unsegmented A");
  const bool cs_fused = colsum_fused(d, p.splits);
  CST_REQUIRE(!d->colsum || cs_fused || (d->batch0 * d->batch1 == 1 && d->M % 8 == 0 && d->lda % 8 == 0),
              "cst_gemm: colsum of this configuration runs as a separate column sum: unbatched, M and lda multiples of 8");
  p.colsum = cs_fused ? d->colsum : nullptr;
  p.colsum_ws = nullptr;
#ifdef CST_TRACE
  if (p.splits == 1 && d->workspace) p.ws = (float*)d->workspace;  // cycle-stamp buffer of tools/gemm8p_trace.py
#endif
  const int64_t nbatch = d->batch0 * d->batch1;
  CST_REQUIRE(cst_ceil_div(d->M, 128) * cst_ceil_div(d->N, 128) < (1ll << 31) && nbatch * p.splits < 65536, "cst_gemm: grid too large");
  if (p.splits > 1) {
    const int64_t need = cst_gemm_workspace(d);
    if (!d->workspace || d->workspace_bytes < need) {
      cst_set_error("cst_gemm: split_k=%d needs %lld workspace bytes, got %lld", p.splits, (long long)need,
                    (long long)d->workspace_bytes);
      return CST_ERR_WORKSPACE;
    }
    p.ws = (float*)d->workspace;
    p.colsum_ws = p.ws + (int64_t)p.splits * d->M * d->N * nbatch;
  } else if (d->colsum && !cs_fused && (!d->workspace || d->workspace_bytes < cst_gemm_workspace(d))) {
    cst_set_error("cst_gemm: colsum needs %lld workspace bytes, got %lld", (long long)cst_gemm_workspace(d), (long long)d->workspace_bytes);
    return CST_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  // Work that is skipped is never credited: with live-tile stamps only the live K blocks count.  The stamps live on the device,
  // so this costs a synchronous read-back — done only while the profiling table is recording (bench.py's untimed roofline step).
  double kfrac = 1.0;
  if (p.k_live && cst_prof_is_on()) {
    const int64_t nt = cst_ceil_div(d->K, 64);
    std::vector<uint32_t> st((size_t)nt);
    if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(st.data(), p.k_live, sizeof(uint32_t) * nt, hipMemcpyDeviceToHost) == hipSuccess) {
      int64_t live = 0;
      for (int64_t i = 0; i < nt; ++i) live += st[i] == p.k_epoch;
      kfrac = (double)live / (double)nt;
    }
  }
  if (p.k_len && cst_prof_is_on()) {
    std::vector<int32_t> kl((size_t)d->batch0);
    if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(kl.data(), p.k_len, sizeof(int32_t) * d->batch0, hipMemcpyDeviceToHost) == hipSuccess) {
      double sum = 0.0;
      for (int64_t b = 0; b < d->batch0; ++b) sum += (double)(kl[b] < 0 ? 0 : (kl[b] < d->K ? kl[b] : d->K));
      kfrac *= sum / ((double)d->batch0 * (double)d->K);
    }
  }
  if (p.m_len && cst_prof_is_on()) {  // conv GEMMs: credit the output tiles that still run their K loop
    std::vector<int32_t> ml((size_t)d->batch0);
    if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(ml.data(), p.m_len, sizeof(int32_t) * d->batch0, hipMemcpyDeviceToHost) == hipSuccess) {
      const int64_t tiles = cst_ceil_div(d->M, 256);
      double live = 0.0;
      for (int64_t b = 0; b < d->batch0; ++b) {
        const int64_t t = ml[b] <= 0 ? 0 : cst_ceil_div((int64_t)ml[b], 256);
        live += (double)(t < tiles ? t : tiles);
      }
      kfrac *= live / ((double)d->batch0 * (double)tiles);
    }
  }
  if (p.m_live && cst_prof_is_on()) {  // dX GEMMs: credit the 256-row output tiles that still run their K loop (8p / 16-wave tile)
    const int64_t nt = cst_ceil_div(d->M, 64);
    std::vector<uint32_t> st((size_t)nt);
    if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(st.data(), p.m_live, sizeof(uint32_t) * nt, hipMemcpyDeviceToHost) == hipSuccess) {
      int64_t live = 0, tiles = cst_ceil_div(d->M, 256);
      for (int64_t t = 0; t < tiles; ++t) {
        bool l = false;
        for (int64_t i = 4 * t; i < 4 * t + 4 && i < nt; ++i) l |= st[i] == p.m_epoch;
        live += l;
      }
      kfrac *= (double)live / (double)tiles;
    }
  }
  const double flops = 2.0 * (double)d->M * (double)d->N * (double)d->K * (double)nbatch * kfrac;
  const double esz = (double)cst_dtype_size(d->dtype);
  const double bytes = ((double)d->M * d->K + (double)d->N * d->K) * esz * nbatch * kfrac + (double)d->M * d->N * cst_dtype_size(d->c_dtype) * nbatch;
  CstProfScope prof(CST_K_GEMM, s, flops, bytes);
  int rc;
  const bool ak = d->a_kmajor != 0, bk = d->b_kmajor != 0;
  // SEG instantiation only where a k-major operand is segmented (grouped conv with channel-major input); mn-major
  // segments are folded into the per-thread base pointer and cost nothing per tile.
  const bool seg = (ak && d->a_seg) || (bk && d->b_seg);
  const bool large = use_large(d, p.splits);
#define CST_GEMM_LAYOUT(T, SEGV, CFG)                                                                                   \
  (ak ? (bk ? launch<T, true, true, SEGV, CFG>(p, d->M, d->N, nbatch, s) : launch<T, true, false, SEGV, CFG>(p, d->M, d->N, nbatch, s)) \
      : (bk ? launch<T, false, true, SEGV, CFG>(p, d->M, d->N, nbatch, s) : launch<T, false, false, SEGV, CFG>(p, d->M, d->N, nbatch, s)))
#define CST_GLDS_LAYOUT(T, CFG, NS)                                                                                     \
  (ak ? (bk ? launch_glds<T, true, true, CFG, NS>(p, d->M, d->N, nbatch, s) : launch_glds<T, true, false, CFG, NS>(p, d->M, d->N, nbatch, s)) \
      : (bk ? launch_glds<T, false, true, CFG, NS>(p, d->M, d->N, nbatch, s) : launch_glds<T, false, false, CFG, NS>(p, d->M, d->N, nbatch, s)))
  // measured on MI355X (tools/bench_kernels.py): the DMA path wins when both operands are k-major (716 vs 661 TF/s, fc1
  // forward); with an mn-major operand the register-staged path + padded LDS rows is faster (690 vs 649, 562 vs 486).
  static const bool glds_all = getenv("CST_GEMM_GLDS_ALL") != nullptr;
  const bool no_glds = getenv_no_glds() || (!(ak && bk) && !glds_all) || d->colsum != nullptr;  // (column sums live in gemm_kernel)
#define CST_GEMM_DISPATCH(T)                                                                         \
  (seg ? CST_GEMM_LAYOUT(T, true, CfgSmall)                                                           \
       : (no_glds ? (large ? CST_GEMM_LAYOUT(T, false, CfgLarge) : CST_GEMM_LAYOUT(T, false, CfgSmall)) \
                  : (large ? CST_GLDS_LAYOUT(T, CfgLarge, 2) : CST_GLDS_LAYOUT(T, CfgSmall, 2))))
  // bf16 problems that fill the chip with 256 x 256 tiles take the 8-phase DMA-pipelined kernel (gemm8p.hip)
  static const bool no_8p = getenv("CST_GEMM_NO_8P") != nullptr, force_8p = getenv("CST_GEMM_FORCE_8P") != nullptr;
  // (A k-major only: with an mn-major A the register-staged 16-wave kernel measured 20-30 % faster on the dW shapes; and not
  //  with an act'(aux_in) epilogue, whose 128 KiB-per-tile operand read is exposed at one workgroup per CU: 0.49 vs 0.42 ms)
  static const bool all_8p = getenv("CST_GEMM_8P_ALL") != nullptr;
  // act'(aux_in) epilogues: taken since the operand vectors are requested in one burst ahead of the epilogue passes (gemm8p.hip)
  static const bool dact_8p = getenv("CST_GEMM_8P_NO_DACT") == nullptr;
  static const bool no_skinny = getenv("CST_GEMM_NO_SKINNY") != nullptr;
  // (256: -0.3 ms per update, 512 no better than 0.  400 looked better in back-to-back launches with the two-stage ring (4064 x 1536 x 512
  //  17.4 -> 15.9 us) and was worse inside the update, where the operands are not L2-warm: 21.5 -> 24.4 us.)
  static const int64_t skinny_tiles = getenv("CST_GEMM_SKINNY_TILES") ? atoll(getenv("CST_GEMM_SKINNY_TILES")) : 256;
  static const bool skinny_ns4 = getenv("CST_GEMM_SKINNY_NS4") != nullptr;   // A/B switches
  static const bool no_mid8p = getenv("CST_GEMM_NO_MID8P") != nullptr;
  // k/k problems of 150-199 tiles of 256 x 256 (7901 x 1536 x 512: 186): the persistent kernel on 3/4 of the CUs still beats 744
  // workgroups of 128 x 128 (19.5 vs 28.4 us)
  const bool mid8p = !no_mid8p && ak && bk && !seg && nbatch == 1 && p.splits == 1 && d->M >= 256 && d->N >= 256 &&
                     cst_ceil_div(d->M, 256) * cst_ceil_div(d->N, 256) >= 150;
  static const bool no_narrow = getenv("CST_GEMM_NO_NARROW") != nullptr;
  // gemm4w.hip (four waves, pinned issue order, three A stages): the N = 768 long-K launches of the step, same bits as the 8-wave
  // kernel, 3-8 % faster inside the update (CST_GEMM_NO_4W=1: the 8-wave kernel everywhere, for A/B runs)
  static const bool no_4w = getenv("CST_GEMM_NO_4W") != nullptr;
  static const bool experiment = getenv("CST_GEMM_EXPERIMENT") != nullptr;
  const char* force_cfg = experiment ? getenv("CST_GEMM_FORCE_CFG") : nullptr;
  if (force_cfg && !*force_cfg) force_cfg = nullptr;
  if (!no_narrow && !seg && !large && ak && bk && d->N <= 64 && d->M > 256) {
    rc = d->dtype == CST_BF16 ? launch_glds<bf16_t, true, true, CfgNarrowN, 2>(p, d->M, d->N, nbatch, s)
                              : launch_glds<float, true, true, CfgNarrowN, 2>(p, d->M, d->N, nbatch, s);
  } else if (!no_narrow && !seg && !large && !ak && !bk && d->M <= 64 && d->N > 256) {
    rc = d->dtype == CST_BF16 ? launch<bf16_t, false, false, false, CfgNarrowM>(p, d->M, d->N, nbatch, s)
                              : launch<float, false, false, false, CfgNarrowM>(p, d->M, d->N, nbatch, s);
  } else if (force_cfg && !ak && !bk && !seg && nbatch == 1 && d->dtype == CST_BF16) {
    // (dW layouts: both operands mn-major; the split count comes from the caller's split_k.  tools/bench_gemm_cfg_dw.py — its
    //  back-to-back launches keep the operands in L2 / Infinity Cache: 64 x 64 tiles with a quarter of the K slices won there
    //  (512 x 512 x 4064: 19.9 -> 16.2 us) and LOST inside the update, where the operands come from HBM and the longer K loop per
    //  workgroup is exposed (22.4 -> 24.6 us; x 7901: 24.8 -> 35.2): not adopted.)
    const std::string f(force_cfg);
    if (f == "small") rc = launch<bf16_t, false, false, false, CfgSmall>(p, d->M, d->N, nbatch, s);
    else if (f == "narrowm") rc = launch<bf16_t, false, false, false, CfgNarrowM>(p, d->M, d->N, nbatch, s);
    else if (f == "narrown") rc = launch<bf16_t, false, false, false, CfgNarrowN>(p, d->M, d->N, nbatch, s);
    else if (f == "skinny") rc = launch<bf16_t, false, false, false, CfgSkinny>(p, d->M, d->N, nbatch, s);
    else if (f == "large") rc = launch<bf16_t, false, false, false, CfgLarge>(p, d->M, d->N, nbatch, s);
    else if (f == "big4r") rc = launch<bf16_t, false, false, false, Cfg<256, 256, 2, 2>>(p, d->M, d->N, nbatch, s);  // 4 waves x (128 x 128), register-staged
    else { cst_set_error("cst_gemm: unknown CST_GEMM_FORCE_CFG %s", force_cfg); return CST_ERR_BAD_ARG; }
  } else if (force_cfg && ak && bk && !seg && nbatch == 1 && p.splits == 1 && d->dtype == CST_BF16) {
    // tools/bench_gemm_cfg.py: one shape through every k/k configuration (CST_GEMM_EXPERIMENT=1, CST_GEMM_FORCE_CFG read per call)
    const std::string f(force_cfg);
    if (f == "skinny4") rc = launch_glds<bf16_t, true, true, CfgSkinny, 4>(p, d->M, d->N, nbatch, s);
    else if (f == "skinny2") rc = launch_glds<bf16_t, true, true, CfgSkinny, 2>(p, d->M, d->N, nbatch, s);
    else if (f == "small2") rc = launch_glds<bf16_t, true, true, CfgSmall, 2>(p, d->M, d->N, nbatch, s);
    else if (f == "small3") rc = launch_glds<bf16_t, true, true, CfgSmall, 3>(p, d->M, d->N, nbatch, s);
    else if (f == "big4") rc = launch_glds<bf16_t, true, true, Cfg<256, 256, 2, 2>, 2>(p, d->M, d->N, nbatch, s);  // 4 waves x (128 x 128): see tools/bench_gemm_big4.py
    else if (f == "big4n") rc = launch_glds<bf16_t, true, true, Cfg<256, 192, 2, 2>, 2>(p, d->M, d->N, nbatch, s);  // 128 x 96 wave tiles: N = 768 is 4 tiles
    else if (f == "big4r") rc = launch<bf16_t, true, true, false, Cfg<256, 256, 2, 2>>(p, d->M, d->N, nbatch, s);          // the same, register-staged
    else if (f == "large") rc = launch<bf16_t, true, true, false, CfgLarge>(p, d->M, d->N, nbatch, s);
    else if (f == "8p" && cst_gemm8p_supported(p, ak, bk, nbatch)) rc = cst_gemm8p_launch(p, ak, bk, nbatch, s);
    else if (f == "4w" && cst_gemm4w_supported(p, ak, bk, nbatch)) rc = cst_gemm4w_launch(p, s);
    else { cst_set_error("cst_gemm: unknown CST_GEMM_FORCE_CFG %s", force_cfg); return CST_ERR_BAD_ARG; }
  } else if (!no_skinny && ak && bk && !seg && nbatch == 1 && p.splits == 1 &&
             (d->M <= 256 || (!large && !mid8p && cst_ceil_div(d->M, 128) * cst_ceil_div(d->N, 128) < skinny_tiles))) {
    // (mid-size problems too — the Linear layers of the 512-wide encoder / decoder: few 128 x 128 tiles per CU leave CUs idle; 64 x 64
    //  tiles quadruple the workgroups.)  Ring depth: four stages keep 48 KB per workgroup in flight, which is what a decode-step
    //  projection (<= 256 rows, a few dozen workgroups, long K) is bound by; with more workgroups than fit two per CU the LDS they
    //  hold is the limit instead — two stages let five of them share a CU (tools/bench_gemm_cfg.py: 7901 x 512 x 512 14.9 -> 12.1 us,
    //  7901 x 512 x 2048 33.7 -> 28.7, 12000 x 512 x 512 17.1 -> 15.3; same bits: the K order per output element does not change).
    const bool two_stage = d->M > 256 && cst_ceil_div(d->M, 64) * cst_ceil_div(d->N, 64) > 512 && !skinny_ns4;
    if (two_stage) rc = d->dtype == CST_BF16 ? launch_glds<bf16_t, true, true, CfgSkinny, 2>(p, d->M, d->N, nbatch, s)
                                             : launch_glds<float, true, true, CfgSkinny, 2>(p, d->M, d->N, nbatch, s);
    else rc = d->dtype == CST_BF16 ? launch_glds<bf16_t, true, true, CfgSkinny, 4>(p, d->M, d->N, nbatch, s)
                                   : launch_glds<float, true, true, CfgSkinny, 4>(p, d->M, d->N, nbatch, s);
  } else if (d->dtype == CST_BF16 && !seg && !no_4w && !d->colsum && cst_gemm4w_supported(p, ak, bk, nbatch)) {
    rc = cst_gemm4w_launch(p, s);  // long K, N a multiple of 192 (the N = 768 family): four waves, pinned issue order (gemm4w.hip)
  } else if (d->dtype == CST_BF16 && !seg && !no_8p && !d->colsum && (ak || all_8p || force_8p) && (!d->dact || (dact_8p && !d->resid && !d->aux_out) || all_8p || force_8p) && (large || mid8p || force_8p) && cst_gemm8p_supported(p, ak, bk, nbatch))
    rc = cst_gemm8p_launch(p, ak, bk, nbatch, s);
  else if (d->dtype == CST_BF16) rc = CST_GEMM_DISPATCH(bf16_t);
  else rc = CST_GEMM_DISPATCH(float);
#undef CST_GEMM_DISPATCH
#undef CST_GEMM_LAYOUT
#undef CST_GLDS_LAYOUT
  if (cst_prof_is_on())
    prof.tag("M=%lld N=%lld K=%lld b=%lld a%c b%c split=%d%s%s%s%s%s%s%s", (long long)d->M, (long long)d->N, (long long)d->K, (long long)nbatch, ak ? 'k' : 'm',
             bk ? 'k' : 'm', p.splits, d->bias ? " bias" : "", d->act ? " act" : "", d->aux_out ? " aux_out" : "", d->dact ? " dact" : "",
             d->resid ? " resid" : "", d->drop_p > 0 ? " drop" : "", (d->k_live || d->m_live || d->k_len || d->m_len) ? " live" : "");
  if (rc != CST_OK) return rc;
  if (p.splits > 1 && d->defer_reduce) {
    // the slabs stay in the caller's workspace for cst_reduce_multi: only epilogues that are a plain conversion can be finished there
    CST_REQUIRE((nbatch == 1 || !d->colsum) && d->alpha == 1.0f && !d->bias && !d->act && !d->dact && !d->aux_out && !d->resid && d->drop_p == 0.0f && d->ldc == d->N &&
                    d->N % 8 == 0 && (!d->colsum || cs_fused),
                "cst_gemm: defer_reduce needs a plain epilogue and a dense C (batched: no colsum; the slabs are [batch][split][M][N])");
    return rc;
  }
  if (p.splits > 1) {
    const int64_t total = d->M * d->N;
    const int64_t work = (p.vec_epi && d->N % 8 == 0) ? total / 8 : total;  // items the reduce kernel's threads walk
    int blocks = (int)(cst_ceil_div(work, 256) < 2048 ? cst_ceil_div(work, 256) : 2048);
    dim3 grid(blocks, (unsigned)nbatch);
    if (d->dtype == CST_BF16) hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(splitk_reduce_kernel<float>, grid, dim3(256), 0, s, p);
    rc = cst_check_launch("cst_gemm split-k reduce");
  }
  if (rc == CST_OK && d->colsum && !cs_fused) {
    // the 16-wave / DMA configurations do not build the by-product: two-launch column sum over A [K rows, M columns], its row-chunk
    // partials behind the split-K slabs of the caller's workspace; the live K stamps are the same 64-row blocks
    void* cws = (char*)d->workspace + (p.splits > 1 ? (int64_t)p.splits * d->M * d->N * (int64_t)sizeof(float) : 0);
    rc = p.k_live ? cst_colsum_typed_live(d->A, d->lda, d->colsum, cws, d->K, d->M, d->dtype, d->dtype, p.k_live, p.k_epoch, stream)
                  : cst_colsum_typed(d->A, d->lda, d->colsum, cws, d->K, d->M, d->dtype, d->dtype, stream);
  }
  return rc;
}

// (A row-split dispatch — whole rounds of 256 x 256 tiles to the persistent kernel, the remaining row slab to the 128 x 128
// configurations — was measured for the N = 768 family (564 tiles = 2.2 rounds): 0.241 vs 0.244 ms, no gain; not kept.
// Second attempt, the slab kept on the persistent kernel as a split-K (x3-4) launch + reduce epilogue: fc2 fwd 0.275 vs 0.278 ms,
// out_proj 0.105 vs 0.093, qkv dX 0.215 vs 0.189 — the 20 %-occupied last round is not a full round's cost (the 52 busy CUs run
// with the whole chip's bandwidth and clock budget), and the slabs + reduce launch cost more than it saves; not kept.)
extern "C" int cst_gemm(const cst_gemm_desc* d, cst_stream stream) { return gemm_one(d, stream, 0); }
