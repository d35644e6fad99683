// gemm8p.hip — the large-problem bf16 GEMM of libcst_hip: 256 x 256 x 64 tile, 8 waves (2 x 4), v_mfma_f32_32x32x16_bf16,
// operands streamed HBM/L2 -> LDS by buffer_load ... lds (16 B per lane, no staging registers, out-of-range lanes read 0),
// an 8-phase K loop (2 K tiles x 4 phases) with counted vmcnt: three 16-KiB half-tiles stay in flight across barriers and
// the DMA queue is never drained inside the loop.
//
// LDS (128 KiB): 2 K-tile buffers x 4 half-tile images of 16 KiB in CONSUMPTION order  B0 | A0 | B1 | A1
//   (A0 = rows [0,128) of the 256-row A tile, A1 = rows [128,256); same for B columns).
//   k-major operand  image [128 rows][128 B], 16-B chunk c of row r stored at chunk c ^ ((r >> 1) & 7) -> ds_read_b128
//                    conflict-free; a lane still fetches one row's chunks from 8 consecutive lanes (full 128-B lines).  (A
//                    chunk-major image with immediate-only addressing measured 20 % slower: its DMA touches 8 lines per 8 lanes.)
//   mn-major operand image: 16 DMA groups of 4 k-rows x 256 B placed 1088 B apart (64 B of padding per group); k-row
//                    k = 16a + 4b + c lives in group 4a + c, row slot b.  The four consecutive k-rows one ds_read_b64_tr_b16
//                    gathers (c = 0..3) then sit in four groups whose bases differ by 64 B mod 256 -> conflict-free, and every
//                    fragment address is one per-lane base + compile-time immediates (no swizzle arithmetic, no extra VGPRs).
//   The DMA writes lane-linear (wave base + lane * 16): the layouts above are produced purely by choosing which source
//   (row, chunk) each lane fetches.  Rows / columns / k beyond the matrix are fetched as zeros (descriptor range check).
// Wave (wm, wn) owns rows {h*128 + wm*64 + [0,64)} x cols {h'*128 + wn*32 + [0,32)}, h,h' in {0,1}: four 64 x 32 quadrants,
//   one per phase:   ph1 reads B0,A0 -> Q00 | ph2 reads B1 -> Q01 | ph3 reads A1 -> Q11 | ph4 reads nothing -> Q10.
// Every phase = { ds_reads ; stage one half-tile (2 DMA / lane) ; s_barrier ; lgkmcnt(0) ; 8 MFMA ; s_barrier }.  The wm = 1
//   waves run one barrier behind the wm = 0 waves (they share SIMDs pairwise), so one group's MFMA section overlaps the other
//   group's LDS/DMA section.
// Stream order of half-tiles s = 4*tile + q (q: 0=B0 1=A0 2=B1 3=A1); 7 are staged in the prologue, phase j stages s = 7 + j.
//   WAR: a region is restaged >= 2 phases after its last ds_read (B0: 1 phase, its reads are retired by lgkmcnt(8) before ph1's
//   first barrier).  RAW: vmcnt(6) in ph4 (before its first barrier) retires everything but the 3 newest half-tiles, i.e. the
//   whole next K tile, which is first read one phase later.
// Persistent: the grid is one workgroup per CU.  A workgroup's first work item (batch/split z, output tile) is v = blockIdx.x; the
//   following ones are CLAIMED from a per-XCD counter (v = gridDim.x + xcd + 8 * atomicAdd(&sched[xcd], 1)), one item ahead: the
//   claim for the item after next is issued while the next item's first K tile streams in, and is read back after the next K
//   loop.  The item order inside an XCD is the one a static stride would give, but a workgroup that starts late — its CU was
//   held by another stream's kernel, e.g. an RCCL all-reduce overlapped with backward — simply takes fewer items instead of
//   doubling the launch time.  The last workgroup to leave zeroes the counters (sched == nullptr: static stride).
//   After an item's K loop the first K tile of the NEXT item is already streaming into buffer 0 while the epilogue runs out of
//   the buffer-1 region, and the epilogue's global stores drain under the next item's MFMAs.
// Tail: with T tiles on a grid of G workgroups the last round holds r = T mod G tiles; when that round would be at most half full
//   (N = 768 at 31 760 rows: 375 tiles = one full round and 119 tiles, i.e. 1.46 rounds of work in the time of 2) its tiles are cut
//   into two 128-row halves — 2 r items for G workgroups.  A half item is the same pipeline with a 128-row A tile: the A1 half-tile
//   DMAs are issued out of range (zero fill, no memory traffic: every counted vmcnt and barrier stays as it is), phases 3 and 4 skip
//   their LDS reads and MFMAs, the epilogue runs one pass.  The K order per output element is unchanged: same bits.
// Epilogue: MFMA operands are swapped (D = B-frag x A-frag), so a lane holds 4 CONSECUTIVE output columns of one row per
//   register quad.  bf16 outputs: alpha/bias in registers -> one bf16 rounding (this is the pre-activation z) -> ds_write_b64
//   into a [128][256] bf16 image (2 passes) -> row-contiguous 16-byte read-back, activation / act' / residual on the way out.
//   fp32 outputs and split-K slabs: fp32 image, 64 rows per pass.
#include "gemm_common.h"
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace {
using namespace cstg;
using T = bf16_t;

constexpr int BM = 256, BN = 256, BK = 64, NTHREADS = 512;
static_assert(BM <= cstg::GEMM_MAX_BM && cstg::GEMM_MAX_BM % BM == 0, "conv0.hip (cst_conv_row_limits) sizes live frames from GEMM_MAX_BM");
constexpr int QB = 17408;           // one half-tile image (k-major images use the first 16 KiB)
constexpr int GSTRIDE = 1088;       // byte distance of the DMA groups of an mn-major image (64 B pad)
constexpr int KSTRIDE = 1024;       // k-major images are dense
constexpr int BUFB = 4 * QB;        // one K tile: B0 | A0 | B1 | A1
constexpr int LDS_BYTES = 2 * BUFB;
constexpr unsigned OOB = 0x80000000u;  // >= num_records of every descriptor: the DMA writes zeros

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// Epilogue LDS traffic goes through inline asm: hipcc orders every C++-visible LDS access behind pending LDS-DMA (the next
// item's first K tile, in flight by design) with an s_waitcnt vmcnt, and inside the read-back loop that wait also drains the
// previous iteration's global store — 17.7 k cycles per 256 x 256 tile for a plain store.  The image region never overlaps a
// DMA target, so only the LDS counter has to be honoured (explicit lgkmcnt waits at the call sites).
__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ void lds_write_b64(unsigned addr, uint2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write_b128(unsigned addr, f32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 lds_read_b128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128_off(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, 0, 0, 0);
}

// MODE selects the epilogue code compiled into an instantiation (the launcher picks it from the descriptor):
//   0  everything (fp32 outputs, split-K slabs, row bias, two extra operands, ...)
//   1  bf16 output through the LDS image, no extra operand: plain / bias / activation / pre-activation output / dropout
//   2  bf16 output, exactly ONE extra operand (act'(aux_in) or a residual), no pre-activation output
// The hot Linear shapes are all mode 1 or 2: without the general path's parameters the register allocator spills 20-30 scalars
// instead of 160, and the operand registers of mode 2 never weigh on the others.
template <bool AK, bool BKM, int MODE>
__global__ __launch_bounds__(NTHREADS) void gemm8p_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  // the wave number as a SCALAR: the DMA destinations (M0 values) derived from it are then scalar arithmetic instead of six VGPRs kept
  // across the K loop and a v_readfirstlane per DMA
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lrow = lane & 31, hi = lane >> 5;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int total = p.nitems;
  const int ktiles = (int)((p.K + BK - 1) / BK);
  const int per = (ktiles + p.splits - 1) / p.splits;
  // The by-value argument block is re-read from the kernarg segment (scalar loads through an opaque pointer) at each use site
  // outside the K loop, so that the ~70 SGPRs of launch parameters are not kept live across the MFMA loop.
  typedef const __attribute__((address_space(4))) GemmParams* kparg_t;
  kparg_t kp = (kparg_t)__builtin_amdgcn_kernarg_segment_ptr();

  // ---- lane constants of the DMA source map ----
  // k-major: group g covers rows 8g..8g+7 of the half; lane -> (row 8g + lane/8, stored chunk lane%8 = source chunk ^ swizzle)
  const int ck = (lane & 7) ^ (((4 * wave) + (lane >> 4)) & 7);   // ((8g + lane/8) >> 1) & 7 is the same for g and g + 8
  // mn-major: group G = 4a + c holds k-rows 16a + 4b + c (b = row slot = lane/16); lane -> chunk lane%16 of that row
  const int klm = 16 * (wave >> 2) + 4 * (lane >> 4) + (wave & 3);
  const int kla = AK ? ck * 8 : klm, klb = BKM ? ck * 8 : klm;    // k inside a K tile (mn-major group 1: +32)
  const int pa = AK ? 8 * wave + (lane >> 3) : (lane & 15) * 8;   // row (k-major) / first column (mn-major)
  const int pb = BKM ? 8 * wave + (lane >> 3) : (lane & 15) * 8;  //   inside a half, group 0
  const unsigned step_a = AK ? (unsigned)(BK * 2) : (unsigned)(BK * p.lda * 2);
  const unsigned step_b = BKM ? (unsigned)(BK * 2) : (unsigned)(BK * p.ldb * 2);
  constexpr bool EXF = MODE == 2;
  const bool fast_epi = MODE != 0 || (!p.c_f32 && p.splits == 1 && p.vec_epi && (p.N % 8) == 0 && p.bias_mode != CST_BIAS_ROW &&
                                (p.bias_mode == CST_BIAS_NONE || p.alpha == 1.0f));
  const bool bias_in_acc = fast_epi && p.bias_mode == CST_BIAS_COL;  // the bias row seeds the accumulators
  char* const bias_lds = smem + 2 * BUFB;                              // 512 B: bias[n0 .. n0 + 256) of the streamed item

  // ---- state of the item whose operands are being streamed ----
  __amdgpu_buffer_rsrc_t ra, rb;
  unsigned va0 = 0, vb0 = 0;     // per-lane byte offset (half 0, group 0) from the descriptor base
  int nt = 0, krem0 = 0, mrem = 0, nrem = 0;
  int hf = 0;                    // the streamed item is a 128-row half tile
  unsigned lda2 = 0, ldb2 = 0;   // leading dimensions in bytes
  int64_t m0 = 0, n0 = 0, cofs = 0, bofs = 0;
  int zcur = 0;

  // work item v -> tile (XCD-aware grouped order, same map as gemm.hip) and operand descriptors
  auto setup = [&](int v) {
    asm volatile("" : "+s"(kp));
    struct { int64_t M, N, K, lda, ldb, batch1, sa0, sa1, sb0, sb1, sc0, sc1, sbias0, sbias1; const void *A, *B, *bias; int splits, tiles_m, tiles_n; } p;
    p.M = kp->M; p.N = kp->N; p.K = kp->K; p.lda = kp->lda; p.ldb = kp->ldb; p.batch1 = kp->batch1;
    p.sa0 = kp->sa0; p.sa1 = kp->sa1; p.sb0 = kp->sb0; p.sb1 = kp->sb1; p.sc0 = kp->sc0; p.sc1 = kp->sc1;
    p.sbias0 = kp->sbias0; p.sbias1 = kp->sbias1; p.A = kp->A; p.B = kp->B; p.bias = kp->bias;
    p.splits = kp->splits; p.tiles_m = kp->tiles_m; p.tiles_n = kp->tiles_n;
    auto fdiv = [](int n, unsigned long long magic) { return (int)(((unsigned long long)(unsigned)n * magic) >> 40); };
    int half = -1, vt = v;
    if (v >= kp->half_from) {  // (only launches with nz == 1 have half items)
      const int o = v - kp->half_from;
      half = o & 1;
      vt = kp->half_from + (o >> 1);
    }
    hf = half >= 0 ? 1 : 0;
    const int z = kp->nz > 1 ? fdiv(vt, kp->magic_ntiles) : 0;
    int id = vt - z * ntiles;
    {
      const int q = ntiles >> 3, r = ntiles & 7, xcd = id & 7, loc = id >> 3;
      id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    if (kp->mrot > 0) {
      // row-limited batches with too few tile groups for the deal below: batch z starts its walk z * mrot positions further on, so
      // the XCD that holds the dead tail of one batch holds the live head of the next (batches are sorted by length: the offsets
      // z mod 8 give every XCD one batch of each length class)
      int zr = z * kp->mrot;
      zr -= fdiv(zr, kp->magic_ntiles) * ntiles;
      id += zr;
      if (id >= ntiles) id -= ntiles;
    }
    const int GROUP_M = kp->group_m, per_group = kp->per_group, gshift = kp->group_shift;
    int grp = fdiv(id, kp->magic_per_group);
    const int rem = id - grp * per_group;
    if (kp->gperm_full > 0 && grp < kp->gperm_full) {
      // Per-batch row limits (m_len): the tiles behind a batch's limit are the TAIL of its m range, and the chunk map above hands
      // XCD x the x-th eighth of every batch's m range — with utterance lengths between a third of the longest and the longest,
      // XCD 0 .. 2 never meet a dead tile and XCD 7 meets almost nothing else; the claim counters are per XCD, so the launch took
      // as long as if nothing had been skipped (conv layer 1 forward inside the update: 2.44 ms, all rows live standalone: 2.57).
      // Instead XCD x gets groups x, x + 8, x + 16, ...: `grp` counts the groups column by column through an 8-column arrangement
      // of the real group numbers (columns 0 .. r - 1 hold q + 1 groups, the others q), so that the contiguous run of an XCD is
      // (about) one column.  A group stays on one XCD: the L2 sees the same 4 x tiles_n patches as before.  Same tiles, same
      // arithmetic per tile: same bits.
      const int q = kp->gperm_q, r = kp->gperm_r, big = (q + 1) * r;
      int col, row;
      if (grp < big) { col = fdiv(grp, kp->magic_gq1); row = grp - col * (q + 1); }
      else { const int t = grp - big, c = fdiv(t, kp->magic_gq); col = r + c; row = t - c * q; }
      grp = row * 8 + col;
    }
    const int gm0 = grp * GROUP_M;
    const int gsz = (p.tiles_m - gm0 < GROUP_M) ? (p.tiles_m - gm0) : GROUP_M;
    int tm, tn;
    if (gsz == GROUP_M && gshift >= 0) { tm = gm0 + (rem & (GROUP_M - 1)); tn = rem >> gshift; }
    else { tm = gm0 + rem % gsz; tn = rem / gsz; }  // the last, partial group of a launch
    m0 = (int64_t)tm * BM + (half == 1 ? 128 : 0);
    n0 = (int64_t)tn * BN;
    zcur = z;
    const int split = p.splits > 1 ? z % p.splits : 0;
    const int64_t bidx = p.splits > 1 ? z / p.splits : z;
    const int64_t b0 = p.batch1 > 1 ? bidx / p.batch1 : bidx, b1 = p.batch1 > 1 ? bidx % p.batch1 : 0;
    const T* A = (const T*)p.A + b0 * p.sa0 + b1 * p.sa1;
    const T* B = (const T*)p.B + b0 * p.sb0 + b1 * p.sb1;
    cofs = b0 * p.sc0 + b1 * p.sc1;
    bofs = b0 * p.sbias0 + b1 * p.sbias1;
    const int kt0 = split * per;
    const int kt1 = (kt0 + per < ktiles) ? kt0 + per : ktiles;
    nt = kt1 > kt0 ? kt1 - kt0 : 0;
    if (kp->m_live) {  // all four 64-row blocks of this tile's A rows stamped dead: no K loop, the epilogue runs on zero accumulators
      typedef const __attribute__((address_space(4))) uint32_t* cptr_t;
      cptr_t ml = (cptr_t)kp->m_live;
      const uint32_t ep = kp->m_epoch;
      const int t0 = (int)(m0 >> 6), tn = (int)((p.M + 63) >> 6);
      bool live = false;
#pragma unroll
      for (int i = 0; i < 4; ++i) live |= (i < 2 || !hf) && (t0 + i < tn) && ml[t0 + i < tn ? t0 + i : t0] == ep;
      if (!live) nt = 0;
    }
    if (kp->m_len) {  // rows of this batch from m_len[b0] on: all zero in A, unread in C
      typedef const __attribute__((address_space(4))) int32_t* iptr_t;
      if (m0 >= (int64_t)((iptr_t)kp->m_len)[b0]) nt = 0;
    }
    krem0 = (int)p.K - kt0 * BK;
    const T* abase = AK ? A + m0 * p.lda + (int64_t)kt0 * BK : A + (int64_t)kt0 * BK * p.lda + m0;
    const T* bbase = BKM ? B + n0 * p.ldb + (int64_t)kt0 * BK : B + (int64_t)kt0 * BK * p.ldb + n0;
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)abase, (short)0, (int)OOB, 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc((void*)bbase, (short)0, (int)OOB, 0x00020000);
    if (bias_in_acc) {  // bias[n0 .. n0+256) -> LDS (columns >= N read 0); consumed when this item's accumulators are seeded
      const int64_t left = p.N - n0;
      const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.bias + bofs + n0), (short)0,
                                                                              (int)((left < 256 ? left : 256) * 2), 0x00020000);
      dma16(rbias, bias_lds, (unsigned)lane * 16);
    }
    lda2 = (unsigned)(p.lda * 2);
    ldb2 = (unsigned)(p.ldb * 2);
    mrem = (int)(p.M - m0 < BM ? p.M - m0 : BM);
    if (hf && mrem > 128) mrem = 128;
    if (mrem <= 0) { mrem = 0; nt = 0; }  // the lower half of a tile that ends in its upper half: nothing to compute or store
    nrem = (int)(p.N - n0 < BN ? p.N - n0 : BN);
    va0 = AK ? (unsigned)pa * lda2 + (unsigned)kla * 2 : (unsigned)klm * lda2 + (unsigned)pa * 2;
    vb0 = BKM ? (unsigned)pb * ldb2 + (unsigned)klb * 2 : (unsigned)klm * ldb2 + (unsigned)pb * 2;
  };

  // stage half-tile Q of K tile `tau` into buffer BUF
  auto stage = [&](auto qc, auto bufc, int tau) {
    constexpr int Q = decltype(qc)::value, BUF = decltype(bufc)::value;
    constexpr bool IS_A = (Q & 1) != 0;
    constexpr int H = Q >> 1;
    constexpr bool KM = IS_A ? AK : BKM;
    constexpr int GS = KM ? KSTRIDE : GSTRIDE;
    const int krem = krem0 - tau * BK;
    const unsigned ld2 = IS_A ? lda2 : ldb2;
    const unsigned base = (IS_A ? va0 : vb0) + (unsigned)tau * (IS_A ? step_a : step_b) + (KM ? (unsigned)(H * 128) * ld2 : (unsigned)(H * 256));
    const int pos = (IS_A ? pa : pb) + H * 128;   // row / first column of this lane inside the 256-wide tile (group 0)
    const int lim = IS_A ? mrem : nrem;
    char* dst = smem + BUF * BUFB + Q * QB + wave * GS;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned v0 = base + (KM ? (unsigned)(64 * j) * ld2 : (unsigned)(32 * j) * ld2);
      const bool ok = ((IS_A ? kla : klb) + (KM ? 0 : 32 * j) < krem) && (pos + (KM ? 64 * j : 0) < lim);
      dma16(IS_A ? ra : rb, dst + j * 8 * GS, ok ? v0 : OOB);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // ---- fragment reads ----
  // k-major: per-lane byte offset of (row, chunk 2kk + hi) inside a half image; rows wm*64 + i*32 + lrow share one swizzle
  const int swr = (lrow >> 1) & 7;
  const int arow = (wm * 64 + lrow) * 128, brow = (wn * 32 + lrow) * 128;
  // mn-major: lane s of a 16-lane group addresses the 8-byte piece (k-row c = s/4, mn 4*(s%4)..+3) of its 4 x 16 block
  const int trl = ((lane & 15) >> 2) * GSTRIDE + hi * 512 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  auto read_tr = [&](Frag<T>& f, const char* q) {
    const v4s_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(q));
    const v4s_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(q + 256));
    u16x8 t;
    t[0] = (unsigned short)x[0]; t[1] = (unsigned short)x[1]; t[2] = (unsigned short)x[2]; t[3] = (unsigned short)x[3];
    t[4] = (unsigned short)y[0]; t[5] = (unsigned short)y[1]; t[6] = (unsigned short)y[2]; t[7] = (unsigned short)y[3];
    f.v = __builtin_bit_cast(bf16x8, t);
  };
  auto read_a = [&](Frag<T>& f, const char* img, int i, int kk) {
    if (AK) f.v = *reinterpret_cast<const bf16x8*>(img + arow + i * 4096 + (((2 * kk + hi) ^ swr) << 4));
    else read_tr(f, img + trl + wm * 128 + i * 64 + kk * 4 * GSTRIDE);
  };
  auto read_b = [&](Frag<T>& f, const char* img, int kk) {
    if (BKM) f.v = *reinterpret_cast<const bf16x8*>(img + brow + (((2 * kk + hi) ^ swr) << 4));
    else read_tr(f, img + trl + wn * 64 + kk * 4 * GSTRIDE);
  };

  // acc[m tile: h*2 + i][n tile: h'] in the SWAPPED layout (A-operand = B fragment): lane -> output row (lrow), register r ->
  // output column 8*(r/4) + 4*hi + r%4 of the 32 x 32 tile.
  f32x16 acc[4][2];
  Frag<T> fa[2][4], fb0[4], fb1[4];

  auto mfma_quadrant = [&](auto hmc, auto hnc) {
    constexpr int HM = decltype(hmc)::value, HN = decltype(hnc)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma16(acc[HM * 2 + i][HN], HN ? fb1[kk] : fb0[kk], fa[i][kk]);
    __builtin_amdgcn_s_setprio(0);
  };

  // one K tile = 4 phases; BUF = the buffer holding tile t
  auto tile_body = [&](auto bufc, int t, int S, const bool half) {
    constexpr int BUF = decltype(bufc)::value;
    using BC = std::integral_constant<int, BUF>;
    using BN_ = std::integral_constant<int, BUF ^ 1>;
    const char* img = smem + BUF * BUFB;
    const int s0 = 7 + 4 * t;
    // ---- phase 1: B0, A0 -> Q00 ----
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) read_b(fb0[kk], img + 0 * QB, kk);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i) read_a(fa[i][kk], img + 1 * QB, i, kk);
    __builtin_amdgcn_sched_barrier(0);
    if (s0 < S) stage(I3{}, BN_{}, t + 1);
    wait_lgkm<AK ? 8 : 15>();  // the B0 reads (issued first) are retired: B0 may be restaged next phase
    __builtin_amdgcn_s_barrier();
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    mfma_quadrant(I0{}, I0{});
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: B1 -> Q01 ----
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) read_b(fb1[kk], img + 2 * QB, kk);
    __builtin_amdgcn_sched_barrier(0);
    if (s0 + 1 < S) stage(I0{}, BC{}, t + 2);
    __builtin_amdgcn_s_barrier();
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    mfma_quadrant(I0{}, I1{});
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: A1 -> Q11 (a half item has no A1: staging, waits and barriers as ever, no reads, no MFMAs) ----
    if (!half) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) read_a(fa[i][kk], img + 3 * QB, i, kk);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s0 + 2 < S) stage(I1{}, BC{}, t + 2);
    __builtin_amdgcn_s_barrier();
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    if (!half) mfma_quadrant(I1{}, I1{});
    __builtin_amdgcn_s_barrier();
    // ---- phase 4: no reads -> Q10; the next K tile must have landed ----
    if (s0 + 3 < S) stage(I2{}, BC{}, t + 2);
    if (s0 + 4 <= S) wait_vm<6>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (!half) mfma_quadrant(I1{}, I0{});
    __builtin_amdgcn_s_barrier();
  };

  // the epilogue image lives in the buffer-1 region (buffer 0 is receiving the next item's first K tile)
  char* const ebase = smem + BUFB;
  const unsigned ebase_off = lds_off(ebase);

  int v = blockIdx.x;
  if (v >= total) return;
  // dynamic item claims (see the header).  Nothing of this is kept in registers across a K loop: the state pointer is re-read
  // from the kernarg segment at each use (nullptr = static stride; the launcher passes it only for multi-round launches on a
  // grid that is a multiple of 8), the XCD index is recomputed from blockIdx.
  const unsigned slot_off = lds_off(smem + 2 * BUFB + 1024);  // two 4-byte slots behind the bias row's DMA footprint, alternating per item
  int par = 0;
  // claim_issue: thread 0 fires the counter increment right after the next item's first-K-tile DMA; claim_land publishes the
  // claimed item through LDS just before the epilogue's first global access, i.e. after the accumulator image has been written
  // (LDS only) — the compiler's wait for the returned value there also covers the DMA issued before it, which have had the same
  // time to land.  The value is live only between those two points (never across a K loop, where no VGPR is free).  (A
  // hand-written atomic with a counted vmcnt was tried first: the register allocator spills the asm's output before it has
  // arrived.)
  int claim = 0;
  auto claim_issue = [&]() {
    int* const sched = p.sched;
    if (sched && tid == 0) claim = __hip_atomic_fetch_add(sched + (blockIdx.x & 7), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto claim_land = [&]() {
    // slot `par ^ 1` was last read one item ago (barriers in between); the LDS store is inline asm for the reason given above
    if (p.sched && tid == 0) {
      const int item = (int)gridDim.x + (int)(blockIdx.x & 7) + 8 * claim;
      asm volatile("ds_write_b32 %0, %1" ::"v"(slot_off + 4u * (unsigned)(par ^ 1)), "v"(item) : "memory");
    }
    par ^= 1;
  };
  setup(v);
  {
    const int S = 4 * nt;
    if (0 < S) stage(I0{}, I0{}, 0);
    if (1 < S) stage(I1{}, I0{}, 0);
    if (2 < S) stage(I2{}, I0{}, 0);
    if (3 < S) stage(I3{}, I0{}, 0);
  }
  claim_issue();
  claim_land();
#ifdef CST_TRACE
  long long* trace = (p.splits == 1 && p.ws) ? reinterpret_cast<long long*>(p.ws) + (int64_t)blockIdx.x * 64 : nullptr;
  int titem = 0;
#define CST_STAMP(k) do { if (trace && tid == 0 && titem < 8) trace[titem * 8 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define CST_STAMP(k) do { } while (0)
#endif
  while (true) {
    CST_STAMP(0);
    // ---- the item being computed (its first K tile is in flight or landed) ----
    const int64_t e_m0 = m0, e_n0 = n0, e_cofs = cofs, e_bofs = bofs;
    const int e_z = zcur, e_nt = nt;
    const bool e_hf = hf != 0;
    const int S = 4 * e_nt;
    if (4 < S) stage(I0{}, I1{}, 1);
    if (5 < S) stage(I1{}, I1{}, 1);
    if (6 < S) stage(I2{}, I1{}, 1);
    if (7 <= S) wait_vm<6>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (bias_in_acc) {
#pragma unroll
      for (int hn = 0; hn < 2; ++hn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint2 raw = *reinterpret_cast<const uint2*>(bias_lds + (hn * 128 + wn * 32 + 8 * g + 4 * hi) * 2);
          const float b0 = __uint_as_float(raw.x << 16), b1 = __uint_as_float(raw.x & 0xffff0000u);
          const float b2 = __uint_as_float(raw.y << 16), b3 = __uint_as_float(raw.y & 0xffff0000u);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[i][hn][4 * g] = b0; acc[i][hn][4 * g + 1] = b1; acc[i][hn][4 * g + 2] = b2; acc[i][hn][4 * g + 3] = b3;
          }
        }
      wait_lgkm<0>();
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    }
    if (wm == 1) __builtin_amdgcn_s_barrier();  // stagger: the wm = 1 waves run one barrier behind
    CST_STAMP(1);
    for (int t = 0; t < e_nt; t += 2) {
      tile_body(I0{}, t, S, e_hf);
      if (t + 1 < e_nt) tile_body(I1{}, t + 1, S, e_hf);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    CST_STAMP(2);

    // ---- start streaming the next item before this one's epilogue ----
    {
      int item;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(item) : "v"(slot_off + 4u * (unsigned)par) : "memory");
      v = __builtin_amdgcn_readfirstlane(p.sched ? item : v + (int)gridDim.x);
    }
    const bool has_next = v < total;
    if (has_next) {
      // the claim for the item after the one set up here (published in the epilogue, read back after that item's K loop) goes out
      // FIRST: the compiler waits for its return value right away (it has to park it across the epilogue), and here that wait covers
      // only the atomic's own round trip — behind the four DMA below it also covered their arrival (1.6 k cycles of wave 0, for which
      // the other seven waves then waited at the epilogue's first barrier)
      claim_issue();
      setup(v);
      CST_STAMP(5);
      const int Sn = 4 * nt;
      if (0 < Sn) stage(I0{}, I0{}, 0);
      if (1 < Sn) stage(I1{}, I0{}, 0);
      if (2 < Sn) stage(I2{}, I0{}, 0);
      if (3 < Sn) stage(I3{}, I0{}, 0);
      CST_STAMP(6);
    }

    CST_STAMP(3);
    asm volatile("" : "+s"(kp));
    GemmParams q;  // epilogue parameters, re-read after the K loop (only the fields used below are materialised)
    q.M = kp->M; q.N = kp->N; q.C = kp->C; q.ldc = kp->ldc; q.bias = kp->bias; q.bias_mode = kp->bias_mode; q.act = kp->act;
    q.aux_out = kp->aux_out; q.ld_aux_out = kp->ld_aux_out; q.dact = kp->dact; q.aux_in = kp->aux_in; q.ld_aux_in = kp->ld_aux_in;
    q.resid = kp->resid; q.ld_resid = kp->ld_resid; q.alpha = kp->alpha; q.splits = kp->splits; q.ws = kp->ws;
    q.c_f32 = kp->c_f32; q.vec_epi = kp->vec_epi; q.drop_thr = kp->drop_thr; q.drop_key = kp->drop_key; q.drop_scale = kp->drop_scale; q.drop_row0 = kp->drop_row0;
    if (fast_epi && e_nt == 0 && q.bias_mode == CST_BIAS_NONE && !q.resid) {
      // A tile whose K loop was skipped (cst_gemm_desc.m_len / m_live: its rows of A are all zero) with an epilogue that maps 0 to 0
      // — no bias, no residual; act(0) = 0, 0 * act'(z) = 0, 0 * mask = 0 — stores zeros straight from registers: no accumulator
      // image, no read-back, no operand loads, no activation arithmetic (the conv stack's frames past an utterance's end: a third
      // of the layer-1 items at the bench's lengths, each of which paid the full 17 k-cycle GELU epilogue to write zeros).
      if (has_next) claim_land();
      const u32x4 z4 = {0u, 0u, 0u, 0u};
      const int64_t col = e_n0 + (tid & 31) * 8;
      if (col < q.N) {
        T* const cb = (T*)q.C + e_cofs + col;
        T* const xb = q.aux_out ? (T*)q.aux_out + e_cofs + col : nullptr;
#pragma unroll 4
        for (int it = 0; it < (e_hf ? 8 : 16); ++it) {
          const int64_t row = e_m0 + (tid >> 5) + 16 * it;
          if (row < q.M) {
            __builtin_nontemporal_store(z4, reinterpret_cast<u32x4*>(cb + row * q.ldc));
            if (xb) __builtin_nontemporal_store(z4, reinterpret_cast<u32x4*>(xb + row * q.ld_aux_out));
          }
        }
      }
    } else if (fast_epi) {
      // bf16 image [128 rows][256 cols], row stride 528 B; pass hm = rows [128 hm, 128 hm + 128) of the tile.
      // Global LOADS and global STORES never share a loop: with stores pending, waiting for a load costs a full vmcnt(0)
      // (loads and stores retire out of order with respect to each other), i.e. one store round trip per iteration.
      //   loop A (only with an aux_in / residual operand): z, operand -> final value, written back into the image in place;
      //   loop B: image -> [pre-activation output] -> [activation, dropout] -> C.
      constexpr int ERS = 528;
      const T* exsrc = MODE == 1 ? nullptr : (q.dact ? (const T*)q.aux_in : (const T*)q.resid);
      const int64_t exld = q.dact ? q.ld_aux_in : q.ld_resid;
      const bool both = q.dact && q.resid;
      // EXF (one extra operand, no pre-activation output): the eight operand vectors of a pass are requested in one burst BEFORE
      // that pass's image traffic — pass 0's here, pass 1's between pass 0's arithmetic and pass 0's stores — so their latency
      // hides behind LDS work, and inside a pass the pieces are finished in registers first and stored afterwards (a load is
      // never waited for with a store of the same pass pending).
      constexpr bool ex_fast = EXF;
      u32x4 exq[8];
      const T* ex_b = nullptr;
      if (ex_fast) {
        int64_t colc = e_n0 + (tid & 31) * 8;
        colc = colc < q.N ? colc : q.N - 8;
        ex_b = exsrc + e_cofs + colc;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          int64_t row = e_m0 + (tid >> 5) + 16 * it;
          row = row < q.M ? row : q.M - 1;
          exq[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ex_b + row * exld));
        }
      }
#pragma unroll
      for (int hm = 0; hm < 2; ++hm) {
        if (hm && e_hf) break;                 // a half item has one pass (rows 128.. belong to the other half's item)
        if (hm) __builtin_amdgcn_s_barrier();  // previous pass fully read back
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int hn = 0; hn < 2; ++hn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
              bf16x4_t pk;
#pragma unroll
              for (int e = 0; e < 4; ++e) pk[e] = static_cast<__bf16>(acc[hm * 2 + i][hn][4 * g + e] * q.alpha);
              lds_write_b64(ebase_off + (wm * 64 + i * 32 + lrow) * ERS + (hn * 128 + wn * 32 + 8 * g + 4 * hi) * 2, __builtin_bit_cast(uint2, pk));
            }
        wait_lgkm<0>();
        __builtin_amdgcn_s_barrier();
        if (hm == 0 && has_next) claim_land();
        if (exsrc && !ex_fast) {
#pragma unroll 2
          for (int it = 0; it < 8; ++it) {
            const int vi = tid + NTHREADS * it;
            const int rl = vi >> 5, cl = (vi & 31) * 8;
            const int64_t row = e_m0 + hm * 128 + rl, col = e_n0 + cl;
            if (row >= q.M || col >= q.N) continue;
            const u32x4 exv = *reinterpret_cast<const u32x4*>(exsrc + e_cofs + row * exld + col);
            const u32x4 zraw = lds_read_b128(ebase_off + rl * ERS + cl * 2);
            wait_lgkm<0>();
            __builtin_amdgcn_sched_barrier(0);
            if (q.aux_out) *reinterpret_cast<u32x4*>((T*)q.aux_out + e_cofs + row * q.ld_aux_out + col) = zraw;  // (not on the hot path)
            float x[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              x[2 * e] = __uint_as_float(zraw[e] << 16);
              x[2 * e + 1] = __uint_as_float(zraw[e] & 0xffff0000u);
            }
            if (q.act == CST_ACT_RELU) {
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.0f);
            } else if (q.act == CST_ACT_GELU) {
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] = gelu_t<T>(x[e]);
            }
            if (q.drop_thr) cst_drop8(x, q.drop_key, (uint64_t)((row + q.drop_row0) * q.N + col), q.drop_thr, q.drop_scale);
            if (q.dact) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                x[2 * e] *= dact_t<T>(__uint_as_float(exv[e] << 16), q.dact);
                x[2 * e + 1] *= dact_t<T>(__uint_as_float(exv[e] & 0xffff0000u), q.dact);
              }
              if (both) {
                float rr[8];
                load8((const T*)q.resid + e_cofs + row * q.ld_resid + col, rr);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] += rr[e];
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                x[2 * e] += __uint_as_float(exv[e] << 16);
                x[2 * e + 1] += __uint_as_float(exv[e] & 0xffff0000u);
              }
            }
            bf16x8 ob;
#pragma unroll
            for (int e = 0; e < 8; ++e) ob[e] = static_cast<__bf16>(x[e]);
            lds_write_b128(ebase_off + rl * ERS + cl * 2, __builtin_bit_cast(f32x4, ob));
          }
          wait_lgkm<0>();  // loop B re-reads this thread's own slots
        }
        const bool post = !exsrc && (q.act != CST_ACT_NONE || q.drop_thr);
        // loop B.  The eight 16-byte pieces of this thread (rows 16 apart, same columns) are read back with ONE burst of ds_read_b128
        // (immediate offsets from one address) and stored as they arrive (counted lgkmcnt): a read -> wait -> store chain per piece
        // exposed one LDS round trip per iteration with only two waves per SIMD to cover it.
        {
          constexpr int RSTEP = 16 * ERS;  // byte distance of consecutive pieces in the image
          const int rl0 = tid >> 5, cl = (tid & 31) * 8;
          const unsigned a0 = ebase_off + rl0 * ERS + cl * 2;
          const int64_t col = e_n0 + cl, row0 = e_m0 + hm * 128 + rl0;
          const bool col_ok = col < q.N;
          u32x4 zr[8];
          zr[0] = lds_read_b128_off<0 * RSTEP>(a0); zr[1] = lds_read_b128_off<1 * RSTEP>(a0);
          zr[2] = lds_read_b128_off<2 * RSTEP>(a0); zr[3] = lds_read_b128_off<3 * RSTEP>(a0);
          zr[4] = lds_read_b128_off<4 * RSTEP>(a0); zr[5] = lds_read_b128_off<5 * RSTEP>(a0);
          zr[6] = lds_read_b128_off<6 * RSTEP>(a0); zr[7] = lds_read_b128_off<7 * RSTEP>(a0);
          T* const cbase = (T*)q.C + e_cofs + row0 * q.ldc + col;
          T* const xbase = (T*)q.aux_out + e_cofs + row0 * q.ld_aux_out + col;
          const bool want_aux = !exsrc && q.aux_out;
          if (ex_fast) {
            wait_lgkm<0>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
              const u32x4 zraw = zr[it], exv = exq[it];
              float x[8];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                x[2 * e] = __uint_as_float(zraw[e] << 16);
                x[2 * e + 1] = __uint_as_float(zraw[e] & 0xffff0000u);
              }
              if (q.act == CST_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.0f);
              } else if (q.act == CST_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = gelu_t<T>(x[e]);
              }
              if (q.drop_thr) cst_drop8(x, q.drop_key, (uint64_t)((row0 + 16 * it + q.drop_row0) * q.N + col), q.drop_thr, q.drop_scale);
              if (q.dact) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  x[2 * e] *= dact_t<T>(__uint_as_float(exv[e] << 16), q.dact);
                  x[2 * e + 1] *= dact_t<T>(__uint_as_float(exv[e] & 0xffff0000u), q.dact);
                }
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  x[2 * e] += __uint_as_float(exv[e] << 16);
                  x[2 * e + 1] += __uint_as_float(exv[e] & 0xffff0000u);
                }
              }
              bf16x8 ob;
#pragma unroll
              for (int e = 0; e < 8; ++e) ob[e] = static_cast<__bf16>(x[e]);
              zr[it] = __builtin_bit_cast(u32x4, ob);
            }
            if (hm == 0 && !e_hf) {  // pass 1's operand vectors, ahead of pass 0's stores
#pragma unroll
              for (int it = 0; it < 8; ++it) {
                int64_t row = e_m0 + 128 + (tid >> 5) + 16 * it;
                row = row < q.M ? row : q.M - 1;
                exq[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ex_b + row * exld));
              }
            }
#pragma unroll
            for (int it = 0; it < 8; ++it)
              if (col_ok && row0 + 16 * it < q.M)
                __builtin_nontemporal_store(zr[it], reinterpret_cast<u32x4*>(cbase + (int64_t)(16 * it) * q.ldc));
            continue;
          }
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            if (it == 0) wait_lgkm<7>(); else if (it == 1) wait_lgkm<6>(); else if (it == 2) wait_lgkm<5>(); else if (it == 3) wait_lgkm<4>();
            else if (it == 4) wait_lgkm<3>(); else if (it == 5) wait_lgkm<2>(); else if (it == 6) wait_lgkm<1>(); else wait_lgkm<0>();
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 zraw = zr[it];
            if (!(col_ok && row0 + 16 * it < q.M)) continue;
            T* cdst = cbase + (int64_t)(16 * it) * q.ldc;
            if (want_aux) __builtin_nontemporal_store(zraw, reinterpret_cast<u32x4*>(xbase + (int64_t)(16 * it) * q.ld_aux_out));
            if (!post) {
              __builtin_nontemporal_store(zraw, reinterpret_cast<u32x4*>(cdst));
              continue;
            }
            float x[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              x[2 * e] = __uint_as_float(zraw[e] << 16);
              x[2 * e + 1] = __uint_as_float(zraw[e] & 0xffff0000u);
            }
            if (q.act == CST_ACT_RELU) {
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.0f);
            } else if (q.act == CST_ACT_GELU) {
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] = gelu_t<T>(x[e]);
            }
            if (q.drop_thr) cst_drop8(x, q.drop_key, (uint64_t)((row0 + 16 * it + q.drop_row0) * q.N + col), q.drop_thr, q.drop_scale);
            {
              bf16x8 ob;
#pragma unroll
              for (int e = 0; e < 8; ++e) ob[e] = static_cast<__bf16>(x[e]);
              __builtin_nontemporal_store(__builtin_bit_cast(u32x4, ob), reinterpret_cast<u32x4*>(cdst));
            }
          }
        }
      }
    } else {
      // fp32 image, 64 rows x (BN + 4) per pass; pass ps = rows [64 ps, 64 ps + 64) = half ps/2 of the wm = ps%2 waves
      constexpr int LDC = BN + 4;
      constexpr int VPR = BN / 8;
      constexpr int ITERS = 64 * VPR / NTHREADS;
      float* wsp = q.splits > 1 ? q.ws + ((int64_t)e_z) * q.M * q.N : nullptr;
      const bool ws_vec = (q.N % 4) == 0;
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        if (ps == 2 && e_hf) break;  // a half item: rows [0, 128) only
        if (ps) __builtin_amdgcn_s_barrier();
        if (wm == (ps & 1)) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int hn = 0; hn < 2; ++hn)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                f32x4 q;
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = acc[(ps >> 1) * 2 + i][hn][4 * g + e];
                lds_write_b128(ebase_off + ((i * 32 + lrow) * LDC + hn * 128 + wn * 32 + 8 * g + 4 * hi) * 4, q);
              }
        }
        wait_lgkm<0>();
        __builtin_amdgcn_s_barrier();
        if (ps == 0 && has_next) claim_land();
#pragma unroll 2
        for (int itv = 0; itv < ITERS; ++itv) {
          const int vi = tid + NTHREADS * itv;
          const int rl = vi / VPR, cl = (vi % VPR) * 8;
          const int64_t row = e_m0 + 64 * ps + rl, col = e_n0 + cl;
          if (row >= q.M || col >= q.N) continue;
          float x[8];
          {
            const u32x4 ua = lds_read_b128(ebase_off + (rl * LDC + cl) * 4), ub = lds_read_b128(ebase_off + (rl * LDC + cl + 4) * 4);
            wait_lgkm<0>();
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 a = __builtin_bit_cast(f32x4, ua), b = __builtin_bit_cast(f32x4, ub);
            x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
          }
          const bool full = col + 8 <= q.N;
          if (wsp) {
            if (full && ws_vec) store8(wsp + row * q.N + col, x);
            else
              for (int e = 0; e < 8 && col + e < q.N; ++e) wsp[row * q.N + col + e] = x[e];
          } else if (full && q.vec_epi) {
            epilogue_store8<T>(q, e_cofs, e_bofs, row, col, x);
          } else {
            for (int e = 0; e < 8 && col + e < q.N; ++e) epilogue_store<T>(q, e_cofs, e_bofs, row, col + e, x[e]);
          }
        }
      }
    }
    CST_STAMP(4);
#ifdef CST_TRACE
    ++titem;
#endif
    if (!has_next) break;
    __builtin_amdgcn_s_barrier();  // the image is fully read back: buffer 1 may receive the next item's second K tile
  }
  int* const sched_exit = p.sched;
  if (int* const sched = sched_exit; sched && tid == 0) {
    // every workgroup passes here once, after its last (failed) claim: the last one re-arms the state for the next launch
    if (atomicAdd(sched + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) __hip_atomic_store(sched + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Scheduling state of the dynamic item claims: a ring of 64-byte slots per device, zeroed once; a launch takes the next slot,
// so launches in flight on different streams never share counters (each launch leaves its slot zeroed).
int* sched_slot() {
  constexpr int SLOTS = 1024, MAXDEV = 64;
  static int* ring[MAXDEV] = {};
  static std::atomic<unsigned> next[MAXDEV];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
  int* r = __atomic_load_n(&ring[dev], __ATOMIC_ACQUIRE);
  if (!r) {
    std::lock_guard<std::mutex> g(mu);
    r = ring[dev];
    if (!r) {
      void* q = nullptr;
      if (hipMalloc(&q, SLOTS * 64) != hipSuccess) return nullptr;
      if (hipMemset(q, 0, SLOTS * 64) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
      r = (int*)q;
      __atomic_store_n(&ring[dev], r, __ATOMIC_RELEASE);
    }
  }
  return r + 16 * (next[dev].fetch_add(1, std::memory_order_relaxed) % SLOTS);
}

std::atomic<int> g_reserved_cus{[] { const char* e = getenv("CST_GEMM_RESERVE_CUS"); return e ? atoi(e) : 0; }()};

template <bool AK, bool BKM, int MODE>
int launch8p(GemmParams p, int64_t nbatch, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8p_kernel<AK, BKM, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  p.tiles_m = (int)cst_ceil_div(p.M, BM);
  p.tiles_n = (int)cst_ceil_div(p.N, BN);
  p.nz = (int)(nbatch * p.splits);
  const int64_t total = (int64_t)p.tiles_m * p.tiles_n * p.nz;
  {
    const int64_t ntl = (int64_t)p.tiles_m * p.tiles_n;
    p.per_group = p.group_m * p.tiles_n;
    if (total * ntl >= (1ll << 40) || total * (int64_t)p.per_group >= (1ll << 40)) { cst_set_error("cst_gemm (8-phase): too many work items"); return CST_ERR_BAD_ARG; }
    p.magic_ntiles = ((1ull << 40) / (unsigned long long)ntl) + 1ull;
    p.magic_per_group = ((1ull << 40) / (unsigned long long)p.per_group) + 1ull;
    p.group_shift = -1;
    for (int sft = 0; sft < 16; ++sft)
      if ((1 << sft) == p.group_m) p.group_shift = sft;
    // round-robin deal of the m-tile groups over the XCDs (setup()): launches with per-batch row limits whose batches have at
    // least two full groups per XCD; CST_GEMM8P_NO_GPERM=1 keeps the contiguous runs (A/B runs)
    static const bool no_gperm = getenv("CST_GEMM8P_NO_GPERM") != nullptr;
    const int gfull = p.tiles_m / p.group_m;
    p.gperm_full = (!no_gperm && p.m_len && gfull >= 16) ? gfull : 0;
    p.mrot = (!no_gperm && p.m_len && p.gperm_full == 0 && ntl >= 2) ? (int)(ntl / 8 > 0 ? ntl / 8 : 1) : 0;
    p.gperm_q = gfull / 8;
    p.gperm_r = gfull % 8;
    p.magic_gq = p.gperm_q > 0 ? ((1ull << 40) / (unsigned long long)p.gperm_q) + 1ull : 0ull;
    p.magic_gq1 = ((1ull << 40) / (unsigned long long)(p.gperm_q + 1)) + 1ull;
  }
  static const int ncu = [] {
    int dev = 0, n = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    if (const char* e = getenv("CST_GEMM8P_CUS")) n = atoi(e);  // experiments: persistent workgroups on a part of the chip only
    return n > 0 ? n : 256;
  }();
  // CUs left to a concurrent stream (the gradient all-reduce of a data-parallel backward: cst_gemm_reserve_cus / CST_GEMM_RESERVE_CUS):
  // a persistent grid one workgroup per CU otherwise holds the whole chip for the length of a launch.  Multiples of 8 keep the
  // per-XCD work claims on.
  const int avail = ncu - g_reserved_cus.load(std::memory_order_relaxed) > 8 ? ncu - g_reserved_cus.load(std::memory_order_relaxed) : 8;
  // the last, partly filled round as half-height items (see the header): unbatched launches whose tail round is at most half full
  const bool no_halves = getenv("CST_GEMM8P_NO_HALVES") != nullptr;  // (read per launch: the tests toggle it in-process)
  int64_t nitems = total;
  p.half_from = 0x7fffffff;
  if (!no_halves && p.nz == 1 && total > avail) {
    const int64_t r = total % avail;
    if (r > 0 && 2 * r <= avail) {  // (the halves must fit ONE round: 137 tiles as 274 halves on 256 CUs run two rounds — 33 500 x 768 x 3072: 0.146 -> 0.193 ms)
      p.half_from = (int)(total - r);
      nitems = total + r;
    }
  }
  p.nitems = (int)nitems;
  dim3 grid((unsigned)(nitems < avail ? nitems : avail), 1, 1);
  static const bool static_walk = getenv("CST_GEMM8P_STATIC") != nullptr;
  // (not under stream capture: a captured launch would pin one slot of the ring for every replay of the graph)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
  p.sched = (!static_walk && !capturing && nitems > avail && grid.x % 8 == 0) ? sched_slot() : nullptr;
  hipLaunchKernelGGL((gemm8p_kernel<AK, BKM, MODE>), grid, dim3(NTHREADS), LDS_BYTES + 1024 + 64, s, p);
  return cst_check_launch("cst_gemm (8-phase)");
}

}  // namespace

extern "C" int cst_gemm_reserve_cus(int n) {
  const int old = g_reserved_cus.load();
  if (n >= 0) g_reserved_cus.store(n > 248 ? 248 : n);
  return old;
}

// The per-lane DMA offsets are 32-bit byte offsets from the tile's first element and must stay below the descriptors' 2 GiB.
bool cst_gemm8p_supported(const cstg::GemmParams& p, bool ak, bool bk, int64_t nbatch) {
  (void)nbatch;
  if (p.a_seg || p.b_seg) return false;
  const int64_t lim = (int64_t)1 << 31;
  const int64_t ea = ak ? (256 * p.lda + p.K) * 2 : (p.K + 64) * p.lda * 2;
  const int64_t eb = bk ? (256 * p.ldb + p.K) * 2 : (p.K + 64) * p.ldb * 2;
  return ea < lim && eb < lim && p.lda > 0 && p.ldb > 0;
}

int cst_gemm8p_launch(cstg::GemmParams p, bool ak, bool bk, int64_t nbatch, hipStream_t s) {
  // the one-extra-operand epilogue instantiation (see EXF): same conditions as the kernel's bf16 image path + exactly one operand
  const bool fast = !p.c_f32 && p.splits == 1 && p.vec_epi && (p.N % 8) == 0 && p.bias_mode != CST_BIAS_ROW &&
                    (p.bias_mode == CST_BIAS_NONE || p.alpha == 1.0f);
  const bool exf = fast && !p.aux_out && ((p.dact != 0) != (p.resid != nullptr)) && (!p.dact || p.aux_in);
  const bool plainf = fast && !p.dact && !p.resid;
#define CST_8P_MODE(MODE)                                                                                          \
  (ak ? (bk ? launch8p<true, true, MODE>(p, nbatch, s) : launch8p<true, false, MODE>(p, nbatch, s))              \
      : (bk ? launch8p<false, true, MODE>(p, nbatch, s) : launch8p<false, false, MODE>(p, nbatch, s)))
  if (exf) return CST_8P_MODE(2);
  if (plainf) return CST_8P_MODE(1);
  return CST_8P_MODE(0);
#undef CST_8P_MODE
}
