// PROBE (not built into libcst_hip.so): the "two co-resident 256 x 128 workgroups per CU" GEMM that DESIGN.md (round 1, §9.1) proposed
// for overlapping one workgroup's epilogue with the other's K loop.  Built, verified (tools/bench_gemm4w.py: every epilogue mode
// bit-compatible with gemm8p within bf16 rounding) and MEASURED on MI355X in round 2 — and rejected:
//   fc1 47 968 x 3072 x 768 plain      gemm8p 0.260 ms | 4 waves/WG 0.302 ms | 8 waves/WG (this file) 0.336 ms
//   fc1 + bias + GELU + aux_out        gemm8p 0.313 ms | 4 waves/WG 0.342 ms | 8 waves/WG 0.381 ms
//   fc2 47 968 x 768 x 3072            gemm8p 0.281 ms | 4 waves/WG 0.327 ms | 8 waves/WG 0.349 ms
//   operands L2-resident (tools/bench_gemm_l2res.py), steady state: gemm8p 1024-1072 TF/s, this kernel 848-898 TF/s
// Why: a 256 x 128 x 32 step moves 1.5x the LDS-DMA instructions per MFMA of a 256 x 256 x 64 tile; a buffer_load ... lds costs its
// wave 100-185 issue cycles, so the K loop of the smaller tile is DMA-issue-bound (45 % matrix-pipe busy with both workgroups
// running, against 52-79 % for gemm8p) and the epilogue overlap has no idle matrix pipe to fill.  The anti-phase start made no
// difference (0.302 vs 0.298 ms).  Kept as a probe so the measurement can be repeated.
//
// gemm4w.hip — the bf16 GEMM of libcst_hip for problems whose EPILOGUE weighs as much as their K loop (the short-K Linear layers of
// the Transformer stacks: K = 768 .. 3072 with bias / GELU / pre-activation output / dropout / act' / residual epilogues).
//
// Why a second large-tile kernel.  gemm8p (256 x 256 tile, 8 waves, ONE workgroup per CU) keeps the matrix pipe at 79 % inside its K
// loop, but a work item's epilogue — 128-256 KB of stores per tile — is bound by the ~8 B/cycle a CU can issue towards HBM: 12-44 k
// cycles with the matrix pipe idle against a 31 k-cycle K loop at K = 768.  Nothing inside one workgroup can hide that (no LDS left to
// park a second accumulator image, no registers for a second accumulator set).  Here the tile is 256 x 128 with FOUR waves (one per
// SIMD) and 74 KiB of LDS, so that TWO workgroups are co-resident on every CU: while one of them drains its epilogue through the
// store queue, the other one owns the matrix pipe.  The second workgroup of a CU starts half an item late (one s_sleep burst per
// launch); both then alternate K loop / epilogue in anti-phase for the rest of the launch.
//
// Tile 256 (M) x 128 (N) x 32 (K step); wave (wm, wn) owns rows wm*128 + [0,128) x cols wn*64 + [0,64): 4 x 2 tiles of 32 x 32,
// v_mfma_f32_32x32x16_bf16 with swapped operands (D = B-frag x A-frag: a lane holds 4 consecutive output columns).
// LDS ring of 3 stages, a stage = A image 16 KiB + B image 8.5 KiB, filled by buffer_load ... lds (16 B per lane, zeros out of range):
//   k-major image  [rows][64 B]: 16-byte chunk c of row r is stored at chunk c ^ ((r >> 2) & 3)  -> ds_read_b128 conflict-free
//   mn-major image (B only: the dX GEMMs): 8 DMA groups of 4 k-rows x 256 B, 1088 B apart; k-row 16a + 4b + c in group 4a + c, slot b
//                  -> ds_read_b64_tr_b16 with immediate-only addressing (the layout of gemm8p, two k16 blocks instead of four)
// K loop, one s_barrier per K step, fragments prefetched half a step ahead in registers:
//     { ds_read F1(t) ; 8 MFMA (k16 block 0 of step t) ; lgkmcnt(0) ; vmcnt(6): stage t+1 landed ; s_barrier ;
//       DMA stage t+3 -> slot t % 3 ; ds_read F0(t+1) ; 8 MFMA (k16 block 1) ; lgkmcnt(0) }
//   RAW: a stage is read only after the counted vmcnt of every wave + the barrier.  WAR: slot t % 3 is re-filled after the barrier
//   that follows the last read of stage t (every wave's lgkmcnt(0) precedes it).  Two stages stay in flight across every barrier.
// Epilogue (same contract as gemm8p's fast path: bf16 C, no split-K, vector-aligned operands): accumulators (+bias, x alpha) -> one
//   bf16 rounding -> LDS image [256][128] (ring region, row stride 272 B) -> row-contiguous 16-byte read-back -> pre-activation
//   output, activation, dropout, act'(aux) multiply, residual -> 16-byte non-temporal stores.
// Persistent: grid = 2 x #CU workgroups, static stride over (batch, tile) items in the XCD-aware grouped tile order of gemm8p.
#include "gemm_common.h"
#include <cstdlib>
#include <type_traits>

namespace {
using namespace cstg;
using T = bf16_t;

constexpr int BM = 256, BN = 128, BK = 32, NTHREADS = 512, NSTAGE = 3;
constexpr int A_BYTES = BM * BK * 2;      // 16384
constexpr int GSTRIDE = 1088;             // mn-major B image: byte distance of the DMA groups
constexpr int B_BYTES = 8 * GSTRIDE;      // 8704 (the k-major B image uses the first 8192)
constexpr int STAGE = A_BYTES + B_BYTES;  // 25088
constexpr int RING = NSTAGE * STAGE;      // 75264
constexpr int ERS = 272;                  // epilogue image row stride (256 B of data + 16 B pad)
constexpr int LDS_BYTES = RING + 1024 + 64;  // + the bias row's 1-KiB DMA footprint
static_assert(BM * ERS <= RING, "the epilogue image lives in the ring region");
constexpr unsigned OOB = 0x80000000u;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// lgkmcnt(0) through the builtin: unlike the asm form it is visible to hipcc's wait-count insertion, which otherwise believes the
// fragment reads of the previous half step are still outstanding and waits for the just-issued prefetch reads instead
__device__ __forceinline__ void wait_lgkm0_known() { __builtin_amdgcn_s_waitcnt(0xC07F); }

__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ void lds_write_b64(unsigned addr, uint2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_write_b128(unsigned addr, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ u32x4 lds_read_b128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, 0, 0, 0);
}

template <bool BKM>
__global__ __launch_bounds__(NTHREADS, 4) void gemm4w_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // The by-value argument block is re-read from the kernarg segment (scalar loads through an opaque pointer) at each use site
  // outside the K loop, so that the ~70 SGPRs of launch parameters are not kept live (and spilled) across the MFMA loop.
  typedef const __attribute__((address_space(4))) GemmParams* kparg_t;
  kparg_t kp = (kparg_t)__builtin_amdgcn_kernarg_segment_ptr();
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;  // 4 x 2 waves, wave tile 64 x 64
  const int lrow = lane & 31, hi = lane >> 5;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int total = ntiles * p.nz;
  const int Kdim = (int)p.K;
  const int ksteps = (Kdim + BK - 1) / BK;
  char* const bias_lds = smem + RING;
  const bool bias_in_acc = p.bias_mode == CST_BIAS_COL;

  // ---- lane constants of the DMA source maps ----
  // k-major image: instruction q covers rows 16q .. 16q+15; lane -> row 16q + lane/4, stored chunk lane%4 = source chunk ^ ((row>>2)&3)
  const int kc = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;     // source k (elements) inside a K step
  const int kr = lane >> 2;                                 // row inside the instruction's 16-row group
  // mn-major B image: instruction j of wave w fills group G = w + 4j: k-row 16j + 4 (lane/16) + w, 16-byte chunk lane%16 of its 128 columns
  const int mk = 16 * (wave >> 2) + 4 * (lane >> 4) + (wave & 3), mc = (lane & 15) * 8;  // wave w fills group G = w

  // ---- fragment read offsets ----
  const int swz = (lrow >> 2) & 3;
  const int a_rd = (wm * 64 + lrow) * 64;                                   // + i * 2048 + (((2kk + hi) ^ swz) << 4)
  const int b_rd = A_BYTES + (wn * 64 + lrow) * 64;                        // + j * 2048 + ...
  const int trl = A_BYTES + ((lane & 15) >> 2) * GSTRIDE + hi * 512 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2 + wn * 128;

  __amdgpu_buffer_rsrc_t ra, rb;
  unsigned va0 = 0, vb0 = 0, lda2 = 0, ldb2 = 0;
  int nt = 0, mrem = 0, nrem = 0;
  int64_t m0 = 0, n0 = 0, cofs = 0, bofs = 0;

  auto setup = [&](int v) {
    asm volatile("" : "+s"(kp));
    struct { int64_t M, N, K, lda, ldb, batch1, sa0, sa1, sb0, sb1, sc0, sc1, sbias0, sbias1; const void *A, *B; int tiles_m, tiles_n, group_m;
             const uint32_t* m_live; uint32_t m_epoch; } p;
    p.M = kp->M; p.N = kp->N; p.K = kp->K; p.lda = kp->lda; p.ldb = kp->ldb; p.batch1 = kp->batch1;
    p.sa0 = kp->sa0; p.sa1 = kp->sa1; p.sb0 = kp->sb0; p.sb1 = kp->sb1; p.sc0 = kp->sc0; p.sc1 = kp->sc1;
    p.sbias0 = kp->sbias0; p.sbias1 = kp->sbias1; p.A = kp->A; p.B = kp->B;
    p.tiles_m = kp->tiles_m; p.tiles_n = kp->tiles_n; p.group_m = kp->group_m; p.m_live = kp->m_live; p.m_epoch = kp->m_epoch;
    const int z = v / ntiles;
    int id = v - z * ntiles;
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
    m0 = (int64_t)tm * BM;
    n0 = (int64_t)tn * BN;
    const int64_t b0 = z / p.batch1, b1 = z % p.batch1;
    const T* A = (const T*)p.A + b0 * p.sa0 + b1 * p.sa1;
    const T* B = (const T*)p.B + b0 * p.sb0 + b1 * p.sb1;
    cofs = b0 * p.sc0 + b1 * p.sc1;
    bofs = b0 * p.sbias0 + b1 * p.sbias1;
    nt = ksteps;
    if (p.m_live) {  // all four 64-row blocks of this tile's A rows stamped dead: no K loop, the epilogue runs on zero accumulators
      typedef const __attribute__((address_space(4))) uint32_t* cptr_t;
      cptr_t ml = (cptr_t)p.m_live;
      const int t0 = (int)(m0 >> 6), tn64 = (int)((p.M + 63) >> 6);
      bool live = false;
#pragma unroll
      for (int i = 0; i < 4; ++i) live |= (t0 + i < tn64) && ml[t0 + i < tn64 ? t0 + i : t0] == p.m_epoch;
      if (!live) nt = 0;
    }
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + m0 * p.lda), (short)0, (int)OOB, 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc((void*)(BKM ? B + n0 * p.ldb : B + n0), (short)0, (int)OOB, 0x00020000);
    lda2 = (unsigned)(p.lda * 2);
    ldb2 = (unsigned)(p.ldb * 2);
    mrem = (int)(p.M - m0 < BM ? p.M - m0 : BM);
    nrem = (int)(p.N - n0 < BN ? p.N - n0 : BN);
    va0 = (unsigned)kr * lda2 + (unsigned)kc * 2;
    vb0 = BKM ? (unsigned)kr * ldb2 + (unsigned)kc * 2 : (unsigned)mk * ldb2 + (unsigned)mc * 2;
  };

  // stage K step `t` of the current item into ring slot `slot`: 3 DMA instructions per wave (two of A, one of B)
  auto stage = [&](int t, int slot) {
    char* const base = smem + slot * STAGE;
    const int k0 = t * BK;
    const bool kok = k0 + kc < Kdim;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = wave * 2 + j;
      const bool ok = kok && (q * 16 + kr < mrem);
      dma16(ra, base + q * 1024, ok ? va0 + (unsigned)(q * 16) * lda2 + (unsigned)(k0 * 2) : OOB);
    }
    if (BKM) {
      const bool ok = kok && (wave * 16 + kr < nrem);
      dma16(rb, base + A_BYTES + wave * 1024, ok ? vb0 + (unsigned)(wave * 16) * ldb2 + (unsigned)(k0 * 2) : OOB);
    } else {
      const bool ok = (k0 + mk < Kdim) && (mc < nrem);
      dma16(rb, base + A_BYTES + wave * GSTRIDE, ok ? vb0 + (unsigned)k0 * ldb2 : OOB);
    }
  };

  auto read_tr = [&](Frag<T>& f, const char* q) {
    const v4s_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(q));
    const v4s_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(q + 256));
    u16x8 t;
    t[0] = (unsigned short)x[0]; t[1] = (unsigned short)x[1]; t[2] = (unsigned short)x[2]; t[3] = (unsigned short)x[3];
    t[4] = (unsigned short)y[0]; t[5] = (unsigned short)y[1]; t[6] = (unsigned short)y[2]; t[7] = (unsigned short)y[3];
    f.v = __builtin_bit_cast(bf16x8, t);
  };
  // fragments of k16 block kk of the stage in ring slot `slot`
  auto read_frags = [&](Frag<T> (&fa)[2], Frag<T> (&fb)[2], int slot, int kk) {
    const char* img = smem + slot * STAGE;
    const int ch = ((2 * kk + hi) ^ swz) << 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (BKM) fb[j].v = *reinterpret_cast<const bf16x8*>(img + b_rd + j * 2048 + ch);
      else read_tr(fb[j], img + trl + j * 64 + kk * 4 * GSTRIDE);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[i].v = *reinterpret_cast<const bf16x8*>(img + a_rd + i * 2048 + ch);
  };

  f32x16 acc[2][2];
  Frag<T> fa0[2], fb0[2], fa1[2], fb1[2];
  auto mfma_block = [&](Frag<T> (&fa)[2], Frag<T> (&fb)[2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) mma16(acc[i][j], fb[j], fa[i]);
    __builtin_amdgcn_s_setprio(0);
  };

  int v = blockIdx.x;
  if (v >= total) return;
  // anti-phase start of the CU's second workgroup: half an item (K loop + epilogue are about equally long on the shapes routed here)
  if (p.sched && (int)blockIdx.x >= (int)gridDim.x / 2) {
    const int units = ksteps * 6;  // x 64 cycles: ~ three quarters of a solo K loop
    for (int i = 0; i < units; i += 127) __builtin_amdgcn_s_sleep(127);
  }
  const unsigned ebase_off = lds_off(smem);

  while (true) {
    setup(v);
    if (bias_in_acc) {  // bias[n0 .. n0+128) -> LDS (columns >= N read 0)
      const int64_t left = kp->N - n0;
      const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)kp->bias + bofs + n0), (short)0,
                                                                              (int)((left < BN ? left : BN) * 2), 0x00020000);
      if (wave == 0) dma16(rbias, bias_lds, (unsigned)lane * 16);
    }
    // ---- prologue: stages 0, 1, 2 in flight ----
    if (0 < nt) stage(0, 0);
    if (1 < nt) stage(1, 1);
    if (2 < nt) stage(2, 2);
    if (nt > 2) wait_vm<6>();
    else if (nt > 1) wait_vm<3>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (bias_in_acc) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint2 raw = *reinterpret_cast<const uint2*>(bias_lds + (wn * 64 + j * 32 + 8 * g + 4 * hi) * 2);
          const float b0 = __uint_as_float(raw.x << 16), b1 = __uint_as_float(raw.x & 0xffff0000u);
          const float b2 = __uint_as_float(raw.y << 16), b3 = __uint_as_float(raw.y & 0xffff0000u);
          const float s = 1.0f / kp->alpha;  // the epilogue multiplies by alpha: bias enters unscaled (alpha == 1 on every call site with a bias)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            acc[i][j][4 * g] = b0 * s; acc[i][j][4 * g + 1] = b1 * s; acc[i][j][4 * g + 2] = b2 * s; acc[i][j][4 * g + 3] = b3 * s;
          }
        }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    }
    if (nt > 0) read_frags(fa0, fb0, 0, 0);
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);

    // ---- K loop ----
    int slot = 0;  // ring slot of stage t
    for (int t = 0; t < nt; ++t) {
      const int nslot = slot == NSTAGE - 1 ? 0 : slot + 1;
      read_frags(fa1, fb1, slot, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm0_known();
      if (t + 2 < nt) wait_vm<3>();   // stage t+1 landed (stage t+2 may still be in flight)
      else wait_vm<0>();
      __builtin_amdgcn_s_barrier();   // every wave: stage t+1 visible, stage t fully read
      if (t + 3 < nt) stage(t + 3, slot);
      if (t + 1 < nt) read_frags(fa0, fb0, nslot, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm0_known();
      slot = nslot;
    }
    // every DMA of this item has landed and was consumed (vmcnt(0) + barrier in the last step, or in the prologue when nt == 0)

    // ---- epilogue ----
    {
      const int64_t e_m0 = m0, e_n0 = n0, e_cofs = cofs;
      asm volatile("" : "+s"(kp));
      GemmParams p;  // epilogue parameters, re-read after the K loop (only the fields used below are materialised)
      p.M = kp->M; p.N = kp->N; p.C = kp->C; p.ldc = kp->ldc; p.act = kp->act; p.aux_out = kp->aux_out; p.ld_aux_out = kp->ld_aux_out;
      p.dact = kp->dact; p.aux_in = kp->aux_in; p.ld_aux_in = kp->ld_aux_in; p.resid = kp->resid; p.ld_resid = kp->ld_resid; p.alpha = kp->alpha;
      p.drop_thr = kp->drop_thr; p.drop_key = kp->drop_key; p.drop_scale = kp->drop_scale; p.drop_row0 = kp->drop_row0;
      const T* exsrc = p.dact ? (const T*)p.aux_in : (const T*)p.resid;
      const int64_t exld = p.dact ? p.ld_aux_in : p.ld_resid;
      const bool both = p.dact && p.resid;
      const float alpha = p.alpha;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
            bf16x4_t pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = static_cast<__bf16>(acc[i][j][4 * g + e] * alpha);
            lds_write_b64(ebase_off + (wm * 64 + i * 32 + lrow) * ERS + (wn * 64 + j * 32 + 8 * g + 4 * hi) * 2, __builtin_bit_cast(uint2, pk));
          }
      wait_lgkm<0>();
      __builtin_amdgcn_s_barrier();
      if (exsrc) {  // loop A: operand loads never share a loop with stores (loads and stores retire out of order with each other)
#pragma unroll 2
        for (int it = 0; it < 8; ++it) {
          const int vi = tid + NTHREADS * it;
          const int rl = vi >> 4, cl = (vi & 15) * 8;
          const int64_t row = e_m0 + rl, col = e_n0 + cl;
          if (row >= p.M || col >= p.N) continue;
          const u32x4 exv = *reinterpret_cast<const u32x4*>(exsrc + e_cofs + row * exld + col);
          const u32x4 zraw = lds_read_b128(ebase_off + rl * ERS + cl * 2);
          wait_lgkm<0>();
          __builtin_amdgcn_sched_barrier(0);
          if (p.aux_out) *reinterpret_cast<u32x4*>((T*)p.aux_out + e_cofs + row * p.ld_aux_out + col) = zraw;  // (not on the hot path)
          float x[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x[2 * e] = __uint_as_float(zraw[e] << 16);
            x[2 * e + 1] = __uint_as_float(zraw[e] & 0xffff0000u);
          }
          if (p.act == CST_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.0f);
          } else if (p.act == CST_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = gelu_t<T>(x[e]);
          }
          if (p.drop_thr) cst_drop8(x, p.drop_key, (uint64_t)((row + p.drop_row0) * p.N + col), p.drop_thr, p.drop_scale);
          if (p.dact) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              x[2 * e] *= dact_t<T>(__uint_as_float(exv[e] << 16), p.dact);
              x[2 * e + 1] *= dact_t<T>(__uint_as_float(exv[e] & 0xffff0000u), p.dact);
            }
            if (both) {
              float rr[8];
              load8((const T*)p.resid + e_cofs + row * p.ld_resid + col, rr);
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
      const bool post = !exsrc && (p.act != CST_ACT_NONE || p.drop_thr);
#pragma unroll 2
      for (int it = 0; it < 8; ++it) {
        const int vi = tid + NTHREADS * it;
        const int rl = vi >> 4, cl = (vi & 15) * 8;
        const int64_t row = e_m0 + rl, col = e_n0 + cl;
        if (row >= p.M || col >= p.N) continue;
        const u32x4 zraw = lds_read_b128(ebase_off + rl * ERS + cl * 2);
        wait_lgkm<0>();
        __builtin_amdgcn_sched_barrier(0);
        T* cdst = (T*)p.C + e_cofs + row * p.ldc + col;
        if (!exsrc && p.aux_out) __builtin_nontemporal_store(zraw, reinterpret_cast<u32x4*>((T*)p.aux_out + e_cofs + row * p.ld_aux_out + col));
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
        if (p.act == CST_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.0f);
        } else if (p.act == CST_ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = gelu_t<T>(x[e]);
        }
        if (p.drop_thr) cst_drop8(x, p.drop_key, (uint64_t)((row + p.drop_row0) * p.N + col), p.drop_thr, p.drop_scale);
        bf16x8 ob;
#pragma unroll
        for (int e = 0; e < 8; ++e) ob[e] = static_cast<__bf16>(x[e]);
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, ob), reinterpret_cast<u32x4*>(cdst));
      }
    }
    v += (int)gridDim.x;
    if (v >= total) break;
    __builtin_amdgcn_s_barrier();  // the image is fully read back: the ring may receive the next item
  }
}

template <bool BKM>
int launch4w(GemmParams p, int64_t nbatch, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel<BKM>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  p.tiles_m = (int)cst_ceil_div(p.M, BM);
  p.tiles_n = (int)cst_ceil_div(p.N, BN);
  p.nz = (int)nbatch;
  const int64_t total = (int64_t)p.tiles_m * p.tiles_n * p.nz;
  static const int ncu = [] {
    int dev = 0, n = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const int64_t slots = 2 * (int64_t)ncu;
  dim3 grid((unsigned)(total < slots ? total : slots), 1, 1);
  static const bool no_phase = getenv("CST_GEMM4W_NO_PHASE") != nullptr;
  // `sched` doubles as the anti-phase switch of this kernel (non-null = delay the second half of the grid by half an item); only
  // worth it when every slot has more than one item
  p.sched = (!no_phase && total >= 2 * slots) ? reinterpret_cast<int*>(1) : nullptr;
  hipLaunchKernelGGL((gemm4w_kernel<BKM>), grid, dim3(NTHREADS), LDS_BYTES, s, p);
  return cst_check_launch("cst_gemm (4-wave co-resident)");
}

}  // namespace

// bf16, A k-major, fast-epilogue contract (bf16 C, no split-K, 16-byte aligned operands, column bias or none); 32-bit DMA offsets
bool cst_gemm4w_supported(const cstg::GemmParams& p, bool ak, bool bk, int64_t nbatch) {
  (void)nbatch;
  if (!ak || p.a_seg || p.b_seg || p.c_f32 || p.splits != 1 || !p.vec_epi || (p.N % 8) != 0 || (p.K % 8) != 0) return false;
  if (p.bias_mode == CST_BIAS_ROW || (p.bias_mode == CST_BIAS_COL && p.alpha != 1.0f)) return false;
  if (p.k_len || p.k_live) return false;
  const int64_t lim = (int64_t)1 << 31;
  const int64_t ea = (256 * p.lda + p.K) * 2;
  const int64_t eb = bk ? (128 * p.ldb + p.K) * 2 : (p.K + 32) * p.ldb * 2;
  return ea < lim && eb < lim && p.lda > 0 && p.ldb > 0 && (p.lda % 8) == 0 && (p.ldb % 8) == 0;
}

int cst_gemm4w_launch(cstg::GemmParams p, bool ak, bool bk, int64_t nbatch, hipStream_t s) {
  (void)ak;
  return bk ? launch4w<true>(p, nbatch, s) : launch4w<false>(p, nbatch, s);
}
