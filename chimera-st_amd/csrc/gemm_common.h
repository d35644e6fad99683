// gemm_common.h — launch parameters and the fused epilogue shared by every GEMM kernel of libcst_hip (gemm.hip, gemm8p.hip).
#pragma once
#include "cst_common.h"

namespace cstg {

// The tallest output tile any GEMM kernel of this library uses, and the granularity tiles start at inside one batch of a batched
// launch.  conv0.hip sizes the frames it writes for the layer-1 conv GEMM from this (cst_conv_row_limits): every row a live tile of
// that GEMM reads must have been written.  Each kernel static_asserts its BM against it.
constexpr int GEMM_MAX_BM = 256;

struct GemmParams {
  int64_t M, N, K;
  const void* A; int64_t lda, a_seg, a_seg_stride;
  const void* B; int64_t ldb, b_seg, b_seg_stride;
  void* C; int64_t ldc;
  const void* bias; int bias_mode; int64_t sbias0, sbias1;
  int act;
  void* aux_out; int64_t ld_aux_out;
  int dact;
  const void* aux_in; int64_t ld_aux_in;
  const void* resid; int64_t ld_resid;
  float alpha;
  uint32_t drop_thr, drop_key; float drop_scale;  // drop_thr == 0: no dropout; element index = (row + drop_row0) * N + col
  int64_t drop_row0;                                // row offset of this launch inside the logical output (row-split launches)
  int64_t batch1;
  int64_t sa0, sa1, sb0, sb1, sc0, sc1;
  int splits;
  float* ws;  // split-K partials [batch][split][M][N]
  int c_f32;
  int vec_epi;  // all epilogue operands 16-byte aligned -> staged, vectorised epilogue
  int tiles_m, tiles_n;
  int nz;  // persistent kernels: number of (batch, split) slices
  // gemm8p: work items [0, half_from) are whole 256 x 256 tiles; item half_from + 2 j + h is the 128-row half h of tile half_from + j
  // (the tiles of a launch's last, partly filled round of the persistent grid: twice as many items of half the height); nitems in all
  int half_from, nitems;
  int group_m;      // gemm8p: m tiles per group of the tile walk (the patch of tiles an XCD works on concurrently)
  // gemm8p: the item -> tile map runs once per work item on the CU's one scalar unit for all eight waves; its divisions by launch
  // constants are multiplications by these (q = (n * magic) >> 40, exact for n * d < 2^40)
  unsigned long long magic_ntiles, magic_per_group;
  int per_group, group_shift;  // group_m * tiles_n; log2(group_m) or -1 if group_m is not a power of two
  // gemm8p, launches with per-batch row limits (m_len: the conv stack): the m-tile GROUPS of a batch are dealt to the XCDs round-robin
  // instead of in contiguous runs (see the kernel's setup()): gperm_full = number of full groups taking part (0 = off),
  // gperm_q / gperm_r = gperm_full / 8 and % 8, with the division magics for gperm_q and gperm_q + 1
  int gperm_full, gperm_q, gperm_r;
  int mrot;  // ... and batches with fewer groups than that: batch z walks its tiles rotated by z * mrot positions (0 = off)
  unsigned long long magic_gq, magic_gq1;
  int split_order;  // gemm_kernel, split-K unbatched: XCD-contiguous (split, tile) item order (see the kernel)
  int* sched;       // gemm8p: 16 zeroed ints of scheduling state (8 per-XCD item counters, 1 exit counter), nullptr = static walk
  const uint32_t* m_live; uint32_t m_epoch;  // rows of A in blocks of 64: an output tile with no live block skips its K loop
  const int32_t* k_len;  // gemm_kernel, per batch0: K rows >= k_len[b0] of A are zero
  const int32_t* m_len;  // per batch0: output tiles with m0 >= m_len[b0] skip their K loop (rows of A all zero, values of C unused)
  const uint32_t* k_live; uint32_t k_epoch;  // gemm_kernel: K blocks of 64 whose stamp != k_epoch are all-zero in A and skipped
  void* colsum;      // gemm_kernel, mn-major A: [batch][M] column sums of A over k (storage dtype), written by the n-tile-0 workgroups
  float* colsum_ws;  // splits > 1: fp32 partials [batch][split][M] behind the slabs; added by the reduce kernel in split order
};

__device__ __forceinline__ int64_t segaddr(int64_t c, int64_t seg, int64_t seg_stride) {
  return seg ? (c / seg) * seg_stride + (c % seg) : c;
}

template <typename T>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, int64_t cofs, int64_t bofs, int64_t row, int64_t col, float v) {
  v *= p.alpha;
  if (p.bias_mode == CST_BIAS_COL) v += DT<T>::ld((const T*)p.bias + bofs + col);
  else if (p.bias_mode == CST_BIAS_ROW) v += DT<T>::ld((const T*)p.bias + bofs + row);
  if (p.aux_out) DT<T>::st((T*)p.aux_out + cofs + row * p.ld_aux_out + col, v);
  v = act_t<T>(v, p.act);
  if (p.drop_thr) v *= cst_drop1(p.drop_key, (uint64_t)((row + p.drop_row0) * p.N + col), p.drop_thr, p.drop_scale);
  if (p.dact) v *= dact_t<T>(DT<T>::ld((const T*)p.aux_in + cofs + row * p.ld_aux_in + col), p.dact);
  if (p.resid) v += DT<T>::ld((const T*)p.resid + cofs + row * p.ld_resid + col);
  if (p.c_f32) ((float*)p.C)[cofs + row * p.ldc + col] = v;
  else DT<T>::st((T*)p.C + cofs + row * p.ldc + col, v);
}

// 8 consecutive columns of one row: the same epilogue as epilogue_store, with 16-byte global accesses.
template <typename T>
__device__ __forceinline__ void epilogue_store8(const GemmParams& p, int64_t cofs, int64_t bofs, int64_t row, int64_t col, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
  if (p.bias_mode == CST_BIAS_COL) {
    float b[8];
    load8((const T*)p.bias + bofs + col, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += b[e];
  } else if (p.bias_mode == CST_BIAS_ROW) {
    const float b = DT<T>::ld((const T*)p.bias + bofs + row);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += b;
  }
  if (p.aux_out) store8((T*)p.aux_out + cofs + row * p.ld_aux_out + col, v);
  if (p.act == CST_ACT_RELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
  } else if (p.act == CST_ACT_GELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = gelu_t<T>(v[e]);
  }
  if (p.drop_thr) cst_drop8(v, p.drop_key, (uint64_t)((row + p.drop_row0) * p.N + col), p.drop_thr, p.drop_scale);
  if (p.dact) {
    float z[8];
    load8((const T*)p.aux_in + cofs + row * p.ld_aux_in + col, z);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= dact_t<T>(z[e], p.dact);
  }
  if (p.resid) {
    float r[8];
    load8((const T*)p.resid + cofs + row * p.ld_resid + col, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += r[e];
  }
  if (p.c_f32) store8((float*)p.C + cofs + row * p.ldc + col, v);
  else store8((T*)p.C + cofs + row * p.ldc + col, v);
}

}  // namespace cstg

// gemm8p.hip: 256 x 256 x 64 bf16 tile, 8 waves, 8-phase DMA pipeline.  Returns CST_OK after the launch.
int cst_gemm8p_launch(cstg::GemmParams p, bool a_kmajor, bool b_kmajor, int64_t nbatch, hipStream_t s);
bool cst_gemm8p_supported(const cstg::GemmParams& p, bool a_kmajor, bool b_kmajor, int64_t nbatch);
// gemm4w.hip: four waves, 256 x 192 tiles, pinned issue order — the long-K N = 768-family launches (same bits as gemm8p's fast path)
int cst_gemm4w_launch(cstg::GemmParams p, hipStream_t s);
bool cst_gemm4w_supported(const cstg::GemmParams& p, bool a_kmajor, bool b_kmajor, int64_t nbatch);

