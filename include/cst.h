/* cst.h — C ABI of libcst_hip.so: the MI355X (gfx950) hot path of Chimera-ST.
 *
 * The reference (Glaciohound/Chimera-ST, a fairseq fork) has NO C ABI on this path: its seam is
 * nn.Module.forward / torch.autograd.Function calling PyTorch ATen.  Each entry
 * point below therefore replaces an ATen call site of the reference; the file:line it replaces
 * is cited per function (paths relative to the reference root).
 *
 * Conventions (SURVEY.md §8b "Native (C-ABI) layer"):
 *   - plain C: raw DEVICE pointers + explicit int64 sizes/strides (in ELEMENTS) + dtype enum +
 *     a hipStream_t passed as void*; no torch types.
 *   - ownership: every buffer (inputs, outputs, workspaces, saved-for-backward) is allocated and
 *     owned by the caller; the library never frees or retains a pointer past return.
 *   - errors: int return, 0 = ok, negative = cst_status; text via cst_last_error() (thread-local).
 *     Nothing throws across the boundary.
 *   - threading: re-entrant; no global mutable state except the optional profiling table and, per
 *     device, one 64 KiB device allocation made on the first large cst_gemm launch: the ring of
 *     work-item counters of the persistent GEMM (csrc/gemm8p.hip; each launch uses one 64-byte
 *     slot and leaves it zeroed).  It is the only memory the library owns.
 *   - all kernels are asynchronous on the given stream.
 *   - accumulation is always fp32; `dtype` is the storage type of activations/weights.
 */
#ifndef CST_H
#define CST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumps on any change of a signature or a descriptor layout (3: cst_gemm_desc.m_len, cst_attn_desc.seq_offsets, workspaces of the
 * fixed-order reductions; 4: cst_attn_desc.kpm_bits / bwd_ws, the separable attention-dropout mask; 5: cst_gemm_desc.colsum;
 * 6: cst_dec_ln_q_cross_attn).
 * cst_version() returns the value the library was built with; chimera-st_amd/lib.py refuses a mismatch. */
#define CST_ABI_VERSION 6

typedef enum { CST_F32 = 0, CST_BF16 = 1 } cst_dtype;

typedef enum {
  CST_OK = 0,
  CST_ERR_BAD_ARG = -1,     /* shape / stride / alignment / dtype not supported */
  CST_ERR_LAUNCH = -2,      /* hipLaunch / hipGetLastError failure */
  CST_ERR_WORKSPACE = -3,   /* caller workspace too small */
  CST_ERR_UNSUPPORTED = -4
} cst_status;

typedef void* cst_stream; /* hipStream_t */

const char* cst_last_error(void);
int cst_version(void);            /* ABI version, bumps on any signature change */
int cst_device_arch_ok(void);     /* 1 if the current device is gfx950, 0 otherwise */

/* ------------------------------------------------------------------------------------------
 * Profiling table (bench.py roofline leg): when enabled every launch below is bracketed with
 * hipEvents on its own stream and its algorithmic flops/bytes are recorded per kernel class.
 * ------------------------------------------------------------------------------------------ */
typedef enum {
  CST_K_GEMM = 0, CST_K_ATTN_FWD, CST_K_ATTN_BWD, CST_K_LAYERNORM, CST_K_CONV0, CST_K_ELEMENTWISE,
  CST_K_LOSS, CST_K_OPTIM, CST_K_NUM
} cst_kernel_class;
void cst_prof_enable(int on);     /* clears the table when switching on */
/* synchronises the recorded events; returns launches; outputs total ms / flops / bytes */
int64_t cst_prof_query(int kernel_class, double* total_ms, double* flops, double* bytes);
/* per-launch records of a class as text, one line per launch in launch order: "<ms> <flops> <bytes> <tag>" (tag = the launcher's
 * description: kernel family and shape).  Returns the bytes needed including the terminating 0 (call with cap = 0 to size buf). */
int64_t cst_prof_dump(int kernel_class, char* buf, int64_t cap);

/* ------------------------------------------------------------------------------------------
 * LayerNorm (eps, affine) — replaces torch.nn.LayerNorm at fairseq/modules/layer_norm.py:30-35
 * (apex FusedLayerNorm seam, :11-28) and Fp32LayerNorm (:38-50).
 *   s = x (+ res);  y = (s - mean) * rstd * gamma + beta
 * x,res,y,sum_out: [rows, cols] row-major contiguous; gamma/beta [cols]; mean/rstd fp32 [rows].
 * res and sum_out may be NULL.  cols % 8 == 0, cols <= 2048.
 * ------------------------------------------------------------------------------------------ */
int cst_layernorm_fwd(const void* x, const void* res, const void* gamma, const void* beta,
                      void* y, void* sum_out, float* mean, float* rstd,
                      int64_t rows, int64_t cols, float eps, int dtype, cst_stream stream);
/* dx = LN backward w.r.t. s (s = x + res is what `s` points to); dgamma/dbeta fp32 [cols]
 * (overwritten) in `grad_dtype` (CST_F32 or `dtype`: accumulated in fp32, rounded once).  dres (optional extra upstream gradient on s, e.g. the residual branch) is
 * added into dx when non-NULL.  workspace: cst_layernorm_bwd_workspace() bytes.  dgamma = dbeta = NULL (ABI 5): the second stage is
 * left to cst_reduce_multi — the row-block partials stay in `workspace` as fp32 [blocks][2][cols] (dgamma row, dbeta row per block),
 * blocks = cst_layernorm_bwd_workspace(rows, cols) / (8 * cols). */
int64_t cst_layernorm_bwd_workspace(int64_t rows, int64_t cols);
int cst_layernorm_bwd(const void* dy, const void* s, const void* gamma, const float* mean,
                      const float* rstd, const void* dres, void* dx, void* dgamma, void* dbeta,
                      void* workspace, int64_t rows, int64_t cols, int dtype, int grad_dtype, cst_stream stream);
/* the same, and additionally stamps tile_live[row / 64] = epoch for every row whose dx is not exactly zero (tile_live: uint32
 * [ceil(rows / 64)], NOT initialised by the caller: a tile is live iff its stamp equals this call's unique, non-zero epoch) —
 * the k_live / k_epoch operand of the weight-gradient GEMMs that consume dx (or a row-wise function of it) */
int cst_layernorm_bwd_tiles(const void* dy, const void* s, const void* gamma, const float* mean,
                            const float* rstd, const void* dres, void* dx, void* dgamma, void* dbeta,
                            void* workspace, int64_t rows, int64_t cols, int dtype, int grad_dtype,
                            uint32_t* tile_live, uint32_t epoch, cst_stream stream);

/* ------------------------------------------------------------------------------------------
 * GEMM on MFMA tiles — replaces F.linear / nn.Conv1d call sites:
 *   q/k/v/out projections  modules/multihead_attention.py:205-224,370
 *   fc1/fc2                modules/transformer_layer.py:144-148,404-408; models/wav2vec/wav2vec2.py:949-953
 *   conv layers 1..6       models/wav2vec/wav2vec2.py:707 (implicit GEMM on channels-last rows)
 *   pos_conv               models/wav2vec/wav2vec2.py:773-779 (grouped; segmented-K implicit GEMM)
 *   subsampler convs       models/speech_to_text/s2t_transformer.py:53-74
 *   post_extract_proj      models/wav2vec/wav2vec2.py:550-551
 *   vocab projection       models/transformer.py:830-836
 * and their autograd backward GEMMs.
 *
 *   C[M,N] = epilogue( alpha * sum_k Aop[m,k] * Bop[k,n] )
 * Operand storage ("k-major" = the reduction index is the contiguous one):
 *   a_kmajor=1: Aop[m,k] at A[m*lda + segaddr_a(k)]      a_kmajor=0: Aop[m,k] at A[k*lda + segaddr_a(m)]
 *   b_kmajor=1: Bop[k,n] at B[n*ldb + segaddr_b(k)]      b_kmajor=0: Bop[k,n] at B[k*ldb + segaddr_b(n)]
 *   segaddr(c) = (c / seg) * seg_stride + c % seg   (seg = 0 means plain c).  seg % 8 == 0.
 * lda/ldb may be SMALLER than the contiguous extent (overlapping rows = implicit-GEMM conv1d).
 * Epilogue, in order:  v = alpha*acc;  v += bias;  [aux_out = v];  v = act(v);
 *                      [v *= act'(aux_in)];  v += resid;  C = v
 * Batching: two batch levels (e.g. utterance x group); offsets in elements.
 * split_k > 1: partial sums go through `workspace` (cst_gemm_workspace bytes) and are reduced
 * into C by a second kernel (bias/act/resid are applied there).
 * ------------------------------------------------------------------------------------------ */
typedef enum { CST_ACT_NONE = 0, CST_ACT_RELU = 1, CST_ACT_GELU = 2 } cst_act;
typedef enum { CST_BIAS_NONE = 0, CST_BIAS_COL = 1, CST_BIAS_ROW = 2 } cst_bias_mode;

typedef struct {
  int dtype;              /* storage dtype of A, B, bias, resid, aux (cst_dtype) */
  int c_dtype;            /* storage dtype of C (cst_dtype; fp32 allowed with bf16 inputs) */
  int a_kmajor, b_kmajor;
  int64_t M, N, K;
  const void* A; int64_t lda, a_seg, a_seg_stride;
  const void* B; int64_t ldb, b_seg, b_seg_stride;
  void* C; int64_t ldc;
  const void* bias; int bias_mode;   /* dtype `dtype`; [N] or [M] */
  int64_t sbias0, sbias1;            /* batch strides of bias (0 = shared; grouped conv: per group) */
  int act;                           /* cst_act applied after bias */
  void* aux_out; int64_t ld_aux_out; /* optional pre-activation copy, dtype `dtype` */
  int dact;                          /* cst_act whose derivative (at aux_in) multiplies v */
  const void* aux_in; int64_t ld_aux_in;
  const void* resid; int64_t ld_resid;
  float alpha;
  float drop_p; uint32_t drop_key;   /* drop_p > 0: v *= keep(drop_key, row*N + col) / (1 - drop_p) after `act`, before dact /
                                        resid (FairseqDropout fused: residual + dropout(linear(x)), dropout(act(fc1 x)) and the
                                        matching mask in the dX-through-activation GEMM); needs batch0*batch1 == 1 */
  int64_t batch0, batch1;            /* >= 1 */
  int64_t sa0, sa1, sb0, sb1, sc0, sc1; /* batch strides (elements) for A, B, C(+aux/resid) */
  int split_k;                       /* 0/1 = none; >1 explicit; -1 = let the library choose */
  void* workspace; int64_t workspace_bytes;
  const uint32_t* k_live; uint32_t k_epoch; /* optional (NULL = off): one stamp per 64 consecutive k indices; a k block whose stamp
                                        != k_epoch is declared ALL-ZERO in A (every row/column of A at those k) and may be skipped —
                                        exact, 0 * b adds nothing.  Used for the weight-gradient GEMMs (k = token), whose dY rows
                                        at padded frames are exactly zero; stamps come from cst_layernorm_bwd_tiles.  Kernels
                                        that do not implement skipping ignore it. */
  const uint32_t* m_live; uint32_t m_epoch; /* optional: the same stamps on the M side — one per 64 consecutive rows of A (= rows
                                        of C); an output tile whose rows of A are all dead skips its K loop (its accumulators
                                        are exactly 0; the epilogue still runs: act', residual, stores).  Used for the dX GEMMs. */
  const int32_t* k_len;              /* optional, [batch0]: rows k >= k_len[b0] of batch b0's A are all zero (a trailing run — the
                                        frames past an utterance's end in the conv weight-gradient GEMMs); the K loop of that batch
                                        stops there and split-K divides the live range.  Exact. */
  const int32_t* m_len;              /* optional, [batch0]: rows m >= m_len[b0] of batch b0's A are all zero AND nobody reads those rows
                                        of C for their value (the frames past an utterance's end in the conv feature extractor: the
                                        reference zeroes them behind the CNN, wav2vec2.py:820-821, and their gradient is exactly zero).
                                        Output tiles that lie entirely behind m_len[b0] skip their K loop and run the epilogue on
                                        zero accumulators (C = epilogue(0): 0 for the bias-free GELU / GELU' epilogues of that stack). */
  void* colsum;                      /* optional (ABI 5), mn-major A only: colsum[batch][m] = sum_k A[k][m] in `dtype`, a by-product of staging
                                        A (the tiles of output column 0 add up the A vectors they load anyway; with split-K the slices'
                                        partial sums ride behind the slabs in the workspace and the reduce kernel adds them in split
                                        order).  With A = dY of a Linear this is the bias gradient (torch's `grad_output.sum(0)` behind
                                        F.linear, modules/multihead_attention.py / transformer_layer.py call sites): the weight-gradient
                                        GEMM dW = dY^T X reads every dY element exactly once per output tile column, so no separate
                                        column-sum pass over dY is needed.  Fixed summation order (bit-reproducible). */
  int defer_reduce;                  /* split-K launches only (ABI 5): 1 = do not launch the reduce; the fp32 slabs [split][M][N] (and the
                                        colsum slices [split][M] behind them) stay in `workspace`, which the caller keeps alive and
                                        hands to cst_reduce_multi later.  Needs a plain epilogue (alpha 1, no bias / activation /
                                        operand) and a caller-owned workspace; batched launches (slabs [batch][split][M][N], no colsum)
                                        let the caller sum over the batch in the same pass.  Ignored when the launch does
                                        not split K (cst_gemm_splits tells). */
} cst_gemm_desc;

int64_t cst_gemm_workspace(const cst_gemm_desc* d);
/* 1 if cst_gemm would produce d->colsum (non-NULL) as a by-product of this launch's own kernel, 0 if it would run the separate
 * two-launch column sum behind it (the 16-wave / DMA configurations): a caller that can get the sums cheaper elsewhere — e.g. from
 * the dropout pass that has to read dY anyway, cst_dropout_colsum — asks first. */
int cst_gemm_colsum_is_fused(const cst_gemm_desc* d);
/* The number of K slices cst_gemm will use for this descriptor (1 = no split, no workspace slabs). */
int cst_gemm_splits(const cst_gemm_desc* d);
/* Persistent GEMM launches use (CUs - n) workgroups from now on (n < 0: query only); returns the previous value.  The data-parallel
 * reducer sets it while bucket all-reduces are in flight under the backward pass, so that the RCCL kernels on the side stream find free
 * CUs instead of waiting for a launch boundary (legacy_distributed_data_parallel.py has no overlap to protect).  Initial value:
 * environment variable CST_GEMM_RESERVE_CUS (default 0). */
int cst_gemm_reserve_cus(int n);
int cst_gemm(const cst_gemm_desc* d, cst_stream stream);

/* ------------------------------------------------------------------------------------------
 * Fused attention (flash-style, scores never reach HBM) — replaces
 * F.multi_head_attention_forward (modules/multihead_attention.py:155-187) and the in-tree
 * bmm/softmax/bmm branch (:326-361):  O = softmax(scale * Q K^T + masks) V, softmax in fp32.
 * Modes covered by the descriptor: padded self-attention (key_padding_mask), memory attention
 * (Tq = M slots, Tk = encoder frames: the column mask of w2v2_transformer_interlingua.py:284-288
 * is "keys = first Tk columns"), causal decoder self-attention, cross-attention, and
 * single-query incremental decode (Tq = 1, Tk = cache length).
 * Q,K,V,O element (b,h,t,d) at ptr[b*sb + h*sh + t*st + d]; D in {32, 64}; d contiguous.
 * key_padding_mask: uint8 [B, Tk] (1 = pad -> -inf), row stride kpm_stride; may be NULL.
 * lse: fp32 [B,H,Tq] log-sum-exp of the scaled masked scores (saved for backward).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int dtype;
  int64_t B, H, Tq, Tk, D;
  const void* Q; int64_t q_sb, q_sh, q_st;
  const void* K; int64_t k_sb, k_sh, k_st;
  const void* V; int64_t v_sb, v_sh, v_st;
  void* O; int64_t o_sb, o_sh, o_st;
  float* lse;
  const uint8_t* key_padding_mask; int64_t kpm_stride;
  int causal;
  float scale;
  float drop_p; uint32_t drop_key;   /* attention dropout (multihead_attention.py:359): P * keep(drop_key, row (b*H+h)*Tq + q, key k) / (1-p),
                                        keep = the separable mask of csrc/cst_common.h (cst_adrop_*; numpy twin rng.keep_mask_attn_numpy);
                                        the backward call must carry the same two values */
  /* backward only */
  const void* dO; int64_t do_sb, do_sh, do_st;
  void* dQ; int64_t dq_sb, dq_sh, dq_st;
  void* dK; int64_t dk_sb, dk_sh, dk_st;
  void* dV; int64_t dv_sb, dv_sh, dv_st;
  float* delta;  /* fp32 [B,H,Tq] workspace: rowsum(dO * O) */
  /* optional int32 [B] (NULL = none): the caller's promise that every key >= kv_len[b] is padding in key_padding_mask
   * (trailing padding of a length-sorted batch).  Those key tiles are skipped in all three kernels — they contribute
   * exact zeros, so results are bit-identical to the unskipped run; key_padding_mask still governs keys < kv_len[b]. */
  const int32_t* kv_len;
  /* optional uint8 [B, H, ceil(Tq/64)] workspace of cst_attn_bwd (NULL = off): the delta pre-pass records which 64-query tiles
   * have a non-zero dO row and the dQ / dK-dV kernels stop at the last such tile — the tiles beyond it contribute exact zeros
   * (dP = dO V^T = 0 and delta = 0 give dS = 0), so the gradients are bit-identical to the full walk. */
  uint8_t* q_flags;
  /* optional int32 [B+1] (NULL = dense [B, T] batches): PACKED self-attention over the row sets of cst_rows_pack.  Q / K / V / O and
   * all gradients are [rows, H*D] (the *_sb batch strides are ignored; Q, K and V share one row stride); sequence b owns rows
   * [seq_offsets[b], seq_offsets[b+1]) as QUERIES and its first kv_len[b] rows as KEYS (kv_len NULL: all of them; the remaining
   * rows are the padding frames kept for their outputs: they attend, nobody attends to them, their dK / dV are zero).  Tq = Tk =
   * the longest sequence: it sizes the grid and strides lse / delta / q_flags [B, H, Tq] and the dropout index space, so that a
   * packed call draws exactly the dropout masks of the dense call it replaces.  key_padding_mask must be NULL. */
  const int32_t* seq_offsets;
  /* optional uint64 [B, ceil(Tk/64)] (NULL = none): key_padding_mask packed one word per 64-key tile (bit k of word j: key 64 j + k is
   * masked; bits of keys >= Tk set).  With it (or without any key_padding_mask) bf16 / D = 64 / non-causal problems run the DMA-staged
   * kernels (csrc/attention_fast.inc), which read tile masks from scalar registers; without it a masked problem runs the generic
   * kernels.  Must describe exactly key_padding_mask. */
  const uint64_t* kpm_bits;
  /* optional workspace of cst_attn_bwd, cst_attn_bwd_workspace() bytes (NULL = generic backward kernels): the dQ kernel writes
   * -lse * log2(e) and -delta / dropout-scale per query, padded to whole 64-query tiles, and (round 5) every query row's 32-byte
   * dropout signature behind them, for the DMA-staged dK/dV kernel. */
  void* bwd_ws;
} cst_attn_desc;

int64_t cst_attn_bwd_workspace(const cst_attn_desc* d);
int cst_attn_fwd(const cst_attn_desc* d, cst_stream stream);
int cst_attn_bwd(const cst_attn_desc* d, cst_stream stream);

/* ------------------------------------------------------------------------------------------
 * wav2vec2 conv layer 0 (Cin = 1) fused with GroupNorm(C groups) and GELU — replaces
 * Conv1d + Fp32GroupNorm + GELU at models/wav2vec/wav2vec2.py:697-753 (layer 0) and
 * modules/fp32_group_norm.py:13-25.  The pre-norm conv output is never written to HBM
 * (two-phase: statistics, then normalise+GELU+store), backward recomputes it from the wave.
 *   wav [B,S] fp32 (the raw samples as the collater hands them over; never down-cast);
 *   w [C,k], gamma, beta [C] in `dtype`;  y [B,L,C] channels-last in `dtype`;
 *   mean,rstd fp32 [B,C];  gram fp32 [B, k*k + k] (per-utterance lag moments, saved for backward);
 *   L = (S-k)/stride + 1;  k <= 16;  C % 8 == 0.
 * ------------------------------------------------------------------------------------------ */
/* Live-frame limits of the conv feature extractor (wav2vec2.py:685-763) for one batch, all layers in ONE launch: which frames of every
 * layer's output anything reads, from the count of real frames behind the LAST layer (frames past an utterance's end are zeroed
 * behind the stack, wav2vec2.py:820-821, and carry exactly zero gradient).  They feed cst_gemm_desc.m_len / k_len of the conv GEMMs.
 *   nz_last int32 [B]: real frames per utterance behind layer L-1;  k, stride: int32 [L] kernel widths / strides;  S: samples.
 *   out int32 [L][1 + smax][B], smax = max stride of layers 1..L-1:
 *     out[i][0][b]     = nz_i[b]: layer i's frames t >= nz_i[b] are unread; nz_{i-1} = (nz_i - 1) stride_i + k_i (0 if nz_i = 0),
 *                        each clamped to the layer's frame count
 *     out[i][1 + r][b] = max(0, ceil((nz_{i-1}[b] - r) / stride_i)) for i >= 1, r < stride_i: the live rows of residue class r of
 *                        layer i's INPUT gradient (the windowed dX GEMMs of functional.conv1d_cl)
 *     out[0][1][b]     = frames of layer 0 that layer 1's live 256-row output tiles read: min(len_0, ceil(nz_1 / 256) 256 stride_1 +
 *                        k_1) — what cst_conv0_gn_gelu_fwd has to write (L >= 2)
 * L <= 8, strides <= 8. */
int cst_conv_row_limits(const int32_t* nz_last, const int32_t* k, const int32_t* stride, int L, int64_t S, int32_t* out, int64_t B,
                        int smax, cst_stream stream);
/* workspace: cst_conv0_fwd_workspace() bytes — per-block partial moments, added up in a fixed order (no atomics: the statistics,
 * and with them the whole forward pass, are bit-reproducible run to run). */
int64_t cst_conv0_fwd_workspace(int64_t B, int64_t S, int k, int stride);
/* frame_limit (optional, int32 [B]): forward — frames t >= frame_limit[b] of utterance b are read by nobody (cst_conv_row_limits
 * out[0][1]: behind the rows the next layer's live 256-row tiles reach) and are neither computed nor written (y keeps whatever the
 * buffer held there; the GroupNorm statistics still cover all L frames, as the reference's do over the zero-padded audio);
 * backward — dy is exactly zero from frame frame_limit[b] on (cst_conv_row_limits out[0][0]) and those frames are not read.
 * NULL = every frame. */
int cst_conv0_gn_gelu_fwd(const float* wav, const void* w, const void* gamma, const void* beta,
                          void* y, float* mean, float* rstd, float* gram, float* workspace, const int32_t* frame_limit, int64_t B,
                          int64_t S, int64_t C, int k, int stride, float eps, int dtype, cst_stream stream);
/* dw [C,k], dgamma [C], dbeta [C] are fp32 and overwritten.
 * workspace: cst_conv0_bwd_workspace() bytes — per-block partial sums [B][blocks][k+2][C] + their fixed-order reduction
 * (no atomics: bit-reproducible gradients). */
int64_t cst_conv0_bwd_workspace(int64_t B, int64_t S, int64_t C, int k, int stride);
int cst_conv0_gn_gelu_bwd(const void* dy, const float* wav, const void* w, const void* gamma,
                          const void* beta, const float* mean, const float* rstd, const float* gram,
                          float* dw, float* dgamma, float* dbeta, float* workspace, const int32_t* frame_limit,
                          int64_t B, int64_t S, int64_t C, int k, int stride, int dtype,
                          cst_stream stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise / reduction helpers (HBM-bound)
 * ------------------------------------------------------------------------------------------ */
/* GLU over the channel pairs (c, c+C) of z [rows, 2C] -> y [rows, C]  (F.glu, s2t_transformer.py:74) */
int cst_glu_fwd(const void* z, void* y, int64_t rows, int64_t C, int dtype, cst_stream stream);
int cst_glu_bwd(const void* dy, const void* z, void* dz, int64_t rows, int64_t C, int dtype, cst_stream stream);
/* dx = dy * act'(z)  (GELU: modules/gelu.py:25; ReLU) */
int cst_act_bwd(const void* dy, const void* z, void* dx, int64_t n, int act, int dtype, cst_stream stream);
/* y = act(x) */
int cst_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, cst_stream stream);
/* out[c] = sum_r x[r, c]  (bias gradients); out fp32 [cols], overwritten */
int cst_colsum(const void* x, int64_t ldx, float* out, int64_t rows, int64_t cols, int dtype, cst_stream stream);
/* Gradient of `residual + dropout(linear(x))` on its way into the Linear's backward, in ONE pass: xd = x * keep(key, index) / (1 - p)
 * (FairseqDropout backward, modules/fairseq_dropout.py:20-37: the mask of the forward call, element index = r * cols + c) stored in
 * `dtype`, and out[c] = sum_r xd[r][c] (the bias gradient) over the STORED values — the same bits as cst_dropout followed by
 * cst_colsum_typed.  x, xd contiguous [rows, cols]; workspace: cst_colsum_workspace(rows, cols) bytes; row_live / epoch as in
 * cst_colsum_typed_live (optional): all-zero 64-row tiles are written as zeros without being read. */
int cst_dropout_colsum(const void* x, void* xd, void* out, void* workspace, int64_t rows, int64_t cols, int dtype, int out_dtype,
                       float p, uint32_t key, const uint32_t* row_live, uint32_t epoch, cst_stream stream);
/* the same sums without atomics: row-chunk partials in `workspace` (cst_colsum_workspace bytes), reduced in a fixed order and
 * written in `out_dtype` (the bias parameter's dtype) — deterministic, no zero-fill and no conversion launch around it.
 * out == NULL (here, in cst_colsum_typed_live and in cst_dropout_colsum): the second stage is left to the caller — the partials stay in
 * `workspace` as fp32 [cst_colsum_workspace / (4 cols)][cols], to be finished by cst_reduce_multi with order 1 (same summation order,
 * same bits), together with the other deferred reductions of a backward pass. */
int64_t cst_colsum_workspace(int64_t rows, int64_t cols);
int cst_colsum_typed(const void* x, int64_t ldx, void* out, void* workspace, int64_t rows, int64_t cols, int dtype, int out_dtype,
                     cst_stream stream);
/* the same with live-tile stamps of x's rows (cst_layernorm_bwd_tiles): 64-row tiles whose stamp != epoch are all zero and skipped */
int cst_colsum_typed_live(const void* x, int64_t ldx, void* out, void* workspace, int64_t rows, int64_t cols, int dtype,
                          int out_dtype, const uint32_t* row_live, uint32_t epoch, cst_stream stream);
/* gradient of a strided conv1d input from the column-gradient rows of the implicit GEMM:
 * dx[b, l, c] = sum_{t,j : t*stride + j = l} dcol[b, t, j*C + c];  optionally multiplied by
 * act'(z[b,l,c]) (GELU of the previous conv layer).  dcol [B, Lout, k*C]; dx/z [B, Lin, C]. */
int cst_col2im1d(const void* dcol, const void* z, void* dx, int64_t B, int64_t Lin, int64_t Lout,
                 int64_t C, int k, int stride, int pad, int dact, int dtype, cst_stream stream);
/* Weight normalisation along the LAST dimension — nn.utils.weight_norm(conv, dim=2) of the wav2vec2 positional convolution
 * (models/wav2vec/wav2vec2.py:773-779): v, w, dw, dv [R, C] contiguous (R = C_out * C_in / groups, C = kernel width), g, dg [C],
 * norm fp32 [C] (saved by the forward call).  w = v * g / ||v[:, c]||;  dv = (g/n) (dw - v dot / n^2), dg = dot / n, dot = sum_r dw v.
 * Fixed summation order.  C % 8 == 0, C <= 256, (C / 8) divides 256.  workspace: cst_weight_norm_workspace(R, C) bytes. */
int64_t cst_weight_norm_workspace(int64_t R, int64_t C);
int cst_weight_norm_fwd(const void* v, const void* g, void* w, float* norm, void* workspace, int64_t R, int64_t C, int dtype, cst_stream stream);
int cst_weight_norm_bwd(const void* v, const void* g, const void* dw, const float* norm, void* dv, void* dg, void* workspace, int64_t R,
                        int64_t C, int dtype, cst_stream stream);
/* dst [C, R] = src [R, C] transposed (contiguous both; R, C multiples of the 16-byte vector).  The Linear dX GEMMs take the weight this
 * way (dY W with W^T k-major: both operands read as the forward GEMM reads them). */
int cst_transpose2d(const void* src, void* dst, int64_t R, int64_t C, int dtype, cst_stream stream);
/* The same for a whole table of matrices in one launch (every Linear weight of a model once per update, after the optimizer step):
 * items_dev = DEVICE array of n entries ordered by tile0 = the number of 64 x 64 tiles of the entries before it (entry 0: 0);
 * total_tiles = their sum over the table.  One dtype per call. */
typedef struct cst_transpose_item {
  const void* src; void* dst;
  int64_t R, C;      /* src [R, C] -> dst [C, R] */
  int64_t tile0;
} cst_transpose_item;
int cst_transpose2d_multi(const cst_transpose_item* items_dev, int n, int64_t total_tiles, int dtype, cst_stream stream);
/* The second stage of many fixed-order reductions in ONE launch (ABI 5).  The reference has no counterpart: its weight / bias /
 * LayerNorm gradients come out of single ATen calls (F.linear / F.layer_norm backward at modules/transformer_layer.py:105-155,
 * multihead_attention.py:326-361); here the weight-gradient GEMMs of the small layers split K, the LayerNorm backward reduces over
 * row blocks, and each used to finish with a launch of its own.  cst_gemm (desc.defer_reduce) and cst_layernorm_bwd (dgamma = dbeta
 * = NULL) leave their fp32 partials in the caller's workspace; this call finishes up to CST_REDUCE_MAX_ITEMS of them:
 *   dst[i] = sum_{p < P} src[p * stride + i],  i < L      (L % 8 == 0, src / dst 16-byte aligned, stride % 4 == 0)
 * in the summation order of the launch-each kernel the item stands in for — order 0: p = 0, 1, 2, ... (cst_gemm's split-K reduce
 * and its colsum slices), order 1: 64 interleaved chains p = g, g + 64, ... added 0..63 (cst_layernorm_bwd's second stage) — so a
 * gradient has the same bits whichever route finished it; dst in dst_dtype (fp32 or bf16).  `items` is a HOST array (it travels in
 * the kernel arguments: no host-to-device copy, no synchronisation); block0 is filled in by the library. */
#define CST_REDUCE_MAX_ITEMS 64
typedef struct cst_reduce_item {
  const float* src; void* dst;
  int64_t stride, L;
  int32_t P, dst_dtype, block0, order;
} cst_reduce_item;
int cst_reduce_multi(const cst_reduce_item* items, int n, cst_stream stream);
/* y[r,:] = mask[r] ? 0 : x[r,:]   (x[padding_mask] = 0, wav2vec2.py:820-821) */
int cst_mask_rows(const void* x, const uint8_t* mask, void* y, int64_t rows, int64_t cols, int dtype, cst_stream stream);

/* Packed (padding-free) row sets.  A length-sorted, right-padded batch [B, T, C] of a row-wise network (LayerNorm, Linear, FFN,
 * self-attention with a key-padding mask: wav2vec2.py:818-845, 937-957) carries, per utterance, `len` real frames followed by
 * padding frames whose values the reference also computes.  Past the reach of the positional convolution (len + conv_pos/2) all
 * padding frames of an utterance enter the layer stack with the SAME vector and keep identical values through every row-wise
 * layer, so the stack may run on  n_b = min(T, len_b + conv_pos/2 + 1)  rows per utterance — the last one standing for all the
 * identical rows behind it — and reproduce the padded result bit for bit (dropout off; with dropout on the padding rows share a
 * mask).  seq_off int32 [B+1]: sequence b owns packed rows [seq_off[b], seq_off[b+1]).
 *   cst_rows_pack   : dst[seq_off[b] + t] = src[b, t]; with tail_sum the last packed row of a sequence receives the SUM of
 *                     src[b, n_b - 1 .. T - 1] in index order (the gradient of cst_rows_unpack with tail_broadcast; deterministic).
 *   cst_rows_unpack : dst[b, t] = src[seq_off[b] + min(t, n_b - 1)] with tail_broadcast, zeros beyond n_b otherwise (the
 *                     gradient of cst_rows_pack without tail_sum).
 * C % 8 == 0. */
int cst_rows_pack(const void* src, const int32_t* seq_off, void* dst, int64_t B, int64_t T, int64_t C, int tail_sum, int dtype,
                  cst_stream stream);
int cst_rows_unpack(const void* src, const int32_t* seq_off, void* dst, int64_t B, int64_t T, int64_t C, int tail_broadcast, int dtype,
                    cst_stream stream);

/* y = x * keep / (1 - p): FairseqDropout (modules/fairseq_dropout.py:20-37) forward; the same call on dy is its backward.
 * The mask is counter-based — keep(key, element index), csrc/cst_common.h — so nothing is stored between the two calls and
 * fused epilogues (cst_gemm / cst_attn_*) regenerate the identical mask from the same key.  n elements, contiguous. */
int cst_dropout(const void* x, void* y, int64_t n, float p, uint32_t key, int dtype, cst_stream stream);
/* y = alpha * x * keep / (1 - p)   (n % 8 == 0): the gradient of `dropout(alpha * x + const)` — the dense-feature branch of
 * cst_embed_pos_fwd below. */
int cst_dropout_scale(const void* x, void* y, int64_t n, float alpha, float p, uint32_t key, int dtype, cst_stream stream);

/* ------------------------------------------------------------------------------------------
 * Token embedding + sinusoidal positions of the TRAINING path — replaces, in one kernel,
 *   embed_scale * nn.Embedding(tokens)            models/transformer.py:744-760 (decoder), w2v2_transformer_interlingua.py:216, :231
 *   + SinusoidalPositionalEmbedding.forward       modules/sinusoidal_positional_embedding.py:60-105: weights.index_select(make_positions(input))
 *     with utils.make_positions                   utils.py:235-245: pad_idx + cumsum(non-pad) for non-pad symbols, pad_idx for pads
 *   + FairseqDropout                              modules/fairseq_dropout.py:20-37 (counter-based mask, see cst_dropout)
 *   out[b,t,:] = dropout(scale * src[b,t,:] + pos_table[position(b,t),:])
 * src: exactly one of `embed` [V, C] (row tokens[b,t]; ids outside [0, V) read the pad row) or `x` [B, T, C] (dense features: the
 *   audio branch of S2T_W2V2_TransformerEncoder, w2v2_transformer.py:352-358).
 * position source: `pad_mask` uint8 [B, T] when given (the encoders pass their padding mask: non-pad <=> mask != pad_idx, as
 *   `tensor.ne(padding_idx)` reads a 0/1 mask), else `tokens` (non-pad <=> tokens != pad_idx).
 * pos_table: fp32 [pos_rows, C], the reference's sin || cos table with row pad_idx zero; pos_rows >= pad_idx + 1 + T; NULL = no
 *   positional term.  C % 8 == 0.  tokens int64 [B, T]. */
int cst_embed_pos_fwd(const int64_t* tokens, const uint8_t* pad_mask, const void* embed, const void* x, const float* pos_table,
                      float scale, int64_t pad_idx, void* out, int64_t B, int64_t T, int64_t C, int64_t V, int64_t pos_rows,
                      float drop_p, uint32_t drop_key, int dtype, cst_stream stream);
/* Gradient of the embedding table of cst_embed_pos_fwd — replaces ATen's embedding_dense_backward (sort by key + segmented
 * reduce, 5 kernels; F.embedding with padding_idx: the pad row receives no gradient):
 *   dE[v,:] = scale * sum over occurrences (b,t) of v != pad_idx, in (b,t) order, of keep(b,t,:) * dy[b,t,:];  all other rows 0.
 * Deterministic (fixed summation order, no atomics).  dy [n, C] (n = B*T), tokens int64 [n], dE [V, C] in grad_dtype (fp32 or
 * dtype; fully overwritten).  C % 8 == 0, C <= 8192. */
int cst_embed_bwd(const void* dy, const int64_t* tokens, void* dE, float scale, int64_t pad_idx, int64_t n, int64_t C, int64_t V,
                  float drop_p, uint32_t drop_key, int dtype, int grad_dtype, cst_stream stream);

/* ------------------------------------------------------------------------------------------
 * Label-smoothed cross entropy over vocabulary logits — replaces fp32 log_softmax
 * (models/fairseq_decoder.py:75-79 -> utils.py:469-473) + label_smoothed_nll_loss
 * (criterions/label_smoothed_cross_entropy.py:13-30), reduce=True, ignore_index=pad.
 *   logits [rows, V] (dtype), target int64 [rows];  out2 fp32[2] = {loss_sum, nll_sum} (overwritten);
 *   lse fp32 [rows] saved for backward;  row_ws fp32 [2 * rows]: the per-row terms, added up in a fixed order by a second
 *   one-workgroup kernel (no atomics: the loss is bit-reproducible run to run).
 * bwd: dlogits = gscale * d loss / d logits  (gscale = upstream grad of the summed loss).
 * ------------------------------------------------------------------------------------------ */
int cst_ls_ce_fwd(const void* logits, const int64_t* target, float* out2, float* lse, float* row_ws,
                  int64_t rows, int64_t V, float eps, int64_t pad_idx, int dtype, cst_stream stream);
int cst_ls_ce_bwd(const void* logits, const int64_t* target, const float* lse, const float* gscale,
                  void* dlogits, int64_t rows, int64_t V, float eps, int64_t pad_idx, int dtype,
                  cst_stream stream);

/* ------------------------------------------------------------------------------------------
 * Contrastive term of TripletSTMTContrastiveCriterion.compute_contrastive
 * (criterions/triplet_st_mt_contrastive.py:154-169): a = audio memory, t = text memory, both [B, M, C] batch-major
 * (M <= 64 slots); c[b,i,j] = cosine(a[b,i], t[b,j]) (fp32, eps 1e-8), logits = c / temp with the AUDIO slot as class
 * dim and target(j) = j;  loss[0] = sum_b sum_j ( logsumexp_i - logits[j][j] )  (overwritten; loss_rows fp32 [B] holds the
 * per-utterance terms, added up in a fixed order).
 * sim fp32 [B,M,M], na/nt fp32 [B,M] (norms) are saved for backward;  bwd: da, dt = gscale[0] * d loss / d a, t.
 * ------------------------------------------------------------------------------------------ */
int cst_contrastive_fwd(const void* a, const void* t, float* loss, float* loss_rows, float* sim, float* na, float* nt,
                        int64_t B, int64_t M, int64_t C, float temp, int dtype, cst_stream stream);
int cst_contrastive_bwd(const void* a, const void* t, const float* sim, const float* na, const float* nt,
                        const float* gscale, void* da, void* dt, int64_t B, int64_t M, int64_t C, float temp,
                        int dtype, cst_stream stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer path — replaces FP16Optimizer's flat-copy / unscale / clip / Adam / copy-back chain
 * (optim/fp16_optimizer.py:16-300; utils.py:323-364; optim/adam.py:146-226).
 * ------------------------------------------------------------------------------------------ */
/* out[0] += sum(x^2).  Deterministic (fixed-order two-stage reduction, no atomics): data-parallel replicas must get
 * bit-identical gradient norms.  workspace: cst_sumsq_workspace() bytes. */
int64_t cst_sumsq_workspace(void);
int cst_sumsq(const void* x, int64_t n, float* out, float* workspace, int dtype, cst_stream stream);
/* One fused pass over flat buffers: g = grad * (*grad_scale); Adam (fairseq semantics: bias
 * correction folded into the step size, decoupled weight decay); master fp32 -> model dtype.
 * grad_scale is a DEVICE fp32 scalar (e.g. clip coefficient / sample_size) so no host sync. */
int cst_adam_step(float* master, float* exp_avg, float* exp_avg_sq, const void* grad, void* model_param,
                  int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int64_t step, const float* grad_scale, int grad_dtype, int param_dtype,
                  cst_stream stream);

/* ------------------------------------------------------------------------------------------
 * Device-resident incremental decoding — replaces the per-step host loop of
 * fairseq/sequence_generator.py:_generate (:286-541), search.py BeamSearch.step (:109-144), finalize_hypos
 * (:575-696), the incremental self-attention branch of modules/multihead_attention.py:189-293 and
 * reorder_incremental_state (:419-437).  Every kernel reads the step counter from DEVICE memory, so one decode step is
 * a fixed launch sequence (capturable in a hipGraph); the host polls *num_remaining instead of synchronising per step.
 *
 * State (all caller-owned device buffers; bbsz = bsz*beam, L1 = max_len+1, LT = max_len+2):
 *   step int32[1]; tokens int64[2][bbsz][LT], scores f32[2][bbsz][L1], anc int32[2][bbsz][L1]: ping-pong halves, step s
 *   reads half (s & 1) and writes half ((s+1) & 1).  anc[h][j] = cache row holding position j of hypothesis h (the K/V
 *   caches are append-only: row h writes slot [h][s]; a beam reorder moves ancestry entries, not cache contents).
 *   cands_to_ignore u8[bsz][beam], finished u8[bsz], nfinal int32[bsz], num_remaining int32[1];
 *   finalized hypotheses in emission order: fin_tokens int64[bsz][beam][L1], fin_pos f32[bsz][beam][L1] (positional
 *   scores), fin_score f32[bsz][beam] (length-normalised when normalize_scores), fin_len int32[bsz][beam].
 * cst_beam_step: fp32 log-softmax of logits [bbsz, vocab] (/temperature), NaN/pad/unk/min-len/max-len masks, + cumulative
 *   score, top-(2*beam) per sentence, eos finalisation, next active hypotheses, then *step += 1.  beam <= 20.
 *   logits: 16-byte aligned, ld_logits a multiple of 16 bytes and >= vocab rounded up to it (rows are read as vectors).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int dtype;                         /* storage type of logits */
  int64_t bsz, beam, vocab, max_len;
  int64_t pad, unk, eos, min_len;
  float unk_penalty, len_penalty, temperature;
  int normalize_scores;
  const void* logits; int64_t ld_logits;
  int32_t* step;
  int64_t* tokens; float* scores; int32_t* anc;
  uint8_t* cands_to_ignore; uint8_t* finished; int32_t* nfinal; int32_t* num_remaining;
  int64_t* fin_tokens; float* fin_pos; float* fin_score; int32_t* fin_len;
  void* workspace;                   /* cst_beam_workspace(bsz, beam) bytes, 16-byte aligned: per-row candidates + step ticket */
} cst_beam_desc;
int64_t cst_beam_workspace(int64_t bsz, int64_t beam);
int cst_beam_init(const cst_beam_desc* d, cst_stream stream);
int cst_beam_step(const cst_beam_desc* d, cst_stream stream);
/* out[h] = scale * embed[tokens[s&1][h][s]] + pos_table[pad_idx + 1 + s], s = *step (models/transformer.py:744-760,
 * sinusoidal_positional_embedding.py:88-95).  embed/out in `dtype`, pos_table fp32 [pos_rows, C]. */
int cst_dec_embed(const int64_t* tokens, const int32_t* step, const void* embed, const float* pos_table, float scale,
                  int64_t pad_idx, void* out, int64_t rows, int64_t C, int64_t max_len, int64_t pos_rows, int dtype,
                  cst_stream stream);
/* Linear layer of one decode step over the hypothesis rows: y[M, N] = act(x[M, K] W[N, K]^T + bias[N]) (+ resid[M, N]) — replaces the
 * F.linear calls of the incremental decoder layer (modules/transformer_layer.py:293-420 at one target position) with a kernel shaped
 * for M = bsz * beam rows: every weight byte is requested within one memory round trip (N/16..N/64 x ceil(M/80) small workgroups,
 * operands global -> MFMA fragments, K split over the four waves and added in wave order: deterministic).  bf16 only;
 * K % 512 == 0; ldx % 8 == 0; x, W 16-byte aligned.  bias / resid may be NULL.  step (optional, device): the launch does nothing
 * once *step > max_len (a captured step replayed past the end of the search). */
int cst_dec_linear(const void* x, const void* W, const void* bias, const void* resid, void* y, int64_t M, int64_t N, int64_t K,
                   int64_t ldx, int64_t ld_resid, int64_t ldy, int act, const int32_t* step, int64_t max_len, int dtype,
                   cst_stream stream);
/* LayerNorm + Linear of one decode step in ONE launch: y = act(LN(x; gamma, beta, eps) W^T + b) (+ resid), the pre-norm sub-blocks of
 * the decoder layer (modules/transformer_layer.py:346-349, :369-372, :403-406).  The caller folds the affine part into the weights
 * once (they are constants while decoding): Wg[n,k] = bf16(W[n,k] gamma[k]);  sg[n] = sum_k Wg[n,k];  sb[n] = sum_k W[n,k] beta[k] +
 * b[n] (fp32 vectors).  The kernel multiplies the raw rows by Wg, gathers every row's sum(x) and sum(x^2) from the operand fragments
 * it loads anyway, and applies  rstd_m (acc - mean_m sg[n]) + sb[n]  in the epilogue.  Same limits as cst_dec_linear. */
int cst_dec_ln_linear(const void* x, const void* Wg, const float* sg, const float* sb, float eps, const void* resid, void* y, int64_t M,
                      int64_t N, int64_t K, int64_t ldx, int64_t ld_resid, int64_t ldy, int act, const int32_t* step, int64_t max_len,
                      int dtype, cst_stream stream);
/* Single-query self-attention at position s = *step: qkv [rows, 3*H*D] (q | k | v of the newest token), caches
 * [rows, L1, H*D]; appends k/v at slot s, attends over positions 0..s through anc (half s & 1), out [rows, H*D].
 * `scale` multiplies QK^T in fp32.  D in {32, 64}. */
int cst_dec_self_attn(const void* qkv, void* kcache, void* vcache, const int32_t* anc, const int32_t* step, void* out,
                      int64_t rows, int64_t H, int64_t D, int64_t max_len, float scale, int dtype, cst_stream stream);
/* Cross attention of one decode step (modules/multihead_attention.py:189-293 with static_kv): q, out [bsz*beam, H*D];
 * kx, vx HEAD-MAJOR [bsz, H, S, D] — the encoder keys / values of a SENTENCE, shared by its beam hypotheses (not replicated);
 * key_padding_mask uint8 [bsz, S] or NULL.  No-op when *step > max_len.  One pass, online softmax; any S.
 * bf16 / D = 64 / beam <= 32 (operands 16-byte aligned): the matrix-core kernel of csrc/attention_fast.inc — the beam rows are one
 * 32-row MFMA block, the four waves of a (sentence, head) workgroup split the keys (32-key tiles, own LDS-DMA rings, no barrier in the
 * loop) and add their (max, sum, output) states in wave order; every other case: a VALU kernel with the same structure. */
int cst_dec_cross_attn(const void* q, const void* kx, const void* vx, const uint8_t* key_padding_mask, void* out,
                       const int32_t* step, int64_t max_len, int64_t bsz, int64_t beam, int64_t H, int64_t D, int64_t S, float scale,
                       int dtype, cst_stream stream);
/* The query projection of that attention and the attention itself as ONE launch (modules/transformer_layer.py:369-372 with
 * normalize_before + multihead_attention.py:189-207): x [bsz*beam, H*D] (row stride ldx) is the decoder's residual stream in front of
 * encoder_attn_layer_norm; Wg / sg / sb / eps are cst_dec_ln_linear's folded operands of q_proj (fp32 [H*D] vectors; Wg = bf16(W gamma),
 * [H*D, K = H*D]) with Wg stored FRAGMENT-MAJOR — element (n, k) at  ((((n / 64) * 2 + n % 64 / 32) * (K / 16) + k / 16) * 64 +
 * 32 * (k / 8 % 2) + n % 32) * 8 + k % 8  — so that a wave's load of one MFMA operand is one contiguous KiB (the weights are constants
 * while decoding: packed once; Python: Wg.view(H, 2, 32, K/16, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous()).  The
 * workgroup of (sentence, head) forms its own 64 query columns for the sentence's beam rows while its first K / V tiles stream in, rounds
 * them to bf16 as the stored q of the two-launch path is, and attends as cst_dec_cross_attn does.  bf16, D = 64, beam <= 32, (H*D) % 256 == 0. */
int cst_dec_ln_q_cross_attn(const void* x, int64_t ldx, const void* Wg, const float* sg, const float* sb, float eps, const void* kx, const void* vx,
                            const uint8_t* key_padding_mask, void* out, const int32_t* step, int64_t max_len, int64_t bsz, int64_t beam, int64_t H,
                            int64_t D, int64_t S, float scale, int dtype, cst_stream stream);

/* ------------------------------------------------------------------------------------------
 * Host-side (CPU) natives of the input pipeline — no device work, no stream.
 * cst_batch_by_size replaces the Cython batch_by_size_fast (fairseq/data/data_utils_fast.pyx:17-67): num_tokens[i] is
 *   the size of the i-th sample IN BATCHING ORDER; batches are consecutive runs of that order; batch_sizes (capacity n)
 *   receives their lengths; returns the number of batches, or a negative cst_status (a sample above max_tokens).
 *   max_tokens / max_sentences <= 0 mean "no limit"; bsz_mult = required_batch_size_multiple.
 * cst_wav_info / cst_wav_read_f32 replace soundfile.read(path, dtype="float32", start=, frames=) for 16-bit PCM WAV
 *   (fairseq/data/audio/audio_utils.py:7-55): samples / 32768 as float32, interleaved if multi-channel; nframes < 0 =
 *   to the end; returns frames read or a negative cst_status.  cst_wav_info returns CST_ERR_UNSUPPORTED for non-PCM16.
 * ------------------------------------------------------------------------------------------ */
int64_t cst_batch_by_size(const int64_t* num_tokens, int64_t n, int64_t max_tokens, int64_t max_sentences, int32_t bsz_mult,
                          int64_t* batch_sizes);
int cst_wav_info(const char* path, int32_t* sample_rate, int32_t* channels, int64_t* frames, int32_t* bits);
int64_t cst_wav_read_f32(const char* path, int64_t start_frame, int64_t nframes, float* out, int64_t capacity_samples);

#ifdef __cplusplus
}
#endif
#endif /* CST_H */
