// embed.hip — token embedding + sinusoidal positions for the TRAINING path:
//   out[b,t,:] = dropout( scale * src[b,t,:] + pos_table[ position(b,t) ] )
//     src       = embed[tokens[b,t]]  (nn.Embedding, models/transformer.py:744-760 / w2v2_transformer_interlingua.py:216)
//                 or a dense feature row x[b,t,:] (the audio encoder of s2t_transformer_w2v2, w2v2_transformer.py:352-358)
//     position  = utils.make_positions (utils.py:235-245): pad_idx + (number of non-pad symbols in [0, t]) for a non-pad symbol,
//                 pad_idx for a pad symbol, evaluated here by a wave-parallel count (no cumsum tensor, no index tensor);
//     pos_table = the sin || cos table of modules/sinusoidal_positional_embedding.py:36-58 — a TABLE, not in-kernel
//                 trigonometry: the rows are data the reference computes once on the host (row pad_idx is zero), re-reading
//                 them costs C * 4 bytes per token out of L2, in-kernel sinf/cosf would make an HBM-bound kernel VALU-bound.
//   backward:  dE[v,:] = scale * sum_{(b,t): tokens = v != pad} keep * dy[b,t,:]   — deterministic, no atomics, no sort: the
//                 FIRST occurrence of every symbol owns the symbol's row and adds up the later occurrences in index order;
//              dx = scale * keep * dy for the dense variant (cst_dropout with alpha).
// One wave per output row, 16-byte vectors along C, HBM-bound: reads C * (dtype + 4) bytes, writes C * dtype bytes per token.
#include "cst_common.h"

namespace {

constexpr int EMB_WAVES = 4;  // rows (waves) per workgroup

// non-pad predicate of make_positions: `tensor.ne(padding_idx)` on the token ids, or on the 0/1 padding mask the encoders pass
__device__ __forceinline__ bool emb_nonpad(const int64_t* tokens, const uint8_t* pad_mask, int64_t i, int pad_idx) {
  return pad_mask ? ((int)pad_mask[i] != pad_idx) : (tokens[i] != (int64_t)pad_idx);
}

template <typename T>
__global__ __launch_bounds__(64 * EMB_WAVES) void embed_pos_fwd_kernel(const int64_t* tokens, const uint8_t* pad_mask, const T* embed, const T* x,
                                                                        const float* pos_table, float scale, int pad_idx, T* out, int B, int Tn, int C,
                                                                        int V, int pos_rows, uint32_t key, uint32_t thr16, float dscale) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * EMB_WAVES + wave;
  if (row >= (int64_t)B * Tn) return;
  const int b = (int)(row / Tn), t = (int)(row % Tn);
  const int64_t base = (int64_t)b * Tn;
  int pos = pad_idx;
  if (pos_table) {
    const bool me = emb_nonpad(tokens, pad_mask, base + t, pad_idx);
    int cnt = 0;
    for (int j = lane; j <= t; j += 64) cnt += emb_nonpad(tokens, pad_mask, base + j, pad_idx) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    pos = me ? pad_idx + cnt : pad_idx;
    if (pos >= pos_rows) pos = pos_rows - 1;  // the host sizes the table to pad_idx + 1 + T: never taken
  }
  const T* src;
  if (embed) {
    int64_t tok = tokens[base + t];
    if (tok < 0 || tok >= V) tok = pad_idx;  // out-of-range ids read the (zero) pad row instead of faulting
    src = embed + tok * C;
  } else {
    src = x + row * C;
  }
  const float* pr = pos_table ? pos_table + (int64_t)pos * C : nullptr;
  T* dst = out + row * C;
  for (int c = lane * 8; c < C; c += 64 * 8) {
    float v[8];
    load8(src + c, v);
    if (pr) {
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(pr + c), p1 = *reinterpret_cast<const f32x4*>(pr + c + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = fmaf(scale, v[e], p0[e]); v[4 + e] = fmaf(scale, v[4 + e], p1[e]); }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= scale;
    }
    if (thr16) cst_drop8(v, key, (uint64_t)(row * C + c), thr16, dscale);
    store8(dst + c, v);
  }
}

// One workgroup per token occurrence i; only the FIRST occurrence of a symbol works: it walks the whole token list in index
// order (256 ids per pass, ordered ballot compaction into LDS), adds scale * keep * dy[j,:] for every occurrence j in that order
// into fp32 registers and writes the row of dE once, rounded once.  Fixed summation order -> bit-reproducible; the rows of
// symbols that do not occur (and the pad row) are zero-filled by the launcher.
template <typename T, typename G>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const T* dy, const int64_t* tokens, G* dE, float scale, int pad_idx, int n, int C, int V,
                                                        uint32_t key, uint32_t thr16, float dscale) {
  __shared__ int s_list[256];
  __shared__ int s_wcnt[4];
  __shared__ int s_flag;
  const int i = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t tok = tokens[i];
  if (tok == (int64_t)pad_idx || tok < 0 || tok >= V) return;  // padding_idx rows receive no gradient (F.embedding)
  // an earlier occurrence owns the row
  if (tid == 0) s_flag = 0;
  __syncthreads();
  int earlier = 0;
  for (int j = tid; j < i; j += 256) earlier |= (tokens[j] == tok) ? 1 : 0;
  if (earlier) s_flag = 1;
  __syncthreads();
  if (s_flag) return;
  constexpr int MAXV = 4;  // 8-element vectors per thread: C <= 256 * 8 * 4
  float acc[MAXV][8];
#pragma unroll
  for (int q = 0; q < MAXV; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[q][e] = 0.0f;
  for (int j0 = i; j0 < n; j0 += 256) {
    const int j = j0 + tid;
    const bool hit = j < n && tokens[j] == tok;
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_wcnt[wave] = __popcll(m);
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += s_wcnt[w];
    const int total = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    if (hit) s_list[off + __popcll(m & ((1ull << lane) - 1ull))] = j;
    __syncthreads();
    for (int h = 0; h < total; ++h) {
      const int64_t r = s_list[h];
#pragma unroll
      for (int q = 0; q < MAXV; ++q) {
        const int c = (q * 256 + tid) * 8;
        if (c < C) {
          float v[8];
          load8(dy + r * C + c, v);
          if (thr16) cst_drop8(v, key, (uint64_t)(r * C + c), thr16, dscale);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[q][e] += v[e];
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < MAXV; ++q) {
    const int c = (q * 256 + tid) * 8;
    if (c < C) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc[q][e] * scale;
      store8(dE + tok * C + c, v);
    }
  }
}

template <typename T>
__global__ void dropout_scale_kernel(const T* x, T* y, int64_t n8, float alpha, uint32_t key, uint32_t thr16, float dscale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    load8(x + i * 8, v);
    if (thr16) cst_drop8(v, key, (uint64_t)i * 8, thr16, dscale);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= alpha;
    store8(y + i * 8, v);
  }
}

}  // namespace

extern "C" int cst_embed_pos_fwd(const int64_t* tokens, const uint8_t* pad_mask, const void* embed, const void* x, const float* pos_table,
                                 float scale, int64_t pad_idx, void* out, int64_t B, int64_t T, int64_t C, int64_t V, int64_t pos_rows,
                                 float drop_p, uint32_t drop_key, int dtype, cst_stream stream) {
  CST_REQUIRE(out && (embed != nullptr) != (x != nullptr), "cst_embed_pos_fwd: exactly one of embed / x must be given");
  CST_REQUIRE(!embed || tokens, "cst_embed_pos_fwd: an embedding lookup needs tokens");
  CST_REQUIRE(!pos_table || tokens || pad_mask, "cst_embed_pos_fwd: positions need tokens or a padding mask");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_embed_pos_fwd: bad dtype %d", dtype);
  CST_REQUIRE(B > 0 && T > 0 && C > 0 && C % 8 == 0 && B * T < (1ll << 31), "cst_embed_pos_fwd: bad shape B=%lld T=%lld C=%lld (C %% 8 == 0)",
              (long long)B, (long long)T, (long long)C);
  CST_REQUIRE(!pos_table || pos_rows >= pad_idx + 1 + T, "cst_embed_pos_fwd: position table has %lld rows, needs %lld", (long long)pos_rows,
              (long long)(pad_idx + 1 + T));
  CST_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "cst_embed_pos_fwd: bad dropout p");
  hipStream_t s = (hipStream_t)stream;
  const size_t es = cst_dtype_size(dtype);
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, (double)B * T * C * (2.0 * es + (pos_table ? 4.0 : 0.0)));
  const uint32_t thr = drop_p > 0.0f ? cst_drop_thr16(drop_p) : 0u;
  const float dscale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const unsigned blocks = (unsigned)cst_ceil_div(B * T, EMB_WAVES);
  if (dtype == CST_BF16)
    hipLaunchKernelGGL(embed_pos_fwd_kernel<bf16_t>, dim3(blocks), dim3(64 * EMB_WAVES), 0, s, tokens, pad_mask, (const bf16_t*)embed, (const bf16_t*)x,
                       pos_table, scale, (int)pad_idx, (bf16_t*)out, (int)B, (int)T, (int)C, (int)V, (int)pos_rows, drop_key, thr, dscale);
  else
    hipLaunchKernelGGL(embed_pos_fwd_kernel<float>, dim3(blocks), dim3(64 * EMB_WAVES), 0, s, tokens, pad_mask, (const float*)embed, (const float*)x,
                       pos_table, scale, (int)pad_idx, (float*)out, (int)B, (int)T, (int)C, (int)V, (int)pos_rows, drop_key, thr, dscale);
  return cst_check_launch("cst_embed_pos_fwd");
}

extern "C" int cst_embed_bwd(const void* dy, const int64_t* tokens, void* dE, float scale, int64_t pad_idx, int64_t n, int64_t C, int64_t V,
                             float drop_p, uint32_t drop_key, int dtype, int grad_dtype, cst_stream stream) {
  CST_REQUIRE(dy && tokens && dE, "cst_embed_bwd: null operand");
  CST_REQUIRE((dtype == CST_F32 || dtype == CST_BF16) && (grad_dtype == CST_F32 || grad_dtype == dtype), "cst_embed_bwd: bad dtype %d / %d", dtype, grad_dtype);
  CST_REQUIRE(n > 0 && n < (1ll << 31) && V > 0 && C > 0 && C % 8 == 0 && C <= 256 * 8 * 4, "cst_embed_bwd: bad shape n=%lld C=%lld (C %% 8 == 0, C <= 8192)",
              (long long)n, (long long)C);
  CST_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "cst_embed_bwd: bad dropout p");
  hipStream_t s = (hipStream_t)stream;
  const size_t gs = cst_dtype_size(grad_dtype);
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, (double)n * C * cst_dtype_size(dtype) + 2.0 * V * C * gs);
  if (hipMemsetAsync(dE, 0, (size_t)V * C * gs, s) != hipSuccess) { cst_set_error("cst_embed_bwd: memset failed"); return CST_ERR_LAUNCH; }
  const uint32_t thr = drop_p > 0.0f ? cst_drop_thr16(drop_p) : 0u;
  const float dscale = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
#define CST_EB(T, G) hipLaunchKernelGGL((embed_bwd_kernel<T, G>), dim3((unsigned)n), dim3(256), 0, s, (const T*)dy, tokens, (G*)dE, scale, (int)pad_idx, (int)n, \
                                        (int)C, (int)V, drop_key, thr, dscale)
  if (dtype == CST_BF16) { if (grad_dtype == CST_BF16) CST_EB(bf16_t, bf16_t); else CST_EB(bf16_t, float); }
  else CST_EB(float, float);
#undef CST_EB
  return cst_check_launch("cst_embed_bwd");
}

extern "C" int cst_dropout_scale(const void* x, void* y, int64_t n, float alpha, float p, uint32_t key, int dtype, cst_stream stream) {
  CST_REQUIRE(x && y && n > 0 && n % 8 == 0, "cst_dropout_scale: bad args (n %% 8 == 0)");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_dropout_scale: bad dtype %d", dtype);
  CST_REQUIRE(p >= 0.0f && p < 1.0f, "cst_dropout_scale: bad dropout p");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, 2.0 * n * cst_dtype_size(dtype));
  const uint32_t thr = p > 0.0f ? cst_drop_thr16(p) : 0u;
  const float dscale = p > 0.0f ? 1.0f / (1.0f - p) : 1.0f;
  int64_t blocks = cst_ceil_div(n / 8, 256);
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  if (dtype == CST_BF16) hipLaunchKernelGGL(dropout_scale_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, n / 8, alpha, key, thr, dscale);
  else hipLaunchKernelGGL(dropout_scale_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x, (float*)y, n / 8, alpha, key, thr, dscale);
  return cst_check_launch("cst_dropout_scale");
}
