// decode.hip — device-resident incremental decoding: the per-step work of fairseq/sequence_generator.py:_generate
// (:286-541) + search.py BeamSearch.step (:109-144) + the single-query self-attention of the incremental MHA branch
// (modules/multihead_attention.py:189-293) as kernels that read the step counter from DEVICE memory, so that one decode
// step (embed -> 6 decoder layers -> vocabulary projection -> beam search bookkeeping) is a fixed launch sequence that the
// host captures once in a hipGraph and replays; the host never synchronises inside the loop (it polls `num_remaining`).
//
// K/V caches are append-only: hypothesis row h writes its new key/value at slot [h][step] and the beam reorder of
// reorder_incremental_state (multihead_attention.py:419-437: index_select on [B*beam, H, t, D] every step) is replaced
// by an ancestry table anc[h][j] = the cache row that holds position j of hypothesis h — reordering moves (t+1) int32
// per hypothesis instead of 2 * layers * t * C cache elements.
#include "cst_common.h"
#include <limits.h>

namespace {

struct BeamP {
  int bsz, beam, vocab, max_len;
  int pad, unk, eos, min_len;
  float unk_penalty, len_penalty, inv_temperature;
  int normalize_scores;
  const void* logits; int64_t ld_logits;
  int32_t* step;
  int64_t* tokens; float* scores; int32_t* anc;
  uint8_t* cands_to_ignore; uint8_t* finished; int32_t* nfinal; int32_t* num_remaining;
  int64_t* fin_tokens; float* fin_pos; float* fin_score; int32_t* fin_len;
};

__global__ void beam_init_kernel(BeamP p) {
  const int h = blockIdx.x, bbsz = p.bsz * p.beam, L1 = p.max_len + 1, LT = p.max_len + 2;
  for (int buf = 0; buf < 2; ++buf) {
    int64_t* tk = p.tokens + ((int64_t)buf * bbsz + h) * LT;
    float* sc = p.scores + ((int64_t)buf * bbsz + h) * L1;
    int32_t* an = p.anc + ((int64_t)buf * bbsz + h) * L1;
    for (int j = threadIdx.x; j < LT; j += blockDim.x) tk[j] = j == 0 ? p.eos : p.pad;
    for (int j = threadIdx.x; j < L1; j += blockDim.x) {
      sc[j] = 0.0f;
      an[j] = j == 0 ? h : 0;
    }
  }
  if (threadIdx.x == 0) {
    p.fin_len[h] = 0;
    p.fin_score[h] = 0.0f;
    if (h % p.beam == 0) {
      const int s = h / p.beam;
      p.finished[s] = 0;
      p.nfinal[s] = 0;
      for (int b = 0; b < p.beam; ++b) p.cands_to_ignore[s * p.beam + b] = 0;
    }
    if (h == 0) {
      *p.step = 0;
      *p.num_remaining = p.bsz;
    }
  }
}

__global__ void beam_advance_kernel(int32_t* step) { *step += 1; }

__device__ __forceinline__ bool cand_better(float x, int i, float y, int j) { return x > y || (x == y && i < j); }

// One workgroup per sentence.  (a) fp32 log-softmax statistics of the `rows` live hypothesis rows (utils.py:469-473 via
// models/fairseq_decoder.py:58-79), (b) the masks of sequence_generator.py:311-331 and the cumulative-score add of search.py:121-126,
// (c) top-(2*beam) over rows*V candidates (search.py:127-135): per-thread sorted lists merged by 2*beam block-wide arg-max
// rounds, (d) the eos / finalize / active-hypothesis bookkeeping of sequence_generator.py:340-499 and finalize_hypos :575-696,
// (e) the token / score / ancestry rows of the next step written into the other half of the ping-pong buffers.
template <typename T, int KMAX>
__global__ __launch_bounds__(1024) void beam_step_kernel(BeamP p) {
  constexpr int KB = KMAX / 2 > 0 ? KMAX / 2 : 1;  // max beam for this instantiation
  const int s = *p.step;
  if (s > p.max_len) return;
  const int sent = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
  const int beam = p.beam, V = p.vocab, K = 2 * beam, bbsz = p.bsz * beam;
  const int L1 = p.max_len + 1, LT = p.max_len + 2;
  const int rows = s == 0 ? 1 : beam;
  const int cur = s & 1, nxt = cur ^ 1;
  const int64_t* tok_old = p.tokens + (int64_t)cur * bbsz * LT;
  int64_t* tok_new = p.tokens + (int64_t)nxt * bbsz * LT;
  const float* sc_old = p.scores + (int64_t)cur * bbsz * L1;
  float* sc_new = p.scores + (int64_t)nxt * bbsz * L1;
  const int32_t* anc_old = p.anc + (int64_t)cur * bbsz * L1;
  int32_t* anc_new = p.anc + (int64_t)nxt * bbsz * L1;
  const T* logits = reinterpret_cast<const T*>(p.logits) + (int64_t)sent * beam * p.ld_logits;
  const float NEG = -INFINITY;

  __shared__ float red_m[16][KB], red_s[16][KB];
  __shared__ float row_lse[KB], row_prev[KB];
  __shared__ float wv[16];
  __shared__ int wi[16];
  __shared__ float c_score[KMAX];
  __shared__ int c_tok[KMAX], c_beam[KMAX], c_em[KMAX];
  __shared__ int act[KB], rec_k[KB], rec_r[KB], n_rec;

  // ---- (a) online max / sum-exp per row ----
  float m[KB], sum[KB];
#pragma unroll
  for (int r = 0; r < KB; ++r) { m[r] = NEG; sum[r] = 0.0f; }
  for (int v = tid; v < V; v += blockDim.x) {
#pragma unroll
    for (int r = 0; r < KB; ++r) {
      if (r < rows) {
        const float x = DT<T>::ld(logits + (int64_t)r * p.ld_logits + v) * p.inv_temperature;
        if (x > m[r]) { sum[r] = sum[r] * expf(m[r] - x) + 1.0f; m[r] = x; }
        else sum[r] += expf(x - m[r]);  // NaN logits propagate into the sum -> lse NaN -> row masked to -inf below
      }
    }
  }
#pragma unroll
  for (int r = 0; r < KB; ++r) {
    if (r < rows) {
      float mm = m[r], ss = sum[r];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(mm, o, 64), s2 = __shfl_xor(ss, o, 64);
        const float M = fmaxf(mm, m2);
        ss = (mm == NEG ? 0.0f : ss * expf(mm - M)) + (m2 == NEG ? 0.0f : s2 * expf(m2 - M));
        mm = M;
      }
      if (lane == 0) { red_m[wave][r] = mm; red_s[wave][r] = ss; }
    }
  }
  __syncthreads();
  if (tid < rows) {
    float mm = NEG, ss = 0.0f;
    for (int w = 0; w < NW; ++w) {
      const float m2 = red_m[w][tid], s2 = red_s[w][tid];
      const float M = fmaxf(mm, m2);
      ss = (mm == NEG ? 0.0f : ss * expf(mm - M)) + (m2 == NEG ? 0.0f : s2 * expf(m2 - M));
      mm = M;
    }
    row_lse[tid] = mm + logf(ss);
    row_prev[tid] = s > 0 ? sc_old[(int64_t)(sent * beam + tid) * L1 + s - 1] : 0.0f;
  }
  __syncthreads();

  // ---- (b)+(c) masked candidate values, per-thread sorted top-KMAX ----
  float lv[KMAX];
  int li[KMAX];
#pragma unroll
  for (int t = 0; t < KMAX; ++t) { lv[t] = NEG; li[t] = INT_MAX; }
  for (int r = 0; r < rows; ++r) {
    const float lse = row_lse[r], prev = row_prev[r];
    const T* lg = logits + (int64_t)r * p.ld_logits;
    for (int v = tid; v < V; v += blockDim.x) {
      float val = DT<T>::ld(lg + v) * p.inv_temperature - lse;
      if (val != val) val = NEG;                       // lprobs[lprobs != lprobs] = -inf        (:311)
      if (v == p.pad) val = NEG;                       // never select pad                        (:313)
      if (v == p.unk) val -= p.unk_penalty;            //                                         (:314)
      if (s >= p.max_len && v != p.eos) val = NEG;     // force eos at max length                 (:317-319)
      if (s < p.min_len && v == p.eos) val = NEG;      // minimum length constraint               (:329-331)
      if (s > 0) val += prev;                          // search.py:125
      const int idx = r * V + v;
      if (cand_better(val, idx, lv[KMAX - 1], li[KMAX - 1])) {
        lv[KMAX - 1] = val; li[KMAX - 1] = idx;
#pragma unroll
        for (int t = KMAX - 1; t > 0; --t) {
          if (cand_better(lv[t], li[t], lv[t - 1], li[t - 1])) {
            const float tv = lv[t]; lv[t] = lv[t - 1]; lv[t - 1] = tv;
            const int ti = li[t]; li[t] = li[t - 1]; li[t - 1] = ti;
          }
        }
      }
    }
  }
  // ---- merge: K block-wide arg-max rounds over the list heads ----
  for (int k = 0; k < K; ++k) {
    float bv = lv[0];
    int bi = li[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(bv, o, 64);
      const int i2 = __shfl_xor(bi, o, 64);
      if (cand_better(v2, i2, bv, bi)) { bv = v2; bi = i2; }
    }
    if (lane == 0) { wv[wave] = bv; wi[wave] = bi; }
    __syncthreads();
    bv = wv[0]; bi = wi[0];
    for (int w = 1; w < NW; ++w)
      if (cand_better(wv[w], wi[w], bv, bi)) { bv = wv[w]; bi = wi[w]; }
    if (li[0] == bi && bi != INT_MAX) {  // this thread owns the winner: pop it
#pragma unroll
      for (int t = 0; t < KMAX - 1; ++t) { lv[t] = lv[t + 1]; li[t] = li[t + 1]; }
      lv[KMAX - 1] = NEG; li[KMAX - 1] = INT_MAX;
    }
    if (tid == 0) {
      c_score[k] = bv;
      c_tok[k] = bi == INT_MAX ? p.pad : bi % V;
      c_beam[k] = bi == INT_MAX ? 0 : bi / V;
    }
    __syncthreads();
  }

  // ---- (d) bookkeeping (one thread; K <= 40 entries) ----
  if (tid == 0) {
    uint8_t* ign = p.cands_to_ignore + sent * beam;
    bool any_top_eos = false;
    int nr = 0;
    int nf = p.nfinal[sent];
    const bool was_finished = p.finished[sent] != 0;
    for (int k = 0; k < K; ++k) {
      bool e = c_tok[k] == p.eos && c_score[k] != NEG;                       // :341
      if (k < beam && ign[k]) e = false;                                     // :346
      c_em[k] = e ? 1 : 0;
      if (k < beam && e) {
        any_top_eos = true;
        if (!was_finished && nf < beam) { rec_k[nr] = k; rec_r[nr] = nf; ++nr; ++nf; }   // finalize_hypos :575-696
      }
    }
    n_rec = nr;
    p.nfinal[sent] = nf;
    if (any_top_eos && !was_finished && (nf == beam || s == p.max_len)) {     // is_finished :698-713
      p.finished[sent] = 1;
      atomicSub(p.num_remaining, 1);
    }
    // active hypotheses: the first `beam` candidates that are not eos / ignored, in candidate order (:465-499)
    int na = 0;
    for (int k = 0; k < K && na < beam; ++k) {
      const bool e = c_em[k] || (k < beam && ign[k]);
      if (!e) act[na++] = k;
    }
    const int n_live = na;
    for (int k = 0; k < K && na < beam; ++k) {
      const bool e = c_em[k] || (k < beam && ign[k]);
      if (e) act[na++] = k;
    }
    for (int i = 0; i < beam; ++i) ign[i] = i >= n_live ? 1 : 0;
  }
  __syncthreads();

  // ---- finalized hypotheses (tokens[bi, 1:step+2] with eos at [step]; positional scores = differences) ----
  for (int q = 0; q < n_rec; ++q) {
    const int k = rec_k[q], r = rec_r[q];
    const int64_t bi = sent * beam + c_beam[k], slot = (int64_t)sent * beam + r;
    const float sc = c_score[k];
    for (int j = tid; j <= s; j += blockDim.x) {
      p.fin_tokens[slot * L1 + j] = j == s ? (int64_t)p.eos : tok_old[bi * LT + j + 1];
      const float cum = j == s ? sc : sc_old[bi * L1 + j];
      const float before = j > 0 ? sc_old[bi * L1 + j - 1] : 0.0f;
      p.fin_pos[slot * L1 + j] = j > 0 ? cum - before : cum;
    }
    if (tid == 0) {
      p.fin_len[slot] = s + 1;
      p.fin_score[slot] = p.normalize_scores ? sc / (float)pow((double)(s + 1), (double)p.len_penalty) : sc;
    }
  }
  // ---- (e) rows of the next step ----
  if (s < p.max_len) {
    for (int i = 0; i < beam; ++i) {
      const int k = act[i];
      const int64_t src = sent * beam + c_beam[k], dst = (int64_t)sent * beam + i;
      for (int j = tid; j <= s; j += blockDim.x) {
        tok_new[dst * LT + j] = tok_old[src * LT + j];
        anc_new[dst * L1 + j] = anc_old[src * L1 + j];
        if (j < s) sc_new[dst * L1 + j] = sc_old[src * L1 + j];
      }
      if (tid == 0) {
        tok_new[dst * LT + s + 1] = c_tok[k];
        sc_new[dst * L1 + s] = c_score[k];
        anc_new[dst * L1 + s + 1] = (int32_t)dst;
      }
    }
  }
}

// x[h] = embed_scale * E[tokens[h][step]] + P[pad + 1 + step]   (models/transformer.py:744-760 on the incremental branch;
// sinusoidal_positional_embedding.py:88-95: the position of the newest token is pad + seq_len for every row)
template <typename T>
__global__ void dec_embed_kernel(const int64_t* tokens, const int32_t* stepp, const T* embed, const float* pos, float scale,
                                 int pad, T* out, int rows, int C, int max_len, int pos_rows) {
  const int s = *stepp;
  if (s > max_len) return;
  const int h = blockIdx.x, LT = max_len + 2;
  const int64_t tok = tokens[((int64_t)(s & 1) * rows + h) * LT + s];
  const int pr = pad + 1 + s < pos_rows ? pad + 1 + s : pos_rows - 1;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
#pragma clang fp contract(off)  // the reference multiplies, rounds, then adds (two ATen kernels): no fused multiply-add here
    const float e = DT<T>::ld(embed + tok * C + c);
    const float se = scale * e;
    DT<T>::st(out + (int64_t)h * C + c, se + pos[(int64_t)pr * C + c]);
  }
}

// Single-query self-attention over the append-only caches; one wave per (hypothesis, head).
template <typename T, int D>
__global__ __launch_bounds__(256) void dec_self_attn_kernel(const T* qkv, T* kc, T* vc, const int32_t* anc2, const int32_t* stepp,
                                                            T* out, int rows, int H, int max_len, float scale) {
  extern __shared__ float smem[];
  const int s = *stepp;
  if (s > max_len) return;
  const int L1 = max_len + 1, C = H * D;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x * (blockDim.x >> 6) + wave;
  const bool valid = item < rows * H;
  const int h = valid ? item / H : 0, head = valid ? item % H : 0;
  float* pr = smem + wave * L1;
  const int32_t* anc = anc2 + ((int64_t)(s & 1) * rows + h) * L1;
  const T* qp = qkv + (int64_t)h * 3 * C + head * D;
  const T* kn = qp + C;
  const T* vn = qp + 2 * C;
  float q[D];
#pragma unroll
  for (int d = 0; d < D; d += 8) {
    float t[8];
    load8(qp + d, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) q[d + e] = t[e];
  }
  if (valid && lane < D) {  // append this step's key / value (read back from the projection registers below, not from the cache)
    kc[((int64_t)h * L1 + s) * C + head * D + lane] = kn[lane];
    vc[((int64_t)h * L1 + s) * C + head * D + lane] = vn[lane];
  }
  float mx = -INFINITY;
  if (valid) {
    for (int j = lane; j <= s; j += 64) {
      const T* kr = j == s ? kn : kc + ((int64_t)anc[j] * L1 + j) * C + head * D;
      float acc = 0.0f;
#pragma unroll
      for (int d = 0; d < D; d += 8) {
        float t[8];
        load8(kr + d, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf(q[d + e], t[e], acc);
      }
      acc *= scale;
      pr[j] = acc;
      mx = fmaxf(mx, acc);
    }
  }
  mx = wave_max(mx);
  float sum = 0.0f;
  if (valid) {
    for (int j = lane; j <= s; j += 64) {
      const float e = expf(pr[j] - mx);
      pr[j] = e;
      sum += e;
    }
  }
  sum = wave_sum(sum);
  __syncthreads();
  if (valid && lane < D) {
    float acc = 0.0f;
    for (int j = 0; j < s; ++j) {
      const float pj = pr[j];
      acc = fmaf(pj, DT<T>::ld(vc + ((int64_t)anc[j] * L1 + j) * C + head * D + lane), acc);
    }
    acc = fmaf(pr[s], DT<T>::ld(vn + lane), acc);
    DT<T>::st(out + (int64_t)h * C + head * D + lane, acc / sum);
  }
}

int to_params(const cst_beam_desc* d, BeamP& p) {
  CST_REQUIRE(d != nullptr, "cst_beam: null descriptor");
  CST_REQUIRE(d->dtype == CST_F32 || d->dtype == CST_BF16, "cst_beam: bad dtype %d", d->dtype);
  CST_REQUIRE(d->bsz > 0 && d->beam > 0 && d->beam <= 20, "cst_beam: bsz %lld / beam %lld (beam <= 20)", (long long)d->bsz, (long long)d->beam);
  CST_REQUIRE(d->vocab > 2 * d->beam + 1 && d->beam * d->vocab < INT_MAX, "cst_beam: vocabulary %lld too small / large for beam %lld",
              (long long)d->vocab, (long long)d->beam);
  CST_REQUIRE(d->max_len >= 1 && d->min_len <= d->max_len, "cst_beam: max_len %lld / min_len %lld", (long long)d->max_len, (long long)d->min_len);
  CST_REQUIRE(d->temperature > 0.0f, "cst_beam: temperature must be positive");
  CST_REQUIRE(d->step && d->tokens && d->scores && d->anc && d->cands_to_ignore && d->finished && d->nfinal && d->num_remaining &&
                  d->fin_tokens && d->fin_pos && d->fin_score && d->fin_len, "cst_beam: null state buffer");
  p.bsz = (int)d->bsz; p.beam = (int)d->beam; p.vocab = (int)d->vocab; p.max_len = (int)d->max_len;
  p.pad = (int)d->pad; p.unk = (int)d->unk; p.eos = (int)d->eos; p.min_len = (int)d->min_len;
  p.unk_penalty = d->unk_penalty; p.len_penalty = d->len_penalty; p.inv_temperature = 1.0f / d->temperature;
  p.normalize_scores = d->normalize_scores;
  p.logits = d->logits; p.ld_logits = d->ld_logits;
  p.step = d->step; p.tokens = d->tokens; p.scores = d->scores; p.anc = d->anc;
  p.cands_to_ignore = d->cands_to_ignore; p.finished = d->finished; p.nfinal = d->nfinal; p.num_remaining = d->num_remaining;
  p.fin_tokens = d->fin_tokens; p.fin_pos = d->fin_pos; p.fin_score = d->fin_score; p.fin_len = d->fin_len;
  return CST_OK;
}

}  // namespace

extern "C" {

int cst_beam_init(const cst_beam_desc* d, cst_stream stream) {
  BeamP p;
  const int rc = to_params(d, p);
  if (rc != CST_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(beam_init_kernel, dim3(p.bsz * p.beam), dim3(256), 0, s, p);
  return cst_check_launch("cst_beam_init");
}

int cst_beam_step(const cst_beam_desc* d, cst_stream stream) {
  BeamP p;
  const int rc = to_params(d, p);
  if (rc != CST_OK) return rc;
  CST_REQUIRE(d->logits != nullptr && d->ld_logits >= d->vocab, "cst_beam_step: null logits / ld_logits < vocab");
  hipStream_t s = (hipStream_t)stream;
  const int K = 2 * p.beam;
  {
    CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, (double)p.bsz * p.beam * p.vocab * cst_dtype_size(d->dtype) * 2.0);
#define CST_BEAM(T, KM) hipLaunchKernelGGL((beam_step_kernel<T, KM>), dim3(p.bsz), dim3(1024), 0, s, p)
#define CST_BEAM_K(T) do { if (K <= 2) CST_BEAM(T, 2); else if (K <= 10) CST_BEAM(T, 10); else if (K <= 20) CST_BEAM(T, 20); else CST_BEAM(T, 40); } while (0)
    if (d->dtype == CST_BF16) CST_BEAM_K(bf16_t); else CST_BEAM_K(float);
#undef CST_BEAM_K
#undef CST_BEAM
    hipLaunchKernelGGL(beam_advance_kernel, dim3(1), dim3(1), 0, s, p.step);
  }
  return cst_check_launch("cst_beam_step");
}

int cst_dec_embed(const int64_t* tokens, const int32_t* step, const void* embed, const float* pos_table, float scale,
                  int64_t pad_idx, void* out, int64_t rows, int64_t C, int64_t max_len, int64_t pos_rows, int dtype,
                  cst_stream stream) {
  CST_REQUIRE(tokens && step && embed && pos_table && out, "cst_dec_embed: null operand");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_dec_embed: bad dtype %d", dtype);
  CST_REQUIRE(rows > 0 && C > 0 && max_len >= 1 && pos_rows > pad_idx + 1, "cst_dec_embed: bad shape");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, (double)rows * C * (2.0 * cst_dtype_size(dtype) + 4.0));
  if (dtype == CST_BF16)
    hipLaunchKernelGGL(dec_embed_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, s, tokens, step, (const bf16_t*)embed, pos_table, scale,
                       (int)pad_idx, (bf16_t*)out, (int)rows, (int)C, (int)max_len, (int)pos_rows);
  else
    hipLaunchKernelGGL(dec_embed_kernel<float>, dim3((unsigned)rows), dim3(256), 0, s, tokens, step, (const float*)embed, pos_table, scale,
                       (int)pad_idx, (float*)out, (int)rows, (int)C, (int)max_len, (int)pos_rows);
  return cst_check_launch("cst_dec_embed");
}

int cst_dec_self_attn(const void* qkv, void* kcache, void* vcache, const int32_t* anc, const int32_t* step, void* out,
                      int64_t rows, int64_t H, int64_t D, int64_t max_len, float scale, int dtype, cst_stream stream) {
  CST_REQUIRE(qkv && kcache && vcache && anc && step && out, "cst_dec_self_attn: null operand");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_dec_self_attn: bad dtype %d", dtype);
  CST_REQUIRE(D == 32 || D == 64, "cst_dec_self_attn: head dim %lld not in {32,64}", (long long)D);
  CST_REQUIRE(rows > 0 && H > 0 && max_len >= 1 && max_len <= 4000, "cst_dec_self_attn: bad shape (max_len <= 4000)");
  hipStream_t s = (hipStream_t)stream;
  const int WPB = 4;
  const unsigned blocks = (unsigned)cst_ceil_div(rows * H, WPB);
  const size_t lds = (size_t)WPB * (max_len + 1) * sizeof(float);
  CstProfScope prof(CST_K_ATTN_FWD, s, 4.0 * rows * H * D * (max_len + 1) * 0.5, 0.0);
#define CST_DSA(T, DD) hipLaunchKernelGGL((dec_self_attn_kernel<T, DD>), dim3(blocks), dim3(64 * WPB), lds, s, (const T*)qkv, (T*)kcache, (T*)vcache, \
                                          anc, step, (T*)out, (int)rows, (int)H, (int)max_len, scale)
  if (dtype == CST_BF16) { if (D == 64) CST_DSA(bf16_t, 64); else CST_DSA(bf16_t, 32); }
  else { if (D == 64) CST_DSA(float, 64); else CST_DSA(float, 32); }
#undef CST_DSA
  return cst_check_launch("cst_dec_self_attn");
}

}  // extern "C"
