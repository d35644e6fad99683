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

__global__ void beam_init_kernel(BeamP p, int32_t* ticket) {
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
      *ticket = 0;
    }
  }
}

__device__ __forceinline__ bool cand_better(float x, int i, float y, int j) { return x > y || (x == y && i < j); }
constexpr int BEAM_MAX = 20, KMAX_ALL = 2 * BEAM_MAX;

// ---- beam search step, kernel 1 of 2: one workgroup per hypothesis ROW -------------------------------------------------------
// (a) fp32 log-softmax statistics of the row (utils.py:469-473 via models/fairseq_decoder.py:58-79), (b) the masks of
// sequence_generator.py:311-331 and the cumulative-score add of search.py:121-126, (c) the row's top-(2*beam) candidates in
// descending (value, then ascending token) order -> cand_val / cand_tok [row][2*beam].
// The row's logits stay in registers (NV 16-byte vectors per thread) between the statistics pass and the selection; the
// selection is 2*beam block-wide arg-max rounds in which only the winning thread rescans its registers (an earlier version kept
// a sorted top-K list per thread for a whole sentence per workgroup: the divergent insertion chains made it 105 us per step).
template <typename T, int NV>
__global__ __launch_bounds__(512) void beam_row_topk_kernel(BeamP p, float* cand_val, int32_t* cand_tok) {
  constexpr int VEC = DT<T>::VEC, NTH = 512, NW = NTH / 64;
  const int s = *p.step;
  if (s > p.max_len) return;
  const int h = blockIdx.x, r = h % p.beam;
  if (s == 0 && r != 0) return;  // all hypotheses are equal at step 0: only the first beam competes (search.py:121-124)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = p.vocab, K = 2 * p.beam, L1 = p.max_len + 1;
  const int nvec = (V + VEC - 1) / VEC;
  const T* lg = reinterpret_cast<const T*>(p.logits) + (int64_t)h * p.ld_logits;
  const float NEG = -INFINITY;
  __shared__ float red_m[NW], red_s[NW];
  __shared__ float sh_lse;
  __shared__ float w_val[NW * KMAX_ALL];
  __shared__ int w_tok[NW * KMAX_ALL];
  static_assert(KMAX_ALL <= 64 && NW <= 64, "the merge keeps one output candidate / one list head per lane of wave 0");

  float x[NV][VEC];
  float mx = NEG;
  bool nan = false;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = tid + i * NTH;
    if (vi < nvec) {
      if constexpr (VEC == 8) { float t8[8]; load8(lg + (int64_t)vi * 8, t8); for (int e = 0; e < 8; ++e) x[i][e] = t8[e]; }
      else { const f32x4 a4 = *reinterpret_cast<const f32x4*>(lg + (int64_t)vi * 4); for (int e = 0; e < 4; ++e) x[i][e] = a4[e]; }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      x[i][e] = (vi < nvec && vi * VEC + e < V) ? x[i][e] * p.inv_temperature : NEG;
      nan = nan || x[i][e] != x[i][e];
      mx = fmaxf(mx, x[i][e]);
    }
  }
  float sum = 0.0f;
  if (mx != NEG) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < VEC; ++e) sum += expf(x[i][e] - mx);
  }
  if (nan) sum = NAN;  // NaN logits: lse NaN -> every candidate of the row becomes -inf (sequence_generator.py:311)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(mx, o, 64), s2 = __shfl_xor(sum, o, 64);
    const float M = fmaxf(mx, m2);
    sum = (mx == NEG ? 0.0f : sum * expf(mx - M)) + (m2 == NEG ? 0.0f : s2 * expf(m2 - M));
    mx = M;
  }
  if (lane == 0) { red_m[wave] = mx; red_s[wave] = sum; }
  __syncthreads();
  if (tid == 0) {
    float mm = NEG, ss = 0.0f;
    for (int w = 0; w < NW; ++w) {
      const float m2 = red_m[w], s2 = red_s[w];
      const float M = fmaxf(mm, m2);
      ss = (mm == NEG ? 0.0f : ss * expf(mm - M)) + (m2 == NEG ? 0.0f : s2 * expf(m2 - M));
      mm = M;
    }
    sh_lse = mm + logf(ss);
  }
  __syncthreads();
  const float lse = sh_lse;
  const float prev = s > 0 ? (p.scores + (int64_t)(s & 1) * p.bsz * p.beam * L1)[(int64_t)h * L1 + s - 1] : 0.0f;
  // candidate values replace the logits in the registers
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = tid + i * NTH;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int v = vi * VEC + e;
      float val = x[i][e] - lse;
      if (val != val) val = NEG;                       // lprobs[lprobs != lprobs] = -inf        (:311)
      if (v == p.pad) val = NEG;                       // never select pad                        (:313)
      if (v == p.unk) val -= p.unk_penalty;            //                                         (:314)
      if (s >= p.max_len && v != p.eos) val = NEG;     // force eos at max length                 (:317-319)
      if (s < p.min_len && v == p.eos) val = NEG;      // minimum length constraint               (:329-331)
      if (s > 0) val += prev;                          // search.py:125
      x[i][e] = (vi < nvec && v < V) ? val : NAN;      // NaN = not a candidate (never compares better)
    }
  }
  // local best that is strictly worse than (tv, ti) — the thread's previously taken candidate
  float tv = INFINITY, bv;
  int ti = -1, bi;
  auto rescan = [&]() {
    bv = NEG; bi = INT_MAX;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = tid + i * NTH;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        // (bitwise, not short-circuit: the && / || form compiled to ~5 exec-mask branches per element — 121 in the kernel, 3 us per scan)
        const float val = x[i][e];
        const int v = vi * VEC + e;
        const bool open = (val < tv) | ((val == tv) & (v > ti));          // NaN (not a candidate) compares false both ways
        const bool take = open & ((val > bv) | ((val == bv) & (v < bi)));
        bv = take ? val : bv;
        bi = take ? v : bi;
      }
    }
  };
  // Selection in two levels, ONE barrier (round 5; the block-wide arg-max per candidate it replaces cost two barriers and an LDS
  // round trip for each of the 2*beam candidates: 37 us per step): every wave first extracts ITS top-K in order — K rounds of a
  // wave-wide arg-max by shuffles, the winning lane rescans its registers — then wave 0 merges the NW sorted lists, lane w holding
  // the head of wave w's list.  The order is total (value, then token), so the result is the block-wide selection's.
  // the thread's best AND second best in one pass: a lane that wins a round usually has its next head at hand, and the wave enters
  // the (divergent, 24-element) rescan only when one of its lanes wins a third time
  float nv = NEG;
  int ni = INT_MAX;
  bv = NEG; bi = INT_MAX;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = tid + i * NTH;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float val = x[i][e];
      const int v = vi * VEC + e;
      const bool b1 = (val > bv) | ((val == bv) & (v < bi));    // NaN (not a candidate): false
      const bool b2 = (val > nv) | ((val == nv) & (v < ni));
      nv = b1 ? bv : (b2 ? val : nv);
      ni = b1 ? bi : (b2 ? v : ni);
      bv = b1 ? val : bv;
      bi = b1 ? v : bi;
    }
  }
  bool have_next = true;
  for (int k = 0; k < K; ++k) {
    float cv = bv;
    int ci = bi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(cv, o, 64);
      const int i2 = __shfl_xor(ci, o, 64);
      const bool b = (v2 > cv) | ((v2 == cv) & (i2 < ci));
      cv = b ? v2 : cv;
      ci = b ? i2 : ci;
    }
    const bool won = bi == ci && ci != INT_MAX;  // this lane owns the wave's winner: take it, bring up its next best
    if (won) {
      tv = bv; ti = bi;
      bv = nv; bi = ni;
    }
    if (__builtin_expect(__any(won && !have_next), 0)) {
      if (won && !have_next) rescan();
    }
    if (won) have_next = false;
    if (lane == 0) { w_val[wave * KMAX_ALL + k] = cv; w_tok[wave * KMAX_ALL + k] = ci; }
  }
  __syncthreads();
  if (wave == 0) {
    int pos = 0;  // lanes 0 .. NW-1: the next unread entry of wave `lane`'s list
    float hv = lane < NW ? w_val[lane * KMAX_ALL] : NEG;
    int hi_ = lane < NW ? w_tok[lane * KMAX_ALL] : INT_MAX;
    float ov = NEG;
    int ot = INT_MAX;
    for (int k = 0; k < K; ++k) {
      float cv = hv;
      int ci = hi_;
#pragma unroll
      for (int o = NW / 2; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(cv, o, 64);
        const int i2 = __shfl_xor(ci, o, 64);
        if (cand_better(v2, i2, cv, ci)) { cv = v2; ci = i2; }
      }
      cv = __shfl(cv, 0, 64);
      ci = __shfl(ci, 0, 64);
      if (lane < NW && hi_ == ci && ci != INT_MAX) {  // this list's head was taken: advance
        ++pos;
        hv = pos < K ? w_val[lane * KMAX_ALL + pos] : NEG;
        hi_ = pos < K ? w_tok[lane * KMAX_ALL + pos] : INT_MAX;
      }
      if (lane == k) { ov = cv; ot = ci == INT_MAX ? p.pad : ci; }
    }
    if (lane < K) {
      cand_val[(int64_t)h * K + lane] = ov;
      cand_tok[(int64_t)h * K + lane] = ot;
    }
  }
}

// generic-width variant: rows too long for registers are re-read from memory (L2-resident) on every scan
template <typename T>
__global__ __launch_bounds__(512) void beam_row_topk_wide_kernel(BeamP p, float* cand_val, int32_t* cand_tok) {
  constexpr int NTH = 512, NW = NTH / 64;
  const int s = *p.step;
  if (s > p.max_len) return;
  const int h = blockIdx.x, r = h % p.beam;
  if (s == 0 && r != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = p.vocab, K = 2 * p.beam, L1 = p.max_len + 1;
  const T* lg = reinterpret_cast<const T*>(p.logits) + (int64_t)h * p.ld_logits;
  const float NEG = -INFINITY;
  __shared__ float red_m[NW], red_s[NW], wv[NW];
  __shared__ int wi[NW];
  __shared__ float sh_lse;
  __shared__ float o_val[KMAX_ALL];
  __shared__ int o_tok[KMAX_ALL];
  float mx = NEG, sum = 0.0f;
  for (int v = tid; v < V; v += NTH) {
    const float xv = DT<T>::ld(lg + v) * p.inv_temperature;
    if (xv > mx) { sum = (mx == NEG ? 0.0f : sum * expf(mx - xv)) + 1.0f; mx = xv; }
    else sum += expf(xv - mx);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(mx, o, 64), s2 = __shfl_xor(sum, o, 64);
    const float M = fmaxf(mx, m2);
    sum = (mx == NEG ? 0.0f : sum * expf(mx - M)) + (m2 == NEG ? 0.0f : s2 * expf(m2 - M));
    mx = M;
  }
  if (lane == 0) { red_m[wave] = mx; red_s[wave] = sum; }
  __syncthreads();
  if (tid == 0) {
    float mm = NEG, ss = 0.0f;
    for (int w = 0; w < NW; ++w) {
      const float m2 = red_m[w], s2 = red_s[w];
      const float M = fmaxf(mm, m2);
      ss = (mm == NEG ? 0.0f : ss * expf(mm - M)) + (m2 == NEG ? 0.0f : s2 * expf(m2 - M));
      mm = M;
    }
    sh_lse = mm + logf(ss);
  }
  __syncthreads();
  const float lse = sh_lse;
  const float prev = s > 0 ? (p.scores + (int64_t)(s & 1) * p.bsz * p.beam * L1)[(int64_t)h * L1 + s - 1] : 0.0f;
  float tv = INFINITY, bv;
  int ti = -1, bi;
  auto rescan = [&]() {
    bv = NEG; bi = INT_MAX;
    for (int v = tid; v < V; v += NTH) {
      float val = DT<T>::ld(lg + v) * p.inv_temperature - lse;
      if (val != val) val = NEG;
      if (v == p.pad) val = NEG;
      if (v == p.unk) val -= p.unk_penalty;
      if (s >= p.max_len && v != p.eos) val = NEG;
      if (s < p.min_len && v == p.eos) val = NEG;
      if (s > 0) val += prev;
      const bool open = val < tv || (val == tv && v > ti);
      if (open && cand_better(val, v, bv, bi)) { bv = val; bi = v; }
    }
  };
  rescan();
  for (int k = 0; k < K; ++k) {
    float cv = bv;
    int ci = bi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(cv, o, 64);
      const int i2 = __shfl_xor(ci, o, 64);
      if (cand_better(v2, i2, cv, ci)) { cv = v2; ci = i2; }
    }
    if (lane == 0) { wv[wave] = cv; wi[wave] = ci; }
    __syncthreads();
    cv = wv[0]; ci = wi[0];
    for (int w = 1; w < NW; ++w)
      if (cand_better(wv[w], wi[w], cv, ci)) { cv = wv[w]; ci = wi[w]; }
    if (bi == ci && ci != INT_MAX) { tv = bv; ti = bi; rescan(); }
    if (tid == 0) { o_val[k] = cv; o_tok[k] = ci == INT_MAX ? p.pad : ci; }
    __syncthreads();
  }
  if (tid < K) {
    cand_val[(int64_t)h * K + tid] = o_val[tid];
    cand_tok[(int64_t)h * K + tid] = o_tok[tid];
  }
}

// ---- beam search step, kernel 2 of 2: one workgroup per SENTENCE -------------------------------------------------------------
// merges the rows' sorted candidate lists into the sentence's top-(2*beam) over beam*V (search.py:127-135: flat index =
// beam*V + token, ties to the smaller flat index), then (d) the eos / finalize / active-hypothesis bookkeeping of
// sequence_generator.py:340-499 and finalize_hypos :575-696, (e) the token / score / ancestry rows of the next step written
// into the other half of the ping-pong buffers; the last workgroup to finish advances the step counter.
__global__ __launch_bounds__(256) void beam_merge_kernel(BeamP p, const float* cand_val, const int32_t* cand_tok, int32_t* ticket) {
  const int s = *p.step;
  const int sent = blockIdx.x, tid = threadIdx.x;
  const int beam = p.beam, K = 2 * beam, bbsz = p.bsz * beam;
  const int L1 = p.max_len + 1, LT = p.max_len + 2;
  const int rows = s == 0 ? 1 : beam;
  const int cur = s & 1, nxt = cur ^ 1;
  const int64_t* tok_old = p.tokens + (int64_t)cur * bbsz * LT;
  int64_t* tok_new = p.tokens + (int64_t)nxt * bbsz * LT;
  const float* sc_old = p.scores + (int64_t)cur * bbsz * L1;
  float* sc_new = p.scores + (int64_t)nxt * bbsz * L1;
  const int32_t* anc_old = p.anc + (int64_t)cur * bbsz * L1;
  int32_t* anc_new = p.anc + (int64_t)nxt * bbsz * L1;
  const float NEG = -INFINITY;
  __shared__ float l_val[BEAM_MAX * KMAX_ALL];
  __shared__ int l_tok[BEAM_MAX * KMAX_ALL];
  __shared__ float c_score[KMAX_ALL];
  __shared__ int c_tok[KMAX_ALL], c_beam[KMAX_ALL], c_em[KMAX_ALL];
  __shared__ int act[BEAM_MAX], rec_k[BEAM_MAX], rec_r[BEAM_MAX], n_rec;
  __shared__ int ign[BEAM_MAX], ign_new[BEAM_MAX];
  if (s <= p.max_len) {
    for (int i = tid; i < rows * K; i += blockDim.x) {
      l_val[i] = cand_val[(int64_t)sent * beam * K + i];
      l_tok[i] = cand_tok[(int64_t)sent * beam * K + i];
    }
    if (tid < beam) ign[tid] = p.cands_to_ignore[sent * beam + tid];
    __syncthreads();
    // rank of every row candidate among the sentence's rows*K candidates (ordered by value desc, then row asc = flat index asc,
    // then position in the row's list): rank < K -> it is the sentence's candidate number `rank`.  One thread per candidate; a
    // serial K-round head merge by one thread cost ~8 us of dependent LDS round trips.
    for (int i = tid; i < rows * K; i += blockDim.x) {
      const float v = l_val[i];
      const int r = i / K;
      int rank = 0;
      for (int o = 0; o < rows * K; ++o) {
        const float vo = l_val[o];
        rank += (vo > v || (vo == v && o < i)) ? 1 : 0;   // lists are sorted within a row, so o < i orders equal values by (row, position)
      }
      if (rank < K) { c_score[rank] = v; c_tok[rank] = l_tok[i]; c_beam[rank] = r; }
    }
    __syncthreads();
    if (tid == 0) {
      // ---- (d) bookkeeping (LDS / registers only: a global access inside these serial loops costs a memory round trip each) ----
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
      for (int i = 0; i < beam; ++i) ign_new[i] = i >= n_live ? 1 : 0;
    }
    __syncthreads();
    if (tid < beam) p.cands_to_ignore[sent * beam + tid] = (uint8_t)ign_new[tid];
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
      // (one flat loop over (row, position): a loop over the rows around a loop over the positions is `beam` dependent memory round trips)
      for (int idx = tid; idx < beam * (s + 1); idx += blockDim.x) {
        const int i = idx / (s + 1), j = idx - i * (s + 1);
        const int64_t src = sent * beam + c_beam[act[i]], dst = (int64_t)sent * beam + i;
        tok_new[dst * LT + j] = tok_old[src * LT + j];
        anc_new[dst * L1 + j] = anc_old[src * L1 + j];
        if (j < s) sc_new[dst * L1 + j] = sc_old[src * L1 + j];
      }
      if (tid < beam) {
        const int k = act[tid];
        const int64_t dst = (int64_t)sent * beam + tid;
        tok_new[dst * LT + s + 1] = c_tok[k];
        sc_new[dst * L1 + s] = c_score[k];
        anc_new[dst * L1 + s + 1] = (int32_t)dst;
      }
    }
  }
  // the last workgroup to get here advances the step (every workgroup has read *p.step by now)
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) {
      *ticket = 0;
      if (s <= p.max_len) *p.step = s + 1;
      __threadfence();
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

// 16-byte vector of T -> floats
template <typename T>
__device__ __forceinline__ void ld_vec(const T* p, float (&v)[DT<T>::VEC]) {
  if constexpr (DT<T>::VEC == 8) { float v8[8]; load8(p, v8); for (int e = 0; e < 8; ++e) v[e] = v8[e]; }
  else { const f32x4 a = *reinterpret_cast<const f32x4*>(p); for (int e = 0; e < 4; ++e) v[e] = a[e]; }
}
// packed 16-byte vector -> floats (keeps in-flight loads at 4 VGPRs each instead of 8 for bf16)
template <typename T>
__device__ __forceinline__ void cvt_vec(const u32x4& r, float (&v)[DT<T>::VEC]) {
  if constexpr (DT<T>::VEC == 8) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(r[i]);
  }
}
template <typename T>
__device__ __forceinline__ void st_vec(T* p, const float (&v)[DT<T>::VEC]) {
  if constexpr (DT<T>::VEC == 8) { float v8[8]; for (int e = 0; e < 8; ++e) v8[e] = v[e]; store8(p, v8); }
  else { f32x4 a = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(p) = a; }
}

// Single-query self-attention over the append-only caches; one wave per (hypothesis, head), no LDS.
// A key / value row of one head is D*sizeof(T) contiguous bytes: LPK = that / 16 lanes cover it with one 16-byte load each, so a
// 64-lane load instruction fetches KPI = 64 / LPK whole rows (full 128-byte lines, not 64 scattered 16-byte pieces).  One pass
// with an online softmax: the K and V loads of UNR * KPI positions are issued together (the loop is a chain of memory round
// trips, so bytes in flight per wave is what sets its speed); scores = per-lane partial dots reduced over the LPK lanes of a key;
// the output = per-slot partial sums reduced over the KPI slots at the end.
template <typename T, int D>
__global__ __launch_bounds__(256) void dec_self_attn_kernel(const T* qkv, T* kc, T* vc, const int32_t* anc2, const int32_t* stepp,
                                                            T* out, int rows, int H, int max_len, float scale) {
  constexpr int VEC = DT<T>::VEC;          // elements per 16-byte vector
  constexpr int LPK = D / VEC;             // lanes per key row
  constexpr int KPI = 64 / LPK;            // key rows per load instruction
  constexpr int UNR = 8;
  const int s = *stepp;
  if (s > max_len) return;
  const int L1 = max_len + 1, C = H * D;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x * (blockDim.x >> 6) + wave;
  if (item >= rows * H) return;            // waves are independent: no block-level barrier below
  const int h = item / H, head = item % H;
  const int32_t* anc = anc2 + ((int64_t)(s & 1) * rows + h) * L1;
  const int slot = lane / LPK, c0 = (lane % LPK) * VEC;
  const T* qp = qkv + (int64_t)h * 3 * C + head * D + c0;
  const T* kn = qp + C;
  const T* vn = qp + 2 * C;
  const int64_t hoff = (int64_t)head * D + c0;
  float q[VEC];
  ld_vec<T>(qp, q);
#pragma unroll
  for (int e = 0; e < VEC; ++e) q[e] *= scale;
  // caches are head-major [rows][H][L1][D]: the positions of one head are contiguous (128-byte rows back to back), so the KPI
  // rows of a load instruction are one contiguous KPI*128-byte run wherever the ancestry does not switch rows
  if (slot == 0) {  // append this step's key / value (slot 0's lanes hold one full row between them)
    *reinterpret_cast<u32x4*>(kc + (((int64_t)h * H + head) * L1 + s) * D + c0) = *reinterpret_cast<const u32x4*>(kn);
    *reinterpret_cast<u32x4*>(vc + (((int64_t)h * H + head) * L1 + s) * D + c0) = *reinterpret_cast<const u32x4*>(vn);
  }
  float m = -INFINITY, l = 0.0f, acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.0f;
  // bf16 (the timed configuration): scores in log2 units (one v_exp_f32 per probability), the 8-lane dot-product sums by DPP instead
  // of ds_bpermute.  fp32 (the parity configuration) keeps expf and the shuffle order its token ids were pinned with.
  constexpr bool FAST = sizeof(T) == 2 && LPK == 8;
  if (FAST) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) q[e] *= 1.4426950408889634f;
  }
  // The ancestry entries of a batch are a memory round trip IN FRONT of its K / V loads (their addresses): the next batch's are
  // requested while this one is worked on, so only the first batch pays for both.
  int an[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int j = u * KPI + slot;
    an[u] = j < s ? anc[j] : 0;
  }
  for (int j0 = 0; j0 <= s; j0 += KPI * UNR) {
    u32x4 tk[UNR], tv[UNR];
    float part[UNR];
    int an_next[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int jn = j0 + KPI * UNR + u * KPI + slot;
      an_next[u] = jn < s ? anc[jn] : 0;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = j0 + u * KPI + slot;
      if (j <= s) {  // position s comes from the projection buffer, not from the cache (written above by other lanes)
        const int64_t off = j == s ? 0 : (((int64_t)an[u] * H + head) * L1 + j) * D + c0;
        tk[u] = *reinterpret_cast<const u32x4*>(j == s ? kn : kc + off);
        tv[u] = *reinterpret_cast<const u32x4*>(j == s ? vn : vc + off);
      }
    }
    float bm = -INFINITY;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = j0 + u * KPI + slot;
      float d = 0.0f;
      if (j <= s) {
        float kf[VEC];
        cvt_vec<T>(tk[u], kf);
#pragma unroll
        for (int e = 0; e < VEC; ++e) d = fmaf(q[e], kf[e], d);
      }
      if constexpr (FAST) {
        d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
        d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x141, 0xF, 0xF, true));  // row_half_mirror: the other quad of the 8
      } else {
#pragma unroll
        for (int o = LPK >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
      }
      part[u] = j <= s ? d : -INFINITY;
      bm = fmaxf(bm, part[u]);
    }
    bm = wave_max(bm);
    const float mn = fmaxf(m, bm);
    const float corr = m == -INFINITY ? 0.0f : (FAST ? __builtin_amdgcn_exp2f(m - mn) : expf(m - mn));
    l *= corr;
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] *= corr;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = j0 + u * KPI + slot;
      if (j <= s) {
        const float pe = FAST ? __builtin_amdgcn_exp2f(part[u] - mn) : expf(part[u] - mn);
        l += pe;
        float vf[VEC];
        cvt_vec<T>(tv[u], vf);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = fmaf(pe, vf[e], acc[e]);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) an[u] = an_next[u];
    m = mn;
  }
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) {
    l += __shfl_xor(l, o, 64);
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] /= l;
  if (slot == 0) st_vec<T>(out + (int64_t)h * C + hoff, acc);
}

// Cross attention of the decode step: one workgroup per (sentence, head).  The encoder keys / values of a sentence are shared
// by its `beam` hypotheses (they are never replicated or reordered: the reference's reorder_encoder_out + static_kv caches hold
// beam copies), so every K / V row is read once per step and used for BQ queries.  4 waves split the source positions; rows are
// fetched as whole 128-byte lines (LPK lanes per row, as in the self-attention kernel).  ONE pass: the K and V loads of a batch
// of UNR * KPI positions are issued together and folded into per-query online-softmax state (running max, sum, output); the
// four waves' states are merged through LDS at the end.  (The first version made three passes — scores to LDS, exp, P V — i.e.
// twice the dependent memory round trips per wave: 51 us per layer against 37 for the flash kernel on the same shape.)
template <typename T, int D, int BQ, int NW>
__global__ __launch_bounds__(NW * 64, 8 / NW) void dec_cross_attn_kernel(const T* q, const T* kx, const T* vx, const uint8_t* kpm, T* out,
                                                                const int32_t* stepp, int max_len, int beam, int H, int S, float scale) {
  constexpr int VEC = DT<T>::VEC, LPK = D / VEC, KPI = 64 / LPK, UNR = 4;
  extern __shared__ float smem[];
  if (*stepp > max_len) return;
  float* sm_m = smem;                    // [NW][BQ]
  float* sm_l = smem + NW * BQ;          // [NW][BQ]
  float* sm_o = smem + 2 * NW * BQ;      // [NW][BQ][D]
  float* sm_q = sm_o + NW * BQ * D;      // [BQ][D] scaled queries (read per batch instead of pinning BQ * VEC registers)
  const int b = blockIdx.x / H, head = blockIdx.x % H, C = H * D;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int slot = lane / LPK, c0 = (lane % LPK) * VEC;
  const T* kb = kx + ((int64_t)b * H + head) * S * D + c0;  // head-major [bsz][H][S][D]: a head's rows are one contiguous stream
  const T* vb = vx + ((int64_t)b * H + head) * S * D + c0;
  const uint8_t* mk = kpm ? kpm + (int64_t)b * S : nullptr;
  for (int q0 = 0; q0 < beam; q0 += BQ) {
    const int nq = beam - q0 < BQ ? beam - q0 : BQ;
    __syncthreads();
    for (int i = tid; i < BQ * D; i += NW * 64) {
      const int qi = i / D, dd = i % D;
      sm_q[i] = qi < nq ? DT<T>::ld(q + ((int64_t)(b * beam + q0 + qi)) * C + head * D + dd) * scale : 0.0f;
    }
    __syncthreads();
    float mq[BQ], lq[BQ], acc[BQ][VEC];
#pragma unroll
    for (int qi = 0; qi < BQ; ++qi) {
      mq[qi] = -INFINITY; lq[qi] = 0.0f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[qi][e] = 0.0f;
    }
    for (int j0 = wave * KPI * UNR; j0 < S; j0 += NW * KPI * UNR) {
      u32x4 tk[UNR], tv[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int j = j0 + u * KPI + slot;
        if (j < S) {
          tk[u] = *reinterpret_cast<const u32x4*>(kb + (int64_t)j * D);
          tv[u] = *reinterpret_cast<const u32x4*>(vb + (int64_t)j * D);
        }
      }
      float part[UNR][BQ];
      float bm[BQ];
#pragma unroll
      for (int qi = 0; qi < BQ; ++qi) bm[qi] = -INFINITY;
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int j = j0 + u * KPI + slot;
        const bool ok = j < S && !(mk && mk[j < S ? j : 0]);
        float kf[VEC];
        cvt_vec<T>(tk[u], kf);
#pragma unroll
        for (int qi = 0; qi < BQ; ++qi) {
          float d = 0.0f;
          if (j < S) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) d = fmaf(sm_q[qi * D + c0 + e], kf[e], d);
          }
#pragma unroll
          for (int o = LPK >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
          part[u][qi] = ok ? d : -INFINITY;
          bm[qi] = fmaxf(bm[qi], part[u][qi]);
        }
      }
#pragma unroll
      for (int qi = 0; qi < BQ; ++qi) {
        const float mn = fmaxf(mq[qi], wave_max(bm[qi]));
        const float corr = (mq[qi] == -INFINITY) ? 0.0f : expf(mq[qi] - mn);
        lq[qi] *= corr;
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[qi][e] *= corr;
        mq[qi] = mn;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int j = j0 + u * KPI + slot;
        if (j < S) {
          float vf[VEC];
          cvt_vec<T>(tv[u], vf);
#pragma unroll
          for (int qi = 0; qi < BQ; ++qi) {
            const float pe = part[u][qi] == -INFINITY ? 0.0f : expf(part[u][qi] - mq[qi]);
            lq[qi] += pe;
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[qi][e] = fmaf(pe, vf[e], acc[qi][e]);
          }
        }
      }
    }
    // per-wave state: reduce the KPI slots (every lane of a slot counted its keys once per lane -> l is per slot), publish
#pragma unroll
    for (int qi = 0; qi < BQ; ++qi) {
#pragma unroll
      for (int o = LPK; o < 64; o <<= 1) {
        lq[qi] += __shfl_xor(lq[qi], o, 64);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[qi][e] += __shfl_xor(acc[qi][e], o, 64);
      }
      if (lane == 0) { sm_m[wave * BQ + qi] = mq[qi]; sm_l[wave * BQ + qi] = lq[qi]; }
      if (slot == 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) sm_o[(wave * BQ + qi) * D + c0 + e] = acc[qi][e];
      }
    }
    __syncthreads();
    for (int i = tid; i < nq * D; i += NW * 64) {
      const int qi = i / D, dd = i % D;
      float M = -INFINITY;
#pragma unroll
      for (int w = 0; w < NW; ++w) M = fmaxf(M, sm_m[w * BQ + qi]);
      float l = 0.0f, o = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float f = sm_m[w * BQ + qi] == -INFINITY ? 0.0f : expf(sm_m[w * BQ + qi] - M);
        l += sm_l[w * BQ + qi] * f;
        o += sm_o[(w * BQ + qi) * D + dd] * f;
      }
      DT<T>::st(out + ((int64_t)(b * beam + q0 + qi)) * C + head * D + dd, o / l);
    }
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

// ---- decode-step Linear (include/cst.h: cst_dec_linear) ---------------------------------------------------------------------------
// y[M, N] = act(x[M, K] W[N, K]^T + b) (+ resid) for the few hundred hypothesis rows of a beam-search step.  The step is bound by
// streaming each weight matrix (2-8 MB) from HBM ONCE; that takes every byte of it requested within one memory round trip, so the
// matrix is cut into many small workgroups that keep nothing but loads in flight:
//   workgroup = NT x 16 weight rows (output columns) x TT x 16 hypothesis rows, 4 waves; wave w owns K quarter w for the whole tile,
//   operands go global -> registers (no LDS staging: each wave reads its 16-byte k chunks of W rows and x rows straight into
//   MFMA fragments; x comes from L2, it is shared by every workgroup), v_mfma_f32_16x16x32_bf16 with D = W-frag x x-frag, so a lane
//   ends up with 4 consecutive output columns of one hypothesis row; the weights of the next UN k-steps are requested before the
//   MFMAs of the current ones.  The four waves' partial sums are added in wave order through LDS (deterministic), then bias /
//   activation / residual and one 8-byte store per lane and tile.
// LN (fused LayerNorm of the rows of x, models/transformer_layer.py:346-349 / :369-372 / :403-406 in front of the projection):
//   LN(x) W^T + b = rstd_m (x Wg^T - mean_m sg) + sb   with Wg = W * gamma (columns scaled, rounded to bf16 once when the engine packs
//   its weights), sg[n] = sum_k Wg[n,k], sb[n] = sum_k W[n,k] beta[k] + b[n] (fp32).  The kernel multiplies the RAW rows by Wg and
//   gathers sum(x), sum(x^2) of every row from the fragments it loads anyway (each workgroup sees whole rows), so the LayerNorm
//   launch and its activation round trip disappear; W is then Wg, `bias` is unused, ln_sg / ln_sb are the two vectors.
template <int NT, int TT, int UN, bool LN, int NW>
__global__ __launch_bounds__(NW * 64) void dec_linear_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W, const bf16_t* __restrict__ bias,
                                                         const bf16_t* resid, bf16_t* y, int M, int N, int K, int64_t ldx, int64_t ldr,
                                                         int64_t ldy, int act, const int32_t* stepp, int max_len, const float* ln_sg,
                                                         const float* ln_sb, float ln_eps) {
  if (stepp && *stepp > max_len) return;
  using f32x4v = __attribute__((ext_vector_type(4))) float;
  __shared__ float red[NW][NT * TT][64][4];
  __shared__ float stat[LN ? NW : 1][TT][16][2];
  float s1[TT], s2[TT];
#pragma unroll
  for (int i = 0; i < TT; ++i) s1[i] = s2[i] = 0.0f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, c = lane >> 4;
  const int n0 = blockIdx.x * (16 * NT), t0 = blockIdx.y * (16 * TT);
  const int kq = K / NW;  // this wave's K range: [wave * kq, (wave + 1) * kq)
  const bf16_t* wp[NT];
  const bf16_t* xp[TT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    int n = n0 + 16 * j + r;
    n = n < N ? n : N - 1;
    wp[j] = W + (int64_t)n * K + wave * kq + c * 8;
  }
#pragma unroll
  for (int i = 0; i < TT; ++i) {
    int t = t0 + 16 * i + r;
    t = t < M ? t : M - 1;
    xp[i] = x + (int64_t)t * ldx + wave * kq + c * 8;
  }
  f32x4v acc[NT][TT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < TT; ++i) acc[j][i] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
  bf16x8 wv[UN][NT], wn[UN][NT], xv[UN][TT];
#pragma unroll
  for (int u = 0; u < UN; ++u)
#pragma unroll
    for (int j = 0; j < NT; ++j) wn[u][j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[j] + 32 * u));
  for (int k = 0; k < kq; k += 32 * UN) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
#pragma unroll
      for (int j = 0; j < NT; ++j) wv[u][j] = wn[u][j];
#pragma unroll
      for (int i = 0; i < TT; ++i) xv[u][i] = *reinterpret_cast<const bf16x8*>(xp[i] + k + 32 * u);
    }
    if (k + 32 * UN < kq) {
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int j = 0; j < NT; ++j) wn[u][j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[j] + k + 32 * UN + 32 * u));
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < TT; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[u][j], xv[u][i], acc[j][i], 0, 0, 0);
    if (LN) {
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int i = 0; i < TT; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = static_cast<float>(xv[u][i][e]);
            s1[i] += f;
            s2[i] = fmaf(f, f, s2[i]);
          }
    }
  }
  if (LN) {  // row sums: over the four k-chunk lane groups of this wave, then (below) over the four waves' K quarters, in a fixed order
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      s1[i] += __shfl_xor(s1[i], 16, 64); s2[i] += __shfl_xor(s2[i], 16, 64);
      s1[i] += __shfl_xor(s1[i], 32, 64); s2[i] += __shfl_xor(s2[i], 32, 64);
      if (c == 0) { stat[wave][i][r][0] = s1[i]; stat[wave][i][r][1] = s2[i]; }
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][j * TT + i][lane][e] = acc[j][i][e];
  __syncthreads();
  for (int tile = wave; tile < NT * TT; tile += NW) {
    const int j = tile / TT, i = tile % TT;
    const int n = n0 + 16 * j + 4 * c, t = t0 + 16 * i + r;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = red[0][tile][lane][e];
#pragma unroll
      for (int w = 1; w < NW; ++w) a += red[w][tile][lane][e];  // wave order: deterministic
      v[e] = a;
    }
    if (t >= M || n >= N) continue;
    float mean = 0.0f, rstd = 1.0f;
    if (LN) {
      float a = stat[0][i][r][0], b2 = stat[0][i][r][1];
#pragma unroll
      for (int w = 1; w < NW; ++w) { a += stat[w][i][r][0]; b2 += stat[w][i][r][1]; }
      mean = a / (float)K;
      const float var = fmaxf(b2 / (float)K - mean * mean, 0.0f);
      rstd = rsqrtf(var + ln_eps);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (n + e < N) {
        float o = LN ? fmaf(rstd, v[e] - mean * ln_sg[n + e], ln_sb[n + e]) : v[e] + (bias ? DT<bf16_t>::ld(bias + n + e) : 0.0f);
        o = act_t<bf16_t>(o, act);
        if (resid) o += DT<bf16_t>::ld(resid + (int64_t)t * ldr + n + e);
        v[e] = o;
      }
    }
    bf16_t* dst = y + (int64_t)t * ldy + n;
    if (n + 4 <= N && ((ldy | n) & 3) == 0) {
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      bf16x4_t pk;
#pragma unroll
      for (int e = 0; e < 4; ++e) pk[e] = static_cast<__bf16>(v[e]);
      *reinterpret_cast<bf16x4_t*>(dst) = pk;
    } else {
      for (int e = 0; e < 4 && n + e < N; ++e) DT<bf16_t>::st(dst + e, v[e]);
    }
  }
}

}  // namespace

extern "C" {
static int dec_linear_impl(const void* x, const void* W, const void* bias, const void* resid, void* y, int64_t M, int64_t N, int64_t K,
                           int64_t ldx, int64_t ld_resid, int64_t ldy, int act, const int32_t* step, int64_t max_len, int dtype,
                           const float* ln_sg, const float* ln_sb, float ln_eps, cst_stream stream) {
  CST_REQUIRE(x && W && y && M > 0 && N > 0 && K > 0, "cst_dec_linear: bad args");
  CST_REQUIRE(dtype == CST_BF16, "cst_dec_linear: bf16 only (fp32 problems go through cst_gemm)");
  CST_REQUIRE(K % 512 == 0 && ldx % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)W % 16) == 0, "cst_dec_linear: K %% 512 == 0 and 16-byte aligned rows required (K=%lld)", (long long)K);
  CST_REQUIRE(M <= 4096, "cst_dec_linear: a decode-step kernel (M=%lld rows)", (long long)M);
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_GEMM, s, 2.0 * M * N * K, 2.0 * ((double)N * K + (double)M * K + (double)M * N));
  const int ty = (int)cst_ceil_div(M, 80);
  const int64_t wg1 = cst_ceil_div(N, 16) * ty;
  const bool ln = ln_sg != nullptr;
  // Waves per workgroup = how finely K is cut: each wave should have one or two batches of UN k-steps (32 columns each), so that
  // nearly all of its weight bytes are requested at once.  Enough workgroups to have every CU pulling weights, few enough that the
  // shared x rows are not re-read from L2 more than needed.
  static const int env_nw = getenv("CST_DEC_LINEAR_NW") ? atoi(getenv("CST_DEC_LINEAR_NW")) : 0;
  int nw = 8;  // measured (tools/bench_dec_linear.py): 8 waves 10.2 / 8.1 / 10.3 us on the three LayerNorm-fused shapes, 4 waves 10.9 / 8.6 / 11.0; 16 no better
  if (env_nw == 4 || env_nw == 8 || env_nw == 16) nw = env_nw;
  while (nw > 4 && K % (nw * 64) != 0) nw >>= 1;
  const int steps = (int)(K / nw / 32);  // k-steps per wave
#define CST_DL(NTv, UNv, LNv, NWv)                                                                                                  \
  hipLaunchKernelGGL((dec_linear_kernel<NTv, 5, UNv, LNv, NWv>), dim3((unsigned)cst_ceil_div(N, 16 * NTv), (unsigned)ty), dim3(NWv * 64), 0, s, \
                     (const bf16_t*)x, (const bf16_t*)W, (const bf16_t*)bias, (const bf16_t*)resid, (bf16_t*)y, (int)M, (int)N, (int)K, ldx, \
                     ld_resid, ldy, act, step, (int)max_len, ln_sg, ln_sb, ln_eps)
#define CST_DL_LN(NTv, UNv, NWv) do { if (ln) CST_DL(NTv, UNv, true, NWv); else CST_DL(NTv, UNv, false, NWv); } while (0)
  if (wg1 <= 320) {  // one 16-column tile per workgroup
    if (nw == 16) { if (steps % 4 == 0) CST_DL_LN(1, 4, 16); else CST_DL_LN(1, 2, 16); }
    else if (nw == 8) { if (steps % 4 == 0) CST_DL_LN(1, 4, 8); else CST_DL_LN(1, 2, 8); }
    else CST_DL_LN(1, 4, 4);
  } else if (wg1 <= 1024) {  // two column tiles; the wave-partial image of 16 waves would not fit the LDS
    const int nw2 = nw == 16 ? 8 : nw, steps2 = (int)(K / nw2 / 32);
    if (nw2 == 8) { if (steps2 % 4 == 0) CST_DL_LN(2, 4, 8); else CST_DL_LN(2, 2, 8); }
    else CST_DL_LN(2, 4, 4);
  } else {
    CST_DL_LN(4, 2, 4);
  }
#undef CST_DL_LN
#undef CST_DL
  return cst_check_launch("cst_dec_linear");
}

int cst_dec_linear(const void* x, const void* W, const void* bias, const void* resid, void* y, int64_t M, int64_t N, int64_t K,
                   int64_t ldx, int64_t ld_resid, int64_t ldy, int act, const int32_t* step, int64_t max_len, int dtype,
                   cst_stream stream) {
  return dec_linear_impl(x, W, bias, resid, y, M, N, K, ldx, ld_resid, ldy, act, step, max_len, dtype, nullptr, nullptr, 0.0f, stream);
}

int cst_dec_ln_linear(const void* x, const void* Wg, const float* sg, const float* sb, float eps, const void* resid, void* y, int64_t M,
                      int64_t N, int64_t K, int64_t ldx, int64_t ld_resid, int64_t ldy, int act, const int32_t* step, int64_t max_len,
                      int dtype, cst_stream stream) {
  CST_REQUIRE(sg && sb && eps > 0.0f, "cst_dec_ln_linear: the folded LayerNorm vectors are required");
  return dec_linear_impl(x, Wg, nullptr, resid, y, M, N, K, ldx, ld_resid, ldy, act, step, max_len, dtype, sg, sb, eps, stream);
}


int cst_beam_init(const cst_beam_desc* d, cst_stream stream) {
  BeamP p;
  const int rc = to_params(d, p);
  if (rc != CST_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  CST_REQUIRE(d->workspace != nullptr, "cst_beam_init: workspace of cst_beam_workspace() bytes required");
  hipLaunchKernelGGL(beam_init_kernel, dim3(p.bsz * p.beam), dim3(256), 0, s, p, reinterpret_cast<int32_t*>(d->workspace));
  return cst_check_launch("cst_beam_init");
}

int64_t cst_beam_workspace(int64_t bsz, int64_t beam) {
  // per row 2*beam (value, token) candidates + the step ticket
  return bsz * beam * 2 * beam * (int64_t)(sizeof(float) + sizeof(int32_t)) + 64;
}

int cst_beam_step(const cst_beam_desc* d, cst_stream stream) {
  BeamP p;
  const int rc = to_params(d, p);
  if (rc != CST_OK) return rc;
  CST_REQUIRE(d->logits != nullptr && d->ld_logits >= d->vocab, "cst_beam_step: null logits / ld_logits < vocab");
  const int64_t vec = d->dtype == CST_BF16 ? 8 : 4;
  CST_REQUIRE(d->ld_logits % vec == 0 && d->ld_logits >= cst_ceil_div(d->vocab, vec) * vec && ((uintptr_t)d->logits % 16) == 0,
              "cst_beam_step: logits rows must be 16-byte aligned and padded to a multiple of %lld elements", (long long)vec);
  CST_REQUIRE(d->workspace != nullptr && ((uintptr_t)d->workspace % 16) == 0, "cst_beam_step: workspace of cst_beam_workspace() bytes required");
  hipStream_t s = (hipStream_t)stream;
  const int64_t rows = (int64_t)p.bsz * p.beam, K = 2 * p.beam;
  int32_t* ticket = reinterpret_cast<int32_t*>(d->workspace);
  float* cand_val = reinterpret_cast<float*>(reinterpret_cast<char*>(d->workspace) + 64);
  int32_t* cand_tok = reinterpret_cast<int32_t*>(cand_val + rows * K);
  {
    CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, (double)rows * p.vocab * cst_dtype_size(d->dtype));
    const int64_t nvec = cst_ceil_div(d->vocab, vec), per_thread = cst_ceil_div(nvec, 512);
#define CST_TOPK(T, NV) hipLaunchKernelGGL((beam_row_topk_kernel<T, NV>), dim3((unsigned)rows), dim3(512), 0, s, p, cand_val, cand_tok)
#define CST_TOPK_T(T) do { if (per_thread <= 1) CST_TOPK(T, 1); else if (per_thread <= 3) CST_TOPK(T, 3); else if (per_thread <= 5) CST_TOPK(T, 5); \
                            else hipLaunchKernelGGL((beam_row_topk_wide_kernel<T>), dim3((unsigned)rows), dim3(512), 0, s, p, cand_val, cand_tok); } while (0)
    if (d->dtype == CST_BF16) CST_TOPK_T(bf16_t); else CST_TOPK_T(float);
#undef CST_TOPK_T
#undef CST_TOPK
    hipLaunchKernelGGL(beam_merge_kernel, dim3(p.bsz), dim3(256), 0, s, p, (const float*)cand_val, (const int32_t*)cand_tok, ticket);
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
  const size_t lds = 0;
  CstProfScope prof(CST_K_ATTN_FWD, s, 4.0 * rows * H * D * (max_len + 1) * 0.5, 0.0);
#define CST_DSA(T, DD) hipLaunchKernelGGL((dec_self_attn_kernel<T, DD>), dim3(blocks), dim3(64 * WPB), lds, s, (const T*)qkv, (T*)kcache, (T*)vcache, \
                                          anc, step, (T*)out, (int)rows, (int)H, (int)max_len, scale)
  if (dtype == CST_BF16) { if (D == 64) CST_DSA(bf16_t, 64); else CST_DSA(bf16_t, 32); }
  else { if (D == 64) CST_DSA(float, 64); else CST_DSA(float, 32); }
#undef CST_DSA
  return cst_check_launch("cst_dec_self_attn");
}

int cst_dec_cross_attn(const void* q, const void* kx, const void* vx, const uint8_t* key_padding_mask, void* out,
                       const int32_t* step, int64_t max_len, int64_t bsz, int64_t beam, int64_t H, int64_t D, int64_t S, float scale,
                       int dtype, cst_stream stream) {
  CST_REQUIRE(q && kx && vx && out && step, "cst_dec_cross_attn: null operand");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_dec_cross_attn: bad dtype %d", dtype);
  CST_REQUIRE(D == 32 || D == 64, "cst_dec_cross_attn: head dim %lld not in {32,64}", (long long)D);
  CST_REQUIRE(bsz > 0 && beam > 0 && H > 0 && S > 0, "cst_dec_cross_attn: bad shape");
  hipStream_t s = (hipStream_t)stream;
  // bf16 / head dim 64 / at most 32 hypotheses per sentence: the matrix-core kernel whose four waves split the keys (attention_fast.inc);
  // CST_DEC_CROSS_VALU=1 keeps the VALU kernel below (A/B runs)
  static const bool valu_only = getenv("CST_DEC_CROSS_VALU") != nullptr;
  if (!valu_only && dtype == CST_BF16 && D == 64 && beam <= 32 && S * 128 < (1ll << 31) && ((uintptr_t)q % 16 == 0) && ((uintptr_t)kx % 16 == 0) &&
      ((uintptr_t)vx % 16 == 0) && ((uintptr_t)out % 16 == 0)) {
    CstProfScope prof(CST_K_ATTN_FWD, s, 4.0 * bsz * beam * H * D * S, 2.0 * bsz * S * H * D * cst_dtype_size(dtype));
    const int rc = cst_fa_dec_cross(q, kx, vx, key_padding_mask, out, step, max_len, bsz, beam, H, S, scale, nullptr, nullptr, nullptr, 0.0f, 0, 0, s);
    return rc != CST_OK ? rc : cst_check_launch("cst_dec_cross_attn");
  }
  const int BQ = beam == 1 ? 1 : (beam <= 5 ? 5 : 8);
  // waves per workgroup = slices the key range is cut into (each wave streams its own keys; more waves = more loads in flight)
  static const int env_nw = getenv("CST_DEC_CROSS_NW") ? atoi(getenv("CST_DEC_CROSS_NW")) : 0;
  const int NWv = env_nw == 8 ? 8 : 4;  // 8 measured no faster end to end (and the flash kernel remains the engine's default: 0.198 vs 0.230 s per batch)
  const size_t lds = ((size_t)2 * NWv * BQ + (size_t)NWv * BQ * D + (size_t)BQ * D) * sizeof(float);
  CstProfScope prof(CST_K_ATTN_FWD, s, 4.0 * bsz * beam * H * D * S, 2.0 * bsz * S * H * D * cst_dtype_size(dtype));
#define CST_DCA_W(T, DD, QQ, WW)                                                                                                \
  do {                                                                                                                          \
    static bool attr = false;                                                                                                   \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_cross_attn_kernel<T, DD, QQ, WW>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr = true; } \
    hipLaunchKernelGGL((dec_cross_attn_kernel<T, DD, QQ, WW>), dim3((unsigned)(bsz * H)), dim3(WW * 64), lds, s, (const T*)q, (const T*)kx, (const T*)vx, \
                       key_padding_mask, (T*)out, step, (int)max_len, (int)beam, (int)H, (int)S, scale);                       \
  } while (0)
#define CST_DCA(T, DD, QQ) do { if (NWv == 8) CST_DCA_W(T, DD, QQ, 8); else CST_DCA_W(T, DD, QQ, 4); } while (0)
#define CST_DCA_Q(T, DD) do { if (BQ == 1) CST_DCA(T, DD, 1); else if (BQ == 5) CST_DCA(T, DD, 5); else CST_DCA(T, DD, 8); } while (0)
  if (dtype == CST_BF16) { if (D == 64) CST_DCA_Q(bf16_t, 64); else CST_DCA_Q(bf16_t, 32); }
  else { if (D == 64) CST_DCA_Q(float, 64); else CST_DCA_Q(float, 32); }
#undef CST_DCA_Q
#undef CST_DCA
#undef CST_DCA_W
  return cst_check_launch("cst_dec_cross_attn");
}

int cst_dec_ln_q_cross_attn(const void* x, int64_t ldx, const void* Wg, const float* sg, const float* sb, float eps, const void* kx, const void* vx,
                            const uint8_t* key_padding_mask, void* out, const int32_t* step, int64_t max_len, int64_t bsz, int64_t beam, int64_t H,
                            int64_t D, int64_t S, float scale, int dtype, cst_stream stream) {
  CST_REQUIRE(x && Wg && sg && sb && kx && vx && out && step, "cst_dec_ln_q_cross_attn: null operand");
  CST_REQUIRE(dtype == CST_BF16 && D == 64 && beam > 0 && beam <= 32, "cst_dec_ln_q_cross_attn: bf16, head dim 64, beam <= 32 (the unfused pair covers the rest)");
  CST_REQUIRE(bsz > 0 && H > 0 && S > 0 && S * 128 < (1ll << 31), "cst_dec_ln_q_cross_attn: bad shape");
  const int64_t K = H * D;  // the query projection is square: embed_dim -> H * D
  CST_REQUIRE(K % 256 == 0 && ldx % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)Wg % 16) == 0 && ((uintptr_t)sg % 16) == 0 && ((uintptr_t)sb % 16) == 0 &&
                  ((uintptr_t)kx % 16) == 0 && ((uintptr_t)vx % 16) == 0 && ((uintptr_t)out % 16) == 0,
              "cst_dec_ln_q_cross_attn: embed dim %% 256 == 0 and 16-byte aligned operands required");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ATTN_FWD, s, 4.0 * bsz * beam * H * D * S + 2.0 * bsz * beam * K * K, 2.0 * bsz * S * H * D * 2 + 2.0 * K * K);
  const int rc = cst_fa_dec_cross(x, kx, vx, key_padding_mask, out, step, max_len, bsz, beam, H, S, scale, Wg, sg, sb, eps, K, ldx, s);
  return rc != CST_OK ? rc : cst_check_launch("cst_dec_ln_q_cross_attn");
}

}  // extern "C"
