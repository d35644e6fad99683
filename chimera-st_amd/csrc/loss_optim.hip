// loss_optim.hip — (1) label-smoothed cross entropy over vocabulary logits, fused
// log-softmax + NLL + smoothing, one workgroup per target row, logits read once forward and
// once backward (HBM-bound; the fp32 log-prob tensor of the reference is never materialised).
// Replaces models/fairseq_decoder.py:75-79 -> utils.py:469-473 and
// criterions/label_smoothed_cross_entropy.py:13-30.
// (2) optimizer path: sum of squares (grad norm) and a single-pass fused Adam with fp32 master
// weights (optim/fp16_optimizer.py:16-300, optim/adam.py:146-226, utils.py:323-364).
#include "cst_common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.0f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t = fmaxf(t, red[w]);
  return t;
}

// loss_row = (1-eps) * (lse - x_t) + (eps/V) * (V*lse - sum_v x_v);   nll_row = lse - x_t
template <typename T>
__global__ __launch_bounds__(256) void ls_ce_fwd_kernel(const T* logits, const int64_t* target, float* rows2, float* lse_out,
                                                        int64_t V, float eps, int64_t pad) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const T* x = logits + row * V;
  float mx = -INFINITY;
  for (int64_t v = threadIdx.x; v < V; v += blockDim.x) mx = fmaxf(mx, DT<T>::ld(x + v));
  mx = block_max(mx, red);
  float se = 0.0f, sx = 0.0f;
  for (int64_t v = threadIdx.x; v < V; v += blockDim.x) {
    const float f = DT<T>::ld(x + v);
    se += __expf(f - mx);
    sx += f;
  }
  se = block_sum(se, red);
  sx = block_sum(sx, red);
  if (threadIdx.x == 0) {
    const float lse = mx + __logf(se);
    lse_out[row] = lse;
    const int64_t t = target[row];
    float l = 0.0f, nll = 0.0f;
    if (t != pad) {
      nll = lse - DT<T>::ld(x + t);
      const float smooth = (float)V * lse - sx;
      l = (1.0f - eps) * nll + (eps / (float)V) * smooth;
    }
    rows2[2 * row] = l;  // per-row terms; sum_pairs_kernel adds them in a fixed order (bit-reproducible loss, no atomics)
    rows2[2 * row + 1] = nll;
  }
}

// out[j] = sum_i v[i * stride + j], j < stride <= 2: ONE block, fixed order (strided per-thread partial sums in double, then a
// fixed LDS tree) — the run-to-run reproducible replacement of "atomicAdd a per-row term into a scalar"
__global__ __launch_bounds__(256) void sum_pairs_kernel(const float* v, int64_t n, int stride, float* out) {
  __shared__ double red[256];
  for (int j = 0; j < stride; ++j) {
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) a += (double)v[i * stride + j];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[j] = (float)red[0];
    __syncthreads();
  }
}

// dlogits[v] = g * ( p_v - (1-eps) * [v == t] - eps/V ),  p_v = exp(x_v - lse);  0 for pad rows
template <typename T>
__global__ __launch_bounds__(256) void ls_ce_bwd_kernel(const T* logits, const int64_t* target, const float* lse,
                                                        const float* gscale, T* dlogits, int64_t V, float eps, int64_t pad) {
  const int64_t row = blockIdx.x;
  const T* x = logits + row * V;
  T* d = dlogits + row * V;
  const int64_t t = target[row];
  const float g = t == pad ? 0.0f : gscale[0];
  const float l = lse[row];
  const float ev = eps / (float)V;
  for (int64_t v = threadIdx.x; v < V; v += blockDim.x) {
    float p = __expf(DT<T>::ld(x + v) - l) - ev;
    if (v == t) p -= (1.0f - eps);
    DT<T>::st(d + v, g * p);
  }
}

// deterministic two-stage sum of squares: per-block partials in a fixed layout, then ONE block adds them in a fixed order
// (replicas of a data-parallel job must compute bit-identical gradient norms, or their clip coefficients — and then their
// parameters — drift apart by ulps every step).
template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T* x, int64_t n, float* part) {
  __shared__ float red[4];
  float acc = 0.0f;
  const int64_t n8 = n / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    load8(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += v[e] * v[e];
  }
  if (blockIdx.x == 0)
    for (int64_t i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) { const float f = DT<T>::ld(x + i); acc += f * f; }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* part, int nparts, float* out) {
  __shared__ float red[4];
  float acc = 0.0f;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) out[0] += acc;
}

// One element's update — the arithmetic of optim/adam.py:146-226 with the bias correction folded into step_size.
// Contraction is pinned OFF here: every product and sum is rounded on its own, whatever loop shape the statement is compiled into —
// the vector loop, the scalar tail and a span of a sharded state (optim.FusedAdam.shard) must give one element the same bits.
__device__ __forceinline__ void adam_one(float g, float& mi, float& vi, float& p, float lr, float beta1, float beta2, float eps, float wd,
                                         float step_size) {
#pragma clang fp contract(off)
  mi = mi * beta1 + (1.0f - beta1) * g;
  vi = vi * beta2 + (1.0f - beta2) * g * g;
  if (wd != 0.0f) p -= wd * lr * p;                  // decoupled weight decay (optim/adam.py:216-219)
  p -= step_size * mi / (sqrtf(vi) + eps);           // (:206-214)
}

// 28 bytes per parameter in one pass.  Four consecutive elements per thread and iteration — 16-byte accesses to the three fp32 state
// arrays, 8 / 16 bytes to the gradient and the parameter (round 5: the one-element form had a 4-byte load per array and iteration in
// flight per thread and ran at 4.9 TB/s); the same operations per element in the same order: the same bits.  VEC4 needs the five
// base pointers 16-byte aligned (checked by the launcher); the tail (n % 4) and unaligned calls take the scalar loop.
template <typename TG, typename TP, bool VEC4>
__global__ __launch_bounds__(256) void adam_kernel(float* master, float* m, float* v, const TG* grad, TP* param, int64_t n, float lr, float beta1,
                                                    float beta2, float eps, float wd, float step_size, const float* grad_scale) {
  const float gs = grad_scale ? grad_scale[0] : 1.0f;
  const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  int64_t done = 0;
  if constexpr (VEC4) {
    const int64_t n4 = n / 4;
    for (int64_t q = tid0; q < n4; q += nth) {
      const int64_t i = 4 * q;
      f32x4 pm = *reinterpret_cast<const f32x4*>(master + i), mm = *reinterpret_cast<const f32x4*>(m + i), vv = *reinterpret_cast<const f32x4*>(v + i);
      float g[4];
      if constexpr (sizeof(TG) == 2) {
        const uint2 raw = *reinterpret_cast<const uint2*>(grad + i);
        g[0] = __uint_as_float(raw.x << 16); g[1] = __uint_as_float(raw.x & 0xffff0000u);
        g[2] = __uint_as_float(raw.y << 16); g[3] = __uint_as_float(raw.y & 0xffff0000u);
      } else {
        const f32x4 raw = *reinterpret_cast<const f32x4*>(grad + i);
        g[0] = raw[0]; g[1] = raw[1]; g[2] = raw[2]; g[3] = raw[3];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float mi = mm[e], vi = vv[e], pp = pm[e];
        adam_one(g[e] * gs, mi, vi, pp, lr, beta1, beta2, eps, wd, step_size);
        mm[e] = mi; vv[e] = vi; pm[e] = pp;
      }
      *reinterpret_cast<f32x4*>(m + i) = mm;
      *reinterpret_cast<f32x4*>(v + i) = vv;
      *reinterpret_cast<f32x4*>(master + i) = pm;
      if constexpr (sizeof(TP) == 2) {
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = static_cast<__bf16>(pm[e]);
        *reinterpret_cast<uint2*>(param + i) = __builtin_bit_cast(uint2, o);
      } else {
        *reinterpret_cast<f32x4*>(param + i) = pm;
      }
    }
    done = n4 * 4;
  }
  for (int64_t i = done + tid0; i < n; i += nth) {
    float mi = m[i], vi = v[i], p = master[i];
    adam_one(DT<TG>::ld(grad + i) * gs, mi, vi, p, lr, beta1, beta2, eps, wd, step_size);
    m[i] = mi;
    v[i] = vi;
    master[i] = p;
    DT<TP>::st(param + i, p);
  }
}

// ---------------------------------------------------------------------------------------
// Contrastive term of the triplet criterion (criterions/triplet_st_mt_contrastive.py:154-169): per utterance
//   c[i][j] = cos(a[i,:], t[j,:]) (audio slot i, text slot j), logits L = c / temp, class dim = AUDIO slot, target(j) = j:
//   loss = sum_b sum_j ( logsumexp_i L[i][j] - L[j][j] ).
// One workgroup per utterance (M <= 64 memory slots, C channels); 8 KFLOP-scale work: plain FMA, LDS tiles of 32 channels.
// ---------------------------------------------------------------------------------------
constexpr int CM = 64, CK = 32;

template <typename T>
__global__ __launch_bounds__(256) void contrastive_fwd_kernel(const T* a, const T* t, float* loss_rows, float* sim, float* na, float* nt,
                                                              int M, int C, float inv_temp) {
  __shared__ float sA[CM][CK + 1], sT[CM][CK + 1], sS[CM][CM + 1], red[4];
  const int b = blockIdx.x, tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const T* ab = a + (int64_t)b * M * C;
  const T* tb = t + (int64_t)b * M * C;
  float acc[4][4] = {}, qa[4] = {}, qt[4] = {};
  for (int k0 = 0; k0 < C; k0 += CK) {
    for (int e = tid; e < CM * CK; e += 256) {
      const int r = e / CK, c = e % CK;
      const bool ok = r < M && k0 + c < C;
      sA[r][c] = ok ? DT<T>::ld(ab + (int64_t)r * C + k0 + c) : 0.0f;
      sT[r][c] = ok ? DT<T>::ld(tb + (int64_t)r * C + k0 + c) : 0.0f;
    }
    __syncthreads();
#pragma unroll 4
    for (int c = 0; c < CK; ++c) {
      float x[4], y[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { x[u] = sA[ty * 4 + u][c]; y[u] = sT[tx * 4 + u][c]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        qa[u] = fmaf(x[u], x[u], qa[u]);
        qt[u] = fmaf(y[u], y[u], qt[u]);
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = fmaf(x[u], y[v], acc[u][v]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = ty * 4 + u, j = tx * 4 + v;
      const float c = acc[u][v] / fmaxf(sqrtf(qa[u]) * sqrtf(qt[v]), 1e-8f);  // torch.cosine_similarity, eps = 1e-8
      sS[i][j] = c;
      if (i < M && j < M) sim[((int64_t)b * M + i) * M + j] = c;
    }
  if (tx == 0)
    for (int u = 0; u < 4; ++u) if (ty * 4 + u < M) na[(int64_t)b * M + ty * 4 + u] = sqrtf(qa[u]);
  if (ty == 0)
    for (int v = 0; v < 4; ++v) if (tx * 4 + v < M) nt[(int64_t)b * M + tx * 4 + v] = sqrtf(qt[v]);
  __syncthreads();
  float l = 0.0f;
  if (tid < M) {  // text slot j = tid: log-sum-exp over the audio slots
    float mx = -INFINITY;
    for (int i = 0; i < M; ++i) mx = fmaxf(mx, sS[i][tid] * inv_temp);
    float se = 0.0f;
    for (int i = 0; i < M; ++i) se += __expf(sS[i][tid] * inv_temp - mx);
    l = mx + __logf(se) - sS[tid][tid] * inv_temp;
  }
  l = block_sum(l, red);
  if (tid == 0) loss_rows[b] = l;  // per-utterance term; summed in a fixed order by sum_pairs_kernel
}

template <typename T>
__global__ __launch_bounds__(256) void contrastive_bwd_kernel(const T* a, const T* t, const float* sim, const float* na, const float* nt,
                                                              const float* gscale, T* da, T* dt, int M, int C, float inv_temp) {
  __shared__ float sW[CM][CM + 1], sRa[CM], sRt[CM], sNa[CM], sNt[CM];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float g = gscale[0] * inv_temp;
  const float* S = sim + (int64_t)b * M * M;
  if (tid < M) { sNa[tid] = fmaxf(na[(int64_t)b * M + tid], 1e-12f); sNt[tid] = fmaxf(nt[(int64_t)b * M + tid], 1e-12f); }
  __syncthreads();
  if (tid < M) {  // column j = tid: G[i][j] = g * (softmax_i(L[:, j]) - [i == j]);  W = G / (|a_i| |t_j|);  r_t[j] = sum_i G c
    const int j = tid;
    float mx = -INFINITY;
    for (int i = 0; i < M; ++i) mx = fmaxf(mx, S[i * M + j] * inv_temp);
    float se = 0.0f;
    for (int i = 0; i < M; ++i) se += __expf(S[i * M + j] * inv_temp - mx);
    float rt = 0.0f;
    for (int i = 0; i < M; ++i) {
      const float c = S[i * M + j];
      const float G = g * (__expf(c * inv_temp - mx) / se - (i == j ? 1.0f : 0.0f));
      rt += G * c;
      sW[i][j] = G;
    }
    sRt[j] = rt;
  }
  __syncthreads();
  if (tid < M) {  // row i = tid: r_a[i] = sum_j G c
    float ra = 0.0f;
    for (int j = 0; j < M; ++j) ra += sW[tid][j] * S[tid * M + j];
    sRa[tid] = ra;
  }
  __syncthreads();
  for (int e = tid; e < M * M; e += 256) { const int i = e / M, j = e % M; sW[i][j] /= fmaxf(sNa[i] * sNt[j], 1e-8f); }
  __syncthreads();
  // this workgroup's 64-channel slice of a and t -> LDS; 4 x 4 register blocks of da / dt
  __shared__ float sA[CM][CM + 1], sT[CM][CM + 1];
  const int d0 = blockIdx.y * CM;
  const T* ab = a + (int64_t)b * M * C;
  const T* tb = t + (int64_t)b * M * C;
  for (int e = tid; e < CM * CM; e += 256) {
    const int r = e / CM, c = e % CM;
    const bool ok = r < M && d0 + c < C;
    sA[r][c] = ok ? DT<T>::ld(ab + (int64_t)r * C + d0 + c) : 0.0f;
    sT[r][c] = ok ? DT<T>::ld(tb + (int64_t)r * C + d0 + c) : 0.0f;
  }
  for (int e = tid; e < CM * CM; e += 256) { const int i = e / CM, j = e % CM; if (i >= M || j >= M) sW[i][j] = 0.0f; }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  float xa[4][4] = {}, xt[4][4] = {};
  for (int j = 0; j < M; ++j) {
    float w[4], wt[4], tv[4], av[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { w[u] = sW[ty * 4 + u][j]; wt[u] = sW[j][ty * 4 + u]; tv[u] = sT[j][tx * 4 + u]; av[u] = sA[j][tx * 4 + u]; }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        xa[u][v] = fmaf(w[u], tv[v], xa[u][v]);    // da[i][d] += W[i][j] t[j][d]
        xt[u][v] = fmaf(wt[u], av[v], xt[u][v]);   // dt[i'][d] += W[j][i'] a[j][d]   (i' = text slot)
      }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = ty * 4 + u;
    if (r >= M) continue;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int d = d0 + tx * 4 + v;
      if (d >= C) continue;
      DT<T>::st(da + ((int64_t)b * M + r) * C + d, xa[u][v] - sRa[r] * sA[r][tx * 4 + v] / (sNa[r] * sNa[r]));
      DT<T>::st(dt + ((int64_t)b * M + r) * C + d, xt[u][v] - sRt[r] * sT[r][tx * 4 + v] / (sNt[r] * sNt[r]));
    }
  }
}

}  // namespace

extern "C" int cst_ls_ce_fwd(const void* logits, const int64_t* target, float* out2, float* lse, float* row_ws, int64_t rows, int64_t V,
                             float eps, int64_t pad_idx, int dtype, cst_stream stream) {
  CST_REQUIRE(logits && target && out2 && lse && row_ws && rows > 0 && V > 0, "cst_ls_ce_fwd: bad args");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_ls_ce_fwd: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_LOSS, s, 0.0, (double)rows * V * cst_dtype_size(dtype));
  if (dtype == CST_BF16) hipLaunchKernelGGL(ls_ce_fwd_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, s, (const bf16_t*)logits, target, row_ws, lse, V, eps, pad_idx);
  else hipLaunchKernelGGL(ls_ce_fwd_kernel<float>, dim3((unsigned)rows), dim3(256), 0, s, (const float*)logits, target, row_ws, lse, V, eps, pad_idx);
  hipLaunchKernelGGL(sum_pairs_kernel, dim3(1), dim3(256), 0, s, row_ws, rows, 2, out2);
  return cst_check_launch("cst_ls_ce_fwd");
}

extern "C" int cst_ls_ce_bwd(const void* logits, const int64_t* target, const float* lse, const float* gscale, void* dlogits,
                             int64_t rows, int64_t V, float eps, int64_t pad_idx, int dtype, cst_stream stream) {
  CST_REQUIRE(logits && target && lse && gscale && dlogits && rows > 0 && V > 0, "cst_ls_ce_bwd: bad args");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_ls_ce_bwd: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_LOSS, s, 0.0, 2.0 * rows * V * cst_dtype_size(dtype));
  if (dtype == CST_BF16) hipLaunchKernelGGL(ls_ce_bwd_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, s, (const bf16_t*)logits, target, lse, gscale, (bf16_t*)dlogits, V, eps, pad_idx);
  else hipLaunchKernelGGL(ls_ce_bwd_kernel<float>, dim3((unsigned)rows), dim3(256), 0, s, (const float*)logits, target, lse, gscale, (float*)dlogits, V, eps, pad_idx);
  return cst_check_launch("cst_ls_ce_bwd");
}

extern "C" int64_t cst_sumsq_workspace(void) { return 2048 * (int64_t)sizeof(float); }

extern "C" int cst_sumsq(const void* x, int64_t n, float* out, float* workspace, int dtype, cst_stream stream) {
  CST_REQUIRE(x && out && workspace && n > 0, "cst_sumsq: bad args");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_sumsq: bad dtype");
  CST_REQUIRE((uintptr_t)x % 16 == 0, "cst_sumsq: x must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_OPTIM, s, 0.0, (double)n * cst_dtype_size(dtype));
  int blocks = (int)(cst_ceil_div(n, 256 * 8) < 2048 ? cst_ceil_div(n, 256 * 8) : 2048);
  if (dtype == CST_BF16) hipLaunchKernelGGL(sumsq_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, n, workspace);
  else hipLaunchKernelGGL(sumsq_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, n, workspace);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, blocks, out);
  return cst_check_launch("cst_sumsq");
}

extern "C" int cst_adam_step(float* master, float* exp_avg, float* exp_avg_sq, const void* grad, void* model_param, int64_t n,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                             const float* grad_scale, int grad_dtype, int param_dtype, cst_stream stream) {
  CST_REQUIRE(master && exp_avg && exp_avg_sq && grad && model_param && n > 0 && step >= 1, "cst_adam_step: bad args");
  hipStream_t s = (hipStream_t)stream;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr * sqrt(bc2) / bc1);
  CstProfScope prof(CST_K_OPTIM, s, 0.0, (double)n * (24.0 + cst_dtype_size(grad_dtype) + cst_dtype_size(param_dtype)));
  int blocks = (int)(cst_ceil_div(n, 256) < 4096 ? cst_ceil_div(n, 256) : 4096);
  const bool vec4 = (((uintptr_t)master | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)grad | (uintptr_t)model_param) & 15) == 0 && n >= 4;
  if (vec4) blocks = (int)(cst_ceil_div(n / 4, 256) < 4096 ? cst_ceil_div(n / 4, 256) : 4096);
#define CST_ADAM(TG, TP)                                                                                                                  \
  do {                                                                                                                                    \
    if (vec4) hipLaunchKernelGGL((adam_kernel<TG, TP, true>), dim3(blocks), dim3(256), 0, s, master, exp_avg, exp_avg_sq, (const TG*)grad, (TP*)model_param, n, lr, beta1, beta2, eps, weight_decay, step_size, grad_scale); \
    else hipLaunchKernelGGL((adam_kernel<TG, TP, false>), dim3(blocks), dim3(256), 0, s, master, exp_avg, exp_avg_sq, (const TG*)grad, (TP*)model_param, n, lr, beta1, beta2, eps, weight_decay, step_size, grad_scale); \
  } while (0)
  if (grad_dtype == CST_BF16 && param_dtype == CST_BF16) CST_ADAM(bf16_t, bf16_t);
  else if (grad_dtype == CST_F32 && param_dtype == CST_F32) CST_ADAM(float, float);
  else if (grad_dtype == CST_F32 && param_dtype == CST_BF16) CST_ADAM(float, bf16_t);
  else if (grad_dtype == CST_BF16 && param_dtype == CST_F32) CST_ADAM(bf16_t, float);
  else CST_REQUIRE(false, "cst_adam_step: bad dtypes %d/%d", grad_dtype, param_dtype);
#undef CST_ADAM
  return cst_check_launch("cst_adam_step");
}

extern "C" int cst_contrastive_fwd(const void* a, const void* t, float* loss, float* loss_rows, float* sim, float* na, float* nt, int64_t B, int64_t M,
                                   int64_t C, float temp, int dtype, cst_stream stream) {
  CST_REQUIRE(a && t && loss && loss_rows && sim && na && nt && B > 0 && M > 0 && M <= 64 && C > 0 && temp > 0.0f, "cst_contrastive_fwd: bad args (M <= 64)");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_contrastive_fwd: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_LOSS, s, 2.0 * B * M * M * C, 2.0 * B * M * C * cst_dtype_size(dtype));
  if (dtype == CST_BF16) hipLaunchKernelGGL(contrastive_fwd_kernel<bf16_t>, dim3((unsigned)B), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)t, loss_rows, sim, na, nt, (int)M, (int)C, 1.0f / temp);
  else hipLaunchKernelGGL(contrastive_fwd_kernel<float>, dim3((unsigned)B), dim3(256), 0, s, (const float*)a, (const float*)t, loss_rows, sim, na, nt, (int)M, (int)C, 1.0f / temp);
  hipLaunchKernelGGL(sum_pairs_kernel, dim3(1), dim3(256), 0, s, loss_rows, B, 1, loss);
  return cst_check_launch("cst_contrastive_fwd");
}

extern "C" int cst_contrastive_bwd(const void* a, const void* t, const float* sim, const float* na, const float* nt, const float* gscale,
                                   void* da, void* dt, int64_t B, int64_t M, int64_t C, float temp, int dtype, cst_stream stream) {
  CST_REQUIRE(a && t && sim && na && nt && gscale && da && dt && B > 0 && M > 0 && M <= 64 && C > 0 && temp > 0.0f, "cst_contrastive_bwd: bad args");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_contrastive_bwd: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_LOSS, s, 4.0 * B * M * M * C, 4.0 * B * M * C * cst_dtype_size(dtype));
  if (dtype == CST_BF16) hipLaunchKernelGGL(contrastive_bwd_kernel<bf16_t>, dim3((unsigned)B, (unsigned)cst_ceil_div(C, 64)), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)t, sim, na, nt, gscale, (bf16_t*)da, (bf16_t*)dt, (int)M, (int)C, 1.0f / temp);
  else hipLaunchKernelGGL(contrastive_bwd_kernel<float>, dim3((unsigned)B, (unsigned)cst_ceil_div(C, 64)), dim3(256), 0, s, (const float*)a, (const float*)t, sim, na, nt, gscale, (float*)da, (float*)dt, (int)M, (int)C, 1.0f / temp);
  return cst_check_launch("cst_contrastive_bwd");
}
