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
__global__ __launch_bounds__(256) void ls_ce_fwd_kernel(const T* logits, const int64_t* target, float* out2, float* lse_out,
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
    if (t != pad) {
      const float nll = lse - DT<T>::ld(x + t);
      const float smooth = (float)V * lse - sx;
      atomicAdd(out2, (1.0f - eps) * nll + (eps / (float)V) * smooth);
      atomicAdd(out2 + 1, nll);
    }
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

template <typename TG, typename TP>
__global__ void adam_kernel(float* master, float* m, float* v, const TG* grad, TP* param, int64_t n, float lr, float beta1,
                            float beta2, float eps, float wd, float step_size, const float* grad_scale) {
  const float gs = grad_scale ? grad_scale[0] : 1.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float g = DT<TG>::ld(grad + i) * gs;
    const float mi = m[i] * beta1 + (1.0f - beta1) * g;
    const float vi = v[i] * beta2 + (1.0f - beta2) * g * g;
    float p = master[i];
    if (wd != 0.0f) p -= wd * lr * p;                  // decoupled weight decay (optim/adam.py:216-219)
    p -= step_size * mi / (sqrtf(vi) + eps);           // bias correction folded into step_size (:206-214)
    m[i] = mi;
    v[i] = vi;
    master[i] = p;
    DT<TP>::st(param + i, p);
  }
}

}  // namespace

extern "C" int cst_ls_ce_fwd(const void* logits, const int64_t* target, float* out2, float* lse, int64_t rows, int64_t V,
                             float eps, int64_t pad_idx, int dtype, cst_stream stream) {
  CST_REQUIRE(logits && target && out2 && lse && rows > 0 && V > 0, "cst_ls_ce_fwd: bad args");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_ls_ce_fwd: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_LOSS, s, 0.0, (double)rows * V * cst_dtype_size(dtype));
  if (dtype == CST_BF16) hipLaunchKernelGGL(ls_ce_fwd_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, s, (const bf16_t*)logits, target, out2, lse, V, eps, pad_idx);
  else hipLaunchKernelGGL(ls_ce_fwd_kernel<float>, dim3((unsigned)rows), dim3(256), 0, s, (const float*)logits, target, out2, lse, V, eps, pad_idx);
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
#define CST_ADAM(TG, TP) hipLaunchKernelGGL((adam_kernel<TG, TP>), dim3(blocks), dim3(256), 0, s, master, exp_avg, exp_avg_sq, (const TG*)grad, (TP*)model_param, n, lr, beta1, beta2, eps, weight_decay, step_size, grad_scale)
  if (grad_dtype == CST_BF16 && param_dtype == CST_BF16) CST_ADAM(bf16_t, bf16_t);
  else if (grad_dtype == CST_F32 && param_dtype == CST_F32) CST_ADAM(float, float);
  else if (grad_dtype == CST_F32 && param_dtype == CST_BF16) CST_ADAM(float, bf16_t);
  else if (grad_dtype == CST_BF16 && param_dtype == CST_F32) CST_ADAM(bf16_t, float);
  else CST_REQUIRE(false, "cst_adam_step: bad dtypes %d/%d", grad_dtype, param_dtype);
#undef CST_ADAM
  return cst_check_launch("cst_adam_step");
}
