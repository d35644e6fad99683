// layernorm.hip — LayerNorm forward/backward for gfx950 (HBM-bound; one wave64 per row,
// 16-byte vector loads, wave-level shuffles for the row statistics, fp32 statistics).
// Replaces torch.nn.LayerNorm / apex FusedLayerNorm at fairseq/modules/layer_norm.py:11-35.
// Fusions: optional residual add in front (s = x + res, optionally written back) and an optional
// extra upstream gradient added to dx in backward (the residual branch), so a pre-/post-norm
// Transformer block needs no separate add kernels.
#include "cst_common.h"

namespace {

constexpr int LN_MAXV = 4;       // 8-element vectors per lane -> cols <= 2048
constexpr int LN_WAVES = 4;      // rows per block iteration
// The kernels are instantiated per NV = vectors per lane actually needed (cols <= 512: 1, <= 1024: 2, else 4): with the generic 4 the
// backward kernel allocates 160 VGPRs (3 waves per SIMD) for the 768-column rows of wav2vec2 that need 2 vectors (100 VGPRs, 5 waves
// per SIMD) — a row is only 1.5 KB, so the bytes in flight, i.e. the HBM rate, scale with the resident waves.

// 8 elements of a row in storage form
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  u32x4 r;
  __device__ __forceinline__ void load(const bf16_t* p) { r = *reinterpret_cast<const u32x4*>(p); }
  __device__ __forceinline__ void unpack(float (&v)[8]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
  }
};
template <> struct Raw8<float> {
  float f[8];
  __device__ __forceinline__ void load(const float* p) { load8(p, f); }
  __device__ __forceinline__ void unpack(float (&v)[8]) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f[i];
  }
};

template <typename T, int NV>
__global__ __launch_bounds__(LN_WAVES * 64) void ln_fwd_kernel(const T* x, const T* res, const T* gamma, const T* beta,
                                                              T* y, T* sum_out, float* mean, float* rstd,
                                                              int64_t rows, int cols, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = cols / 8;
  // gamma / beta stay in registers in their storage form for the whole grid-stride loop; the NEXT row's loads are issued before this
  // row's statistics (a row is 1-2 KB: one row per wave in flight left the memory pipe idle during the two wave reductions)
  Raw8<T> gq[NV], bq[NV], nx[NV], nr[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < nvec) { gq[i].load(gamma + (lane + 64 * i) * 8); bq[i].load(beta + (lane + 64 * i) * 8); }
  const int64_t row0 = (int64_t)blockIdx.x * LN_WAVES + wave, rstep = (int64_t)gridDim.x * LN_WAVES;
  if (row0 < rows) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nvec) {
        nx[i].load(x + row0 * cols + (lane + 64 * i) * 8);
        if (res) nr[i].load(res + row0 * cols + (lane + 64 * i) * 8);
      }
  }
  for (int64_t row = row0; row < rows; row += rstep) {
    float v[NV][8];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + 64 * i;
      if (vi < nvec) {
        nx[i].unpack(v[i]);
        if (res) {
          float r[8];
          nr[i].unpack(r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[i][e] += r[e];
        }
      }
    }
    if (row + rstep < rows) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < nvec) {
          nx[i].load(x + (row + rstep) * cols + (lane + 64 * i) * 8);
          if (res) nr[i].load(res + (row + rstep) * cols + (lane + 64 * i) * 8);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + 64 * i;
      if (vi < nvec) {
        if (sum_out) store8(sum_out + row * cols + vi * 8, v[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
      }
    }
    const float mu = wave_sum(s) / (float)cols;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mu; q += d * d; }
      }
    const float rs = rsqrtf(wave_sum(q) / (float)cols + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + 64 * i;
      if (vi < nvec) {
        float g[8], b[8], o[8];
        gq[i].unpack(g);
        bq[i].unpack(b);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mu) * rs * g[e] + b[e];
        store8(y + row * cols + vi * 8, o);
      }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma;  partial dgamma/dbeta per block.
template <typename T, int NV>
__global__ __launch_bounds__(LN_WAVES * 64, (NV <= 2 ? 4 : 2)) void ln_bwd_kernel(const T* dy, const T* s, const T* gamma, const float* mean,
                                                              const float* rstd, const T* dres, T* dx, float* part,
                                                              int64_t rows, int cols, uint32_t* tile_live, uint32_t epoch) {
  __shared__ float red[LN_WAVES][64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = cols / 8;
  float dg[NV][8], db[NV][8], gm[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg[i][e] = 0.0f; db[i][e] = 0.0f; gm[i][e] = 0.0f; }
    if (lane + 64 * i < nvec) load8(gamma + (lane + 64 * i) * 8, gm[i]);
  }
  for (int64_t row = (int64_t)blockIdx.x * LN_WAVES + wave; row < rows; row += (int64_t)gridDim.x * LN_WAVES) {
    const float mu = mean[row], rs = rstd[row];
    // the row is kept in its STORAGE form between the two passes (4 VGPRs per 8 bf16 elements instead of 8 + 8 floats of g and
    // xhat) and unpacked twice: 16-32 fewer live registers buy a fourth wave per SIMD
    Raw8<T> rd[NV], rx[NV];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + 64 * i;
      if (vi < nvec) {
        rd[i].load(dy + row * cols + vi * 8);
        rx[i].load(s + row * cols + vi * 8);
        float d[8], xs[8];
        rd[i].unpack(d);
        rx[i].unpack(xs);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (xs[e] - mu) * rs;
          const float g = d[e] * gm[i][e];
          s1 += g;
          s2 += g * xh;
          dg[i][e] += d[e] * xh;
          db[i][e] += d[e];
        }
      }
    }
    s1 = wave_sum(s1) / (float)cols;
    s2 = wave_sum(s2) / (float)cols;
    bool nz = false;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + 64 * i;
      if (vi < nvec) {
        float d[8], xs[8], o[8];
        rd[i].unpack(d);
        rx[i].unpack(xs);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rs * (d[e] * gm[i][e] - s1 - (xs[e] - mu) * rs * s2);
        if (dres) {
          float r[8];
          load8(dres + row * cols + vi * 8, r);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += r[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) nz |= (o[e] != 0.0f);
        store8(dx + row * cols + vi * 8, o);
      }
    }
    // a row with any non-zero dx marks its 64-row tile live (same value from every writer: a benign race); rows that are exactly
    // zero in fp32 are exactly zero as stored, so an unstamped tile may be skipped by the GEMMs that reduce over rows
    if (tile_live && __builtin_amdgcn_ballot_w64(nz) != 0 && lane == 0) tile_live[row >> 6] = epoch;
  }
  // block reduction of the per-wave partials, one vector slot at a time
  float* pg = part + (int64_t)blockIdx.x * 2 * cols;
  float* pb = pg + cols;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = pass ? db[i][e] : dg[i][e];
      __syncthreads();
      if (wave == 0 && lane + 64 * i < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float a = 0.0f;
#pragma unroll
          for (int w = 0; w < LN_WAVES; ++w) a += red[w][lane * 8 + e];
          (pass ? pb : pg)[(lane + 64 * i) * 8 + e] = a;
        }
      }
    }
  }
}

// partial [nblocks][2][cols] -> dgamma/dbeta: block = 16 columns x 64 row-groups (48 blocks at 768 columns instead of 12: the
// pass is latency-bound — 1024 partial rows, two loads each — so more, shorter chains win over wider coalescing)
template <typename TO>
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* part, TO* dgamma, TO* dbeta, int nblocks, int cols) {
  __shared__ float ra[64][17], rb[64][17];
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float a = 0.0f, b = 0.0f;
  if (c < cols)
    for (int i = g; i < nblocks; i += 64) {
      a += part[(int64_t)i * 2 * cols + c];
      b += part[(int64_t)i * 2 * cols + cols + c];
    }
  ra[g][cl] = a;
  rb[g][cl] = b;
  __syncthreads();
  if (g == 0 && c < cols) {
#pragma unroll 8
    for (int k = 1; k < 64; ++k) { a += ra[k][cl]; b += rb[k][cl]; }
    DT<TO>::st(dgamma + c, a);  // written in the parameter dtype: no conversion launches after the backward
    DT<TO>::st(dbeta + c, b);
  }
}

int ln_blocks(int64_t rows, int cap = 1024) {
  int64_t b = cst_ceil_div(rows, LN_WAVES);
  return (int)(b < cap ? b : cap);
}
inline int ln_nv(int64_t cols) { return cols <= 512 ? 1 : (cols <= 1024 ? 2 : 4); }

}  // namespace

extern "C" int cst_layernorm_fwd(const void* x, const void* res, const void* gamma, const void* beta, void* y,
                                 void* sum_out, float* mean, float* rstd, int64_t rows, int64_t cols, float eps,
                                 int dtype, cst_stream stream) {
  CST_REQUIRE(x && gamma && beta && y && mean && rstd, "cst_layernorm_fwd: null tensor");
  CST_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 8 * 64 * LN_MAXV, "cst_layernorm_fwd: cols=%lld must be a multiple of 8 and <= %d", (long long)cols, 8 * 64 * LN_MAXV);
  hipStream_t s = (hipStream_t)stream;
  const double bytes = (double)rows * cols * cst_dtype_size(dtype) * (2.0 + (res ? 1.0 : 0.0) + (sum_out ? 1.0 : 0.0));
  CstProfScope prof(CST_K_LAYERNORM, s, 0.0, bytes);
  dim3 grid(ln_blocks(rows, 2048));  // 8 blocks of 4 waves per CU: the forward kernel needs < 64 VGPRs
  CST_REQUIRE(dtype == CST_BF16 || dtype == CST_F32, "cst_layernorm_fwd: bad dtype %d", dtype);
#define CST_LNF(T, NVV) hipLaunchKernelGGL((ln_fwd_kernel<T, NVV>), grid, dim3(LN_WAVES * 64), 0, s, (const T*)x, (const T*)res, (const T*)gamma, (const T*)beta, (T*)y, (T*)sum_out, mean, rstd, rows, (int)cols, eps)
#define CST_LNF_NV(T) do { const int nv = ln_nv(cols); if (nv == 1) CST_LNF(T, 1); else if (nv == 2) CST_LNF(T, 2); else CST_LNF(T, 4); } while (0)
  if (dtype == CST_BF16) CST_LNF_NV(bf16_t);
  else CST_LNF_NV(float);
#undef CST_LNF_NV
#undef CST_LNF
  return cst_check_launch("cst_layernorm_fwd");
}

constexpr int LN_BWD_BLOCKS = 1024;  // 4 blocks of 4 waves per CU at <= 128 VGPRs (NV <= 2)
extern "C" int64_t cst_layernorm_bwd_workspace(int64_t rows, int64_t cols) {
  return (int64_t)ln_blocks(rows, LN_BWD_BLOCKS) * 2 * cols * (int64_t)sizeof(float);
}

static int layernorm_bwd_impl(const void* dy, const void* sx, const void* gamma, const float* mean, const float* rstd,
                              const void* dres, void* dx, void* dgamma, void* dbeta, void* workspace, int64_t rows,
                              int64_t cols, int dtype, int grad_dtype, uint32_t* tile_live, uint32_t epoch, cst_stream stream) {
  CST_REQUIRE(dy && sx && gamma && mean && rstd && dx && workspace && ((dgamma != nullptr) == (dbeta != nullptr)), "cst_layernorm_bwd: null tensor");
  CST_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 8 * 64 * LN_MAXV, "cst_layernorm_bwd: cols=%lld must be a multiple of 8 and <= %d", (long long)cols, 8 * 64 * LN_MAXV);
  hipStream_t s = (hipStream_t)stream;
  const double bytes = (double)rows * cols * cst_dtype_size(dtype) * (3.0 + (dres ? 1.0 : 0.0));
  CstProfScope prof(CST_K_LAYERNORM, s, 0.0, bytes);
  const int nb = ln_blocks(rows, LN_BWD_BLOCKS);
  CST_REQUIRE(dtype == CST_BF16 || dtype == CST_F32, "cst_layernorm_bwd: bad dtype %d", dtype);
#define CST_LNB(T, NVV) hipLaunchKernelGGL((ln_bwd_kernel<T, NVV>), dim3(nb), dim3(LN_WAVES * 64), 0, s, (const T*)dy, (const T*)sx, (const T*)gamma, mean, rstd, (const T*)dres, (T*)dx, (float*)workspace, rows, (int)cols, tile_live, epoch)
#define CST_LNB_NV(T) do { const int nv = ln_nv(cols); if (nv == 1) CST_LNB(T, 1); else if (nv == 2) CST_LNB(T, 2); else CST_LNB(T, 4); } while (0)
  if (dtype == CST_BF16) CST_LNB_NV(bf16_t);
  else CST_LNB_NV(float);
#undef CST_LNB_NV
#undef CST_LNB
  int rc = cst_check_launch("cst_layernorm_bwd");
  if (rc != CST_OK) return rc;
  CST_REQUIRE(grad_dtype == CST_F32 || grad_dtype == dtype, "cst_layernorm_bwd: grad_dtype must be f32 or dtype");
  if (!dgamma) return rc;  // deferred: the row-block partials [blocks][2][cols] stay in `workspace` for cst_reduce_multi (include/cst.h)
  if (grad_dtype == CST_BF16)
    hipLaunchKernelGGL(ln_bwd_reduce_kernel<bf16_t>, dim3((unsigned)cst_ceil_div(cols, 16)), dim3(1024), 0, s, (const float*)workspace, (bf16_t*)dgamma, (bf16_t*)dbeta, nb, (int)cols);
  else
    hipLaunchKernelGGL(ln_bwd_reduce_kernel<float>, dim3((unsigned)cst_ceil_div(cols, 16)), dim3(1024), 0, s, (const float*)workspace, (float*)dgamma, (float*)dbeta, nb, (int)cols);
  return cst_check_launch("cst_layernorm_bwd reduce");
}

extern "C" int cst_layernorm_bwd(const void* dy, const void* sx, const void* gamma, const float* mean, const float* rstd,
                                 const void* dres, void* dx, void* dgamma, void* dbeta, void* workspace, int64_t rows,
                                 int64_t cols, int dtype, int grad_dtype, cst_stream stream) {
  return layernorm_bwd_impl(dy, sx, gamma, mean, rstd, dres, dx, dgamma, dbeta, workspace, rows, cols, dtype, grad_dtype, nullptr, 0, stream);
}

extern "C" int cst_layernorm_bwd_tiles(const void* dy, const void* sx, const void* gamma, const float* mean, const float* rstd,
                                       const void* dres, void* dx, void* dgamma, void* dbeta, void* workspace, int64_t rows,
                                       int64_t cols, int dtype, int grad_dtype, uint32_t* tile_live, uint32_t epoch, cst_stream stream) {
  CST_REQUIRE(tile_live && epoch != 0, "cst_layernorm_bwd_tiles: tile_live / non-zero epoch required");
  return layernorm_bwd_impl(dy, sx, gamma, mean, rstd, dres, dx, dgamma, dbeta, workspace, rows, cols, dtype, grad_dtype, tile_live, epoch, stream);
}
