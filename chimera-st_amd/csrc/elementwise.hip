// elementwise.hip — HBM-bound helpers: GLU, activation fwd/bwd, column sums (bias grads),
// col2im for strided conv1d input gradients, row masking.  All use 16-byte (8-element) vectors
// and grid-stride loops capped at ~2048 workgroups (cdna guide G11/G13).
#include "cst_common.h"

namespace {

inline int ew_blocks(int64_t items) {
  int64_t b = cst_ceil_div(items, 256);
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

template <typename T>
__global__ void glu_fwd_kernel(const T* z, T* y, int64_t rows, int64_t C) {
  const int64_t cv = C / 8, total = rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv, c = (i % cv) * 8;
    float a[8], g[8], o[8];
    load8(z + r * 2 * C + c, a);
    load8(z + r * 2 * C + C + c, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = a[e] / (1.0f + __expf(-g[e]));
    store8(y + r * C + c, o);
  }
}

template <typename T>
__global__ void glu_bwd_kernel(const T* dy, const T* z, T* dz, int64_t rows, int64_t C) {
  const int64_t cv = C / 8, total = rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv, c = (i % cv) * 8;
    float a[8], g[8], d[8], da[8], dg[8];
    load8(z + r * 2 * C + c, a);
    load8(z + r * 2 * C + C + c, g);
    load8(dy + r * C + c, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float sg = 1.0f / (1.0f + __expf(-g[e]));
      da[e] = d[e] * sg;
      dg[e] = d[e] * a[e] * sg * (1.0f - sg);
    }
    store8(dz + r * 2 * C + c, da);
    store8(dz + r * 2 * C + C + c, dg);
  }
}

template <typename T>
__global__ void act_bwd_kernel(const T* dy, const T* z, T* dx, int64_t n8, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float d[8], zz[8];
    load8(dy + i * 8, d);
    load8(z + i * 8, zz);
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] *= dact_t<T>(zz[e], act);
    store8(dx + i * 8, d);
  }
}

template <typename T>
__global__ void act_fwd_kernel(const T* x, T* y, int64_t n8, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    load8(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = act_t<T>(v[e], act);
    store8(y + i * 8, v);
  }
}

// block = 256 threads = 16 row-lanes x 16 column vectors (128 columns): each half-wave reads a 256-B row segment, 4 row
// loads in flight per thread; LDS combine over the 16 row-lanes; one fp32 atomic per column per block (<= ~200 blocks per
// column, so the atomics are not the bottleneck).
// DROP: x is first multiplied by its dropout factors (FairseqDropout backward: the same mask as forward, element index = r * cols + c),
// the masked rows are written to xd in the storage dtype, and the column sums are taken over those STORED values — the same bits as
// cst_dropout followed by cst_colsum_typed, in one pass over the gradient instead of three.
template <typename T, bool PART = false, bool DROP = false>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, int64_t ldx, float* out, int64_t rows, int64_t cols, int64_t rows_per,
                                                     const uint32_t* row_live = nullptr, uint32_t epoch = 0, T* xd = nullptr,
                                                     uint32_t dkey = 0, uint32_t dthr = 0, float dscale = 1.0f) {
  __shared__ float red[16][16 * 8 + 1];
  auto masked = [&](float (&v)[8], int64_t r, int64_t c0) {  // mask, round to storage, store; v returns the stored values
    cst_drop8(v, dkey, (uint64_t)(r * cols + c0), dthr, dscale);
    store8(xd + r * ldx + c0, v);
    if (sizeof(T) == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = bf16_bits_to_f32(f32_to_bf16_bits(v[e]));
    } else {
      // fp32: the stored value is the rounded product; keep the compiler from contracting x * factor into the sums below (an fma
      // would add the unrounded product and differ from dropout-then-colsum in the last bit)
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(v[e]));
    }
  };
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t cv = (int64_t)blockIdx.x * 16 + cl;
  const bool ok = cv * 8 < cols;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per;
  const int64_t r1 = r0 + rows_per < rows ? r0 + rows_per : rows;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (ok) {
    int64_t r = r0 + g;
    for (; r + 48 < r1; r += 64) {
      // rows_per is a multiple of 64 when stamps are given: this iteration is exactly the 64-row tile (r - g) / 64; a tile whose
      // stamp is not this epoch holds only zero rows and adds nothing
      if (row_live && row_live[(r - g) >> 6] != epoch) {
        if (DROP) {  // an all-zero tile stays all zero under the mask
          float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          store8(xd + r * ldx + cv * 8, z); store8(xd + (r + 16) * ldx + cv * 8, z);
          store8(xd + (r + 32) * ldx + cv * 8, z); store8(xd + (r + 48) * ldx + cv * 8, z);
        }
        continue;
      }
      float v0[8], v1[8], v2[8], v3[8];
      load8(x + r * ldx + cv * 8, v0);
      load8(x + (r + 16) * ldx + cv * 8, v1);
      load8(x + (r + 32) * ldx + cv * 8, v2);
      load8(x + (r + 48) * ldx + cv * 8, v3);
      if (DROP) { masked(v0, r, cv * 8); masked(v1, r + 16, cv * 8); masked(v2, r + 32, cv * 8); masked(v3, r + 48, cv * 8); }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
    }
    for (; r < r1; r += 16) {
      float v[8];
      load8(x + r * ldx + cv * 8, v);
      if (DROP) masked(v, r, cv * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[g][cl * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x;
    float a = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) a += red[k][c];
    const int64_t col = (int64_t)blockIdx.x * 128 + c;
    if (col < cols) {
      if (PART) out[(int64_t)blockIdx.y * cols + col] = a;  // row-chunk partial, reduced in fixed order by colsum_reduce_kernel
      else atomicAdd(out + col, a);
    }
  }
}

// out[c] = sum over the row chunks of part[chunk][c], written in the gradient's dtype (fixed summation order: deterministic)
template <typename TO>
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* part, TO* out, int chunks, int64_t cols) {
  __shared__ float red[64][17];
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t c = (int64_t)blockIdx.x * 16 + cl;
  float a = 0.0f;
  if (c < cols)
    for (int i = g; i < chunks; i += 64) a += part[(int64_t)i * cols + c];
  red[g][cl] = a;
  __syncthreads();
  if (g == 0 && c < cols) {
#pragma unroll 8
    for (int k = 1; k < 64; ++k) a += red[k][cl];
    DT<TO>::st(out + c, a);
  }
}

template <typename T>
__global__ void col2im1d_kernel(const T* dcol, const T* z, T* dx, int64_t B, int64_t Lin, int64_t Lout, int64_t C, int k,
                                int stride, int pad, int dact) {
  const int64_t cv = C / 8, total = B * Lin * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = (i % cv) * 8, l = (i / cv) % Lin, b = i / (cv * Lin);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < k; ++j) {
      const int64_t num = l + pad - j;
      if (num < 0 || num % stride) continue;
      const int64_t t = num / stride;
      if (t >= Lout) continue;
      float v[8];
      load8(dcol + ((b * Lout + t) * k + j) * C + c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
    if (dact) {
      float zz[8];
      load8(z + (b * Lin + l) * C + c, zz);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] *= dact_t<T>(zz[e], dact);
    }
    store8(dx + (b * Lin + l) * C + c, acc);
  }
}

template <typename T>
__global__ void mask_rows_kernel(const T* x, const uint8_t* mask, T* y, int64_t rows, int64_t cols) {
  const int64_t cv = cols / 8, total = rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv;
    float v[8];
    load8(x + i * 8, v);
    if (mask[r]) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.0f;
    }
    store8(y + i * 8, v);
  }
}

// y = x * keep(key, idx) / (1 - p): FairseqDropout (modules/fairseq_dropout.py) forward AND backward (same call on dy)
template <typename T>
__global__ void dropout_kernel(const T* x, T* y, int64_t n, uint32_t key, uint32_t thr16, float scale) {
  const int64_t n8 = n / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    load8(x + i * 8, v);
    cst_drop8(v, key, (uint64_t)i * 8, thr16, scale);
    store8(y + i * 8, v);
  }
  if (blockIdx.x == 0)
    for (int64_t i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) DT<T>::st(y + i, DT<T>::ld(x + i) * cst_drop1(key, (uint64_t)i, thr16, scale));
}

// ---- packed rows: [B, T, C] <-> [N, C] with sequence b at rows [off[b], off[b+1]) --------------------------------------
// pack:   dst[off[b] + t] = src[b, t] for t < n_b - 1;  the LAST packed row of a sequence takes src[b, n_b - 1] (tail == 0) or the
//         sum of src[b, n_b - 1 .. T - 1] (tail == 1: the gradient of `unpack` with broadcast), added in a fixed order.
// unpack: dst[b, t] = src[off[b] + min(t, n_b - 1)] (tail == 1: rows beyond the sequence repeat its last row) or 0 beyond (tail == 0).
// One wave per row, 16-byte vectors; grid (ceil(T / 4), B).
template <typename T>
__device__ __forceinline__ void copy8(T* d, const T* s) {  // 8 elements: one 16-byte vector (bf16) or two (fp32)
  constexpr int NV = (int)(8 * sizeof(T) / 16);
#pragma unroll
  for (int i = 0; i < NV; ++i) reinterpret_cast<u32x4*>(d)[i] = reinterpret_cast<const u32x4*>(s)[i];
}
template <typename T>
__device__ __forceinline__ void zero8(T* d) {
  constexpr int NV = (int)(8 * sizeof(T) / 16);
  const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < NV; ++i) reinterpret_cast<u32x4*>(d)[i] = z;
}

template <typename T>
__global__ __launch_bounds__(256) void rows_pack_kernel(const T* src, const int32_t* off, T* dst, int Tn, int C, int tail) {
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + wave;
  const int n = off[b + 1] - off[b];
  if (t >= n) return;
  const T* s = src + ((int64_t)b * Tn + t) * C;
  T* d = dst + ((int64_t)off[b] + t) * C;
  if (t < n - 1 || !tail) {
    for (int c = lane * 8; c < C; c += 512) copy8(d + c, s + c);
    return;
  }
  for (int c = lane * 8; c < C; c += 512) {  // last row: rows n-1 .. Tn-1 in index order, fp32 accumulation, one rounding
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int r = t;
    for (; r + 8 <= Tn; r += 8) {  // eight rows' loads in flight, added in the same index order (the chain is one row long otherwise)
      float v[8][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) load8(src + ((int64_t)b * Tn + r + u) * C + c, v[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += v[u][e];
    }
    for (; r < Tn; ++r) {
      float v[8];
      load8(src + ((int64_t)b * Tn + r) * C + c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += v[e];
    }
    store8(d + c, a);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rows_unpack_kernel(const T* src, const int32_t* off, T* dst, int Tn, int C, int tail) {
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + wave;
  if (t >= Tn) return;
  const int n = off[b + 1] - off[b];
  T* d = dst + ((int64_t)b * Tn + t) * C;
  if (t >= n && !tail) {
    for (int c = lane * 8; c < C; c += 512) zero8(d + c);
    return;
  }
  const T* s = src + ((int64_t)off[b] + (t < n ? t : n - 1)) * C;
  for (int c = lane * 8; c < C; c += 512) copy8(d + c, s + c);
}

}  // namespace

#define CST_EW_DISPATCH(kern, grid, block, s, dtype, ...)                                        \
  do {                                                                                           \
    if (dtype == CST_BF16) hipLaunchKernelGGL(kern<bf16_t>, grid, block, 0, s, __VA_ARGS__);     \
    else hipLaunchKernelGGL(kern<float>, grid, block, 0, s, __VA_ARGS__);                        \
  } while (0)

extern "C" int cst_glu_fwd(const void* z, void* y, int64_t rows, int64_t C, int dtype, cst_stream stream) {
  CST_REQUIRE(z && y && rows > 0 && C > 0 && C % 8 == 0, "cst_glu_fwd: bad args (C=%lld must be a multiple of 8)", (long long)C);
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_glu_fwd: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, 3.0 * rows * C * cst_dtype_size(dtype));
  if (dtype == CST_BF16) hipLaunchKernelGGL(glu_fwd_kernel<bf16_t>, dim3(ew_blocks(rows * C / 8)), dim3(256), 0, s, (const bf16_t*)z, (bf16_t*)y, rows, C);
  else hipLaunchKernelGGL(glu_fwd_kernel<float>, dim3(ew_blocks(rows * C / 8)), dim3(256), 0, s, (const float*)z, (float*)y, rows, C);
  return cst_check_launch("cst_glu_fwd");
}

extern "C" int cst_glu_bwd(const void* dy, const void* z, void* dz, int64_t rows, int64_t C, int dtype, cst_stream stream) {
  CST_REQUIRE(dy && z && dz && rows > 0 && C > 0 && C % 8 == 0, "cst_glu_bwd: bad args");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_glu_bwd: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, 5.0 * rows * C * cst_dtype_size(dtype));
  if (dtype == CST_BF16) hipLaunchKernelGGL(glu_bwd_kernel<bf16_t>, dim3(ew_blocks(rows * C / 8)), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)z, (bf16_t*)dz, rows, C);
  else hipLaunchKernelGGL(glu_bwd_kernel<float>, dim3(ew_blocks(rows * C / 8)), dim3(256), 0, s, (const float*)dy, (const float*)z, (float*)dz, rows, C);
  return cst_check_launch("cst_glu_bwd");
}

extern "C" int cst_act_bwd(const void* dy, const void* z, void* dx, int64_t n, int act, int dtype, cst_stream stream) {
  CST_REQUIRE(dy && z && dx && n > 0 && n % 8 == 0, "cst_act_bwd: n=%lld must be a positive multiple of 8", (long long)n);
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_act_bwd: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, 3.0 * n * cst_dtype_size(dtype));
  if (dtype == CST_BF16) hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(ew_blocks(n / 8)), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)z, (bf16_t*)dx, n / 8, act);
  else hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(ew_blocks(n / 8)), dim3(256), 0, s, (const float*)dy, (const float*)z, (float*)dx, n / 8, act);
  return cst_check_launch("cst_act_bwd");
}

extern "C" int cst_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, cst_stream stream) {
  CST_REQUIRE(x && y && n > 0 && n % 8 == 0, "cst_act_fwd: n=%lld must be a positive multiple of 8", (long long)n);
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_act_fwd: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, 2.0 * n * cst_dtype_size(dtype));
  if (dtype == CST_BF16) hipLaunchKernelGGL(act_fwd_kernel<bf16_t>, dim3(ew_blocks(n / 8)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, n / 8, act);
  else hipLaunchKernelGGL(act_fwd_kernel<float>, dim3(ew_blocks(n / 8)), dim3(256), 0, s, (const float*)x, (float*)y, n / 8, act);
  return cst_check_launch("cst_act_fwd");
}

extern "C" int cst_colsum(const void* x, int64_t ldx, float* out, int64_t rows, int64_t cols, int dtype, cst_stream stream) {
  CST_REQUIRE(x && out && rows > 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0, "cst_colsum: cols/ldx must be multiples of 8");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_colsum: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, (double)rows * cols * cst_dtype_size(dtype));
  if (hipMemsetAsync(out, 0, sizeof(float) * cols, s) != hipSuccess) { cst_set_error("cst_colsum: memset failed"); return CST_ERR_LAUNCH; }
  const int64_t cblocks = cst_ceil_div(cols / 8, 16);
  int64_t chunks = cst_ceil_div(1536, cblocks);  // ~1536 workgroups in total
  if (chunks > cst_ceil_div(rows, 64)) chunks = cst_ceil_div(rows, 64);
  if (chunks < 1) chunks = 1;
  const int64_t rows_per = cst_ceil_div(rows, chunks);
  dim3 grid((unsigned)cblocks, (unsigned)cst_ceil_div(rows, rows_per));
  if (dtype == CST_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, ldx, out, rows, cols, rows_per);
  else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ldx, out, rows, cols, rows_per);
  return cst_check_launch("cst_colsum");
}

static void colsum_grid(int64_t rows, int64_t cols, int64_t& cblocks, int64_t& nchunks, int64_t& rows_per) {
  cblocks = cst_ceil_div(cols / 8, 16);
  int64_t chunks = cst_ceil_div(1536, cblocks);  // ~1536 workgroups in total
  if (chunks > cst_ceil_div(rows, 64)) chunks = cst_ceil_div(rows, 64);
  if (chunks < 1) chunks = 1;
  rows_per = cst_ceil_div(cst_ceil_div(rows, chunks), 64) * 64;  // whole 64-row tiles per chunk (live-tile stamps)
  nchunks = cst_ceil_div(rows, rows_per);
}

extern "C" int64_t cst_colsum_workspace(int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  int64_t cb, nc, rp;
  colsum_grid(rows, cols, cb, nc, rp);
  return nc * cols * (int64_t)sizeof(float);
}

static int colsum_typed_impl(const void* x, int64_t ldx, void* out, void* workspace, int64_t rows, int64_t cols, int dtype,
                             int out_dtype, const uint32_t* row_live, uint32_t epoch, cst_stream stream) {
  // out == NULL: the row-chunk partials stay in `workspace` ([cst_colsum_workspace / (4 cols)][cols] fp32) for cst_reduce_multi, order 1
  CST_REQUIRE(x && workspace && rows > 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0, "cst_colsum_typed: cols/ldx must be multiples of 8");
  CST_REQUIRE((dtype == CST_F32 || dtype == CST_BF16) && (out_dtype == CST_F32 || out_dtype == CST_BF16), "cst_colsum_typed: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, (double)rows * cols * cst_dtype_size(dtype));
  int64_t cblocks, nchunks, rows_per;
  colsum_grid(rows, cols, cblocks, nchunks, rows_per);
  dim3 grid((unsigned)cblocks, (unsigned)nchunks);
  float* part = (float*)workspace;
  if (dtype == CST_BF16) hipLaunchKernelGGL((colsum_kernel<bf16_t, true>), grid, dim3(256), 0, s, (const bf16_t*)x, ldx, part, rows, cols, rows_per, row_live, epoch);
  else hipLaunchKernelGGL((colsum_kernel<float, true>), grid, dim3(256), 0, s, (const float*)x, ldx, part, rows, cols, rows_per, row_live, epoch);
  if (!out) return cst_check_launch("cst_colsum_typed");
  const dim3 rgrid((unsigned)cst_ceil_div(cols, 16));
  if (out_dtype == CST_BF16) hipLaunchKernelGGL(colsum_reduce_kernel<bf16_t>, rgrid, dim3(1024), 0, s, (const float*)part, (bf16_t*)out, (int)nchunks, cols);
  else hipLaunchKernelGGL(colsum_reduce_kernel<float>, rgrid, dim3(1024), 0, s, (const float*)part, (float*)out, (int)nchunks, cols);
  return cst_check_launch("cst_colsum_typed");
}

extern "C" int cst_dropout_colsum(const void* x, void* xd, void* out, void* workspace, int64_t rows, int64_t cols, int dtype, int out_dtype,
                                  float p, uint32_t key, const uint32_t* row_live, uint32_t epoch, cst_stream stream) {
  CST_REQUIRE(x && xd && workspace && rows > 0 && cols > 0 && cols % 8 == 0, "cst_dropout_colsum: cols must be a multiple of 8");
  CST_REQUIRE((dtype == CST_F32 || dtype == CST_BF16) && (out_dtype == CST_F32 || out_dtype == CST_BF16), "cst_dropout_colsum: bad dtype");
  CST_REQUIRE(p > 0.0f && p < 1.0f, "cst_dropout_colsum: p must be in (0, 1)");
  CST_REQUIRE(!row_live || epoch != 0, "cst_dropout_colsum: stamps need a non-zero epoch");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, 2.0 * (double)rows * cols * cst_dtype_size(dtype));
  int64_t cblocks, nchunks, rows_per;
  colsum_grid(rows, cols, cblocks, nchunks, rows_per);
  dim3 grid((unsigned)cblocks, (unsigned)nchunks);
  float* part = (float*)workspace;
  const uint32_t thr = cst_drop_thr16(p);
  const float scale = 1.0f / (1.0f - p);
  if (dtype == CST_BF16) hipLaunchKernelGGL((colsum_kernel<bf16_t, true, true>), grid, dim3(256), 0, s, (const bf16_t*)x, cols, part, rows, cols, rows_per, row_live, epoch, (bf16_t*)xd, key, thr, scale);
  else hipLaunchKernelGGL((colsum_kernel<float, true, true>), grid, dim3(256), 0, s, (const float*)x, cols, part, rows, cols, rows_per, row_live, epoch, (float*)xd, key, thr, scale);
  if (!out) return cst_check_launch("cst_dropout_colsum");  // partials left for cst_reduce_multi (order 1), as in cst_colsum_typed
  const dim3 rgrid((unsigned)cst_ceil_div(cols, 16));
  if (out_dtype == CST_BF16) hipLaunchKernelGGL(colsum_reduce_kernel<bf16_t>, rgrid, dim3(1024), 0, s, (const float*)part, (bf16_t*)out, (int)nchunks, cols);
  else hipLaunchKernelGGL(colsum_reduce_kernel<float>, rgrid, dim3(1024), 0, s, (const float*)part, (float*)out, (int)nchunks, cols);
  return cst_check_launch("cst_dropout_colsum");
}

extern "C" int cst_colsum_typed(const void* x, int64_t ldx, void* out, void* workspace, int64_t rows, int64_t cols, int dtype,
                                int out_dtype, cst_stream stream) {
  return colsum_typed_impl(x, ldx, out, workspace, rows, cols, dtype, out_dtype, nullptr, 0, stream);
}

extern "C" int cst_colsum_typed_live(const void* x, int64_t ldx, void* out, void* workspace, int64_t rows, int64_t cols, int dtype,
                                     int out_dtype, const uint32_t* row_live, uint32_t epoch, cst_stream stream) {
  CST_REQUIRE(row_live && epoch != 0, "cst_colsum_typed_live: stamps / non-zero epoch required");
  return colsum_typed_impl(x, ldx, out, workspace, rows, cols, dtype, out_dtype, row_live, epoch, stream);
}

extern "C" int cst_col2im1d(const void* dcol, const void* z, void* dx, int64_t B, int64_t Lin, int64_t Lout, int64_t C, int k,
                            int stride, int pad, int dact, int dtype, cst_stream stream) {
  CST_REQUIRE(dcol && dx && B > 0 && Lin > 0 && Lout > 0 && C > 0 && C % 8 == 0 && k >= 1 && stride >= 1 && pad >= 0, "cst_col2im1d: bad args");
  CST_REQUIRE(!dact || z, "cst_col2im1d: dact needs z");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_col2im1d: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, ((double)B * Lout * k * C + (double)B * Lin * C * (dact ? 2.0 : 1.0)) * cst_dtype_size(dtype));
  const int blocks = ew_blocks(B * Lin * C / 8);
  if (dtype == CST_BF16) hipLaunchKernelGGL(col2im1d_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)dcol, (const bf16_t*)z, (bf16_t*)dx, B, Lin, Lout, C, k, stride, pad, dact);
  else hipLaunchKernelGGL(col2im1d_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)dcol, (const float*)z, (float*)dx, B, Lin, Lout, C, k, stride, pad, dact);
  return cst_check_launch("cst_col2im1d");
}

extern "C" int cst_mask_rows(const void* x, const uint8_t* mask, void* y, int64_t rows, int64_t cols, int dtype, cst_stream stream) {
  CST_REQUIRE(x && mask && y && rows > 0 && cols > 0 && cols % 8 == 0, "cst_mask_rows: bad args");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_mask_rows: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, 2.0 * rows * cols * cst_dtype_size(dtype));
  const int blocks = ew_blocks(rows * cols / 8);
  if (dtype == CST_BF16) hipLaunchKernelGGL(mask_rows_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, mask, (bf16_t*)y, rows, cols);
  else hipLaunchKernelGGL(mask_rows_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, mask, (float*)y, rows, cols);
  return cst_check_launch("cst_mask_rows");
}

extern "C" int cst_dropout(const void* x, void* y, int64_t n, float p, uint32_t key, int dtype, cst_stream stream) {
  CST_REQUIRE(x && y && n > 0 && p >= 0.0f && p < 1.0f, "cst_dropout: bad args");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_dropout: bad dtype");
  CST_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0), "cst_dropout: x/y must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, 2.0 * n * cst_dtype_size(dtype));
  const int blocks = ew_blocks(cst_ceil_div(n, 8));
  const uint32_t thr = cst_drop_thr16(p);
  const float scale = 1.0f / (1.0f - p);
  if (dtype == CST_BF16) hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, n, key, thr, scale);
  else hipLaunchKernelGGL(dropout_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, (float*)y, n, key, thr, scale);
  return cst_check_launch("cst_dropout");
}


extern "C" int cst_rows_pack(const void* src, const int32_t* seq_off, void* dst, int64_t B, int64_t T, int64_t C, int tail_sum, int dtype,
                             cst_stream stream) {
  CST_REQUIRE(src && seq_off && dst && B > 0 && T > 0 && C > 0, "cst_rows_pack: bad args");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_rows_pack: bad dtype %d", dtype);
  CST_REQUIRE(C % 8 == 0 && B < 65536, "cst_rows_pack: C=%lld must be a multiple of 8", (long long)C);
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, 2.0 * B * T * C * cst_dtype_size(dtype));
  const dim3 grid((unsigned)cst_ceil_div(T, 4), (unsigned)B);
  if (dtype == CST_BF16) hipLaunchKernelGGL(rows_pack_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)src, seq_off, (bf16_t*)dst, (int)T, (int)C, tail_sum);
  else hipLaunchKernelGGL(rows_pack_kernel<float>, grid, dim3(256), 0, s, (const float*)src, seq_off, (float*)dst, (int)T, (int)C, tail_sum);
  return cst_check_launch("cst_rows_pack");
}

extern "C" int cst_rows_unpack(const void* src, const int32_t* seq_off, void* dst, int64_t B, int64_t T, int64_t C, int tail_broadcast, int dtype,
                               cst_stream stream) {
  CST_REQUIRE(src && seq_off && dst && B > 0 && T > 0 && C > 0, "cst_rows_unpack: bad args");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_rows_unpack: bad dtype %d", dtype);
  CST_REQUIRE(C % 8 == 0 && B < 65536, "cst_rows_unpack: C=%lld must be a multiple of 8", (long long)C);
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, 2.0 * B * T * C * cst_dtype_size(dtype));
  const dim3 grid((unsigned)cst_ceil_div(T, 4), (unsigned)B);
  if (dtype == CST_BF16) hipLaunchKernelGGL(rows_unpack_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)src, seq_off, (bf16_t*)dst, (int)T, (int)C, tail_broadcast);
  else hipLaunchKernelGGL(rows_unpack_kernel<float>, grid, dim3(256), 0, s, (const float*)src, seq_off, (float*)dst, (int)T, (int)C, tail_broadcast);
  return cst_check_launch("cst_rows_unpack");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// weight normalisation along the last dimension (nn.utils.weight_norm(conv, dim=2) of the wav2vec2 positional convolution,
// models/wav2vec/wav2vec2.py:773-779): v [R, C] (R = C_out * C_in / groups, C = kernel width: one norm per kernel tap), g [C]:
//   w[r, c] = v[r, c] * g[c] / ||v[:, c]||          dv = (g / n) * (dw - v * dot / n^2),  dg = dot / n,  dot[c] = sum_r dw * v
// Two launches each way, fixed summation order (bit-reproducible): row-chunk partial sums per column, then every block of the apply
// pass adds the partials of its columns in chunk order and scales its rows.  Replaces ten ATen reduce / elementwise kernels per
// update (0.46 ms in profiles/r02_kernel_stats.txt).
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int WN_ROWS = 256;  // rows per partial chunk

template <typename T, bool DOT>
__global__ __launch_bounds__(256) void wn_partial_kernel(const T* v, const T* dw, float* part, int64_t R, int C) {
  // thread -> (row lane, 8-column group); C % 8 == 0, C <= 256
  const int gpr = C / 8, rl = threadIdx.x / gpr, cg = threadIdx.x % gpr, rpp = 256 / gpr;
  __shared__ float red[256 * 8];
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t r0 = (int64_t)blockIdx.x * WN_ROWS;
  if (rl < rpp)
    for (int64_t r = r0 + rl; r < r0 + WN_ROWS && r < R; r += rpp) {
      float a[8], b[8];
      load8(v + r * C + cg * 8, a);
      if (DOT) load8(dw + r * C + cg * 8, b);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += DOT ? a[e] * b[e] : a[e] * a[e];
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = acc[e];
  __syncthreads();
  if ((int)threadIdx.x < C) {  // column c: add the row lanes in order
    const int c = threadIdx.x, g8 = c / 8, e = c % 8;
    float s = 0.0f;
    for (int l = 0; l < rpp; ++l) s += red[(l * gpr + g8) * 8 + e];
    part[(int64_t)blockIdx.x * C + c] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void wn_apply_fwd_kernel(const T* v, const T* g, T* w, float* norm, const float* part, int nchunks, int64_t R, int C) {
  __shared__ float scale[256];
  if ((int)threadIdx.x < C) {
    float s = 0.0f;
    for (int k = 0; k < nchunks; ++k) s += part[(int64_t)k * C + threadIdx.x];
    const float n = sqrtf(s);
    if (blockIdx.x == 0) norm[threadIdx.x] = n;
    scale[threadIdx.x] = DT<T>::ld(g + threadIdx.x) / n;
  }
  __syncthreads();
  const int gpr = C / 8;
  const int64_t total = R * gpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % gpr);
    float a[8];
    load8(v + i * 8, a);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] *= scale[cg * 8 + e];
    store8(w + i * 8, a);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void wn_apply_bwd_kernel(const T* v, const T* g, const T* dw, const float* norm, T* dv, T* dg, const float* part,
                                                           int nchunks, int64_t R, int C) {
  __shared__ float sc[256], co[256];
  if ((int)threadIdx.x < C) {
    float dot = 0.0f;
    for (int k = 0; k < nchunks; ++k) dot += part[(int64_t)k * C + threadIdx.x];
    const float n = norm[threadIdx.x], gg = DT<T>::ld(g + threadIdx.x);
    sc[threadIdx.x] = gg / n;
    co[threadIdx.x] = dot / (n * n);
    if (blockIdx.x == 0) DT<T>::st(dg + threadIdx.x, dot / n);
  }
  __syncthreads();
  const int gpr = C / 8;
  const int64_t total = R * gpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % gpr);
    float a[8], b[8];
    load8(v + i * 8, a);
    load8(dw + i * 8, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = sc[cg * 8 + e] * (b[e] - a[e] * co[cg * 8 + e]);
    store8(dv + i * 8, a);
  }
}
}  // namespace

extern "C" int64_t cst_weight_norm_workspace(int64_t R, int64_t C) { return cst_ceil_div(R, WN_ROWS) * C * (int64_t)sizeof(float); }

extern "C" int cst_weight_norm_fwd(const void* v, const void* g, void* w, float* norm, void* workspace, int64_t R, int64_t C, int dtype,
                                   cst_stream stream) {
  CST_REQUIRE(v && g && w && norm && workspace && R > 0, "cst_weight_norm_fwd: null tensor");
  CST_REQUIRE(C % 8 == 0 && C >= 8 && C <= 256 && 256 % (C / 8) == 0, "cst_weight_norm_fwd: C = %lld (need a multiple of 8 up to 256 whose eighth divides 256)", (long long)C);
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, (double)R * C * cst_dtype_size(dtype) * 3);
  const int nch = (int)cst_ceil_div(R, WN_ROWS);
  const int blocks = (int)(cst_ceil_div(R * (C / 8), 256) < 1024 ? cst_ceil_div(R * (C / 8), 256) : 1024);
  if (dtype == CST_BF16) {
    hipLaunchKernelGGL((wn_partial_kernel<bf16_t, false>), dim3(nch), dim3(256), 0, s, (const bf16_t*)v, (const bf16_t*)nullptr, (float*)workspace, R, (int)C);
    hipLaunchKernelGGL(wn_apply_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)v, (const bf16_t*)g, (bf16_t*)w, norm, (const float*)workspace, nch, R, (int)C);
  } else {
    hipLaunchKernelGGL((wn_partial_kernel<float, false>), dim3(nch), dim3(256), 0, s, (const float*)v, (const float*)nullptr, (float*)workspace, R, (int)C);
    hipLaunchKernelGGL(wn_apply_fwd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)v, (const float*)g, (float*)w, norm, (const float*)workspace, nch, R, (int)C);
  }
  return cst_check_launch("cst_weight_norm_fwd");
}

extern "C" int cst_weight_norm_bwd(const void* v, const void* g, const void* dw, const float* norm, void* dv, void* dg, void* workspace, int64_t R,
                                   int64_t C, int dtype, cst_stream stream) {
  CST_REQUIRE(v && g && dw && norm && dv && dg && workspace && R > 0, "cst_weight_norm_bwd: null tensor");
  CST_REQUIRE(C % 8 == 0 && C >= 8 && C <= 256 && 256 % (C / 8) == 0, "cst_weight_norm_bwd: C = %lld", (long long)C);
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, (double)R * C * cst_dtype_size(dtype) * 5);
  const int nch = (int)cst_ceil_div(R, WN_ROWS);
  const int blocks = (int)(cst_ceil_div(R * (C / 8), 256) < 1024 ? cst_ceil_div(R * (C / 8), 256) : 1024);
  if (dtype == CST_BF16) {
    hipLaunchKernelGGL((wn_partial_kernel<bf16_t, true>), dim3(nch), dim3(256), 0, s, (const bf16_t*)v, (const bf16_t*)dw, (float*)workspace, R, (int)C);
    hipLaunchKernelGGL(wn_apply_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)v, (const bf16_t*)g, (const bf16_t*)dw, norm, (bf16_t*)dv, (bf16_t*)dg,
                       (const float*)workspace, nch, R, (int)C);
  } else {
    hipLaunchKernelGGL((wn_partial_kernel<float, true>), dim3(nch), dim3(256), 0, s, (const float*)v, (const float*)dw, (float*)workspace, R, (int)C);
    hipLaunchKernelGGL(wn_apply_bwd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)v, (const float*)g, (const float*)dw, norm, (float*)dv, (float*)dg,
                       (const float*)workspace, nch, R, (int)C);
  }
  return cst_check_launch("cst_weight_norm_bwd");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dst[c, r] = src[r, c]: a [R, C] matrix transposed through a 64 x 64 LDS tile (16-byte global accesses both ways).  Used for the
// weight operand of the Linear dX GEMMs: with W^T [K_in, N_out] the GEMM dY W reads BOTH operands k-major (ds_read_b128 fragments),
// as the forward GEMM does; with W [N_out, K_in] its B operand goes through transpose reads, which measured 15-20 % slower on the
// 31 760-row shapes (tools/gemm_shapes_in_step.py).  R % 8 == 0, C % 8 == 0.
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
template <typename T>
__device__ __forceinline__ void transpose_tile(const T* src, T* dst, int64_t R, int64_t C, int64_t r0, int64_t c0, T* tile) {
  constexpr int V = DT<T>::VEC, TS = 64, LDT = TS + V;   // padded rows: column reads walk different banks
  constexpr int VPR = TS / V;
  for (int v = threadIdx.x; v < TS * VPR; v += 256) {
    const int r = v / VPR, cv = (v % VPR) * V;
    u32x4 x = {0, 0, 0, 0};
    if (r0 + r < R && c0 + cv < C) x = *reinterpret_cast<const u32x4*>(src + (r0 + r) * C + c0 + cv);
    *reinterpret_cast<u32x4*>(tile + r * LDT + cv) = x;
  }
  __syncthreads();
  for (int v = threadIdx.x; v < TS * VPR; v += 256) {
    const int c = v / VPR, rv = (v % VPR) * V;   // output row c (a source column), V consecutive source rows
    if (c0 + c < C && r0 + rv < R) {
      T out[V];
#pragma unroll
      for (int e = 0; e < V; ++e) out[e] = tile[(rv + e) * LDT + c];
      *reinterpret_cast<u32x4*>(dst + (c0 + c) * R + r0 + rv) = *reinterpret_cast<const u32x4*>(out);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void transpose2d_kernel(const T* src, T* dst, int64_t R, int64_t C) {
  __shared__ __attribute__((aligned(16))) T tile[64 * (64 + DT<T>::VEC)];
  transpose_tile<T>(src, dst, R, C, (int64_t)blockIdx.y * 64, (int64_t)blockIdx.x * 64, tile);
}

// every matrix of a table in ONE launch: block -> (matrix, tile) by a binary search over the matrices' first-tile numbers
template <typename T>
__global__ __launch_bounds__(256) void transpose2d_multi_kernel(const cst_transpose_item* items, int n) {
  __shared__ __attribute__((aligned(16))) T tile[64 * (64 + DT<T>::VEC)];
  const int64_t b = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {  // last item whose first tile is <= b (uniform: scalar loads)
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].tile0 <= b) lo = mid; else hi = mid - 1;
  }
  const cst_transpose_item it = items[lo];
  const int64_t t = b - it.tile0, tx = (it.C + 63) / 64;
  transpose_tile<T>((const T*)it.src, (T*)it.dst, it.R, it.C, (t / tx) * 64, (t % tx) * 64, tile);
}
}  // namespace

extern "C" int cst_transpose2d_multi(const cst_transpose_item* items_dev, int n, int64_t total_tiles, int dtype, cst_stream stream) {
  CST_REQUIRE(items_dev && n > 0 && total_tiles > 0 && total_tiles < (1ll << 31), "cst_transpose2d_multi: bad table");
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_transpose2d_multi: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, 2.0 * (double)total_tiles * 4096 * cst_dtype_size(dtype));
  if (dtype == CST_BF16) hipLaunchKernelGGL(transpose2d_multi_kernel<bf16_t>, dim3((unsigned)total_tiles), dim3(256), 0, s, items_dev, n);
  else hipLaunchKernelGGL(transpose2d_multi_kernel<float>, dim3((unsigned)total_tiles), dim3(256), 0, s, items_dev, n);
  return cst_check_launch("cst_transpose2d_multi");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// cst_reduce_multi — the second stage of MANY fixed-order reductions in one launch.  The backward pass of one update ends ~140
// two-stage reductions (split-K slabs of the small layers' weight gradients, bias-gradient slices riding with them, the
// LayerNorm dgamma / dbeta row-block partials) with a tiny launch each; their results are only read when the gradients are
// gathered, so the first stages leave their partials in place and ONE launch finishes them all (functional / kernels.py: deferred
// reductions, flushed before anything reads a gradient).
// Item: dst[i] = sum_{p < P} src[p * stride + i], i < L (L % 8 == 0), fp32 partials, dst in fp32 or bf16 — in EXACTLY the order of
// the launch-each kernel it stands in for, so that a gradient has the same bits whichever route produced it (an accumulated
// update mixes the routes: its first micro-batch may be deferred, the later ones add into an existing gradient and are not):
//   order 0  p = 0, 1, 2, ... one after the other                 (splitk_reduce_kernel of gemm.hip; its bias-gradient slices)
//   order 1  64 interleaved chains p = g, g + 64, ..., then the chains 0..63 one after the other   (ln_bwd_reduce_kernel)
// Blocks of 1024 threads: order 0 — 8 consecutive elements per thread; order 1 — 64 columns (16 lanes x 4) x 64 chains: the per-column
// order of layernorm.hip's / colsum's second stage, with 16-byte loads.
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
struct ReduceTable { cst_reduce_item it[CST_REDUCE_MAX_ITEMS]; int n; };

__global__ __launch_bounds__(1024) void reduce_multi_kernel(ReduceTable t) {
  __shared__ float ra[64][65];
  const int b = blockIdx.x;
  int lo = 0, hi = t.n - 1;
  while (lo < hi) {  // last item whose first block is <= b (uniform)
    const int mid = (lo + hi + 1) >> 1;
    if (t.it[mid].block0 <= b) lo = mid; else hi = mid - 1;
  }
  const cst_reduce_item& it = t.it[lo];
  const int64_t blk = b - it.block0;
  if (it.order == 0) {
    const int64_t i0 = (blk * 1024 + threadIdx.x) * 8;
    if (i0 >= it.L) return;
    const float* src = it.src + i0;
    float v[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    int sp = 0;
    for (; sp + 4 <= it.P; sp += 4) {  // four partials in flight, added in order (the loop of splitk_reduce_kernel)
      f32x4 a[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u][0] = *reinterpret_cast<const f32x4*>(src + (int64_t)(sp + u) * it.stride);
        a[u][1] = *reinterpret_cast<const f32x4*>(src + (int64_t)(sp + u) * it.stride + 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += a[u][0][e]; v[4 + e] += a[u][1][e]; }
    }
    for (; sp < it.P; ++sp) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + (int64_t)sp * it.stride), a1 = *reinterpret_cast<const f32x4*>(src + (int64_t)sp * it.stride + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += a0[e]; v[4 + e] += a1[e]; }
    }
    if (it.dst_dtype == CST_BF16) store8((bf16_t*)it.dst + i0, v);
    else store8((float*)it.dst + i0, v);
    return;
  }
  // order 1: 64 columns per block — 16 lanes x 4 consecutive columns (one 16-byte load each: 256-byte row segments) x 64 chains
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t c = blk * 64 + cl * 4;
  f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
  if (c < it.L) {
    const float* src = it.src + c;
    int i = g;
    for (; i + 192 < it.P; i += 256) {  // four loads of this chain in flight, added in chain order
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (int64_t)i * it.stride), v1 = *reinterpret_cast<const f32x4*>(src + (int64_t)(i + 64) * it.stride);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + (int64_t)(i + 128) * it.stride), v3 = *reinterpret_cast<const f32x4*>(src + (int64_t)(i + 192) * it.stride);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = (((a[e] + v0[e]) + v1[e]) + v2[e]) + v3[e];
    }
    for (; i < it.P; i += 64) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)i * it.stride);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) ra[g][cl * 4 + e] = a[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int64_t cc = blk * 64 + threadIdx.x;
    if (cc < it.L) {
      float t = ra[0][threadIdx.x];
#pragma unroll 8
      for (int k = 1; k < 64; ++k) t += ra[k][threadIdx.x];
      if (it.dst_dtype == CST_BF16) DT<bf16_t>::st((bf16_t*)it.dst + cc, t);
      else ((float*)it.dst)[cc] = t;
    }
  }
}
}  // namespace

extern "C" int cst_reduce_multi(const cst_reduce_item* items, int n, cst_stream stream) {
  CST_REQUIRE(items && n > 0 && n <= CST_REDUCE_MAX_ITEMS, "cst_reduce_multi: 1..%d items per call", CST_REDUCE_MAX_ITEMS);
  ReduceTable t;
  int64_t blocks = 0;
  double bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    const cst_reduce_item& it = items[i];
    CST_REQUIRE(it.src && it.dst && it.L > 0 && it.L % 8 == 0 && it.P > 0 && it.stride % 4 == 0 && ((uintptr_t)it.src % 16) == 0 && ((uintptr_t)it.dst % 16) == 0,
                "cst_reduce_multi: item %d: L, stride, alignment", i);
    CST_REQUIRE((it.dst_dtype == CST_F32 || it.dst_dtype == CST_BF16) && (it.order == 0 || it.order == 1), "cst_reduce_multi: item %d: bad dst_dtype / order", i);
    t.it[i] = it;
    t.it[i].block0 = (int32_t)blocks;
    blocks += it.order == 0 ? cst_ceil_div(it.L, 8192) : cst_ceil_div(it.L, 64);
    bytes += (double)it.L * ((double)it.P * 4.0 + cst_dtype_size(it.dst_dtype));
  }
  CST_REQUIRE(blocks < (1ll << 31), "cst_reduce_multi: too many blocks");
  t.n = n;
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, bytes);
  hipLaunchKernelGGL(reduce_multi_kernel, dim3((unsigned)blocks), dim3(1024), 0, s, t);
  return cst_check_launch("cst_reduce_multi");
}

extern "C" int cst_transpose2d(const void* src, void* dst, int64_t R, int64_t C, int dtype, cst_stream stream) {
  CST_REQUIRE(src && dst && R > 0 && C > 0, "cst_transpose2d: null tensor");
  const int v = dtype == CST_BF16 ? 8 : 4;
  CST_REQUIRE(R % v == 0 && C % v == 0, "cst_transpose2d: R and C must be multiples of %d", v);
  hipStream_t s = (hipStream_t)stream;
  CstProfScope prof(CST_K_ELEMENTWISE, s, 0.0, 2.0 * (double)R * C * cst_dtype_size(dtype));
  dim3 grid((unsigned)cst_ceil_div(C, 64), (unsigned)cst_ceil_div(R, 64));
  if (dtype == CST_BF16) hipLaunchKernelGGL(transpose2d_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, R, C);
  else hipLaunchKernelGGL(transpose2d_kernel<float>, grid, dim3(256), 0, s, (const float*)src, (float*)dst, R, C);
  return cst_check_launch("cst_transpose2d");
}
