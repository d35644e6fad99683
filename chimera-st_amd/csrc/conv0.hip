// conv0.hip — wav2vec2 conv layer 0 (Cin = 1, k = 10, stride 5) fused with GroupNorm(C groups,
// fp32 statistics) and GELU.  Replaces Conv1d + Fp32GroupNorm + GELU at
// fairseq/models/wav2vec/wav2vec2.py:697-753 (layer 0) / modules/fp32_group_norm.py:13-25.
//
// HBM-bound: the op's floor is "read the wave once, write y once" (512 x L0 outputs per
// utterance, 98 MB in bf16 for 30 s).  Two tricks keep it at that floor:
//  (1) GroupNorm statistics WITHOUT materialising the conv output: u[c,t] = sum_j w[c,j] x[s t + j]
//      is linear in x, so  sum_t u = w_c . m   and  sum_t u^2 = w_c^T G w_c  with the k-vector
//      m[j] = sum_t x[s t + j] and the k x k Gram matrix G[j,j'] = sum_t x[s t + j] x[s t + j'] of
//      the utterance — one cheap pass over the 1.9 MB wave instead of a pass over 512 channels.
//  (2) the normalisation is folded into the filter: y = GELU(sum_j (a_c w[c,j]) x[.] + b_c),
//      a_c = gamma_c rstd_c, b_c = beta_c - mean_c a_c; output written channels-last [B, L, C]
//      (one wave stores 1 KiB contiguous per frame) so the next conv is an implicit GEMM.
// Backward needs ONE pass over dy: per (b,c) it accumulates s1 = sum dz, s2 = sum dz*uhat and
// r[j] = sum_t dz x[s t + j]; dW/dgamma/dbeta then follow in closed form from (s1, s2, r, m, G).
#include "cst_common.h"
#include "gemm_common.h"  // GEMM_MAX_BM: the tile height the layer-1 conv GEMM reads layer-0 frames in

namespace {

constexpr int KMAX = 16;

// ---- pass A: per-utterance moments m[k], G[k][k]: per-block partials in a fixed layout [B][gridDim.x][k*k+k] (no atomics:
//      the statistics — and with them every activation of the network — are bit-reproducible run to run) ----------------
template <int KC>  // KC = k when k is a supported compile-time size (no padded taps: 55 instead of 136 products per frame at k = 10), else KMAX
__global__ void conv0_gram_kernel(const float* wav, float* part, int64_t S, int64_t L, int k, int stride) {
  const int64_t b = blockIdx.y;
  const float* x = wav + b * S;
  const int nacc = k * k + k;
  float* g = part + (b * gridDim.x + blockIdx.x) * nacc;
  __shared__ float red[4][KMAX * KMAX + KMAX];
  float m[KC], G[KC * (KC + 1) / 2];
#pragma unroll
  for (int j = 0; j < KC; ++j) m[j] = 0.0f;
#pragma unroll
  for (int j = 0; j < KC * (KC + 1) / 2; ++j) G[j] = 0.0f;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < L; t += (int64_t)gridDim.x * blockDim.x) {
    float xv[KC];
#pragma unroll
    for (int j = 0; j < KC; ++j) xv[j] = j < k ? x[t * stride + j] : 0.0f;
    int idx = 0;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      m[j] += xv[j];
#pragma unroll
      for (int j2 = j; j2 < KC; ++j2) G[idx++] += xv[j] * xv[j2];
    }
  }
  const int wave = threadIdx.x >> 6;
  int idx = 0;
#pragma unroll
  for (int j = 0; j < KC; ++j) {
    const float mj = wave_sum(m[j]);
    if ((threadIdx.x & 63) == 0 && j < k) red[wave][k * k + j] = mj;
#pragma unroll
    for (int j2 = j; j2 < KC; ++j2) {
      const float gj = wave_sum(G[idx++]);
      if ((threadIdx.x & 63) == 0 && j < k && j2 < k) {
        red[wave][j * k + j2] = gj;
        red[wave][j2 * k + j] = gj;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nacc; i += blockDim.x) g[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);  // blockDim.x == 256
}

// ---- pass B: mean/rstd per (b,c) from the moments --------------------------------------------
template <typename T>
__global__ void conv0_stats_kernel(const T* w, const float* part, int nparts, float* gram, float* mean, float* rstd, int64_t C, int64_t L, int k,
                                   float eps) {
  // fixed-order sum of the block partials (every block of an utterance redoes it: nparts * 110 floats out of L2), kept in LDS;
  // block 0 also publishes it as gram[b] for the backward pass
  __shared__ float g[KMAX * KMAX + KMAX];
  const int64_t b = blockIdx.y;
  const int nacc = k * k + k;
  for (int i = threadIdx.x; i < nacc; i += blockDim.x) {
    double a = 0.0;
    for (int q = 0; q < nparts; ++q) a += (double)part[(b * nparts + q) * nacc + i];
    g[i] = (float)a;
    if (blockIdx.x == 0) gram[b * nacc + i] = (float)a;
  }
  __syncthreads();
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double wm = 0.0, wgw = 0.0;
  for (int j = 0; j < k; ++j) {
    const double wj = (double)DT<T>::ld(w + c * k + j);
    wm += wj * (double)g[k * k + j];
    double row = 0.0;
    for (int j2 = 0; j2 < k; ++j2) row += (double)g[j * k + j2] * (double)DT<T>::ld(w + c * k + j2);
    wgw += wj * row;
  }
  const double mu = wm / (double)L;
  double var = wgw / (double)L - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[b * C + c] = (float)mu;
  rstd[b * C + c] = (float)(1.0 / sqrt(var + (double)eps));
}

// ---- pass C: y = GELU(a_c * conv + b_c), channels-last ---------------------------------------
// block = 256 threads; lane -> (frame-in-group, 8-channel vector); frames of a block share an LDS copy of the wave span.
constexpr int C0_TB = 64;  // frames per block

template <typename T>
__global__ __launch_bounds__(256) void conv0_fwd_kernel(const float* wav, const T* w, const T* gamma, const T* beta,
                                                        const float* mean, const float* rstd, T* y, int64_t S, int64_t L,
                                                        int C, int k, int stride) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sw = sm;                 // [k][C] folded filter a_c * w[c][j]
  float* sb = sw + k * C;         // [C]    b_c
  float* sx = sb + C;             // [C0_TB*stride + k] wave span
  const int64_t b = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * C0_TB;
  const int nt = (int)((L - t0 < C0_TB) ? (L - t0) : C0_TB);
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    const float a = DT<T>::ld(gamma + i) * rstd[b * C + i];
    sb[i] = DT<T>::ld(beta + i) - mean[b * C + i] * a;
    for (int j = 0; j < k; ++j) sw[j * C + i] = a * DT<T>::ld(w + (int64_t)i * k + j);
  }
  const int span = (nt - 1) * stride + k;
  for (int i = threadIdx.x; i < span; i += blockDim.x) sx[i] = wav[b * S + t0 * stride + i];
  __syncthreads();
  const int cvecs = C / 8;
  for (int item = threadIdx.x; item < nt * cvecs; item += blockDim.x) {
    const int tl = item / cvecs, cv = item % cvecs;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = sb[cv * 8 + e];
    for (int j = 0; j < k; ++j) {
      const float xv = sx[tl * stride + j];
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(sw + j * C + cv * 8);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(sw + j * C + cv * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[e] += w0[e] * xv; acc[4 + e] += w1[e] * xv; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = gelu_t<T>(acc[e]);
    store8(y + (b * L + t0 + tl) * C + cv * 8, acc);
  }
}

// ---- backward pass over dy: per (b,c) accumulate s1, s2, r[0..k) --------------------------------
// partial layout in ws: [B][gridDim.x][k+2][C]  (row 0 = s1, row 1 = s2, rows 2.. = r[j]); conv0_bwd_reduce_kernel adds the blocks of
// an utterance in a fixed order into [B][k+2][C] (no atomics: bit-reproducible gradients)
constexpr int C0_BWD_TB = 2048;  // frames per block (fewer, larger blocks: the per-block result goes out as one partial row set)

template <typename T>
__global__ __launch_bounds__(256) void conv0_bwd_kernel(const T* dy, const float* wav, const T* w, const T* gamma,
                                                        const T* beta, const float* mean, const float* rstd, float* ws,
                                                        int64_t S, int64_t L, int C, int k, int stride) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sw = sm;            // [k][C] raw filter
  float* sa = sw + k * C;    // [C] gamma, [C] beta, [C] mean, [C] rstd
  float* sx = sa + 4 * C;    // wave span
  const int64_t b = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * C0_BWD_TB;
  const int nt = (int)((L - t0 < C0_BWD_TB) ? (L - t0) : C0_BWD_TB);
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    sa[i] = DT<T>::ld(gamma + i);
    sa[C + i] = DT<T>::ld(beta + i);
    sa[2 * C + i] = mean[b * C + i];
    sa[3 * C + i] = rstd[b * C + i];
    for (int j = 0; j < k; ++j) sw[j * C + i] = DT<T>::ld(w + (int64_t)i * k + j);
  }
  const int span = (nt - 1) * stride + k;
  for (int i = threadIdx.x; i < span; i += blockDim.x) sx[i] = wav[b * S + t0 * stride + i];
  __syncthreads();
  const int cvecs = C / 8;
  // thread -> fixed channel vector, strided over frames so partial sums stay in registers
  const int tpc = blockDim.x / cvecs > 0 ? blockDim.x / cvecs : 1;  // threads per channel vector
  const int cv = threadIdx.x % cvecs, tsl = threadIdx.x / cvecs;
  const bool active = tsl < tpc;
  float s1[8], s2[8], r[KMAX][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.0f; s2[e] = 0.0f; }
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) r[j][e] = 0.0f;
  for (int ccv = cv; ccv < cvecs; ccv += blockDim.x) {  // (runs once: cvecs <= blockDim.x)
    for (int tl = tsl; active && tl < nt; tl += tpc) {
      float d[8], u[8];
      load8(dy + (b * L + t0 + tl) * C + ccv * 8, d);
#pragma unroll
      for (int e = 0; e < 8; ++e) u[e] = 0.0f;
      float xv[KMAX];
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        xv[j] = j < k ? sx[tl * stride + j] : 0.0f;
        if (j < k) {
#pragma unroll
          for (int e = 0; e < 8; ++e) u[e] += sw[j * C + ccv * 8 + e] * xv[j];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = ccv * 8 + e;
        const float uh = (u[e] - sa[2 * C + c]) * sa[3 * C + c];
        const float z = uh * sa[c] + sa[C + c];
        const float dz = d[e] * dgelu_t<T>(z);
        s1[e] += dz;
        s2[e] += dz * uh;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) r[j][e] += dz * xv[j];
      }
    }
    // combine the frame-lanes that share this channel vector through LDS (one 8-float row of the accumulator set at a
    // time, so the staging area is 8 KiB), then ONE store per value per block.
    float* o = ws + (b * gridDim.x + blockIdx.x) * (k + 2) * (int64_t)C + ccv * 8;
#pragma unroll
    for (int row = 0; row < KMAX + 2; ++row) {
      if (row < k + 2) {  // wave-uniform
        __syncthreads();
        float* mine = sm + (size_t)threadIdx.x * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) mine[e] = row == 0 ? s1[e] : (row == 1 ? s2[e] : r[row >= 2 ? row - 2 : 0][e]);
        __syncthreads();
        if (tsl == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = 0.0f;
            for (int t2 = 0; t2 < tpc; ++t2) v += sm[(size_t)(t2 * cvecs + ccv) * 8 + e];
            o[(int64_t)row * C + e] = v;
          }
        }
      }
    }
  }
}

// =====================================================================================================================
// Register-resident variants (k == KC compile-time, C % 4 == 0, C/4 divides 256): every thread owns FOUR channels and keeps
// their KC filter taps (and, in backward, the KC+2 accumulators per channel) in VGPRs; the only LDS traffic in the frame loop
// is the broadcast read of the wave span.  ~2.5x fewer LDS instructions and ~half the VGPRs of the 8-channel kernels above.
// =====================================================================================================================
template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <>
__device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
}
template <>
__device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
  const uint2 r = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
  const f32x4 a = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p) = a;
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
  using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  bf16x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = static_cast<__bf16>(v[e]);
  // (layer 0's output is 2-3 GB per batch, written once and far larger than the caches: streaming store)
  __builtin_nontemporal_store(__builtin_bit_cast(u32x2_t, r), reinterpret_cast<u32x2_t*>(p));
}

constexpr int C0R_TB = 128;  // frames per block (forward)

template <typename T, int KC>
__global__ __launch_bounds__(256) void conv0_fwd_reg_kernel(const float* wav, const T* w, const T* gamma, const T* beta,
                                                            const float* mean, const float* rstd, T* y, int64_t S, int64_t L,
                                                            int C, int stride, const int32_t* flim) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sx = sm;
  const int64_t b = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * C0R_TB;
  const int64_t Lb = (flim && flim[b] < L) ? flim[b] : L;  // frames from flim[b] on are read by nobody: not computed, not stored
  if (t0 >= Lb) return;
  const int nt = (int)((Lb - t0 < C0R_TB) ? (Lb - t0) : C0R_TB);
  const int tpf = C / 4, fp = 256 / tpf;       // threads per frame, frames in flight
  const int cq = threadIdx.x % tpf, fl = threadIdx.x / tpf;
  const int c0 = cq * 4;
  float wf[KC][4], bb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float a = DT<T>::ld(gamma + c0 + e) * rstd[b * C + c0 + e];
    bb[e] = DT<T>::ld(beta + c0 + e) - mean[b * C + c0 + e] * a;
#pragma unroll
    for (int j = 0; j < KC; ++j) wf[j][e] = a * DT<T>::ld(w + (int64_t)(c0 + e) * KC + j);
  }
  const int span = (nt - 1) * stride + KC;
  for (int i = threadIdx.x; i < span; i += 256) sx[i] = wav[b * S + t0 * stride + i];
  __syncthreads();
  for (int tl = fl; tl < nt; tl += fp) {
    float acc[4] = {bb[0], bb[1], bb[2], bb[3]};
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      const float xv = sx[tl * stride + j];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(wf[j][e], xv, acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = gelu_t<T>(acc[e]);
    store4(y + (b * L + t0 + tl) * C + c0, acc);
  }
}

constexpr int C0R_BWD_TB = 2048;  // frames per block (backward)

template <typename T, int KC>
__global__ __launch_bounds__(256) void conv0_bwd_reg_kernel(const T* dy, const float* wav, const T* w, const T* gamma,
                                                            const T* beta, const float* mean, const float* rstd, float* ws,
                                                            int64_t S, int64_t L, int C, int stride, const int32_t* blim) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sx = sm;
  const int64_t b = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * C0R_BWD_TB;
  const int64_t Lb = (blim && blim[b] < L) ? blim[b] : L;  // dy is exactly zero from frame blim[b] on: those frames add nothing
  const int64_t left = Lb - t0;
  const int nt = left <= 0 ? 0 : (int)(left < C0R_BWD_TB ? left : C0R_BWD_TB);  // 0: this block only writes its zero partials
  const int tpf = C / 4, fp = 256 / tpf;
  const int cq = threadIdx.x % tpf, fl = threadIdx.x / tpf;
  const int c0 = cq * 4;
  float wr[KC][4], g[4], be[4], mu[4], rs[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    g[e] = DT<T>::ld(gamma + c0 + e);
    be[e] = DT<T>::ld(beta + c0 + e);
    mu[e] = mean[b * C + c0 + e];
    rs[e] = rstd[b * C + c0 + e];
#pragma unroll
    for (int j = 0; j < KC; ++j) wr[j][e] = DT<T>::ld(w + (int64_t)(c0 + e) * KC + j);
  }
  const int span = nt > 0 ? (nt - 1) * stride + KC : 0;
  for (int i = threadIdx.x; i < span; i += 256) sx[i] = wav[b * S + t0 * stride + i];
  __syncthreads();
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, r[KC][4];
#pragma unroll
  for (int j = 0; j < KC; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) r[j][e] = 0.0f;
  for (int tl = fl; tl < nt; tl += fp) {
    float d[4], xv[KC], u[4] = {0, 0, 0, 0};
    load4<T>(dy + (b * L + t0 + tl) * C + c0, d);
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      xv[j] = sx[tl * stride + j];
#pragma unroll
      for (int e = 0; e < 4; ++e) u[e] = fmaf(wr[j][e], xv[j], u[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float uh = (u[e] - mu[e]) * rs[e];
      const float dz = d[e] * dgelu_t<T>(fmaf(uh, g[e], be[e]));
      s1[e] += dz;
      s2[e] = fmaf(dz, uh, s2[e]);
#pragma unroll
      for (int j = 0; j < KC; ++j) r[j][e] = fmaf(dz, xv[j], r[j][e]);
    }
  }
  // combine the `fp` frame-lanes of each channel quad through LDS (4 floats per thread per round), one store per value
  float* o = ws + (b * gridDim.x + blockIdx.x) * (KC + 2) * (int64_t)C + c0;
#pragma unroll
  for (int row = 0; row < KC + 2; ++row) {
    __syncthreads();
    float* mine = sm + (size_t)threadIdx.x * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) mine[e] = row == 0 ? s1[e] : (row == 1 ? s2[e] : r[row >= 2 ? row - 2 : 0][e]);
    __syncthreads();
    if (fl == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = 0.0f;
        for (int t2 = 0; t2 < fp; ++t2) v += sm[(size_t)(t2 * tpf + cq) * 4 + e];
        o[(int64_t)row * C + e] = v;
      }
    }
  }
}

// ---- the backward pass with both contractions on the matrix cores (bf16 storage, k = 10, C = 512) -----------------------------
// The register kernel above is bound by VALU issue: 38 lane operations per output element, 20 of them the two 10-tap contractions
// (u = sum_j w[c][j] x[s t + j] recomputed, r[j][c] += dz x[s t + j]).  Here both are bf16 MFMAs (v_mfma_f32_16x16x32_bf16, fp32
// accumulate) and the VALU keeps GroupNorm + GELU' + the dz rounding (14 operations per element, packed two frames per instruction):
//   u  : M = 16 frames, N = 16 channels, K = 32 taps (10 real: the weight operand is zero beyond).  The fp32 wave enters as hi + lo
//        bf16 halves (hi = upper 16 bits, lo = bf16(x - hi): x = hi + lo to 2^-17 — two MFMAs), the weights are bf16 as stored.
//        D: lane (q = l & 15, g = l >> 4) holds channel q of the set, frames 4g .. 4g + 3 of the 16-frame tile.
//   dz : VALU on those values (the lane owns 4 consecutive channels 4q + e of a 64-channel block: dy arrives as 8-byte loads);
//        s2 = rstd * sum dz (u - mean) is one packed fma.
//   r  : v_mfma_f32_16x16x16_bf16, M = 16 "taps" (row 10 is a row of ones: its sum is s1 = sum dz), N = 16 channels, K = 16 frames
//        (k = 4g + i: exactly the frames u left in the lane — no exchange, no staging).  dz enters rounded to bf16 — the
//        precision every other weight gradient of the bf16 path has (they are GEMMs over bf16-stored gradients) — the wave again
//        as hi + lo.  D: lane holds channel q, tap rows 4g .. 4g + 3.
// (First built with r on v_mfma_f32_16x16x4_f32, exact fp32: correct, and no faster than the register kernel — counters showed
// VALU-busy + MFMA-busy = 87 % of the kernel's cycles: the fp32-input MFMA runs at the fp32 vector rate and does not overlap the VALU.)
// Block = 8 waves over 2048 frames, one 64-channel block per wave (no cross-wave sums); 4 waves per SIMD (two blocks per CU).
// Partials have the layout of the register kernel ([B][blocks][k + 2][C]): the reduce and finish kernels are shared.
constexpr int CONV0_DY_AUX = 2;  // nt: the gradient is read once (2-3 GB per batch)
constexpr int C0M_PAD = 96;  // floats behind the staged wave: zeros[32] (operand reads past the last real frame) | ones[16] | zeros[48]
__global__ __launch_bounds__(512, 4) void conv0_bwd_mfma_kernel(const bf16_t* dy, const float* wav, const bf16_t* w, const bf16_t* gamma,
                                                                const bf16_t* beta, const float* mean, const float* rstd, float* ws,
                                                                int64_t S, int64_t L, int stride, const int32_t* blim) {
  constexpr int KC = 10, C = 512;
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int span_max = (C0R_BWD_TB - 1) * stride + KC;
  float* const sx = sm;                                   // [span_max + C0M_PAD]: wave | zeros[32] | ones[16] | zeros[48]
  const int ones_at = span_max + 32, zeros_at = span_max + 48;
  const int sx_floats = (span_max + C0M_PAD + 3) & ~3;
  bf16_t* const wl = reinterpret_cast<bf16_t*>(sm + sx_floats);                       // [C][16] taps, zero beyond KC
  float* const pr = sm + sx_floats + C * 8;                                           // [C][4]: mean, rstd * gamma, beta, rstd
  const int64_t b = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * C0R_BWD_TB;
  const int64_t Lb = (blim && blim[b] < L) ? blim[b] : L;
  const int64_t left = Lb - t0;
  const int nt = left <= 0 ? 0 : (int)(left < C0R_BWD_TB ? left : C0R_BWD_TB);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane & 15, g = lane >> 4;
  const int cbase = wave * 64;
  // ---- stage: wave span (zero behind it), weights, per-channel constants ----
  const int span = nt > 0 ? (nt - 1) * stride + KC : 0;
  for (int i = tid; i < span_max + C0M_PAD; i += 512)
    sx[i] = i < span ? wav[b * S + t0 * stride + i] : ((i >= ones_at && i < zeros_at) ? 1.0f : 0.0f);
  for (int i = tid; i < C * 16; i += 512) {
    const int c = i >> 4, j = i & 15;
    wl[i] = j < KC ? w[c * KC + j] : static_cast<bf16_t>(0.0f);
  }
  for (int c = tid; c < C; c += 512) {
    const float rs = rstd[b * C + c];
    pr[4 * c + 0] = mean[b * C + c];
    pr[4 * c + 1] = rs * static_cast<float>(gamma[c]);
    pr[4 * c + 2] = static_cast<float>(beta[c]);
    pr[4 * c + 3] = rs;
  }
  __syncthreads();
  f32x4 R[4];
  v2f s2p[4];  // sum dz (u - mean): even / odd frames of the lane's (added in that order at the end)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    R[e] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    s2p[e] = v2f{0.0f, 0.0f};
  }
  const int nsteps = (nt + 15) >> 4;  // 16 frames per step
  // dy of (frame ts + 4g + i, channels cbase + 4q .. + 3): one 8-byte BUFFER load each — the descriptor ends behind the block's last
  // real frame, so rows beyond nt (the block's partial last step, the prefetch behind the last step) come back as zeros through the
  // range check: dz = 0, nothing is added, no clamps and no tail code in the loop.
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (b * L + t0) * (int64_t)C), (short)0, nt * C * 2, 0x00020000);
  const int dy_lane = (4 * g * C + cbase + 4 * q) * 2;
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  typedef short s16x4_t __attribute__((ext_vector_type(4)));
  u32x2_t dn[4];
  auto load_dy = [&](int st) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dn[i] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rdy, dy_lane + i * C * 2, st * 16 * C * 2, CONV0_DY_AUX));
  };
  // x = hi + lo of two neighbouring values: hi = the upper 16 bits (exact in bf16; one byte permute packs two), lo = bf16(x - hi)
  auto split2 = [](float x0, float x1) -> uint2 {
    const unsigned b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
    const v2f lo = v2f{x0, x1} - v2f{__uint_as_float(b0 & 0xffff0000u), __uint_as_float(b1 & 0xffff0000u)};
    return uint2{__builtin_amdgcn_perm(b1, b0, 0x07060302u), __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_t))};
  };
  // r operand: tap row m = q: x[s (ts + 4g + i) + m]; row 10 reads the strip of ones, rows 11 .. 15 the strip of zeros
  const int r_off = q < KC ? (4 * g * stride + q) : (q == KC ? ones_at : zeros_at);
  const int r_mul = q < KC ? 16 * stride : 0;
  const int r_inc = q < KC ? stride : 0;
  // per-lane LDS bases: what depends on e is a compile-time offset from them
  // (lanes g >= 2 carry taps 16 .. 31 of the u MFMA: their weight operand comes from the strip of zeros)
  const bf16_t* const wl_lane = g < 2 ? wl + (cbase + 4 * q) * 16 + 8 * g : reinterpret_cast<const bf16_t*>(sx + zeros_at);
  const float* const pr_lane = pr + 4 * (cbase + 4 * q);
  load_dy(0);
  for (int st = 0; st < nsteps; ++st) {
    u32x2_t dc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dc[i] = dn[i];
    load_dy(st + 1);  // the next step's rows travel under this step's arithmetic
    // u operand: frame row q of the tile, taps 8 (g & 1) .. + 7;  r operand: tap row q over the lane's frames 4g .. 4g + 3
    u32x4 hw, lw;
    u32x2_t rh, rl;
    {
      const float* xs = sx + (st * 16 + q) * stride + 8 * (g & 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint2 hl = split2(xs[2 * j], xs[2 * j + 1]);
        hw[j] = hl.x;
        lw[j] = hl.y;
      }
      const float* xr = sx + r_off + st * r_mul;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const uint2 hl = split2(xr[2 * jp * r_inc], xr[(2 * jp + 1) * r_inc]);
        rh[jp] = hl.x;
        rl[jp] = hl.y;
      }
    }
    const bf16x8 ah = __builtin_bit_cast(bf16x8, hw), al = __builtin_bit_cast(bf16x8, lw);
    const s16x4_t rah = __builtin_bit_cast(s16x4_t, rh), ral = __builtin_bit_cast(s16x4_t, rl);
#pragma unroll
    for (int ep = 0; ep < 4; ep += 2) {
      f32x4 u[2];  // the four MFMAs of two channel sets first: the VALU below meets finished results
#pragma unroll
      for (int eo = 0; eo < 2; ++eo) {
        const bf16x8 bw = *reinterpret_cast<const bf16x8*>(wl_lane + (ep + eo) * 16);
        u[eo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bw, f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
        u[eo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bw, u[eo], 0, 0, 0);
      }
#pragma unroll
      for (int eo = 0; eo < 2; ++eo) {
        const int e = ep + eo;
        const f32x4 prm = *reinterpret_cast<const f32x4*>(pr_lane + 4 * e);
        const v2f mu2 = {prm[0], prm[0]}, rg2 = {prm[1], prm[1]}, be2 = {prm[2], prm[2]};
        // the lane's two frame pairs walk the polynomial side by side (a dependent packed fma right behind its producer waits)
        v2f d[2], t[2], xc[2], x2[2], p[2];
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
          const unsigned w0 = e < 2 ? dc[2 * ip][0] : dc[2 * ip][1], w1 = e < 2 ? dc[2 * ip + 1][0] : dc[2 * ip + 1][1];
          d[ip] = v2f{__uint_as_float((e & 1) ? (w0 & 0xffff0000u) : (w0 << 16)), __uint_as_float((e & 1) ? (w1 & 0xffff0000u) : (w1 << 16))};
          t[ip] = v2f{u[eo][2 * ip], u[eo][2 * ip + 1]} - mu2;
        }
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
          const v2f a = __builtin_elementwise_fma(t[ip], rg2, be2);
          xc[ip] = v2f{__builtin_amdgcn_fmed3f(a[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(a[1], -4.0f, 4.0f)};
        }
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) x2[ip] = xc[ip] * xc[ip];
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
          p[ip] = __builtin_elementwise_fma(v2f{-1.5577683143419563e-08f, -1.5577683143419563e-08f}, x2[ip], v2f{1.1633505891950335e-06f, 1.1633505891950335e-06f});
        constexpr float PC[6] = {-3.7250658351695165e-05f, 0.0006728997686877847f, -0.0075911665335297585f, 0.0555923730134964f,
                                 -0.26155415177345276f, 0.7965189218521118f};  // (dgelu_poly_f, cst_common.h)
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
          for (int ip = 0; ip < 2; ++ip) p[ip] = __builtin_elementwise_fma(p[ip], x2[ip], v2f{PC[k], PC[k]});
        u32x2_t zb;
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
          const v2f gp = __builtin_elementwise_fma(p[ip], xc[ip], v2f{0.5f, 0.5f});
          const v2f z = d[ip] * gp;
          s2p[e] = __builtin_elementwise_fma(z, t[ip], s2p[e]);
          zb[ip] = __builtin_bit_cast(unsigned, __builtin_convertvector(z, bf16x2_t));
        }
        // r: K = 16 frames = this tile (v_mfma_f32_16x16x16_bf16: k = 4g + i is exactly the frame layout u left in the lane)
        const s16x4_t bz = __builtin_bit_cast(s16x4_t, zb);
        R[e] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(rah, bz, R[e], 0, 0, 0);
        R[e] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ral, bz, R[e], 0, 0, 0);
      }
    }
  }
  // ---- s2 over the four frame groups of the wave (lane exchange, g order); every wave owns its 64 channels: no cross-wave sums ----
  float* const o = ws + (b * gridDim.x + blockIdx.x) * (KC + 2) * (int64_t)C;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = cbase + 4 * q + e;
    const float v = s2p[e][0] + s2p[e][1];
    const float v1 = __shfl(v, q + 16, 64), v2 = __shfl(v, q + 32, 64), v3 = __shfl(v, q + 48, 64);
    if (g == 0) o[C + c] = (((v + v1) + v2) + v3) * pr[4 * c + 3];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int tap = 4 * g + rr;                       // rows 0 .. 9: r[tap]; row 10: s1
      if (tap <= KC) o[(tap == KC ? 0 : 2 + tap) * C + c] = R[e][rr];
    }
  }
}

// fixed-order sum of an utterance's block partials: part [B][nblk][rows][C] -> acc [B][rows][C]; grid (B, rows), one thread per channel
__global__ void conv0_bwd_reduce_kernel(const float* part, float* acc, int nblk, int rows, int C) {
  const int64_t b = blockIdx.x;
  const int row = blockIdx.y;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = 0.0f;
    for (int q = 0; q < nblk; ++q) a += part[((b * nblk + q) * rows + row) * (int64_t)C + c];
    acc[(b * rows + row) * (int64_t)C + c] = a;
  }
}

// closed-form finish: dW[c][j], dgamma[c], dbeta[c] summed over utterances.
//   du = (gamma rstd) (dz - s1/L - uhat s2/L);  dW[c][j] = sum_t du x[s t + j]
//   sum_t uhat x_j = rstd (sum_j' w[c][j'] G[j'][j] - mean m[j])
// One thread per (utterance, channel) evaluates its term in double (the k x k contraction: ~1000 fp64 operations); the utterances of
// a channel are then added in index order through LDS.  (The first version looped the utterances inside one thread per channel:
// 512 threads on the whole chip, 465 us per call.)  Block = FB utterances x FC channels.
constexpr int FB = 32, FC = 8;
template <typename T>
__global__ __launch_bounds__(FB * FC) void conv0_bwd_finish_kernel(const float* ws, const float* gram, const T* w, const T* gamma, const float* mean,
                                                                    const float* rstd, float* dw, float* dgamma, float* dbeta, int64_t B, int64_t C,
                                                                    int64_t L, int k) {
  __shared__ double red[FB][FC][KMAX + 2];
  const int cl = threadIdx.x % FC, bl = threadIdx.x / FC;
  const int64_t c = (int64_t)blockIdx.x * FC + cl;
  double tot[KMAX + 2];
  for (int j = 0; j < k + 2; ++j) tot[j] = 0.0;
  for (int64_t b0 = 0; b0 < B; b0 += FB) {
    const int64_t b = b0 + bl;
    double acc[KMAX + 2];
    for (int j = 0; j < k + 2; ++j) acc[j] = 0.0;
    if (b < B && c < C) {
      const double gam = (double)DT<T>::ld(gamma + c);
      const float* a = ws + b * (k + 2) * C;
      const float* g = gram + b * (k * k + k);
      const double s1 = a[c], s2 = a[C + c], rs = rstd[b * C + c], mu = mean[b * C + c];
      acc[0] = s2;  // dgamma
      acc[1] = s1;  // dbeta
      for (int j = 0; j < k; ++j) {
        double wg = 0.0;
        for (int j2 = 0; j2 < k; ++j2) wg += (double)DT<T>::ld(w + c * k + j2) * (double)g[j2 * k + j];
        const double uhx = rs * (wg - mu * (double)g[k * k + j]);
        acc[2 + j] = gam * rs * ((double)a[(2 + j) * C + c] - s1 / (double)L * (double)g[k * k + j] - s2 / (double)L * uhx);
      }
    }
    for (int j = 0; j < k + 2; ++j) red[bl][cl][j] = acc[j];
    __syncthreads();
    if (bl == 0)
      for (int j = 0; j < k + 2; ++j)
        for (int q = 0; q < FB; ++q) tot[j] += red[q][cl][j];  // utterances in index order: deterministic
    __syncthreads();
  }
  if (bl == 0 && c < C) {
    dgamma[c] = (float)tot[0];
    dbeta[c] = (float)tot[1];
    for (int j = 0; j < k; ++j) dw[c * k + j] = (float)tot[2 + j];
  }
}

}  // namespace

static int conv0_gram_blocks(int64_t L) { return (int)(cst_ceil_div(L, 256 * 8) < 64 ? cst_ceil_div(L, 256 * 8) : 64); }

extern "C" int64_t cst_conv0_fwd_workspace(int64_t B, int64_t S, int k, int stride) {
  const int64_t L = S >= k ? (S - k) / stride + 1 : 0;
  return B * (int64_t)conv0_gram_blocks(L) * (k * k + k) * (int64_t)sizeof(float);
}

extern "C" int cst_conv0_gn_gelu_fwd(const float* wav, const void* w, const void* gamma, const void* beta, void* y,
                                     float* mean, float* rstd, float* gram, float* workspace, const int32_t* frame_limit, int64_t B,
                                     int64_t S, int64_t C, int k, int stride, float eps, int dtype, cst_stream stream) {
  CST_REQUIRE(wav && w && gamma && beta && y && mean && rstd && gram && workspace, "cst_conv0_gn_gelu_fwd: null tensor");
  CST_REQUIRE(k >= 1 && k <= KMAX && stride >= 1 && S >= k, "cst_conv0_gn_gelu_fwd: unsupported k=%d stride=%d S=%lld", k, stride, (long long)S);
  CST_REQUIRE(C % 8 == 0 && C >= 8 && C / 8 <= 256, "cst_conv0_gn_gelu_fwd: C=%lld must be a multiple of 8 and <= 2048", (long long)C);
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_conv0_gn_gelu_fwd: bad dtype %d", dtype);
  const int64_t L = (S - k) / stride + 1;
  hipStream_t s = (hipStream_t)stream;
  const double bytes = (double)B * S * 4.0 * 2.0 + (double)B * L * C * cst_dtype_size(dtype);
  CstProfScope prof(CST_K_CONV0, s, 2.0 * (double)B * L * C * k, bytes);
  const int gb = conv0_gram_blocks(L);
  if (k == 10) hipLaunchKernelGGL(conv0_gram_kernel<10>, dim3(gb, (unsigned)B), dim3(256), 0, s, wav, workspace, S, L, k, stride);
  else hipLaunchKernelGGL(conv0_gram_kernel<KMAX>, dim3(gb, (unsigned)B), dim3(256), 0, s, wav, workspace, S, L, k, stride);
  const size_t lds = sizeof(float) * ((size_t)k * C + C + (size_t)C0_TB * stride + k);
  dim3 sg((unsigned)cst_ceil_div(C, 128), (unsigned)B), fg((unsigned)cst_ceil_div(L, C0_TB), (unsigned)B);
  const bool reg_path = k == 10 && C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0;
  const size_t lds_r = sizeof(float) * ((size_t)C0R_TB * stride + k);
  dim3 fgr((unsigned)cst_ceil_div(L, C0R_TB), (unsigned)B);
  if (dtype == CST_BF16) {
    hipLaunchKernelGGL(conv0_stats_kernel<bf16_t>, sg, dim3(128), 0, s, (const bf16_t*)w, workspace, gb, gram, mean, rstd, C, L, k, eps);
    if (reg_path) hipLaunchKernelGGL((conv0_fwd_reg_kernel<bf16_t, 10>), fgr, dim3(256), lds_r, s, wav, (const bf16_t*)w, (const bf16_t*)gamma, (const bf16_t*)beta, mean, rstd, (bf16_t*)y, S, L, (int)C, stride, frame_limit);
    else hipLaunchKernelGGL(conv0_fwd_kernel<bf16_t>, fg, dim3(256), lds, s, wav, (const bf16_t*)w, (const bf16_t*)gamma, (const bf16_t*)beta, mean, rstd, (bf16_t*)y, S, L, (int)C, k, stride);
  } else {
    hipLaunchKernelGGL(conv0_stats_kernel<float>, sg, dim3(128), 0, s, (const float*)w, workspace, gb, gram, mean, rstd, C, L, k, eps);
    if (reg_path) hipLaunchKernelGGL((conv0_fwd_reg_kernel<float, 10>), fgr, dim3(256), lds_r, s, wav, (const float*)w, (const float*)gamma, (const float*)beta, mean, rstd, (float*)y, S, L, (int)C, stride, frame_limit);
    else hipLaunchKernelGGL(conv0_fwd_kernel<float>, fg, dim3(256), lds, s, wav, (const float*)w, (const float*)gamma, (const float*)beta, mean, rstd, (float*)y, S, L, (int)C, k, stride);
  }
  return cst_check_launch("cst_conv0_gn_gelu_fwd");
}

static int conv0_bwd_blocks(int64_t L, int k, int64_t C) {
  const bool reg_path = k == 10 && C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0;
  return (int)cst_ceil_div(L, reg_path ? C0R_BWD_TB : C0_BWD_TB);
}

/* [B][nblk][k+2][C] block partials followed by the reduced [B][k+2][C] */
extern "C" int64_t cst_conv0_bwd_workspace(int64_t B, int64_t S, int64_t C, int k, int stride) {
  const int64_t L = S >= k ? (S - k) / stride + 1 : 0;
  return B * (int64_t)(conv0_bwd_blocks(L, k, C) + 1) * (k + 2) * C * (int64_t)sizeof(float);
}

extern "C" int cst_conv0_gn_gelu_bwd(const void* dy, const float* wav, const void* w, const void* gamma, const void* beta,
                                     const float* mean, const float* rstd, const float* gram, float* dw, float* dgamma,
                                     float* dbeta, float* workspace, const int32_t* frame_limit, int64_t B, int64_t S, int64_t C, int k,
                                     int stride, int dtype, cst_stream stream) {
  CST_REQUIRE(dy && wav && w && gamma && beta && mean && rstd && gram && dw && dgamma && dbeta && workspace, "cst_conv0_gn_gelu_bwd: null tensor");
  CST_REQUIRE(k >= 1 && k <= KMAX && stride >= 1 && S >= k, "cst_conv0_gn_gelu_bwd: unsupported k=%d stride=%d", k, stride);
  CST_REQUIRE(C % 8 == 0 && C >= 8 && C / 8 <= 256, "cst_conv0_gn_gelu_bwd: C=%lld must be a multiple of 8 and <= 2048", (long long)C);
  CST_REQUIRE(dtype == CST_F32 || dtype == CST_BF16, "cst_conv0_gn_gelu_bwd: bad dtype %d", dtype);
  const int64_t L = (S - k) / stride + 1;
  hipStream_t s = (hipStream_t)stream;
  const double bytes = (double)B * S * 4.0 + (double)B * L * C * cst_dtype_size(dtype);
  CstProfScope prof(CST_K_CONV0, s, 4.0 * (double)B * L * C * k, bytes);
  const int nblk = conv0_bwd_blocks(L, k, C);
  float* acc = workspace + (size_t)B * nblk * (k + 2) * C;  // the reduced [B][k+2][C] behind the partials
  const dim3 rg((unsigned)B, (unsigned)(k + 2));
  size_t lds = sizeof(float) * ((size_t)k * C + 4 * C + (size_t)C0_BWD_TB * stride + k);
  dim3 grid((unsigned)cst_ceil_div(L, C0_BWD_TB), (unsigned)B), fg((unsigned)cst_ceil_div(C, FC));
  const bool reg_path = k == 10 && C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0;
  if (reg_path) {
    size_t lds_r = sizeof(float) * ((size_t)C0R_BWD_TB * stride + k);
    if (lds_r < sizeof(float) * 256 * 4) lds_r = sizeof(float) * 256 * 4;
    dim3 gr((unsigned)cst_ceil_div(L, C0R_BWD_TB), (unsigned)B);
    static const bool no_mfma = getenv("CST_CONV0_NO_MFMA") != nullptr;  // (A/B switch: the register kernel)
    if (dtype == CST_BF16 && C == 512 && !no_mfma) {
      const int span_max = (C0R_BWD_TB - 1) * stride + 10;
      const size_t lds_m = sizeof(float) * (size_t)(((span_max + C0M_PAD + 3) & ~3) + 512 * 8 + 512 * 4);
      hipLaunchKernelGGL(conv0_bwd_mfma_kernel, gr, dim3(512), lds_m, s, (const bf16_t*)dy, wav, (const bf16_t*)w, (const bf16_t*)gamma, (const bf16_t*)beta, mean, rstd, workspace, S, L, stride, frame_limit);
      hipLaunchKernelGGL(conv0_bwd_reduce_kernel, rg, dim3(256), 0, s, workspace, acc, nblk, k + 2, (int)C);
      hipLaunchKernelGGL(conv0_bwd_finish_kernel<bf16_t>, fg, dim3(FB * FC), 0, s, acc, gram, (const bf16_t*)w, (const bf16_t*)gamma, mean, rstd, dw, dgamma, dbeta, B, C, L, k);
    } else if (dtype == CST_BF16) {
      hipLaunchKernelGGL((conv0_bwd_reg_kernel<bf16_t, 10>), gr, dim3(256), lds_r, s, (const bf16_t*)dy, wav, (const bf16_t*)w, (const bf16_t*)gamma, (const bf16_t*)beta, mean, rstd, workspace, S, L, (int)C, stride, frame_limit);
      hipLaunchKernelGGL(conv0_bwd_reduce_kernel, rg, dim3(256), 0, s, workspace, acc, nblk, k + 2, (int)C);
      hipLaunchKernelGGL(conv0_bwd_finish_kernel<bf16_t>, fg, dim3(FB * FC), 0, s, acc, gram, (const bf16_t*)w, (const bf16_t*)gamma, mean, rstd, dw, dgamma, dbeta, B, C, L, k);
    } else {
      hipLaunchKernelGGL((conv0_bwd_reg_kernel<float, 10>), gr, dim3(256), lds_r, s, (const float*)dy, wav, (const float*)w, (const float*)gamma, (const float*)beta, mean, rstd, workspace, S, L, (int)C, stride, frame_limit);
      hipLaunchKernelGGL(conv0_bwd_reduce_kernel, rg, dim3(256), 0, s, workspace, acc, nblk, k + 2, (int)C);
      hipLaunchKernelGGL(conv0_bwd_finish_kernel<float>, fg, dim3(FB * FC), 0, s, acc, gram, (const float*)w, (const float*)gamma, mean, rstd, dw, dgamma, dbeta, B, C, L, k);
    }
    return cst_check_launch("cst_conv0_gn_gelu_bwd");
  }
  if (dtype == CST_BF16) {
    hipLaunchKernelGGL(conv0_bwd_kernel<bf16_t>, grid, dim3(256), lds, s, (const bf16_t*)dy, wav, (const bf16_t*)w, (const bf16_t*)gamma, (const bf16_t*)beta, mean, rstd, workspace, S, L, (int)C, k, stride);
    hipLaunchKernelGGL(conv0_bwd_reduce_kernel, rg, dim3(256), 0, s, workspace, acc, nblk, k + 2, (int)C);
    hipLaunchKernelGGL(conv0_bwd_finish_kernel<bf16_t>, fg, dim3(FB * FC), 0, s, acc, gram, (const bf16_t*)w, (const bf16_t*)gamma, mean, rstd, dw, dgamma, dbeta, B, C, L, k);
  } else {
    hipLaunchKernelGGL(conv0_bwd_kernel<float>, grid, dim3(256), lds, s, (const float*)dy, wav, (const float*)w, (const float*)gamma, (const float*)beta, mean, rstd, workspace, S, L, (int)C, k, stride);
    hipLaunchKernelGGL(conv0_bwd_reduce_kernel, rg, dim3(256), 0, s, workspace, acc, nblk, k + 2, (int)C);
    hipLaunchKernelGGL(conv0_bwd_finish_kernel<float>, fg, dim3(FB * FC), 0, s, acc, gram, (const float*)w, (const float*)gamma, mean, rstd, dw, dgamma, dbeta, B, C, L, k);
  }
  return cst_check_launch("cst_conv0_gn_gelu_bwd");
}

// ---- live-frame limits of the whole conv stack (include/cst.h: cst_conv_row_limits) -----------------------------------------------
namespace {
struct ConvSpec8 { int k[8], s[8], len[8]; };
__global__ void conv_row_limits_kernel(const int32_t* nz_last, ConvSpec8 sp, int L, int smax, int32_t* out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int nz[8];
  int cur = nz_last[b];
  for (int i = L - 1; i >= 1; --i) {
    cur = cur < sp.len[i] ? cur : sp.len[i];
    nz[i] = cur;
    cur = cur > 0 ? (cur - 1) * sp.s[i] + sp.k[i] : 0;
  }
  nz[0] = cur < sp.len[0] ? cur : sp.len[0];
  const int64_t plane = (int64_t)(1 + smax) * B;
  for (int i = 0; i < L; ++i) {
    out[i * plane + b] = nz[i];
    for (int r = 0; r < smax; ++r) {
      int v = 0;
      if (i >= 1 && r < sp.s[i]) {
        const int d = nz[i - 1] - r;
        v = d > 0 ? (d + sp.s[i] - 1) / sp.s[i] : 0;
      } else if (i == 0 && r == 0 && L >= 2) {  // rows of layer 0 under layer 1's live GEMM tiles (tiles start at multiples of BM per batch)
        constexpr int BM = cstg::GEMM_MAX_BM;
        const int64_t need = (int64_t)((nz[1] + BM - 1) / BM) * BM * sp.s[1] + sp.k[1];
        v = need < sp.len[0] ? (int)need : sp.len[0];
      } else if (i == 0 && r == 0) {
        v = sp.len[0];
      }
      out[i * plane + (int64_t)(1 + r) * B + b] = v;
    }
  }
}
}  // namespace

extern "C" int cst_conv_row_limits(const int32_t* nz_last, const int32_t* k, const int32_t* stride, int L, int64_t S, int32_t* out,
                                   int64_t B, int smax, cst_stream stream) {
  CST_REQUIRE(nz_last && k && stride && out && B > 0 && L >= 1 && L <= 8 && smax >= 1 && smax <= 8, "cst_conv_row_limits: bad args");
  ConvSpec8 sp;
  int64_t len = S;
  for (int i = 0; i < 8; ++i) {
    sp.k[i] = i < L ? k[i] : 1;
    sp.s[i] = i < L ? stride[i] : 1;
    CST_REQUIRE(sp.k[i] >= 1 && sp.s[i] >= 1 && sp.s[i] <= 8 && (i == 0 || i >= L || sp.s[i] <= smax), "cst_conv_row_limits: bad layer %d", i);
    if (i < L) len = len >= sp.k[i] ? (len - sp.k[i]) / sp.s[i] + 1 : 0;
    sp.len[i] = (int)len;
  }
  hipLaunchKernelGGL(conv_row_limits_kernel, dim3((unsigned)cst_ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, nz_last, sp, L, smax, out, (int)B);
  return cst_check_launch("cst_conv_row_limits");
}
