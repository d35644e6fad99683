// VALU issue cost probe (gfx950): counter ticks per instruction for the integer / transcendental ops used by the dropout hash
// (one wave per SIMD, 16 independent chains: throughput, not latency).  s_memtime ticks are a fixed-frequency clock; the
// v_fma_f32 row (known: 4 core cycles per wave64 instruction) calibrates the others.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2048
#define CH 16
template <int OP>
__global__ void k(unsigned* out, long long* cyc, unsigned seed) {
  unsigned a[CH]; float f[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { a[j] = threadIdx.x * 2654435761u + seed + j * 977u; f[j] = 1.0f + (a[j] & 0xffff) * 1e-6f; }
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      if (OP == 0) a[j] = a[j] * 0x9E3779B1u + i;                 // v_mul_lo_u32 + v_add (or v_mad_u64_u32?)
      if (OP == 1) a[j] = __umul24(a[j], 0x9E3779u) + i;          // v_mad_u32_u24
      if (OP == 2) a[j] = (a[j] ^ (a[j] >> 15)) + i;              // v_lshrrev + v_xad
      if (OP == 3) f[j] = __builtin_amdgcn_exp2f(f[j]) - 1.5f;     // v_exp_f32 + v_add
      if (OP == 4) f[j] = fmaf(f[j], 1.0001f, 0.5f);               // v_fma_f32
      if (OP == 5) a[j] = __umulhi(a[j], 0x9E3779B1u) + i;         // v_mul_hi_u32 + v_add
      if (OP == 6) f[j] = __builtin_amdgcn_rcpf(f[j]) + 1.5f;      // v_rcp_f32 + v_add
    }
  }
  long long t1 = __builtin_readcyclecounter();
  unsigned r = 0; float g = 0;
#pragma unroll
  for (int j = 0; j < CH; ++j) { r ^= a[j]; g += f[j]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r ^ __float_as_uint(g);
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[OP] = t1 - t0;
}
int main() {
  unsigned* out; long long* cyc;
  hipMalloc(&out, 1 << 20); hipMallocManaged(&cyc, 64);
  const char* names[] = {"v_mul_lo_u32 (+add)", "v_mad_u32_u24", "v_lshrrev + v_xad (2 ops)", "v_exp_f32 + v_add (2 ops)", "v_fma_f32", "v_mul_hi_u32 + v_add (2 ops)", "v_rcp_f32 + v_add (2 ops)"};
  hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, out, cyc, 1u); hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, out, cyc, 1u);
  hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, out, cyc, 1u); hipLaunchKernelGGL(k<3>, dim3(1), dim3(256), 0, 0, out, cyc, 1u);
  hipLaunchKernelGGL(k<4>, dim3(1), dim3(256), 0, 0, out, cyc, 1u); hipLaunchKernelGGL(k<5>, dim3(1), dim3(256), 0, 0, out, cyc, 1u);
  hipLaunchKernelGGL(k<6>, dim3(1), dim3(256), 0, 0, out, cyc, 1u);
  hipDeviceSynchronize();
  for (int i = 0; i < 7; ++i) printf("%-32s %8.3f ticks per chain step\n", names[i], (double)cyc[i] / N / CH);
  return 0;
}
