// Which bf16 MFMA shape sustains the higher rate on a power-limited MI355X?    The vendor's best GEMM kernel for
// the step's shapes (Custom_Cijk_..._MT256x256x64_MI16x16x1: profiles/r03_hipblaslt_kernel_names.csv) is built from
// v_mfma_f32_16x16x32_bf16 with 128 x 128 wave tiles; every GEMM of this library uses v_mfma_f32_32x32x16_bf16.  Per MAC the 16x16x32
// form moves half the accumulator bytes and twice the operand bytes through the register file (0.5 vs 0.625 B/MAC in all) and its
// nominal issue rate is ~5 % lower (17 vs 32 cycles for half the MACs).  The chip clocks to its power budget, so what counts is the
// SUSTAINED rate with changing operand bits.  This probe runs the bare accumulate loops — no memory traffic in the loop, operands
// rotated among preloaded random fragment sets — one workgroup of 4 (or 8) waves per CU:
//     a) 32x32x16, one wave per SIMD, 128 x 128 wave tile (16 accumulator tiles of 16 registers)
//     b) 16x16x32, one wave per SIMD, 128 x 128 wave tile (64 accumulator tiles of 4 registers)
//     c) 32x32x16, two waves per SIMD, 128 x 64 wave tiles (the 8-wave kernel's register budget)
// and prints TF/s and the shader clock (cycles of wave 0 / wall time).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_rate.bin tools/probes/mfma_rate.hip && tools/probes/mfma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NSETMAX = 4;  // fragment sets rotated through (operand bits change from step to step): 4 with one wave per SIMD, 2 with two

template <int TI, int TJ, int NSET, int OCC>
__global__ __launch_bounds__(256, OCC) void k32(const bf16x8* frags, float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 fa[NSET][TI], fb[NSET][TJ];
#pragma unroll
  for (int s = 0; s < NSET; ++s) {
#pragma unroll
    for (int i = 0; i < TI; ++i) fa[s][i] = frags[((s * 16 + i) * 64 + lane)];
#pragma unroll
    for (int j = 0; j < TJ; ++j) fb[s][j] = frags[((s * 16 + 8 + j) * 64 + lane)];
  }
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < NSET; ++s)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s][j], fa[s][i], acc[i][j], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int TI, int TJ, int NSET, int OCC>
__global__ __launch_bounds__(256, OCC) void k16(const bf16x8* frags, float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 fa[NSET / 2][TI], fb[NSET / 2][TJ];   // a set covers 32 k: two sets = the k range of four 32x32x16 sets
#pragma unroll
  for (int s = 0; s < NSET / 2; ++s) {
#pragma unroll
    for (int i = 0; i < TI; ++i) fa[s][i] = frags[((s * 16 + i) * 64 + lane)];
#pragma unroll
    for (int j = 0; j < TJ; ++j) fb[s][j] = frags[((s * 16 + 8 + j) * 64 + lane)];
  }
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < NSET / 2; ++s)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[s][j], fa[s][i], acc[i][j], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  int ncu = 256;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  std::vector<unsigned short> h((size_t)NSETMAX * 16 * 64 * 8);
  unsigned x = 12345u;
  for (auto& v : h) {  // bf16 bit patterns of uniform [-1, 1) values
    x = x * 1664525u + 1013904223u;
    const float f = ((x >> 8) * (1.0f / 8388608.0f)) - 1.0f;
    unsigned u; memcpy(&u, &f, 4);
    v = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
  bf16x8* frags; float* out; long long* cyc;
  hipMalloc(&frags, h.size() * 2); hipMalloc(&out, (size_t)ncu * 2 * 512 * 4); hipMalloc(&cyc, ncu * 2 * 8);
  hipMemcpy(frags, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, double macs_per_wg) {
    for (int rep = 0; rep < 3; ++rep) {
      launch();  // warm-up (and clock ramp)
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      long long c0 = 0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
      printf("%-58s %8.3f ms  %7.0f TF/s  shader clock %.2f GHz (%lld cycles)\n", name, ms, 2.0 * macs_per_wg * ncu / (ms * 1e-3) / 1e12,
             (double)c0 / (ms * 1e-3) / 1e9, c0);
    }
  };
  // a) and b): 4 waves x 128 x 128 wave tile, k = 64 per loop iteration
  const double macs4 = 4.0 * 128 * 128 * 64 * iters;   // NSET = 4 sets of k16 (two of k32) per iteration
  run("a) 32x32x16, 1 wave/SIMD, 128x128 wave tile", [&] { hipLaunchKernelGGL((k32<4, 4, 4, 1>), dim3(ncu), dim3(256), 0, 0, frags, out, cyc, iters); }, macs4);
  run("b) 16x16x32, 1 wave/SIMD, 128x128 wave tile", [&] { hipLaunchKernelGGL((k16<8, 8, 4, 1>), dim3(ncu), dim3(256), 0, 0, frags, out, cyc, iters); }, macs4);
  // c): 8 waves (two workgroups of 4 waves per CU would need co-residency: launch 2 x ncu workgroups, 2 per CU fit by registers)
  const double macs8 = 4.0 * 128 * 64 * 32 * iters;    // NSET = 2: k = 32 per iteration
  run("c) 32x32x16, 2 waves/SIMD (2 WG/CU), 128x64 wave tiles", [&] { hipLaunchKernelGGL((k32<4, 2, 2, 2>), dim3(2 * ncu), dim3(256), 0, 0, frags, out, cyc, iters); }, macs8 * 2);
  run("d) 16x16x32, 2 waves/SIMD (2 WG/CU), 128x64 wave tiles", [&] { hipLaunchKernelGGL((k16<8, 4, 2, 2>), dim3(2 * ncu), dim3(256), 0, 0, frags, out, cyc, iters); }, macs8 * 2);
  return 0;
}
