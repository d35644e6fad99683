#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int rowstride) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  // hypothesis: within each 16-lane group, lane i supplies the address of row (i/4)?? Just give each lane: row = lane%16 ... we print for two addressings
  // addressing A: lane l -> &lds[(l & 15) * rowstride + (l >> 4) * 4]   (16 rows (k) x 4 consecutive cols per group)
  const unsigned short* p = &lds[(lane & 15) * rowstride + (lane >> 4) * 4];
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)r[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int rs : {64, 16}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, rs);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("rowstride %d\n", rs);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
