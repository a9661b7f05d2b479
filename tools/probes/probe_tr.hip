// Probe of ds_read_b64_tr_b16 semantics on gfx950 (tools/probes: measurement helpers, not product code).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int pitch) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;   // value = element index = row*pitch + col
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  // group g covers rows 4g..4g+3; lane i points at (row 4g + (i>>2), col 4*(i&3))
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + (4*g + (i>>2))*pitch + 4*(i&3)));
  for (int e=0;e<4;++e) out[l*4+e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 64*4*2);
  for (int pitch : {16, 72}) {
    k<<<1,64>>>(d, pitch); short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pitch %d\n", pitch);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e=0;e<4;++e) printf(" (r%d,c%d)", h[l*4+e]/pitch, h[l*4+e]%pitch); printf("\n"); }
  }
  return 0;
}
