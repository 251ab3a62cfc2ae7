// Probe of ds_read_b64_tr_b16 (gfx950): which (row, column) of the LDS image lands in which lane /
// element.  Image: [32 keys][64 dims] bf16, row stride 128 B (no swizzle).  A 16-lane group g reads
// a [4 keys][16 dims] block: lane j of the group passes the address of key kb + (j >> 2), dims
// db + 4 (j & 3) .. + 3 (8 contiguous bytes).  Expected (MI355X guide, T10): the lane receives
// COLUMN db + j of the block, elements = keys kb .. kb + 3.
// hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_probe.hip -o gpurun_out/tr16_probe && gpurun_out/tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void probe(int which, int kb, float* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[32 * 64];
  for (int i = threadIdx.x; i < 32 * 64; i += 64) {
    const int key = i >> 6, dim = i & 63;
    const float v = which == 0 ? (float)key : (float)dim;
    lds[i] = (unsigned short)(__float_as_uint(v) >> 16);   // exact: small integers
  }
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
  const int db = 16 * (g & 1) + 32 * (g >> 1);   // four different column blocks, one per group
  const int off = (kb + (j >> 2)) * 128 + (db + 4 * (j & 3)) * 2;
  typedef __attribute__((address_space(3))) s16x4* lp;
  const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)((char*)lds + off));
  for (int e = 0; e < 4; ++e)
    out[lane * 4 + e] = __uint_as_float(((unsigned)(unsigned short)r[e]) << 16);
}

int main() {
  float *d, hk[256], hd[256];
  hipMalloc(&d, sizeof(hk));
  int bad = 0;
  for (int kb = 0; kb <= 8; kb += 8) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, 0, kb, d);
    hipMemcpy(hk, d, sizeof(hk), hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, 1, kb, d);
    hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, j = l & 15, db = 16 * (g & 1) + 32 * (g >> 1);
      printf("kb %d lane %2d:", kb, l);
      for (int e = 0; e < 4; ++e) {
        printf(" (k%2d,d%2d)", (int)hk[l * 4 + e], (int)hd[l * 4 + e]);
        if ((int)hk[l * 4 + e] != kb + e || (int)hd[l * 4 + e] != db + j) ++bad;
      }
      printf("\n");
    }
  }
  printf("mismatches against the expected mapping (lane j of a group = column db + j, element e = key kb + e): %d\n", bad);
  return 0;
}
