// Why do some boxes run every one-wave-per-SIMD kernel ~25 us slower per launch (ffn_x6f 86 ->
// 110 us, x6r 32 -> 74 us, visits r06b / r06q) while the 8-wave tile kernels are unaffected?
// Launch-cost probe: the same trivial work under different resource shapes, back to back.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/launch_probe.hip -o /tmp/lp && /tmp/lp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NREG, int MINB>
__global__ __launch_bounds__(256, MINB) void probe_kernel(const float* in, float* out, int iters) {
  extern __shared__ char lds[];
  float r[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) r[i] = in[(threadIdx.x + i * 256) & 4095];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = __builtin_fmaf(r[i], 1.0001f, 0.5f);
#pragma unroll
    for (int i = 0; i < NREG; ++i) asm volatile("" : "+v"(r[i]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NREG; ++i) s += r[i];
  if (lds != nullptr && threadIdx.x == 0) reinterpret_cast<volatile float*>(lds)[0] = s;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NREG, int MINB>
int run(const char* what, size_t lds_bytes, int blocks, const float* in, float* out) {
  auto k = probe_kernel<NREG, MINB>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds_bytes, 0, in, out, 8);
  CK(hipDeviceSynchronize());
  const int n = 400;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds_bytes, 0, in, out, 8);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k)));
  printf("%-44s regs %3d  LDS %6zu B  blocks %4d : %7.2f us per launch\n", what, fa.numRegs, lds_bytes, blocks, ms * 1e3 / n);
  return 0;
}

int main() {
  float *in, *out;
  CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 1024 * 256 * 4));
  CK(hipMemset(in, 0, 4096 * 4));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("%s  CUs %d  clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
  for (int rep = 0; rep < 2; ++rep) {
    run<32, 4>("32 values / lane, 4 blocks per CU", 0, 248, in, out);
    run<200, 2>("200 values / lane, launch_bounds(256, 2)", 0, 248, in, out);
    run<32, 1>("32 values / lane, launch_bounds(256, 1)", 0, 248, in, out);
    run<300, 1>("300 values / lane, launch_bounds(256, 1)", 0, 248, in, out);
    run<440, 1>("440 values / lane, launch_bounds(256, 1)", 0, 248, in, out);
    run<32, 4>("32 values / lane + 100 KB LDS", 100 * 1024, 248, in, out);
    run<32, 4>("32 values / lane + 150 KB LDS", 150 * 1024, 248, in, out);
    run<300, 1>("300 values / lane + 150 KB LDS", 150 * 1024, 248, in, out);
    run<300, 1>("300 values / lane + 150 KB LDS, 496 blocks", 150 * 1024, 496, in, out);
  }
  return 0;
}
