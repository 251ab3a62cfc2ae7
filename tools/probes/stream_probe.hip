// How fast can ONE CU stream an L2-resident image into registers (the weight stream of the
// row-block GEMMs, gemm_x6r.hip: ~33 B/clk per CU measured, tools/bench_x6r_blocks.py), and does
// the kind of load or the number of loads in flight change it?  Every block (4 waves, one per
// SIMD) reads the SAME image (1-KB records, lane x 16 B: perfectly coalesced) `passes` times.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND 0: global_load_dwordx4; 1: the same nontemporal; 2: buffer_load_dwordx4 ... lds (LDS-DMA,
// gfx950: 16 bytes per lane straight into LDS, no VGPRs)
template <int U, int KIND>
__global__ __launch_bounds__(256, 1) void stream_kernel(const char* img, int records, int passes,
                                                        float* out) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(img), 0, records * 1024, 0x00020000);
  for (int p = 0; p < passes; ++p) {
    for (int r0 = wave * U; r0 + U <= records; r0 += 4 * U) {
      if constexpr (KIND == 2) {
#pragma unroll
        for (int i = 0; i < U; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + (wave * U + i) * 1024), 16,
                                                   lane * 16, (r0 + i) * 1024, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        f32x4 v[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
          const f32x4* q = reinterpret_cast<const f32x4*>(img + (size_t)(r0 + i) * 1024) + lane;
          v[i] = KIND == 1 ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int i = 0; i < U; ++i) acc += v[i];
      }
    }
  }
  if (KIND == 2) acc[0] = reinterpret_cast<float*>(lds)[threadIdx.x];
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int U, int KIND>
int run(const char* what, const char* img, int records, int blocks, float* out, double ghz) {
  auto k = stream_kernel<U, KIND>;
  const size_t lds = KIND == 2 ? (size_t)4 * U * 1024 : 0;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds ? lds : 1024)));
  const int passes = 8;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, img, records, passes, out);
  CK(hipDeviceSynchronize());
  const int n = 20;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, img, records, passes, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / n - 3.0;    // (~3 us of launch per kernel)
  const double bytes = (double)records * 1024 * passes;
  printf("%-34s %3d KB in flight per CU, %3d blocks: %7.1f us per launch, %6.1f GB/s per CU = %5.1f B/clk at %.1f GHz, %6.2f TB/s all\n",
         what, 4 * U, blocks, us + 3.0, bytes / us / 1e3, bytes / us / 1e3 / ghz, ghz, bytes * blocks / us / 1e6);
  return 0;
}

int main() {
  const int records = 1152;                 // 1.18 MB: the QKV weight image at d = 256
  char* img; float* out;
  CK(hipMalloc(&img, (size_t)records * 1024)); CK(hipMalloc(&out, 256 * 256 * 4));
  CK(hipMemset(img, 0, (size_t)records * 1024));
  const double ghz = 2.4;
  for (int blocks : {1, 31, 248}) {
    run<4, 0>("global_load_dwordx4", img, records, blocks, out, ghz);
    run<8, 0>("global_load_dwordx4", img, records, blocks, out, ghz);
    run<16, 0>("global_load_dwordx4", img, records, blocks, out, ghz);
    run<32, 0>("global_load_dwordx4", img, records, blocks, out, ghz);
    run<16, 1>("global_load_dwordx4 nt", img, records, blocks, out, ghz);
    run<8, 2>("buffer_load_dwordx4 lds", img, records, blocks, out, ghz);
    run<16, 2>("buffer_load_dwordx4 lds", img, records, blocks, out, ghz);
    run<32, 2>("buffer_load_dwordx4 lds", img, records, blocks, out, ghz);
  }
  return 0;
}
