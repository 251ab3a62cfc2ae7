// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 operand / scale association (gfx950).
// hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_scale_probe.hip -o gpurun_out/probe && gpurun_out/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// cfg: a_mask[hi][half]: which 16-byte halves of lane-group hi hold 1.0 (else 0)
__global__ void probe(const int* cfg, float* out) {
  const int l = threadIdx.x, hi = l >> 5;
  const int one4 = 0x38383838;  // e4m3 1.0 x 4
  i32x8 a, b;
  for (int r = 0; r < 8; ++r) {
    const int half = r >> 2;
    a[r] = cfg[hi * 2 + half] ? one4 : 0;
    b[r] = cfg[4 + hi * 2 + half] ? one4 : 0;
  }
  const int sa = cfg[8 + hi];      // dword of 4 scale bytes per lane group
  const int sb = cfg[10 + hi];
  f32x16 acc = {0};
  if (cfg[12] == 0)
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
  else if (cfg[12] == 1)
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 1, sa, 1, sb);
  else if (cfg[12] == 2)
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 2, sa, 2, sb);
  else
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 3, sa, 3, sb);
  if (l == 0) out[0] = acc[0];
}

int main() {
  int* d; float* o;
  hipMalloc(&d, 64 * 4); hipMalloc(&o, 4);
  struct T { const char* name; int c[13]; } tests[] = {
    {"all ones, scales 1 -> 64", {1,1,1,1, 1,1,1,1, 0x7f7f7f7f,0x7f7f7f7f, 0x7f7f7f7f,0x7f7f7f7f, 0}},
    {"A lanes<32 only -> 32", {1,1,0,0, 1,1,1,1, 0x7f7f7f7f,0x7f7f7f7f, 0x7f7f7f7f,0x7f7f7f7f, 0}},
    {"A lanes<32 first half only -> 16", {1,0,0,0, 1,1,1,1, 0x7f7f7f7f,0x7f7f7f7f, 0x7f7f7f7f,0x7f7f7f7f, 0}},
    {"all ones, sa lanes<32 = 2 (byte0), lanes>=32 = 1 -> 96 if per-lane", {1,1,1,1, 1,1,1,1, 0x7f7f7f80,0x7f7f7f7f, 0x7f7f7f7f,0x7f7f7f7f, 0}},
    {"A = lanes>=32 first half; sa: lanes<32 x2, lanes>=32 x4 -> 64 if scale is per lane", {0,0,1,0, 1,1,1,1, 0x7f7f7f80,0x7f7f7f81, 0x7f7f7f7f,0x7f7f7f7f, 0}},
    {"A = lanes>=32 second half; same scales -> 64 if per lane", {0,0,0,1, 1,1,1,1, 0x7f7f7f80,0x7f7f7f81, 0x7f7f7f7f,0x7f7f7f7f, 0}},
    {"A = lanes<32 second half; same scales -> 32 if per lane", {0,1,0,0, 1,1,1,1, 0x7f7f7f80,0x7f7f7f81, 0x7f7f7f7f,0x7f7f7f7f, 0}},
    {"op_sel 1: all ones, sa byte1 = x2 (both groups), byte0 = x8 -> 128", {1,1,1,1, 1,1,1,1, 0x7f7f8082,0x7f7f8082, 0x7f7f7f7f,0x7f7f7f7f, 1}},
    {"op_sel 2: all ones, sa byte2 = x2, others x8 -> 128", {1,1,1,1, 1,1,1,1, (int)0x82808282,(int)0x82808282, 0x7f7f7f7f,0x7f7f7f7f, 2}},
    {"op_sel 3: all ones, sa byte3 = x2, others x8 -> 128", {1,1,1,1, 1,1,1,1, (int)0x80828282,(int)0x80828282, 0x7f7f7f7f,0x7f7f7f7f, 3}},
    {"op_sel 2 on B: sb byte2 = x4 -> 256", {1,1,1,1, 1,1,1,1, 0x7f7f7f7f,0x7f7f7f7f, (int)0x7f817f7f,(int)0x7f817f7f, 2}},
  };
  for (auto& t : tests) {
    hipMemcpy(d, t.c, sizeof(t.c), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
    float r; hipMemcpy(&r, o, 4, hipMemcpyDeviceToHost);
    printf("%-90s = %g\n", t.name, r);
  }
  return 0;
}
