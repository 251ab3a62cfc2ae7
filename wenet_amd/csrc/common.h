// Shared declarations for libwenet_amd (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <string>

#include "tune.h"

namespace wn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error plumbing -------------------------------------------------------
void set_error(const std::string& msg);
#define WN_STR2(x) #x
#define WN_STR(x) WN_STR2(x)
#define WN_HIP(expr)                                                         \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      ::wn::set_error(std::string(__FILE__ ":" WN_STR(__LINE__) ": ") +      \
                      #expr + " -> " + hipGetErrorString(_e));               \
      return -2;                                                             \
    }                                                                        \
  } while (0)
#define WN_CHECK(cond, msg)                                                  \
  do {                                                                       \
    if (!(cond)) {                                                           \
      ::wn::set_error(std::string(__FILE__ ":" WN_STR(__LINE__) ": ") +      \
                      (msg));                                                \
      return -1;                                                             \
    }                                                                        \
  } while (0)
#define WN_TRY(expr)                                                         \
  do {                                                                       \
    int _r = (expr);                                                         \
    if (_r != 0) return _r;                                                  \
  } while (0)

// hipFuncSetAttribute is per DEVICE: a process may hold handles on several GPUs
// (wn_model_create takes a device), so the "already done" memo of a launcher is a bit per
// device of the calling thread, not one flag (round-4 advice).  Setting the attribute twice is
// harmless, so two threads racing here both set it before either launches.
#define WN_MAX_DYN_LDS(kern, bytes)                                                          \
  do {                                                                                       \
    static std::atomic<uint64_t> _mask{0};                                                   \
    int _dev = 0;                                                                            \
    WN_HIP(hipGetDevice(&_dev));                                                             \
    const uint64_t _bit = 1ull << (_dev & 63);                                               \
    if (!(_mask.load(std::memory_order_acquire) & _bit)) {                                   \
      WN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                        \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
      _mask.fetch_or(_bit, std::memory_order_release);                                       \
    }                                                                                        \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device helpers -------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// SiLU of the convolution module (convolution.py:146: x * sigmoid(x)) on v_exp_f32 / v_rcp_f32 --
// ~1 ulp each, the form the fused feed-forward kernels always used (gemm_epilogue.h silu_fast) --
// instead of the exact expf and the IEEE division sequence: 7.5 k of the depthwise-conv prologue's
// 19 k cycles went into those (round 4 stamps); round 5.
// WN_EXACT_TRANSCENDENTALS (a validation build: WN_EXACT=1 python -m wenet_amd.build): the
// exponentials and reciprocals of the SiLU / GLU gates and of the CTC log-softmax sum run on the
// exact library routines (expf, exp2f, IEEE division) -- the reference's own operations --
// instead of v_exp_f32 / v_rcp_f32.  The product library is never built this way; the build
// exists so that the 1-2 ulp the fast forms move can be measured end to end (DESIGN.md,
// deviations).  The attention softmaxes are not part of it: their exp always ran on v_exp_f32.
#ifdef WN_EXACT_TRANSCENDENTALS
__device__ __forceinline__ float wn_exp(float x) { return expf(x); }
__device__ __forceinline__ float wn_exp2(float x) { return exp2f(x); }
__device__ __forceinline__ float wn_rcp(float x) { return 1.0f / x; }
#else
__device__ __forceinline__ float wn_exp(float x) { return __expf(x); }
__device__ __forceinline__ float wn_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float wn_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif
__device__ __forceinline__ float silu_f(float x) {
  return x * wn_rcp(1.0f + wn_exp(-x));
}
__device__ __forceinline__ float sigmoid_f(float x) {
  return 1.0f / (1.0f + expf(-x));
}
#endif

// ---- GEMM -----------------------------------------------------------------
// C[M,N] = epilogue(A[M,K] * W[N,K]^T), fp32 in / fp32 accumulate on
// v_mfma_f32_32x32x2_f32.  All pointers are device pointers.
enum GemmAct { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_GELU = 3 };  // GELU: exact (erf)

struct GemmArgs {
  const float* A = nullptr;  // [M, lda] (plain) or gathered (a_row_off)
  const float* W = nullptr;  // [N, K] row-major (torch Linear layout)
  const float* bias = nullptr;   // [N] or null
  const float* resid = nullptr;  // [M, ldr] or null: C = resid + alpha*act(..)
  float* C = nullptr;            // [M, ldc]
  int M = 0, N = 0, K = 0;
  int lda = 0, ldc = 0, ldr = 0;
  float alpha = 1.0f;
  int act = ACT_NONE;
  bool glu = false;  // W rows permuted [32 a | 32 gate] per 64; C has N/2 cols
  // implicit 3x3/stride-2 conv A operand (subsampling conv2): element (row,k)
  // lives at A[a_row_off[row] + (tap/3)*conv_sy + (tap%3)*conv_sx + k%conv_C],
  // tap = k / conv_C.  conv_C == K degenerates to plain gathered rows
  // A[a_row_off[row] + k] (1-D conv over channels-last frames: the taps of one
  // output are consecutive input rows).
  const int64_t* a_row_off = nullptr;
  int conv_C = 0;
  int64_t conv_sy = 0, conv_sx = 0;
  // bf16-storage form of the bf16 mode (gemm_bf16s.hip): A is a bf16 matrix (A
  // reinterpreted, lda in bf16 elements); C is written as bf16 (ldc in elements).
  // Needs the bf16 image of the weight slab (t_wslab_*), plain A, no residual
  // with bf16 C.
  bool a_bf16 = false, c_bf16 = false;
  // MXFP8 form (gemm_bf16p.hip, ET = 1): A is an e4m3 matrix (lda in elements) with
  // block scales a_scale, W comes with w_scale; scales are dwords [K/128][pitch]
  // (4 E8M0 bytes of the 4 k blocks of a K tile, K-tile-major).  c_mx: C is written as
  // MXFP8 (ldc in elements) with its block scales in c_scale (the k blocks of the
  // next GEMM).
  bool fp8 = false, c_mx = false;
  const unsigned* a_scale = nullptr; const unsigned* w_scale = nullptr;
  unsigned* c_scale = nullptr;
  int a_scale_pitch = 0, w_scale_pitch = 0, c_scale_pitch = 0;
  // measurement (gemm_bf16p.hip, wn_tune_set("lp_probe")): bit 2 = shader-clock stamps of one
  // block (bit 3: block 0 instead of the middle one, bit 4: the last one)
  int probe = 0;
};

int gemm_f32(const GemmArgs& a, hipStream_t stream);

// Operand precision of the calling host thread's GEMMs (wn_model_set_precision):
// 0 = fp32 operands (gemm.hip, the default and the parity mode), 1 = operands
// rounded to bf16 on the way into LDS, fp32 accumulate (gemm_bf16.hip).  Set by
// the C-ABI entry points from their handle (PrecisionScope in model.hip); one
// host thread drives one handle, so a thread-local is the handle's state.
enum GemmPrecision { PREC_F32 = 0, PREC_BF16 = 1 };
extern thread_local int t_gemm_prec;
int gemm_bf16(const GemmArgs& a, hipStream_t stream);  // called by gemm_f32
// bf16 image of the calling handle's weight slab (PrecisionScope): W pointers
// inside [t_wslab_f32, t_wslab_f32 + t_wslab_elems) map to t_wslab_bf16 + offset.
extern thread_local const float* t_wslab_f32;
extern thread_local const void* t_wslab_bf16;
extern thread_local int64_t t_wslab_elems;
int gemm_bf16_stored(const GemmArgs& a, const void* Wh, hipStream_t stream);
// 256x256 direct-to-LDS pipelined kernel for the large shapes (gemm_bf16p.hip)
bool gemm_bf16p_supported(const GemmArgs& a);
int gemm_bf16_pipelined(const GemmArgs& a, const void* Wh, hipStream_t stream);
int gemm_mxfp8(const GemmArgs& a, const void* Wq, hipStream_t stream);
int gemm_lp_clocks(unsigned long long* out);   // [8 waves][8] stamps of the pipelined kernel
// fp32 [rows][ld] -> e4m3 [rows][K] + block scales [K/128][pitch] dwords
int mx_quantize(const float* x, int ld, int rows, int K, void* q, unsigned* scale, int pitch,
                hipStream_t s);
enum { PREC_FP8 = 2 };  // bf16 mode with MXFP8 FFN GEMMs (wn_model_set_precision)
int convert_f32_to_bf16(const float* x, void* y, int64_t n, hipStream_t s);

}  // namespace wn
