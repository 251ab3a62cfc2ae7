// Whisper log-mel spectrogram on the GPU.
//
// Replaces processor.compute_log_mel_spectrogram
// (wenet/dataset/processor.py:320-369): torch.stft(n_fft=400, hop=160, periodic
// hann, center=True / reflect padding), |.|^2 of all frames but the last,
// librosa slaney mel filters, log10(clamp(., 1e-10)), floor at (utterance max
// - 8), (x + 4) / 4.
//
// n_fft = 400 is not a power of two, and the transform is tiny, so the DFT is a
// GEMM on the fp32 matrix cores: frames[T, 416] (windowed, zero-padded K) times
// the [402, 416] matrix of cos / -sin rows, then |.|^2 -> [T, 224] times the
// [n_mels, 224] mel matrix.  The kernels here do the framing (reflect padding,
// window), the power spectrum, and the two log passes (per-utterance max, then
// floor + affine), all over the packed frames of the whole batch.
#include "kernels.h"

namespace wn {

namespace {

constexpr int N_FFT = 400;
constexpr int HOP = 160;

__global__ __launch_bounds__(128) void logmel_frame_kernel(LogMelArgs a) {
  const int row = blockIdx.x;  // packed frame
  const int b = a.row_utt[row];
  const int t = row - a.frame_off[b];
  const int64_t n = a.sample_off[b + 1] - a.sample_off[b];
  const float* x = a.pcm + a.sample_off[b];
  float* dst = a.frames + (int64_t)row * LOGMEL_K1;
  for (int i = threadIdx.x; i < LOGMEL_K1; i += 128) {
    float v = 0.f;
    if (i < N_FFT) {
      int64_t j = (int64_t)t * HOP + i - N_FFT / 2;  // center=True
      if (j < 0) j = -j;                              // reflect (no edge repeat)
      if (j >= n) j = 2 * (n - 1) - j;
      v = x[j] * a.window[i];
    }
    dst[i] = v;
  }
}

__global__ __launch_bounds__(256) void logmel_power_kernel(const float* spec,
                                                           float* pw, int rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * LOGMEL_K2) return;
  const int r = i / LOGMEL_K2, k = i - r * LOGMEL_K2;
  float v = 0.f;
  if (k <= N_FFT / 2) {
    const float re = spec[(int64_t)r * LOGMEL_NS + k];
    const float im = spec[(int64_t)r * LOGMEL_NS + (N_FFT / 2 + 1) + k];
    v = re * re + im * im;
  }
  pw[i] = v;
}

// pass 1: log10(clamp) in place + per-utterance maximum (one block per utterance)
__global__ __launch_bounds__(256) void logmel_log_kernel(float* mel, int n_mels,
                                                         const int* frame_off,
                                                         const int* n_frames,
                                                         float* umax) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float* p = mel + (int64_t)frame_off[b] * n_mels;
  const int64_t n = (int64_t)n_frames[b] * n_mels;
  float mx = -INFINITY;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const float v = log10f(fmaxf(p[i], 1e-10f));
    p[i] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) umax[b] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// pass 2: floor at max - 8, (x + 4) / 4, scatter into the padded (B, Tmax, n_mels)
__global__ __launch_bounds__(128) void logmel_norm_kernel(const float* mel, int n_mels,
                                                          const int* frame_off,
                                                          const int* n_frames,
                                                          const float* umax,
                                                          int max_frames, float* feats) {
  const int b = blockIdx.y, t = blockIdx.x;
  float* dst = feats + ((int64_t)b * max_frames + t) * n_mels;
  if (t < n_frames[b]) {
    const float* src = mel + (int64_t)(frame_off[b] + t) * n_mels;
    const float lo = umax[b] - 8.0f;
    for (int i = threadIdx.x; i < n_mels; i += 128)
      dst[i] = (fmaxf(src[i], lo) + 4.0f) / 4.0f;
  } else {
    for (int i = threadIdx.x; i < n_mels; i += 128) dst[i] = 0.f;
  }
}

}  // namespace

int logmel_frames(const LogMelArgs& a, int rows, hipStream_t s) {
  hipLaunchKernelGGL(logmel_frame_kernel, dim3(rows), dim3(128), 0, s, a);
  WN_HIP(hipGetLastError());
  return 0;
}

int logmel_power(const float* spec, float* pw, int rows, hipStream_t s) {
  hipLaunchKernelGGL(logmel_power_kernel, dim3(cdiv(rows * LOGMEL_K2, 256)), dim3(256),
                     0, s, spec, pw, rows);
  WN_HIP(hipGetLastError());
  return 0;
}

int logmel_finish(float* mel, int n_mels, const int* frame_off, const int* n_frames,
                  float* umax, int B, int max_frames, float* feats, hipStream_t s) {
  hipLaunchKernelGGL(logmel_log_kernel, dim3(B), dim3(256), 0, s, mel, n_mels,
                     frame_off, n_frames, umax);
  hipLaunchKernelGGL(logmel_norm_kernel, dim3(max_frames, B), dim3(128), 0, s, mel,
                     n_mels, frame_off, n_frames, umax, max_frames, feats);
  WN_HIP(hipGetLastError());
  return 0;
}

}  // namespace wn
