// Kaldi-compatible log-mel filterbank on the GPU.
//
// Replaces processor.compute_fbank (wenet/dataset/processor.py:226-256), i.e.
// torchaudio.compliance.kaldi.fbank(waveform * 32768, num_mel_bins, 25 ms,
// 10 ms, dither 0, energy_floor 0, povey window) with Kaldi defaults, following
// the reference's own C++ statement of that arithmetic
// (runtime/core/frontend/fbank.h:250-327): snip_edges framing, DC removal,
// pre-emphasis 0.97, povey window, zero pad to 512, FFT, power spectrum of bins
// [0, 256), HTK triangular mel filters (20 Hz .. Nyquist), log(max(e, eps)).
//
// One 256-thread block per (utterance, frame): the 400 samples are read with
// coalesced loads (4 B/sample, the only HBM traffic besides the 80 outputs),
// the 512-point radix-2 FFT runs in LDS (one butterfly per thread per stage),
// the mel filters are applied from a CSR table.
#include "kernels.h"

namespace wn {

namespace {

constexpr int NFFT = 512;
constexpr int FRAME_LEN = 400;
constexpr int FRAME_SHIFT = 160;

__global__ __launch_bounds__(256) void fbank_kernel(FbankArgs a) {
  const int b = blockIdx.y, fr = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* out = a.feats + ((int64_t)b * a.max_frames + fr) * a.n_mel;
  if (fr >= a.n_frames[b]) {  // zero padding (processor.py:559-561)
    for (int i = tid; i < a.n_mel; i += 256) out[i] = 0.f;
    return;
  }
  __shared__ float re[NFFT], im[NFFT];
  __shared__ float red[4];
  const float* src = a.pcm + a.sample_off[b] + (int64_t)fr * FRAME_SHIFT;
  // scale to int16 range (processor.py:245) and remove the DC offset
  float v0 = tid < FRAME_LEN ? src[tid] * 32768.0f : 0.f;
  float v1 = tid + 256 < FRAME_LEN ? src[tid + 256] * 32768.0f : 0.f;
  float sum = wave_sum(v0 + v1);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)FRAME_LEN;
  re[tid] = v0 - mean;
  re[tid + 256] = tid + 256 < FRAME_LEN ? v1 - mean : 0.f;
  __syncthreads();
  // pre-emphasis (fbank.h:221-226) + window, written bit-reversed for the FFT
  float w0 = 0.f, w1 = 0.f;
  if (tid < FRAME_LEN) {
    const float prev = tid > 0 ? re[tid - 1] : re[0];
    w0 = (re[tid] - 0.97f * prev) * a.window[tid];
  }
  if (tid + 256 < FRAME_LEN) {
    const int i = tid + 256;
    w1 = (re[i] - 0.97f * re[i - 1]) * a.window[i];
  }
  __syncthreads();
  re[__brev((unsigned)tid) >> 23] = w0;
  re[__brev((unsigned)(tid + 256)) >> 23] = w1;
  im[tid] = 0.f;
  im[tid + 256] = 0.f;
  __syncthreads();
  // radix-2 decimation-in-time, 9 stages, twiddle = exp(-2 pi i k / 512)
#pragma unroll
  for (int st = 0; st < 9; ++st) {
    const int half = 1 << st;
    const int grp = tid >> st, k = tid & (half - 1);
    const int i0 = (grp << (st + 1)) + k, i1 = i0 + half;
    const int tw = k << (8 - st);
    const float c = a.twiddle[2 * tw], s = a.twiddle[2 * tw + 1];
    const float xr = re[i1], xi = im[i1];
    const float tr = xr * c - xi * s, ti = xr * s + xi * c;
    const float ur = re[i0], ui = im[i0];
    re[i0] = ur + tr; im[i0] = ui + ti;
    re[i1] = ur - tr; im[i1] = ui - ti;
    __syncthreads();
  }
  const float pw = re[tid] * re[tid] + im[tid] * im[tid];  // bins [0, 256)
  __syncthreads();
  re[tid] = pw;
  __syncthreads();
  for (int mbin = tid; mbin < a.n_mel; mbin += 256) {
    const int st = a.mel_start[mbin], n = a.mel_len[mbin];
    const float* wt = a.mel_w + a.mel_off[mbin];
    float e = 0.f;
    for (int i = 0; i < n; ++i) e += wt[i] * re[st + i];
    e = fmaxf(e, 1.1920928955078125e-07f);  // FLT_EPSILON (fbank.h:306)
    out[mbin] = logf(e);
  }
}

// Polyphase sinc resampler (torchaudio.functional.resample as called by
// processor.resample, processor.py:177-196): output sample t = j * new + i is the
// dot product of phase i's K taps with the input window starting at
// j * orig - width (zero outside the signal).  One thread per output sample;
// the taps of a phase are read by consecutive lanes of other phases -> the table
// stays in L1/L2, the signal is read ~K/orig times from cache.  HBM-bound.
__global__ __launch_bounds__(256) void resample_kernel(
    const float* __restrict__ x, int64_t n_in, const float* __restrict__ taps, int K,
    int width, int orig, int nnew, float* __restrict__ out, int64_t n_out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out) return;
  const int64_t j = t / nnew;
  const int i = (int)(t - j * nnew);
  const float* w = taps + (int64_t)i * K;
  const int64_t base = j * orig - width;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    const int64_t src = base + k;
    const float v = (src >= 0 && src < n_in) ? x[src] : 0.f;
    acc = fmaf(w[k], v, acc);
  }
  out[t] = acc;
}

}  // namespace

int resample_sinc(const float* x, int64_t n_in, const float* taps, int K, int width,
                  int orig, int nnew, float* out, int64_t n_out, hipStream_t s) {
  WN_CHECK(n_out > 0 && K > 0, "resample: empty");
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)cdiv(n_out, (int64_t)256)),
                     dim3(256), 0, s, x, n_in, taps, K, width, orig, nnew, out, n_out);
  WN_HIP(hipGetLastError());
  return 0;
}

int fbank_kaldi(const FbankArgs& a, hipStream_t s) {
  WN_CHECK(a.B > 0 && a.max_frames > 0, "fbank: empty");
  hipLaunchKernelGGL(fbank_kernel, dim3(a.max_frames, a.B), dim3(256), 0, s, a);
  WN_HIP(hipGetLastError());
  return 0;
}

}  // namespace wn
