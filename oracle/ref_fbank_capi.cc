// Test infrastructure: C entry point around the REFERENCE's own C++ fbank
// (runtime/core/frontend/fbank.h, compiled from /root/reference by
// oracle/Makefile into oracle/_ref/libref_fbank.so).  Used to pin
// oracle/wenet_oracle.py::fbank and to generate tests/golden/fbank_*.npz.
#include <vector>

// the mel filter bank is a private member; the pin below reads it (test code only)
#define private public
#include "frontend/fbank.h"
#undef private

extern "C" int ref_fbank(const float* wave_s16_scale, int n_samples, int num_bins,
                         int sample_rate, int frame_length, int frame_shift,
                         float* out, int max_frames) {
  // the Python path calls kaldi.fbank(waveform * (1 << 15), ...,
  // dither=0, energy_floor=0) (wenet/dataset/processor.py:245-254); the C++
  // class is its in-tree restatement with the same defaults
  // (frontend/feature_pipeline.h:55-63).
  wenet::Fbank fb(num_bins, sample_rate, frame_length, frame_shift);
  std::vector<float> wave(wave_s16_scale, wave_s16_scale + n_samples);
  std::vector<std::vector<float>> feat;
  const int n = fb.Compute(wave, &feat);
  if (n > max_frames) return -n;
  for (int t = 0; t < n; ++t)
    for (int b = 0; b < num_bins; ++b) out[t * num_bins + b] = feat[t][b];
  return n;
}

// The REFERENCE's own wav reader (runtime/core/frontend/wav.h:59-134): interleaved
// samples as raw integer values.  Pins wenet_amd.model.read_wav (PCM16 / PCM32, mono
// and multi-channel, files with extra chunks).  Returns the number of interleaved
// values, or a negative count if `out` is too small; -1: the reader refused the file.
#include "frontend/wav.h"

extern "C" int ref_wav_read(const char* path, float* out, int max_values, int* channels,
                            int* sample_rate, int* bits) {
  wenet::WavReader r;
  if (!r.Open(path)) return -1;
  *channels = r.num_channel();
  *sample_rate = r.sample_rate();
  *bits = r.bits_per_sample();
  const int n = r.num_samples() * r.num_channel();
  if (n > max_values) return -n;
  for (int i = 0; i < n; ++i) out[i] = r.data()[i];
  return n;
}

// The REFERENCE's C++ frontend in its Whisper configuration
// (frontend/feature_pipeline.h:64-72: 25 ms / 10 ms frames, low_freq 0, no
// pre-emphasis, log10 with floor 1e-10, periodic Hanning window, Slaney mel scale and
// filter normalisation, input scaled to unit, WhisperNorm) on a 512-point FFT.  The
// Python path (processor.py:320-369) uses a 400-point STFT with reflect padding, so
// this cannot pin the whole log-mel pipeline; it pins the pieces the oracle restates:
// the window, the Slaney filter construction (evaluated on this 512-point grid), the
// log10 / floor and the max - 8 / (x + 4) / 4 normalisation.
static wenet::Fbank make_whisper_fbank(int num_bins) {
  return wenet::Fbank(num_bins, 16000, 400, 160, /*low_freq=*/0.0f, /*pre_emphasis=*/false,
                      /*scale_input_to_unit=*/true, /*log_floor=*/1e-10f,
                      wenet::LogBase::kBase10, wenet::WindowType::kHanning,
                      wenet::MelType::kSlaney, wenet::NormalizationType::kWhisper);
}

extern "C" int ref_whisper_fbank(const float* wave_s16_scale, int n_samples, int num_bins,
                                 float* out, int max_frames) {
  wenet::Fbank fb = make_whisper_fbank(num_bins);
  std::vector<float> wave(wave_s16_scale, wave_s16_scale + n_samples);
  std::vector<std::vector<float>> feat;
  const int n = fb.Compute(wave, &feat);
  if (n > max_frames) return -n;
  for (int t = 0; t < n; ++t)
    for (int b = 0; b < num_bins; ++b) out[t * num_bins + b] = feat[t][b];
  return n;
}

// dense (num_bins, 256) Slaney filter weights of that configuration + its window
extern "C" void ref_slaney_filters(int num_bins, float* weights, float* window400) {
  wenet::Fbank fb = make_whisper_fbank(num_bins);
  for (int b = 0; b < num_bins; ++b) {
    for (int i = 0; i < 256; ++i) weights[b * 256 + i] = 0.f;
    const int first = fb.bins_[b].first;
    for (size_t k = 0; k < fb.bins_[b].second.size(); ++k)
      weights[b * 256 + first + k] = fb.bins_[b].second[k];
  }
  for (int i = 0; i < 400; ++i) window400[i] = fb.window_[i];
}
