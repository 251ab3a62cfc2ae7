// Test infrastructure: C entry point around the REFERENCE's own C++ fbank
// (runtime/core/frontend/fbank.h, compiled from /root/reference by
// oracle/Makefile into oracle/_ref/libref_fbank.so).  Used to pin
// oracle/wenet_oracle.py::fbank and to generate tests/golden/fbank_*.npz.
#include <vector>

#include "frontend/fbank.h"

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
