// host_data.hip — host-side (CPU) native pieces of the input pipeline; no device code.
//   cst_batch_by_size : the reference's Cython batch_by_size_fast (fairseq/data/data_utils_fast.pyx:17-67)
//   cst_wav_info/read : PCM WAV slice decoding — what the reference gets from libsndfile through
//                       soundfile.read(path, dtype="float32", start=offset, frames=length)
//                       (fairseq/data/audio/audio_utils.py:7-55): 16-bit PCM samples / 32768.
#include "cst_common.h"
#include <stdio.h>
#include <string.h>

namespace {

inline bool batch_full(int64_t num_sentences, int64_t num_tokens, int64_t max_tokens, int64_t max_sentences) {
  if (num_sentences == 0) return false;
  if (max_sentences > 0 && num_sentences == max_sentences) return true;
  if (max_tokens > 0 && num_tokens > max_tokens) return true;
  return false;
}

struct WavHeader {
  int channels, bits, rate, format;
  int64_t data_offset, data_bytes;
};

int parse_wav(FILE* f, WavHeader& h) {
  unsigned char b[12];
  if (fread(b, 1, 12, f) != 12 || memcmp(b, "RIFF", 4) != 0 || memcmp(b + 8, "WAVE", 4) != 0) return -1;
  bool have_fmt = false;
  for (;;) {
    unsigned char ch[8];
    if (fread(ch, 1, 8, f) != 8) return -1;
    const uint32_t sz = (uint32_t)ch[4] | ((uint32_t)ch[5] << 8) | ((uint32_t)ch[6] << 16) | ((uint32_t)ch[7] << 24);
    if (memcmp(ch, "fmt ", 4) == 0) {
      unsigned char fm[40];
      const uint32_t n = sz < 40 ? sz : 40;
      if (n < 16 || fread(fm, 1, n, f) != n) return -1;
      h.format = fm[0] | (fm[1] << 8);
      h.channels = fm[2] | (fm[3] << 8);
      h.rate = (int)((uint32_t)fm[4] | ((uint32_t)fm[5] << 8) | ((uint32_t)fm[6] << 16) | ((uint32_t)fm[7] << 24));
      h.bits = fm[14] | (fm[15] << 8);
      if (h.format == 0xFFFE && n >= 26) h.format = fm[24] | (fm[25] << 8);  // WAVE_FORMAT_EXTENSIBLE: sub-format GUID prefix
      if (fseek(f, (long)(sz - n) + (long)(sz & 1), SEEK_CUR) != 0) return -1;
      have_fmt = true;
    } else if (memcmp(ch, "data", 4) == 0) {
      if (!have_fmt) return -1;
      h.data_offset = ftell(f);
      h.data_bytes = sz;
      return 0;
    } else {
      if (fseek(f, (long)sz + (long)(sz & 1), SEEK_CUR) != 0) return -1;
    }
  }
}

}  // namespace

extern "C" {

int64_t cst_batch_by_size(const int64_t* num_tokens, int64_t n, int64_t max_tokens, int64_t max_sentences, int32_t bsz_mult,
                          int64_t* batch_sizes) {
  if (!num_tokens || !batch_sizes || n < 0 || bsz_mult < 1) { cst_set_error("cst_batch_by_size: bad argument"); return CST_ERR_BAD_ARG; }
  int64_t start = 0, sample_len = 0, nb = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t nt = num_tokens[i];
    if (nt > sample_len) sample_len = nt;
    if (max_tokens > 0 && sample_len > max_tokens) {
      cst_set_error("sentence at position %lld of size %lld exceeds max_tokens limit of %lld!", (long long)i, (long long)sample_len, (long long)max_tokens);
      return CST_ERR_BAD_ARG;
    }
    const int64_t cur = i - start;
    if (batch_full(cur, (cur + 1) * sample_len, max_tokens, max_sentences)) {
      const int64_t a = bsz_mult * (cur / bsz_mult), b = cur % bsz_mult;
      const int64_t mod_len = a > b ? a : b;
      batch_sizes[nb++] = mod_len;
      start += mod_len;
      sample_len = 0;
      for (int64_t j = start; j <= i; ++j) sample_len = num_tokens[j] > sample_len ? num_tokens[j] : sample_len;
    }
  }
  if (n - start > 0) batch_sizes[nb++] = n - start;
  return nb;
}

int cst_wav_info(const char* path, int32_t* sample_rate, int32_t* channels, int64_t* frames, int32_t* bits) {
  FILE* f = path ? fopen(path, "rb") : nullptr;
  if (!f) { cst_set_error("cst_wav_info: cannot open %s", path ? path : "(null)"); return CST_ERR_BAD_ARG; }
  WavHeader h{};
  const int rc = parse_wav(f, h);
  fclose(f);
  if (rc != 0 || h.channels < 1 || h.bits < 8) { cst_set_error("cst_wav_info: %s is not a RIFF/WAVE PCM file", path); return CST_ERR_BAD_ARG; }
  if (sample_rate) *sample_rate = h.rate;
  if (channels) *channels = h.channels;
  if (bits) *bits = h.bits;
  if (frames) *frames = h.data_bytes / (h.channels * (h.bits / 8));
  return h.format == 1 && h.bits == 16 ? CST_OK : CST_ERR_UNSUPPORTED;
}

int64_t cst_wav_read_f32(const char* path, int64_t start_frame, int64_t nframes, float* out, int64_t capacity_samples) {
  FILE* f = path ? fopen(path, "rb") : nullptr;
  if (!f) { cst_set_error("cst_wav_read_f32: cannot open %s", path ? path : "(null)"); return CST_ERR_BAD_ARG; }
  WavHeader h{};
  if (parse_wav(f, h) != 0 || h.format != 1 || h.bits != 16 || h.channels < 1) {
    fclose(f);
    cst_set_error("cst_wav_read_f32: %s is not 16-bit PCM WAV", path);
    return CST_ERR_UNSUPPORTED;
  }
  const int64_t total = h.data_bytes / (2 * h.channels);
  if (start_frame < 0) start_frame = 0;
  if (start_frame > total) start_frame = total;
  int64_t want = nframes < 0 ? total - start_frame : nframes;
  if (start_frame + want > total) want = total - start_frame;
  const int64_t nsamp = want * h.channels;
  if (!out || nsamp > capacity_samples) { fclose(f); cst_set_error("cst_wav_read_f32: output buffer too small (%lld samples needed)", (long long)nsamp); return CST_ERR_WORKSPACE; }
  if (fseek(f, (long)(h.data_offset + start_frame * 2 * h.channels), SEEK_SET) != 0) { fclose(f); return CST_ERR_BAD_ARG; }
  int16_t buf[4096];
  int64_t done = 0;
  while (done < nsamp) {
    const size_t chunk = (size_t)((nsamp - done) < 4096 ? (nsamp - done) : 4096);
    const size_t got = fread(buf, 2, chunk, f);
    for (size_t i = 0; i < got; ++i) out[done + (int64_t)i] = (float)buf[i] * (1.0f / 32768.0f);  // libsndfile's int16 -> float scaling
    done += (int64_t)got;
    if (got < chunk) break;
  }
  fclose(f);
  return done / h.channels;
}

}  // extern "C"
