// Host-side replay of the reference's episode sampler draws (test_phase/datasets/samplers.py:19-35): per episode one
// `np.random.choice(n_cat, n_cls, replace=False)` and, per chosen class, one `np.random.choice(catlocs[c], n_per, replace=False)` from the GLOBAL
// legacy numpy generator.  numpy's legacy `choice(replace=False)` is `permutation(n)[:k]`, i.e. a full Fisher-Yates shuffle of arange(n) driven by
// `random_interval` (masked rejection sampling on 32-bit MT19937 outputs) - ~3000 generator calls per 5-way episode of a 600-image class, ~100 us
// in numpy.  At 4000+ episodes/s per GPU (x 8 GPUs replaying one stream) the host sampler becomes the bottleneck of test_few_shot.evaluate, so the
// same arithmetic runs here on the generator state taken from `np.random.get_state()` and handed back with `set_state()`: the index stream and the
// generator state afterwards are bit-identical to numpy's (tests/test_sampler_native_cpu.py).  No GPU code in this file.
#include <stdint.h>
#include <stdlib.h>

#include <vector>

#include "../../include/fsvit.h"

int fsvit_set_error(int code, const char* fmt, ...);      // engine.hip

namespace {

struct Mt {
  uint32_t* key;
  int pos;
  void gen() {
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    const int N = 624, M = 397;
    uint32_t y;
    int i = 0;
    for (; i < N - M; ++i) { y = (key[i] & UPPER) | (key[i + 1] & LOWER); key[i] = key[i + M] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1)) & MATRIX_A); }
    for (; i < N - 1; ++i) { y = (key[i] & UPPER) | (key[i + 1] & LOWER); key[i] = key[i + (M - N)] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1)) & MATRIX_A); }
    y = (key[N - 1] & UPPER) | (key[0] & LOWER);
    key[N - 1] = key[M - 1] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1)) & MATRIX_A);
    pos = 0;
  }
  inline uint32_t next32() {
    if (pos == 624) gen();
    uint32_t y = key[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  // numpy random_interval(max): uniform on [0, max], max < 2^32 here
  inline uint32_t interval(uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = max;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    while ((v = next32() & mask) > max) {}
    return v;
  }
  // permutation(n): shuffle of arange(n), numpy's loop order (i = n-1 .. 1, j = interval(i), swap)
  inline void permutation(int64_t* buf, int64_t n) {
    for (int64_t i = 0; i < n; ++i) buf[i] = i;
    for (int64_t i = n - 1; i >= 1; --i) {
      const int64_t j = (int64_t)interval((uint32_t)i);
      const int64_t tmp = buf[i]; buf[i] = buf[j]; buf[j] = tmp;
    }
  }
};

}  // namespace

extern "C" int fsvit_sampler_draw(unsigned int* mt_key, int* mt_pos, const long long* cat_offsets, const long long* cat_items, int n_cat, int n_batch,
                                  int ep_per_batch, int n_cls, int n_per, long long* out) {
  if (!mt_key || !mt_pos || !cat_offsets || !cat_items || !out) return fsvit_set_error(FSVIT_ERR_ARG, "fsvit_sampler_draw: null argument");
  if (*mt_pos < 0 || *mt_pos > 624 || n_cat <= 0 || n_cls <= 0 || n_cls > n_cat || n_per <= 0 || n_batch < 0 || ep_per_batch <= 0)
    return fsvit_set_error(FSVIT_ERR_ARG, "fsvit_sampler_draw: bad argument");
  int64_t longest = n_cat;
  for (int c = 0; c < n_cat; ++c) {
    const int64_t len = cat_offsets[c + 1] - cat_offsets[c];
    if (len < n_per) return fsvit_set_error(FSVIT_ERR_ARG, "fsvit_sampler_draw: class %d has %lld items, fewer than n_per = %d", c, (long long)len, n_per);
    if (len >= ((int64_t)1 << 32)) return fsvit_set_error(FSVIT_ERR_ARG, "fsvit_sampler_draw: class too large");
    if (len > longest) longest = len;
  }
  std::vector<int64_t> perm((size_t)longest), cls((size_t)n_cat);
  Mt mt{mt_key, *mt_pos};
  long long* o = out;
  for (int b = 0; b < n_batch; ++b)
    for (int e = 0; e < ep_per_batch; ++e) {
      mt.permutation(cls.data(), n_cat);
      for (int k = 0; k < n_cls; ++k) {
        const int64_t c = cls[(size_t)k], base = cat_offsets[c], len = cat_offsets[c + 1] - base;
        mt.permutation(perm.data(), len);
        for (int i = 0; i < n_per; ++i) *o++ = cat_items[base + perm[(size_t)i]];
      }
    }
  *mt_pos = mt.pos;
  return 0;
}
