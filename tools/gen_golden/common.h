// common.h -- helpers of the golden-vector generators (tools/gen_golden/*.cpp).
//
// The generators run ONLY in the build container: they compile the REFERENCE's own headers
// (/root/reference/include/ac_dsp/*.h, included by path on the compiler command line, never copied) over this
// repo's ac_types subset (include/ac_types) and dump raw-integer input / output vectors of the reference's loops
// to tests/golden/ref_hdr/*.json.  The committed JSON files are data; they are what pins the oracle and the HIP
// engine to the reference's actual `acc += reg[i] * coeffs[i]` source instead of to a restatement of it.
// (The number formats underneath are still this repo's ac_types subset: hlslibs/ac_types is not in the image.)
#pragma once
#include <ac_channel.h>
#include <ac_fixed.h>
#include <ac_int.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace gg {

inline uint64_t splitmix64(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// uniform raw word of T (all W bits random, sign-/zero-extended)
template <class T> inline T rnd(uint64_t &s) {
  const int W = T::width;
  __int128 v = (__int128)(int64_t)splitmix64(s);
  if (W < 64) { v = T::sign ? (v >> (64 - W)) : (__int128)((uint64_t)splitmix64(s) >> (64 - W)); }
  return T::from_raw128(v);
}
// random raw word limited to `bits` significant bits (small coefficients / inputs)
template <class T> inline T rnd_bits(uint64_t &s, int bits) {
  __int128 v = (__int128)((int64_t)splitmix64(s) >> (64 - bits));
  if (!T::sign) { v = v < 0 ? -v : v; }
  return T::from_raw128(v);
}
template <class T> inline long long raw(const T &x) { return (long long)x.raw128(); }

template <class T> inline std::string fmt_json() {
  char b[96];
  snprintf(b, sizeof b, "[%d, %d, %d, %d, %d]", (int)T::width, (int)T::i_width, (int)T::sign, (int)T::q_mode, (int)T::o_mode);
  return b;
}

struct Json {
  FILE *f;
  bool first_case;
  explicit Json(const std::string &path) : f(fopen(path.c_str(), "w")), first_case(true) {
    if (!f) { perror(path.c_str()); exit(1); }
    fprintf(f, "{\"generator\": \"tools/gen_golden (reference headers of hlslibs/ac_dsp v2026.1.1 over include/ac_types)\", \"cases\": [\n");
  }
  ~Json() { fprintf(f, "\n]}\n"); fclose(f); }
  void begin(const std::string &name) {
    fprintf(f, "%s{\"name\": \"%s\"", first_case ? "" : ",\n", name.c_str());
    first_case = false;
  }
  void end() { fprintf(f, "}"); }
  void num(const char *k, long long v) { fprintf(f, ", \"%s\": %lld", k, v); }
  void str(const char *k, const std::string &v) { fprintf(f, ", \"%s\": \"%s\"", k, v.c_str()); }
  void rawjson(const char *k, const std::string &v) { fprintf(f, ", \"%s\": %s", k, v.c_str()); }
  void arr(const char *k, const std::vector<long long> &v) {
    fprintf(f, ", \"%s\": [", k);
    for (size_t i = 0; i < v.size(); i++) { fprintf(f, "%s%lld", i ? "," : "", v[i]); }
    fprintf(f, "]");
  }
};

static const char *const kFtypeNames[] = {"SHIFT_REG", "ROTATE_SHIFT", "C_BUFF", "FOLD_EVEN", "FOLD_ODD", "TRANSPOSED", "FOLD_EVEN_ANTI", "FOLD_ODD_ANTI"};

// call-split pattern shared by the generators: sizes of consecutive run() calls covering n samples
inline std::vector<int> splits(int n, int mode) {
  std::vector<int> s;
  if (mode == 0) { s.push_back(n); return s; }
  const int pat[] = {1, 2, 9, 10, 10, 77, 130};
  int left = n, i = 0;
  while (left > 0) {
    int k = pat[i++ % 7];
    if (k > left) { k = left; }
    s.push_back(k);
    left -= k;
  }
  return s;
}

}  // namespace gg
