// ac_mv_avg.h -- drop-in for hlslibs/ac_dsp's include/ac_dsp/ac_mv_avg.h, backed by the MI355X engine.
//
// Same class template, constructor and run() signature as the reference (ac_mv_avg.h:135-196): the constructor borrows
// the coefficient array, run() reads the pending n_sample words (the last one wins, :151-163) and then filters whole
// frames of n_sample inputs while data is available (:164-189).  The per-sample window / MAC loop of ac_mv_avg_core
// (:111-123) is replaced by one batched call into libacdsp (acdsp_mvavg_*: mv_avg.hip).  The reference builds its core
// object inside run(), so no filter state survives a call -- neither here.
//
// ac_window_1d_flag (from <ac_window.h>, part of ac_types, not of ac_dsp) supplies the boundary behaviour in the reference;
// include/ac_types/ac_window.h states the semantics this engine implements.
#ifndef _INCLUDED_AC_MV_AVG_H_
#define _INCLUDED_AC_MV_AVG_H_

#include <ac_fixed.h>
#include <ac_window.h>
#include <ac_channel.h>
#include <mc_scverify.h>
#include <ac_dsp/acdsp_engine.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

template < int MAX_SAMPLE, int TAPS, ac_window_mode WIN_TYPE, class IN_TYPE, class OUT_TYPE, class ACC_TYPE, class COEFF_TYPE, class S_TYPE >
class ac_mv_avg
{
public:
  // public in the reference too: testbenches read the coefficient array back through it
  const COEFF_TYPE *const cff_ptr;

  ac_mv_avg(const COEFF_TYPE *const c_ptr) : cff_ptr(c_ptr), h_(0) { }
  ~ac_mv_avg() { if (h_) { acdsp_mvavg_destroy(h_); } }

  void CCS_BLOCK(run)(ac_channel < IN_TYPE > &data_in, ac_channel < OUT_TYPE > &data_out, ac_channel < S_TYPE > &n_sample) {
    bool have = false;
    S_TYPE n_sample_t = 0, sample = 0;
    while (n_sample.available(1)) {
      n_sample_t = n_sample.read();
      if (WIN_TYPE == AC_WIN) { sample = n_sample_t; } else { sample = n_sample_t + TAPS / 2; }   // S_TYPE arithmetic, as in the reference
      have = true;
    }
    if (!data_in.available(1)) { return; }
    if (!have) { die("run() with data but no n_sample word: the reference would use an uninitialised count"); }
    const long long n = (long long)n_sample_t.to_int64();
    const long long want = (WIN_TYPE == AC_WIN) ? n : n + TAPS / 2;
    if ((long long)sample.to_int64() != want) { die("n_sample + TAPS/2 does not fit S_TYPE: the reference's frame loop would break early"); }
    if (n < 1 || n > MAX_SAMPLE) { die("n_sample outside 1..MAX_SAMPLE: the reference's frame loop would lose alignment"); }
    std::vector<int64_t> x;
    while (data_in.available(1)) { x.push_back(acdsp::raw_of(data_in.read())); }
    if (x.size() % (size_t)n != 0) { die("input does not hold whole frames: the reference would read an empty channel"); }
    ensure();
    std::vector<int64_t> c((size_t)TAPS);
    for (int i = 0; i < TAPS; i++) { c[(size_t)i] = acdsp::raw_of(cff_ptr[i]); }   // borrowed pointer: read at every call
    acdsp::check(acdsp_mvavg_set_coeffs(h_, c.data()), "acdsp_mvavg_set_coeffs");
    const int64_t n_frames = (int64_t)(x.size() / (size_t)n);
    const int64_t opf = acdsp_mvavg_out_per_frame(h_, n);
    const int ib = acdsp_elem_bytes(IN_TYPE::width), ob = acdsp_elem_bytes(OUT_TYPE::width);
    std::vector<unsigned char> bi, bo((size_t)(opf * n_frames > 0 ? opf * n_frames : 1) * (size_t)ob);
    acdsp::pack(x, ib, bi);
    int64_t n_out = 0;
    acdsp::check(acdsp_mvavg_run_host(h_, bi.data(), n, n_frames, bo.data(), opf * n_frames > 0 ? opf * n_frames : 1, &n_out), "acdsp_mvavg_run_host");
    for (int64_t i = 0; i < n_out; i++) {
      data_out.write(acdsp::from_raw<OUT_TYPE>(acdsp::unpack_one(&bo[(size_t)i * (size_t)ob], ob, OUT_TYPE::sign)));
    }
  }

private:
  static void die(const char *why) {
    fprintf(stderr, "ac_mv_avg (MI355X engine): %s\n", why);
    abort();
  }
  void ensure() {
    if (h_) { return; }
    acdsp_mvavg_desc_t d;
    d.max_sample = MAX_SAMPLE; d.taps = TAPS;
    d.win_mode = WIN_TYPE == AC_WIN ? ACDSP_WIN_PLAIN : (WIN_TYPE == AC_MIRROR ? ACDSP_WIN_MIRROR : ACDSP_WIN_CLIP);
    d.n_objects = 1;
    d.in = acdsp::fmt_of<IN_TYPE>(); d.coeff = acdsp::fmt_of<COEFF_TYPE>(); d.acc = acdsp::fmt_of<ACC_TYPE>(); d.out = acdsp::fmt_of<OUT_TYPE>();
    d.device = acdsp::default_device(); d.flags = 0;
    acdsp::check(acdsp_mvavg_create(&d, &h_), "acdsp_mvavg_create");
  }
  ac_mv_avg(const ac_mv_avg &);
  ac_mv_avg &operator=(const ac_mv_avg &);
  acdsp_mvavg_t h_;
};

#endif
