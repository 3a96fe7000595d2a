// ac_window.h -- minimal 1-D sliding window with start / end-of-line flags (subset of the AC window classes).
//
// hlslibs/ac_dsp's ac_mv_avg.h includes <ac_window.h> and uses exactly four things of it
// (reference include/ac_dsp/ac_mv_avg.h:100-122,146): the enum ac_window_mode with AC_WIN / AC_MIRROR / AC_CLIP,
// ac_window_1d_flag<T, AC_WN, AC_WMODE>::write(T, sol, eol), ::valid() and ::operator[](int).  The class itself ships
// with Catapult / hlslibs, NOT with ac_dsp, and is absent from this image, so its behaviour is RESTATED here from how
// ac_mv_avg drives it and from the documented meaning of the three modes -- parity unpinned by any reference artefact:
//   * the window holds the AC_WN most recent samples; operator[](0) is the centre, negative indices are older samples
//     (i = -AC_WN/2 the oldest, +AC_WN/2 the newest);
//   * sol marks the first sample of a line, eol the last; ac_mv_avg keeps writing AC_WN/2 more (don't-care) samples after
//     eol in the boundary modes so that the last real samples reach the centre (ac_mv_avg.h:146-166);
//   * AC_WIN: no boundary processing -- valid() only while all AC_WN entries belong to the current line
//     (n - AC_WN + 1 valid positions per line of n samples);
//   * AC_CLIP: positions before the line's first / after its last sample read that first / last sample;
//     AC_MIRROR: they read the sample mirrored about the first / last one (x[-k] = x[k], x[n-1+k] = x[n-1-k]);
//     in both, valid() while the centre holds a sample of the current line (n valid positions per line).
#ifndef AC_DSP_AMD_AC_WINDOW_H
#define AC_DSP_AMD_AC_WINDOW_H

enum ac_window_mode { AC_WIN = 1, AC_MIRROR = 2, AC_CLIP = 4 };

template <class T, int AC_WN, ac_window_mode AC_WMODE = AC_WIN>
class ac_window_1d_flag {
  static_assert(AC_WN >= 1 && (AC_WN & 1) == 1, "ac_dsp_amd ac_window_1d_flag: odd window sizes only");
  enum { H = AC_WN / 2 };

public:
  ac_window_1d_flag() : since_sol_(-1), eol_at_(-1) {
    for (int i = 0; i < AC_WN; i++) { data_[i] = T(); }
  }

  void write(T src, bool sol, bool eol) {
    for (int i = 0; i + 1 < AC_WN; i++) { data_[i] = data_[i + 1]; }
    data_[AC_WN - 1] = src;
    if (sol) { since_sol_ = 0; eol_at_ = -1; }
    else if (since_sol_ >= 0) { since_sol_++; }
    if (eol && since_sol_ >= 0) { eol_at_ = since_sol_; }
  }

  // position (within the line) of the sample at the window centre
  bool valid() const {
    if (since_sol_ < 0) { return false; }
    const int centre = since_sol_ - H;
    if (AC_WMODE == AC_WIN) { return since_sol_ >= AC_WN - 1 && (eol_at_ < 0 || since_sol_ <= eol_at_); }
    return centre >= 0 && (eol_at_ < 0 || centre <= eol_at_);
  }

  T &operator[](int i) {
    int pos = since_sol_ - H + i;          // line position the caller asks for
    const int last = eol_at_ >= 0 ? eol_at_ : since_sol_;
    if (AC_WMODE == AC_CLIP) {
      if (pos < 0) { pos = 0; }
      if (pos > last) { pos = last; }
    } else if (AC_WMODE == AC_MIRROR) {
      // reflect about the first / last sample until inside (a line shorter than the reach bounces)
      while (pos < 0 || pos > last) {
        if (last == 0) { pos = 0; break; }
        if (pos < 0) { pos = -pos; }
        if (pos > last) { pos = 2 * last - pos; }
      }
    }
    // the sample of line position p sits (since_sol_ - p) writes back
    int back = since_sol_ - pos;
    if (back < 0) { back = 0; }
    if (back > AC_WN - 1) { back = AC_WN - 1; }
    return data_[AC_WN - 1 - back];
  }

private:
  T data_[AC_WN];
  int since_sol_;   // writes since (and including) the last sol, minus one; -1 before the first line
  int eol_at_;      // line position of the eol sample, -1 while the line is open
};

#endif
