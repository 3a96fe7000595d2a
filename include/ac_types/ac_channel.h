// ac_channel.h -- minimal FIFO channel type for the ac_dsp_amd engine.
//
// Independent from-scratch subset of the AC Datatypes `ac_channel<T>` (an
// un-vendored dependency of hlslibs/ac_dsp, included at reference
// include/ac_dsp/ac_fir_const_coeffs.h:90).  Call sites it has to serve:
// read/write/available in the run() drain loops (reference
// ac_fir_const_coeffs.h:325-353) and debug_size in the CIC testbenches
// (reference tests/rtest_ac_cic_dec_full.cpp:108).
#ifndef AC_DSP_AMD_AC_CHANNEL_H
#define AC_DSP_AMD_AC_CHANNEL_H
#define __AC_CHANNEL_H

#include <cstdio>
#include <cstdlib>
#include <deque>
#include <iostream>

template <class T>
class ac_channel {
  std::deque<T> q;

public:
  typedef T element_type;

  ac_channel() {}
  explicit ac_channel(int init) { for (int i = 0; i < init; i++) { q.push_back(T()); } }
  ac_channel(int init, T val) { for (int i = 0; i < init; i++) { q.push_back(val); } }

  // Blocking read: in C simulation an empty channel is a design error.
  T read() {
    if (q.empty()) {
      fprintf(stderr, "ac_channel: read from an empty channel\n");
      abort();
    }
    T t = q.front();
    q.pop_front();
    return t;
  }
  void read(T &t) { t = read(); }
  void write(const T &t) { q.push_back(t); }

  bool nb_read(T &t) {
    if (q.empty()) { return false; }
    t = q.front();
    q.pop_front();
    return true;
  }
  bool nb_write(const T &t) { q.push_back(t); return true; }

  bool available(unsigned int k) const { return q.size() >= k; }
  unsigned int size() const { return (unsigned int)q.size(); }
  unsigned int debug_size() const { return (unsigned int)q.size(); }
  bool empty() const { return q.empty(); }
  void reset() { q.clear(); }

  // Indexed peek (element 0 is the next one to be read).
  const T &operator[](unsigned int i) const { return q[i]; }
  T &operator[](unsigned int i) { return q[i]; }
};

template <class T>
inline std::ostream &operator<<(std::ostream &os, const ac_channel<T> &c) {
  for (unsigned int i = 0; i < c.size(); i++) { os << (i ? " " : "") << c[i]; }
  return os;
}

#endif
