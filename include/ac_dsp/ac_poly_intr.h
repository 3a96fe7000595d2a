// ac_poly_intr.h -- drop-in for hlslibs/ac_dsp's polyphase FIR interpolator, MI355X back end.
//
// Same class template and run() signature as the reference (include/ac_dsp/ac_poly_intr.h:275-320), including its own
// FTYPE enum { FOLD_EVEN, FOLD_ODD, FOLD_ANTI } under the __POLY_FILTER_TYPES_ENUM_DEF__ guard (:68-72) -- as in the
// reference, this header therefore cannot share a translation unit with the FIR headers' FTYPE.
// run() reads ONE flag from read_ctrl_chan per call (:290): true -> the control and coefficient structs are read from
// their channels (:291-294); false -> ONE input sample goes through the core selected by ftype, which writes IF outputs
// (the folded cores one sample late, nothing for the very first sample: accumulator banks acc_a / acc_b, :153-175).
// The cores run as a HIP kernel behind include/acdsp.h (acdsp_polyintr_*); one launch per sample here, whole bursts
// through the C ABI directly.
#ifndef _INCLUDED_AC_POLY_INTR_H_
#define _INCLUDED_AC_POLY_INTR_H_

#include <ac_fixed.h>
#include <ac_int.h>
#include <ac_channel.h>

#ifndef __POLY_FILTER_TYPES_ENUM_DEF__
#define __POLY_FILTER_TYPES_ENUM_DEF__
typedef enum { FOLD_EVEN, FOLD_ODD, FOLD_ANTI } FTYPE;
#endif

#include <mc_scverify.h>
#include <ac_dsp/acdsp_engine.h>

#include <vector>

template < class IN_TYPE, class COEFF_TYPE, class ACC_TYPE, class OUT_TYPE, class STR_CTRL_TYPE, class STR_COEFF_TYPE, int NTAPS, int COEFFSZ, int IF, FTYPE ftype >
class ac_poly_intr
{
public:
  ac_poly_intr() : h_(0), have_ctrl_(false) { }
  ~ac_poly_intr() { if (h_) { acdsp_polyintr_destroy(h_); } }

#pragma hls_pipeline_init_interval 1
#pragma hls_design interface
  void CCS_BLOCK(run)( ac_channel < IN_TYPE > &data_in, ac_channel < OUT_TYPE > &data_out, ac_channel < STR_CTRL_TYPE > &ctrl_st, ac_channel < STR_COEFF_TYPE > &coeffs_st, ac_channel < bool > &read_ctrl_chan) {
    bool read_ctrl = read_ctrl_chan.read();
    if (read_ctrl) {
      ctrl_t = ctrl_st.read();
      coeffs_t = coeffs_st.read();
      have_ctrl_ = true;
      return;
    }
    ensure();
    if (have_ctrl_) {
      std::vector<int64_t> c((size_t)COEFFSZ);
      std::vector<unsigned char> sg((size_t)IF), cr((size_t)IF);
      for (int i = 0; i < COEFFSZ; i++) { c[(size_t)i] = acdsp::raw_of(coeffs_t.coeffs[i]); }
      for (int j = 0; j < IF; j++) { sg[(size_t)j] = ctrl_t.sign[j] ? 1 : 0; cr[(size_t)j] = (unsigned char)(unsigned)(int)ctrl_t.corr[j]; }
      if (c != last_c_ || sg != last_sg_ || cr != last_cr_) {
        acdsp::check(acdsp_polyintr_set_ctrl(h_, c.data(), sg.data(), cr.data()), "acdsp_polyintr_set_ctrl");
        last_c_ = c; last_sg_ = sg; last_cr_ = cr;
      }
    }
    std::vector<int64_t> raw(1, acdsp::raw_of(data_in.read()));
    const int ib = acdsp_elem_bytes(IN_TYPE::width), ob = acdsp_elem_bytes(OUT_TYPE::width);
    std::vector<unsigned char> bi, bo((size_t)IF * (size_t)ob);
    acdsp::pack(raw, ib, bi);
    int64_t n_out = 0;
    acdsp::check(acdsp_polyintr_run_host(h_, bi.data(), 1, bo.data(), IF, &n_out), "acdsp_polyintr_run_host");
    for (int64_t i = 0; i < n_out; i++) {
      data_out.write(acdsp::from_raw<OUT_TYPE>(acdsp::unpack_one(&bo[(size_t)i * ob], ob, OUT_TYPE::sign)));
    }
  }

private:
  ac_poly_intr(const ac_poly_intr &);
  ac_poly_intr &operator=(const ac_poly_intr &);
  void ensure() {
    if (h_) { return; }
    acdsp_polyintr_desc_t d;
    d.n_taps = NTAPS; d.coeff_sz = COEFFSZ; d.ifac = IF; d.ftype = (int)ftype; d.n_channels = 1;
    d.in = acdsp::fmt_of<IN_TYPE>(); d.coeff = acdsp::fmt_of<COEFF_TYPE>(); d.acc = acdsp::fmt_of<ACC_TYPE>(); d.out = acdsp::fmt_of<OUT_TYPE>();
    d.device = acdsp::default_device(); d.flags = 0;
    acdsp::check(acdsp_polyintr_create(&d, &h_), "acdsp_polyintr_create");
  }
  acdsp_polyintr_t h_;
  bool have_ctrl_;
  STR_CTRL_TYPE ctrl_t;
  STR_COEFF_TYPE coeffs_t;
  std::vector<int64_t> last_c_;
  std::vector<unsigned char> last_sg_, last_cr_;
};

#endif
