// ac_poly_dec.h -- drop-in for hlslibs/ac_dsp's polyphase FIR decimator, MI355X back end.
//
// Same class template and run() signature as the reference (include/ac_dsp/ac_poly_dec.h:82-107):
// coefficients arrive as a struct { COEFF_TYPE coeffs[NTAPS*DF]; } on their own channel (the last one queued
// wins, :101-106), data is consumed in whole groups of DF samples (`while (data_in.available(DF))`, :109 --
// fewer than DF left-over samples stay in the channel) and every group yields one output.  The loop nest
// (:112-128) runs as HIP kernels behind include/acdsp.h.
#ifndef _INCLUDED_AC_POLY_DEC_H_
#define _INCLUDED_AC_POLY_DEC_H_

#include <ac_fixed.h>
#include <ac_channel.h>
#include <mc_scverify.h>
#include <ac_dsp/acdsp_engine.h>

template < class IN_TYPE, class COEFF_TYPE, class STR_COEFF_TYPE, class ACC_TYPE, class OUT_TYPE, int NTAPS, int DF >
class ac_poly_dec
{
public:
  ac_poly_dec() : h_(0), have_coeffs_(false) {}
  ~ac_poly_dec() { if (h_) { acdsp_polydec_destroy(h_); } }

#pragma hls_pipeline_init_interval 1
#pragma hls_design interface
  void CCS_BLOCK(run)( ac_channel < IN_TYPE > &data_in, ac_channel < OUT_TYPE > &data_out, ac_channel < STR_COEFF_TYPE > &coeffs_st ) {
    while (coeffs_st.available(1)) { coeffs_t = coeffs_st.read(); have_coeffs_ = true; }
    std::vector<int64_t> raw;
    while (data_in.available(DF)) {
      for (int i = 0; i < DF; i++) { raw.push_back(acdsp::raw_of(data_in.read())); }
    }
    if (raw.empty()) { return; }
    ensure();
    std::vector<int64_t> c((size_t)NTAPS * DF, 0);   // coefficients are zero until the first struct arrives (don't care in the reference)
    if (have_coeffs_) { for (int i = 0; i < NTAPS * DF; i++) { c[(size_t)i] = acdsp::raw_of(coeffs_t.coeffs[i]); } }
    if (c != last_c_) { acdsp::check(acdsp_polydec_set_coeffs(h_, c.data()), "acdsp_polydec_set_coeffs"); last_c_ = c; }
    const int ib = acdsp_elem_bytes(IN_TYPE::width), ob = acdsp_elem_bytes(OUT_TYPE::width);
    std::vector<unsigned char> bi, bo(raw.size() / DF * (size_t)ob);
    acdsp::pack(raw, ib, bi);
    acdsp::check(acdsp_polydec_run_host(h_, bi.data(), (int64_t)raw.size(), bo.data()), "acdsp_polydec_run_host");
    for (size_t i = 0; i < raw.size() / DF; i++) {
      data_out.write(acdsp::from_raw<OUT_TYPE>(acdsp::unpack_one(&bo[i * ob], ob, OUT_TYPE::sign)));
    }
  }

private:
  ac_poly_dec(const ac_poly_dec &);             // the engine handle is not shared: no copies
  ac_poly_dec &operator=(const ac_poly_dec &);
  void ensure() {
    if (h_) { return; }
    acdsp_polydec_desc_t d;
    d.n_taps = NTAPS; d.df = DF; d.n_channels = 1;
    d.in = acdsp::fmt_of<IN_TYPE>(); d.coeff = acdsp::fmt_of<COEFF_TYPE>(); d.acc = acdsp::fmt_of<ACC_TYPE>(); d.out = acdsp::fmt_of<OUT_TYPE>();
    d.device = acdsp::default_device(); d.flags = 0;
    acdsp::check(acdsp_polydec_create(&d, &h_), "acdsp_polydec_create");
  }
  acdsp_polydec_t h_;
  bool have_coeffs_;
  STR_COEFF_TYPE coeffs_t;
  std::vector<int64_t> last_c_;
};

#endif
