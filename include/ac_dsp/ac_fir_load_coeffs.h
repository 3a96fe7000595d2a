// ac_fir_load_coeffs.h -- drop-in for hlslibs/ac_dsp's loadable-coefficient FIR, MI355X back end.
//
// Same class template and run() signature as the reference
// (include/ac_dsp/ac_fir_load_coeffs.h:300-320).  run() keeps the reference's
// two phases (:324-364): (1) if a load flag is queued it is consumed, and when
// it is true *and* N_TAPS coefficients are queued they replace the stored set;
// (2) the input channel is drained through the filter.  The per-sample cores
// fir_load_coeffs_core::firLoadCoeffs* (:180-278) are the batched HIP kernels
// behind include/acdsp.h.  Coefficients are "don't care" until the first load
// (:311); here they start as zeros.
#ifndef _INCLUDED_AC_FIR_LOAD_COEFFS_H_
#define _INCLUDED_AC_FIR_LOAD_COEFFS_H_

#include <ac_fixed.h>
#include <ac_int.h>
#include <ac_channel.h>

#ifndef __FIR_FILTER_TYPES_ENUM_DEF__
#define __FIR_FILTER_TYPES_ENUM_DEF__
typedef enum { SHIFT_REG, ROTATE_SHIFT, C_BUFF, FOLD_EVEN, FOLD_ODD, TRANSPOSED, FOLD_EVEN_ANTI, FOLD_ODD_ANTI } FTYPE;
#endif

#include <mc_scverify.h>
#include <ac_dsp/acdsp_engine.h>

#ifndef __SYNTHESIS__
#include <iostream>
using namespace std;  // the reference header leaks this (ac_fir_load_coeffs.h:113); testbenches rely on it
#endif

template < class IN_TYPE, class OUT_TYPE, class COEFF_TYPE, class ACC_TYPE, unsigned N_TAPS, FTYPE ftype >
class ac_fir_load_coeffs
{
public:
  ac_fir_load_coeffs() : engine(ACDSP_FIR_LOAD, (int)ftype, (int)N_TAPS) {
    for (unsigned i = 0; i < N_TAPS; i++) { coeffs[i] = COEFF_TYPE(0); }
  }

#pragma hls_pipeline_init_interval 1
#pragma hls_design interface
  void CCS_BLOCK(run)(ac_channel < IN_TYPE > &data_in, ac_channel < COEFF_TYPE > &coeffs_ch, ac_channel < OUT_TYPE > &data_out, ac_channel < bool > &ld) {
    if (ld.available(1)) {
      bool ld_t = ld.read();
      if (ld_t && coeffs_ch.available(N_TAPS)) {
        for (unsigned i = 0; i < N_TAPS; i++) { coeffs[i] = coeffs_ch.read(); }
      }
    }
    std::vector<IN_TYPE> burst;
    while (data_in.available(1)) { burst.push_back(data_in.read()); }
    if (burst.empty()) { return; }
    std::vector<OUT_TYPE> result;
    engine.run_values_c(burst, result, coeffs);
    for (size_t i = 0; i < result.size(); i++) { data_out.write(result[i]); }
  }

private:
  acdsp::fir_engine<IN_TYPE, OUT_TYPE, COEFF_TYPE, ACC_TYPE> engine;
  COEFF_TYPE coeffs[N_TAPS];  // internal array that stores coefficient data
};

#endif
