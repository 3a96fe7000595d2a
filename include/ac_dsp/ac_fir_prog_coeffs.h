// ac_fir_prog_coeffs.h -- drop-in for hlslibs/ac_dsp's programmable-coefficient FIR, MI355X back end.
//
// Same class template and run() signature as the reference
// (include/ac_dsp/ac_fir_prog_coeffs.h:261-277): N_TAPS is an `int` here, ftype
// defaults to SHIFT_REG, and run() consumes at most ONE sample per call
// (`if (data_in.available(1))`, :281) with the coefficient array passed on that
// call.  A call of one sample is far below what a GPU launch is worth: calls of
// fewer than ACDSP_HOST_SMALL_MACS samples x taps (default 8192) run in the
// header's own ac_fixed loop (acdsp::fir_engine::run_values_c, the filter state
// moving between host and device as a state blob); with ACDSP_HOST_SMALL_MACS=0
// every call is one (tiny) GPU launch.  Streams that can be batched should use
// acdsp::fir_engine directly (include/ac_dsp/acdsp_engine.h).
#ifndef _INCLUDED_AC_FIR_PROG_COEFFS_H_
#define _INCLUDED_AC_FIR_PROG_COEFFS_H_

#include <ac_fixed.h>
#include <ac_int.h>
#include <ac_channel.h>

#ifndef __FIR_FILTER_TYPES_ENUM_DEF__
#define __FIR_FILTER_TYPES_ENUM_DEF__
typedef enum { SHIFT_REG, ROTATE_SHIFT, C_BUFF, FOLD_EVEN, FOLD_ODD, TRANSPOSED, FOLD_EVEN_ANTI, FOLD_ODD_ANTI } FTYPE;
#endif

#include <mc_scverify.h>
#include <ac_dsp/acdsp_engine.h>

template < class IN_TYPE, class OUT_TYPE, class COEFF_TYPE, class ACC_TYPE, int N_TAPS, FTYPE ftype = SHIFT_REG >
class ac_fir_prog_coeffs
{
public:
  ac_fir_prog_coeffs() : engine(ACDSP_FIR_PROG, (int)ftype, N_TAPS) { }

#pragma hls_pipeline_init_interval 1
#pragma hls_design interface
  void CCS_BLOCK(run)(ac_channel<IN_TYPE> &data_in, ac_channel<OUT_TYPE> &data_out, const COEFF_TYPE coeffs[N_TAPS]) {
    if (data_in.available(1)) {
      std::vector<IN_TYPE> one(1, data_in.read());
      std::vector<OUT_TYPE> result;
      engine.run_values_c(one, result, coeffs);   // one sample: below the break-even of a launch, the host-side loop of acdsp_engine.h
      data_out.write(result[0]);
    }
  }

private:
  acdsp::fir_engine<IN_TYPE, OUT_TYPE, COEFF_TYPE, ACC_TYPE> engine;
};

#endif
