// ac_cic_intr_full.h -- drop-in for hlslibs/ac_dsp's full-precision CIC interpolator, MI355X back end.
//
// Same class template and run() signature as the reference
// (include/ac_dsp/ac_cic_intr_full.h:137-153).  Comb chain, zero-stuffing
// sequencer and integrator chain (:173-215, cores in ac_cic_full_core.h:143-160,
// 211-255) are one HIP kernel behind include/acdsp.h.  Output count per call is
// the reference's: L inputs give (L-1)*R+1 results on the first call (minus the
// N-1 start-up values dropped at :209-213) and L*R on later calls, because the
// R-1 zero-stuffed results after the last input only appear once more input
// arrives (:200-205).
#ifndef _INCLUDED_AC_CIC_INTR_FULL_H_
#define _INCLUDED_AC_CIC_INTR_FULL_H_

#include <ac_channel.h>
#include <ac_fixed.h>

#include <mc_scverify.h>
#include <ac_dsp/acdsp_engine.h>

#ifndef __SYNTHESIS__
#include <iostream>
using namespace std;  // leaked by the reference through ac_cic_full_core.h:53; testbenches rely on it
#endif

#ifndef AC_DSP_AMD_CIC_POWER_DEF
#define AC_DSP_AMD_CIC_POWER_DEF
template <int base, int expon>
struct power {
  enum { value = base * power<base, expon - 1>::value };
};
template <int base>
struct power<base, 0> {
  enum { value = 1 };
};
#endif

template <class IN_TYPE, unsigned R, unsigned M, unsigned N>
struct find_inter_type_cic_intr {
  enum {
    W    = IN_TYPE::width,
    I    = IN_TYPE::i_width,
    S    = IN_TYPE::sign,
    outF = W - I,
    outW = ac::log2_ceil<power<R, N - 1>::value*power<M, N>::value>::val + W + int(!S),
    outI = outW - outF
  };
  typedef ac_fixed<outW, outI, true> INT_TYPE;
};

template < class IN_TYPE, class OUT_TYPE, unsigned R_, unsigned M_, unsigned N_ >
class ac_cic_intr_full
{
public:
  ac_cic_intr_full() : engine(true, R_, M_, N_) { }

#pragma hls_pipeline_init_interval 1
#pragma hls_design interface
  void CCS_BLOCK(run)(ac_channel < IN_TYPE > &data_in, ac_channel < OUT_TYPE > &data_out) {
    std::vector<IN_TYPE> burst;
    while (data_in.available(1)) { burst.push_back(data_in.read()); }
    std::vector<OUT_TYPE> result;
    engine.run_values(burst, result);
    for (size_t i = 0; i < result.size(); i++) { data_out.write(result[i]); }
  }

private:
  typedef typename find_inter_type_cic_intr <IN_TYPE, R_, M_, N_>::INT_TYPE INT_TYPE;
  static_assert(INT_TYPE::width <= 128, "ac_dsp_amd engine: CIC intermediate type limited to 128 bits");
  acdsp::cic_engine<IN_TYPE, OUT_TYPE> engine;
};

#endif
