// ac_fir_reg_share.h -- drop-in for hlslibs/ac_dsp's register-sharing FIR, MI355X back end.
//
// Same class template, constructor and member signatures as the reference
// (include/ac_dsp/ac_fir_reg_share.h:267-313): the shift register is an array OWNED BY THE CALLER (several filter
// objects may share it, hence the name), run() takes one sample, the coefficient memory and one output by reference.
// What run() does here:
//   1. firShiftReg(data_in) on the caller's array, exactly as the reference (:120-126, :286) -- the array stays
//      readable / shareable between calls;
//   2. the coefficient memory is resolved to tap order: tap t of the MAC loops reads
//      coeffs[(t / BLK_SZ) * MEM_WORD_WIDTH + BLK_OFFSET + t % BLK_SZ]  (:143-146 and the four fold cores);
//   3. the MAC core selected by ftype (SHIFT_REG, FOLD_EVEN, FOLD_EVEN_ANTI, FOLD_ODD, FOLD_ODD_ANTI; ascending tap
//      order, :136-260) is evaluated on the GPU from the register contents (stateless: the register array, not the
//      engine handle, is the filter state).
// One launch per sample: this class exists for source compatibility; streams use
// acdsp::fir_engine<IN, OUT, COEFF, ACC>(ACDSP_FIR_REG_SHARE, ftype, N_TAPS, n_channels) with tap-ordered coefficients.
// Where the reference would index outside reg[] / coeffs[] (tap count of the loop not a multiple of BLK_SZ, coefficient
// address >= N_TAPS) or leave the output unassigned (other ftypes) this header aborts with a message.
#ifndef _INCLUDED_AC_FIR_REG_SHARE_H_
#define _INCLUDED_AC_FIR_REG_SHARE_H_

#include <ac_fixed.h>
#include <ac_int.h>
#include <ac_channel.h>

#ifndef __FIR_FILTER_TYPES_ENUM_DEF__
#define __FIR_FILTER_TYPES_ENUM_DEF__
typedef enum { SHIFT_REG, ROTATE_SHIFT, C_BUFF, FOLD_EVEN, FOLD_ODD, TRANSPOSED, FOLD_EVEN_ANTI, FOLD_ODD_ANTI } FTYPE;
#endif

#include <mc_scverify.h>
#include <ac_dsp/acdsp_engine.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef ac_fixed < 16, 1, true > DEFAULT_TYPE;

template < int N_TAPS = 2, class IN_TYPE = DEFAULT_TYPE, class OUT_TYPE = DEFAULT_TYPE, class COEFF_TYPE = DEFAULT_TYPE,
           class ACC_TYPE = DEFAULT_TYPE, int MEM_WORD_WIDTH = 1, int BLK_SZ = 1, int BLK_OFFSET = 0, FTYPE ftype = SHIFT_REG >
class ac_fir_reg_share
{
public:
  ac_fir_reg_share(IN_TYPE *ptr_t) : ptr(ptr_t), engine(ACDSP_FIR_REG_SHARE, (int)ftype, N_TAPS) { }

  void CCS_BLOCK(run)(IN_TYPE &data_in, COEFF_TYPE coeffs[N_TAPS], OUT_TYPE &data_out) {
    for (int i = N_TAPS - 1; i >= 0; i--) { ptr[i] = (i == 0) ? data_in : ptr[i - 1]; }
    const int count = (ftype == SHIFT_REG) ? N_TAPS
                      : (ftype == FOLD_EVEN || ftype == FOLD_EVEN_ANTI) ? N_TAPS / 2
                      : (ftype == FOLD_ODD || ftype == FOLD_ODD_ANTI) ? ((N_TAPS - 1) / 2) + 1 : -1;
    if (count < 0) { die("run() has no branch for this ftype: the reference leaves data_out unassigned"); }
    if (BLK_SZ < 1 || count % BLK_SZ != 0) { die("the MAC loop's tap count is not a multiple of BLK_SZ: the reference reads outside reg[]"); }
    std::vector<COEFF_TYPE> tap((size_t)N_TAPS, COEFF_TYPE(0));
    for (int t = 0; t < count; t++) {
      const int addr = (t / BLK_SZ) * MEM_WORD_WIDTH + BLK_OFFSET + t % BLK_SZ;
      if (addr < 0 || addr >= N_TAPS) { die("coefficient address outside coeffs[N_TAPS]"); }
      tap[(size_t)t] = coeffs[addr];
    }
    engine.set_coeffs(tap.data());
    // the register array is the state: replay it oldest-first through a cleared handle, keep the newest output
    std::vector<IN_TYPE> x((size_t)N_TAPS);
    for (int i = 0; i < N_TAPS; i++) { x[(size_t)i] = ptr[N_TAPS - 1 - i]; }
    std::vector<OUT_TYPE> y;
    engine.reset();
    engine.run_values(x, y);
    data_out = y[(size_t)N_TAPS - 1];
  }

  void ac_firProgCoeffs_delay_line(OUT_TYPE &core_out) {
    core_out = ptr[N_TAPS - 1];
  }

private:
  static void die(const char *why) {
    fprintf(stderr, "ac_fir_reg_share (MI355X engine): %s\n", why);
    abort();
  }
  IN_TYPE *ptr;
  acdsp::fir_engine<IN_TYPE, OUT_TYPE, COEFF_TYPE, ACC_TYPE> engine;
};

#endif
