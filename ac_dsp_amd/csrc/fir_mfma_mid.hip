// fir_mfma_mid.hip -- second translation unit of the int8 MFMA FIR: the register-resident kernel shapes for 11 / 13 / 15 / 17 K-blocks
// (258 - 513 taps) at one wave per SIMD.  The kernels and launch_nb_hs live in fir_mfma.hip; this unit only instantiates them
// (launch_fir_mfma_mid), so that the two compile side by side.
#define ACDSP_FIR_TU_MID 1
#include "fir_mfma.hip"
