// fir_mfma_mid2.hip -- translation unit 3 of the register-resident shapes of the int8 MFMA FIR: 19 / 21 / 23 / 25 K-blocks at one wave per
// SIMD (see fir_mfma_mid.hip).  Only instantiates; the kernels live in fir_mfma.hip.
#define ACDSP_FIR_TU_MID 2
#include "fir_mfma.hip"
