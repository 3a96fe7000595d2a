// fir_mfma_mid3.hip -- translation unit 4 of the register-resident shapes of the int8 MFMA FIR: 27 / 29 / 31 K-blocks at one wave per
// SIMD (see fir_mfma_mid.hip).  Only instantiates; the kernels live in fir_mfma.hip.
#define ACDSP_FIR_TU_MID 3
#include "fir_mfma.hip"
