// fir_mfma_alt.hip -- translation unit 5 of the int8 MFMA FIR: the pipelined kernel's instantiations for OUT_TYPEs of fewer than 16 bits (NAR)
// and for 4-byte output containers (W4), up to 9 K-blocks (see fir_mfma.hip: MfmaArgs, launch_nb_hs).  Only instantiates.
#define ACDSP_FIR_TU_MID 4
#include "fir_mfma.hip"
