// fir_mfma_alt2.hip -- translation unit 6 of the int8 MFMA FIR: the NAR instantiations (OUT_TYPEs of fewer than 16 bits, general rounding /
// overflow modes) of the pipelined kernel WITH a band skip, 5 .. 9 K-blocks (see fir_mfma.hip: launch_alt2_hs).  Only instantiates.
#define ACDSP_FIR_TU_MID 5
#include "fir_mfma.hip"
