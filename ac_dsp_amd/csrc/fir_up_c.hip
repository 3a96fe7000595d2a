// fir_up_c.hip -- translation unit 3 of the interpolating MFMA kernel: 32-bit samples (CIC interpolators, two or three coefficient digit
// planes).  Only instantiates; the kernel lives in fir_up.hip.
#define ACDSP_UP_TU 2
#include "fir_up.hip"
