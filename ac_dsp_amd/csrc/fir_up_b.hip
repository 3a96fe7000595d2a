// fir_up_b.hip -- translation unit 2 of the interpolating MFMA kernel: 16-bit samples, three coefficient digit planes (ac_poly_intr's pair
// taps, CIC interpolators whose boxcar^N taps pass 2^15).  Only instantiates; the kernel lives in fir_up.hip.
#define ACDSP_UP_TU 1
#include "fir_up.hip"
