// cic2_e.hip -- fifth translation unit of cic2.hip (compile time): the shapes of unit 4 in ACDSP_CIC2_SHAPES
#define ACDSP_CIC2_PART 4
#include "cic2.hip"
