// cic2_d.hip -- fourth translation unit of cic2.hip (compile time): the shapes of unit 3 in ACDSP_CIC2_SHAPES
#define ACDSP_CIC2_PART 3
#include "cic2.hip"
