// cic2_c.hip -- third translation unit of cic2.hip (compile time): the shapes of unit 2 in ACDSP_CIC2_SHAPES
#define ACDSP_CIC2_PART 2
#include "cic2.hip"
