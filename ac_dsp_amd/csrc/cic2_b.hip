// cic2_b.hip -- second translation unit of cic2.hip (compile time): the shapes of unit 1 in ACDSP_CIC2_SHAPES
#define ACDSP_CIC2_PART 1
#include "cic2.hip"
