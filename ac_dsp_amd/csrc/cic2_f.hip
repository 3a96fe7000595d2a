// cic2_f.hip -- sixth translation unit of cic2.hip (compile time): the shapes of unit 5 in ACDSP_CIC2_SHAPES
#define ACDSP_CIC2_PART 5
#include "cic2.hip"
