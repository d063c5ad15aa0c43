// One leaf of the split product build (crossclr_kernels_fast.h, CROSSCLR_SPLIT): this translation unit defines the launcher(s) guarded by
// CROSSCLR_TU_FWDP and thereby instantiates their kernel templates; crossclr_api.cpp only declares them.
#ifndef CROSSCLR_SPLIT
#error "tu_*.cpp are compiled by build.py with -DCROSSCLR_SPLIT"
#endif
#define CROSSCLR_TU_FWDP 1
#include "crossclr_kernels_fast.h"
