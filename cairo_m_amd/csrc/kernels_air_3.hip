// part 3 of the per-component AIR kernels (split only to parallelise compilation)
#define CM_AIR_PART 3
#include "kernels_air.inc"
