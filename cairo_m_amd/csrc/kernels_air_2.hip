// part 2 of the per-component AIR kernels (split only to parallelise compilation)
#define CM_AIR_PART 2
#include "kernels_air.inc"
