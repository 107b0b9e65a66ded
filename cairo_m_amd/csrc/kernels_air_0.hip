// part 0 of the per-component AIR kernels (split only to parallelise compilation)
#define CM_AIR_PART 0
#include "kernels_air.inc"
