// part 4 of the per-component AIR kernels: the small-component batch kernel of the LogUp phase
#define CM_AIR_PART 4
#include "kernels_air.inc"
