// part 5 of the per-component AIR kernels: the small-component batch kernel of the constraint phase
#define CM_AIR_PART 5
#include "kernels_air.inc"
