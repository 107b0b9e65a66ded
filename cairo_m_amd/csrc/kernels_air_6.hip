// part 6 of the per-component AIR kernels: trace + histogram of the small opcode components in one launch
#define CM_AIR_PART 6
#include "kernels_air.inc"
