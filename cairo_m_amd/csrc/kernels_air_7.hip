// part 7 of the per-component AIR kernels: trace + histogram of a large opcode component in one launch
#define CM_AIR_PART 7
#include "kernels_air.inc"
