// pair_style dpd/meso (fp64 arithmetic): the fp64 instantiations of the ring kernel, compiled as a translation unit of their own
// (pair_ring.hip holds the kernel, the fp32 instantiations and the launcher)
#define RG_UNIT 2
#include "pair_ring.hip"
