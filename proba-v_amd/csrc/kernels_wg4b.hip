// conv3_wgrad_w4_kernel, part 1 of its instances (kernels_wg4.hip: WG4_PART1) -- a translation unit of its own for the sake of the build time only.
#define WG4_PART 1
#include "kernels_wg4.hip"
