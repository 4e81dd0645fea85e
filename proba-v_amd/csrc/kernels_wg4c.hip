// conv3_wgrad_w4_kernel, part 2 of its instances (kernels_wg4.hip: WG4_PART2) -- a translation unit of its own for the sake of the build time only.
#define WG4_PART 2
#include "kernels_wg4.hip"
