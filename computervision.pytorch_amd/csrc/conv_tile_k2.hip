// conv_tile.hip's kernel for 3 and 4 pixel groups per wave (one of four translation units, compiled in parallel)
#define CVX_TILE_MT_A 3
#define CVX_TILE_MT_B 4
#define CVX_TILE_LAUNCH_FN cvx_conv_tile_launch_k2
#include "conv_tile_kernel.inc.h"
