// conv_tile.hip's kernel for 1 and 2 pixel groups per wave (one of four translation units, compiled in parallel)
#define CVX_TILE_MT_A 1
#define CVX_TILE_MT_B 2
#define CVX_TILE_LAUNCH_FN cvx_conv_tile_launch_k1
#include "conv_tile_kernel.inc.h"
