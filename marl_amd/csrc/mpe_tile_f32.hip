#define MPE_TU_T float
#include "mpe_tile_tu.h"
