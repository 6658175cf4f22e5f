#define MPE_TU_T double
#include "mpe_tile_tu.h"
