#define MPE_TU_T float
#include "mpe_hier_tu.h"
