#define MPE_TU_T double
#include "mpe_hier_tu.h"
