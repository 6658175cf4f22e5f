#define MPE_TU_T float
#include "mpe_small_tu.h"
