#define MPE_TU_T double
#include "mpe_small_tu.h"
