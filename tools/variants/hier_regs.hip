// scratch: register pressure of one hier_kernel instantiation under different occupancy hints
#include "../../marl_amd/csrc/mpe_hier_kernel.h"
#ifndef TT
#define TT float
#endif
namespace mpe {
template __global__ void hier_kernel<TT, 5, 10, 2, 256, true, true>(const KParams<TT>);
template __global__ void hier_kernel<TT, 5, 10, 2, 256, true, false>(const KParams<TT>);
template __global__ void hier_kernel<TT, 5, 10, 2, 256, false, true>(const KParams<TT>);
}
