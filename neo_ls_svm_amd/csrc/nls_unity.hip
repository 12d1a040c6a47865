// Single translation unit of libneolssvm_hip.so (the kernel headers define non-inline __global__ functions,
// so the two host files are compiled together).
#include "nls_comm.hip"
#include "nls_evd.hip"
#include "nls_lib.hip"
#include "nls_dual.hip"
#include "nls_prestep.hip"
#include "nls_group.hip"
