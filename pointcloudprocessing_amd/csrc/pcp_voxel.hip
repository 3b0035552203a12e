// Voxelizer / voxel stager kernels and their C ABI (see include/pcp.h).
#include <hip/hip_runtime.h>

#include "../../include/pcp.h"
#include "pcp_host.h"
