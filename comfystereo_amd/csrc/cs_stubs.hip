// cs_stubs.hip -- placeholders for kernels that are not written yet (they fail loudly).
#include "cs_kernels.h"
namespace cs {
int launch_blur(const float*, int, int, int, double, double, double, int, float*, float*, float*, float*, uint32_t*, int, hipStream_t) { return CS_EINVAL; }
size_t hybrid_workspace_bytes(int, int, int) { return 0; }
int hybrid_max_width() { return 0; }
int launch_hybrid(const RowArgs&, void*, hipStream_t) { return CS_EINVAL; }
size_t gpuwarp_workspace_bytes(int, int, int) { return 0; }
int gpuwarp_max_width() { return 0; }
int launch_gpuwarp_plain(const float*, const float*, int, int, int, double, double, double, double, float*, uint8_t*, uint32_t*, void*, hipStream_t) { return CS_EINVAL; }
int launch_gpuwarp_node(const cs_params*, const float*, const float*, const float*, int, uint32_t*, float*, float*, float*, float*, int, int, void*, hipStream_t) { return CS_EINVAL; }
}
