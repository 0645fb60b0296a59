// wavesim runtime state — TEST INFRASTRUCTURE ONLY (see wavesim.h).
#include "wavesim.h"
namespace wavesim {
thread_local BlockCtx* g_block = nullptr;
thread_local int g_lane = 0;
thread_local int g_wave = 0;
}  // namespace wavesim
thread_local dim3 threadIdx;
thread_local dim3 blockIdx;
thread_local dim3 blockDim;
thread_local dim3 gridDim;
