// Stand-in for <hip/hip_runtime.h> when kernel sources are compiled for the wave simulator (tests only).
#include "../wavesim.h"
