// TEST INFRASTRUCTURE (see hip_runtime.h next to this file): hipExtLaunchKernelGGL lives there.
#pragma once
#include <hip/hip_runtime.h>
