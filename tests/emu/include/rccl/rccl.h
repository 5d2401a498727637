// TEST INFRASTRUCTURE: the handful of RCCL types pm_gather.hip names (it binds the functions with
// dlopen), for the CPU emulation build.  The values are those of rccl.h (ncclUint8 = 1, 128-byte id).
#pragma once
#include <hip/hip_runtime.h>
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
