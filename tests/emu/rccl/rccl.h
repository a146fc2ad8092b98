// Stand-in for <rccl/rccl.h> in the CPU builds of tests/emu: the few type names the device library declares function
// pointers with. The library opens RCCL at run time and only for groups of distinct GPUs, which the emulation never forms.
#pragma once
typedef struct EmuNcclComm *ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
