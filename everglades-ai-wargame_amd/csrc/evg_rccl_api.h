// The part of RCCL's C API (rccl/rccl.h, the NCCL 2 ABI) that evg_abi.hip calls through pointers resolved with dlsym at run time: evg_comm_* /
// evg_gather_returns, include/evg.h.  Declared here so that libevg.so builds on a ROCm installation without the RCCL development files; where the real
// header is present the declarations are checked against it at compile time.
#pragma once
#include <cstddef>

#if defined(__has_include) && __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
static_assert(sizeof(ncclUniqueId) == 128 && (int)ncclSuccess == 0 && (int)ncclFloat == 7, "evg_rccl_api.h: RCCL's ABI differs from the local declarations");
#else
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;               // anything else is a failure; the text comes from ncclGetErrorString
typedef enum { ncclFloat = 7 } ncclDataType_t;
#endif
