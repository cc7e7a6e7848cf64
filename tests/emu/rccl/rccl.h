// TEST INFRASTRUCTURE: the RCCL names sdk_amd/csrc/comm.cpp uses, for the host emulation (see ../hip/hip_runtime.h).
//
// Ranks are PROCESSES on this host; a communicator is a POSIX shared-memory segment named by the unique id: a process-shared
// barrier and one staging slot per rank.  A collective = every rank copies its send buffer into its slot, barrier, every rank
// computes its result from all slots, barrier.  That is the SEMANTICS of ncclReduceScatter / ncclAllGather (element counts,
// rank order, in-place forms), written independently of RCCL, so the call sequence of comm.cpp -- which so far has only met a
// real RCCL at one rank -- is checked against it with 2 and 4 ranks (tests/test_emulated_library.py).  Nothing about xGMI,
// rings or time.
#pragma once
#include <hip/hip_runtime.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclUint32 = 3, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#define NCCL_UNIQUE_ID_BYTES 128
struct ncclUniqueId {
  char internal[NCCL_UNIQUE_ID_BYTES];
};
struct emu_nccl_comm;
typedef emu_nccl_comm* ncclComm_t;

const char* ncclGetErrorString(ncclResult_t r);
ncclResult_t ncclGetVersion(int* version);
ncclResult_t ncclGetUniqueId(ncclUniqueId* id);
ncclResult_t ncclCommInitRank(ncclComm_t* c, int nranks, ncclUniqueId id, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t c);
// every rank sends nranks * recvcount elements; rank r receives the element-wise sum of all ranks' block r
ncclResult_t ncclReduceScatter(const void* send, void* recv, size_t recvcount, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s);
// every rank sends sendcount elements; every rank receives nranks * sendcount, rank k's block at k * sendcount
ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t t, ncclComm_t c, hipStream_t s);
