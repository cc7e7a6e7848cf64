// TEST INFRASTRUCTURE: the RCCL names sdk_amd/csrc/comm.cpp uses, for the host emulation (see ../hip/hip_runtime.h).  One rank
// only: a collective over one rank is a copy.  A communicator of more ranks is refused -- several ranks on one device go through
// the library's loopback transport (sp_comm_create_custom), which needs no RCCL.
#pragma once
#include <hip/hip_runtime.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclUint32 = 3, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#define NCCL_UNIQUE_ID_BYTES 128
struct ncclUniqueId {
  char internal[128];
};
struct emu_nccl_comm {
  int nranks;
};
typedef emu_nccl_comm* ncclComm_t;

inline const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "emulated RCCL: one rank only"; }
inline ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id->internal, 0x5a, sizeof(id->internal));
  return ncclSuccess;
}
inline ncclResult_t ncclCommInitRank(ncclComm_t* c, int nranks, ncclUniqueId, int rank) {
  if (nranks != 1 || rank != 0) return ncclInvalidArgument;
  *c = new emu_nccl_comm{1};
  return ncclSuccess;
}
inline ncclResult_t ncclCommDestroy(ncclComm_t c) {
  delete c;
  return ncclSuccess;
}
inline size_t emu_nccl_size(ncclDataType_t t) { return t == ncclUint64 ? 8 : 4; }
inline ncclResult_t ncclReduceScatter(const void* send, void* recv, size_t recvcount, ncclDataType_t t, ncclRedOp_t, ncclComm_t, hipStream_t) {
  if (send != recv) std::memmove(recv, send, recvcount * emu_nccl_size(t));
  return ncclSuccess;
}
inline ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t t, ncclComm_t, hipStream_t) {
  if (send != recv) std::memmove(recv, send, sendcount * emu_nccl_size(t));
  return ncclSuccess;
}
