// comm.h -- RCCL (xGMI) collectives behind the C ABI (include/lrcn.h "data parallelism").  librccl is opened at run time
// (dlopen "librccl.so.1": the copy already mapped by the host program, e.g. torch's, or the system one), so liblrcn_hip has no
// link-time dependency on it and single-GPU users never load it.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct LrcnComm;  // opaque: RCCL communicator + the entry points resolved from librccl

// nullptr on failure, with a message in err (size errn)
LrcnComm *comm_create(int world, int rank, const void *unique_id128, char *err, size_t errn);
void comm_destroy(LrcnComm *c);
int comm_world(const LrcnComm *c);
// local check: librccl can be opened and every entry point used here resolves; 0 = yes
int comm_available(char *err, size_t errn);
// 128-byte RCCL unique id (host buffer); 0 on success
int comm_unique_id(void *out128, char *err, size_t errn);
// in-place all-reduce(SUM) of `count` floats on `stream`; several calls may be bracketed by comm_group_begin / _end
int comm_allreduce_f32(LrcnComm *c, float *buf, size_t count, hipStream_t stream, char *err, size_t errn);
int comm_group_begin(LrcnComm *c);
int comm_group_end(LrcnComm *c);
