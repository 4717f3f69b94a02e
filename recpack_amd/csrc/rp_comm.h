// The communicator's operations for the library's own distributed entry points (rp_ease_dist.hip implements them).
#pragma once
#include "rp_common.h"

int rp_comm_world(const rphip_comm* c);
int rp_comm_rank(const rphip_comm* c);
bool rp_comm_is_loopback(const rphip_comm* c);  // rphip_comm_loopback_create: the other ranks' data never arrives
// loop-back only: the all-gathers that follow put this rank's own part into EVERY slot whatever its size (large parts are otherwise
// only written, as a peer's incoming data would be) - for callers that compute with what the other ranks sent
void rp_comm_loop_replicate(rphip_comm* c, bool on);
// stream-ordered on `st`; see rphip_comm_ops in include/recpack_hip.h for the semantics
int rp_comm_broadcast(rphip_comm* c, void* buf, int64_t bytes, int root, hipStream_t st);
int rp_comm_all_gather(rphip_comm* c, const void* send, void* recv, int64_t bytes_per_rank, hipStream_t st);
int rp_comm_exchange(rphip_comm* c, const void* send, const int64_t* soff, const int64_t* sbytes, void* recv, const int64_t* roff,
                     const int64_t* rbytes, hipStream_t st);
int rp_comm_all_reduce_sum_i32(rphip_comm* c, int32_t* buf, int64_t count, hipStream_t st);
