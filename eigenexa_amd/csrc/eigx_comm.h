// eigx_comm.h -- collective wrappers over RCCL (see comm.hip).
#pragma once
#include "eigx_context.h"

namespace eigx {

struct ncclUniqueIdBlob { char internal[128]; };

struct CommState {
  void* world = nullptr;  // ncclComm_t
  void* x = nullptr;      // ranks sharing my py (size Px)
  void* y = nullptr;      // ranks sharing my px (size Py)
  bool callbacks = false; // host-staged test transport (see comm.hip)
};

bool comm_uses_callbacks(const Context& ctx);

enum CommGroup { COMM_WORLD = 0, COMM_X = 1, COMM_Y = 2 };

int comm_size(const Context& ctx, CommGroup grp);
// all in place on device buffers, enqueued on stream s; no-ops for groups of one rank
void comm_allreduce_sum(const Context& ctx, CommGroup grp, double* buf, size_t count, hipStream_t s);
void comm_allreduce_max(const Context& ctx, CommGroup grp, double* buf, size_t count, hipStream_t s);
void comm_bcast(const Context& ctx, CommGroup grp, double* buf, size_t count, int root, hipStream_t s);
void comm_allgather(const Context& ctx, CommGroup grp, const double* send, double* recv, size_t count,
                    hipStream_t s);

}  // namespace eigx
