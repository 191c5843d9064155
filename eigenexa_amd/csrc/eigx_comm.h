// eigx_comm.h -- inter-GPU transport of the multi-rank solvers (see comm.hip).
//
// Two layers:
//   * peer windows: every rank allocates its communication buffers collectively (comm_buffer), exports them with
//     hipIpcGetMemHandle and maps the other ranks' copies.  Kernels then STORE into the peers' HBM directly
//     (xGMI is point-to-point and every GPU pair of a node is linked) and signal with 8-byte epoch flags; a
//     one-wave wait kernel polls the flags on the consumer's stream.  This carries the latency-bound per-step
//     exchange of the reduction (replaces the hand-written reproducible allreduce, src/comm.F:2035-2580) and,
//     when RCCL cannot be used (several ranks sharing one GPU in the tests), the bulk collectives as well.
//   * RCCL communicators world / X / Y (ncclCommSplit, the MPI_Comm_split of src/eigen_libs0.F:579-585) for the
//     bulk collectives (panel allgather, eigenvector redistribution) when every rank owns its own GPU.
#pragma once
#include "eigx_context.h"
#include <string>
#include <map>

namespace eigx {

constexpr int EIGX_MAXP = 8;   // ranks of one xGMI node

// A device buffer that every rank allocated collectively and that is mapped on every peer.
struct PeerBuf {
  double* local = nullptr;
  double* peer[EIGX_MAXP] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // peer[me] == local
  size_t bytes = 0;
  bool mapped = false;   // false: plain local allocation (RCCL-only transport)
};

enum CommGroup { COMM_WORLD = 0, COMM_X = 1, COMM_Y = 2 };

// flag channels: one per (stream, purpose) so that operations in flight on different streams never share an epoch
enum CommChannel { CH_STEP = 0, CH_BULK = 1, CH_SIDE = 2, CH_STEP2 = 3, CH_STEPX = 4, CH_STEPX2 = 5, CH_COUNT = 6 };

struct CommState;

// members of a group as world ranks, in group order; returns the group size and my index in it
int comm_group(const Context& ctx, CommGroup grp, int* members, int* my_index);
int comm_size(const Context& ctx, CommGroup grp);
bool comm_failed(const Context& ctx);            // a collective timed out or RCCL returned an error
bool comm_shared_device(const Context& ctx);     // at least two ranks share a GPU (tests)
double comm_seconds(Context& ctx, bool reset);   // time spent in communication since the last reset (this rank)
int64_t comm_held_bytes(const Context& ctx);     // bytes of communication windows this rank holds (retired ones included)

// collectively (re)allocated, peer-mapped buffer; grows on demand (every rank must ask for the same size)
PeerBuf* comm_buffer(Context& ctx, const std::string& name, size_t bytes);

// recv.local[recv_off + r*count + i] = send_r[send_stride * (my index) + i] for every member r of the group
// (send_stride = 0: allgather; send_stride = count: all-to-all).  Enqueued on s; ch names the flag channel.
void comm_exchange(Context& ctx, CommGroup grp, const double* send, size_t send_stride, PeerBuf* recv, size_t recv_off,
                   size_t count, hipStream_t s, CommChannel ch);
// in place on a device buffer; every rank gets bit-identical results (fixed summation order over the members)
void comm_allreduce_sum(Context& ctx, CommGroup grp, double* buf, size_t count, hipStream_t s, CommChannel ch = CH_BULK);
void comm_allreduce_max(Context& ctx, CommGroup grp, double* buf, size_t count, hipStream_t s, CommChannel ch = CH_BULK);
// plain allgather into a local buffer (staged through an internal peer window)
void comm_allgather(Context& ctx, CommGroup grp, const double* send, double* recv, size_t count, hipStream_t s,
                    CommChannel ch = CH_BULK);

// exchange of large pieces into a plain local buffer: recv[r * count + i] = (member r's send)[(my index) * send_stride + i]
// (send_stride = count: all-to-all, 0: allgather); bounded hipIpc footprint (a bounce window of at most `bounce` doubles,
// eigx_tune key 9), one RCCL collective / grouped send + receive on a node
void comm_exchange_big(Context& ctx, CommGroup grp, const double* send, size_t send_stride, double* recv, size_t count,
                       hipStream_t s);
int comm_set_bounce(int doubles);

// ---- per-step exchanges of the reduction (double-buffered by message parity) -------------------------------------
// Two windows, one flag channel each.  which = 0 ("Y", CH_STEP): after the local mat-vec every rank sends the sums of its
// local rows / columns to the rank that OWNS the row for the panel work (band_reduce.hip: rows are dealt to the ranks in
// groups of 16), plus its share of the panel dot products and three scalars to everybody.  which = 1 ("X", CH_STEPX): the
// owner finishes W and the next x for its rows and sends them to everybody.  (The reference does the same two rounds per
// column with allreduces over X and Y: src/eigen_prd_t2.F:179,:203 and src/eigen_prd_t6_3.F:174,:296.)
// View handed to the producer kernel: where rank `me`'s message of the given parity lives in every rank's window, and
// which flag word announces it.
struct StepPeers {
  double* slot[EIGX_MAXP];                 // slot[q] = start of my message area (parity 0) in rank q's window
  unsigned long long* flag[EIGX_MAXP];     // flag[q] = my arrival flag (parity 0) in rank q's flag block; parity 1 at +EIGX_MAXP
  size_t parity_stride;                    // doubles between the parity-0 and parity-1 message areas
  size_t src_stride;                       // doubles between the messages of two sources in a window (reader side)
  unsigned* counter;                       // local last-workgroup counter
  int n;                                   // ranks
};
// (re)allocates the step window `which` for messages of msg_doubles each; returns my window (all sources, both parities):
// message of source q, parity p at  win + p * parity_stride + q * src_stride
double* comm_step_window(Context& ctx, int which, size_t msg_doubles, StepPeers* peers);
// one-wave kernel on s: wait until every rank's message `epoch` (1-based) of window `which` has arrived
void comm_step_wait(Context& ctx, int which, unsigned long long epoch, hipStream_t s);
// The same wait folded into the consumer kernel's prologue (saves the wait kernel's launch): the default when every rank
// has its own GPU -- on a shared card thousands of spinning consumer workgroups could keep the producers off the CUs --
// or when EIGX_FUSE_WAIT=1 asks for it (tests at sizes whose grids leave room); EIGX_FUSE_WAIT=0 forces the wait kernel.
// n = 0: nothing to wait for.
struct StepWait {
  const unsigned long long* flag;   // my flag block of the channel: parity p, source q at flag[p * EIGX_MAXP + q]
  int* err;
  unsigned long long* ticks;
  long long limit_ticks;
  unsigned long long epoch;
  int n;
  int naps;                         // ~60-ns naps between two polls
};
bool comm_step_wait_fused(const Context& ctx);
StepWait comm_step_wait_args(Context& ctx, int which, unsigned long long epoch);
// Collective form of the per-step exchanges (selected when the peer windows are not usable, or by EIGX_STEP=coll):
// the producer kernel writes its message into peers.slot[0] (a local send buffer; peers.n == 1, no flags) and
// comm_step_allgather delivers every rank's message into every rank's step window at the given parity --
// ncclAllGather over the world communicator on a node, the same group semantics through the peer windows when ranks
// share a card.  The consumer follows in stream order.
bool comm_step_collective(const Context& ctx);
void comm_step_allgather(Context& ctx, int which, const double* sendmsg, int parity, hipStream_t s);
// JSON description of the transports in use and of the init-time self-test (eigx_comm_info)
int comm_info(const Context& ctx, char* buf, int len);
// first epoch number of the next reduction's messages in window `which` (epochs are monotone over the life of the
// communicator); nmsg = an upper bound of the messages that follow
unsigned long long comm_step_epoch_base(Context& ctx, int which, unsigned long long nmsg);

}  // namespace eigx
