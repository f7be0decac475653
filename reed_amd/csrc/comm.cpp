// Data-parallel gradient reduction on RCCL over xGMI (replaces accelerate -> DistributedDataParallel's
// bucketed NCCL all-reduce, image/train.py:151,293-295,401).  One process per GPU; the communicator
// owns a side stream so bucket reductions overlap the rest of backward on the compute stream:
//   allreduce_avg(bucket): comm stream waits for "bucket's grads written" (event on the compute stream),
//                          then ncclAllReduce(avg) in place;
//   sync():                compute stream waits for the last reduction before grad-norm / AdamW.
#include <rccl/rccl.h>
#include <string.h>

#include "../../include/reed_hip.h"
#include "common.hpp"

struct ReedComm {
  ncclComm_t comm;
  hipStream_t stream;
  hipEvent_t ready, done, gdone;
  int rank, world;
};

#define NCCL_TRY(x)                                                          \
  do {                                                                       \
    ncclResult_t r__ = (x);                                                  \
    if (r__ != ncclSuccess) {                                                \
      reed_set_error("RCCL error %d (%s) at %s", (int)r__, ncclGetErrorString(r__), #x); \
      return 2000 + (int)r__;                                                \
    }                                                                        \
  } while (0)
#define HIP_TRY(x)                                                           \
  do {                                                                       \
    hipError_t e__ = (x);                                                    \
    if (e__ != hipSuccess) {                                                 \
      reed_set_error("HIP error %s at %s", hipGetErrorString(e__), #x);      \
      return (int)e__;                                                       \
    }                                                                        \
  } while (0)

extern "C" int reed_comm_unique_id(void* out128) {
  REED_CHECK_ARG(out128, "comm_unique_id: null pointer");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  memcpy(out128, &id, sizeof(id));
  return REED_OK;
}

static int comm_init_body(ReedComm* c, const void* id128) {
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  NCCL_TRY(ncclCommInitRank(&c->comm, c->world, id, c->rank));
  int lo, hi;
  HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
  HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
  HIP_TRY(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->gdone, hipEventDisableTiming));
  return REED_OK;
}

extern "C" int reed_comm_init(const void* id128, int rank, int world, void** comm_out) {
  REED_CHECK_ARG(id128 && comm_out && world >= 1 && rank >= 0 && rank < world, "comm_init: bad args");
  ReedComm* c = new ReedComm();
  memset(c, 0, sizeof(*c));
  c->rank = rank;
  c->world = world;
  int rc = comm_init_body(c, id128);
  if (rc != REED_OK) {   // nothing half-built survives a failed init
    if (c->gdone) (void)hipEventDestroy(c->gdone);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->comm) (void)ncclCommAbort(c->comm);
    delete c;
    return rc;
  }
  *comm_out = c;
  return REED_OK;
}

extern "C" int reed_comm_allreduce_avg(void* comm, float* buf, int64_t count, void* compute_stream) {
  ReedComm* c = (ReedComm*)comm;
  REED_CHECK_ARG(c && buf && count > 0, "comm_allreduce_avg: bad args");
  HIP_TRY(hipEventRecord(c->ready, (hipStream_t)compute_stream));
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ready, 0));
  NCCL_TRY(ncclAllReduce(buf, buf, (size_t)count, ncclFloat32, ncclAvg, c->comm, c->stream));
  HIP_TRY(hipEventRecord(c->done, c->stream));
  return REED_OK;
}

// The same average as reduce-scatter + all-gather, both in place on the bucket: every rank reduces the 1 / world slice it
// owns (ncclReduceScatter with recv = send + rank * chunk) and the slices are gathered back (ncclAllGather with
// send = recv + rank * chunk). On a fully connected xGMI node each phase is one direct exchange with the 7 peers
// (SURVEY.md §5); which of the two forms RCCL runs faster is for the 8-GPU node to say (REED_COMM_ALGO=rsag).
// The count % world tail elements go through a plain all-reduce.
extern "C" int reed_comm_allreduce_avg_rsag(void* comm, float* buf, int64_t count, void* compute_stream) {
  ReedComm* c = (ReedComm*)comm;
  REED_CHECK_ARG(c && buf && count > 0, "comm_allreduce_avg_rsag: bad args");
  HIP_TRY(hipEventRecord(c->ready, (hipStream_t)compute_stream));
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ready, 0));
  const int64_t chunk = count / c->world, tail = count - chunk * c->world;
  if (chunk > 0) {
    float* mine = buf + chunk * c->rank;
    NCCL_TRY(ncclReduceScatter(buf, mine, (size_t)chunk, ncclFloat32, ncclAvg, c->comm, c->stream));
    NCCL_TRY(ncclAllGather(mine, buf, (size_t)chunk, ncclFloat32, c->comm, c->stream));
  }
  if (tail > 0) {
    float* t = buf + chunk * c->world;
    NCCL_TRY(ncclAllReduce(t, t, (size_t)tail, ncclFloat32, ncclAvg, c->comm, c->stream));
  }
  HIP_TRY(hipEventRecord(c->done, c->stream));
  return REED_OK;
}

extern "C" int reed_comm_sync(void* comm, void* compute_stream) {
  ReedComm* c = (ReedComm*)comm;
  REED_CHECK_ARG(c, "comm_sync: null communicator");
  HIP_TRY(hipStreamWaitEvent((hipStream_t)compute_stream, c->done, 0));
  return REED_OK;
}

// Factor exchange for weight gradients whose contraction length is the LOCAL batch (the adaLN matrix: dW = dmod^T
// silu(c), K = b): every rank contributes its [b, rows] factor, receives all of them ([world * b, rows], rank-major) and
// forms the global-batch product itself — world * b * rows bf16 on the wire instead of an all-reduce of rows * D fp32.
// Same stream discipline as the bucket reductions; completion has its own event so that the consumer does not wait
// for bucket reductions queued behind the last gather.
extern "C" int reed_comm_allgather(void* comm, const void* send, void* recv, int64_t bytes, void* compute_stream) {
  ReedComm* c = (ReedComm*)comm;
  REED_CHECK_ARG(c && send && recv && bytes > 0, "comm_allgather: bad args");
  HIP_TRY(hipEventRecord(c->ready, (hipStream_t)compute_stream));
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ready, 0));
  NCCL_TRY(ncclAllGather(send, recv, (size_t)bytes, ncclUint8, c->comm, c->stream));
  HIP_TRY(hipEventRecord(c->gdone, c->stream));
  HIP_TRY(hipEventRecord(c->done, c->stream));
  return REED_OK;
}

extern "C" int reed_comm_sync_gather(void* comm, void* compute_stream) {
  ReedComm* c = (ReedComm*)comm;
  REED_CHECK_ARG(c, "comm_sync_gather: null communicator");
  HIP_TRY(hipStreamWaitEvent((hipStream_t)compute_stream, c->gdone, 0));
  return REED_OK;
}

extern "C" int reed_comm_broadcast(void* comm, float* buf, int64_t count, int root, void* compute_stream) {
  ReedComm* c = (ReedComm*)comm;
  REED_CHECK_ARG(c && buf && count > 0, "comm_broadcast: bad args");
  HIP_TRY(hipEventRecord(c->ready, (hipStream_t)compute_stream));
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ready, 0));
  NCCL_TRY(ncclBroadcast(buf, buf, (size_t)count, ncclFloat32, root, c->comm, c->stream));
  HIP_TRY(hipEventRecord(c->done, c->stream));
  HIP_TRY(hipStreamWaitEvent((hipStream_t)compute_stream, c->done, 0));
  return REED_OK;
}

extern "C" int reed_comm_destroy(void* comm) {
  ReedComm* c = (ReedComm*)comm;
  if (!c) return REED_OK;
  (void)hipStreamSynchronize(c->stream);
  ncclCommDestroy(c->comm);
  (void)hipEventDestroy(c->ready);
  (void)hipEventDestroy(c->done);
  (void)hipEventDestroy(c->gdone);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return REED_OK;
}
