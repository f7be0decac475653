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

extern "C" int reed_comm_init(const void* id128, int rank, int world, void** comm_out) {
  REED_CHECK_ARG(id128 && comm_out && world >= 1 && rank >= 0 && rank < world, "comm_init: bad args");
  ReedComm* c = new ReedComm();
  c->rank = rank;
  c->world = world;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  NCCL_TRY(ncclCommInitRank(&c->comm, world, id, rank));
  int lo, hi;
  HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
  HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
  HIP_TRY(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->gdone, hipEventDisableTiming));
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
