/* TEST INFRASTRUCTURE, not product code: a stand-in for librccl.so that lets ONE GPU play several devices.
 *
 * csrc/multi.hip (nbody_create_multi) drives one sub-context per device from one caller thread and moves data between
 * them with RCCL: a grouped in-place ncclAllGather of the positions and the symmetric algorithm's all-to-all as grouped
 * ncclSend / ncclRecv.  The test box has one GPU, and RCCL refuses two ranks on one device — so every `k > 0` index of that
 * file would never run.  With NBODY_RCCL_LIB naming this library and NBODY_MULTI_SHARE_DEVICE=1, multi.hip lists device 0
 * several times and the eight entry points below implement the same collectives as stream-ordered copies between the
 * parts' buffers:
 *
 *   - calls between ncclGroupStart and the matching ncclGroupEnd are only noted; the copies are issued at ncclGroupEnd;
 *   - every participating communicator records a "ready" event on its stream; a copy INTO rank k's buffer is issued on
 *     rank k's stream behind the source rank's ready event (the source's data is what its stream had produced when the
 *     collective was called);
 *   - then every rank records "done" and every rank's stream waits for every other rank's done: nobody overwrites a send
 *     buffer that a peer is still reading — the completion semantics a real collective gives the streams.
 *
 * Semantics checked, like RCCL: a rank missing from a grouped all-gather, unequal counts, a send without its receive ->
 * ncclInvalidUsage.  Counters (fake_rccl_counters) let the tests prove that the data really went through here.
 *
 *   gcc -shared -fPIC -O1 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/fake_rccl.c -o libfake_rccl.so -L/opt/rocm/lib -lamdhip64
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7,
               ncclFloat64 = 8 } ncclDataType_t;   /* the values of rccl.h (ncclFloat = ncclFloat32, ncclDouble = ncclFloat64) */

struct clique;
typedef struct fake_comm {
  int rank, n_ranks, device;
  struct clique *cl;
  hipEvent_t ready, done;
} fake_comm;
typedef fake_comm *ncclComm_t;

struct clique { int n, alive; fake_comm *c; };

enum { OP_ALLGATHER, OP_SEND, OP_RECV };
typedef struct { int kind, peer, matched; fake_comm *comm; const void *send; void *recv; size_t bytes; hipStream_t stream; } op_t;

#define MAX_OPS 8192
static op_t ops[MAX_OPS];
static int n_ops = 0, depth = 0;
static int64_t n_allgather = 0, n_send = 0, n_recv = 0, n_copies = 0, bytes_copied = 0, n_groups = 0;

#define EXPORT __attribute__((visibility("default")))

static size_t type_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}

static ncclResult_t copy_into(const op_t *dst_op, void *dst, const op_t *src_op, const void *src, size_t bytes) {
  /* on the destination rank's stream, behind the source rank's ready event */
  if (hipSetDevice(dst_op->comm->device) != hipSuccess) return ncclUnhandledCudaError;
  if (src_op->comm != dst_op->comm && hipStreamWaitEvent(dst_op->stream, src_op->comm->ready, 0) != hipSuccess) return ncclUnhandledCudaError;
  if (src != dst && bytes) {
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, dst_op->stream) != hipSuccess) return ncclUnhandledCudaError;
    ++n_copies; bytes_copied += (int64_t)bytes;
  }
  return ncclSuccess;
}

static ncclResult_t flush(void) {
  ncclResult_t rc = ncclSuccess;
  /* one stream per communicator and group: what multi.hip does, and what keeps the event logic below simple */
  for (int a = 0; a < n_ops && rc == ncclSuccess; ++a)
    for (int b = 0; b < a; ++b)
      if (ops[a].comm == ops[b].comm && ops[a].stream != ops[b].stream) rc = ncclInvalidUsage;
  /* ready events: once per communicator */
  for (int a = 0; a < n_ops && rc == ncclSuccess; ++a) {
    int first = 1;
    for (int b = 0; b < a; ++b) if (ops[b].comm == ops[a].comm) first = 0;
    if (!first) continue;
    if (hipSetDevice(ops[a].comm->device) != hipSuccess || hipEventRecord(ops[a].comm->ready, ops[a].stream) != hipSuccess) rc = ncclUnhandledCudaError;
  }
  for (int a = 0; a < n_ops && rc == ncclSuccess; ++a) {
    op_t *o = &ops[a];
    if (o->kind == OP_ALLGATHER) {
      for (int q = 0; q < o->comm->n_ranks && rc == ncclSuccess; ++q) {
        const op_t *src = NULL;
        for (int b = 0; b < n_ops; ++b)
          if (ops[b].kind == OP_ALLGATHER && ops[b].comm == &o->comm->cl->c[q]) { src = &ops[b]; break; }
        if (!src || src->bytes != o->bytes) { rc = ncclInvalidUsage; break; }     /* a rank missing from the group / unequal counts */
        rc = copy_into(o, (char *)o->recv + (size_t)q * o->bytes, src, src->send, o->bytes);
      }
    } else if (o->kind == OP_RECV) {
      op_t *src = NULL;
      if (o->peer < 0 || o->peer >= o->comm->n_ranks) { rc = ncclInvalidArgument; break; }
      for (int b = 0; b < n_ops; ++b)
        if (ops[b].kind == OP_SEND && !ops[b].matched && ops[b].comm == &o->comm->cl->c[o->peer] && ops[b].peer == o->comm->rank) { src = &ops[b]; break; }
      if (!src || src->bytes != o->bytes) { rc = ncclInvalidUsage; break; }
      src->matched = 1; o->matched = 1;
      rc = copy_into(o, o->recv, src, src->send, o->bytes);
    }
  }
  for (int a = 0; a < n_ops && rc == ncclSuccess; ++a)
    if (ops[a].kind == OP_SEND && !ops[a].matched) rc = ncclInvalidUsage;           /* a send nobody receives */
  /* done events, then everybody waits for everybody */
  for (int a = 0; a < n_ops && rc == ncclSuccess; ++a) {
    int first = 1;
    for (int b = 0; b < a; ++b) if (ops[b].comm == ops[a].comm) first = 0;
    if (!first) continue;
    if (hipSetDevice(ops[a].comm->device) != hipSuccess || hipEventRecord(ops[a].comm->done, ops[a].stream) != hipSuccess) rc = ncclUnhandledCudaError;
  }
  for (int a = 0; a < n_ops && rc == ncclSuccess; ++a) {
    int first = 1;
    for (int b = 0; b < a; ++b) if (ops[b].comm == ops[a].comm) first = 0;
    if (!first) continue;
    if (hipSetDevice(ops[a].comm->device) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
    for (int b = 0; b < n_ops && rc == ncclSuccess; ++b) {
      int first_b = 1;
      for (int c = 0; c < b; ++c) if (ops[c].comm == ops[b].comm) first_b = 0;
      if (!first_b || ops[b].comm == ops[a].comm || ops[b].comm->cl != ops[a].comm->cl) continue;
      if (hipStreamWaitEvent(ops[a].stream, ops[b].comm->done, 0) != hipSuccess) rc = ncclUnhandledCudaError;
    }
  }
  n_ops = 0;
  ++n_groups;
  return rc;
}

static ncclResult_t note(int kind, fake_comm *comm, const void *send, void *recv, size_t bytes, int peer, hipStream_t stream) {
  if (!comm || !comm->cl || !comm->cl->alive) return ncclInvalidArgument;
  if (n_ops == MAX_OPS) return ncclInternalError;
  ops[n_ops++] = (op_t){kind, peer, 0, comm, send, recv, bytes, stream};
  return depth == 0 ? flush() : ncclSuccess;
}

EXPORT ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *devlist) {
  if (!comms || n < 1) return ncclInvalidArgument;
  struct clique *cl = calloc(1, sizeof *cl);
  fake_comm *c = calloc((size_t)n, sizeof *c);
  if (!cl || !c) { free(cl); free(c); return ncclSystemError; }
  cl->n = n; cl->alive = n; cl->c = c;
  for (int k = 0; k < n; ++k) {
    c[k].rank = k; c[k].n_ranks = n; c[k].device = devlist ? devlist[k] : k; c[k].cl = cl;
    if (hipSetDevice(c[k].device) != hipSuccess || hipEventCreateWithFlags(&c[k].ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c[k].done, hipEventDisableTiming) != hipSuccess)
      return ncclUnhandledCudaError;
    comms[k] = &c[k];
  }
  return ncclSuccess;
}

EXPORT ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (!comm || !comm->cl) return ncclInvalidArgument;
  struct clique *cl = comm->cl;
  (void)hipSetDevice(comm->device);
  (void)hipEventDestroy(comm->ready); (void)hipEventDestroy(comm->done);
  comm->cl = NULL;
  if (--cl->alive == 0) { free(cl->c); free(cl); }
  return ncclSuccess;
}

EXPORT ncclResult_t ncclGroupStart(void) { ++depth; return ncclSuccess; }

EXPORT ncclResult_t ncclGroupEnd(void) {
  if (depth == 0) return ncclInvalidUsage;
  if (--depth > 0) return ncclSuccess;
  return flush();
}

EXPORT ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
  ++n_allgather;
  return note(OP_ALLGATHER, comm, sendbuff, recvbuff, sendcount * type_size(datatype), -1, stream);
}

EXPORT ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  ++n_send;
  return note(OP_SEND, comm, sendbuff, NULL, count * type_size(datatype), peer, stream);
}

EXPORT ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  ++n_recv;
  return note(OP_RECV, comm, NULL, recvbuff, count * type_size(datatype), peer, stream);
}

EXPORT const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake RCCL: unhandled HIP error";
    case ncclInvalidArgument: return "fake RCCL: invalid argument";
    case ncclInvalidUsage: return "fake RCCL: invalid usage (a rank missing from a group, unequal counts, or an unmatched send/recv)";
    default: return "fake RCCL: internal error";
  }
}

/* {all-gathers, sends, receives, copies issued, bytes copied, groups flushed} since the library was loaded */
EXPORT void fake_rccl_counters(int64_t out[6]) {
  out[0] = n_allgather; out[1] = n_send; out[2] = n_recv; out[3] = n_copies; out[4] = bytes_copied; out[5] = n_groups;
}
