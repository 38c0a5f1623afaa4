// Test double for librccl (tests only; selected with HX_RCCL_LIB): the nine entry points libhx binds (csrc/comm.hip), for
// ranks that SHARE one GPU -- RCCL itself refuses two ranks on one device, and the GPU boxes of this project have one.
// It lets the library-issued exchange path (hx_comm_set_plan / hx_comm_exchange / hx_comm_allreduce, the callbacks of
// hx_cocg_sharded / hx_gmres_sharded / the sharded V-cycle) run end to end with N processes: same plans, same offsets,
// same call sequence as on N GPUs; only the transport differs.
//
// ASYNCHRONOUS, STREAM-ORDERED (round 4; the round-3 version synchronised the caller's stream inside every call, so a missing
// event between a producer kernel and an exchange -- or two operations of one communicator on two streams -- could not fail a
// test).  Like RCCL, every operation is ENQUEUED on the stream it is given and returns at once:
//     1. device -> pinned outbox copies of the sends            (hipMemcpyAsync on the stream)
//     2. a host function on the stream (hipLaunchHostFunc):     publish my message list, barrier over the ranks, copy the peers'
//                                                               messages for me (or the reduced sum) into my pinned inbox, barrier
//     3. pinned inbox -> device copies of the receives          (hipMemcpyAsync on the stream)
// so an operation sees exactly what stream order lets it see.  State per communicator is ONE outbox / inbox / message table:
// operations of one communicator in flight on two streams overwrite each other's staging and fail the matching check below --
// the single-stream rule of csrc/comm.hip is enforced by the stand-in, not assumed.  Operations must reach the host function in
// the same order on every rank of the communicator (NCCL's rule).  HX_FAKE_RCCL_SYNC=1 restores the synchronous transport.
//
// Transport: one POSIX shared-memory segment per communicator (a sense-reversing barrier, per-rank message tables and outboxes,
// registered with the HIP runtime as pinned memory).  Messages between a pair of ranks match in issue order, as in NCCL.
// Reductions add the ranks' contributions in rank order (deterministic).  Not a performance path and not part of the product.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct fakeComm* ncclComm_t;
}

namespace {
const size_t OUTBOX = 96u << 20;          // bytes per rank
const int MAX_MSG = 64;
const int MAX_RANKS = 16;
struct Msg { int peer; size_t off, bytes; };
struct Header {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<int> failed;                // any rank saw a mismatch: every later operation reports it
  int kind[MAX_RANKS];                    // what the rank's pending operation is (0 group of sends/receives, 1 all-reduce): checked
  size_t count[MAX_RANKS];
  int n_msg[MAX_RANKS];
  Msg msg[MAX_RANKS][MAX_MSG];
};
size_t dtype_size(ncclDataType_t t) {
  switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: return 2; case ncclInt32: case ncclUint32: case ncclFloat32: return 4; default: return 8; }
}
struct Op { bool send; void* ptr; size_t bytes; int peer; hipStream_t s; };
thread_local bool g_in_group = false;
thread_local std::vector<Op> g_ops;
thread_local fakeComm* g_group_comm = nullptr;
bool sync_mode() { static const bool on = getenv("HX_FAKE_RCCL_SYNC") && atoi(getenv("HX_FAKE_RCCL_SYNC")) != 0; return on; }
}

struct fakeComm {
  int rank, world;
  Header* hdr;
  char* box;          // world outboxes behind the header (shared, pinned)
  char* inbox;        // this rank's landing area (process-local, pinned)
  size_t map_bytes;
  bool registered = false;
  char name[64];
  // Sense-reversing barrier WITH A DEADLINE (HX_FAKE_RCCL_TIMEOUT_S, default 60 s): it runs inside a host function on the
  // stream, so a rank that raised or exited before enqueueing the matching operation would otherwise leave the other ranks'
  // streams -- and every hipDeviceSynchronize / pool wait behind them -- blocked for good.  On expiry the communicator is
  // marked failed (every later operation reports ncclInternalError), the rank says what it was waiting in, and the stream drains.
  bool barrier(const char* what = "operation") {
    if (hdr->failed.load()) return false;
    const int gen = hdr->generation.load();
    if (hdr->arrived.fetch_add(1) + 1 == world) { hdr->arrived.store(0); hdr->generation.fetch_add(1); return true; }
    static const double limit = getenv("HX_FAKE_RCCL_TIMEOUT_S") ? atof(getenv("HX_FAKE_RCCL_TIMEOUT_S")) : 60.0;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (hdr->generation.load() == gen) {
      sched_yield();
      if ((++spins & 1023u) == 0) {
        if (hdr->failed.load()) return false;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
          fprintf(stderr, "[fake_rccl] rank %d: barrier of %s timed out after %.0f s (%d of %d ranks arrived): a peer never enqueued "
                          "the matching operation\n", rank, what, limit, hdr->arrived.load(), world);
          hdr->failed.store(1);
          return false;
        }
      }
    }
    return true;
  }
  char* outbox(int r) { return box + (size_t)r * OUTBOX; }
};

// ---- the host-side half of an operation: runs inside the stream (host function) or inline (synchronous mode)
struct Recv { int peer; size_t bytes, in_off; };
struct Job {
  fakeComm* c;
  int kind;                               // 0: group, 1: all-reduce
  std::vector<Msg> sends;                 // my messages (offsets into my outbox)
  std::vector<Recv> recvs;                // what I expect, in issue order; in_off = where it lands in my inbox
  size_t count = 0; int dtype = 0, op = 0; // all-reduce
};

static void fail(fakeComm* c, const char* what) {
  fprintf(stderr, "[fake_rccl] rank %d: %s\n", c->rank, what);
  c->hdr->failed.store(1);
}

static void run_job(Job* j) {
  fakeComm* c = j->c;
  Header* H = c->hdr;
  H->kind[c->rank] = j->kind;
  H->count[c->rank] = j->kind ? j->count * (size_t)(j->dtype ? 8 : 4) : 0;
  H->n_msg[c->rank] = (int)j->sends.size();
  for (size_t q = 0; q < j->sends.size(); ++q) H->msg[c->rank][q] = j->sends[q];
  if (!c->barrier(j->kind ? "an all-reduce" : "a send/receive group")) return;     // failed or timed out: no copies, the stream drains
  // every rank must be in the same kind of operation (a group meeting an all-reduce = operations issued in different orders,
  // or two operations of this communicator racing on two streams)
  for (int r = 0; r < c->world; ++r)
    if (H->kind[r] != j->kind || (j->kind && H->count[r] != H->count[c->rank])) fail(c, "ranks are in different operations of the communicator");
  if (j->kind == 0) {
    int taken[MAX_RANKS] = {0};
    for (const Recv& rv : j->recvs) {
      const int p = rv.peer;
      int seen = 0;
      const Msg* hit = nullptr;
      for (int q = 0; q < H->n_msg[p]; ++q) {
        const Msg& m = H->msg[p][q];
        if (m.peer != c->rank) continue;
        if (seen++ == taken[p]) { hit = &m; break; }
      }
      if (!hit || hit->bytes != rv.bytes) { fail(c, "a receive has no matching send of its size"); continue; }
      ++taken[p];
      memcpy(c->inbox + rv.in_off, c->outbox(p) + hit->off, rv.bytes);
    }
  } else {
    const size_t n = j->count;
    if (j->dtype) {
      double* a = (double*)c->inbox;
      memcpy(a, c->outbox(0), n * 8);
      for (int r = 1; r < c->world; ++r) { const double* b = (const double*)c->outbox(r);
        for (size_t i = 0; i < n; ++i) a[i] = j->op == 0 ? a[i] + b[i] : (a[i] > b[i] ? a[i] : b[i]); }
    } else {
      float* a = (float*)c->inbox;
      memcpy(a, c->outbox(0), n * 4);
      for (int r = 1; r < c->world; ++r) { const float* b = (const float*)c->outbox(r);
        for (size_t i = 0; i < n; ++i) a[i] = j->op == 0 ? a[i] + b[i] : (a[i] > b[i] ? a[i] : b[i]); }
    }
  }
  (void)c->barrier(j->kind ? "an all-reduce (release)" : "a send/receive group (release)");       // outboxes and tables may be reused
}
static void host_fn(void* p) { Job* j = (Job*)p; run_job(j); delete j; }

// enqueue one operation on stream s: sends (device -> outbox), host half, receives (inbox -> device)
static ncclResult_t enqueue(fakeComm* c, Job* j, const std::vector<Op>& sends, const std::vector<Op>& recvs, hipStream_t s) {
  if (c->hdr->failed.load()) { delete j; return ncclInternalError; }
  const bool sync = sync_mode() || !c->registered;
  if (sync && hipStreamSynchronize(s) != hipSuccess) { delete j; return ncclUnhandledCudaError; }
  for (size_t q = 0; q < sends.size(); ++q) {
    char* dst = c->outbox(c->rank) + j->sends[q].off;
    const hipError_t e = sync ? hipMemcpy(dst, sends[q].ptr, sends[q].bytes, hipMemcpyDeviceToHost)
                              : hipMemcpyAsync(dst, sends[q].ptr, sends[q].bytes, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) { delete j; return ncclUnhandledCudaError; }
  }
  std::vector<Recv> landing = j->recvs;
  const int kind = j->kind;
  const size_t red_bytes = j->count * (size_t)(j->dtype ? 8 : 4);
  if (sync) { run_job(j); delete j; }
  else if (hipLaunchHostFunc(s, host_fn, j) != hipSuccess) { delete j; return ncclUnhandledCudaError; }
  for (size_t q = 0; q < recvs.size(); ++q) {
    const char* src = c->inbox + (kind ? 0 : landing[q].in_off);
    const size_t bytes = kind ? red_bytes : landing[q].bytes;
    const hipError_t e = sync ? hipMemcpy(recvs[q].ptr, src, bytes, hipMemcpyHostToDevice)
                              : hipMemcpyAsync(recvs[q].ptr, src, bytes, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return ncclUnhandledCudaError;
  }
  return c->hdr->failed.load() ? ncclInternalError : ncclSuccess;
}

static ncclResult_t run_group(fakeComm* c, std::vector<Op>& ops) {
  Job* j = new Job();
  j->c = c; j->kind = 0;
  std::vector<Op> sends, recvs;
  size_t off = 0, in_off = 0;
  hipStream_t s = ops.empty() ? nullptr : ops[0].s;
  for (Op& o : ops) {
    if (o.s != s) { fprintf(stderr, "[fake_rccl] a group spans two streams\n"); delete j; return ncclInvalidArgument; }
    if (o.send) {
      if (off + o.bytes > OUTBOX || (int)j->sends.size() >= MAX_MSG) { fprintf(stderr, "[fake_rccl] outbox overflow\n"); delete j; return ncclInternalError; }
      j->sends.push_back(Msg{o.peer, off, o.bytes});
      sends.push_back(o);
      off += (o.bytes + 63) & ~(size_t)63;
    } else {
      if (in_off + o.bytes > OUTBOX) { fprintf(stderr, "[fake_rccl] inbox overflow\n"); delete j; return ncclInternalError; }
      j->recvs.push_back(Recv{o.peer, o.bytes, in_off});
      recvs.push_back(o);
      in_off += (o.bytes + 63) & ~(size_t)63;
    }
  }
  return enqueue(c, j, sends, recvs, s);
}

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/hxfake_%d_%ld", (int)getpid(), (long)random());
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAX_RANKS) return ncclInvalidArgument;
  fakeComm* c = new fakeComm();
  c->rank = rank; c->world = nranks;
  snprintf(c->name, sizeof(c->name), "%s", id.internal);
  c->map_bytes = sizeof(Header) + (size_t)nranks * OUTBOX;
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) return ncclSystemError;
  void* m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return ncclSystemError;
  c->hdr = (Header*)m;            // a fresh segment is zero-filled: arrived = generation = 0
  c->box = (char*)m + sizeof(Header);
  // pinned staging: asynchronous copies read / write host memory at their point in the stream, not at the call
  if (!sync_mode()) {
    // (touch the pages first: the segment is sparse until written)
    memset(c->outbox(rank), 0, OUTBOX);
    c->registered = hipHostRegister(m, c->map_bytes, hipHostRegisterPortable) == hipSuccess &&
                    hipHostMalloc((void**)&c->inbox, OUTBOX, hipHostMallocDefault) == hipSuccess;
    if (!c->registered) { (void)hipGetLastError(); fprintf(stderr, "[fake_rccl] rank %d: pinned staging unavailable, synchronous transport\n", rank); }
  }
  if (!c->registered) c->inbox = (char*)malloc(OUTBOX);
  const bool met = c->barrier("ncclCommInitRank");
  if (rank == 0) shm_unlink(c->name);      // every rank has it mapped (or gave up): the name can go
  if (!met || !c->barrier("ncclCommInitRank (release)")) return ncclSystemError;
  *out = c;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (c) {
    (void)hipDeviceSynchronize();          // no host function of this communicator may still be queued
    if (c->registered) { (void)hipHostUnregister((void*)c->hdr); (void)hipHostFree(c->inbox); }
    else free(c->inbox);
    munmap((void*)c->hdr, c->map_bytes);
    delete c;
  }
  return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) { case ncclSuccess: return "success"; case ncclUnhandledCudaError: return "HIP error"; case ncclSystemError: return "system error";
               case ncclInvalidArgument: return "invalid argument"; default: return "internal error (operations of a communicator mismatched across ranks or streams)"; }
}
ncclResult_t ncclGroupStart() { g_in_group = true; g_ops.clear(); g_group_comm = nullptr; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  g_in_group = false;
  if (!g_group_comm) return ncclSuccess;     // (an empty group has no communicator to synchronise with)
  ncclResult_t rc = run_group(g_group_comm, g_ops);
  g_ops.clear();
  return rc;
}
static ncclResult_t p2p(bool send, void* ptr, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
  if (!c || peer < 0 || peer >= c->world) return ncclInvalidArgument;
  Op o{send, ptr, count * dtype_size(t), peer, s};
  if (g_in_group) { g_ops.push_back(o); g_group_comm = c; return ncclSuccess; }
  std::vector<Op> one{o};
  return run_group(c, one);
}
ncclResult_t ncclSend(const void* p, size_t n, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) { return p2p(true, (void*)p, n, t, peer, c, s); }
ncclResult_t ncclRecv(void* p, size_t n, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) { return p2p(false, p, n, t, peer, c, s); }
ncclResult_t ncclAllReduce(const void* in, void* out, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
  if (!c || (t != ncclFloat32 && t != ncclFloat64) || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;
  const size_t bytes = count * dtype_size(t);
  if (bytes > OUTBOX) return ncclInternalError;
  Job* j = new Job();
  j->c = c; j->kind = 1; j->count = count; j->dtype = t == ncclFloat64; j->op = op == ncclMax;
  j->sends.push_back(Msg{-1, 0, bytes});
  std::vector<Op> sends{Op{true, (void*)in, bytes, -1, s}}, recvs{Op{false, out, bytes, -1, s}};
  return enqueue(c, j, sends, recvs, s);
}
}
