// Test double for librccl (tests only; selected with HX_RCCL_LIB): the nine entry points libhx binds (csrc/comm.hip), for
// ranks that SHARE one GPU -- RCCL itself refuses two ranks on one device, and the GPU boxes of this project have one.
// It lets the library-issued exchange path (hx_comm_set_plan / hx_comm_exchange / hx_comm_allreduce, the callbacks of
// hx_cocg_sharded / hx_gmres_sharded / the sharded V-cycle) run end to end with N processes: same plans, same offsets,
// same call sequence as on N GPUs; only the transport differs.
//
// Transport: one POSIX shared-memory segment per communicator: a sense-reversing barrier + one outbox per rank.  Every
// operation synchronises the caller's stream, stages through the host and is collective over ALL ranks of the communicator
// (a rank without neighbours still calls hx_comm_exchange: the library issues an empty group then).  Messages between a
// pair of ranks match in issue order, as in NCCL.  Reductions add the ranks' contributions in rank order (deterministic).
// Not a performance path and not part of the product.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
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
struct Msg { int peer; size_t off, bytes; };
struct Header {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  int n_msg[16];
  Msg msg[16][MAX_MSG];
};
size_t dtype_size(ncclDataType_t t) {
  switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: return 2; case ncclInt32: case ncclUint32: case ncclFloat32: return 4; default: return 8; }
}
struct Op { bool send; void* ptr; size_t bytes; int peer; hipStream_t s; };
thread_local bool g_in_group = false;
thread_local std::vector<Op> g_ops;
thread_local fakeComm* g_group_comm = nullptr;
}

struct fakeComm {
  int rank, world;
  Header* hdr;
  char* box;          // world outboxes behind the header
  size_t map_bytes;
  char name[64];
  void barrier() {
    const int gen = hdr->generation.load();
    if (hdr->arrived.fetch_add(1) + 1 == world) { hdr->arrived.store(0); hdr->generation.fetch_add(1); }
    else while (hdr->generation.load() == gen) sched_yield();
  }
  char* outbox(int r) { return box + (size_t)r * OUTBOX; }
};

static ncclResult_t run_group(fakeComm* c, std::vector<Op>& ops) {
  // 1) my sends -> my outbox (after everything queued on the streams so far)
  size_t off = 0;
  int nm = 0;
  for (Op& o : ops) {
    if (hipStreamSynchronize(o.s) != hipSuccess) return ncclUnhandledCudaError;
    if (!o.send) continue;
    if (off + o.bytes > OUTBOX || nm >= MAX_MSG) { fprintf(stderr, "[fake_rccl] outbox overflow\n"); return ncclInternalError; }
    if (hipMemcpy(c->outbox(c->rank) + off, o.ptr, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    c->hdr->msg[c->rank][nm++] = Msg{o.peer, off, o.bytes};
    off += o.bytes;
  }
  c->hdr->n_msg[c->rank] = nm;
  c->barrier();
  // 2) my receives: the k-th message a peer addressed to me, in issue order
  int taken[16] = {0};
  ncclResult_t rc = ncclSuccess;
  for (Op& o : ops) {
    if (o.send) continue;
    const int p = o.peer;
    int seen = 0;
    const Msg* hit = nullptr;
    for (int q = 0; q < c->hdr->n_msg[p]; ++q) {
      const Msg& m = c->hdr->msg[p][q];
      if (m.peer != c->rank) continue;
      if (seen++ == taken[p]) { hit = &m; break; }
    }
    if (!hit || hit->bytes != o.bytes) {
      fprintf(stderr, "[fake_rccl] rank %d: receive of %zu bytes from %d has no matching send (%zu)\n", c->rank, o.bytes, p, hit ? hit->bytes : (size_t)0);
      rc = ncclInvalidArgument;
      continue;
    }
    ++taken[p];
    if (hipMemcpy(o.ptr, c->outbox(p) + hit->off, o.bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
  }
  c->barrier();       // outboxes may be reused
  return rc;
}

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/hxfake_%d_%ld", (int)getpid(), (long)random());
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > 16) return ncclInvalidArgument;
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
  c->barrier();
  if (rank == 0) shm_unlink(c->name);      // every rank has it mapped: the name can go
  c->barrier();
  *out = c;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (c) { munmap((void*)c->hdr, c->map_bytes); delete c; }
  return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) { case ncclSuccess: return "success"; case ncclUnhandledCudaError: return "HIP error"; case ncclSystemError: return "system error";
               case ncclInvalidArgument: return "invalid argument"; default: return "internal error"; }
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
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->outbox(c->rank), in, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  c->barrier();
  std::vector<char> acc(bytes);
  memcpy(acc.data(), c->outbox(0), bytes);
  for (int r = 1; r < c->world; ++r) {
    if (t == ncclFloat64) { double* a = (double*)acc.data(); const double* b = (const double*)c->outbox(r);
      for (size_t i = 0; i < count; ++i) a[i] = op == ncclSum ? a[i] + b[i] : (a[i] > b[i] ? a[i] : b[i]); }
    else { float* a = (float*)acc.data(); const float* b = (const float*)c->outbox(r);
      for (size_t i = 0; i < count; ++i) a[i] = op == ncclSum ? a[i] + b[i] : (a[i] > b[i] ? a[i] : b[i]); }
  }
  ncclResult_t rc = hipMemcpy(out, acc.data(), bytes, hipMemcpyHostToDevice) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
  c->barrier();
  return rc;
}
}
