// RCCL exchanges issued from the library, on the stream the kernels run on (one process per GPU, xGMI underneath).
//
// What the reference gets from PETSc's distributed Vec / Mat under `mpirun -np N` (ghost updates of MatMult:
// VecScatter, reached from eigensolvers.py:62,113 through EPSSolve / PEPSolve; inner products: MPI_Allreduce inside
// BVDot / VecNorm) is done here by two primitives on an `hx_comm`:
//   * exchange(plan): every rank sends listed entries of its OWNED block to the ranks that hold them as ghosts and
//     receives its own ghost entries -- point-to-point ncclSend / ncclRecv grouped into one call (theta-slabs: two ring
//     neighbours; xGMI links are point-to-point, so this is the natural transport, not a ring collective);
//   * allreduce: sum / max of a few scalars (Krylov inner products: <= 100 numbers, latency-bound) or of one coarse
//     right-hand side.
// Both are enqueued on the caller's stream: no host synchronisation, no Python frame per exchange.  EVERY call into the
// communicator -- blocking halos, split halos, all-reduces -- is issued on that ONE stream, in program order, hence in the same
// order on every rank: RCCL gives no ordering guarantee between operations of one communicator in flight on different
// streams.  The split halo overlaps the other way round: the exchange stays on the caller's stream, the INTERIOR slices of the
// product it feeds run on a side stream of the communicator (hx_comm_halo_begin hands it out), two events order the two.  hx_comm_halo /
// hx_comm_allreduce have the hx_halo_fn / hx_allreduce_fn signatures and are handed to hx_amg_set_sharded with the
// communicator as context, so hx_cocg_sharded / hx_gmres_sharded / the V-cycle run a whole solve inside the library.
//
// librccl is bound at RUN TIME (dlopen of the library torch already loaded, else librccl.so.1): libhx.so itself has no
// link dependency on it, so the single-GPU path and the CPU-side symbol tests never need it.  HX_RCCL_LIB overrides
// the library path (tests/fake_rccl: a host-staged stand-in with the same entry points for two ranks sharing one GPU,
// which RCCL itself refuses).
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include <rccl/rccl.h>

#include "hx_common.h"

namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char name[256] = {0};
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.lib) return HX_OK;
  const char* over = getenv("HX_RCCL_LIB");
  void* h = nullptr;
  if (over && *over) {
    h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
    if (!h) return hx_fail(HX_ERR_COMM, "hx_comm: dlopen(HX_RCCL_LIB)", dlerror());
    snprintf(g_rccl.name, sizeof(g_rccl.name), "%s", over);
  } else {
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* nm : names) {            // the copy the process already holds (torch's), if any
      h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
      if (h) { snprintf(g_rccl.name, sizeof(g_rccl.name), "%s (already loaded)", nm); break; }
    }
    if (!h) for (const char* nm : {"librccl.so.1", "librccl.so"}) {
      h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (h) { snprintf(g_rccl.name, sizeof(g_rccl.name), "%s", nm); break; }
    }
    if (!h) return hx_fail(HX_ERR_COMM, "hx_comm", "librccl.so not found (set HX_RCCL_LIB)");
  }
#define BIND(field, sym)                                                            \
  g_rccl.field = (decltype(g_rccl.field))dlsym(h, sym);                             \
  if (!g_rccl.field) return hx_fail(HX_ERR_COMM, "hx_comm: missing RCCL symbol", sym)
  BIND(GetUniqueId, "ncclGetUniqueId");
  BIND(CommInitRank, "ncclCommInitRank");
  BIND(CommDestroy, "ncclCommDestroy");
  BIND(AllReduce, "ncclAllReduce");
  BIND(Send, "ncclSend");
  BIND(Recv, "ncclRecv");
  BIND(GroupStart, "ncclGroupStart");
  BIND(GroupEnd, "ncclGroupEnd");
  BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
  g_rccl.lib = h;
  return HX_OK;
}

#define HX_NCCL(call)                                                                              \
  do {                                                                                             \
    ncclResult_t r_ = (call);                                                                      \
    if (r_ != ncclSuccess) return hx_fail(HX_ERR_COMM, #call, g_rccl.GetErrorString(r_));          \
  } while (0)

struct Plan {
  bool set = false;
  std::vector<int> send_peer, recv_peer;
  std::vector<int64_t> send_cnt;           // entries per send peer
  std::vector<int64_t> send_lo;            // >= 0: the list is the contiguous owned range [lo, lo + cnt): sent in place
  std::vector<int64_t> send_stage;         // else: offset of the gathered list in the staging buffer
  std::vector<int64_t> recv_off, recv_cnt; // per recv peer: destination element range
  int32_t* d_idx = nullptr;                // concatenated owned indices of the gathered lists (device)
  int64_t n_gather = 0;
  void* stage = nullptr;                   // n_gather * 16 bytes
  // a peer whose entries fill SEVERAL separate ranges (the layers of a deep-halo plan) still gets ONE message: it lands in
  // `rstage` and k_comm_scatter moves entry i to dst[d_ridx[i]]
  std::vector<int64_t> recv_stage;         // >= 0: offset in rstage (else the message lands in dst at recv_off)
  int32_t* d_ridx = nullptr;
  int64_t n_scatter = 0;
  void* rstage = nullptr;
};
}  // namespace

struct hx_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  std::vector<Plan> plans;
  int64_t n_exchange = 0, n_allreduce = 0, n_overlapped = 0;
  // split halo (hx_comm_halo_begin / _end): the exchange is enqueued on the caller's stream like every other operation of the
  // communicator; the interior slices of the product run meanwhile on `istream`, ordered against the caller's stream by two
  // events.  `pending`: a begin without its end (one exchange may be split at a time).
  hipStream_t istream = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  bool pending = false;
  hipStream_t op_stream = nullptr;       // the stream the communicator's operations were last issued on (single-stream rule)
  int64_t n_stream_switches = 0;         // times an operation arrived on another stream than the previous one
};

// Single-stream rule: operations of one communicator issued on different streams are ordered against each other here (the new
// stream waits for everything the previous one holds), and counted -- the product path never switches (tests assert 0).
static int same_stream(hx_comm* c, hx_stream s) {
  hipStream_t st = hx_s(s);
  if (c->op_stream == st) return HX_OK;
  if (c->op_stream != nullptr || c->n_exchange + c->n_allreduce > 0) {
    hipEvent_t ev;
    HX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HX_HIP(hipEventRecord(ev, c->op_stream));
    HX_HIP(hipStreamWaitEvent(st, ev, 0));
    HX_HIP(hipEventDestroy(ev));
    ++c->n_stream_switches;
  }
  c->op_stream = st;
  return HX_OK;
}

// ---- gather of listed entries (4-, 8- or 16-byte elements) into the contiguous send buffer
template <typename T>
__global__ void k_comm_gather(int64_t n, const int32_t* __restrict__ idx, const T* __restrict__ src, T* __restrict__ dst) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src[idx[i]];
}

template <typename T>
__global__ void k_comm_scatter(int64_t n, const int32_t* __restrict__ idx, const T* __restrict__ src, T* __restrict__ dst) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[idx[i]] = src[i];
}

extern "C" int hx_comm_unique_id(void* out, int nbytes) {
  HX_REQUIRE(out && nbytes >= (int)sizeof(ncclUniqueId), "need a buffer of HX_COMM_ID_BYTES bytes");
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  HX_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(out, &id, sizeof(id));
  return HX_OK;
}

extern "C" int hx_comm_create(hx_comm** out, int rank, int world, const void* id, int nbytes) {
  HX_REQUIRE(out && id && world >= 1 && rank >= 0 && rank < world && nbytes >= (int)sizeof(ncclUniqueId), "bad argument");
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  hx_comm* c = new hx_comm();
  c->rank = rank; c->world = world;
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
  if (r != ncclSuccess) { delete c; return hx_fail(HX_ERR_COMM, "ncclCommInitRank", g_rccl.GetErrorString(r)); }
  *out = c;
  return HX_OK;
}

extern "C" int hx_comm_destroy(hx_comm* c) {
  if (!c) return HX_OK;
  // nothing of this communicator may still be in flight when its staging buffers go back to the pool
  if (c->istream) (void)hipStreamSynchronize(c->istream);
  if (c->op_stream || c->n_exchange + c->n_allreduce > 0) (void)hipStreamSynchronize(c->op_stream);
  for (Plan& p : c->plans) { (void)hx_free(p.d_idx); (void)hx_free(p.stage); (void)hx_free(p.d_ridx); (void)hx_free(p.rstage); }
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->istream) { hx_pool_register_stream(c->istream, false); (void)hipStreamDestroy(c->istream); }
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
  return HX_OK;
}

extern "C" const char* hx_comm_library(void) { return g_rccl.name; }

extern "C" int hx_comm_counters(const hx_comm* c, int64_t* n_exchange, int64_t* n_allreduce, int reset) {
  HX_REQUIRE(c, "null communicator");
  if (n_exchange) *n_exchange = c->n_exchange;
  if (n_allreduce) *n_allreduce = c->n_allreduce;
  if (reset) { hx_comm* m = const_cast<hx_comm*>(c); m->n_exchange = 0; m->n_allreduce = 0; m->n_overlapped = 0; }
  return HX_OK;
}
// number of halo exchanges that ran split (hx_comm_halo_begin / _end: overlapped with the interior slices of their product)
extern "C" int64_t hx_comm_overlapped(const hx_comm* c) { return c ? c->n_overlapped : 0; }
// times an operation of the communicator arrived on another stream than the one before it (0 on the product path)
extern "C" int64_t hx_comm_stream_switches(const hx_comm* c) { return c ? c->n_stream_switches : 0; }

// Exchange plan `plan` (>= 0; halo plans use the hierarchy level as their number).  HOST arrays:
//   send_peer[ns], send_ptr[ns + 1] into send_idx (owned element indices, ascending per peer);
//   recv_peer[nr], recv_off[nr], recv_cnt[nr]: element range of the destination buffer that peer's entries fill.
// A send list that is one contiguous run is sent in place; the others are gathered into a staging buffer first.
extern "C" int hx_comm_set_plan(hx_comm* c, int plan, int ns, const int32_t* send_peer, const int64_t* send_ptr,
                                const int32_t* send_idx, int nr, const int32_t* recv_peer, const int64_t* recv_off,
                                const int64_t* recv_cnt) {
  HX_REQUIRE(c && plan >= 0 && plan < 64 && ns >= 0 && nr >= 0, "bad argument");
  HX_REQUIRE(ns == 0 || (send_peer && send_ptr && send_idx), "missing send lists");
  HX_REQUIRE(nr == 0 || (recv_peer && recv_off && recv_cnt), "missing receive ranges");
  if ((int)c->plans.size() <= plan) c->plans.resize(plan + 1);
  Plan& P = c->plans[plan];
  (void)hx_free(P.d_idx); (void)hx_free(P.stage); (void)hx_free(P.d_ridx); (void)hx_free(P.rstage);
  P = Plan();
  // ONE message per peer and direction: lists / ranges of a peer that is named several times (per peer one entry per layer of a
  // deep-halo plan, in layer order on both sides) are concatenated in the order given -- nothing rests on how the
  // communication library matches several messages between one pair of ranks
  std::vector<int32_t> gathered, scattered;
  {
    std::vector<int> peers;
    for (int q = 0; q < ns; ++q) {
      HX_REQUIRE(send_peer[q] >= 0 && send_peer[q] < c->world && send_ptr[q + 1] >= send_ptr[q], "bad send list");
      if (std::find(peers.begin(), peers.end(), (int)send_peer[q]) == peers.end()) peers.push_back(send_peer[q]);
    }
    for (int peer : peers) {
      std::vector<int32_t> list;
      for (int q = 0; q < ns; ++q)
        if (send_peer[q] == peer) list.insert(list.end(), send_idx + send_ptr[q], send_idx + send_ptr[q + 1]);
      bool run = !list.empty();
      for (size_t i = 1; i < list.size() && run; ++i) run = list[i] == list[i - 1] + 1;
      P.send_peer.push_back(peer);
      P.send_cnt.push_back((int64_t)list.size());
      P.send_lo.push_back(run ? (int64_t)list[0] : -1);
      P.send_stage.push_back((int64_t)gathered.size());
      if (!run) gathered.insert(gathered.end(), list.begin(), list.end());
    }
    peers.clear();
    for (int q = 0; q < nr; ++q) {
      HX_REQUIRE(recv_peer[q] >= 0 && recv_peer[q] < c->world && recv_off[q] >= 0 && recv_cnt[q] >= 0, "bad receive range");
      HX_REQUIRE(recv_off[q] + recv_cnt[q] < ((int64_t)1 << 31), "receive range beyond int32 positions");
      if (std::find(peers.begin(), peers.end(), (int)recv_peer[q]) == peers.end()) peers.push_back(recv_peer[q]);
    }
    for (int peer : peers) {
      int64_t first = -1, next = -1, total = 0;
      bool one_range = true;
      for (int q = 0; q < nr; ++q)
        if (recv_peer[q] == peer && recv_cnt[q] > 0) {
          if (first < 0) first = recv_off[q]; else if (recv_off[q] != next) one_range = false;
          next = recv_off[q] + recv_cnt[q];
          total += recv_cnt[q];
        }
      P.recv_peer.push_back(peer);
      P.recv_cnt.push_back(total);
      P.recv_off.push_back(first < 0 ? 0 : first);
      if (one_range) { P.recv_stage.push_back(-1); continue; }
      P.recv_stage.push_back((int64_t)scattered.size());
      for (int q = 0; q < nr; ++q)
        if (recv_peer[q] == peer)
          for (int64_t i = 0; i < recv_cnt[q]; ++i) scattered.push_back((int32_t)(recv_off[q] + i));
    }
  }
  P.n_scatter = (int64_t)scattered.size();
  if (P.n_scatter > 0) {
    HX_HIP(hx_malloc((void**)&P.d_ridx, sizeof(int32_t) * (size_t)P.n_scatter));
    HX_HIP(hipMemcpy(P.d_ridx, scattered.data(), sizeof(int32_t) * (size_t)P.n_scatter, hipMemcpyHostToDevice));
    HX_HIP(hx_malloc(&P.rstage, 16 * (size_t)P.n_scatter));
  }
  P.n_gather = (int64_t)gathered.size();
  if (P.n_gather > 0) {
    HX_HIP(hx_malloc((void**)&P.d_idx, sizeof(int32_t) * (size_t)P.n_gather));
    // (a pooled block may still be read by kernels of a non-blocking stream: hx_malloc orders the block against its last
    // users; the copy itself is synchronous with respect to the host)
    HX_HIP(hipMemcpy(P.d_idx, gathered.data(), sizeof(int32_t) * (size_t)P.n_gather, hipMemcpyHostToDevice));
    HX_HIP(hx_malloc(&P.stage, 16 * (size_t)P.n_gather));
  }
  P.set = true;
  return HX_OK;
}

// dst[recv ranges] <- the peers' listed entries of their src; src[send lists] -> the peers.  src and dst may be the same
// buffer (halo: the owned block of x_ext feeds the peers' ghost blocks).  elem_bytes in {4, 8, 16}.  Stream-ordered.
extern "C" int hx_comm_exchange(hx_comm* c, int plan, const void* src, void* dst, int elem_bytes, hx_stream s) {
  HX_REQUIRE(c && src && dst && plan >= 0 && plan < (int)c->plans.size() && c->plans[plan].set, "unknown exchange plan");
  HX_REQUIRE(elem_bytes == 4 || elem_bytes == 8 || elem_bytes == 16, "element size must be 4, 8 or 16 bytes");
  Plan& P = c->plans[plan];
  const int ns = (int)P.send_peer.size(), nr = (int)P.recv_peer.size();
  { const int rs = same_stream(c, s); if (rs) return rs; }
  ++c->n_exchange;
  if (ns == 0 && nr == 0) return HX_OK;
  const int64_t* stage_off = P.send_stage.data();
  const int64_t* cnt = P.send_cnt.data();
  if (P.n_gather > 0) HX_DEBUG_BOUNDS("hx_comm_exchange (send lists of the plan)", P.d_idx, P.n_gather, nullptr, 0, src, elem_bytes, hx_s(s));
  if (P.n_scatter > 0) HX_DEBUG_BOUNDS("hx_comm_exchange (receive ranges of the plan)", P.d_ridx, P.n_scatter, nullptr, 0, dst, elem_bytes, hx_s(s));
  if (P.n_gather > 0) {
    const int g = hx_grid(P.n_gather, 256);
    if (elem_bytes == 16) k_comm_gather<double2><<<g, 256, 0, hx_s(s)>>>(P.n_gather, P.d_idx, (const double2*)src, (double2*)P.stage);
    else if (elem_bytes == 8) k_comm_gather<float2><<<g, 256, 0, hx_s(s)>>>(P.n_gather, P.d_idx, (const float2*)src, (float2*)P.stage);
    else k_comm_gather<float><<<g, 256, 0, hx_s(s)>>>(P.n_gather, P.d_idx, (const float*)src, (float*)P.stage);
    HX_LAUNCH_CHECK("k_comm_gather");
  }
  HX_NCCL(g_rccl.GroupStart());
  ncclResult_t bad = ncclSuccess;
  for (int q = 0; q < ns && bad == ncclSuccess; ++q) {
    const char* p = P.send_lo[q] >= 0 ? (const char*)src + (size_t)P.send_lo[q] * elem_bytes
                                      : (const char*)P.stage + (size_t)stage_off[q] * elem_bytes;
    bad = g_rccl.Send(p, (size_t)cnt[q] * elem_bytes, ncclChar, P.send_peer[q], c->comm, hx_s(s));
  }
  for (int q = 0; q < nr && bad == ncclSuccess; ++q) {
    char* p = P.recv_stage[q] >= 0 ? (char*)P.rstage + (size_t)P.recv_stage[q] * elem_bytes : (char*)dst + (size_t)P.recv_off[q] * elem_bytes;
    bad = g_rccl.Recv(p, (size_t)P.recv_cnt[q] * elem_bytes, ncclChar, P.recv_peer[q], c->comm, hx_s(s));
  }
  const ncclResult_t ge = g_rccl.GroupEnd();
  if (bad != ncclSuccess) return hx_fail(HX_ERR_COMM, "ncclSend/ncclRecv", g_rccl.GetErrorString(bad));
  HX_NCCL(ge);
  if (P.n_scatter > 0) {        // messages that fill several separate ranges: from the staging buffer to their places
    const int g = hx_grid(P.n_scatter, 256);
    if (elem_bytes == 16) k_comm_scatter<double2><<<g, 256, 0, hx_s(s)>>>(P.n_scatter, P.d_ridx, (const double2*)P.rstage, (double2*)dst);
    else if (elem_bytes == 8) k_comm_scatter<float2><<<g, 256, 0, hx_s(s)>>>(P.n_scatter, P.d_ridx, (const float2*)P.rstage, (float2*)dst);
    else k_comm_scatter<float><<<g, 256, 0, hx_s(s)>>>(P.n_scatter, P.d_ridx, (const float*)P.rstage, (float*)dst);
    HX_LAUNCH_CHECK("k_comm_scatter");
  }
  return HX_OK;
}

// hx_halo_fn: ghost block of a level's extended vector (plan number = level)
extern "C" int hx_comm_halo(void* ctx, int level, void* x_ext, int elem_bytes, hx_stream s) {
  return hx_comm_exchange((hx_comm*)ctx, level, x_ext, x_ext, elem_bytes, s);
}

// Split form of the halo (hx_halo_begin_fn / hx_halo_end_fn for hx_amg_set_halo_overlap).  _begin enqueues the exchange on `s`
// (after everything enqueued there so far) and hands out the communicator's side stream, made to wait for the same point of
// `s`: the caller launches THERE the kernels that read only the OWNED block of x_ext (the interior slices of a row-sharded
// product) -- they run while the ghost entries travel.  _end makes `s` wait for the side stream; the boundary slices then
// follow on `s`, behind the exchange.  One exchange may be split at a time.
extern "C" int hx_comm_halo_begin(void* ctx, int level, void* x_ext, int elem_bytes, hx_stream s, hx_stream* interior) {
  hx_comm* c = (hx_comm*)ctx;
  HX_REQUIRE(c && x_ext && interior, "bad argument");
  HX_REQUIRE(!c->pending, "hx_comm_halo_begin: the previous split exchange was not ended");
  if (!c->istream) {
    HX_HIP(hipStreamCreateWithFlags(&c->istream, hipStreamNonBlocking));
    hx_pool_register_stream(c->istream, true);
    HX_HIP(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    HX_HIP(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  }
  HX_HIP(hipEventRecord(c->ev_ready, hx_s(s)));
  HX_HIP(hipStreamWaitEvent(c->istream, c->ev_ready, 0));
  const int rc = hx_comm_exchange(c, level, x_ext, x_ext, elem_bytes, s);
  if (rc) return rc;                         // (nothing was launched on the side stream yet: nothing to drain)
  c->pending = true;
  *interior = (hx_stream)c->istream;
  ++c->n_overlapped;
  return HX_OK;
}
extern "C" int hx_comm_halo_end(void* ctx, int level, hx_stream s) {
  hx_comm* c = (hx_comm*)ctx;
  (void)level;
  HX_REQUIRE(c && c->pending, "hx_comm_halo_end: no split exchange pending");
  c->pending = false;
  HX_HIP(hipEventRecord(c->ev_done, c->istream));
  HX_HIP(hipStreamWaitEvent(hx_s(s), c->ev_done, 0));
  return HX_OK;
}

// Every rank's `bytes`-long block into everybody's recv buffer (world x bytes, rank order): one group of point-to-point messages on
// the caller's stream, like an exchange -- the right-hand sides of the replicated direct solve (slepc.ReplicatedLUSolver).
extern "C" int hx_comm_allgather(void* ctx, const void* send, void* recv, int64_t bytes, hx_stream s) {
  hx_comm* c = (hx_comm*)ctx;
  HX_REQUIRE(c && send && recv && bytes >= 0, "bad argument");
  { const int rs = same_stream(c, s); if (rs) return rs; }
  ++c->n_exchange;
  if (bytes == 0) return HX_OK;
  char* out = (char*)recv;
  HX_HIP(hipMemcpyAsync(out + (size_t)c->rank * (size_t)bytes, send, (size_t)bytes, hipMemcpyDeviceToDevice, hx_s(s)));
  if (c->world == 1) return HX_OK;
  HX_NCCL(g_rccl.GroupStart());
  ncclResult_t bad = ncclSuccess;
  for (int p = 0; p < c->world && bad == ncclSuccess; ++p)
    if (p != c->rank) bad = g_rccl.Send(send, (size_t)bytes, ncclChar, p, c->comm, hx_s(s));
  for (int p = 0; p < c->world && bad == ncclSuccess; ++p)
    if (p != c->rank) bad = g_rccl.Recv(out + (size_t)p * (size_t)bytes, (size_t)bytes, ncclChar, p, c->comm, hx_s(s));
  const ncclResult_t ge = g_rccl.GroupEnd();
  if (bad != ncclSuccess) return hx_fail(HX_ERR_COMM, "ncclSend/ncclRecv (all-gather)", g_rccl.GetErrorString(bad));
  HX_NCCL(ge);
  return HX_OK;
}

// hx_allreduce_fn: in-place reduction of `count` reals (dtype 0: float32, 1: float64; op 0: sum, 1: max)
extern "C" int hx_comm_allreduce(void* ctx, void* buf, int64_t count, int dtype, int op, hx_stream s) {
  hx_comm* c = (hx_comm*)ctx;
  HX_REQUIRE(c && buf && count >= 0 && (dtype == 0 || dtype == 1) && (op == 0 || op == 1), "bad argument");
  { const int rs = same_stream(c, s); if (rs) return rs; }
  ++c->n_allreduce;
  if (count == 0 || c->world == 1) return HX_OK;
  HX_NCCL(g_rccl.AllReduce(buf, buf, (size_t)count, dtype ? ncclFloat64 : ncclFloat32, op ? ncclMax : ncclSum, c->comm,
                           hx_s(s)));
  return HX_OK;
}
