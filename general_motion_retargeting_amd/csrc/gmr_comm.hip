// gmr_comm.hip -- the collectives of the multi-GPU path behind the C-ABI (include/gmr_hip.h, "multi-GPU"): ONE RCCL
// communicator per process (one rank per GPU), used for the single broadcast of the packed robot model + task set
// from rank 0 over xGMI and for the job-level barrier / timing reductions of the drivers.  There is no per-step
// collective in the retargeting path (streams are independent).
//
// No PyTorch: librccl.so is opened at run time (dlopen), the ncclUniqueId travels from rank 0 to the peers over a
// plain TCP socket at MASTER_ADDR : port (the launcher's rendezvous variables), nothing else is needed.
#include <arpa/inet.h>
#include <dlfcn.h>
#include <errno.h>
#include <hip/hip_runtime.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <time.h>
#include <unistd.h>

#include <new>

#include "../../include/gmr_hip.h"
#include "gmr_internal.h"

namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.so) return GMR_OK;
  const char* names[] = {getenv("GMR_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* so = nullptr;
  for (const char* n : names) {
    if (!n || !*n) continue;
    so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (so) break;
  }
  if (!so) return gmr_fail(GMR_ERR_COMM, "librccl.so not found (set GMR_RCCL_LIBRARY): %s", dlerror());
#define SYM(field, name)                                                            \
  *(void**)(&g_rccl.field) = dlsym(so, name);                                       \
  if (!g_rccl.field) { dlclose(so); return gmr_fail(GMR_ERR_COMM, "librccl: symbol %s missing", name); }
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(Broadcast, "ncclBroadcast")
  SYM(AllReduce, "ncclAllReduce")
  SYM(AllGather, "ncclAllGather")
  SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  g_rccl.so = so;
  return GMR_OK;
}

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int send_all(int fd, const void* buf, size_t n) {
  const char* p = (const char*)buf;
  while (n) {
    ssize_t k = send(fd, p, n, MSG_NOSIGNAL);
    if (k < 0) { if (errno == EINTR) continue; return -1; }
    p += k; n -= (size_t)k;
  }
  return 0;
}
int recv_all(int fd, void* buf, size_t n) {
  char* p = (char*)buf;
  while (n) {
    ssize_t k = recv(fd, p, n, 0);
    if (k == 0) return -1;
    if (k < 0) { if (errno == EINTR) continue; return -1; }
    p += k; n -= (size_t)k;
  }
  return 0;
}

#define NCCL_TRY(call)                                                                                       \
  do {                                                                                                       \
    ncclResult_t _r = (call);                                                                                \
    if (_r != ncclSuccess) return gmr_fail(GMR_ERR_COMM, "%s: %s", #call, g_rccl.GetErrorString(_r));        \
  } while (0)
#define HIPC_TRY(call)                                                                                       \
  do {                                                                                                       \
    hipError_t _e = (call);                                                                                  \
    if (_e != hipSuccess) return gmr_fail(GMR_ERR_HIP, "%s: %s", #call, hipGetErrorString(_e));              \
  } while (0)

}  // namespace

struct gmr_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  hipStream_t stream = nullptr;
  void* d_scratch = nullptr;     // small device staging for the host-buffer conveniences
  size_t scratch_bytes = 0;
};

extern "C" {

// Rank 0 hands `bytes` bytes of `payload` to every other rank: it listens on addr:port, every peer connects (retrying
// while rank 0 is not up yet), sends its rank and receives the payload.  Plain TCP, no GPU: also the CPU-tested half
// of gmr_comm_create.
int gmr_bootstrap_exchange(int rank, int world, const char* addr, int port, void* payload, size_t bytes, double timeout_s) {
  if (world < 1 || rank < 0 || rank >= world || !payload) return gmr_fail(GMR_ERR_ARG, "bootstrap: bad rank/world");
  if (world == 1) return GMR_OK;
  if (!addr || !*addr) addr = "127.0.0.1";
  char ports[16];
  snprintf(ports, sizeof ports, "%d", port);
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_UNSPEC;
  hints.ai_socktype = SOCK_STREAM;
  if (rank == 0) hints.ai_flags = AI_PASSIVE;
  int gai = getaddrinfo(addr, ports, &hints, &res);
  if (gai != 0 || !res) return gmr_fail(GMR_ERR_COMM, "bootstrap: cannot resolve %s:%d (%s)", addr, port, gai_strerror(gai));
  const double t_end = now_s() + timeout_s;
  int rc = GMR_OK;
  if (rank == 0) {
    int ls = socket(res->ai_family, SOCK_STREAM, 0);
    int one = 1;
    if (ls >= 0) setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    if (ls < 0 || bind(ls, res->ai_addr, res->ai_addrlen) != 0 || listen(ls, world) != 0) {
      rc = gmr_fail(GMR_ERR_COMM, "bootstrap: rank 0 cannot listen on %s:%d (%s)", addr, port, strerror(errno));
      if (ls >= 0) close(ls);
      freeaddrinfo(res);
      return rc;
    }
    timeval tv{1, 0};
    setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);     // accept() wakes up once a second to check the deadline
    int served = 0;
    char seen[4096] = {0};
    while (served < world - 1) {
      int fd = accept(ls, nullptr, nullptr);
      if (fd < 0) {
        if (now_s() > t_end) { rc = gmr_fail(GMR_ERR_COMM, "bootstrap: only %d of %d peers connected within %.0f s", served, world - 1, timeout_s); break; }
        continue;
      }
      int32_t peer = -1;
      timeval tv2{10, 0};
      setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv2, sizeof tv2);
      if (recv_all(fd, &peer, 4) == 0 && peer > 0 && peer < world && peer < 4096 && !seen[peer] && send_all(fd, payload, bytes) == 0) {
        seen[peer] = 1;
        served++;
      }
      close(fd);
    }
    close(ls);
  } else {
    for (;;) {
      int fd = socket(res->ai_family, SOCK_STREAM, 0);
      if (fd >= 0 && connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
        int32_t me = rank;
        timeval tv{30, 0};
        setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        const bool ok = send_all(fd, &me, 4) == 0 && recv_all(fd, payload, bytes) == 0;
        close(fd);
        if (ok) break;
      } else if (fd >= 0) {
        close(fd);
      }
      if (now_s() > t_end) { rc = gmr_fail(GMR_ERR_COMM, "bootstrap: rank %d could not reach rank 0 at %s:%d within %.0f s", rank, addr, port, timeout_s); break; }
      usleep(50 * 1000);
    }
  }
  freeaddrinfo(res);
  return rc;
}

// Replaces nothing in the reference (it has mp.Pool on one CPU: scripts/smplx_to_robot_dataset.py:241-242); this is
// `gmr_broadcast_model`'s communicator of SURVEY.md section 8(b).  Call after gmr_set_device(local_rank).
int gmr_comm_create(int rank, int world, const char* master_addr, int port, gmr_comm_t** out) {
  if (!out) return gmr_fail(GMR_ERR_ARG, "null out pointer");
  if (world < 1 || rank < 0 || rank >= world) return gmr_fail(GMR_ERR_ARG, "bad rank/world");
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  memset(&id, 0, sizeof id);
  // RCCL prints a start-up banner (version, host, library path) on stdout; callers own stdout (bench.py prints ONE JSON
  // line there): send whatever RCCL prints while it initialises to stderr
  fflush(stdout);
  const int saved_stdout = dup(STDOUT_FILENO);
  if (saved_stdout >= 0) dup2(STDERR_FILENO, STDOUT_FILENO);
  struct RestoreStdout {
    int fd;
    ~RestoreStdout() { if (fd >= 0) { fflush(stdout); dup2(fd, STDOUT_FILENO); close(fd); } }
  } restore{saved_stdout};
  if (rank == 0) NCCL_TRY(g_rccl.GetUniqueId(&id));
  double timeout_s = 120.0;
  if (const char* t = getenv("GMR_COMM_TIMEOUT")) timeout_s = atof(t) > 0 ? atof(t) : timeout_s;
  rc = gmr_bootstrap_exchange(rank, world, master_addr, port, &id, sizeof id, timeout_s);
  if (rc) return rc;
  gmr_comm* c = new (std::nothrow) gmr_comm;
  if (!c) return gmr_fail(GMR_ERR_ARG, "out of host memory");
  c->rank = rank; c->world = world;
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { delete c; return gmr_fail(GMR_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString(r)); }
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) { c->scratch_bytes = 1 << 16; e = hipMalloc(&c->d_scratch, c->scratch_bytes); }
  if (e != hipSuccess) { (void)g_rccl.CommDestroy(c->comm); delete c; return gmr_fail(GMR_ERR_HIP, "gmr_comm_create: %s", hipGetErrorString(e)); }
  *out = c;
  return GMR_OK;
}

int gmr_comm_destroy(gmr_comm_t* c) {
  if (!c) return GMR_OK;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return GMR_OK;
}

int gmr_comm_rank(const gmr_comm_t* c) { return c ? c->rank : 0; }
int gmr_comm_world(const gmr_comm_t* c) { return c ? c->world : 1; }

// The ONE data collective of the path: `bytes` bytes of a DEVICE buffer from `root` to every rank, on `stream`.
int gmr_comm_broadcast_dev(gmr_comm_t* c, void* d_buf, size_t bytes, int root, void* stream) {
  if (!c || !d_buf) return gmr_fail(GMR_ERR_ARG, "null comm / buffer");
  NCCL_TRY(g_rccl.Broadcast(d_buf, d_buf, bytes, ncclUint8, root, c->comm, (hipStream_t)stream));
  return GMR_OK;
}

static int need_scratch(gmr_comm* c, size_t bytes) {
  if (bytes <= c->scratch_bytes) return GMR_OK;
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  c->d_scratch = nullptr; c->scratch_bytes = 0;
  HIPC_TRY(hipMalloc(&c->d_scratch, bytes));
  c->scratch_bytes = bytes;
  return GMR_OK;
}

// gmr_broadcast_model of SURVEY.md section 8(b): host bytes (the packed gmr_model_t + gmr_taskset_t, 24 KB) of rank
// `root` to the same host buffer on every rank: H2D, RCCL broadcast over xGMI, D2H, synchronised.
int gmr_comm_broadcast(gmr_comm_t* c, void* buf, size_t bytes, int root) {
  if (!c || !buf) return gmr_fail(GMR_ERR_ARG, "null comm / buffer");
  if (bytes == 0) return GMR_OK;
  int rc = need_scratch(c, bytes);
  if (rc) return rc;
  if (c->rank == root) HIPC_TRY(hipMemcpyAsync(c->d_scratch, buf, bytes, hipMemcpyHostToDevice, c->stream));
  NCCL_TRY(g_rccl.Broadcast(c->d_scratch, c->d_scratch, bytes, ncclUint8, root, c->comm, c->stream));
  HIPC_TRY(hipMemcpyAsync(buf, c->d_scratch, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPC_TRY(hipStreamSynchronize(c->stream));
  return GMR_OK;
}

// job-level reductions of the drivers (timing): element-wise max / sum of n doubles, result on every rank
static int allreduce_f64(gmr_comm* c, double* inout, int n, ncclRedOp_t op) {
  if (!c || !inout || n < 0) return gmr_fail(GMR_ERR_ARG, "null comm / buffer");
  if (n == 0) return GMR_OK;
  int rc = need_scratch(c, (size_t)n * 8);
  if (rc) return rc;
  HIPC_TRY(hipMemcpyAsync(c->d_scratch, inout, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  NCCL_TRY(g_rccl.AllReduce(c->d_scratch, c->d_scratch, (size_t)n, ncclFloat64, op, c->comm, c->stream));
  HIPC_TRY(hipMemcpyAsync(inout, c->d_scratch, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  HIPC_TRY(hipStreamSynchronize(c->stream));
  return GMR_OK;
}
int gmr_comm_allreduce_max(gmr_comm_t* c, double* inout, int n) { return allreduce_f64(c, inout, n, ncclMax); }
int gmr_comm_allreduce_sum(gmr_comm_t* c, double* inout, int n) { return allreduce_f64(c, inout, n, ncclSum); }

// every rank contributes n doubles; out[world][n] on every rank
int gmr_comm_allgather(gmr_comm_t* c, const double* in, double* out, int n) {
  if (!c || !in || !out || n < 0) return gmr_fail(GMR_ERR_ARG, "null comm / buffer");
  if (n == 0) return GMR_OK;
  const size_t one = (size_t)n * 8;
  int rc = need_scratch(c, one * (size_t)(c->world + 1));
  if (rc) return rc;
  char* d = (char*)c->d_scratch;
  HIPC_TRY(hipMemcpyAsync(d, in, one, hipMemcpyHostToDevice, c->stream));
  NCCL_TRY(g_rccl.AllGather(d, d + one, (size_t)n, ncclFloat64, c->comm, c->stream));
  HIPC_TRY(hipMemcpyAsync(out, d + one, one * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
  HIPC_TRY(hipStreamSynchronize(c->stream));
  return GMR_OK;
}

// all ranks have reached this point and their device work is complete
int gmr_comm_barrier(gmr_comm_t* c) {
  HIPC_TRY(hipDeviceSynchronize());
  double one = 1.0;
  int rc = allreduce_f64(c, &one, 1, ncclSum);
  if (rc) return rc;
  if ((int)(one + 0.5) != c->world) return gmr_fail(GMR_ERR_COMM, "barrier: %d of %d ranks", (int)(one + 0.5), c->world);
  return GMR_OK;
}

}  // extern "C"
