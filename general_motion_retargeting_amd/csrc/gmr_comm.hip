// gmr_comm.hip -- the collectives of the multi-GPU path behind the C-ABI (include/gmr_hip.h, "multi-GPU"): ONE RCCL
// communicator per process (one rank per GPU), used for the single broadcast of the packed robot model + task set
// from rank 0 over xGMI and for the job-level barrier / timing reductions of the drivers.  There is no per-step
// collective in the retargeting path (streams are independent).
//
// No PyTorch: librccl.so is opened at run time (dlopen).  The ranks first form a CONTROL STAR over plain TCP (rank 0
// listens at MASTER_ADDR : port, every peer connects and stays connected); the ncclUniqueId travels over it, and --
// the point of keeping it open -- every step of the RCCL bring-up is AGREED on by all ranks before the next one starts
// (library found on every rank?  ncclCommInitRank succeeded on every rank?), so a failure on one rank becomes the same
// clear error on all of them instead of a hang in somebody's rendezvous.  The star also carries a "tcp" backend of
// the same interface (the job-level plumbing only: barrier, timing reductions, the 24 KB broadcast): the CPU
// rehearsal of the N > 1 path, and -- opt-in, GMR_COMM_FALLBACK=tcp -- what a job falls back to collectively when RCCL
// cannot be brought up, labelled as such.
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
  ncclResult_t (*GetVersion)(int*) = nullptr;      // optional
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.so) return GMR_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* so = nullptr;
  const char* forced = getenv("GMR_RCCL_LIBRARY");
  if (forced && *forced) {                     // an explicit choice is not silently replaced by another library
    so = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    if (!so) return gmr_fail(GMR_ERR_COMM, "GMR_RCCL_LIBRARY=%s does not load: %s", forced, dlerror());
  }
  for (const char* n : names) {
    if (so) break;
    so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
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
  *(void**)(&g_rccl.GetVersion) = dlsym(so, "ncclGetVersion");
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

enum { BACKEND_RCCL = 0, BACKEND_TCP = 1 };

struct gmr_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  int backend = BACKEND_RCCL;
  hipStream_t stream = nullptr;
  void* d_scratch = nullptr;     // small device staging for the host-buffer conveniences
  size_t scratch_bytes = 0;
  // control star: rank 0 holds one socket per peer (index = rank), a peer holds its socket to rank 0
  int* fds = nullptr;
  int fd0 = -1;
  bool has_device = false;
  char label[256] = {0};         // "rccl-2.27.7" / "tcp" / "tcp (fallback: ...)"
};

namespace {

void star_close(gmr_comm* c) {
  if (c->fds) { for (int r = 1; r < c->world; r++) if (c->fds[r] >= 0) close(c->fds[r]); delete[] c->fds; c->fds = nullptr; }
  if (c->fd0 >= 0) { close(c->fd0); c->fd0 = -1; }
}

void sock_opts(int fd, double timeout_s) {
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
  timeval tv{(time_t)timeout_s, 0};
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
  setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
}

// Form the star: rank 0 accepts world - 1 peers (each announces its rank), everybody keeps the connection.
int star_connect(gmr_comm* c, const char* addr, int port, double timeout_s, double io_timeout_s) {
  if (c->world == 1) return GMR_OK;
  if (!addr || !*addr) addr = "127.0.0.1";
  char ports[16];
  snprintf(ports, sizeof ports, "%d", port);
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_UNSPEC;
  hints.ai_socktype = SOCK_STREAM;
  if (c->rank == 0) hints.ai_flags = AI_PASSIVE;
  int gai = getaddrinfo(addr, ports, &hints, &res);
  if (gai != 0 || !res) return gmr_fail(GMR_ERR_COMM, "comm: cannot resolve %s:%d (%s)", addr, port, gai_strerror(gai));
  const double t_end = now_s() + timeout_s;
  int rc = GMR_OK;
  if (c->rank == 0) {
    c->fds = new int[c->world];
    for (int r = 0; r < c->world; r++) c->fds[r] = -1;
    int ls = socket(res->ai_family, SOCK_STREAM, 0);
    int one = 1;
    if (ls >= 0) setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    if (ls < 0 || bind(ls, res->ai_addr, res->ai_addrlen) != 0 || listen(ls, c->world) != 0) {
      rc = gmr_fail(GMR_ERR_COMM, "comm: rank 0 cannot listen on %s:%d (%s)", addr, port, strerror(errno));
      if (ls >= 0) close(ls);
      freeaddrinfo(res);
      return rc;
    }
    timeval tv{1, 0};
    setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);     // accept() wakes up once a second to check the deadline
    int have = 0;
    while (have < c->world - 1) {
      int fd = accept(ls, nullptr, nullptr);
      if (fd < 0) {
        if (now_s() > t_end) { rc = gmr_fail(GMR_ERR_COMM, "comm: only %d of %d peers connected to rank 0 at %s:%d within %.0f s", have, c->world - 1, addr, port, timeout_s); break; }
        continue;
      }
      sock_opts(fd, 10.0);
      int32_t peer = -1;
      if (recv_all(fd, &peer, 4) == 0 && peer > 0 && peer < c->world && c->fds[peer] < 0) {
        sock_opts(fd, io_timeout_s);
        c->fds[peer] = fd;
        have++;
      } else {
        close(fd);
      }
    }
    close(ls);
  } else {
    for (;;) {
      int fd = socket(res->ai_family, SOCK_STREAM, 0);
      if (fd >= 0 && connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
        sock_opts(fd, io_timeout_s);
        int32_t me = c->rank;
        if (send_all(fd, &me, 4) == 0) { c->fd0 = fd; break; }
        close(fd);
      } else if (fd >= 0) {
        close(fd);
      }
      if (now_s() > t_end) { rc = gmr_fail(GMR_ERR_COMM, "comm: rank %d could not reach rank 0 at %s:%d within %.0f s", c->rank, addr, port, timeout_s); break; }
      usleep(50 * 1000);
    }
  }
  freeaddrinfo(res);
  return rc;
}

// gather `n` bytes per rank at rank 0 (all[world][n], rank 0 only); then rank 0 sends `m` bytes of `reply` to everyone
int star_gather(gmr_comm* c, const void* mine, void* all, size_t n) {
  if (c->rank == 0) {
    memcpy(all, mine, n);
    for (int r = 1; r < c->world; r++)
      if (recv_all(c->fds[r], (char*)all + (size_t)r * n, n) != 0) return gmr_fail(GMR_ERR_COMM, "comm: lost rank %d (%s)", r, strerror(errno));
  } else if (send_all(c->fd0, mine, n) != 0) {
    return gmr_fail(GMR_ERR_COMM, "comm: rank %d lost rank 0 (%s)", c->rank, strerror(errno));
  }
  return GMR_OK;
}
int star_scatter_same(gmr_comm* c, void* buf, size_t m) {
  if (c->rank == 0) {
    for (int r = 1; r < c->world; r++)
      if (send_all(c->fds[r], buf, m) != 0) return gmr_fail(GMR_ERR_COMM, "comm: lost rank %d (%s)", r, strerror(errno));
  } else if (recv_all(c->fd0, buf, m) != 0) {
    return gmr_fail(GMR_ERR_COMM, "comm: rank %d lost rank 0 (%s)", c->rank, strerror(errno));
  }
  return GMR_OK;
}

// Every rank reports (ok, message); everybody learns whether ALL were ok and, if not, the first failing rank's message.
struct Verdict { int32_t ok; int32_t rank; char msg[200]; };
int star_agree(gmr_comm* c, bool ok, const char* msg, Verdict* out) {
  Verdict mine{};
  mine.ok = ok ? 1 : 0; mine.rank = c->rank;
  if (!ok && msg) snprintf(mine.msg, sizeof mine.msg, "%s", msg);
  Verdict* all = c->rank == 0 ? new Verdict[c->world] : nullptr;
  int rc = c->world > 1 ? star_gather(c, &mine, all, sizeof mine) : GMR_OK;
  Verdict v = mine;
  if (rc == GMR_OK && c->rank == 0 && c->world > 1) {
    v = all[0];
    for (int r = 0; r < c->world; r++) if (!all[r].ok) { v = all[r]; break; }
  }
  delete[] all;
  if (rc == GMR_OK && c->world > 1) rc = star_scatter_same(c, &v, sizeof v);
  if (rc) return rc;
  *out = v;
  return GMR_OK;
}

int tcp_allreduce(gmr_comm* c, double* inout, int n, int op /* 0 max, 1 sum */) {
  if (c->world == 1 || n == 0) return GMR_OK;
  double* all = c->rank == 0 ? new double[(size_t)c->world * n] : nullptr;
  int rc = star_gather(c, inout, all, (size_t)n * 8);
  if (rc == GMR_OK && c->rank == 0)
    for (int r = 1; r < c->world; r++)
      for (int i = 0; i < n; i++) {
        const double v = all[(size_t)r * n + i];
        inout[i] = op == 0 ? (v > inout[i] ? v : inout[i]) : inout[i] + v;
      }
  delete[] all;
  if (rc) return rc;
  return star_scatter_same(c, inout, (size_t)n * 8);
}

}  // namespace

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
// GMR_COMM_BACKEND=tcp builds the star only; GMR_COMM_FALLBACK=tcp lets a job whose RCCL bring-up fails continue on the
// star (decided by all ranks together; the label says so).  Without it a failure is the same error on every rank.
int gmr_comm_create(int rank, int world, const char* master_addr, int port, gmr_comm_t** out) {
  if (!out) return gmr_fail(GMR_ERR_ARG, "null out pointer");
  if (world < 1 || rank < 0 || rank >= world) return gmr_fail(GMR_ERR_ARG, "bad rank/world");
  const char* be = getenv("GMR_COMM_BACKEND");
  const bool want_tcp = be && strcmp(be, "tcp") == 0;
  const char* fb = getenv("GMR_COMM_FALLBACK");
  const bool may_fall_back = fb && strcmp(fb, "tcp") == 0;
  double timeout_s = 120.0;
  if (const char* t = getenv("GMR_COMM_TIMEOUT")) timeout_s = atof(t) > 0 ? atof(t) : timeout_s;
  double io_timeout_s = 1800.0;   // a rank may wait this long for a peer inside a barrier (e.g. rank 0's one-GPU leg)
  if (const char* t = getenv("GMR_COMM_IO_TIMEOUT")) io_timeout_s = atof(t) > 0 ? atof(t) : io_timeout_s;

  gmr_comm* c = new (std::nothrow) gmr_comm;
  if (!c) return gmr_fail(GMR_ERR_ARG, "out of host memory");
  c->rank = rank; c->world = world;
  int ndev = 0;
  c->has_device = hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
  int rc = star_connect(c, master_addr, port, timeout_s, io_timeout_s);
  if (rc) { star_close(c); delete c; return rc; }
  if (want_tcp) {
    c->backend = BACKEND_TCP;
    snprintf(c->label, sizeof c->label, "tcp");
    *out = c;
    return GMR_OK;
  }

  // RCCL prints a start-up banner (version, host, library path) on stdout; callers own stdout (bench.py prints ONE JSON
  // line there): send whatever RCCL prints while it initialises to stderr
  fflush(stdout);
  const int saved_stdout = dup(STDOUT_FILENO);
  if (saved_stdout >= 0) dup2(STDERR_FILENO, STDOUT_FILENO);
  struct RestoreStdout {
    int fd;
    ~RestoreStdout() { if (fd >= 0) { fflush(stdout); dup2(fd, STDOUT_FILENO); close(fd); } }
  } restore{saved_stdout};

  // step 1 (agreed): the library loads on every rank, rank 0 has an id
  ncclUniqueId id;
  memset(&id, 0, sizeof id);
  char why[200] = {0};
  bool ok = load_rccl() == GMR_OK;
  if (!ok) snprintf(why, sizeof why, "%s", gmr_last_error());
  if (ok && !c->has_device) { ok = false; snprintf(why, sizeof why, "no HIP device visible"); }
  if (ok && rank == 0) {
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) { ok = false; snprintf(why, sizeof why, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r)); }
  }
  Verdict v{};
  rc = star_agree(c, ok, why, &v);
  if (rc == GMR_OK && v.ok && world > 1) rc = star_scatter_same(c, &id, sizeof id);
  // step 2 (agreed): ncclCommInitRank on every rank
  if (rc == GMR_OK && v.ok) {
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    ok = r == ncclSuccess;
    if (!ok) { c->comm = nullptr; snprintf(why, sizeof why, "ncclCommInitRank: %s", g_rccl.GetErrorString(r)); }
    if (ok) {
      hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
      if (e == hipSuccess) { c->scratch_bytes = 1 << 16; e = hipMalloc(&c->d_scratch, c->scratch_bytes); }
      if (e != hipSuccess) { ok = false; snprintf(why, sizeof why, "gmr_comm_create: %s", hipGetErrorString(e)); }
    }
    rc = star_agree(c, ok, why, &v);
  }
  if (rc == GMR_OK && v.ok) {
    int ver = 0;
    if (g_rccl.GetVersion && g_rccl.GetVersion(&ver) == ncclSuccess && ver > 0)
      snprintf(c->label, sizeof c->label, "rccl-%d.%d.%d", ver / 10000, (ver / 100) % 100, ver % 100);
    else
      snprintf(c->label, sizeof c->label, "rccl");
    c->backend = BACKEND_RCCL;
    *out = c;
    return GMR_OK;
  }
  // some rank failed (the same verdict on all of them), or the star itself broke (rc != 0)
  if (c->comm) { (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
  if (rc == GMR_OK && may_fall_back) {
    c->backend = BACKEND_TCP;
    snprintf(c->label, sizeof c->label, "tcp (fallback: RCCL bring-up failed on rank %d: %.150s)", v.rank, v.msg);
    if (rank == 0) fprintf(stderr, "[gmr comm] %s\n", c->label);
    *out = c;
    return GMR_OK;
  }
  char msg[256];
  if (rc == GMR_OK) snprintf(msg, sizeof msg, "RCCL bring-up failed on rank %d: %.200s", v.rank, v.msg);
  else snprintf(msg, sizeof msg, "%.250s", gmr_last_error());
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  star_close(c);
  delete c;
  return gmr_fail(GMR_ERR_COMM, "%s", msg);
}

int gmr_comm_destroy(gmr_comm_t* c) {
  if (!c) return GMR_OK;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  star_close(c);
  delete c;
  return GMR_OK;
}

int gmr_comm_rank(const gmr_comm_t* c) { return c ? c->rank : 0; }
int gmr_comm_world(const gmr_comm_t* c) { return c ? c->world : 1; }
const char* gmr_comm_backend(const gmr_comm_t* c) { return c ? c->label : "none"; }

// The ONE data collective of the path: `bytes` bytes of a DEVICE buffer from `root` to every rank, on `stream`.
int gmr_comm_broadcast_dev(gmr_comm_t* c, void* d_buf, size_t bytes, int root, void* stream) {
  if (!c || !d_buf) return gmr_fail(GMR_ERR_ARG, "null comm / buffer");
  if (c->backend != BACKEND_RCCL) return gmr_fail(GMR_ERR_COMM, "gmr_comm_broadcast_dev needs the RCCL backend (this communicator: %s)", c->label);
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
  if (root < 0 || root >= c->world) return gmr_fail(GMR_ERR_ARG, "bad root");
  if (bytes == 0) return GMR_OK;
  if (c->backend == BACKEND_TCP) {
    if (c->world == 1) return GMR_OK;
    if (root != 0) {                                    // root -> rank 0 first
      if (c->rank == root && send_all(c->fd0, buf, bytes) != 0) return gmr_fail(GMR_ERR_COMM, "comm: lost rank 0");
      if (c->rank == 0 && recv_all(c->fds[root], buf, bytes) != 0) return gmr_fail(GMR_ERR_COMM, "comm: lost rank %d", root);
    }
    return star_scatter_same(c, buf, bytes);
  }
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
  if (c->backend == BACKEND_TCP) return tcp_allreduce(c, inout, n, op == ncclMax ? 0 : 1);
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
  if (c->backend == BACKEND_TCP) {
    if (c->world == 1) { memcpy(out, in, one); return GMR_OK; }
    int rc = star_gather(c, in, out, one);
    if (rc) return rc;
    return star_scatter_same(c, out, one * (size_t)c->world);
  }
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
  if (!c) return gmr_fail(GMR_ERR_ARG, "null comm");
  if (c->has_device) HIPC_TRY(hipDeviceSynchronize());
  double one = 1.0;
  int rc = allreduce_f64(c, &one, 1, ncclSum);
  if (rc) return rc;
  if ((int)(one + 0.5) != c->world) return gmr_fail(GMR_ERR_COMM, "barrier: %d of %d ranks", (int)(one + 0.5), c->world);
  return GMR_OK;
}

}  // extern "C"
