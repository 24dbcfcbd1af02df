// gmr_abi.hip -- the C-ABI of libgmrhip.so (include/gmr_hip.h): handles, memory/stream/event
// helpers and the host side of the two launches.  No compute happens here.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "../../include/gmr_hip.h"
#include "gmr_fk_tree.h"
#include "gmr_ik_layout.h"
#include "gmr_ik_wide_layout.h"
#include "gmr_internal.h"

static_assert(sizeof(gmr_model_t) % 8 == 0, "gmr_model_t must be 8-byte sized");
static_assert(sizeof(gmr_taskset_t) % 8 == 0, "gmr_taskset_t must be 8-byte sized");
static_assert(offsetof(gmr_model_t, timestep) % 8 == 0, "double block of gmr_model_t misaligned");
static_assert(offsetof(gmr_taskset_t, damping) % 8 == 0, "double block of gmr_taskset_t misaligned");

extern "C" hipError_t gmr_launch_ik_streams(const uint4*, const gmr::IkLayout*, const gmr::IkParams*, int, int,
                                            const double*, const double*, const int32_t*, int, double*, int32_t*,
                                            int32_t*, double*, double*, hipStream_t, unsigned long long*);
extern "C" hipError_t gmr_ik_set_max_smem(int nvp, int nw, int tree_small, int bytes);
extern "C" hipError_t gmr_launch_ik_wide(const char*, const gmr::WideLayout*, const gmr::IkParams*, int, int, const double*,
                                         const double*, const int32_t*, int, double*, int32_t*, int32_t*, double*, double*,
                                         hipStream_t, unsigned long long*, void*);
struct gmr_wide_job_desc {       // (gmr_ik_wide.hip)
  const char* d_image; const gmr::WideLayout* L; const gmr::IkParams* P;
  int S, T;
  const double* d_q0; const double* d_human; const int32_t* d_len;
  double* d_q_out; int32_t* d_nsolve; int32_t* d_status; double* d_tgt_out; double* d_err_out;
};
extern "C" hipError_t gmr_launch_ik_wide_group(const gmr_wide_job_desc*, int, int, hipStream_t, unsigned long long*, void*);
extern "C" hipError_t gmr_launch_ik_wide_window(const gmr_wide_job_desc*, int, int, hipStream_t, unsigned long long*, void*, int, int);
extern "C" void* gmr_ik_wide_pool_create();
extern "C" void gmr_ik_wide_pool_destroy(void*);
extern "C" void gmr_ik_wide_pool_set_chunk(void*, int);
extern "C" hipError_t gmr_ik_wide_attributes(int* num_regs, int* lds_bytes, int* max_waves_per_cu);
extern "C" hipError_t gmr_launch_fk_batch(const gmr::FkTree*, const gmr::FkTree*, int, const float*, const float*, const float*,
                                          float*, float*, float*, float*, hipStream_t);
extern "C" int gmr_fk_blocks(int B);
extern "C" hipError_t gmr_launch_fk_segment_min(const float*, int, const int32_t*, int, float*, hipStream_t);

// up to this many streams a launch uses the 4-wave (main + 3 helpers) shape; measured crossover on MI355X
// (tools/shape_sweep.py, G1): S=256 1.04M vs 0.93M frames/s, S=384 1.24M vs 1.40M (NW=4 vs NW=1): the switch
// sits just above one workgroup per CU (256 CUs)
#define GMR_HELPER_MAX_STREAMS 300

namespace {
thread_local char g_err[512] = "";
}
int gmr_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
namespace {
#define fail gmr_fail
#define HIP_TRY(call)                                                                         \
  do {                                                                                        \
    hipError_t _e = (call);                                                                   \
    if (_e != hipSuccess) return fail(GMR_ERR_HIP, "%s: %s", #call, hipGetErrorString(_e));   \
  } while (0)
}  // namespace

struct gmr_solver {
  gmr_model_t model;
  gmr_taskset_t ts;
  gmr::IkLayout layout;          // one wave per stream (many streams)
  gmr::IkLayout layout4;         // main wave + 3 helpers per stream (few streams: latency shape)
  uint4* d_image4 = nullptr;
  int force_waves = 0;           // 0 = choose by stream count, 1 or 4 = forced (gmr_solver_set_waves)
  gmr::IkParams params;
  uint4* d_image = nullptr;      // host-built LDS image of the constants (gmr_ik_layout.h)
  gmr::WideLayout wide;          // throughput shape (gmr_ik_wide.hip): ok = 0 when the robot does not fit it
  char* d_wide = nullptr;        // its global image of the constants
  void* wide_pool = nullptr;     // queue workspaces of its queued dispatch mode (one per HIP stream)
  char* ws = nullptr;            // grow-only device workspace of the host-buffer entry point
  size_t ws_bytes = 0;
  char* pin = nullptr;           // pinned host staging for small calls (one H2D + one D2H per call)
  static constexpr size_t kPinBytes = 1u << 20;
  // the sliced host pipeline (gmr_retarget_group): HIP streams and one device workspace per slice in flight
  static constexpr int kPipe = 4;
  hipStream_t pipe_stream[kPipe] = {nullptr, nullptr, nullptr, nullptr};
  char* pipe_ws[kPipe] = {nullptr, nullptr, nullptr, nullptr};
  size_t pipe_bytes[kPipe] = {0, 0, 0, 0};
};

struct gmr_fk {
  gmr::FkTree tree;
  gmr::FkTree* d_tree = nullptr;
  float* d_min_part = nullptr;
  int min_part_cap = 0;
};

extern "C" {

const char* gmr_last_error(void) { return g_err; }

const char* gmr_backend_info(void) {
  static thread_local char buf[256];
  int dev = -1;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
    snprintf(buf, sizeof buf, "hip:%s device=%d name=%s CUs=%d", p.gcnArchName, dev, p.name, p.multiProcessorCount);
  else
    snprintf(buf, sizeof buf, "hip:no-device");
  return buf;
}

int gmr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int gmr_set_device(int device) { HIP_TRY(hipSetDevice(device)); return GMR_OK; }
size_t gmr_sizeof_model(void) { return sizeof(gmr_model_t); }
size_t gmr_sizeof_taskset(void) { return sizeof(gmr_taskset_t); }

int gmr_malloc(void** ptr, size_t bytes) {
  if (!ptr) return fail(GMR_ERR_ARG, "gmr_malloc: null out pointer");
  HIP_TRY(hipMalloc(ptr, bytes ? bytes : 8));
  return GMR_OK;
}
int gmr_free(void* ptr) { if (ptr) HIP_TRY(hipFree(ptr)); return GMR_OK; }
int gmr_host_alloc(void** ptr, size_t bytes) {
  if (!ptr) return fail(GMR_ERR_ARG, "gmr_host_alloc: null out pointer");
  HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 8, hipHostMallocDefault));
  return GMR_OK;
}
int gmr_host_free(void* ptr) { if (ptr) HIP_TRY(hipHostFree(ptr)); return GMR_OK; }
int gmr_host_register(void* ptr, size_t bytes) {
  if (!ptr) return fail(GMR_ERR_ARG, "gmr_host_register: null pointer");
  HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  return GMR_OK;
}
int gmr_host_unregister(void* ptr) { if (ptr) HIP_TRY(hipHostUnregister(ptr)); return GMR_OK; }
int gmr_memset(void* ptr, int value, size_t bytes, void* stream) {
  HIP_TRY(hipMemsetAsync(ptr, value, bytes, (hipStream_t)stream));
  return GMR_OK;
}
int gmr_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return GMR_OK;
}
int gmr_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return GMR_OK;
}
int gmr_stream_create(void** stream) {
  hipStream_t s;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (void*)s;
  return GMR_OK;
}
int gmr_stream_destroy(void* stream) { HIP_TRY(hipStreamDestroy((hipStream_t)stream)); return GMR_OK; }
int gmr_stream_sync(void* stream) {
  if (stream) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  else HIP_TRY(hipDeviceSynchronize());
  return GMR_OK;
}
int gmr_event_create(void** event) {
  hipEvent_t e;
  HIP_TRY(hipEventCreate(&e));
  *event = (void*)e;
  return GMR_OK;
}
int gmr_event_destroy(void* event) { HIP_TRY(hipEventDestroy((hipEvent_t)event)); return GMR_OK; }
int gmr_event_record(void* event, void* stream) {
  HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return GMR_OK;
}
int gmr_event_elapsed_ms(void* start, void* stop, float* ms) {
  HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
  HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return GMR_OK;
}

// ---- solver ---------------------------------------------------------------------------------
static int validate(const gmr_model_t* m, const gmr_taskset_t* t) {
  if (!m || !t) return fail(GMR_ERR_ARG, "null model/taskset");
  if (m->magic != GMR_MAGIC_MODEL || m->version != GMR_ABI_VERSION) return fail(GMR_ERR_ARG, "bad model blob");
  if (t->magic != GMR_MAGIC_TASKSET || t->version != GMR_ABI_VERSION) return fail(GMR_ERR_ARG, "bad taskset blob");
  if (m->nbody < 1 || m->nbody > GMR_MAX_BODIES || m->nbody > 64) return fail(GMR_ERR_ARG, "nbody out of range");
  if (m->nhinge < 0 || m->nhinge > GMR_MAX_HINGES) return fail(GMR_ERR_ARG, "nhinge out of range");
  if (m->nv != m->nhinge + 6 || m->nq != m->nhinge + 7 || m->nv > GMR_MAX_DOF) return fail(GMR_ERR_ARG, "nq/nv inconsistent");
  if (!(m->timestep > 0.0)) return fail(GMR_ERR_ARG, "timestep must be positive");
  if (m->parent[0] != -1) return fail(GMR_ERR_ARG, "body 0 must be the root");
  for (int b = 1; b < m->nbody; b++) {
    if (m->parent[b] < 0 || m->parent[b] >= b) return fail(GMR_ERR_ARG, "parent[%d] invalid", b);
    if (m->depth[b] != m->depth[m->parent[b]] + 1 || m->depth[b] >= GMR_MAX_DEPTH)
      return fail(GMR_ERR_ARG, "depth[%d] invalid", b);
    for (int d = 0; d <= m->depth[b]; d++)
      if (m->chain[b][d] < 0 || m->chain[b][d] >= m->nbody) return fail(GMR_ERR_ARG, "chain[%d][%d] invalid", b, d);
    if (m->chain[b][m->depth[b]] != b) return fail(GMR_ERR_ARG, "chain[%d] does not end at the body", b);
    if (m->body_hinge[b] < -1 || m->body_hinge[b] >= m->nhinge) return fail(GMR_ERR_ARG, "body_hinge[%d] invalid", b);
  }
  for (int h = 0; h < m->nhinge; h++)
    if (m->hinge_body[h] < 1 || m->hinge_body[h] >= m->nbody || m->body_hinge[m->hinge_body[h]] != h)
      return fail(GMR_ERR_ARG, "hinge_body[%d] invalid", h);
  if (t->nhuman < 1 || t->nhuman > GMR_MAX_HUMAN || t->human_root < 0 || t->human_root >= t->nhuman)
    return fail(GMR_ERR_ARG, "nhuman/human_root out of range");
  for (int s = 0; s < 2; s++) {
    if (t->ntask[s] < 0 || t->ntask[s] > GMR_MAX_TASKS) return fail(GMR_ERR_ARG, "ntask out of range");
    if (t->use_stage[s] && t->ntask[s] == 0) return fail(GMR_ERR_ARG, "stage %d enabled without tasks", s + 1);
    if (t->npair[s] < 0 || t->npair[s] > GMR_MAX_PAIRS) return fail(GMR_ERR_ARG, "npair out of range");
    int p = 0;
    for (int k = 0; k < t->ntask[s]; k++) {
      if (t->task_body[s][k] < 0 || t->task_body[s][k] >= m->nbody) return fail(GMR_ERR_ARG, "task body invalid");
      if (t->task_human[s][k] < 0 || t->task_human[s][k] >= t->nhuman) return fail(GMR_ERR_ARG, "task human invalid");
      if (t->task_col0[s][k] != p) return fail(GMR_ERR_ARG, "task_col0 not contiguous");
      int n = t->task_ncol[s][k];
      if (n < 3 || p + n > t->npair[s]) return fail(GMR_ERR_ARG, "task_ncol invalid");   // (base translations may be pruned)
      for (int c = 0; c < n; c++) {
        int d = t->pair_dof[s][p + c];
        if (t->pair_task[s][p + c] != k || d < 0 || d >= m->nv) return fail(GMR_ERR_ARG, "pair table invalid");
        if (c > 0 && d <= t->pair_dof[s][p + c - 1]) return fail(GMR_ERR_ARG, "pair dofs not ascending");
        if (t->pair_index[s][k][d] != p + c) return fail(GMR_ERR_ARG, "pair_index inconsistent");
      }
      p += n;
    }
    if (p != t->npair[s]) return fail(GMR_ERR_ARG, "npair inconsistent");
  }
  if (t->max_iter < 0 || t->max_iter > 1000) return fail(GMR_ERR_ARG, "max_iter out of range");
  return GMR_OK;
}

int gmr_solver_create(const gmr_model_t* model, const gmr_taskset_t* taskset, gmr_solver_t** out) {
  if (!out) return fail(GMR_ERR_ARG, "null out pointer");
  int rc = validate(model, taskset);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(GMR_ERR_NO_DEVICE, "no HIP device visible");
  gmr_solver* s = new (std::nothrow) gmr_solver;
  if (!s) return fail(GMR_ERR_ARG, "out of host memory");
  s->model = *model;
  s->ts = *taskset;
  if (gmr::ik_padded_nv(s->model.nv) < 0) { delete s; return fail(GMR_ERR_ARG, "nv = %d > 48 is not supported", s->model.nv); }
  s->params = gmr::make_ik_params(s->model, s->ts);
  hipError_t e = hipSuccess;
  for (int v = 0; v < 2 && e == hipSuccess; v++) {
    const int nw = v == 0 ? 1 : 4;
    gmr::IkSchedule sch = gmr::make_ik_schedule(s->model, s->ts, v == 0 ? 64 : 64 * (nw - 1));
    gmr::IkLayout& lay = v == 0 ? s->layout : s->layout4;
    lay = gmr::make_ik_layout(s->model, s->ts, sch, nw);
    if (lay.smem_bytes > 160 * 1024 - 1024) { delete s; return fail(GMR_ERR_ARG, "robot too large for LDS"); }
    std::vector<char> img = gmr::make_ik_image(s->model, s->ts, sch, lay);
    uint4** dst = v == 0 ? &s->d_image : &s->d_image4;
    if ((e = hipMalloc((void**)dst, img.size())) != hipSuccess) break;
    if ((e = hipMemcpy(*dst, img.data(), img.size(), hipMemcpyHostToDevice)) != hipSuccess) break;
    e = gmr_ik_set_max_smem(lay.nvp, nw, lay.tree_small, lay.smem_bytes);
  }
  s->wide = gmr::WideLayout{};
  if (e == hipSuccess && !getenv("GMR_IK_NO_WIDE")) {   // (diagnostic switch: A/B against the one-wavefront kernel of gmr_ik.hip)
    std::vector<char> img;
    s->wide = gmr::make_wide_layout(s->model, s->ts, &img);
    if (s->wide.ok) {
      if ((e = hipMalloc((void**)&s->d_wide, img.size())) == hipSuccess)
        e = hipMemcpy(s->d_wide, img.data(), img.size(), hipMemcpyHostToDevice);
      if (e == hipSuccess) s->wide_pool = gmr_ik_wide_pool_create();
    }
  }
  if (e != hipSuccess) {
    if (s->d_image) (void)hipFree(s->d_image);
    if (s->d_image4) (void)hipFree(s->d_image4);
    if (s->d_wide) (void)hipFree(s->d_wide);
    gmr_ik_wide_pool_destroy(s->wide_pool);
    delete s;
    return fail(GMR_ERR_HIP, "gmr_solver_create: %s", hipGetErrorString(e));
  }
  *out = s;
  return GMR_OK;
}

int gmr_solver_destroy(gmr_solver_t* s) {
  if (!s) return GMR_OK;
  (void)hipFree(s->d_image);
  (void)hipFree(s->d_image4);
  if (s->d_wide) (void)hipFree(s->d_wide);
  gmr_ik_wide_pool_destroy(s->wide_pool);
  if (s->ws) (void)hipFree(s->ws);
  if (s->pin) (void)hipHostFree(s->pin);
  for (int i = 0; i < gmr_solver::kPipe; i++) {
    if (s->pipe_stream[i]) { (void)hipStreamSynchronize(s->pipe_stream[i]); (void)hipStreamDestroy(s->pipe_stream[i]); }
    if (s->pipe_ws[i]) (void)hipFree(s->pipe_ws[i]);
  }
  delete s;
  return GMR_OK;
}

int gmr_solver_dims(const gmr_solver_t* s, int* nq, int* nv, int* nhuman) {
  if (!s) return fail(GMR_ERR_ARG, "null solver");
  if (nq) *nq = s->model.nq;
  if (nv) *nv = s->model.nv;
  if (nhuman) *nhuman = s->ts.nhuman;
  return GMR_OK;
}

int gmr_solver_set_waves(gmr_solver_t* s, int waves_per_stream) {
  if (!s) return fail(GMR_ERR_ARG, "null solver");
  if (waves_per_stream != 0 && waves_per_stream != 1 && waves_per_stream != 4)
    return fail(GMR_ERR_ARG, "waves_per_stream must be 0 (auto), 1 or 4");
  s->force_waves = waves_per_stream;
  return GMR_OK;
}

int gmr_solver_set_dispatch(gmr_solver_t* s, int frames_per_item) {
  if (!s) return fail(GMR_ERR_ARG, "null solver");
  if (frames_per_item < 0) return fail(GMR_ERR_ARG, "frames_per_item must be >= 0");
  if (s->wide_pool) gmr_ik_wide_pool_set_chunk(s->wide_pool, frames_per_item);
  return GMR_OK;
}

int gmr_retarget_lds_bytes(const gmr_solver_t* s) { return s ? s->layout4.smem_bytes : 0; }

int gmr_retarget_streams_dev(gmr_solver_t* s, int S, int T, const double* d_q0, const double* d_human,
                             const int32_t* d_len, int flags, double* d_q_out, int32_t* d_nsolve,
                             int32_t* d_status, double* d_tgt_out, double* d_err_out, void* stream) {
  if (!s) return fail(GMR_ERR_ARG, "null solver");
  if (S < 0 || T < 0) return fail(GMR_ERR_ARG, "negative S/T");
  if (S == 0 || T == 0) return GMR_OK;
  if (!d_q0 || !d_human || !d_q_out || !d_nsolve || !d_status) return fail(GMR_ERR_ARG, "null device buffer");
  // few streams: 4 waves per stream (helpers share the wide assembly phases, shorter per-frame
  // latency); many streams: 1 wave per stream (more streams resident, more frames per second)
  const bool wide = s->layout4.tree_ok && (s->force_waves ? s->force_waves == 4 : S <= GMR_HELPER_MAX_STREAMS);
  if (!wide && s->wide.ok)     // throughput shape: two resident wavefronts per SIMD (gmr_ik_wide.hip)
    HIP_TRY(gmr_launch_ik_wide(s->d_wide, &s->wide, &s->params, S, T, d_q0, d_human, d_len, flags, d_q_out, d_nsolve,
                               d_status, d_tgt_out, d_err_out, (hipStream_t)stream, nullptr, s->wide_pool));
  else
    HIP_TRY(gmr_launch_ik_streams(wide ? s->d_image4 : s->d_image, wide ? &s->layout4 : &s->layout, &s->params, S, T,
                                  d_q0, d_human, d_len, flags, d_q_out, d_nsolve, d_status, d_tgt_out, d_err_out,
                                  (hipStream_t)stream, nullptr));
  return GMR_OK;
}


// ---- group launches: several (robot, task set) jobs as one scheduling domain ---------------------------------------
static bool job_takes_wide_shape(const gmr_solver* s, long long total_streams) {
  const bool helpers = s->layout4.tree_ok && (s->force_waves ? s->force_waves == 4 : total_streams <= GMR_HELPER_MAX_STREAMS);
  return !helpers && s->wide.ok;
}

int gmr_retarget_group_dev(const gmr_job_t* jobs, int njobs, int flags, void* stream) {
  if (njobs < 0 || (njobs > 0 && !jobs)) return fail(GMR_ERR_ARG, "bad job list");
  long long total = 0;
  for (int j = 0; j < njobs; j++) {
    const gmr_job_t& J = jobs[j];
    if (!J.solver) return fail(GMR_ERR_ARG, "job %d: null solver", j);
    if (J.S < 0 || J.T < 0) return fail(GMR_ERR_ARG, "job %d: negative S/T", j);
    if (J.S == 0 || J.T == 0) continue;
    if (!J.q0 || !J.human || !J.q_out || !J.nsolve || !J.status) return fail(GMR_ERR_ARG, "job %d: null device buffer", j);
    total += J.S;
  }
  // The throughput kernel is one instance for every robot of its size class: all jobs that take that shape at this
  // width (the width of the GROUP decides, not a job's own) go out as launches of up to 8 jobs; the others -- robots
  // that do not decompose, solvers forced to the latency shape, groups too small for the throughput shape -- one by one.
  gmr_wide_job_desc wd[8];
  int nw = 0;
  void* pool = nullptr;
  auto flush = [&]() -> int {
    if (nw == 0) return GMR_OK;
    hipError_t e = gmr_launch_ik_wide_group(wd, nw, flags, (hipStream_t)stream, nullptr, pool);
    nw = 0;
    if (e != hipSuccess) return fail(GMR_ERR_HIP, "group launch: %s", hipGetErrorString(e));
    return GMR_OK;
  };
  for (int j = 0; j < njobs; j++) {
    const gmr_job_t& J = jobs[j];
    if (J.S == 0 || J.T == 0) continue;
    gmr_solver* s = J.solver;
    if (njobs > 1 && job_takes_wide_shape(s, total)) {
      if (nw == 0) pool = s->wide_pool;
      wd[nw++] = gmr_wide_job_desc{s->d_wide, &s->wide, &s->params, J.S, J.T, J.q0, J.human, J.len, J.q_out, J.nsolve, J.status,
                                   J.tgt_out, J.err_out};
      if (nw == 8) { int rc = flush(); if (rc) return rc; }
    } else {
      int rc = gmr_retarget_streams_dev(s, J.S, J.T, J.q0, J.human, J.len, flags, J.q_out, J.nsolve, J.status, J.tgt_out,
                                        J.err_out, stream);
      if (rc) return rc;
    }
  }
  return flush();
}

// One WINDOW of every stream's frames: [t_begin, t_end).  The windows of a batch must be launched in order on ONE stream,
// starting at t_begin = 0; the per-stream state between them (QP bound sets, status) stays in the first job's workspace of
// that stream; q continues from the previous window's last q_out row.  Only batches that take the throughput shape as a
// whole (every job's robot decomposes, more than 300 streams, no solver forced to four wavefronts) can be windowed.
static bool group_is_windowable(const gmr_job_t* jobs, int njobs) {
  long long total = 0;
  int n = 0;
  for (int j = 0; j < njobs; j++) if (jobs[j].S > 0 && jobs[j].T > 0) { total += jobs[j].S; n++; }
  if (n == 0 || n > 8) return false;
  for (int j = 0; j < njobs; j++)
    if (jobs[j].S > 0 && jobs[j].T > 0 && !job_takes_wide_shape(jobs[j].solver, total)) return false;
  return true;
}

int gmr_retarget_group_window_dev(const gmr_job_t* jobs, int njobs, int flags, int t_begin, int t_end, void* stream) {
  if (njobs < 0 || (njobs > 0 && !jobs)) return fail(GMR_ERR_ARG, "bad job list");
  if (t_begin < 0 || t_end <= t_begin) return fail(GMR_ERR_ARG, "bad window [%d, %d)", t_begin, t_end);
  for (int j = 0; j < njobs; j++) {
    const gmr_job_t& J = jobs[j];
    if (!J.solver) return fail(GMR_ERR_ARG, "job %d: null solver", j);
    if (J.S < 0 || J.T < 0) return fail(GMR_ERR_ARG, "job %d: negative S/T", j);
    if (J.S > 0 && J.T > 0 && (!J.q0 || !J.human || !J.q_out || !J.nsolve || !J.status)) return fail(GMR_ERR_ARG, "job %d: null device buffer", j);
  }
  if (!group_is_windowable(jobs, njobs)) return fail(GMR_ERR_ARG, "this batch does not take the throughput shape as a whole: launch it unwindowed");
  gmr_wide_job_desc wd[8];
  int nw = 0;
  void* pool = nullptr;
  for (int j = 0; j < njobs; j++) {
    const gmr_job_t& J = jobs[j];
    if (J.S == 0 || J.T == 0) continue;
    gmr_solver* s = J.solver;
    if (nw == 0) pool = s->wide_pool;
    wd[nw++] = gmr_wide_job_desc{s->d_wide, &s->wide, &s->params, J.S, J.T, J.q0, J.human, J.len, J.q_out, J.nsolve, J.status,
                                 J.tgt_out, J.err_out};
  }
  if (nw == 0) return GMR_OK;
  hipError_t e = gmr_launch_ik_wide_window(wd, nw, flags, (hipStream_t)stream, nullptr, pool, t_begin, t_end);
  if (e != hipSuccess) return fail(GMR_ERR_HIP, "window launch: %s", hipGetErrorString(e));
  return GMR_OK;
}

// Host buffers of a LONG, NARROW batch (a few thousand streams of hundreds of frames: BASELINE.json configs[3], a dataset
// batch), overlapped in TIME: the batch is launched as W consecutive windows of frames; the H2D copies of window w + 1
// (strided: a window of a stream's frames is one piece of its row) run under the kernel of window w, the D2H copies of
// window w - 1 likewise.  (Cutting such a batch by streams loses: launches narrower than the resident width are bound by
// their longest stream.)  Three HIP streams, one device workspace holding the whole batch; results are bit-identical to
// one launch -- the per-stream state travels from window to window in device memory.
static int retarget_group_windows(const gmr_job_t* jobs, int njobs, int flags, gmr_solver* owner, int W) {
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  struct Off { size_t q0, h, len, qo, ns, st, tg, er; };
  std::vector<Off> off((size_t)njobs);
  size_t need = 0;
  int maxT = 0;
  for (int j = 0; j < njobs; j++) {
    const gmr_job_t& J = jobs[j];
    if (J.S == 0 || J.T == 0) continue;
    const size_t nq = J.solver->model.nq, nh = J.solver->ts.nhuman, S = (size_t)J.S, T = (size_t)J.T;
    Off& o = off[j];
    o.q0 = need; need += up(S * nq * 8);
    o.h = need; need += up(S * T * nh * 56);
    o.len = need; need += up(S * 4);
    o.qo = need; need += up(S * T * nq * 8);
    o.ns = need; need += up(S * T * 8);
    o.st = need; need += up(S * 4);
    o.tg = need; need += J.tgt_out ? up(S * T * nh * 56) : 0;
    o.er = need; need += J.err_out ? up(S * T * 16) : 0;
    maxT = std::max(maxT, J.T);
  }
  for (int i = 0; i < 3; i++)
    if (!owner->pipe_stream[i]) HIP_TRY(hipStreamCreateWithFlags(&owner->pipe_stream[i], hipStreamNonBlocking));
  hipStream_t s_in = owner->pipe_stream[0], s_k = owner->pipe_stream[1], s_out = owner->pipe_stream[2];
  if (need > owner->pipe_bytes[0]) {
    for (int i = 0; i < 3; i++) HIP_TRY(hipStreamSynchronize(owner->pipe_stream[i]));
    if (owner->pipe_ws[0]) (void)hipFree(owner->pipe_ws[0]);
    owner->pipe_ws[0] = nullptr; owner->pipe_bytes[0] = 0;
    HIP_TRY(hipMalloc((void**)&owner->pipe_ws[0], need));
    owner->pipe_bytes[0] = need;
  }
  char* d = owner->pipe_ws[0];
  W = std::max(2, std::min(W, maxT / 2));
  int wl = (maxT + W - 1) / W;
  wl = (wl + 3) / 4 * 4;                              // whole queue items (4 frames) per window
  std::vector<gmr_job_t> dj((size_t)njobs);
  std::vector<hipEvent_t> ev;
  int rc = GMR_OK;
  hipError_t e = hipSuccess;
  auto fin = [&](int code) {                          // drain and release whatever was started
    for (int i = 0; i < 3; i++) (void)hipStreamSynchronize(owner->pipe_stream[i]);
    for (hipEvent_t x : ev) (void)hipEventDestroy(x);
    return code;
  };
  auto new_event = [&](hipEvent_t* out) { e = hipEventCreateWithFlags(out, hipEventDisableTiming); if (e == hipSuccess) ev.push_back(*out); return e; };
  for (int j = 0; j < njobs && rc == GMR_OK; j++) {
    const gmr_job_t& J = jobs[j];
    gmr_job_t& D = dj[j];
    D = J;
    if (J.S == 0 || J.T == 0) continue;
    const Off& o = off[j];
    const size_t nq = J.solver->model.nq, nh = J.solver->ts.nhuman, S = (size_t)J.S, T = (size_t)J.T;
    if ((e = hipMemcpyAsync(d + o.q0, J.q0, S * nq * 8, hipMemcpyHostToDevice, s_in)) != hipSuccess ||
        (J.len && (e = hipMemcpyAsync(d + o.len, J.len, S * 4, hipMemcpyHostToDevice, s_in)) != hipSuccess))
      rc = fail(GMR_ERR_HIP, "H2D copy: %s", hipGetErrorString(e));
    if (rc == GMR_OK && (J.len || J.tgt_out || J.err_out)) {      // rows the kernel does not write come back as zeros
      const size_t end = J.err_out ? o.er + up(S * T * 16) : (J.tgt_out ? o.tg + up(S * T * nh * 56) : o.st + up(S * 4));
      if ((e = hipMemsetAsync(d + o.qo, 0, end - o.qo, s_k)) != hipSuccess) rc = fail(GMR_ERR_HIP, "memset: %s", hipGetErrorString(e));
    }
    D.q0 = (const double*)(d + o.q0); D.human = (const double*)(d + o.h); D.len = J.len ? (const int32_t*)(d + o.len) : nullptr;
    D.q_out = (double*)(d + o.qo); D.nsolve = (int32_t*)(d + o.ns); D.status = (int32_t*)(d + o.st);
    D.tgt_out = J.tgt_out ? (double*)(d + o.tg) : nullptr; D.err_out = J.err_out ? (double*)(d + o.er) : nullptr;
  }
  for (int tb = 0; tb < maxT && rc == GMR_OK; tb += wl) {
    const int te = std::min(tb + wl, maxT);
    for (int j = 0; j < njobs && rc == GMR_OK; j++) {
      const gmr_job_t& J = jobs[j];
      if (J.S == 0 || J.T == 0 || tb >= J.T) continue;
      const size_t row = (size_t)J.solver->ts.nhuman * 56, pitch = (size_t)J.T * row, width = (size_t)(std::min(te, (int)J.T) - tb) * row;
      if ((e = hipMemcpy2DAsync(d + off[j].h + (size_t)tb * row, pitch, (const char*)J.human + (size_t)tb * row, pitch, width, (size_t)J.S,
                                hipMemcpyHostToDevice, s_in)) != hipSuccess)
        rc = fail(GMR_ERR_HIP, "H2D window copy: %s", hipGetErrorString(e));
    }
    hipEvent_t e_in = nullptr, e_k = nullptr;
    if (rc == GMR_OK && (new_event(&e_in) != hipSuccess || (e = hipEventRecord(e_in, s_in)) != hipSuccess ||
                         (e = hipStreamWaitEvent(s_k, e_in, 0)) != hipSuccess))
      rc = fail(GMR_ERR_HIP, "window events: %s", hipGetErrorString(e));
    if (rc == GMR_OK) rc = gmr_retarget_group_window_dev(dj.data(), njobs, flags, tb, te, s_k);
    if (rc == GMR_OK && (new_event(&e_k) != hipSuccess || (e = hipEventRecord(e_k, s_k)) != hipSuccess ||
                         (e = hipStreamWaitEvent(s_out, e_k, 0)) != hipSuccess))
      rc = fail(GMR_ERR_HIP, "window events: %s", hipGetErrorString(e));
    for (int j = 0; j < njobs && rc == GMR_OK; j++) {
      const gmr_job_t& J = jobs[j];
      if (J.S == 0 || J.T == 0 || tb >= J.T) continue;
      const Off& o = off[j];
      const size_t nq = J.solver->model.nq, nh = J.solver->ts.nhuman, T = (size_t)J.T, S = (size_t)J.S, n = (size_t)(std::min(te, (int)J.T) - tb);
      auto back = [&](void* host, size_t dev_off, size_t row) {
        return hipMemcpy2DAsync((char*)host + (size_t)tb * row, T * row, d + dev_off + (size_t)tb * row, T * row, n * row, S, hipMemcpyDeviceToHost, s_out);
      };
      if ((e = back(J.q_out, o.qo, nq * 8)) != hipSuccess || (e = back(J.nsolve, o.ns, 8)) != hipSuccess ||
          (J.tgt_out && (e = back(J.tgt_out, o.tg, nh * 56)) != hipSuccess) || (J.err_out && (e = back(J.err_out, o.er, 16)) != hipSuccess))
        rc = fail(GMR_ERR_HIP, "D2H window copy: %s", hipGetErrorString(e));
    }
  }
  for (int j = 0; j < njobs && rc == GMR_OK; j++) {
    const gmr_job_t& J = jobs[j];
    if (J.S == 0 || J.T == 0) continue;
    if ((e = hipMemcpyAsync(J.status, d + off[j].st, (size_t)J.S * 4, hipMemcpyDeviceToHost, s_out)) != hipSuccess)
      rc = fail(GMR_ERR_HIP, "D2H copy: %s", hipGetErrorString(e));
  }
  for (int i = 0; i < 3 && rc == GMR_OK; i++)
    if ((e = hipStreamSynchronize(owner->pipe_stream[i])) != hipSuccess) rc = fail(GMR_ERR_HIP, "kernel / copies: %s", hipGetErrorString(e));
  return fin(rc);
}

// Host buffers, sliced and overlapped: slice k holds the streams [S_j k / n, S_j (k + 1) / n) of every job; its H2D
// copies, its (group) launch and its D2H copies go to HIP stream k mod 4 in that order, so the copies of one slice run
// under the kernels of its neighbours.  Slices are cut only while every slice still has a few thousand streams: a
// launch narrower than the resident width is bound by its longest stream, and slices that share a HIP stream would
// run those one after the other.  Pinned host memory (gmr_host_alloc / gmr_host_register) makes the copies truly
// asynchronous; pageable memory works, staged by the runtime.
int gmr_retarget_group(const gmr_job_t* jobs, int njobs, int flags, int slices) {
  if (njobs < 0 || (njobs > 0 && !jobs)) return fail(GMR_ERR_ARG, "bad job list");
  long long total = 0;
  size_t in_bytes = 0;
  gmr_solver* owner = nullptr;
  for (int j = 0; j < njobs; j++) {
    const gmr_job_t& J = jobs[j];
    if (!J.solver) return fail(GMR_ERR_ARG, "job %d: null solver", j);
    if (J.S < 0 || J.T < 0) return fail(GMR_ERR_ARG, "job %d: negative S/T", j);
    if (J.S == 0 || J.T == 0) continue;
    if (!J.q0 || !J.human || !J.q_out || !J.nsolve || !J.status) return fail(GMR_ERR_ARG, "job %d: null host buffer", j);
    if (!owner) owner = J.solver;
    total += J.S;
    in_bytes += (size_t)J.S * J.T * J.solver->ts.nhuman * 56;
  }
  if (!owner) return GMR_OK;
  const long long min_slice_streams = 4096;          // twice the resident width of an MI355X (8 wavefronts x 256 CUs)
  int n;
  if (slices > 0) n = (int)std::min<long long>(slices, total);                                   // the caller's choice
  else if (slices < 0) n = 1;
  else n = (int)std::max<long long>(1, std::min<long long>(std::min<long long>(16, (long long)(in_bytes >> 26)),   // ~64 MB of input per slice
                                                            total / min_slice_streams));
  // a batch too narrow to be cut by streams but long enough to be cut in time (slices < 0: |slices| windows, the caller's choice)
  {
    int maxT = 0;
    for (int j = 0; j < njobs; j++) if (jobs[j].S > 0) maxT = std::max(maxT, (int)jobs[j].T);
    const bool want = slices < 0 || (slices == 0 && n == 1 && maxT >= 32 && in_bytes >= ((size_t)64 << 20));
    if (want && maxT >= 4 && group_is_windowable(jobs, njobs) && !getenv("GMR_NO_WINDOWS"))
      return retarget_group_windows(jobs, njobs, flags, owner, slices < 0 ? -slices : (int)std::min<size_t>(8, in_bytes >> 26));
  }
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  int rc = GMR_OK;
  hipError_t e = hipSuccess;
  const int nslot = std::min(n, (int)gmr_solver::kPipe);
  for (int i = 0; i < nslot; i++)
    if (!owner->pipe_stream[i]) HIP_TRY(hipStreamCreateWithFlags(&owner->pipe_stream[i], hipStreamNonBlocking));
  std::vector<gmr_job_t> dj((size_t)njobs);
  for (int k = 0; k < n && rc == GMR_OK; k++) {
    const int slot = k % gmr_solver::kPipe;
    hipStream_t st = owner->pipe_stream[slot];
    // layout of this slice in the slot's workspace
    size_t need = 0;
    struct Off { size_t q0, h, len, qo, ns, st, tg, er; int s0, S; };
    std::vector<Off> off((size_t)njobs);
    for (int j = 0; j < njobs; j++) {
      const gmr_job_t& J = jobs[j];
      Off& o = off[j];
      o.s0 = (int)((long long)J.S * k / n);
      o.S = (int)((long long)J.S * (k + 1) / n) - o.s0;
      if (J.S == 0 || J.T == 0) { o.S = 0; continue; }
      const size_t nq = J.solver->model.nq, nh = J.solver->ts.nhuman, S = (size_t)o.S, T = (size_t)J.T;
      o.q0 = need; need += up(S * nq * 8);
      o.h = need; need += up(S * T * nh * 56);
      o.len = need; need += up(S * 4);
      o.qo = need; need += up(S * T * nq * 8);
      o.ns = need; need += up(S * T * 8);
      o.st = need; need += up(S * 4);
      o.tg = need; need += J.tgt_out ? up(S * T * nh * 56) : 0;
      o.er = need; need += J.err_out ? up(S * T * 16) : 0;
    }
    if (need > owner->pipe_bytes[slot]) {
      HIP_TRY(hipStreamSynchronize(st));
      if (owner->pipe_ws[slot]) (void)hipFree(owner->pipe_ws[slot]);
      owner->pipe_ws[slot] = nullptr; owner->pipe_bytes[slot] = 0;
      HIP_TRY(hipMalloc((void**)&owner->pipe_ws[slot], need));
      owner->pipe_bytes[slot] = need;
    }
    char* d = owner->pipe_ws[slot];
    for (int j = 0; j < njobs && rc == GMR_OK; j++) {
      const gmr_job_t& J = jobs[j];
      const Off& o = off[j];
      gmr_job_t& D = dj[j];
      D = J;
      D.S = o.S;
      if (o.S == 0) continue;
      const size_t nq = J.solver->model.nq, nh = J.solver->ts.nhuman, S = (size_t)o.S, T = (size_t)J.T, s0 = (size_t)o.s0;
      if ((e = hipMemcpyAsync(d + o.q0, J.q0 + s0 * nq, S * nq * 8, hipMemcpyHostToDevice, st)) != hipSuccess ||
          (e = hipMemcpyAsync(d + o.h, J.human + s0 * T * nh * 7, S * T * nh * 56, hipMemcpyHostToDevice, st)) != hipSuccess ||
          (J.len && (e = hipMemcpyAsync(d + o.len, J.len + s0, S * 4, hipMemcpyHostToDevice, st)) != hipSuccess))
        rc = fail(GMR_ERR_HIP, "H2D copy: %s", hipGetErrorString(e));
      // rows the kernel does not write come back as zeros (see gmr_hip.h)
      if (rc == GMR_OK && (J.len || J.tgt_out || J.err_out)) {
        const size_t end = J.err_out ? o.er + up(S * T * 16) : (J.tgt_out ? o.tg + up(S * T * nh * 56) : o.st + up(S * 4));
        if ((e = hipMemsetAsync(d + o.qo, 0, end - o.qo, st)) != hipSuccess) rc = fail(GMR_ERR_HIP, "memset: %s", hipGetErrorString(e));
      }
      D.q0 = (const double*)(d + o.q0); D.human = (const double*)(d + o.h); D.len = J.len ? (const int32_t*)(d + o.len) : nullptr;
      D.q_out = (double*)(d + o.qo); D.nsolve = (int32_t*)(d + o.ns); D.status = (int32_t*)(d + o.st);
      D.tgt_out = J.tgt_out ? (double*)(d + o.tg) : nullptr; D.err_out = J.err_out ? (double*)(d + o.er) : nullptr;
    }
    if (rc == GMR_OK) rc = gmr_retarget_group_dev(dj.data(), njobs, flags, st);
    for (int j = 0; j < njobs && rc == GMR_OK; j++) {
      const gmr_job_t& J = jobs[j];
      const Off& o = off[j];
      if (o.S == 0) continue;
      const size_t nq = J.solver->model.nq, nh = J.solver->ts.nhuman, S = (size_t)o.S, T = (size_t)J.T, s0 = (size_t)o.s0;
      if ((e = hipMemcpyAsync(J.q_out + s0 * T * nq, d + o.qo, S * T * nq * 8, hipMemcpyDeviceToHost, st)) != hipSuccess ||
          (e = hipMemcpyAsync(J.nsolve + s0 * T * 2, d + o.ns, S * T * 8, hipMemcpyDeviceToHost, st)) != hipSuccess ||
          (e = hipMemcpyAsync(J.status + s0, d + o.st, S * 4, hipMemcpyDeviceToHost, st)) != hipSuccess ||
          (J.tgt_out && (e = hipMemcpyAsync(J.tgt_out + s0 * T * nh * 7, d + o.tg, S * T * nh * 56, hipMemcpyDeviceToHost, st)) != hipSuccess) ||
          (J.err_out && (e = hipMemcpyAsync(J.err_out + s0 * T * 2, d + o.er, S * T * 16, hipMemcpyDeviceToHost, st)) != hipSuccess))
        rc = fail(GMR_ERR_HIP, "D2H copy: %s", hipGetErrorString(e));
    }
  }
  for (int i = 0; i < nslot; i++)
    if ((e = hipStreamSynchronize(owner->pipe_stream[i])) != hipSuccess && rc == GMR_OK)
      rc = fail(GMR_ERR_HIP, "kernel / copies: %s", hipGetErrorString(e));
  return rc;
}

#ifdef GMR_IK_PROFILE
// diagnostic builds only (tools/phase_profile.py): per-stream phase cycle counters
int gmr_retarget_streams_prof(gmr_solver_t* s, int S, int T, const double* d_q0, const double* d_human, int flags,
                              double* d_q_out, int32_t* d_nsolve, int32_t* d_status, unsigned long long* d_prof) {
  const bool wide = s->layout4.tree_ok && (s->force_waves ? s->force_waves == 4 : S <= GMR_HELPER_MAX_STREAMS);
  if (!wide && s->wide.ok)
    HIP_TRY(gmr_launch_ik_wide(s->d_wide, &s->wide, &s->params, S, T, d_q0, d_human, nullptr, flags, d_q_out, d_nsolve,
                               d_status, nullptr, nullptr, nullptr, d_prof, nullptr));
  else
    HIP_TRY(gmr_launch_ik_streams(wide ? s->d_image4 : s->d_image, wide ? &s->layout4 : &s->layout, &s->params, S, T,
                                  d_q0, d_human, nullptr, flags, d_q_out, d_nsolve, d_status, nullptr, nullptr, nullptr, d_prof));
  return GMR_OK;
}
#endif

int gmr_retarget_streams(gmr_solver_t* s, int S, int T, const double* q0, const double* human, const int32_t* len,
                         int flags, double* q_out, int32_t* nsolve, int32_t* status, double* tgt_out, double* err_out) {
  if (!s) return fail(GMR_ERR_ARG, "null solver");
  if (S < 0 || T < 0) return fail(GMR_ERR_ARG, "negative S/T");
  if (S == 0 || T == 0) return GMR_OK;
  if (!q0 || !human || !q_out || !nsolve || !status) return fail(GMR_ERR_ARG, "null host buffer");
  const size_t nq = s->model.nq, nh = s->ts.nhuman;
  const size_t b_q0 = (size_t)S * nq * 8, b_h = (size_t)S * T * nh * 7 * 8, b_qo = (size_t)S * T * nq * 8;
  const size_t b_ns = (size_t)S * T * 2 * 4, b_st = (size_t)S * 4, b_len = (size_t)S * 4;
  const size_t b_tg = tgt_out ? b_h : 0, b_er = err_out ? (size_t)S * T * 2 * 8 : 0;
  if (b_h >= ((size_t)32 << 20)) {       // large batches: sliced, copies overlapped with the kernels (gmr_retarget_group)
    const gmr_job_t job{s, S, T, q0, human, len, q_out, nsolve, status, tgt_out, err_out};
    return gmr_retarget_group(&job, 1, flags, 0);
  }
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  size_t o_q0 = 0, o_h = o_q0 + up(b_q0), o_len = o_h + up(b_h), o_qo = o_len + up(b_len), o_ns = o_qo + up(b_qo),
         o_st = o_ns + up(b_ns), o_tg = o_st + up(b_st), o_er = o_tg + up(b_tg), total = o_er + up(b_er);
  // grow-only workspace kept on the handle: a per-frame caller (retarget() once per frame) pays no
  // hipMalloc/hipFree per call.  Like the reference object, a handle is not re-entrant on this path.
  if (total > s->ws_bytes) {
    if (s->ws) (void)hipFree(s->ws);
    s->ws = nullptr;
    s->ws_bytes = 0;
    size_t want = total < (1u << 20) ? (1u << 20) : total;
    HIP_TRY(hipMalloc((void**)&s->ws, want));
    s->ws_bytes = want;
  }
  char* d = s->ws;
  int rc = GMR_OK;
  hipError_t e;
  // buffers are laid out [q0 | human | len | q_out | nsolve | status | tgt_out | err_out]; small calls (the per-frame
  // API) go through pinned staging so that a call is 1 H2D + 1 launch + 1 D2H + 1 sync
  const size_t in_bytes = o_qo, out_bytes = total - o_qo;
  const bool small = total <= gmr_solver::kPinBytes;
  if (small && !s->pin) HIP_TRY(hipHostMalloc((void**)&s->pin, gmr_solver::kPinBytes, hipHostMallocDefault));
  if (small) {
    memcpy(s->pin + o_q0, q0, b_q0);
    memcpy(s->pin + o_h, human, b_h);
    if (len) memcpy(s->pin + o_len, len, b_len);
    if ((e = hipMemcpyAsync(d, s->pin, in_bytes, hipMemcpyHostToDevice, nullptr)) != hipSuccess)
      rc = fail(GMR_ERR_HIP, "H2D copy: %s", hipGetErrorString(e));
  } else if ((e = hipMemcpyAsync(d + o_q0, q0, b_q0, hipMemcpyHostToDevice, nullptr)) != hipSuccess ||
             (e = hipMemcpyAsync(d + o_h, human, b_h, hipMemcpyHostToDevice, nullptr)) != hipSuccess ||
             (len && (e = hipMemcpyAsync(d + o_len, len, b_len, hipMemcpyHostToDevice, nullptr)) != hipSuccess)) {
    rc = fail(GMR_ERR_HIP, "H2D copy: %s", hipGetErrorString(e));
  }
  // frames at or beyond len[s] are not touched by the kernel, nor are the tgt_out / err_out rows of the frames after a
  // stream's status turned non-OK (err_out: including the failing frame): hand them back as zeros, never as what an
  // earlier call left in the grow-only workspace
  if (rc == GMR_OK && (len || tgt_out || err_out) && (e = hipMemsetAsync(d + o_qo, 0, out_bytes, nullptr)) != hipSuccess)
    rc = fail(GMR_ERR_HIP, "memset: %s", hipGetErrorString(e));
  if (rc == GMR_OK)
    rc = gmr_retarget_streams_dev(s, S, T, (double*)(d + o_q0), (double*)(d + o_h), len ? (int32_t*)(d + o_len) : nullptr,
                                  flags, (double*)(d + o_qo), (int32_t*)(d + o_ns), (int32_t*)(d + o_st),
                                  tgt_out ? (double*)(d + o_tg) : nullptr, err_out ? (double*)(d + o_er) : nullptr, nullptr);
  if (rc == GMR_OK && small) {
    if ((e = hipMemcpyAsync(s->pin + o_qo, d + o_qo, out_bytes, hipMemcpyDeviceToHost, nullptr)) != hipSuccess ||
        (e = hipStreamSynchronize(nullptr)) != hipSuccess)
      rc = fail(GMR_ERR_HIP, "kernel / D2H copy: %s", hipGetErrorString(e));
    else {
      memcpy(q_out, s->pin + o_qo, b_qo);
      memcpy(nsolve, s->pin + o_ns, b_ns);
      memcpy(status, s->pin + o_st, b_st);
      if (tgt_out) memcpy(tgt_out, s->pin + o_tg, b_tg);
      if (err_out) memcpy(err_out, s->pin + o_er, b_er);
    }
  } else if (rc == GMR_OK) {
    if ((e = hipMemcpyAsync(q_out, d + o_qo, b_qo, hipMemcpyDeviceToHost, nullptr)) != hipSuccess ||
        (e = hipMemcpyAsync(nsolve, d + o_ns, b_ns, hipMemcpyDeviceToHost, nullptr)) != hipSuccess ||
        (e = hipMemcpyAsync(status, d + o_st, b_st, hipMemcpyDeviceToHost, nullptr)) != hipSuccess ||
        (tgt_out && (e = hipMemcpyAsync(tgt_out, d + o_tg, b_tg, hipMemcpyDeviceToHost, nullptr)) != hipSuccess) ||
        (err_out && (e = hipMemcpyAsync(err_out, d + o_er, b_er, hipMemcpyDeviceToHost, nullptr)) != hipSuccess) ||
        (e = hipStreamSynchronize(nullptr)) != hipSuccess)
      rc = fail(GMR_ERR_HIP, "kernel / D2H copy: %s", hipGetErrorString(e));
  }
  return rc;
}

// ---- post-hoc FK ----------------------------------------------------------------------------
int gmr_fk_create(int nbody, const int32_t* parent, const float* local_t, const float* local_r,
                  const int32_t* dof_idx, const double* axis, int ndof, gmr_fk_t** out) {
  if (!out || !parent || !local_t || !local_r || !dof_idx || !axis) return fail(GMR_ERR_ARG, "null argument");
  if (nbody < 1 || nbody > gmr::FK_MAX_BODIES) return fail(GMR_ERR_ARG, "nbody out of range");
  if (ndof < 0) return fail(GMR_ERR_ARG, "ndof negative");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(GMR_ERR_NO_DEVICE, "no HIP device visible");
  gmr_fk* k = new (std::nothrow) gmr_fk;
  if (!k) return fail(GMR_ERR_ARG, "out of host memory");
  memset(&k->tree, 0, sizeof k->tree);
  gmr::FkTree& t = k->tree;
  t.nbody = nbody; t.ndof = ndof;
  int depth[gmr::FK_MAX_BODIES];
  int maxd = 1;
  if (parent[0] != -1) { delete k; return fail(GMR_ERR_ARG, "body 0 must be the root"); }
  depth[0] = 0;
  for (int b = 1; b < nbody; b++) {
    if (parent[b] < 0 || parent[b] >= b) { delete k; return fail(GMR_ERR_ARG, "parent[%d] invalid", b); }
    depth[b] = depth[parent[b]] + 1;
    if (depth[b] + 1 > maxd) maxd = depth[b] + 1;
  }
  if (maxd > gmr::FK_MAX_DEPTH) { delete k; return fail(GMR_ERR_ARG, "tree too deep"); }
  t.maxd = maxd;
  // LDS slots: bodies with >= 2 children are parked; a later child reloads its parent from the slot
  {
    int nchild[gmr::FK_MAX_BODIES] = {0};
    for (int b = 1; b < nbody; b++) nchild[parent[b]]++;
    int ns = 0;
    for (int b = 0; b < nbody; b++) t.save_slot[b] = (short)(nchild[b] >= 2 ? ns++ : -1);
    t.nslot = ns > 0 ? ns : 1;
    t.load_slot[0] = -1;
    for (int b = 1; b < nbody; b++) t.load_slot[b] = (short)(parent[b] == b - 1 ? -1 : t.save_slot[parent[b]]);
    for (int b = 0; b < nbody; b++) t.parent[b] = (short)(b == 0 ? 0 : parent[b]);
  }
  for (int b = 0; b < nbody; b++) {
    if (dof_idx[b] >= ndof) { delete k; return fail(GMR_ERR_ARG, "dof_idx[%d] out of range", b); }
    t.dof_idx[b] = dof_idx[b];
    if (dof_idx[b] >= 0) t.dof_body[dof_idx[b]] = (short)b;
    t.depth[b] = (short)depth[b];
    int c = b;
    for (int d = depth[b]; d >= 0; d--) { t.chain[b * maxd + d] = (short)c; c = parent[c]; }
    // normalize(axis) = axis / max(|axis|, 1e-9) in float64 (torch_utils.py:57-59), once
    double an = sqrt(axis[3 * b] * axis[3 * b] + axis[3 * b + 1] * axis[3 * b + 1] + axis[3 * b + 2] * axis[3 * b + 2]);
    if (an < 1e-9) an = 1e-9;
    for (int a = 0; a < 3; a++) { t.local_t[3 * b + a] = local_t[3 * b + a]; t.axis[3 * b + a] = axis[3 * b + a] / an; }
    for (int a = 0; a < 4; a++) t.local_r[4 * b + a] = local_r[4 * b + a];
  }
  for (int b = 0; b < nbody; b++) {
    gmr::FkBodyRec& r = t.rec[b];
    for (int a = 0; a < 3; a++) { r.t[a] = t.local_t[3 * b + a]; r.axis[a] = t.axis[3 * b + a]; }
    for (int a = 0; a < 4; a++) r.r[a] = t.local_r[4 * b + a];
    r.dof_idx = t.dof_idx[b];
    r.next_park = 0;
    r.meta = (t.dof_idx[b] >= 0 ? 1u : 0u) | ((uint32_t)(t.load_slot[b] + 1) << 8) | ((uint32_t)(t.save_slot[b] + 1) << 16) |
             ((uint32_t)t.parent[b] << 24);
    // exact zeros and ones the walk does not multiply by (fk_body): a unit local rotation, a hinge axis +-e_k, zero
    // components of the local translation
    if (!getenv("GMR_FK_NO_SPECIAL")) {
      for (int a = 0; a < 3; a++) if (r.t[a] == 0.0f) r.next_park |= 1u << (16 + a);
      if (r.r[0] == 0.0f && r.r[1] == 0.0f && r.r[2] == 0.0f && r.r[3] == 1.0f) r.meta |= 2u;
      for (int a = 0; a < 3 && t.dof_idx[b] >= 0; a++)
        if (fabs(r.axis[a]) == 1.0 && r.axis[(a + 1) % 3] == 0.0 && r.axis[(a + 2) % 3] == 0.0) {
          r.meta |= (uint32_t)(a + 1) << 2;
          r.axis[0] = r.axis[a];                // the one component the walk reads for such a hinge
          break;
        }
    }
  }
  // the split walk: per-wavefront body lists with their own records (gmr_fk_tree.h)
  {
    const int maxw = getenv("GMR_FK_WAVES") ? std::max(1, std::min(gmr::FK_MAX_WAVES, atoi(getenv("GMR_FK_WAVES")))) : gmr::FK_MAX_WAVES;
    const char* why = gmr::fk_build_split(t, parent, maxw);
    if (why) { delete k; return fail(GMR_ERR_ARG, "split walk: %s", why); }
  }
  hipError_t e;
  if ((e = hipMalloc((void**)&k->d_tree, sizeof(gmr::FkTree))) != hipSuccess ||
      (e = hipMemcpy(k->d_tree, &k->tree, sizeof(gmr::FkTree), hipMemcpyHostToDevice)) != hipSuccess) {
    if (k->d_tree) (void)hipFree(k->d_tree);
    delete k;
    return fail(GMR_ERR_HIP, "gmr_fk_create: %s", hipGetErrorString(e));
  }
  *out = k;
  return GMR_OK;
}

int gmr_fk_destroy(gmr_fk_t* k) {
  if (!k) return GMR_OK;
  (void)hipFree(k->d_tree);
  if (k->d_min_part) (void)hipFree(k->d_min_part);
  delete k;
  return GMR_OK;
}

int gmr_fk_batch_dev(gmr_fk_t* k, int B, const float* d_root_pos, const float* d_root_rot, const float* d_dof,
                     float* d_body_pos, float* d_body_rot, float* d_min_z, void* stream) {
  if (!k) return fail(GMR_ERR_ARG, "null fk handle");
  if (B < 0) return fail(GMR_ERR_ARG, "negative B");
  if (B == 0) return GMR_OK;
  if (!d_root_pos || !d_root_rot || !d_body_pos || (k->tree.ndof > 0 && !d_dof)) return fail(GMR_ERR_ARG, "null device buffer");
  if (d_min_z) {
    int blocks = gmr_fk_blocks(B);
    if (blocks > k->min_part_cap) {  // grows only; not graph-capturable on the first call of a size
      if (k->d_min_part) (void)hipFree(k->d_min_part);
      k->d_min_part = nullptr;
      HIP_TRY(hipMalloc((void**)&k->d_min_part, (size_t)blocks * sizeof(float)));
      k->min_part_cap = blocks;
    }
  }
  HIP_TRY(gmr_launch_fk_batch(k->d_tree, &k->tree, B, d_root_pos, d_root_rot, d_dof, d_body_pos,
                              d_body_rot, k->d_min_part, d_min_z, (hipStream_t)stream));
  return GMR_OK;
}

int gmr_fk_batch(gmr_fk_t* k, int B, const float* root_pos, const float* root_rot, const float* dof, float* body_pos,
                 float* body_rot, float* min_z) {
  if (!k) return fail(GMR_ERR_ARG, "null fk handle");
  if (B < 0) return fail(GMR_ERR_ARG, "negative B");
  if (B == 0) return GMR_OK;
  if (!root_pos || !root_rot || !body_pos) return fail(GMR_ERR_ARG, "null host buffer");
  const size_t nb = k->tree.nbody, nd = k->tree.ndof;
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  size_t b_rp = (size_t)B * 12, b_rr = (size_t)B * 16, b_d = (size_t)B * nd * 4, b_bp = (size_t)B * nb * 12,
         b_br = (size_t)B * nb * 16;
  size_t o_rp = 0, o_rr = o_rp + up(b_rp), o_d = o_rr + up(b_rr), o_bp = o_d + up(b_d), o_br = o_bp + up(b_bp),
         o_mz = o_br + up(b_br), total = o_mz + 256;
  char* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, total));
  int rc = GMR_OK;
  hipError_t e;
  if ((e = hipMemcpy(d + o_rp, root_pos, b_rp, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(d + o_rr, root_rot, b_rr, hipMemcpyHostToDevice)) != hipSuccess ||
      (nd && (e = hipMemcpy(d + o_d, dof, b_d, hipMemcpyHostToDevice)) != hipSuccess))
    rc = fail(GMR_ERR_HIP, "H2D copy: %s", hipGetErrorString(e));
  if (rc == GMR_OK)
    rc = gmr_fk_batch_dev(k, B, (float*)(d + o_rp), (float*)(d + o_rr), (float*)(d + o_d), (float*)(d + o_bp),
                          body_rot ? (float*)(d + o_br) : nullptr, min_z ? (float*)(d + o_mz) : nullptr, nullptr);
  if (rc == GMR_OK) {
    if ((e = hipDeviceSynchronize()) != hipSuccess ||
        (e = hipMemcpy(body_pos, d + o_bp, b_bp, hipMemcpyDeviceToHost)) != hipSuccess ||
        (body_rot && (e = hipMemcpy(body_rot, d + o_br, b_br, hipMemcpyDeviceToHost)) != hipSuccess) ||
        (min_z && (e = hipMemcpy(min_z, d + o_mz, 4, hipMemcpyDeviceToHost)) != hipSuccess))
      rc = fail(GMR_ERR_HIP, "kernel / D2H copy: %s", hipGetErrorString(e));
  }
  (void)hipFree(d);
  return rc;
}

int gmr_fk_segment_min_z_dev(gmr_fk_t* k, const float* d_body_pos, const int32_t* d_seg_start, int nseg, float* d_seg_min,
                             void* stream) {
  if (!k) return fail(GMR_ERR_ARG, "null fk handle");
  if (nseg < 0) return fail(GMR_ERR_ARG, "negative nseg");
  if (nseg == 0) return GMR_OK;
  if (!d_body_pos || !d_seg_start || !d_seg_min) return fail(GMR_ERR_ARG, "null device buffer");
  HIP_TRY(gmr_launch_fk_segment_min(d_body_pos, k->tree.nbody, d_seg_start, nseg, d_seg_min, (hipStream_t)stream));
  return GMR_OK;
}

// Many clips in one launch: frames of clip g are rows [seg_start[g], seg_start[g + 1]) of the inputs.
int gmr_fk_batch_segments(gmr_fk_t* k, int B, const float* root_pos, const float* root_rot, const float* dof, int nseg,
                          const int32_t* seg_start, float* body_pos, float* seg_min_z) {
  if (!k) return fail(GMR_ERR_ARG, "null fk handle");
  if (B < 0 || nseg < 0) return fail(GMR_ERR_ARG, "negative B / nseg");
  if (B == 0) { for (int g = 0; g < nseg && seg_min_z; g++) seg_min_z[g] = INFINITY; return GMR_OK; }
  if (!root_pos || !root_rot || (k->tree.ndof > 0 && !dof)) return fail(GMR_ERR_ARG, "null host buffer");
  if (nseg > 0 && (!seg_start || !seg_min_z)) return fail(GMR_ERR_ARG, "null segment buffers");
  for (int g = 0; g < nseg; g++)
    if (seg_start[g] < 0 || seg_start[g] > seg_start[g + 1] || seg_start[g + 1] > B) return fail(GMR_ERR_ARG, "seg_start must ascend within [0, B]");
  const size_t nb = k->tree.nbody, nd = k->tree.ndof;
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t b_rp = (size_t)B * 12, b_rr = (size_t)B * 16, b_d = (size_t)B * nd * 4, b_bp = (size_t)B * nb * 12,
               b_ss = (size_t)(nseg + 1) * 4, b_sm = (size_t)nseg * 4;
  const size_t o_rp = 0, o_rr = o_rp + up(b_rp), o_d = o_rr + up(b_rr), o_bp = o_d + up(b_d), o_ss = o_bp + up(b_bp),
               o_sm = o_ss + up(b_ss), total = o_sm + up(b_sm) + 256;
  char* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, total));
  int rc = GMR_OK;
  hipError_t e;
  if ((e = hipMemcpy(d + o_rp, root_pos, b_rp, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(d + o_rr, root_rot, b_rr, hipMemcpyHostToDevice)) != hipSuccess ||
      (nd && (e = hipMemcpy(d + o_d, dof, b_d, hipMemcpyHostToDevice)) != hipSuccess) ||
      (nseg && (e = hipMemcpy(d + o_ss, seg_start, b_ss, hipMemcpyHostToDevice)) != hipSuccess))
    rc = fail(GMR_ERR_HIP, "H2D copy: %s", hipGetErrorString(e));
  if (rc == GMR_OK)
    rc = gmr_fk_batch_dev(k, B, (float*)(d + o_rp), (float*)(d + o_rr), (float*)(d + o_d), (float*)(d + o_bp), nullptr, nullptr, nullptr);
  if (rc == GMR_OK && nseg)
    rc = gmr_fk_segment_min_z_dev(k, (float*)(d + o_bp), (int32_t*)(d + o_ss), nseg, (float*)(d + o_sm), nullptr);
  if (rc == GMR_OK) {
    if ((e = hipDeviceSynchronize()) != hipSuccess ||
        (body_pos && (e = hipMemcpy(body_pos, d + o_bp, b_bp, hipMemcpyDeviceToHost)) != hipSuccess) ||
        (nseg && (e = hipMemcpy(seg_min_z, d + o_sm, b_sm, hipMemcpyDeviceToHost)) != hipSuccess))
      rc = fail(GMR_ERR_HIP, "kernel / D2H copy: %s", hipGetErrorString(e));
  }
  (void)hipFree(d);
  return rc;
}

}  // extern "C"
