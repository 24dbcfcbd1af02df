/*
 * gmr_hip.h -- C-ABI of libgmrhip.so: the MI355X (gfx950) implementation of GMR's retargeting
 * hot path.  Plain C, plain pointers and sizes, no torch / no C++ types: this is exactly what a
 * ctypes (or cffi / cgo / JNI) binding in the reference would bind.  INTEGRATION.md shows the
 * reference-side stub.
 *
 * Conventions: every function returns 0 on success and a negative code on error
 * (gmr_last_error() returns a thread-local message); handles are immutable after creation, so
 * batch calls on different HIP streams are re-entrant.  "dev" entry points take device pointers
 * and a hipStream_t (as void*; NULL = the default stream) and never synchronise; the entry points
 * without the suffix take host pointers, copy, launch and synchronise; they stage through a per-handle
 * device workspace, so at most one host-pointer call per handle at a time (like the reference object,
 * which is not thread-safe either: SURVEY.md section 8b).
 *
 * Row IDs (H1..H10) refer to SURVEY.md section 8(a).
 */
#ifndef GMR_HIP_H
#define GMR_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "gmr_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GMR_OK 0
#define GMR_ERR_ARG (-1)     /* bad argument / bad blob                                   */
#define GMR_ERR_HIP (-2)     /* a HIP runtime call failed                                  */
#define GMR_ERR_NO_DEVICE (-3)
#define GMR_ERR_COMM (-4)    /* RCCL / bootstrap failure                                   */

/* retarget flags */
#define GMR_FLAG_OFFSET_TO_GROUND 1 /* retarget(human_data, offset_to_ground=True), motion_retarget.py:139 */
#define GMR_FLAG_EVAL_ONLY 2        /* update_targets() without solve: preprocess + residual norms only, q_out = q in
                                       (motion_retarget.py:117-136 followed by error1()/error2(), :188-200)          */

/* per-stream status written by the IK kernel */
#define GMR_STATUS_OK 0
#define GMR_STATUS_QP_FAILED (-1)   /* Cholesky breakdown / non-finite input; mink would raise (section 8b) */
#define GMR_STATUS_QP_MAXITER (-2)  /* active-set iteration cap hit                                         */

typedef struct gmr_solver gmr_solver_t; /* device-resident (model, task set)                    */
typedef struct gmr_fk gmr_fk_t;         /* device-resident KinematicsModel tree (float32 path)  */

/* ---- library / device ------------------------------------------------------------------- */
const char* gmr_last_error(void);
const char* gmr_backend_info(void);     /* "hip:gfx950 ..." ; replaces nothing, diagnostic      */
int gmr_device_count(void);
int gmr_set_device(int device);
size_t gmr_sizeof_model(void);          /* ABI check against the Python-side struct layouts      */
size_t gmr_sizeof_taskset(void);

/* ---- device memory / streams / events (so that the Python host needs no torch) ----------- */
int gmr_malloc(void** ptr, size_t bytes);
int gmr_free(void* ptr);
int gmr_memset(void* ptr, int value, size_t bytes, void* stream);
int gmr_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int gmr_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int gmr_stream_create(void** stream);
int gmr_stream_destroy(void* stream);
int gmr_stream_sync(void* stream);      /* NULL = device synchronize                             */
int gmr_event_create(void** event);
int gmr_event_destroy(void* event);
int gmr_event_record(void* event, void* stream);
int gmr_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop`          */

/* ---- H1: solver state ------------------------------------------------------------------- */
/* Replaces GeneralMotionRetargeting.__init__ + setup_retarget_configuration
 * (motion_retarget.py:13-114): uploads the packed robot model and task set.  The blobs are the
 * same bytes rank 0 broadcasts to its peers. */
int gmr_solver_create(const gmr_model_t* model, const gmr_taskset_t* taskset, gmr_solver_t** out);
int gmr_solver_destroy(gmr_solver_t* solver);
int gmr_solver_dims(const gmr_solver_t* solver, int* nq, int* nv, int* nhuman);
/* Launch shape of the IK kernel: 1 = one wavefront per stream (most streams resident), 4 = one main
 * wavefront + 3 helper wavefronts per stream (shortest per-frame latency), 0 = automatic (4 up to 300
 * streams per launch, 1 above).  The 4-wavefront shape needs a robot whose dofs split into <= 4 limbs of <= 8
 * and a trunk of <= 10 (all shipped robots do); other robots always run the 1-wavefront shape.  Results agree
 * to rounding between the shapes; no reference analogue. */
int gmr_solver_set_waves(gmr_solver_t* solver, int waves_per_stream);
/* Dispatch of the 1-wavefront shape when streams outnumber the GPU's resident wavefronts (9 per CU):
 * frames_per_item > 0 = a resident set of wavefronts serves (stream, frames_per_item frames) items from a device-side
 * FIFO, so all streams advance together and the launch does not end on a few late, long streams (default 4);
 * 0 = one workgroup per stream for all of its frames.  Results are bit-identical either way.  The reference's analogue
 * is the chunking of `mp.Pool.map` over files (scripts/smplx_to_robot_dataset.py:241-242). */
int gmr_solver_set_dispatch(gmr_solver_t* solver, int frames_per_item);

/* ---- H2-H7: the retargeting loop ---------------------------------------------------------- */
/* Replaces the caller loop `for frame in frames: qpos = retargeter.retarget(frame)`
 * (scripts/smplx_to_robot_dataset.py:85-87) around GeneralMotionRetargeting.retarget
 * (motion_retarget.py:139-185) for S independent streams of up to T frames each, with the time
 * loop on the device (frames of one stream are sequentially dependent: warm start, :75).
 *
 *   q0      f64 [S][nq]            configuration before the first frame (qpos0 for a fresh object)
 *   human   f64 [S][T][nhuman][7]  raw human_data of the bodies of the scale table, packed order
 *                                  (pos xyz, quat wxyz); a body absent from the caller's dict is
 *                                  encoded with pos[0] = NaN
 *   len     i32 [S] or NULL        frames of stream s (<= T); NULL = all T
 *   q_out   f64 [S][T][nq]         qpos after each frame (what retarget() returns, :185)
 *   nsolve  i32 [S][T][2]          solve_ik calls per stage (1 + loop iterations, :147-161)
 *   status  i32 [S]                GMR_STATUS_*
 *   tgt_out f64 [S][T][nhuman][7]  or NULL: the preprocessed targets of every frame, i.e. `scaled_human_data`
 *                                  = the poses handed to task.set_target (:117-136, :203-270), as the kernel
 *                                  computed them (absent bodies stay NaN rows)
 *   err_out f64 [S][T][2]          or NULL: error1() / error2() (:188-200) at the configuration each frame
 *                                  ends with (0 for a stage the config does not use)
 * Rows the kernel does not write: frames at or beyond len[s]; and, for a stream whose status is not GMR_STATUS_OK,
 * the tgt_out rows of the frames AFTER the failing one and the err_out rows from the failing frame on (q_out keeps the
 * last good configuration for those frames, nsolve 0).  The host entry point returns such rows as zeros; the device
 * entry point leaves the caller's memory untouched there -- clear tgt_out / err_out first if stale bytes matter.
 */
int gmr_retarget_streams_dev(gmr_solver_t* solver, int S, int T, const double* d_q0, const double* d_human,
                             const int32_t* d_len, int flags, double* d_q_out, int32_t* d_nsolve,
                             int32_t* d_status, double* d_tgt_out, double* d_err_out, void* stream);
int gmr_retarget_streams(gmr_solver_t* solver, int S, int T, const double* q0, const double* human,
                         const int32_t* len, int flags, double* q_out, int32_t* nsolve, int32_t* status,
                         double* tgt_out, double* err_out);
/* ---- several (robot, task set) jobs as ONE scheduling domain -------------------------------------------------------
 * BASELINE.json configs[3] ("all 6 robots mixed-DoF batch"; SURVEY.md section 8d: "per-robot kernels or one kernel
 * with per-stream model index"): the reference would run one mp.Pool worker per file whatever its robot
 * (scripts/smplx_to_robot_dataset.py:241-242).  Every robot of the throughput kernel's size class runs the same
 * kernel instance, so the jobs of a group share one resident grid and one device-side queue of (job, stream, chunk)
 * items: the group is balanced as a whole and its launch ends within one chunk of its last stream.  Results are
 * bit-identical to launching every job by itself.  Jobs that do not take the throughput shape (a robot that does not
 * decompose, a solver forced to 4 wavefronts, a group of <= 300 streams) are launched one by one on the same stream. */
typedef struct gmr_job {
  gmr_solver_t* solver;
  int32_t S, T;
  const double* q0;        /* [S][nq]              the buffers of gmr_retarget_streams, per job                     */
  const double* human;     /* [S][T][nhuman][7]                                                                      */
  const int32_t* len;      /* [S] or NULL                                                                            */
  double* q_out;           /* [S][T][nq]                                                                             */
  int32_t* nsolve;         /* [S][T][2]                                                                              */
  int32_t* status;         /* [S]                                                                                    */
  double* tgt_out;         /* [S][T][nhuman][7] or NULL                                                              */
  double* err_out;         /* [S][T][2] or NULL                                                                      */
} gmr_job_t;
/* device pointers, asynchronous on `stream` */
int gmr_retarget_group_dev(const gmr_job_t* jobs, int njobs, int flags, void* stream);
/* device pointers, ONE WINDOW of every stream's frames, [t_begin, t_end): the windows of a batch are launched in order on one
 * stream starting at t_begin = 0; q continues from the previous window's last q_out row, the QP's bound sets and the status
 * travel in device memory.  Bit-identical to one launch.  Only for batches that take the throughput shape as a whole (every
 * robot decomposes, more than 300 streams in total, at most 8 jobs); GMR_ERR_ARG otherwise. */
int gmr_retarget_group_window_dev(const gmr_job_t* jobs, int njobs, int flags, int t_begin, int t_end, void* stream);
/* HOST pointers: the streams are cut into `slices` slices (0 = automatic: about 64 MB of input each, never fewer than
 * 4 096 streams per slice, at most 16; > 0: exactly that many, at most one per stream) whose H2D copies, launch and D2H copies go to one of four HIP streams, so that
 * copy(k+1) || kernel(k) || copy-back(k-1); synchronises before returning.  A batch too narrow for that (fewer than 8 192
 * streams) but long (>= 32 frames, >= 64 MB of input) is cut in TIME instead: consecutive windows of frames
 * (gmr_retarget_group_window_dev), the strided copies of window w + 1 under the kernel of window w; slices < 0 asks for
 * |slices| windows explicitly.  Use pinned host memory (below) for the
 * copies to be asynchronous.  gmr_retarget_streams takes this path by itself for inputs of 32 MB and more. */
int gmr_retarget_group(const gmr_job_t* jobs, int njobs, int flags, int slices);
/* pinned (page-locked) host memory for the host-pointer entry points; gmr_host_register pins a caller's own buffer */
int gmr_host_alloc(void** ptr, size_t bytes);
int gmr_host_free(void* ptr);
int gmr_host_register(void* ptr, size_t bytes);
int gmr_host_unregister(void* ptr);

/* LDS bytes per stream of the IK kernel for this solver (occupancy reporting). */
int gmr_retarget_lds_bytes(const gmr_solver_t* solver);

/* ---- H8-H9: post-hoc batched FK (float32) ------------------------------------------------- */
/* Replaces KinematicsModel(xml, device="cuda:0") + forward_kinematics
 * (kinematics_model.py:69-170, 213-246), i.e. ~2.6k ATen launches per call -> one kernel.
 * Tree arrays are those of the reference's own XML reader (mjcf.parse_kinematics_tree):
 * local_r xyzw un-normalised, axis f64 per body (zeros where dof_idx < 0). */
int gmr_fk_create(int nbody, const int32_t* parent, const float* local_t, const float* local_r,
                  const int32_t* dof_idx, const double* axis, int ndof, gmr_fk_t** out);
int gmr_fk_destroy(gmr_fk_t* fk);
/*   root_pos f32 [B][3], root_rot f32 [B][4] xyzw, dof f32 [B][ndof]
 *   body_pos f32 [B][nbody][3], body_rot f32 [B][nbody][4] (may be NULL)
 *   min_z    f32 [1] (may be NULL): min over all frames and bodies of body_pos z -- the reduction
 *            of the dataset scripts' height adjustment (smplx_to_robot_dataset.py:118-126)        */
int gmr_fk_batch_dev(gmr_fk_t* fk, int B, const float* d_root_pos, const float* d_root_rot, const float* d_dof,
                     float* d_body_pos, float* d_body_rot, float* d_min_z, void* stream);
int gmr_fk_batch(gmr_fk_t* fk, int B, const float* root_pos, const float* root_rot, const float* dof,
                 float* body_pos, float* body_rot, float* min_z);

/* Many clips in ONE launch (the dataset drivers): the frames of clip g are rows [seg_start[g], seg_start[g + 1]) of the
 * inputs; seg_min_z[g] = min over the clip's frames and bodies of body_pos z -- the per-clip reduction of the height
 * adjustment (smplx_to_robot_dataset.py:118-126), +inf for an empty clip.  body_pos may be NULL (only the minima wanted). */
int gmr_fk_segment_min_z_dev(gmr_fk_t* fk, const float* d_body_pos, const int32_t* d_seg_start, int nseg, float* d_seg_min,
                             void* stream);
int gmr_fk_batch_segments(gmr_fk_t* fk, int B, const float* root_pos, const float* root_rot, const float* dof, int nseg,
                          const int32_t* seg_start /* [nseg + 1] */, float* body_pos, float* seg_min_z /* [nseg] */);

/* ---- N1: SMPL-X frame extraction (the step in front of the loop) ---------- */
/* A kinematic tree of J <= 64 joints (parents[j] < j, joint 0 the root) and the joints whose poses are
 * wanted: sel[nsel] (one output row each, in this order; nsel = 0 -> all J joints, row = joint).  For the
 * retargeting loop sel lists the SMPL-X joints of the solver's packed human bodies, so that the output IS
 * the `human` argument of gmr_retarget_streams. */
typedef struct gmr_smplx gmr_smplx_t;
int gmr_smplx_create(int J, const int32_t* parents, int nsel, const int32_t* sel, gmr_smplx_t** out);
int gmr_smplx_destroy(gmr_smplx_t* h);
int gmr_smplx_rows(const gmr_smplx_t* h);   /* rows per output frame (nsel, or J) */
/* Replaces the joints of `body_model(betas, global_orient, body_pose, transl, ...)` as called at
 * general_motion_retargeting/utils/smpl.py:12-34, without the mesh: joints f32[N][J][3] from the rest
 * joints j_rest f64[J][3] (J_regressor applied to the shaped template, once per clip), the axis-angle
 * poses full_pose f32[N][J][3] and transl f32[N][3].  (The body model is third-party: parity unpinned.) */
int gmr_smplx_joints_dev(gmr_smplx_t* h, int N, const double* d_j_rest, const float* d_full_pose,
                         const float* d_transl, float* d_joints, void* stream);
int gmr_smplx_joints(gmr_smplx_t* h, int N, const double* j_rest, const float* full_pose, const float* transl,
                     float* joints);
/* Replaces get_smplx_data_offline_fast (utils/smpl.py:109-197) for one clip -- and get_smplx_data
 * (:44-73) when target_time is NULL (then Nout == N, no interpolation):
 *   full_pose f32[N][J][3], joints f32[N][jstride][3] (jstride >= J: the model appends landmark joints),
 *   target_time f64[Nout] = np.linspace(0, N-1, Nout), Nout = N // int(src_fps / tgt_fps) (:120-127)
 *   out f64[Nout][rows][7] = position xyz, global orientation quaternion wxyz per selected joint. */
int gmr_smplx_align_dev(gmr_smplx_t* h, int N, int jstride, const float* d_full_pose, const float* d_joints,
                        int Nout, const double* d_target_time, double* d_out, void* stream);
int gmr_smplx_align(gmr_smplx_t* h, int N, int jstride, const float* full_pose, const float* joints, int Nout,
                    const double* target_time, double* out);
/* Both steps for one clip in one call, host buffers: the body model's joints (all J) and the alignment of the selected rows,
 * nothing but out f64[Nout][rows][7] coming back (no joints on the host, no gather between the steps).  Replaces
 * load_smplx_file's forward pass + get_smplx_data_offline_fast (utils/smpl.py:12-41, :109-197) for a dataset driver that
 * needs only the packed frames.  target_time == NULL: no fps alignment, Nout == N. */
int gmr_smplx_frames(gmr_smplx_t* h, int N, const double* j_rest, const float* full_pose, const float* transl, int Nout,
                     const double* target_time, double* out);
/* The same on COMPACT inputs -- only what the alignment reads, FRAME-MINOR: pose_c f32[npose][3][N] = the axis-angle poses
 * of the selection's ancestor closure in walk order, joints_c f32[nrow][3][N] = the joint of every output row (numpy:
 * full_pose[:, pose_joints].transpose(1, 2, 0)); the two joint lists come from gmr_smplx_compact_layout (either output may
 * be NULL).  The host entry point gmr_smplx_align gathers these itself, so only 34 of the 110 joint triples of a G1 frame
 * cross the bus, and a wavefront's load of one component reads one contiguous run of source frames. */
int gmr_smplx_compact_layout(const gmr_smplx_t* h, int32_t* pose_joints, int* npose, int32_t* row_joints, int* nrow);
int gmr_smplx_align_compact_dev(gmr_smplx_t* h, int N, const float* d_pose_c, const float* d_joints_c, int Nout,
                                const double* d_target_time, double* d_out, void* stream);

/* ---- multi-GPU: one rank per GPU, ONE broadcast, no per-step collective ------------ */
/* The reference parallelises over files with mp.Pool on one CPU (scripts/smplx_to_robot_dataset.py:241-242); here
 * streams shard over the ranks of one node and the only data that crosses ranks is the packed robot model + task set.
 * RCCL is opened at run time (dlopen of librccl.so; GMR_RCCL_LIBRARY overrides); no PyTorch involved. */
typedef struct gmr_comm gmr_comm_t;
/* The ranks form a control star over TCP at master_addr:port (the launcher's MASTER_ADDR and a port derived from
 * MASTER_PORT); rank 0's ncclUniqueId travels over it and every step of the RCCL bring-up is agreed on by ALL ranks,
 * so a failure on one rank is the same error (GMR_ERR_COMM, gmr_last_error names the rank and the reason) on every
 * rank -- never a hang in a half-formed communicator.  Call after gmr_set_device(local_rank).
 * Environment: GMR_COMM_BACKEND=tcp -- the star alone carries the (job-level, host-buffer) operations below: the CPU
 * rehearsal of the N > 1 path; GMR_COMM_FALLBACK=tcp -- a job whose RCCL bring-up fails continues on the star, decided
 * collectively, and gmr_comm_backend() says so; GMR_COMM_TIMEOUT (s, default 120) bounds the rendezvous. */
int gmr_comm_create(int rank, int world, const char* master_addr, int port, gmr_comm_t** out);
/* "rccl-<version>", "tcp" or "tcp (fallback: <why RCCL could not be brought up>)"; valid until gmr_comm_destroy */
const char* gmr_comm_backend(const gmr_comm_t* comm);
int gmr_comm_destroy(gmr_comm_t* comm);
int gmr_comm_rank(const gmr_comm_t* comm);
int gmr_comm_world(const gmr_comm_t* comm);
/* `gmr_broadcast_model` of SURVEY.md section 8(b): `bytes` host bytes (gmr_model_t + gmr_taskset_t, 24 KB) of rank
 * `root` into the same buffer on every rank (H2D, ncclBroadcast over xGMI, D2H, synchronised). */
int gmr_comm_broadcast(gmr_comm_t* comm, void* buf, size_t bytes, int root);
int gmr_comm_broadcast_dev(gmr_comm_t* comm, void* d_buf, size_t bytes, int root, void* stream);
/* job-level plumbing of the drivers (not in the data path): device-synchronising barrier, timing reductions */
int gmr_comm_barrier(gmr_comm_t* comm);
int gmr_comm_allreduce_max(gmr_comm_t* comm, double* inout, int n);
int gmr_comm_allreduce_sum(gmr_comm_t* comm, double* inout, int n);
int gmr_comm_allgather(gmr_comm_t* comm, const double* in, double* out /* [world][n] */, int n);
/* the bootstrap alone (plain TCP, no GPU): rank 0 hands `bytes` bytes to every peer */
int gmr_bootstrap_exchange(int rank, int world, const char* addr, int port, void* payload, size_t bytes, double timeout_s);

#ifdef __cplusplus
}
#endif
#endif /* GMR_HIP_H */
