/*
 * gmr_types.h -- flat, pointer-free descriptions of a robot model and of an IK task set.
 *
 * These two PODs are what crosses the C-ABI (include/gmr_hip.h), what rank 0 broadcasts to the
 * other ranks (one RCCL broadcast of sizeof(gmr_model_t)+sizeof(gmr_taskset_t) bytes) and what the
 * HIP kernels stage into LDS.  They replace, for the retargeting hot path, the state the reference
 * builds in GeneralMotionRetargeting.__init__ / setup_retarget_configuration
 * (reference general_motion_retargeting/motion_retarget.py:13-114): the compiled MuJoCo model
 * (mj.MjModel.from_xml_path, :27), the two lists of mink.FrameTask (:80-114) and the
 * scale / offset tables (:42-54, :90-94).
 *
 * Layout rules: int32 fields first, then doubles; fixed-size arrays; no pointers; no padding
 * surprises (every double array starts on an 8-byte boundary because the int32 block has an even
 * number of entries -- checked by static_asserts in the implementation and by
 * gmr_sizeof_model()/gmr_sizeof_taskset() against the Python-side numpy dtypes).
 */
#ifndef GMR_TYPES_H
#define GMR_TYPES_H

#include <stdint.h>

#define GMR_MAGIC_MODEL   0x474d524d /* "GMRM" */
#define GMR_MAGIC_TASKSET 0x474d5254 /* "GMRT" */
#define GMR_ABI_VERSION   1

#define GMR_MAX_BODIES 48  /* robot bodies, floating base = body 0 (G1: 38)            */
#define GMR_MAX_HINGES 40  /* hinge joints (G1: 29)                                     */
#define GMR_MAX_DOF    46  /* nv = 6 + nhinge                                           */
#define GMR_MAX_NQ     47  /* nq = 7 + nhinge                                           */
#define GMR_MAX_DEPTH  20  /* bodies on a root->leaf path, root included (G1: 13)       */
#define GMR_MAX_TASKS  16  /* frame tasks per stage (Hi: 15)                            */
#define GMR_MAX_HUMAN  16  /* human bodies consumed per frame (G1: 14)                  */
#define GMR_MAX_PAIRS  384 /* (task, ancestor dof) pairs per stage (G1: 157)            */

#ifdef __cplusplus
extern "C" {
#endif

/* Kinematic subset of a compiled MJCF, MuJoCo conventions, quaternions wxyz. */
typedef struct gmr_model_t {
  int32_t magic, version;
  int32_t nbody, nhinge, nq, nv;
  int32_t parent[GMR_MAX_BODIES];        /* -1 for body 0                                  */
  int32_t depth[GMR_MAX_BODIES];         /* 0 for body 0                                   */
  int32_t body_hinge[GMR_MAX_BODIES];    /* hinge index carried by this body, or -1        */
  int32_t hinge_body[GMR_MAX_HINGES];
  int32_t limited[GMR_MAX_HINGES];
  /* chain[b][d], d = 0..depth[b]: the bodies on the path root -> b (chain[b][depth[b]] == b) */
  int32_t chain[GMR_MAX_BODIES][GMR_MAX_DEPTH];
  double timestep;
  double body_pos[GMR_MAX_BODIES][3];
  double body_quat[GMR_MAX_BODIES][4];   /* normalised                                     */
  double hinge_axis[GMR_MAX_HINGES][3];  /* body-local, normalised                         */
  double range_lo[GMR_MAX_HINGES];
  double range_hi[GMR_MAX_HINGES];
  double qpos0[GMR_MAX_NQ + 1];
} gmr_model_t;

/* The two-stage frame-task set plus the per-frame target preprocessing tables. */
typedef struct gmr_taskset_t {
  int32_t magic, version;
  int32_t nhuman;                         /* human bodies in the packed input, <= GMR_MAX_HUMAN   */
  int32_t human_root;                     /* index of human_root_name in that list                 */
  int32_t max_iter;                       /* motion_retarget.py:56                                 */
  int32_t _pad0;
  int32_t use_stage[2];                   /* use_ik_match_table1 / 2                               */
  int32_t ntask[2];
  int32_t npair[2];
  int32_t is_foot[GMR_MAX_HUMAN];         /* name contains "foot"/"Foot" (motion_retarget.py:260)  */
  int32_t task_body[2][GMR_MAX_TASKS];    /* robot body index of the task frame                    */
  int32_t task_human[2][GMR_MAX_TASKS];   /* index into the packed human list                      */
  int32_t task_col0[2][GMR_MAX_TASKS];    /* first pair of the task                                */
  int32_t task_ncol[2][GMR_MAX_TASKS];    /* dofs listed for the task: base (6, or 3 without a position cost) + hinges root -> body */
  int32_t pair_task[2][GMR_MAX_PAIRS];
  int32_t pair_dof[2][GMR_MAX_PAIRS];     /* ascending within a task                               */
  int32_t pair_index[2][GMR_MAX_TASKS][GMR_MAX_DOF]; /* pair id of (task, dof) or -1            */
  double damping;                         /* solve_ik damping, motion_retarget.py:19               */
  double lm_damping;                      /* FrameTask lm_damping, :88                             */
  double tol;                             /* stop threshold, :153                                  */
  double limit_gain;                      /* mink ConfigurationLimit gain (0.95)                   */
  double ground_offset;                   /* offset_human_data_to_ground, :255                     */
  double w_pos[2][GMR_MAX_TASKS];
  double w_rot[2][GMR_MAX_TASKS];
  double scale[GMR_MAX_HUMAN];            /* human_scale_table * height ratio, :36-43              */
  double pos_off[GMR_MAX_HUMAN][3];       /* table-1 offset minus ground, :91                      */
  double quat_off[GMR_MAX_HUMAN][4];      /* table-1 rotation offset, wxyz, normalised, :92-94     */
} gmr_taskset_t;

#ifdef __cplusplus
}
#endif
#endif /* GMR_TYPES_H */
