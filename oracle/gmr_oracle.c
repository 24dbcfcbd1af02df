/*
 * gmr_oracle.c -- CPU restatement (plain C, FP64) of GMR's per-frame IK retargeting path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY STATUS:
 *   - preprocessing (orc_preprocess) and the float32 post-hoc FK (orc_fk_f32) are PINNED by golden
 *     vectors generated in the build container from the reference's own importable Python
 *     (tests/golden/make_golden.py -> tests/golden/ npz files);
 *   - the IK numerics (FK, frame Jacobian, SE(3) log / Jlog, QP assembly, box-QP, integration;
 *     rows H4-H7) live in third-party libraries that are neither vendored in the reference nor
 *     installed here: mink, mujoco, qpsolvers, daqp -- all UNPINNED in the reference
 *     (requirements.txt:7-9, setup.py:16-20; daqp is not declared at all).  Their published
 *     algorithms are restated below from the reference's call sites
 *     (general_motion_retargeting/motion_retarget.py:74-200) and SURVEY.md Appendix A.
 *     For these rows: "parity unpinned".  They are cross-checked by independent properties
 *     (finite-difference Jacobians, KKT residuals, scipy box-QP, known-answer IK) in tests/.
 *
 * Conventions: quaternions wxyz; SE(3) tangent order [v(3); w(3)]; qpos = [xyz, wxyz, hinges].
 * Every function cites the reference line (or the Appendix-A item) it follows.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/gmr_types.h"

#define NB GMR_MAX_BODIES
#define NV GMR_MAX_DOF
#define NK GMR_MAX_TASKS
#define NHUM GMR_MAX_HUMAN

/* ------------------------------------------------------------------------------------------ */
/* small algebra                                                                               */
/* ------------------------------------------------------------------------------------------ */
static void quat_mul(double r[4], const double a[4], const double b[4]) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}

/* mju_normalize4: leaves near-unit quaternions untouched, resets degenerate ones. */
static void quat_normalize_mj(double q[4]) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < 1e-15) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else if (fabs(n - 1.0) > 1e-15) { double s = 1.0 / n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
}

/* mju_quat2Mat */
static void quat_to_mat(double m[9], const double q[4]) {
  double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
  double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  double q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02);
  m[3] = 2 * (q12 + q03); m[5] = 2 * (q23 - q01);
  m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01);
}

static void mat_vec(double r[3], const double m[9], const double v[3]) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void matT_vec(double r[3], const double m[9], const double v[3]) {
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  double y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  double z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void cross3(double r[3], const double a[3], const double b[3]) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static void skew(double K[9], const double w[3]) {
  K[0] = 0; K[1] = -w[2]; K[2] = w[1];
  K[3] = w[2]; K[4] = 0; K[5] = -w[0];
  K[6] = -w[1]; K[7] = w[0]; K[8] = 0;
}
static void mat3_mul(double C[9], const double A[9], const double B[9]) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, T, sizeof T);
}

/* mju_axisAngle2Quat */
static void axis_angle_quat(double q[4], const double axis[3], double angle) {
  if (angle == 0.0) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  double s = sin(0.5 * angle);
  q[0] = cos(0.5 * angle); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}

/* ------------------------------------------------------------------------------------------ */
/* H2: target preprocessing  (reference motion_retarget.py:117-124, 203-270)                   */
/* ------------------------------------------------------------------------------------------ */
/* scipy Rotation.from_quat(scalar_first=True) normalises; `*` composes and normalises again. */
static void quat_unit(double q[4]) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/*
 * in : human[nhuman][7]  raw (pos xyz, quat wxyz) of the bodies of the scale table, packed order.
 *      A body whose pos[0] is NaN is "absent from the caller's dict" and is skipped like the
 *      reference skips it (scale_human_data iterates the input dict, :218).
 * out: tgt[nhuman][7]    scaled + offset (+ optionally grounded) targets.
 */
void orc_preprocess(const gmr_taskset_t* ts, const double* human, int offset_to_ground, double* tgt) {
  int nh = ts->nhuman, root = ts->human_root;
  const double* rp = human + 7 * root;
  /* scale_human_data, :209-232 */
  double srp[3] = {ts->scale[root] * rp[0], ts->scale[root] * rp[1], ts->scale[root] * rp[2]};
  for (int b = 0; b < nh; b++) {
    const double* in = human + 7 * b;
    double* o = tgt + 7 * b;
    if (b == root) { o[0] = srp[0]; o[1] = srp[1]; o[2] = srp[2]; }
    else for (int a = 0; a < 3; a++) o[a] = (in[a] - rp[a]) * ts->scale[b] + srp[a];
    for (int a = 0; a < 4; a++) o[3 + a] = in[3 + a];
  }
  /* offset_human_data, :234-250 (table-1 offsets only, :121) */
  for (int b = 0; b < nh; b++) {
    double* o = tgt + 7 * b;
    if (o[0] != o[0]) continue;
    double q[4] = {o[3], o[4], o[5], o[6]}, qo[4], uq[4], m[9], g[3];
    quat_unit(q);
    memcpy(qo, ts->quat_off[b], sizeof qo);
    quat_unit(qo);
    quat_mul(uq, q, qo);
    quat_unit(uq);
    quat_to_mat(m, uq);
    mat_vec(g, m, ts->pos_off[b]);
    o[0] += g[0]; o[1] += g[1]; o[2] += g[2];
    o[3] = uq[0]; o[4] = uq[1]; o[5] = uq[2]; o[6] = uq[3];
  }
  /* offset_human_data_to_ground, :252-270 */
  if (offset_to_ground) {
    double lowest = INFINITY;
    for (int b = 0; b < nh; b++) {
      const double* o = tgt + 7 * b;
      if (!ts->is_foot[b] || o[0] != o[0]) continue;
      if (o[2] < lowest) lowest = o[2];
    }
    for (int b = 0; b < nh; b++) tgt[7 * b + 2] = tgt[7 * b + 2] - lowest + ts->ground_offset;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* H6 (second half): forward kinematics, mj_kinematics semantics              */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  double xpos[NB][3], xquat[NB][4], xmat[NB][9];
  double xaxis[NB][3]; /* world hinge axis of the hinge carried by body b */
} orc_fk_t;

void orc_fk(const gmr_model_t* m, double* q, orc_fk_t* k) {
  /* free joint: position and (normalised) orientation straight from qpos */
  quat_normalize_mj(q + 3);
  memcpy(k->xpos[0], q, 3 * sizeof(double));
  memcpy(k->xquat[0], q + 3, 4 * sizeof(double));
  quat_to_mat(k->xmat[0], k->xquat[0]);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->parent[b];
    double v[3], quat[4];
    mat_vec(v, k->xmat[p], m->body_pos[b]);
    for (int a = 0; a < 3; a++) k->xpos[b][a] = k->xpos[p][a] + v[a];
    quat_mul(quat, k->xquat[p], m->body_quat[b]);
    int h = m->body_hinge[b];
    if (h >= 0) {
      double mm[9], qloc[4];
      quat_to_mat(mm, quat);
      mat_vec(k->xaxis[b], mm, m->hinge_axis[h]);        /* xaxis before the joint rotation */
      axis_angle_quat(qloc, m->hinge_axis[h], q[7 + h]); /* qpos0 of hinges is 0 (no `ref`)   */
      quat_mul(quat, quat, qloc);
      /* joint pos == 0 for every supported robot => no off-centre correction, anchor == xpos */
    }
    quat_normalize_mj(quat);
    memcpy(k->xquat[b], quat, sizeof quat);
    quat_to_mat(k->xmat[b], quat);
  }
}

/* flat copy for tests: xpos[nbody][3], xquat[nbody][4] */
void orc_fk_flat(const gmr_model_t* m, const double* q_in, double* xpos, double* xquat) {
  orc_fk_t k;
  double q[GMR_MAX_NQ + 1];
  memcpy(q, q_in, m->nq * sizeof(double));
  orc_fk(m, q, &k);
  for (int b = 0; b < m->nbody; b++) {
    memcpy(xpos + 3 * b, k.xpos[b], 3 * sizeof(double));
    memcpy(xquat + 4 * b, k.xquat[b], 4 * sizeof(double));
  }
}

/* ------------------------------------------------------------------------------------------ */
/* H4: SE(3) log of the body->target transform  (mink FrameTask.compute_error; App. A.5)       */
/* ------------------------------------------------------------------------------------------ */
/* SO3.log of a unit quaternion, |w| <= pi; small-angle branch on |vec|^2 < 1e-10. */
void orc_so3_log(const double q[4], double w[3]) {
  double qw = q[0];
  double n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  double f;
  if (n2 < 1e-10) {
    f = 2.0 / qw - 2.0 / 3.0 * n2 / (qw * qw * qw);
  } else {
    double n = sqrt(n2);
    if (fabs(qw) < 1e-10) f = (qw > 0.0 ? 1.0 : -1.0) * M_PI / n;
    else f = 2.0 * atan2(qw < 0 ? -n : n, fabs(qw)) / n;
  }
  w[0] = f * q[1]; w[1] = f * q[2]; w[2] = f * q[3];
}

/* coefficient of K^2 in V^-1 = I - K/2 + c K^2 :  c = (1 - (t/2) cot(t/2)) / t^2 */
static double vinv_coef(double t2) {
  if (t2 < 1e-2) /* series of (1 - x cot x)/(4x^2), x = t/2: truncation < 1e-19 at t2 = 1e-2 */
    return 1.0 / 12.0 + t2 * (1.0 / 720.0 + t2 * (1.0 / 30240.0 + t2 * (1.0 / 1209600.0 + t2 / 47900160.0)));
  double t = sqrt(t2), h = 0.5 * t;
  return (1.0 - h * cos(h) / sin(h)) / t2;
}

/* e = log(T_wb^-1 T_wt) = [V^-1(w) p_bt ; w] */
void orc_se3_log_rel(const double pb[3], const double qb[4], const double Rb[9],
                     const double pt[3], const double qt[4], double e[6]) {
  double qbc[4] = {qb[0], -qb[1], -qb[2], -qb[3]}, qbt[4], d[3], pbt[3], w[3], K[9], K2[9];
  quat_mul(qbt, qbc, qt);
  for (int a = 0; a < 3; a++) d[a] = pt[a] - pb[a];
  matT_vec(pbt, Rb, d);
  orc_so3_log(qbt, w);
  double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double c = vinv_coef(t2);
  skew(K, w);
  mat3_mul(K2, K, K);
  for (int i = 0; i < 3; i++) {
    double s = 0;
    for (int j = 0; j < 3; j++) s += ((i == j ? 1.0 : 0.0) - 0.5 * K[3 * i + j] + c * K2[3 * i + j]) * pbt[j];
    e[i] = s;
  }
  e[3] = w[0]; e[4] = w[1]; e[5] = w[2];
}

/* ------------------------------------------------------------------------------------------ */
/* H5(ii): inverse left Jacobian of SE(3) at e  (mink SE3.jlog of T_tb == Jl^-1(e); App. A.5)   */
/* ------------------------------------------------------------------------------------------ */
/* Jl^-1(e) = [[A, -A Q A], [0, A]],  A = I - K/2 + a K^2 (same coefficient as V^-1),
 * Q per Barfoot (7.86b).  When |w|^2 < 1e-10 mink returns the 6x6 identity. */
void orc_se3_jlinv(const double e[6], double A[9], double B[9]) {
  const double* rho = e;
  const double* w = e + 3;
  double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  memset(A, 0, 9 * sizeof(double));
  memset(B, 0, 9 * sizeof(double));
  A[0] = A[4] = A[8] = 1.0;
  if (t2 < 1e-10) return;
  double a = vinv_coef(t2);
  double W[9], W2[9], V[9];
  skew(W, w); skew(V, rho);
  mat3_mul(W2, W, W);
  for (int i = 0; i < 9; i++) A[i] += -0.5 * W[i] + a * W2[i];
  /* Q */
  double c1, c2, c3;
  if (t2 < 1e-2) { /* series: relative truncation error < 1e-16 at t2 = 1e-2 */
    c1 = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0 - t2 * t2 * t2 / 362880.0;
    c2 = -1.0 / 24.0 + t2 / 720.0 - t2 * t2 / 40320.0 + t2 * t2 * t2 / 3628800.0;
    c3 = -1.0 / 120.0 + t2 / 5040.0 - t2 * t2 / 362880.0 + t2 * t2 * t2 / 39916800.0;
  } else {
    double t = sqrt(t2), s = sin(t), c = cos(t);
    c1 = (t - s) / (t2 * t);
    c2 = (1.0 - 0.5 * t2 - c) / (t2 * t2);
    c3 = (t - s - t2 * t / 6.0) / (t2 * t2 * t);
  }
  double WV[9], VW[9], WVW[9], WWV[9], VWW[9], WVWW[9], WWVW[9], Q[9];
  mat3_mul(WV, W, V); mat3_mul(VW, V, W);
  mat3_mul(WVW, WV, W); mat3_mul(WWV, W, WV); mat3_mul(VWW, VW, W);
  mat3_mul(WVWW, WVW, W); mat3_mul(WWVW, W, WVW);
  double c4 = -0.5 * (c2 - 3.0 * c3);
  for (int i = 0; i < 9; i++)
    Q[i] = 0.5 * V[i] + c1 * (WV[i] + VW[i] + WVW[i]) - c2 * (WWV[i] + VWW[i] - 3.0 * WVW[i])
           + c4 * (WVWW[i] + WWVW[i]);
  double AQ[9];
  mat3_mul(AQ, A, Q);
  mat3_mul(B, AQ, A);
  for (int i = 0; i < 9; i++) B[i] = -B[i];
}

/* ------------------------------------------------------------------------------------------ */
/* H4/H5: task errors, QP objective and bounds                                                  */
/* ------------------------------------------------------------------------------------------ */
/* errors of every task of a stage and their unweighted 2-norm (motion_retarget.py:188-200) */
double orc_stage_error(const gmr_model_t* m, const gmr_taskset_t* ts, int stage, const orc_fk_t* k,
                       const double* tgt, double e[NK][6]) {
  (void)m;
  double ss = 0;
  for (int i = 0; i < ts->ntask[stage]; i++) {
    int b = ts->task_body[stage][i];
    const double* t = tgt + 7 * ts->task_human[stage][i];
    orc_se3_log_rel(k->xpos[b], k->xquat[b], k->xmat[b], t, t + 3, e[i]);
    for (int a = 0; a < 6; a++) ss += e[i][a] * e[i][a];
  }
  return sqrt(ss);
}

/* body-frame Jacobian of body b (6 x nv, rows [lin; ang]): mj_jacBody rotated by R_wb^T
 * (mink Configuration.get_frame_jacobian; App. A.4) */
void orc_body_jacobian(const gmr_model_t* m, const orc_fk_t* k, int b, double J[6][NV]) {
  int nv = m->nv;
  for (int r = 0; r < 6; r++) for (int d = 0; d < nv; d++) J[r][d] = 0.0;
  const double* Rb = k->xmat[b];
  for (int d = 0; d < 3; d++) { /* free joint translation: world axes */
    double lin[3] = {0, 0, 0}, o[3];
    lin[d] = 1.0;
    matT_vec(o, Rb, lin);
    for (int r = 0; r < 3; r++) J[r][d] = o[r];
  }
  for (int d = 0; d < 3; d++) { /* free joint rotation: body-local axes of the root */
    double ang[3] = {k->xmat[0][d], k->xmat[0][3 + d], k->xmat[0][6 + d]};
    double off[3], lin[3], o[3];
    for (int a = 0; a < 3; a++) off[a] = k->xpos[b][a] - k->xpos[0][a];
    cross3(lin, ang, off);
    matT_vec(o, Rb, lin);
    for (int r = 0; r < 3; r++) J[r][3 + d] = o[r];
    matT_vec(o, Rb, ang);
    for (int r = 0; r < 3; r++) J[3 + r][3 + d] = o[r];
  }
  for (int c = b; c > 0; c = m->parent[c]) { /* hinges on the path root -> b */
    int h = m->body_hinge[c];
    if (h < 0) continue;
    double off[3], lin[3], o[3];
    for (int a = 0; a < 3; a++) off[a] = k->xpos[b][a] - k->xpos[c][a]; /* anchor == xpos[c] */
    cross3(lin, k->xaxis[c], off);
    matT_vec(o, Rb, lin);
    for (int r = 0; r < 3; r++) J[r][6 + h] = o[r];
    matT_vec(o, Rb, k->xaxis[c]);
    for (int r = 0; r < 3; r++) J[3 + r][6 + h] = o[r];
  }
}

/* task Jacobian J = -Jl^-1(e) J_body (mink FrameTask.compute_jacobian) */
void orc_task_jacobian(const gmr_model_t* m, const orc_fk_t* k, int b, const double e[6], double J[6][NV]) {
  double Jb[6][NV], A[9], B[9];
  orc_body_jacobian(m, k, b, Jb);
  orc_se3_jlinv(e, A, B);
  for (int d = 0; d < m->nv; d++) {
    double lin[3] = {Jb[0][d], Jb[1][d], Jb[2][d]}, ang[3] = {Jb[3][d], Jb[4][d], Jb[5][d]};
    double al[3], ba[3], aa[3];
    mat_vec(al, A, lin); mat_vec(ba, B, ang); mat_vec(aa, A, ang);
    for (int r = 0; r < 3; r++) { J[r][d] = -(al[r] + ba[r]); J[3 + r][d] = -aa[r]; }
  }
}

/* H = damping I + sum_k (W J_k)^T (W J_k) + lm |W e_k|^2 I ;  c = sum_k (W J_k)^T (W e_k)
 * (mink Task.compute_qp_objective, _compute_qp_objective; App. A.6)
 * bounds of mink ConfigurationLimit(gain 0.95): limited hinges only. */
void orc_build_qp(const gmr_model_t* m, const gmr_taskset_t* ts, int stage, const double* q,
                  const orc_fk_t* k, const double e[NK][6], double* H, double* c, double* lo, double* hi) {
  int nv = m->nv;
  for (int i = 0; i < nv * nv; i++) H[i] = 0.0;
  for (int i = 0; i < nv; i++) { c[i] = 0.0; H[i * nv + i] = ts->damping; }
  for (int t = 0; t < ts->ntask[stage]; t++) {
    double J[6][NV], w[6], we[6];
    orc_task_jacobian(m, k, ts->task_body[stage][t], e[t], J);
    for (int r = 0; r < 6; r++) w[r] = r < 3 ? ts->w_pos[stage][t] : ts->w_rot[stage][t];
    double mu = 0;
    for (int r = 0; r < 6; r++) { we[r] = w[r] * e[t][r]; mu += we[r] * we[r]; }
    mu *= ts->lm_damping;
    for (int i = 0; i < nv; i++) {
      for (int j = 0; j < nv; j++) {
        double s = 0;
        for (int r = 0; r < 6; r++) s += (w[r] * J[r][i]) * (w[r] * J[r][j]);
        H[i * nv + j] += s;
      }
      H[i * nv + i] += mu;
      double s = 0;
      for (int r = 0; r < 6; r++) s += we[r] * (w[r] * J[r][i]);
      c[i] += s;
    }
  }
  for (int i = 0; i < 6; i++) { lo[i] = -INFINITY; hi[i] = INFINITY; }
  for (int h = 0; h < m->nhinge; h++) {
    if (m->limited[h]) {
      hi[6 + h] = ts->limit_gain * (m->range_hi[h] - q[7 + h]);
      lo[6 + h] = -ts->limit_gain * (q[7 + h] - m->range_lo[h]);
    } else { lo[6 + h] = -INFINITY; hi[6 + h] = INFINITY; }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* H5(v): strictly convex box-constrained QP, primal active set with dense Cholesky            */
/* (stands in for DAQP behind qpsolvers: same unique minimiser; App. A.6)                      */
/* ------------------------------------------------------------------------------------------ */
static int chol_solve(int n, double* K, double* rhs) { /* in-place LL^T and solve; 0 ok */
  for (int j = 0; j < n; j++) {
    double d = K[j * n + j];
    for (int p = 0; p < j; p++) d -= K[j * n + p] * K[j * n + p];
    if (!(d > 0.0)) return -1;
    d = sqrt(d);
    K[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = K[i * n + j];
      for (int p = 0; p < j; p++) s -= K[i * n + p] * K[j * n + p];
      K[i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; i++) {
    double s = rhs[i];
    for (int p = 0; p < i; p++) s -= K[i * n + p] * rhs[p];
    rhs[i] = s / K[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = rhs[i];
    for (int p = i + 1; p < n; p++) s -= K[p * n + i] * rhs[p];
    rhs[i] = s / K[i * n + i];
  }
  return 0;
}

/* returns number of factorisations (>0) or <0 on failure */
int orc_solve_box_qp(int n, const double* H, const double* c, const double* lo, const double* hi, double* x) {
  int st[NV];                 /* 0 free, -1 at lower, +1 at upper */
  double K[NV * NV], xe[NV];
  double cmax = 0;
  for (int i = 0; i < n; i++) {
    st[i] = 0;
    x[i] = 0.0 < lo[i] ? lo[i] : (0.0 > hi[i] ? hi[i] : 0.0);
    if (fabs(c[i]) > cmax) cmax = fabs(c[i]);
  }
  const double dual_tol = 1e-13 * (1.0 + cmax);
  int nfact = 0;
  for (int it = 0; it < 8 * n + 8; it++) {
    /* equality-constrained subproblem on the working set: fixed rows/cols -> identity */
    for (int i = 0; i < n; i++) {
      if (st[i]) {
        for (int j = 0; j < n; j++) K[i * n + j] = K[j * n + i] = 0.0;
      }
    }
    for (int i = 0; i < n; i++) {
      if (st[i]) { K[i * n + i] = 1.0; xe[i] = x[i]; continue; }
      double r = -c[i];
      for (int j = 0; j < n; j++) {
        if (st[j]) r -= H[i * n + j] * x[j];
        else K[i * n + j] = H[i * n + j];
      }
      xe[i] = r;
    }
    if (chol_solve(n, K, xe)) return -1;
    nfact++;
    /* ratio test */
    double alpha = 1.0;
    int blk = -1, side = 0;
    for (int i = 0; i < n; i++) {
      if (st[i]) continue;
      double p = xe[i] - x[i];
      if (p < 0.0 && xe[i] < lo[i]) {
        double a = (lo[i] - x[i]) / p;
        if (a < alpha) { alpha = a; blk = i; side = -1; }
      } else if (p > 0.0 && xe[i] > hi[i]) {
        double a = (hi[i] - x[i]) / p;
        if (a < alpha) { alpha = a; blk = i; side = 1; }
      }
    }
    if (blk >= 0) {
      if (alpha < 0.0) alpha = 0.0;
      for (int i = 0; i < n; i++) if (!st[i]) x[i] += alpha * (xe[i] - x[i]);
      x[blk] = side < 0 ? lo[blk] : hi[blk];
      st[blk] = side;
      continue;
    }
    for (int i = 0; i < n; i++) x[i] = xe[i];
    /* multipliers of the working set: g = Hx + c; need g >= 0 at lower, g <= 0 at upper */
    double worst = dual_tol;
    int rel = -1;
    for (int i = 0; i < n; i++) {
      if (!st[i]) continue;
      double g = c[i];
      for (int j = 0; j < n; j++) g += H[i * n + j] * x[j];
      double viol = st[i] < 0 ? -g : g;
      if (viol > worst) { worst = viol; rel = i; }
    }
    if (rel < 0) return nfact;
    st[rel] = 0;
  }
  return -2;
}

/* ------------------------------------------------------------------------------------------ */
/* H6 (first half): mj_integratePos with v = dq/dt, dt  (App. A.7)                             */
/* ------------------------------------------------------------------------------------------ */
void orc_integrate(const gmr_model_t* m, double* q, const double* dq) {
  double dt = m->timestep;
  double v[NV];
  for (int i = 0; i < m->nv; i++) v[i] = dq[i] / dt; /* solve_ik returns dq / dt */
  for (int a = 0; a < 3; a++) q[a] += dt * v[a];
  /* mju_quatIntegrate(quat, angvel, dt): body-local angular velocity */
  double ax[3] = {v[3], v[4], v[5]};
  double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  if (n < 1e-15) { ax[0] = 1; ax[1] = ax[2] = 0; n = 0; }
  else { ax[0] /= n; ax[1] /= n; ax[2] /= n; }
  double qrot[4], qn[4];
  axis_angle_quat(qrot, ax, dt * n);
  quat_normalize_mj(q + 3);
  quat_mul(qn, q + 3, qrot);
  memcpy(q + 3, qn, sizeof qn);
  for (int h = 0; h < m->nhinge; h++) q[7 + h] += dt * v[6 + h];
}

/* ------------------------------------------------------------------------------------------ */
/* H7: one retarget() call  (motion_retarget.py:139-185; App. A.8)                              */
/* ------------------------------------------------------------------------------------------ */
/* q: in = configuration before the frame, out = configuration after; nsolve[2]: solve_ik calls
 * per stage; returns 0 ok, <0 QP failure */
int orc_retarget_frame(const gmr_model_t* m, const gmr_taskset_t* ts, double* q, const double* human,
                       int offset_to_ground, int* nsolve, double* tgt_out) {
  double tgt[NHUM * 7], e[NK][6], H[NV * NV], c[NV], lo[NV], hi[NV], dq[NV];
  orc_fk_t k;
  orc_preprocess(ts, human, offset_to_ground, tgt);
  if (tgt_out) memcpy(tgt_out, tgt, sizeof(double) * 7 * ts->nhuman);
  orc_fk(m, q, &k);
  nsolve[0] = nsolve[1] = 0;
  for (int stage = 0; stage < 2; stage++) {
    if (!ts->use_stage[stage]) continue;
    double curr = orc_stage_error(m, ts, stage, &k, tgt, e);
    int num_iter = 0;
    for (;;) {
      orc_build_qp(m, ts, stage, q, &k, e, H, c, lo, hi);
      if (orc_solve_box_qp(m->nv, H, c, lo, hi, dq) < 0) return -1;
      orc_integrate(m, q, dq);
      orc_fk(m, q, &k);
      double next = orc_stage_error(m, ts, stage, &k, tgt, e);
      nsolve[stage]++;
      /* first solve is unconditional (:147-151); then `while curr - next > tol and n < max_iter` */
      if (nsolve[stage] > 1) num_iter++;
      if (!(curr - next > ts->tol && num_iter < ts->max_iter)) break;
      curr = next;
    }
  }
  return 0;
}

/* streams: q0[S][nq], human[S][T][nhuman][7] -> q_out[S][T][nq], nsolve[S][T][2], status[S].
 * The time loop is sequential per stream (warm start, motion_retarget.py:75); streams are
 * independent (OpenMP over streams when compiled with -fopenmp). */
void orc_retarget_streams(const gmr_model_t* m, const gmr_taskset_t* ts, int S, int T, const double* q0,
                          const double* human, int offset_to_ground, double* q_out, int* nsolve,
                          int* status, int nthreads) {
  int nq = m->nq, nh = ts->nhuman;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int s = 0; s < S; s++) {
    double q[GMR_MAX_NQ + 1];
    memcpy(q, q0 + (size_t)s * nq, nq * sizeof(double));
    status[s] = 0;
    for (int t = 0; t < T; t++) {
      size_t f = (size_t)s * T + t;
      if (status[s] == 0) {
        int rc = orc_retarget_frame(m, ts, q, human + f * nh * 7, offset_to_ground, nsolve + 2 * f, NULL);
        if (rc < 0) status[s] = rc;
      } else { nsolve[2 * f] = nsolve[2 * f + 1] = 0; }
      memcpy(q_out + f * nq, q, nq * sizeof(double));
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Parity-risk audit (tools/parity_risk.py): how far could the genuine stack (DAQP behind        */
/* qpsolvers, primal tolerance ~1e-6) be from this restatement, which solves every QP exactly?   */
/* ------------------------------------------------------------------------------------------ */
/* DAQP is a dual active-set method: its iterates solve the equality-constrained problem of a working set
 * exactly and it stops as soon as no constraint is violated by more than its primal tolerance.  The emulation
 * below has that termination rule: a bound violated by <= ptol is never added to the working set and the
 * returned x is NOT clipped (qpsolvers hands DAQP's x to mink unchanged).  ptol = 0 gives the exact minimiser.
 * Returns the number of factorisations or <0. */
int orc_solve_box_qp_relaxed(int n, const double* H, const double* c, const double* lo, const double* hi,
                             double ptol, double* x) {
  int st[NV];
  double K[NV * NV], xe[NV];
  double cmax = 0;
  for (int i = 0; i < n; i++) { st[i] = 0; if (fabs(c[i]) > cmax) cmax = fabs(c[i]); }
  const double dual_tol = 1e-12 * (1.0 + cmax);
  for (int it = 0; it < 16 * n + 16; it++) {
    for (int i = 0; i < n; i++) {
      if (st[i]) {
        for (int j = 0; j < n; j++) K[i * n + j] = K[j * n + i] = 0.0;
        K[i * n + i] = 1.0;
        xe[i] = st[i] < 0 ? lo[i] : hi[i];
        continue;
      }
      double r = -c[i];
      for (int j = 0; j < n; j++) {
        if (st[j]) r -= H[i * n + j] * (st[j] < 0 ? lo[j] : hi[j]);
        else K[i * n + j] = H[i * n + j];
      }
      xe[i] = r;
    }
    for (int i = 0; i < n; i++) if (st[i]) for (int j = 0; j < n; j++) if (j != i) K[j * n + i] = 0.0;
    if (chol_solve(n, K, xe)) return -1;
    /* most violated bound among the free variables (beyond the primal tolerance) joins the working set */
    double worst = ptol;
    int add = -1, side = 0;
    for (int i = 0; i < n; i++) {
      if (st[i]) continue;
      if (lo[i] - xe[i] > worst) { worst = lo[i] - xe[i]; add = i; side = -1; }
      if (xe[i] - hi[i] > worst) { worst = xe[i] - hi[i]; add = i; side = 1; }
    }
    if (add >= 0) { st[add] = side; continue; }
    /* multipliers of the working set */
    double wd = dual_tol;
    int rel = -1;
    for (int i = 0; i < n; i++) {
      if (!st[i]) continue;
      double g = c[i];
      for (int j = 0; j < n; j++) g += H[i * n + j] * xe[j];
      double viol = st[i] < 0 ? -g : g;
      if (viol > wd) { wd = viol; rel = i; }
    }
    if (rel >= 0) { st[rel] = 0; continue; }
    for (int i = 0; i < n; i++) x[i] = xe[i];
    return it + 1;
  }
  return -2;
}

static double audit_uniform(uint64_t* s) {   /* xorshift64*: uniform in [-1, 1) */
  *s ^= *s >> 12; *s ^= *s << 25; *s ^= *s >> 27;
  return (double)((*s * 2685821657736338717ull) >> 11) / 4503599627370496.0 - 1.0;
}

/* One retarget() call with the audit hooks.  qp_ptol > 0: every QP is solved by the relaxed-tolerance emulation;
 * qp_noise > 0: every component of every QP solution is moved by an independent uniform amount in
 * [-qp_noise, qp_noise] (a pessimistic stand-in for "any solver that is accurate to qp_noise").
 * margins[0] = min over this frame's stop-rule decisions of |(curr - next) - tol| (motion_retarget.py:153,172);
 * margins[1] = min over this frame's solves of the distance of a FREE limited joint's step from its bound
 *              (how close an inactive bound is to switching);
 * margins[2] = min over this frame's solves of |multiplier| of an ACTIVE bound (how close it is to releasing). */
int orc_retarget_frame_audit(const gmr_model_t* m, const gmr_taskset_t* ts, double* q, const double* human,
                             int offset_to_ground, double qp_ptol, double qp_noise, uint64_t* rng, int* nsolve,
                             double* margins) {
  double tgt[NHUM * 7], e[NK][6], H[NV * NV], c[NV], lo[NV], hi[NV], dq[NV];
  orc_fk_t k;
  const int nv = m->nv;
  orc_preprocess(ts, human, offset_to_ground, tgt);
  orc_fk(m, q, &k);
  nsolve[0] = nsolve[1] = 0;
  margins[0] = margins[1] = margins[2] = INFINITY;
  for (int stage = 0; stage < 2; stage++) {
    if (!ts->use_stage[stage]) continue;
    double curr = orc_stage_error(m, ts, stage, &k, tgt, e);
    int num_iter = 0;
    for (;;) {
      orc_build_qp(m, ts, stage, q, &k, e, H, c, lo, hi);
      int rc = qp_ptol > 0.0 ? orc_solve_box_qp_relaxed(nv, H, c, lo, hi, qp_ptol, dq)
                             : orc_solve_box_qp(nv, H, c, lo, hi, dq);
      if (rc < 0) return -1;
      for (int i = 6; i < nv; i++) {
        if (!(lo[i] > -INFINITY)) continue;
        const double gl = dq[i] - lo[i], gh = hi[i] - dq[i];
        const double tolb = 1e-12 * (1.0 + fabs(lo[i]) + fabs(hi[i]));
        if (gl > tolb && gh > tolb) {            /* free: distance to the nearer bound */
          const double gmin = gl < gh ? gl : gh;
          if (gmin < margins[1]) margins[1] = gmin;
        } else {                                 /* on a bound: |multiplier| */
          double g = c[i];
          for (int j = 0; j < nv; j++) g += H[i * nv + j] * dq[j];
          if (fabs(g) < margins[2]) margins[2] = fabs(g);
        }
      }
      if (qp_noise > 0.0) for (int i = 0; i < nv; i++) dq[i] += qp_noise * audit_uniform(rng);
      orc_integrate(m, q, dq);
      orc_fk(m, q, &k);
      double next = orc_stage_error(m, ts, stage, &k, tgt, e);
      nsolve[stage]++;
      if (nsolve[stage] > 1) num_iter++;
      if (num_iter < ts->max_iter) {             /* the decision is the error test (not the iteration cap) */
        const double mg = fabs((curr - next) - ts->tol);
        if (mg < margins[0]) margins[0] = mg;
      }
      if (!(curr - next > ts->tol && num_iter < ts->max_iter)) break;
      curr = next;
    }
  }
  return 0;
}

/* margins[S][T][3] as in orc_retarget_frame_audit; stream s draws its noise from seed + s */
void orc_retarget_streams_audit(const gmr_model_t* m, const gmr_taskset_t* ts, int S, int T, const double* q0,
                                const double* human, int offset_to_ground, double qp_ptol, double qp_noise,
                                uint64_t seed, double* q_out, int* nsolve, int* status, double* margins,
                                int nthreads) {
  int nq = m->nq, nh = ts->nhuman;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int s = 0; s < S; s++) {
    double q[GMR_MAX_NQ + 1];
    uint64_t rng = (seed + (uint64_t)s) * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull;
    if (!rng) rng = 1;
    memcpy(q, q0 + (size_t)s * nq, nq * sizeof(double));
    status[s] = 0;
    for (int t = 0; t < T; t++) {
      size_t f = (size_t)s * T + t;
      if (status[s] == 0) {
        int rc = orc_retarget_frame_audit(m, ts, q, human + f * nh * 7, offset_to_ground, qp_ptol, qp_noise, &rng,
                                          nsolve + 2 * f, margins + 3 * f);
        if (rc < 0) status[s] = rc;
      } else { nsolve[2 * f] = nsolve[2 * f + 1] = 0; margins[3 * f] = margins[3 * f + 1] = margins[3 * f + 2] = INFINITY; }
      memcpy(q_out + f * nq, q, nq * sizeof(double));
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* H9: post-hoc batched FK, float32, KinematicsModel semantics                                  */
/* (reference kinematics_model.py:172-182, 213-246; torch_utils.py:57-75,117-138,353-359)      */
/* ------------------------------------------------------------------------------------------ */
/* Tree arrays as parsed by the reference's own XML reader (un-normalised xyzw body quats, f32).
 * The hinge quaternion is formed like axis_angle_to_quat does under torch type promotion: sin/cos
 * of the float32 half angle, products and the normalisation in float64, rounded to float32 on
 * assignment (kinematics_model.py:32).  quat_mul is the plain Hamilton product here (the
 * reference uses an 8-multiplication rearrangement, torch_utils.py:117-138: same value up to
 * float32 rounding; the fixture tolerance covers it). */
static void quat_mul_xyzw_f32(float r[4], const float a[4], const float b[4]) {
  float x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  float y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  float z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  float w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  r[0] = x; r[1] = y; r[2] = z; r[3] = w;
}
static void quat_rotate_xyzw_f32(float r[3], const float q[4], const float v[3]) {
  float w = q[3];
  float s = 2.0f * w * w - 1.0f;
  float cx = q[1] * v[2] - q[2] * v[1], cy = q[2] * v[0] - q[0] * v[2], cz = q[0] * v[1] - q[1] * v[0];
  float d = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
  r[0] = v[0] * s + cx * w * 2.0f + q[0] * d * 2.0f;
  r[1] = v[1] * s + cy * w * 2.0f + q[1] * d * 2.0f;
  r[2] = v[2] * s + cz * w * 2.0f + q[2] * d * 2.0f;
}

void orc_fk_f32(int nbody, const int32_t* parent, const float* local_t, const float* local_r,
                const int32_t* dof_idx, const double* axis, int ndof, int B, const float* root_pos,
                const float* root_rot, const float* dof, float* body_pos, float* body_rot) {
  for (int f = 0; f < B; f++) {
    float* bp = body_pos + (size_t)f * nbody * 3;
    float* br = body_rot + (size_t)f * nbody * 4;
    memcpy(bp, root_pos + 3 * f, 3 * sizeof(float));
    memcpy(br, root_rot + 4 * f, 4 * sizeof(float));
    for (int j = 1; j < nbody; j++) {
      float jr[4] = {0, 0, 0, 1};
      if (dof_idx[j] >= 0) {
        float th = dof[(size_t)f * ndof + dof_idx[j]] / 2.0f;
        double s = (double)sinf(th), c = (double)cosf(th);
        const double* a = axis + 3 * j;
        double an = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        if (an < 1e-9) an = 1e-9;
        double q[4] = {a[0] / an * s, a[1] / an * s, a[2] / an * s, c};
        double qn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        if (qn < 1e-9) qn = 1e-9;
        for (int i = 0; i < 4; i++) jr[i] = (float)(q[i] / qn);
      }
      int p = parent[j];
      float wt[3], cr[4];
      quat_rotate_xyzw_f32(wt, br + 4 * p, local_t + 3 * j);
      for (int a = 0; a < 3; a++) bp[3 * j + a] = bp[3 * p + a] + wt[a];
      quat_mul_xyzw_f32(cr, local_r + 4 * j, jr);
      quat_mul_xyzw_f32(br + 4 * j, br + 4 * p, cr);
    }
  }
}

/* ABI helpers for the Python wrapper */
int orc_sizeof_model(void) { return (int)sizeof(gmr_model_t); }
int orc_sizeof_taskset(void) { return (int)sizeof(gmr_taskset_t); }
int orc_has_openmp(void) {
#ifdef _OPENMP
  return 1;
#else
  return 0;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* flat entry points for the property tests (finite differences, KKT checks)                   */
/* ------------------------------------------------------------------------------------------ */
/* e_out[K][6]; returns the unweighted norm */
double orc_stage_error_flat(const gmr_model_t* m, const gmr_taskset_t* ts, int stage, const double* q_in,
                            const double* tgt, double* e_out) {
  orc_fk_t k;
  double q[GMR_MAX_NQ + 1], e[NK][6];
  memcpy(q, q_in, m->nq * sizeof(double));
  orc_fk(m, q, &k);
  double E = orc_stage_error(m, ts, stage, &k, tgt, e);
  memcpy(e_out, e, sizeof(double) * 6 * ts->ntask[stage]);
  return E;
}

/* J_out[K][6][nv] (dense, row-major with row length nv) */
void orc_task_jacobians_flat(const gmr_model_t* m, const gmr_taskset_t* ts, int stage, const double* q_in,
                             const double* tgt, double* J_out) {
  orc_fk_t k;
  double q[GMR_MAX_NQ + 1], e[NK][6], J[6][NV];
  memcpy(q, q_in, m->nq * sizeof(double));
  orc_fk(m, q, &k);
  orc_stage_error(m, ts, stage, &k, tgt, e);
  for (int t = 0; t < ts->ntask[stage]; t++) {
    orc_task_jacobian(m, &k, ts->task_body[stage][t], e[t], J);
    for (int r = 0; r < 6; r++)
      for (int d = 0; d < m->nv; d++) J_out[((size_t)t * 6 + r) * m->nv + d] = J[r][d];
  }
}

/* H[nv][nv], c, lo, hi of the QP solve_ik would build at q */
void orc_build_qp_flat(const gmr_model_t* m, const gmr_taskset_t* ts, int stage, const double* q_in,
                       const double* tgt, double* H, double* c, double* lo, double* hi) {
  orc_fk_t k;
  double q[GMR_MAX_NQ + 1], e[NK][6];
  memcpy(q, q_in, m->nq * sizeof(double));
  orc_fk(m, q, &k);
  orc_stage_error(m, ts, stage, &k, tgt, e);
  orc_build_qp(m, ts, stage, q, &k, e, H, c, lo, hi);
}
