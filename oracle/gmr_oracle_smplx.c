/*
 * gmr_oracle_smplx.c -- CPU restatement (plain C, FP64) of the SMPL-X frame extraction that feeds the
 * retargeting loop ("next" row N1).  TEST INFRASTRUCTURE ONLY: linked into
 * libgmr_oracle.so; only tests/, __graft_entry__.smoke() and measurement tools may call it.
 *
 * Parity status
 *   orc_smplx_align   PINNED by tests/golden/g_smplx.npz (outputs of the reference's own
 *                     general_motion_retargeting/utils/smpl.py:44-197 run in the build container).
 *                     Follows: slerp() :76-107, get_smplx_data_offline_fast() :109-197 (fps alignment
 *                     :123-169, orientation chain :173-195), get_smplx_data() :44-73; the SciPy pieces it
 *                     calls are restated from SciPy 1.15's documented formulas (Rotation.from_rotvec /
 *                     as_rotvec / from_quat / __mul__, interpolate.interp1d(kind="linear")).
 *   orc_smplx_joints  parity UNPINNED: the SMPL-X body model lives in the third-party `smplx` package
 *                     (requirements.txt:22, unpinned VCS URL), absent here together with its model files.
 *                     Restates the published joints-only part of its forward pass (Pavlakos et al. 2019;
 *                     smplx lbs.py batch_rodrigues / batch_rigid_transform as recalled): Rodrigues per
 *                     joint, chain of rigid transforms over the kinematic tree, + transl.  The reference's
 *                     call site is utils/smpl.py:12-34.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define SX_MAX_JOINTS 64

/* ---- SciPy Rotation pieces (quaternions xyzw like SciPy stores them) ------------------------------- */
static void sx_from_rotvec(const double v[3], double q[4]) { /* Rotation.from_rotvec */
  double a = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  double scale;
  if (a <= 1e-3) {
    double a2 = a * a;
    scale = 0.5 - a2 / 48.0 + a2 * a2 / 3840.0;
  } else {
    scale = sin(a / 2.0) / a;
  }
  q[0] = scale * v[0]; q[1] = scale * v[1]; q[2] = scale * v[2]; q[3] = cos(a / 2.0);
}

static void sx_as_rotvec(const double qin[4], double v[3]) { /* Rotation.as_rotvec */
  double q[4] = {qin[0], qin[1], qin[2], qin[3]};
  if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  double a = 2.0 * atan2(sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]), q[3]);
  double scale;
  if (a <= 1e-3) {
    double a2 = a * a;
    scale = 2.0 + a2 / 12.0 + 7.0 * a2 * a2 / 2880.0;
  } else {
    scale = a / sin(a / 2.0);
  }
  v[0] = scale * q[0]; v[1] = scale * q[1]; v[2] = scale * q[2];
}

static void sx_normalize4(double q[4]) { /* Rotation.from_quat normalises */
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

static void sx_compose(const double p[4], const double q[4], double r[4]) { /* Rotation.__mul__: p * q */
  double cx = p[1] * q[2] - p[2] * q[1], cy = p[2] * q[0] - p[0] * q[2], cz = p[0] * q[1] - p[1] * q[0];
  r[0] = p[3] * q[0] + q[3] * p[0] + cx;
  r[1] = p[3] * q[1] + q[3] * p[1] + cy;
  r[2] = p[3] * q[2] + q[3] * p[2] + cz;
  r[3] = p[3] * q[3] - p[0] * q[0] - p[1] * q[1] - p[2] * q[2];
  sx_normalize4(r);
}

/* slerp(rot1, rot2, t) of utils/smpl.py:76-107; q1, q2 xyzw unit; result normalised (R.from_quat) */
void orc_smplx_slerp(const double q1in[4], const double q2in[4], double t, double out[4]) {
  double q1[4] = {q1in[0], q1in[1], q1in[2], q1in[3]}, q2[4] = {q2in[0], q2in[1], q2in[2], q2in[3]};
  sx_normalize4(q1);                                                  /* :83-84 */
  sx_normalize4(q2);
  double dot = q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2] + q1[3] * q2[3];   /* :87 */
  if (dot < 0.0) { for (int i = 0; i < 4; i++) q2[i] = -q2[i]; dot = -dot; }    /* :90-92 */
  if (dot > 0.9995) {                                                            /* :95-96 */
    for (int i = 0; i < 4; i++) out[i] = q1[i] + t * (q2[i] - q1[i]);
    sx_normalize4(out);
    return;
  }
  double th0 = acos(dot), th = th0 * t;                                          /* :99-100 */
  double st = sin(th), st0 = sin(th0);
  double s0 = cos(th) - dot * st / st0, s1 = st / st0;                           /* :104-105 */
  for (int i = 0; i < 4; i++) out[i] = s0 * q1[i] + s1 * q2[i];
  sx_normalize4(out);
}

static void sx_slerp_rotvec(const float* a, const float* b, double alpha, double v[3]) {
  /* :137-140 / :153-156: R.from_rotvec(float32 row) twice, slerp, .as_rotvec() */
  double va[3] = {a[0], a[1], a[2]}, vb[3] = {b[0], b[1], b[2]}, qa[4], qb[4], q[4];
  sx_from_rotvec(va, qa);
  sx_from_rotvec(vb, qb);
  orc_smplx_slerp(qa, qb, alpha, q);
  sx_as_rotvec(q, v);
}

/*
 * N source frames, J joints (parents[i] < i, parents[0] = -1), `joints` rows have jstride >= J joints.
 * target_time != NULL: the fps-alignment branch (:123-169) for the Nout times of np.linspace(0, N-1, Nout);
 * target_time == NULL: no alignment (Nout must equal N), also what get_smplx_data returns per frame (:44-73).
 * sel (nsel joints, or NULL = all J) chooses the rows written: out[Nout][nsel][7] = pos xyz, quat wxyz.
 */
int orc_smplx_align(int N, int J, int jstride, const int32_t* parents, const float* full_pose, const float* joints,
                    int Nout, const double* target_time, int nsel, const int32_t* sel, double* out) {
  if (J < 1 || J > SX_MAX_JOINTS || N < 1 || jstride < J) return -1;
  if (!target_time && Nout != N) return -1;
  if (target_time && N < 2) return -1;               /* interp1d needs two samples */
  int nrow = sel ? nsel : J;
#pragma omp parallel for schedule(static)
  for (int o = 0; o < Nout; o++) {
    double quat[SX_MAX_JOINTS][4], pos[SX_MAX_JOINTS][3];
    if (target_time) {
      double t = target_time[o];
      int idx1 = (int)floor(t);                       /* :133-135 */
      int idx2 = idx1 + 1 < N - 1 ? idx1 + 1 : N - 1;
      double alpha = t - idx1;
      /* interp1d linear (:160-165): lo = searchsorted_left(x, t).clip(1, N-1) - 1 */
      int ss = (int)ceil(t);
      if (ss < 1) ss = 1;
      if (ss > N - 1) ss = N - 1;
      int lo = ss - 1, hi = ss;
      for (int j = 0; j < J; j++) {
        double v[3], ql[4];
        sx_slerp_rotvec(full_pose + ((size_t)idx1 * J + j) * 3, full_pose + ((size_t)idx2 * J + j) * 3, alpha, v);
        sx_from_rotvec(v, ql);                        /* :181-186 */
        if (j == 0) { for (int k = 0; k < 4; k++) quat[0][k] = ql[k]; }
        else sx_compose(quat[parents[j]], ql, quat[j]);
        for (int c = 0; c < 3; c++) {
          float ylo = joints[((size_t)lo * jstride + j) * 3 + c], yhi = joints[((size_t)hi * jstride + j) * 3 + c];
          float d = yhi - ylo;                        /* float32 difference, as NumPy evaluates y_hi - y_lo */
          double slope = (double)d / (double)(hi - lo);
          pos[j][c] = slope * (t - (double)lo) + (double)ylo;
        }
      }
    } else {
      for (int j = 0; j < J; j++) {
        const float* r = full_pose + ((size_t)o * J + j) * 3;
        double v[3] = {r[0], r[1], r[2]}, ql[4];
        sx_from_rotvec(v, ql);
        if (j == 0) { for (int k = 0; k < 4; k++) quat[0][k] = ql[k]; }
        else sx_compose(quat[parents[j]], ql, quat[j]);
        for (int c = 0; c < 3; c++) pos[j][c] = joints[((size_t)o * jstride + j) * 3 + c];
      }
    }
    for (int r = 0; r < nrow; r++) {
      int j = sel ? sel[r] : r;
      double* w = out + ((size_t)o * nrow + r) * 7;
      w[0] = pos[j][0]; w[1] = pos[j][1]; w[2] = pos[j][2];
      w[3] = quat[j][3]; w[4] = quat[j][0]; w[5] = quat[j][1]; w[6] = quat[j][2];   /* as_quat(scalar_first=True) */
    }
  }
  return 0;
}

/* ---- joints-only SMPL-X forward (parity unpinned, see header) --------------------------------------- */
/* j_rest f64[J][3] (= J_regressor (v_template + shapedirs beta), once per clip), full_pose f32[N][J][3],
 * transl f32[N][3] -> joints f32[N][J][3].  FP64 inside, rounded once to float32 (the reference's model
 * computes this in float32 torch, so agreement with it is at float32 rounding level by construction). */
int orc_smplx_joints(int N, int J, const int32_t* parents, const double* j_rest, const float* full_pose,
                     const float* transl, float* joints) {
  if (J < 1 || J > SX_MAX_JOINTS || N < 0) return -1;
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; n++) {
    double Rg[SX_MAX_JOINTS][9], pg[SX_MAX_JOINTS][3];
    for (int j = 0; j < J; j++) {
      const float* r = full_pose + ((size_t)n * J + j) * 3;
      /* batch_rodrigues: angle = |v + 1e-8|, axis = v / angle, R = I + sin K + (1 - cos) K^2 */
      double vx = r[0], vy = r[1], vz = r[2];
      double ax = vx + 1e-8, ay = vy + 1e-8, az = vz + 1e-8;
      double ang = sqrt(ax * ax + ay * ay + az * az);
      double x = vx / ang, y = vy / ang, z = vz / ang, s = sin(ang), c1 = 1.0 - cos(ang);
      double K[9] = {0, -z, y, z, 0, -x, -y, x, 0}, K2[9], Rl[9];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
          double acc = 0;
          for (int k = 0; k < 3; k++) acc += K[a * 3 + k] * K[k * 3 + b];
          K2[a * 3 + b] = acc;
        }
      for (int i = 0; i < 9; i++) Rl[i] = (i % 4 == 0 ? 1.0 : 0.0) + s * K[i] + c1 * K2[i];
      if (j == 0) {
        for (int i = 0; i < 9; i++) Rg[0][i] = Rl[i];
        for (int c = 0; c < 3; c++) pg[0][c] = j_rest[c];
      } else {
        int p = parents[j];
        double rel[3] = {j_rest[j * 3] - j_rest[p * 3], j_rest[j * 3 + 1] - j_rest[p * 3 + 1], j_rest[j * 3 + 2] - j_rest[p * 3 + 2]};
        for (int a = 0; a < 3; a++) {
          for (int b = 0; b < 3; b++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += Rg[p][a * 3 + k] * Rl[k * 3 + b];
            Rg[j][a * 3 + b] = acc;
          }
          pg[j][a] = pg[p][a] + Rg[p][a * 3] * rel[0] + Rg[p][a * 3 + 1] * rel[1] + Rg[p][a * 3 + 2] * rel[2];
        }
      }
    }
    for (int j = 0; j < J; j++)
      for (int c = 0; c < 3; c++) joints[((size_t)n * J + j) * 3 + c] = (float)(pg[j][c] + (double)transl[(size_t)n * 3 + c]);
  }
  return 0;
}
