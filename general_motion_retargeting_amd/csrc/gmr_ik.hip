// gmr_ik.hip -- the retargeting hot path (rows H2-H7 of SURVEY.md section 8a) as ONE gfx950 kernel.
//
// One 64-lane wavefront per motion stream (clip / robot instance).  Frames of a stream are
// sequentially dependent (the configuration is warm-started from the previous frame, reference
// motion_retarget.py:75,139-185), so the time loop runs on the device and the parallel width of a
// launch is the number of streams.  Everything a stream touches between two frames lives in LDS:
//
//   constants  joint-local transforms (body pos/quat, hinge axes), limits, task tables
//   state      q, FK (xpos/xquat/world hinge axes), targets, task residuals e_k, -Jl^-1(e_k),
//              the weighted per-(task,dof) Jacobian columns, H, its working copy, c, bounds
//
// Per frame only nhuman*7 doubles are read from HBM (coalesced, contiguous) and nq doubles are
// written, i.e. 1 072 B/frame for G1: the kernel is bound by the dependent FP64 chain, not by HBM
// ("Roofline").
//
// Lane mapping: FK -> lane = body (each lane walks its own root->body chain in registers, no
// level-by-level LDS hand-off); residuals / Jl^-1 -> lane = task; Jacobian columns -> lane = (task,
// ancestor dof) pair; H accumulation -> lane = (column a, column b) of the current task's block
// (the Jacobian of a task is non-zero only on the dofs of its root->frame path, so J^T W^2 J is
// accumulated block-sparse: 11k instead of 103k multiply-adds for G1); QP -> lane = row.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/gmr_hip.h"
#include "gmr_device_math.h"
#include "gmr_ik_layout.h"

namespace gmr {

#define WSYNC() __syncthreads()

struct IkSmem {
  double* base;
  const IkLayout* L;
  __device__ __forceinline__ double* d(int off) const { return base + off; }
};

// ---------------------------------------------------------------------------------------------
// FK: mj_kinematics semantics (App. A.3).  lane b < nb.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void fk_wave(const IkLayout& L, double* sm, const short* chain, const short* depth,
                                        const short* body_hinge, int lane) {
  const int nb = L.nb;
  double* q = sm + L.q;
  double* lq = sm + L.lq;
  // step A: local quaternion of every body (body quat * hinge rotation); body 0: normalised root quat
  if (lane < nb) {
    d4 r;
    if (lane == 0) {
      r = qnormalize(d4{q[3], q[4], q[5], q[6]});
      q[3] = r.w; q[4] = r.x; q[5] = r.y; q[6] = r.z;
    } else {
      const double* bq = sm + L.body_quat + 4 * lane;
      r = d4{bq[0], bq[1], bq[2], bq[3]};
      int h = body_hinge[lane];
      if (h >= 0) {
        const double* ax = sm + L.axis + 3 * lane;
        double th = q[7 + h];
        if (th != 0.0) r = qmul(r, axis_angle(d3{ax[0], ax[1], ax[2]}, th));
      }
    }
    lq[4 * lane + 0] = r.w; lq[4 * lane + 1] = r.x; lq[4 * lane + 2] = r.y; lq[4 * lane + 3] = r.z;
  }
  WSYNC();
  // step B: every lane walks root -> its body
  if (lane < nb) {
    d3 pos = {q[0], q[1], q[2]};
    d4 quat = {lq[0], lq[1], lq[2], lq[3]};
    const int dep = depth[lane];
    const short* ch = chain + lane * L.maxd;
    for (int dd = 1; dd <= dep; dd++) {
      int c = ch[dd];
      const double* bp = sm + L.body_pos + 3 * c;
      pos = pos + qrot(quat, d3{bp[0], bp[1], bp[2]});
      quat = qnormalize(qmul(quat, d4{lq[4 * c], lq[4 * c + 1], lq[4 * c + 2], lq[4 * c + 3]}));
    }
    double* xp = sm + L.xpos + 3 * lane;
    double* xq = sm + L.xquat + 4 * lane;
    xp[0] = pos.x; xp[1] = pos.y; xp[2] = pos.z;
    xq[0] = quat.w; xq[1] = quat.x; xq[2] = quat.y; xq[3] = quat.z;
    if (body_hinge[lane] >= 0) {
      const double* ax = sm + L.axis + 3 * lane;
      d3 aw = qrot(quat, d3{ax[0], ax[1], ax[2]});
      double* xa = sm + L.xaxis + 3 * lane;
      xa[0] = aw.x; xa[1] = aw.y; xa[2] = aw.z;
    }
  }
  WSYNC();
}

// ---------------------------------------------------------------------------------------------
// residuals of the stage's tasks (mink FrameTask.compute_error) and their unweighted norm
// (motion_retarget.py:188-200).  lane k < K.  Returns E in every lane.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double errors_wave(const IkLayout& L, double* sm, const short* task_body,
                                              const short* task_human, int K, int lane) {
  double ss = 0.0;
  if (lane < K) {
    int b = task_body[lane], h = task_human[lane];
    const double* xp = sm + L.xpos + 3 * b;
    const double* xq = sm + L.xquat + 4 * b;
    const double* tg = sm + L.tgt + 7 * h;
    double e[6];
    se3_log_rel(d3{xp[0], xp[1], xp[2]}, d4{xq[0], xq[1], xq[2], xq[3]}, d3{tg[0], tg[1], tg[2]},
                d4{tg[3], tg[4], tg[5], tg[6]}, e);
    double* eo = sm + L.e + 6 * lane;
#pragma unroll
    for (int r = 0; r < 6; r++) { eo[r] = e[r]; ss += e[r] * e[r]; }
  }
  ss = wave_sum(ss);
  WSYNC();
  return sqrt(ss);
}

// ---------------------------------------------------------------------------------------------
// QP assembly (mink compute_qp_objective + ConfigurationLimit; App. A.4-A.6)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void build_qp_wave(const IkLayout& L, double* sm, int stage, const short* task_body,
                                              const short* task_col0, const short* task_ncol,
                                              const short* pair_task, const short* pair_dof,
                                              const short* pair_index, const short* hinge_body,
                                              const short* limited, double damping, double lm_damping,
                                              double limit_gain, int lane) {
  const int K = L.K[stage], P = L.P[stage], nv = L.nv, ldh = L.ldh;
  const double* wpos = sm + L.wpos[stage];
  const double* wrot = sm + L.wrot[stage];
  // (a) lane = task: M_k = -Jl^-1(e_k) (blocks -A, -B), weighted residual, LM term
  double mu = 0.0;
  if (lane < K) {
    const double* e = sm + L.e + 6 * lane;
    double ee[6];
#pragma unroll
    for (int r = 0; r < 6; r++) ee[r] = e[r];
    m3 A, B;
    se3_jlinv(ee, A, B);
    double* M = sm + L.M + 18 * lane;
#pragma unroll
    for (int i = 0; i < 9; i++) { M[i] = -A.a[i]; M[9 + i] = -B.a[i]; }
    double wp = wpos[lane], wr = wrot[lane];
    double* we = sm + L.we + 6 * lane;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      double v = (r < 3 ? wp : wr) * ee[r];
      we[r] = v;
      mu += v * v;
    }
  }
  mu = lm_damping * wave_sum(mu);
  // zero H (both triangles are written below)
  double* H = sm + L.H;
  for (int i = lane; i < nv * ldh; i += 64) H[i] = 0.0;
  WSYNC();
  // (b) lane = (task, dof) pair: weighted task-Jacobian column W_k * (-Jl^-1(e_k)) * J_body[:, d]
  double* Jw = sm + L.Jw;
  const double* xpos = sm + L.xpos;
  const double* xquat = sm + L.xquat;
  for (int p = lane; p < P; p += 64) {
    int k = pair_task[p], dof = pair_dof[p];
    int b = task_body[k];
    d3 pb = {xpos[3 * b], xpos[3 * b + 1], xpos[3 * b + 2]};
    d4 qb = {xquat[4 * b], xquat[4 * b + 1], xquat[4 * b + 2], xquat[4 * b + 3]};
    d3 lin, ang;
    if (dof < 3) {
      lin = d3{dof == 0 ? 1.0 : 0.0, dof == 1 ? 1.0 : 0.0, dof == 2 ? 1.0 : 0.0};
      ang = d3{0.0, 0.0, 0.0};
    } else if (dof < 6) {
      d4 q0 = {xquat[0], xquat[1], xquat[2], xquat[3]};
      int a = dof - 3;
      ang = qrot(q0, d3{a == 0 ? 1.0 : 0.0, a == 1 ? 1.0 : 0.0, a == 2 ? 1.0 : 0.0});
      lin = cross(ang, pb - d3{xpos[0], xpos[1], xpos[2]});
    } else {
      int c = hinge_body[dof - 6];
      const double* xa = sm + L.xaxis + 3 * c;
      ang = d3{xa[0], xa[1], xa[2]};
      lin = cross(ang, pb - d3{xpos[3 * c], xpos[3 * c + 1], xpos[3 * c + 2]});
    }
    d3 jl = qrot_inv(qb, lin), ja = qrot_inv(qb, ang);   // body-frame Jacobian column
    const double* M = sm + L.M + 18 * k;
    double wp = wpos[k], wr = wrot[k];
    double* o = Jw + 6 * p;
    // [ -A  -B ] [jl]      rows 0..2 (scaled by w_pos)
    // [  0  -A ] [ja]      rows 3..5 (scaled by w_rot)
#pragma unroll
    for (int r = 0; r < 3; r++) {
      double top = M[3 * r] * jl.x + M[3 * r + 1] * jl.y + M[3 * r + 2] * jl.z + M[9 + 3 * r] * ja.x +
                   M[9 + 3 * r + 1] * ja.y + M[9 + 3 * r + 2] * ja.z;
      double bot = M[3 * r] * ja.x + M[3 * r + 1] * ja.y + M[3 * r + 2] * ja.z;
      o[r] = wp * top;
      o[3 + r] = wr * bot;
    }
  }
  WSYNC();
  // (c) lane = dof: c = sum_k (W J_k)^T (W e_k); bounds of the limited hinges
  if (lane < nv) {
    double cc = 0.0;
    for (int k = 0; k < K; k++) {
      int p = pair_index[k * nv + lane];
      if (p >= 0) {
        const double* j = Jw + 6 * p;
        const double* we = sm + L.we + 6 * k;
        cc += j[0] * we[0] + j[1] * we[1] + j[2] * we[2] + j[3] * we[3] + j[4] * we[4] + j[5] * we[5];
      }
    }
    (sm + L.c)[lane] = cc;
    double lo = -INFINITY, hi = INFINITY;
    if (lane >= 6 && limited[lane - 6]) {
      double th = (sm + L.q)[7 + lane - 6];
      hi = limit_gain * ((sm + L.range_hi)[lane - 6] - th);
      lo = -limit_gain * (th - (sm + L.range_lo)[lane - 6]);
    }
    (sm + L.lo)[lane] = lo;
    (sm + L.hi)[lane] = hi;
  }
  // (d) H += (W J_k)^T (W J_k), one task block at a time; lane = (a, b) with a >= b
  for (int k = 0; k < K; k++) {
    int n = task_ncol[k], c0 = task_col0[k];
    int npair = n * (n + 1) / 2;
    for (int idx = lane; idx < npair; idx += 64) {
      int a = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
      while (a * (a + 1) / 2 > idx) a--;
      while ((a + 1) * (a + 2) / 2 <= idx) a++;
      int b = idx - a * (a + 1) / 2;
      const double* ja = Jw + 6 * (c0 + a);
      const double* jb = Jw + 6 * (c0 + b);
      double s = ja[0] * jb[0] + ja[1] * jb[1] + ja[2] * jb[2] + ja[3] * jb[3] + ja[4] * jb[4] + ja[5] * jb[5];
      int da = pair_dof[c0 + a], db = pair_dof[c0 + b];
      H[da * ldh + db] += s;
      if (da != db) H[db * ldh + da] += s;
    }
    WSYNC();
  }
  if (lane < nv) H[lane * ldh + lane] += damping + mu;
  WSYNC();
}

// ---------------------------------------------------------------------------------------------
// box-constrained strictly convex QP: primal active set, dense Cholesky in LDS (lane = row).
// Same unique minimiser as DAQP behind qpsolvers (App. A.6).  Returns 0 ok / <0 failure; the
// solution is left in sm[L.x].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int solve_qp_wave(const IkLayout& L, double* sm, int lane) {
  const int n = L.nv, ldh = L.ldh;
  const double* H = sm + L.H;
  double* Kf = sm + L.Kf;
  const double* cvec = sm + L.c;
  double* xs = sm + L.x;
  const bool act = lane < n;
  const double lo = act ? (sm + L.lo)[lane] : 0.0, hi = act ? (sm + L.hi)[lane] : 0.0;
  const double ci = act ? cvec[lane] : 0.0;
  double x = act ? fmin(fmax(0.0, lo), hi) : 0.0;
  int st = 0;  // 0 free, -1 at lower, +1 at upper
  const double dual_tol = 1e-13 * (1.0 + wave_max(fabs(ci)));
  if (act) xs[lane] = x;
  WSYNC();
  for (int it = 0; it < 8 * n + 8; it++) {
    const unsigned long long fixed = __ballot(act && st != 0);
    // working copy K (lower triangle incl. diagonal) and right-hand side
    double rhs = 0.0;
    if (act) {
      if (st != 0) rhs = x;
      else {
        rhs = -ci;
        if (fixed) {
          for (int j = 0; j < n; j++)
            if ((fixed >> j) & 1ull) rhs -= H[lane * ldh + j] * xs[j];
        }
      }
      for (int j = 0; j <= lane; j++) {
        bool fx = (st != 0) || ((fixed >> j) & 1ull);
        Kf[lane * ldh + j] = fx ? (j == lane ? 1.0 : 0.0) : H[lane * ldh + j];
      }
    }
    WSYNC();
    // Cholesky, left-looking: lane i >= j forms s = K[i][j] - sum_{p<j} L[i][p] L[j][p]
    int fail = 0;
    for (int j = 0; j < n; j++) {
      double s = 0.0;
      if (act && lane >= j) {
        s = Kf[lane * ldh + j];
        const double* ri = Kf + lane * ldh;
        const double* rj = Kf + j * ldh;
        double s2 = 0.0;
        int p = 0;
        for (; p + 1 < j; p += 2) { s -= ri[p] * rj[p]; s2 -= ri[p + 1] * rj[p + 1]; }
        if (p < j) s -= ri[p] * rj[p];
        s += s2;
      }
      double djj = __shfl(s, j, 64);
      if (!(djj > 0.0)) { fail = 1; break; }
      double dinv = 1.0 / sqrt(djj);
      if (act && lane >= j) Kf[lane * ldh + j] = (lane == j) ? djj * dinv : s * dinv;
      WSYNC();
    }
    if (fail) return GMR_STATUS_QP_FAILED;
    // forward substitution L y = rhs (column sweep), then L^T x = y
    double b = rhs;
    for (int j = 0; j < n; j++) {
      double yj = __shfl(b, j, 64) / Kf[j * ldh + j];
      if (lane == j) b = yj;
      else if (act && lane > j) b -= Kf[lane * ldh + j] * yj;
    }
    for (int j = n - 1; j >= 0; j--) {
      double xj = __shfl(b, j, 64) / Kf[j * ldh + j];
      if (lane == j) b = xj;
      else if (lane < j) b -= Kf[j * ldh + lane] * xj;
    }
    const double xe = b;
    // ratio test against the bounds of the free variables
    double alpha = 2.0;
    int side = 0;
    if (act && st == 0) {
      double p = xe - x;
      if (p < 0.0 && xe < lo) { alpha = (lo - x) / p; side = -1; }
      else if (p > 0.0 && xe > hi) { alpha = (hi - x) / p; side = 1; }
    }
    double amin = wave_min(alpha);
    if (amin < 1.0) {
      unsigned long long m = __ballot(alpha == amin);
      int blk = __ffsll((long long)m) - 1;
      double a = fmax(amin, 0.0);
      if (act && st == 0) x += a * (xe - x);
      if (lane == blk) { x = side < 0 ? lo : hi; st = side; }
      if (act) xs[lane] = x;
      WSYNC();
      continue;
    }
    x = xe;
    if (act) xs[lane] = x;
    WSYNC();
    if (!fixed) return GMR_STATUS_OK;
    // multipliers of the working set: g = H x + c
    double viol = 0.0;
    if (act && st != 0) {
      double g = ci;
      for (int j = 0; j < n; j++) g += H[lane * ldh + j] * xs[j];
      viol = st < 0 ? -g : g;
    }
    double vmax = wave_max(viol);
    if (!(vmax > dual_tol)) return GMR_STATUS_OK;
    unsigned long long m = __ballot(viol == vmax);
    int rel = __ffsll((long long)m) - 1;
    if (lane == rel) st = 0;
  }
  return GMR_STATUS_QP_MAXITER;
}

// ---------------------------------------------------------------------------------------------
// mj_integratePos with v = dq/dt (App. A.7): lane 0 the free joint, lane 6+h hinge h
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void integrate_wave(const IkLayout& L, double* sm, double dt, int lane) {
  double* q = sm + L.q;
  const double* dq = sm + L.x;
  if (lane == 0) {
    double v[6];
#pragma unroll
    for (int i = 0; i < 6; i++) v[i] = dq[i] / dt;
    q[0] += dt * v[0]; q[1] += dt * v[1]; q[2] += dt * v[2];
    double n = sqrt(v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
    d4 quat = qnormalize(d4{q[3], q[4], q[5], q[6]});
    if (n >= 1e-15) {
      double inv = 1.0 / n;
      double ang = dt * n;
      if (ang != 0.0) quat = qmul(quat, axis_angle(d3{v[3] * inv, v[4] * inv, v[5] * inv}, ang));
    }
    q[3] = quat.w; q[4] = quat.x; q[5] = quat.y; q[6] = quat.z;
  } else if (lane >= 6 && lane < L.nv) {
    double v = dq[lane] / dt;
    q[7 + lane - 6] += dt * v;
  }
  WSYNC();
}

// ---------------------------------------------------------------------------------------------
// target preprocessing (motion_retarget.py:203-270).  lane b < nhuman.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void preprocess_wave(const IkLayout& L, double* sm, const short* is_foot, int human_root,
                                                double ground_offset, int flags, int lane) {
  const double* raw = sm + L.raw;
  double* tgt = sm + L.tgt;
  double z = INFINITY;
  d3 p = {0, 0, 0};
  d4 uq = {1, 0, 0, 0};
  const bool on = lane < L.nhum;
  if (on) {
    const double* in = raw + 7 * lane;
    const double* rp = raw + 7 * human_root;
    const double* sc = sm + L.scale;
    double sr = sc[human_root];
    d3 srp = {sr * rp[0], sr * rp[1], sr * rp[2]};
    if (lane == human_root) p = srp;
    else {
      double s = sc[lane];
      p = d3{(in[0] - rp[0]) * s + srp.x, (in[1] - rp[1]) * s + srp.y, (in[2] - rp[2]) * s + srp.z};
    }
    const double* qo = sm + L.quat_off + 4 * lane;
    const double* po = sm + L.pos_off + 3 * lane;
    d4 q = qnormalize(d4{in[3], in[4], in[5], in[6]});
    uq = qnormalize(qmul(q, qnormalize(d4{qo[0], qo[1], qo[2], qo[3]})));
    p = p + qrot(uq, d3{po[0], po[1], po[2]});
    if (is_foot[lane] && p.x == p.x) z = p.z;
  }
  if (flags & GMR_FLAG_OFFSET_TO_GROUND) {
    double lowest = wave_min(z);
    p.z = p.z - lowest + ground_offset;
  }
  if (on) {
    double* o = tgt + 7 * lane;
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = uq.w; o[4] = uq.x; o[5] = uq.y; o[6] = uq.z;
  }
  WSYNC();
}

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ik_streams_kernel(const gmr_model_t* __restrict__ model,
                                                        const gmr_taskset_t* __restrict__ ts, IkLayout L, int S,
                                                        int T, const double* __restrict__ q0,
                                                        const double* __restrict__ human,
                                                        const int32_t* __restrict__ len, int flags,
                                                        double* __restrict__ q_out, int32_t* __restrict__ nsolve,
                                                        int32_t* __restrict__ status) {
  extern __shared__ __align__(16) double smem[];
  const int lane = threadIdx.x;
  const int s = blockIdx.x;
  if (s >= S) return;
  double* sm = smem;
  short* si = reinterpret_cast<short*>(smem + L.n_double);
  short* chain = si + L.i_chain;
  short* depth = si + L.i_depth;
  short* body_hinge = si + L.i_body_hinge;
  short* hinge_body = si + L.i_hinge_body;
  short* limited = si + L.i_limited;
  short* is_foot = si + L.i_is_foot;

  // ---- stage the constants into LDS --------------------------------------------------------
  const int nb = L.nb, nh = L.nh, nv = L.nv, nq = L.nq, nhum = L.nhum;
  for (int i = lane; i < nb; i += 64) {
    for (int a = 0; a < 3; a++) (sm + L.body_pos)[3 * i + a] = model->body_pos[i][a];
    for (int a = 0; a < 4; a++) (sm + L.body_quat)[4 * i + a] = model->body_quat[i][a];
    int h = model->body_hinge[i];
    for (int a = 0; a < 3; a++) (sm + L.axis)[3 * i + a] = h >= 0 ? model->hinge_axis[h][a] : 0.0;
    depth[i] = (short)model->depth[i];
    body_hinge[i] = (short)h;
    for (int d = 0; d < L.maxd; d++) chain[i * L.maxd + d] = (short)model->chain[i][d];
  }
  for (int i = lane; i < nh; i += 64) {
    (sm + L.range_lo)[i] = model->range_lo[i];
    (sm + L.range_hi)[i] = model->range_hi[i];
    hinge_body[i] = (short)model->hinge_body[i];
    limited[i] = (short)model->limited[i];
  }
  for (int i = lane; i < nhum; i += 64) {
    (sm + L.scale)[i] = ts->scale[i];
    for (int a = 0; a < 3; a++) (sm + L.pos_off)[3 * i + a] = ts->pos_off[i][a];
    for (int a = 0; a < 4; a++) (sm + L.quat_off)[4 * i + a] = ts->quat_off[i][a];
    is_foot[i] = (short)ts->is_foot[i];
  }
  for (int st = 0; st < 2; st++) {
    short* tb = si + L.i_task_body[st];
    short* th = si + L.i_task_human[st];
    short* c0 = si + L.i_task_col0[st];
    short* nc = si + L.i_task_ncol[st];
    short* pt = si + L.i_pair_task[st];
    short* pd = si + L.i_pair_dof[st];
    short* pi = si + L.i_pair_index[st];
    for (int k = lane; k < L.K[st]; k += 64) {
      tb[k] = (short)ts->task_body[st][k];
      th[k] = (short)ts->task_human[st][k];
      c0[k] = (short)ts->task_col0[st][k];
      nc[k] = (short)ts->task_ncol[st][k];
      (sm + L.wpos[st])[k] = ts->w_pos[st][k];
      (sm + L.wrot[st])[k] = ts->w_rot[st][k];
    }
    for (int p = lane; p < L.P[st]; p += 64) {
      pt[p] = (short)ts->pair_task[st][p];
      pd[p] = (short)ts->pair_dof[st][p];
    }
    for (int i = lane; i < L.K[st] * nv; i += 64) pi[i] = (short)ts->pair_index[st][i / nv][i % nv];
  }
  const double damping = ts->damping, lm_damping = ts->lm_damping, tol = ts->tol, limit_gain = ts->limit_gain;
  const double ground_offset = ts->ground_offset, dt = model->timestep;
  const int max_iter = ts->max_iter, human_root = ts->human_root;
  const int use0 = ts->use_stage[0], use1 = ts->use_stage[1];

  for (int i = lane; i < nq; i += 64) (sm + L.q)[i] = q0[(size_t)s * nq + i];
  WSYNC();
  fk_wave(L, sm, chain, depth, body_hinge, lane);

  const int Ts = len ? min(len[s], T) : T;
  const size_t fstride = (size_t)nhum * 7;
  const double* hs = human + (size_t)s * T * fstride;
  int stat = GMR_STATUS_OK;
  // first frame's raw targets
  double r0 = 0.0, r1 = 0.0;
  if (Ts > 0) {
    if (lane < (int)fstride) r0 = hs[lane];
    if (lane + 64 < (int)fstride) r1 = hs[lane + 64];
  }
  for (int t = 0; t < Ts; t++) {
    if (lane < (int)fstride) (sm + L.raw)[lane] = r0;
    if (lane + 64 < (int)fstride) (sm + L.raw)[lane + 64] = r1;
    // prefetch the next frame (nhuman*7 <= 128 doubles): the loads stay in flight during the solve
    if (t + 1 < Ts) {
      const double* nx = hs + (size_t)(t + 1) * fstride;
      if (lane < (int)fstride) r0 = nx[lane];
      if (lane + 64 < (int)fstride) r1 = nx[lane + 64];
    }
    WSYNC();
    int ns0 = 0, ns1 = 0;
    if (stat == GMR_STATUS_OK) {
      preprocess_wave(L, sm, is_foot, human_root, ground_offset, flags, lane);
      for (int stage = 0; stage < 2; stage++) {
        if (!(stage == 0 ? use0 : use1)) continue;
        const short* tb = si + L.i_task_body[stage];
        const short* th = si + L.i_task_human[stage];
        const int K = L.K[stage];
        double curr = errors_wave(L, sm, tb, th, K, lane);
        int nsol = 0, num_iter = 0;
        for (;;) {
          build_qp_wave(L, sm, stage, tb, si + L.i_task_col0[stage], si + L.i_task_ncol[stage],
                        si + L.i_pair_task[stage], si + L.i_pair_dof[stage], si + L.i_pair_index[stage],
                        hinge_body, limited, damping, lm_damping, limit_gain, lane);
          int rc = solve_qp_wave(L, sm, lane);
          if (rc != GMR_STATUS_OK) { stat = rc; break; }
          integrate_wave(L, sm, dt, lane);
          fk_wave(L, sm, chain, depth, body_hinge, lane);
          double next = errors_wave(L, sm, tb, th, K, lane);
          nsol++;
          if (nsol > 1) num_iter++;
          if (!(curr - next > tol && num_iter < max_iter)) break;
          curr = next;
        }
        if (stage == 0) ns0 = nsol; else ns1 = nsol;
        if (stat != GMR_STATUS_OK) break;
      }
    }
    const size_t f = (size_t)s * T + t;
    for (int i = lane; i < nq; i += 64) q_out[f * nq + i] = (sm + L.q)[i];
    if (lane == 0) { nsolve[2 * f] = ns0; nsolve[2 * f + 1] = ns1; }
    WSYNC();
  }
  if (lane == 0) status[s] = stat;
}

}  // namespace gmr

// host-side launcher used by gmr_abi.hip
extern "C" hipError_t gmr_launch_ik_streams(const gmr_model_t* d_model, const gmr_taskset_t* d_ts,
                                            const gmr::IkLayout* L, int S, int T, const double* d_q0,
                                            const double* d_human, const int32_t* d_len, int flags,
                                            double* d_q_out, int32_t* d_nsolve, int32_t* d_status,
                                            hipStream_t stream) {
  if (S <= 0 || T <= 0) return hipSuccess;
  hipLaunchKernelGGL(gmr::ik_streams_kernel, dim3(S), dim3(64), L->smem_bytes, stream, d_model, d_ts, *L, S, T,
                     d_q0, d_human, d_len, flags, d_q_out, d_nsolve, d_status);
  return hipGetLastError();
}

extern "C" hipError_t gmr_ik_set_max_smem(int bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gmr::ik_streams_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
