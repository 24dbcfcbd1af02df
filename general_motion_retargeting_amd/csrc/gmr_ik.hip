// gmr_ik.hip -- the retargeting hot path (rows H2-H7 of SURVEY.md section 8a) as ONE gfx950 kernel.
//
// One workgroup per motion stream (clip / robot instance): one 64-lane wavefront (NW = 1, many streams) or a main
// wavefront plus three helpers (NW = 4, few streams).  Frames of a stream are sequentially dependent (the
// configuration is warm-started from the previous frame, reference motion_retarget.py:75,139-185), so the time
// loop runs on the device and the parallel width of a launch is the number of streams.  Everything a stream
// touches between two frames lives in LDS (at compile-time offsets, gmr_ik_layout.h) or in registers:
//
//   LDS constants  joint-local transforms (body pos/quat, hinge axes), limits, task tables, solve parameters
//                  (NW = 4: also the static H-assembly schedule; NW = 1 streams it from global memory)
//   LDS state      q, FK ping-pong (pos, quat per body), world hinge axes, targets, residuals e_k,
//                  -Jl^-1(e_k), weighted per-(task,dof) Jacobian columns, H, c, bounds
//   registers      the working copy of H during the factorisation: lane i holds row i
//
// Per frame only nhuman*7 doubles are read from HBM (coalesced, contiguous) and nq doubles are
// written, i.e. 1 072 B/frame for G1: the kernel is bound by the dependent FP64 chain, not by HBM
// ("Roofline").
//
// Lane mapping per phase
//   FK          lane = body; log2(depth) rounds of pointer jumping over the kinematic tree
//               (transform composition is associative), ping-pong buffers in LDS
//   residuals   lane = task: e_k = log(T_wb^-1 T_wt); DPP row reduction for |e|
//   Jl^-1       lane = task
//   Jacobian    lane = (task, ancestor dof) pair: weighted column W_k (-Jl^-1) J_body[:, d]
//   H           lane = owner of a set of H entries (static LPT schedule, gmr_ik_layout.h): the
//               Jacobian of a task is non-zero only on its root->frame path, so H is assembled
//               block-sparse (6k instead of 103k multiply-adds for G1), one store per entry
//   QP          tree-structured (gmr_ik_tree.h): the dofs split into <= 4 limbs and a trunk, H is block-arrowhead.
//               NW = 4: one limb per wavefront; NW = 1: one limb per 16-lane DPP row of the single wavefront.
//               Robots that do not decompose (NW = 1 only): lane = row of H, dense Cholesky with the row in
//               registers, pivot column broadcast through v_readlane, L^T through one LDS transpose.
//               Box constraints: block principal pivoting (all violated bounds / multipliers are exchanged at
//               once, Murty's single exchange as the finite-termination fallback), warm-started from the
//               previous solve's active set.
//   NW = 4      the helpers (i) turn each new FK state into the body Jacobians while the main wavefront
//               evaluates residuals and Jl^-1, (ii) share the weighted Jacobian columns, (iii) assemble H while
//               the main wavefront gathers c and the bounds, (iv) eliminate one limb each in the QP.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/gmr_hip.h"
#include "gmr_device_math.h"
#include "gmr_ik_layout.h"
#include "gmr_ik_prof.h"

namespace gmr {

// Synchronisation inside ONE wave's phases.  NW == 1: the block is the wave, __syncthreads() is the
// cheapest full fence.  NW > 1 (helper waves present): the main wave must not touch the workgroup
// barrier outside the helped phases; LDS operations of one wave execute in order, so a workgroup-scope
// fence (s_waitcnt + compiler ordering) is all that is needed between its own phases.
template <int NW>
__device__ __forceinline__ void wsync() {
  if (NW == 1) __syncthreads();
  else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}
#define WSYNC() wsync<NW>()

#ifndef GMR_IK_MIN_WAVES
#define GMR_IK_MIN_WAVES 1
#endif
enum { QP_DENSE = 0, QP_TREE_SMALL = 1, QP_TREE = 2 };   // solver carried by a kernel instance
enum { CMD_BUILD = 1, CMD_EXIT = 2, CMD_JBODY = 3 };   // BUILD: assemble the QP, then solve it together; JBODY: body Jacobians

// value of lane `src` (wave-uniform index) as a scalar operand
__device__ __forceinline__ double readlane_d(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// value of lane `addr / 4` (per-lane index), all lanes active
__device__ __forceinline__ double bpermute_d(int addr, double v) {
  int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
  int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double dot6(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

}  // namespace gmr
#include "gmr_ik_tree.h"
namespace gmr {

// ---------------------------------------------------------------------------------------------
// FK: mj_kinematics semantics (App. A.3), evaluated by pointer jumping.  lane b < nb.
// Result: (pos, quat) per body at sm[L.o.xa + 7 b], world hinge axes at sm[L.o.xaxis + 3 b].
// ---------------------------------------------------------------------------------------------
// A body lane's constants of the walk (parent-relative pose, hinge, source lanes of the jumping rounds): read once per
// launch (fk_lane), kept in registers -- at every evaluation they were an index -> address -> data chain through LDS.
struct FkLane {
  int dep, hinge, src[IK_MAX_HOPS];
  d4 bq; d3 bp, ax;
};

template <class LT>
__device__ __forceinline__ FkLane fk_lane(const LT& L, const double* sm, const short* hop, const short* depth,
                                          const short* body_hinge, int lane) {
  FkLane F;
  const int b = lane < L.nb ? lane : 0;
  F.dep = lane < L.nb ? depth[b] : 0;
  F.hinge = lane < L.nb ? body_hinge[b] : -1;
#pragma unroll
  for (int r = 0; r < IK_MAX_HOPS; r++) F.src[r] = r < L.nhop ? 4 * hop[r * L.o.cap.nb + b] : 0;   // byte address of the source lane
  const double* bq = sm + L.o.body_quat + 4 * b;
  const double* bp = sm + L.o.body_pos + 3 * b;
  const double* ax = sm + L.o.axis + 3 * b;
  F.bq = d4{bq[0], bq[1], bq[2], bq[3]};
  F.bp = d3{bp[0], bp[1], bp[2]};
  F.ax = d3{ax[0], ax[1], ax[2]};
  return F;
}

template <int NW, class LT>
__device__ __forceinline__ void fk_wave(const LT& L, double* sm, const FkLane& F, int lane, Prof& pr) {
  lane = fresh_lane(lane);
  PROF_BEGIN(pr);
  const int nb = L.nb;
  double* q = sm + L.o.q;
  d3 pos = {0, 0, 0};
  d4 quat = {1, 0, 0, 0};
  const int dep = F.dep;
  // round 0 input: transform of every body relative to its parent (body 0: world pose)
  if (lane < nb) {
    if (lane == 0) {
      quat = qnormalize(d4{q[3], q[4], q[5], q[6]});
      q[3] = quat.w; q[4] = quat.x; q[5] = quat.y; q[6] = quat.z;
      pos = d3{q[0], q[1], q[2]};
    } else {
      quat = F.bq;
      pos = F.bp;
      if (F.hinge >= 0) {
        const double* sc = sm + L.o.hsc + 2 * F.hinge;   // sin, cos of q[7 + h] / 2 (hinge_sincos / integrate_wave)
        const double s = sc[0], c = sc[1];
        quat = qmul(quat, d4{c, F.ax.x * s, F.ax.y * s, F.ax.z * s});
      }
    }
  }
  // after round r a body's transform is relative to its ancestor 2^(r+1) levels up (or the world).  The bodies are the
  // lanes of this one wavefront, so a round takes the ancestor's transform straight out of that lane's registers
  // (ds_bpermute: every lane is read as it was BEFORE the round, which is what pointer jumping needs): no staging
  // buffer, no fence, no dependent index -> address -> data chain through LDS.
#pragma unroll
  for (int r = 0; r < IK_MAX_HOPS; r++) {
    if (r >= L.nhop) break;
    const int src = F.src[r];
    const d3 pa = {bpermute_d(src, pos.x), bpermute_d(src, pos.y), bpermute_d(src, pos.z)};
    const d4 qa = {bpermute_d(src, quat.w), bpermute_d(src, quat.x), bpermute_d(src, quat.y), bpermute_d(src, quat.z)};
    if (lane < nb && dep >= (1 << r)) {
      pos = pa + qrot(qa, pos);
      quat = qmul(qa, quat);
    }
  }
  if (lane < nb) {
    quat = qnormalize(quat);
    double* o = sm + L.o.xa + 7 * lane;
    o[0] = pos.x; o[1] = pos.y; o[2] = pos.z; o[3] = quat.w; o[4] = quat.x; o[5] = quat.y; o[6] = quat.z;
    if (F.hinge >= 0) {
      d3 aw = qrot(quat, F.ax);
      double* xa = sm + L.o.xaxis + 3 * lane;
      xa[0] = aw.x; xa[1] = aw.y; xa[2] = aw.z;
    }
  }
  WSYNC();
  PROF_END(pr, PH_FK);
}

// ---------------------------------------------------------------------------------------------
// residuals of the stage's tasks (mink FrameTask.compute_error) and their unweighted norm
// (motion_retarget.py:188-200).  lane k < K.  Returns E in every lane.
// ---------------------------------------------------------------------------------------------
template <int NW, class LT>
__device__ __forceinline__ double errors_wave(const LT& L, double* sm, const short* task_body,
                                              const short* task_human, int K, int lane, Prof& pr) {
  lane = fresh_lane(lane);
  PROF_BEGIN(pr);
  double ss = 0.0;
  if (lane < K) {
    int b = task_body[lane], h = task_human[lane];
    const double* x = sm + L.o.xa + 7 * b;
    const double* tg = sm + L.o.tgt + 7 * h;
    double e[6], aux[5];
    se3_log_rel5(d3{x[0], x[1], x[2]}, d4{x[3], x[4], x[5], x[6]}, d3{tg[0], tg[1], tg[2]},
                 d4{tg[3], tg[4], tg[5], tg[6]}, e, aux);
    double* eo = sm + L.o.e + 6 * lane;
#pragma unroll
    for (int r = 0; r < 6; r++) { eo[r] = e[r]; ss += e[r] * e[r]; }
    double* ao = sm + L.o.eaux + 5 * lane;     // a, sin|w|, cos|w|, |w|, 1/|w|: reused by the Jl^-1 phase
#pragma unroll
    for (int r = 0; r < 5; r++) ao[r] = aux[r];
  }
  ss = row0_sum(ss);                 // tasks live in lanes 0..K-1, K <= 16
  WSYNC();
  PROF_END(pr, PH_ERR);
  return sqrt(ss);
}

// ---------------------------------------------------------------------------------------------
// QP assembly (mink compute_qp_objective + ConfigurationLimit; App. A.4-A.6), in four phases so
// that helper waves can share the two wide ones (Jacobian columns, H entries)
// ---------------------------------------------------------------------------------------------
struct StageTabs {
  const short* task_body; const short* task_human; const short* pair_task; const short* pair_dof;
  const short* pair_index; const uint2* items;      // 64-bit schedule items (gmr_ik_layout.h)
};

// (a) lane = 16 j + task: column j of M_k = -Jl^-1(e_k) (blocks -A, -B; three lanes per task), weighted residual;
// returns the LM term mu
template <int NW, class LT>
__device__ __forceinline__ double jlog_phase(const LT& L, double* sm, int stage, double lm_damping, int lane,
                                             Prof& pr) {
  lane = fresh_lane(lane);
  PROF_BEGIN(pr);
  const int K = L.K[stage];
  const double* wpos = sm + L.o.wpos[stage];
  const double* wrot = sm + L.o.wrot[stage];
  double mu = 0.0;
  const int k = lane & 15, j = lane >> 4;
  if (k < K && j < 3) {
    const double* e = sm + L.o.e + 6 * k;
    double ee[6];
#pragma unroll
    for (int r = 0; r < 6; r++) ee[r] = e[r];
    const double* ax = sm + L.o.eaux + 5 * k;
    const double aux[5] = {ax[0], ax[1], ax[2], ax[3], ax[4]};
    double Ac[3], Bc[3];
    se3_jlinv_col5(ee, aux, j, Ac, Bc);
    double* M = sm + L.o.M + 18 * k + j;
#pragma unroll
    for (int i = 0; i < 3; i++) { M[3 * i] = -Ac[i]; M[9 + 3 * i] = -Bc[i]; }
    if (j == 0) {
      double wp = wpos[k], wr = wrot[k];
      double* we = sm + L.o.we + 6 * k;
#pragma unroll
      for (int r = 0; r < 6; r++) {
        double v = (r < 3 ? wp : wr) * ee[r];
        we[r] = v;
        mu += v * v;
      }
    }
  }
  mu = lm_damping * row0_sum(mu);
  WSYNC();
  PROF_END(pr, PH_JLOG);
  return mu;
}

// (b) virtual lane = (task, dof) pair: weighted task-Jacobian column W_k * (-Jl^-1(e_k)) * J_body[:, d]
template <class LT>
__device__ __forceinline__ void pairs_phase(const LT& L, double* sm, int stage, const StageTabs& tb,
                                            const short* hinge_body, int vlane, int nvl) {
  const int P = L.P[stage];
  const double* wpos = sm + L.o.wpos[stage];
  const double* wrot = sm + L.o.wrot[stage];
  double* Jw = sm + L.o.Jw;
  double* cpart = sm + L.o.cpart;
  const double* X = sm + L.o.xa;
  for (int p = vlane; p < P; p += nvl) {
    const unsigned info = (unsigned short)tb.pair_task[p];       // [3:0] task, [9:4] dof, [15:10] task body
    const int k = info & 15u, dof = (info >> 4) & 63u, b = info >> 10;
    const int c = tb.pair_dof[p];                                  // body of hinge dof - 6 (0 for the base)
    d3 pb = {X[7 * b], X[7 * b + 1], X[7 * b + 2]};
    d4 qb = {X[7 * b + 3], X[7 * b + 4], X[7 * b + 5], X[7 * b + 6]};
    d3 lin, ang;
    if (dof < 3) {
      lin = d3{dof == 0 ? 1.0 : 0.0, dof == 1 ? 1.0 : 0.0, dof == 2 ? 1.0 : 0.0};
      ang = d3{0.0, 0.0, 0.0};
    } else if (dof < 6) {
      d4 q0 = {X[3], X[4], X[5], X[6]};
      int a = dof - 3;
      ang = qrot(q0, d3{a == 0 ? 1.0 : 0.0, a == 1 ? 1.0 : 0.0, a == 2 ? 1.0 : 0.0});
      lin = cross(ang, pb - d3{X[0], X[1], X[2]});
    } else {
      const double* xa = sm + L.o.xaxis + 3 * c;
      ang = d3{xa[0], xa[1], xa[2]};
      lin = cross(ang, pb - d3{X[7 * c], X[7 * c + 1], X[7 * c + 2]});
    }
    d3 jl = qrot_inv(qb, lin), ja = qrot_inv(qb, ang);   // body-frame Jacobian column
    const double* M = sm + L.o.M + 18 * k;
    const double* we = sm + L.o.we + 6 * k;
    double wp = wpos[k], wr = wrot[k];
    double* o = Jw + 6 * p;
    double cp = 0.0;
    // [ -A  -B ] [jl]      rows 0..2 (scaled by w_pos)
    // [  0  -A ] [ja]      rows 3..5 (scaled by w_rot)
#pragma unroll
    for (int r = 0; r < 3; r++) {
      double top = M[3 * r] * jl.x + M[3 * r + 1] * jl.y + M[3 * r + 2] * jl.z + M[9 + 3 * r] * ja.x +
                   M[9 + 3 * r + 1] * ja.y + M[9 + 3 * r + 2] * ja.z;
      double bot = M[3 * r] * ja.x + M[3 * r + 1] * ja.y + M[3 * r + 2] * ja.z;
      top *= wp; bot *= wr;
      o[r] = top;
      o[3 + r] = bot;
      cp += top * we[r] + bot * we[3 + r];
    }
    cpart[p] = cp;   // this column's contribution to c = sum_k (W J_k)^T (W e_k)
  }
  if (vlane < 6) Jw[6 * L.o.cap.p + vlane] = 0.0;   // the row that schedule items without a term read
  if (vlane == 6) cpart[L.o.cap.p] = 0.0;           // and the c share absent (task, dof) pairs gather
}

// (b1) 4-wavefront shape, helpers only, right after the FK and concurrently with the main wavefront's residual
// and Jl^-1 phases: the body-frame Jacobian column of every (task, dof) pair -- it depends on the FK state alone.
// Stored where the weighted column will go (Jw[p][6] = [jl; ja]).
template <class LT>
__device__ __forceinline__ void jbody_phase(const LT& L, double* sm, int stage, const StageTabs& tb, int vlane, int nvl) {
  const int P = L.P[stage];
  double* Jw = sm + L.o.Jw;
  const double* X = sm + L.o.xa;
  for (int p = vlane; p < P; p += nvl) {
    const unsigned info = (unsigned short)tb.pair_task[p];       // [3:0] task, [9:4] dof, [15:10] task body
    const int dof = (info >> 4) & 63u, b = info >> 10;
    const int c = tb.pair_dof[p];
    d3 pb = {X[7 * b], X[7 * b + 1], X[7 * b + 2]};
    d4 qb = {X[7 * b + 3], X[7 * b + 4], X[7 * b + 5], X[7 * b + 6]};
    d3 lin, ang;
    if (dof < 3) {
      lin = d3{dof == 0 ? 1.0 : 0.0, dof == 1 ? 1.0 : 0.0, dof == 2 ? 1.0 : 0.0};
      ang = d3{0.0, 0.0, 0.0};
    } else if (dof < 6) {
      d4 q0 = {X[3], X[4], X[5], X[6]};
      int a = dof - 3;
      ang = qrot(q0, d3{a == 0 ? 1.0 : 0.0, a == 1 ? 1.0 : 0.0, a == 2 ? 1.0 : 0.0});
      lin = cross(ang, pb - d3{X[0], X[1], X[2]});
    } else {
      const double* xa = sm + L.o.xaxis + 3 * c;
      ang = d3{xa[0], xa[1], xa[2]};
      lin = cross(ang, pb - d3{X[7 * c], X[7 * c + 1], X[7 * c + 2]});
    }
    d3 jl = qrot_inv(qb, lin), ja = qrot_inv(qb, ang);
    double* o = Jw + 6 * p;
    o[0] = jl.x; o[1] = jl.y; o[2] = jl.z; o[3] = ja.x; o[4] = ja.y; o[5] = ja.z;
  }
}

// (b2) all wavefronts, after Jl^-1: Jw[p] = W_k (-Jl^-1(e_k)) Jb[p] in place, and the column's share of c
template <class LT>
__device__ __forceinline__ void pairs_from_jbody(const LT& L, double* sm, int stage, const StageTabs& tb, int vlane, int nvl,
                                                 const uint32_t* zero_off, int k_first) {
  // k_first: task of pair `vlane` (the first trip's, read once per launch by the caller; P <= nvl: the only trip)
  const int P = L.P[stage];
  const double* wpos = sm + L.o.wpos[stage];
  const double* wrot = sm + L.o.wrot[stage];
  double* Jw = sm + L.o.Jw;
  double* cpart = sm + L.o.cpart;
  for (int p = vlane; p < P; p += nvl) {
    const int k = p == vlane ? k_first : ((unsigned short)tb.pair_task[p] & 15u);
    double* o = Jw + 6 * p;
    const d3 jl = {o[0], o[1], o[2]}, ja = {o[3], o[4], o[5]};
    const double* M = sm + L.o.M + 18 * k;
    const double* we = sm + L.o.we + 6 * k;
    const double wp = wpos[k], wr = wrot[k];
    double cp = 0.0;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      double top = M[3 * r] * jl.x + M[3 * r + 1] * jl.y + M[3 * r + 2] * jl.z + M[9 + 3 * r] * ja.x +
                   M[9 + 3 * r + 1] * ja.y + M[9 + 3 * r + 2] * ja.z;
      double bot = M[3 * r] * ja.x + M[3 * r + 1] * ja.y + M[3 * r + 2] * ja.z;
      top *= wp; bot *= wr;
      o[r] = top;
      o[3 + r] = bot;
      cp += top * we[r] + bot * we[3 + r];
    }
    cpart[p] = cp;
  }
  if (vlane < 6) Jw[6 * L.o.cap.p + vlane] = 0.0;   // the row that schedule items without a term read
  if (vlane == 6) cpart[L.o.cap.p] = 0.0;           // and the c share absent (task, dof) pairs gather
  if (vlane >= 64 && vlane - 64 < L.nzero[stage])   // the H cells that half entries are added to (a helper lane each)
    *reinterpret_cast<double*>(reinterpret_cast<char*>(sm + L.o.H) + zero_off[vlane - 64]) = 0.0;
}

// (c) lane = dof: gather c; bounds of the limited hinges (mink ConfigurationLimit)
template <class LT>
__device__ __forceinline__ void cvec_phase(const LT& L, double* sm, int stage, const StageTabs& tb,
                                           const short* limited, double limit_gain, int lane) {
  lane = fresh_lane(lane);
  const int nv = L.nv;
  const char* cpart = reinterpret_cast<const char*>(sm + L.o.cpart);
  if (lane < nv) {
    // the table holds byte offsets into cpart; absent pairs and absent tasks name its zero slot: no compare, no select
    int off[GMR_MAX_TASKS];
#pragma unroll
    for (int k = 0; k < GMR_MAX_TASKS; k++) off[k] = (unsigned short)tb.pair_index[k * L.o.nvp + lane];
    double cc = 0.0;
#pragma unroll
    for (int k = 0; k < GMR_MAX_TASKS; k++) cc += *reinterpret_cast<const double*>(cpart + off[k]);
    (sm + L.o.c)[lane] = cc;
    double lo = -INFINITY, hi = INFINITY;
    if (lane >= 6 && limited[lane - 6]) {
      double th = (sm + L.o.q)[7 + lane - 6];
      hi = limit_gain * ((sm + L.o.range_hi)[lane - 6] - th);
      lo = -limit_gain * (th - (sm + L.o.range_lo)[lane - 6]);
    }
    (sm + L.o.lo)[lane] = lo;
    (sm + L.o.hi)[lane] = hi;
  }
}

// (d) H: every (virtual) lane sums the terms of the entries it owns (static schedule) and stores each
// once.  The schedule is padded to a wave-uniform number of slots and stored [slot][lane]; two terms per
// trip, operands of both loaded (as 16-B pieces) before any store (Jw read-only, H write-only).
struct __attribute__((aligned(16))) dd2 { double x, y; };   // Jw rows start on 16-byte boundaries: one b128 read per piece
__device__ __forceinline__ double dot6v(const double* a, const double* b) {
  const dd2* pa = reinterpret_cast<const dd2*>(a);
  const dd2* pb = reinterpret_cast<const dd2*>(b);
  dd2 a0 = pa[0], a1 = pa[1], a2 = pa[2], b0 = pb[0], b1 = pb[1], b2 = pb[2];
  return (a0.x * b0.x + a0.y * b0.y + a1.x * b1.x) + (a1.y * b1.y + a2.x * b2.x + a2.y * b2.y);
}
__device__ __forceinline__ double item_dot(const char* Jb, uint32_t lo) {
  return dot6v(reinterpret_cast<const double*>(Jb + (lo & 0xffffu)), reinterpret_cast<const double*>(Jb + (lo >> 16)));
}
__device__ __forceinline__ void item_store(char* Hb, uint32_t hi, double acc, double diag) {
  const double v = acc + ((hi & (1u << 30)) ? diag : 0.0);
  *reinterpret_cast<double*>(Hb + (hi & 0x7fffu)) = v;
  *reinterpret_cast<double*>(Hb + ((hi >> 15) & 0x7fffu)) = v;
}

// a half entry's sum is ADDED to its (zeroed) cells: two addends per cell, so the order does not matter
__device__ __forceinline__ void item_add(char* Hb, uint32_t hi, double acc, double diag) {
  const double v = acc + ((hi & (1u << 30)) ? diag : 0.0);
  const uint32_t o1 = hi & 0x7fffu, o2 = (hi >> 15) & 0x7fffu;
  atomicAdd(reinterpret_cast<double*>(Hb + o1), v);
  if (o2 != o1) atomicAdd(reinterpret_cast<double*>(Hb + o2), v);
}

template <class LT>
__device__ __forceinline__ void hacc_phase(const LT& L, double* sm, int stage, const StageTabs& tb, double diag,
                                           int vlane, const uint2* first) {   // first: the first trip's items, loaded by the caller
  const int nl = L.nlanes, ntrip = L.ntrip[stage];
  const bool paired = vlane < L.pair_lanes;       // the whole first helper wavefront, or nobody
  char* __restrict__ Hb = reinterpret_cast<char*>(sm + L.o.H);
  const char* __restrict__ Jb = reinterpret_cast<const char*>(sm + L.o.Jw);
  const uint2* items = tb.items + vlane;
  double acc = 0.0;
  uint2 n[4];
#pragma unroll
  for (int k = 0; k < 4; k++) n[k] = first[k];
  for (int it = 0; it < ntrip; it += 4) {                  // four slots per trip: their row reads are in flight together
    uint2 w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = n[k];
    if (it + 4 < ntrip) {
#pragma unroll
      for (int k = 0; k < 4; k++) n[k] = items[(it + 4 + k) * nl];
    }
    double s[4];
#pragma unroll
    for (int k = 0; k < 4; k++) s[k] = item_dot(Jb, w[k].x);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      acc += s[k];
      if (paired) {   // wave-uniform: lanes 2i / 2i+1 hold the two halves of one entry, closed in the same slot
        if (__any((int)(w[k].y >> 31))) {
          double tot = acc + dpp_swap_pairs(acc);           // own + neighbour (commutative: the same bits in both lanes)
          if ((w[k].y >> 31) && !(vlane & 1)) {
            if (vlane < L.atomic_lanes[stage]) item_add(Hb, w[k].y, tot, diag); else item_store(Hb, w[k].y, tot, diag);
          }
          if (w[k].y >> 31) acc = 0.0;
        }
      } else if (w[k].y >> 31) {
        item_store(Hb, w[k].y, acc, diag);
        acc = 0.0;
      }
    }
  }
}

// (d') one wavefront per stream: the schedule is read from the global image (the same words for every
// stream on the CU: vector-L1 hits) instead of LDS, four slots per trip with the next trip in flight.
template <class LT>
__device__ __forceinline__ void hacc_phase_g(const LT& L, double* sm, int stage,
                                             const uint2* __restrict__ gitems, double diag, int lane) {
  const int ntrip = L.ntrip[stage];                         // a multiple of 4, 64 lanes
  char* __restrict__ Hb = reinterpret_cast<char*>(sm + L.o.H);
  const char* __restrict__ Jb = reinterpret_cast<const char*>(sm + L.o.Jw);
  const uint2* p = gitems + lane;
  uint2 n[4];
#pragma unroll
  for (int k = 0; k < 4; k++) n[k] = p[k * 64];
  double acc = 0.0;
  for (int it = 0; it < ntrip; it += 4) {
    uint2 w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = n[k];
    if (it + 4 < ntrip) {
#pragma unroll
      for (int k = 0; k < 4; k++) n[k] = p[(it + 4 + k) * 64];
    }
    double s[4];
#pragma unroll
    for (int k = 0; k < 4; k++) s[k] = item_dot(Jb, w[k].x);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      acc += s[k];
      if (w[k].y >> 31) {
        item_store(Hb, w[k].y, acc, diag);
        acc = 0.0;
      }
    }
  }
}

// the whole assembly as seen from the MAIN wave.  NW == 1: all four phases by this wave.  NW > 1:
// the Jacobian columns are shared by all NW waves, then the helpers assemble H (64*(NW-1) virtual
// lanes) while this wave gathers c and the bounds; three workgroup barriers per assembly.
template <int NW, class LT>
__device__ __forceinline__ void build_qp_main(const LT& L, double* sm, int stage, const StageTabs& tb,
                                              const short* hinge_body, const short* limited, int* ctl, int& epoch,
                                              double damping, double lm_damping, double limit_gain, int lane,
                                              int pk_main, Prof& pr) {   // pk_main: task of this lane's pair in this table
  double mu = jlog_phase<NW>(L, sm, stage, lm_damping, lane, pr);
  const double diag = damping + mu;
  if (NW == 1) {
    PROF_BEGIN(pr);
    pairs_phase(L, sm, stage, tb, hinge_body, lane, 64);
    WSYNC();
    PROF_END(pr, PH_PAIRS);
    PROF_BEGIN(pr);
    cvec_phase(L, sm, stage, tb, limited, limit_gain, lane);
    PROF_END(pr, PH_CVEC);
    PROF_BEGIN(pr);
    hacc_phase_g(L, sm, stage, tb.items, diag, lane);
    WSYNC();
    PROF_END(pr, PH_HACC);
  } else {
    PROF_BEGIN(pr);
    if (lane == 0) { int* c = ctl + 2 * (epoch & 1); c[0] = CMD_BUILD; c[1] = stage; (sm + L.o.scal)[0] = diag; }
    epoch++;
    __syncthreads();                      // B1: helpers see the command; M / we and the body Jacobians are final
    pairs_from_jbody(L, sm, stage, tb, lane, 64 * NW, nullptr, pk_main);      // (the main wavefront's lanes zero nothing)
    __syncthreads();                      // B2: all Jacobian columns written
    PROF_END(pr, PH_PAIRS);
    PROF_BEGIN(pr);
    cvec_phase(L, sm, stage, tb, limited, limit_gain, lane);
    PROF_END(pr, PH_CVEC);
    PROF_BEGIN(pr);
    __syncthreads();                      // B3: H complete (helpers)
    PROF_END(pr, PH_HACC);
  }
}

// helper waves (NW > 1): serve assembly requests until the main wave says EXIT
template <int NW, int QP, class LT>
__device__ __forceinline__ void helper_loop(const LT& L, double* sm, const uint32_t* sw, const short* si,
                                            const short* hinge_body, const int* ctl, int wave, int lane,
                                            Prof& hp) {
  TreeState bs = {0ull, 0ull};            // bound sets of the QP (identical in every wavefront)
  const TreeRows<16> rows_s = tree_rows<7, 9, false>(L, si, wave, lane);
  const TreeRows<18> rows_l = tree_rows<8, 10, false>(L, si, wave, lane);
  int pk[2];                              // task of this lane's (task, dof) pair in either table
#pragma unroll
  for (int st = 0; st < 2; st++) pk[st] = (unsigned short)(si + L.o.i_pair_task[st])[min(wave * 64 + lane, L.o.cap.p - 1)] & 15u;
  for (int epoch = 0;; epoch++) {         // command n sits in mailbox slot n & 1 (see the main wavefront)
    PROF_BEGIN(hp);
    __syncthreads();                      // B1 (or the EXIT barrier)
    PROF_END(hp, PH_PRE);                 // (helper stamps reuse the slots: PRE = idle at B1)
    const int cmd = ctl[2 * (epoch & 1)];
    if (cmd == CMD_EXIT) return;
    const int stage = ctl[2 * (epoch & 1) + 1];
    StageTabs tb = {si + L.o.i_task_body[stage], si + L.o.i_task_human[stage], si + L.o.i_pair_task[stage],
                    si + L.o.i_pair_dof[stage], si + L.o.i_pair_index[stage],
                    reinterpret_cast<const uint2*>(sw + L.w_items[stage])};
    if (cmd == CMD_JBODY) {               // the main wavefront is evaluating the residuals meanwhile
      PROF_BEGIN(hp);
      jbody_phase(L, sm, stage, tb, (wave - 1) * 64 + lane, 64 * (NW - 1));
      PROF_END(hp, PH_PAIRS);
      continue;
    }
    const double diag = (sm + L.o.scal)[0];
    // the first trip of this wavefront's H share: constants of the stage, asked for now so that they have arrived when
    // the barrier after the column phase opens (instead of starting an item -> row -> FMA chain there)
    uint2 first[4];
#pragma unroll
    for (int k = 0; k < 4; k++) first[k] = tb.items[k * L.nlanes + (wave - 1) * 64 + lane];
    PROF_BEGIN(hp);
    pairs_from_jbody(L, sm, stage, tb, wave * 64 + lane, 64 * NW, sw + L.o.w_zero + stage * IK_MAX_ZERO, pk[stage]);
    PROF_END(hp, PH_PAIRS);
    PROF_BEGIN(hp);
    __syncthreads();                      // B2
    PROF_END(hp, PH_CVEC);                // wait at B2
    PROF_BEGIN(hp);
    hacc_phase(L, sm, stage, tb, diag, (wave - 1) * 64 + lane, first);
    PROF_END(hp, PH_HACC);
    PROF_BEGIN(hp);
    __syncthreads();                      // B3
    PROF_END(hp, PH_JLOG);                // wait at B3
    // NW > 1 is only launched for robots that decompose
    if (QP == QP_TREE_SMALL) (void)solve_qp_tree<7, 9, false, true>(L, sm, const_cast<uint32_t*>(sw), si, wave, lane, bs, hp, rows_s);
    else (void)solve_qp_tree<8, 10, false, false>(L, sm, const_cast<uint32_t*>(sw), si, wave, lane, bs, hp, rows_l);
  }
}

// ---------------------------------------------------------------------------------------------
// box-constrained strictly convex QP  min 1/2 x^T H x + c^T x, lo <= x <= hi  (App. A.6): same
// unique minimiser as DAQP behind qpsolvers.  Block principal pivoting over the bound sets; each
// round is one dense Cholesky of H with the rows/columns of the fixed variables replaced by the
// identity.  `st` (0 free, -1 at lower, +1 at upper) is carried from solve to solve (warm start).
// Returns 0 ok / <0 failure; the solution is left in sm[L.o.x].
// ---------------------------------------------------------------------------------------------
template <int NVP, int NW, class LT>
__device__ __forceinline__ int solve_qp_regs(const LT& L, double* sm, int lane, int& st, Prof& pr) {
  constexpr int LDK = NVP + 1;
  const int n = L.nv, ldh = L.o.ldh;
  const double* H = sm + L.o.H;
  double* Kt = sm + L.o.Kt;
  const bool act = lane < n;
  const bool row = lane < NVP;
  const double lo = act ? (sm + L.o.lo)[lane] : 0.0, hi = act ? (sm + L.o.hi)[lane] : 0.0;
  const double ci = act ? (sm + L.o.c)[lane] : 0.0;
  const double* Hrow = H + (act ? lane : 0) * ldh;
  if (!act || (st < 0 && !(lo > -INFINITY)) || (st > 0 && !(hi < INFINITY))) st = 0;
  const double dual_tol = 1e-13 * (1.0 + rows3_max(fabs(ci)));   // dofs live in lanes 0..nv-1, nv <= 48
  const double ptol_lo = 1e-12 * (1.0 + fabs(lo)), ptol_hi = 1e-12 * (1.0 + fabs(hi));
  int pcount = 3, ninf_best = NVP + 1;
  for (int it = 0; it < 100; it++) {
    PROF_BEGIN(pr);
    PROF_COUNT(pr, PH_NFACT);
    const unsigned long long fixedm = __ballot(act && st != 0);
    const double xfix = st < 0 ? lo : (st > 0 ? hi : 0.0);
    // right-hand side: fixed rows keep their bound, free rows get -c_F - H_FA x_A
    double rhs = act ? (st != 0 ? xfix : -ci) : 0.0;
    {
      unsigned long long m = fixedm;
      while (m) {
        int j = __ffsll((long long)m) - 1;
        m &= m - 1;
        double xj = readlane_d(xfix, j);
        if (act && st == 0) rhs -= Hrow[j] * xj;
      }
    }
    // lane i's row of the working matrix: H_ij for free i, free j <= i; identity for fixed / padding
    const bool self_fixed = (st != 0) || !act;
    double r[NVP];
#pragma unroll
    for (int k = 0; k < NVP; k++) {
      const bool kfixed = (k >= n) || ((fixedm >> k) & 1ull);  // wave-uniform
      double h = (act && k < n && k <= lane) ? Hrow[k] : 0.0;
      double v = (self_fixed || kfixed) ? 0.0 : h;
      if (k == lane && self_fixed) v = 1.0;
      r[k] = v;
    }
    PROF_END(pr, PH_KBUILD);
    PROF_BEGIN(pr);
    // right-looking Cholesky, row per lane, pivot column broadcast by readlane
    double mydinv = 1.0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NVP; j++) {
      double dj = readlane_d(r[j], j);
      bad = bad || !(dj > 0.0);
      double dinv = fast_rsqrt(dj);
      double l = r[j] * dinv;
      r[j] = l;
      if (lane == j) mydinv = dinv;
      // trailing update, four columns per group: the four SGPR pairs are read first so that the
      // VALU-writes-SGPR wait states of one pair are covered by the reads of the next
#pragma unroll
      for (int k0 = j + 1; k0 < NVP; k0 += 4) {
        double lk[4];
#pragma unroll
        for (int u = 0; u < 4; u++) lk[u] = (k0 + u < NVP) ? readlane_d(l, k0 + u) : 0.0;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (k0 + u < NVP) r[k0 + u] = fma(-l, lk[u], r[k0 + u]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (bad) return GMR_STATUS_QP_FAILED;
    PROF_END(pr, PH_CHOL);
    PROF_BEGIN(pr);
    // forward substitution L y = rhs (column sweep)
    double b = rhs;
#pragma unroll
    for (int j = 0; j < NVP; j++) {
      double yj = readlane_d(b * mydinv, j);
      double upd = fma(-r[j], yj, b);
      b = lane == j ? yj : (lane > j ? upd : b);
    }
    // L^T through LDS: lane i needs column i of L
    if (row) {
#pragma unroll
      for (int k = 0; k < NVP; k++) Kt[lane * LDK + k] = r[k];
    }
    WSYNC();
    double lt[NVP];
#pragma unroll
    for (int j = 0; j < NVP; j++) lt[j] = row ? Kt[j * LDK + lane] : 0.0;
#pragma unroll
    for (int j = NVP - 1; j >= 0; j--) {
      double xj = readlane_d(b * mydinv, j);
      double upd = fma(-lt[j], xj, b);
      b = lane == j ? xj : (lane < j ? upd : b);
    }
    double x = b;
    WSYNC();  // Kt is rewritten by the next round
    PROF_END(pr, PH_SUBST);
    PROF_BEGIN(pr);
    // violated bounds (free set) and multipliers (fixed set): g = H x + c
    double g = 0.0;
    if (fixedm) {
      g = ci;
#pragma unroll
      for (int j = 0; j < NVP; j++) {
        double xj = readlane_d(x, j);
        if (j < n) g += (act ? Hrow[j] : 0.0) * xj;
      }
    }
    int newst = st;
    bool viol = false;
    if (act) {
      if (st == 0) {
        if (x < lo - ptol_lo) { viol = true; newst = -1; }
        else if (x > hi + ptol_hi) { viol = true; newst = 1; }
      } else if (st < 0) {
        if (g < -dual_tol) { viol = true; newst = 0; }
      } else {
        if (g > dual_tol) { viol = true; newst = 0; }
      }
    }
    const unsigned long long vm = __ballot(viol);
    PROF_END(pr, PH_MULT);
    if (vm == 0ull) {
      if (act) (sm + L.o.x)[lane] = fmin(fmax(x, lo), hi);
      WSYNC();
      return GMR_STATUS_OK;
    }
    const int ninf = __popcll(vm);
    if (ninf < ninf_best) { ninf_best = ninf; pcount = 3; st = newst; }
    else if (pcount > 0) { pcount--; st = newst; }
    else {  // Murty: exchange only the highest-index violated variable
      int top = 63 - __clzll((long long)vm);
      if (lane == top) st = newst;
    }
  }
  return GMR_STATUS_QP_MAXITER;
}

// ---------------------------------------------------------------------------------------------
// (sin, cos) of every hinge's half angle for a configuration that did not come out of integrate_wave (q0)
template <class LT>
__device__ __forceinline__ void hinge_sincos(const LT& L, double* sm, int lane) {
  if (lane >= 6 && lane < L.nv) {
    double s, c;
    sincos_small(0.5 * (sm + L.o.q)[7 + lane - 6], &s, &c);
    double* sc = sm + L.o.hsc + 2 * (lane - 6);
    sc[0] = s; sc[1] = c;
  }
}

// mj_integratePos with v = dq/dt (App. A.7): lane 0 the free joint, lane 6+h hinge h
// ---------------------------------------------------------------------------------------------
template <int NW, class LT>
__device__ __forceinline__ void integrate_wave(const LT& L, double* sm, double dt, int lane, Prof& pr) {
  lane = fresh_lane(lane);
  PROF_BEGIN(pr);
  double* q = sm + L.o.q;
  const double* dq = sm + L.o.x;
  // ONE sincos serves both uses of this phase: lane 0 needs the half angle of the base rotation increment,
  // the hinge lanes the half angle of their new joint value (consumed by the next FK, which then needs none)
  const bool base = lane == 0, hinge = lane >= 6 && lane < L.nv;
  double half = 0.0, inv = 0.0;
  bool rotate = false;
  if (base) {
    // v = dq / dt followed by dt * v (solve_ik / mj_integratePos) is dq to 1 ulp: integrate dq directly
    q[0] += dq[0]; q[1] += dq[1]; q[2] += dq[2];
    const double n2 = dq[3] * dq[3] + dq[4] * dq[4] + dq[5] * dq[5];
    rotate = n2 >= 1e-30 * dt * dt;
    if (rotate) { inv = fast_rsqrt(n2); half = 0.5 * (n2 * inv); }
  } else if (hinge) {
    const double th = q[7 + lane - 6] + dq[lane];
    q[7 + lane - 6] = th;
    half = 0.5 * th;
  }
  double s, c;
  sincos_small(half, &s, &c);
  if (base) {
    d4 quat = qnormalize(d4{q[3], q[4], q[5], q[6]});
    if (rotate) quat = qmul(quat, d4{c, dq[3] * inv * s, dq[4] * inv * s, dq[5] * inv * s});
    q[3] = quat.w; q[4] = quat.x; q[5] = quat.y; q[6] = quat.z;
  } else if (hinge) {
    double* sc = sm + L.o.hsc + 2 * (lane - 6);
    sc[0] = s; sc[1] = c;
  }
  WSYNC();
  PROF_END(pr, PH_INTEG);
}

// ---------------------------------------------------------------------------------------------
// target preprocessing (motion_retarget.py:203-270).  lane b < nhuman.
// ---------------------------------------------------------------------------------------------
template <int NW, class LT>
__device__ __forceinline__ void preprocess_wave(const LT& L, double* sm, const short* is_foot, int human_root,
                                                double ground_offset, int flags, int lane, Prof& pr) {
  lane = fresh_lane(lane);
  PROF_BEGIN(pr);
  const double* raw = sm + L.o.raw;
  double* tgt = sm + L.o.tgt;
  double z = INFINITY;
  d3 p = {0, 0, 0};
  d4 uq = {1, 0, 0, 0};
  const bool on = lane < L.nhum;
  if (on) {
    const double* in = raw + 7 * lane;
    const double* rp = raw + 7 * human_root;
    const double* sc = sm + L.o.scale;
    double sr = sc[human_root];
    d3 srp = {sr * rp[0], sr * rp[1], sr * rp[2]};
    if (lane == human_root) p = srp;
    else {
      double s = sc[lane];
      p = d3{(in[0] - rp[0]) * s + srp.x, (in[1] - rp[1]) * s + srp.y, (in[2] - rp[2]) * s + srp.z};
    }
    const double* qo = sm + L.o.quat_off + 4 * lane;
    const double* po = sm + L.o.pos_off + 3 * lane;
    d4 q = qnormalize(d4{in[3], in[4], in[5], in[6]});
    uq = qnormalize(qmul(q, qnormalize(d4{qo[0], qo[1], qo[2], qo[3]})));
    p = p + qrot(uq, d3{po[0], po[1], po[2]});
    if (is_foot[lane] && p.x == p.x) z = p.z;
  }
  if (flags & GMR_FLAG_OFFSET_TO_GROUND) {
    double lowest = row0_min(z);      // human bodies live in lanes 0..nhum-1, nhum <= 16
    p.z = p.z - lowest + ground_offset;
  }
  if (on) {
    double* o = tgt + 7 * lane;
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = uq.w; o[4] = uq.x; o[5] = uq.y; o[6] = uq.z;
  }
  WSYNC();
  PROF_END(pr, PH_PRE);
}

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
// QP: which box-QP solver the instance carries (one per instance: the dense solver's 2 x NVP row registers would
// otherwise set the register allocation of robots that never run it)
template <int NVP, int NW, int QP>
__global__ __launch_bounds__(64 * NW, GMR_IK_MIN_WAVES) void ik_streams_kernel(const uint4* __restrict__ image, IkLay<NVP, NW> L, IkParams P,
                                                             int S, int T, const double* __restrict__ q0,
                                                             const double* __restrict__ human,
                                                             const int32_t* __restrict__ len, int flags,
                                                             double* __restrict__ q_out,
                                                             int32_t* __restrict__ nsolve,
                                                             int32_t* __restrict__ status,
                                                             double* __restrict__ tgt_out,
                                                             double* __restrict__ err_out,
                                                             unsigned long long* __restrict__ prof_out) {
  extern __shared__ __align__(16) double smem[];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x;
  if (s >= S) return;
  double* sm = smem;
  Prof pr;
#ifdef GMR_IK_PROFILE
  for (int i = 0; i < PH_COUNT; i++) pr.acc[i] = 0;
  const unsigned long long k_t0 = __builtin_amdgcn_s_memtime(), k_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  short* si = reinterpret_cast<short*>(smem + L.o.n_double);
  uint32_t* sw = reinterpret_cast<uint32_t*>(si + L.o.n_short);
  int* ctl = reinterpret_cast<int*>(sw + L.o.w_ctl);
  short* hop = si + L.o.i_hop;
  short* depth = si + L.o.i_depth;
  short* body_hinge = si + L.o.i_body_hinge;
  short* hinge_body = si + L.o.i_hinge_body;
  short* limited = si + L.o.i_limited;
  short* is_foot = si + L.o.i_is_foot;

  // ---- stage the constants: one coalesced copy of the host-built LDS image ---------------------
  const int nv = L.nv, nq = L.nq, nhum = L.nhum;
  {
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const int n16 = (L.smem_bytes + 15) >> 4;
    for (int i = threadIdx.x; i < n16; i += 64 * NW) dst[i] = image[i];
  }
  __syncthreads();
  if (NW > 1 && wave > 0) {
    helper_loop<NW, QP>(L, sm, sw, si, hinge_body, ctl, wave, lane, pr);
#ifdef GMR_IK_PROFILE
    if (wave == 1 && lane == 0 && prof_out)      // second row of stamps: helper wavefront 1
      for (int i = 0; i < PH_COUNT; i++) prof_out[((size_t)S + s) * PH_COUNT + i] = pr.acc[i];
#endif
    return;
  }
  const double* prm = sm + L.o.params;   // damping, lm_damping, tol, limit_gain, ground_offset, dt: read where used
  const int max_iter = P.max_iter, human_root = P.human_root;
  const int use0 = P.use0, use1 = P.use1;

  for (int i = lane; i < nq; i += 64) (sm + L.o.q)[i] = q0[(size_t)s * nq + i];
  WSYNC();
  hinge_sincos(L, sm, lane);
  WSYNC();
  const FkLane fkc = fk_lane(L, sm, hop, depth, body_hinge, lane);
  int pk_main[2];                         // task of this lane's (task, dof) pair in either table (column phase)
#pragma unroll
  for (int st = 0; st < 2; st++) pk_main[st] = (unsigned short)(si + L.o.i_pair_task[st])[lane] & 15u;
  fk_wave<NW>(L, sm, fkc, lane, pr);

  const int Ts = len ? min(len[s], T) : T;
  const size_t fstride = (size_t)nhum * 7;
  const double* hs = human + (size_t)s * T * fstride;
  int stat = GMR_STATUS_OK;
  int qp_state = 0;     // this lane's bound state of the previous solve (QP warm start)
  TreeState tree_state = {0ull, 0ull};
  // (only the instance's own solver's table is live: the others are dead code per template instance)
  const TreeRows<16> rows_s = (NW > 1 && QP == QP_TREE_SMALL) ? tree_rows<7, 9, false>(L, si, 0, lane) : TreeRows<16>{};
  const TreeRows<18> rows_l = (NW > 1 && QP != QP_TREE_SMALL) ? tree_rows<8, 10, false>(L, si, 0, lane) : TreeRows<18>{};
  const TreeRows<16> rows_r = (NW == 1 && QP == QP_TREE_SMALL) ? tree_rows<7, 9, true>(L, si, 0, lane) : TreeRows<16>{};
  int h_stage = -1;     // stage whose sparsity pattern H currently holds
  // Commands to the helper wavefronts go through a two-slot mailbox, alternating per command: a helper reads the slot
  // of command n right after the barrier that publishes it, the main wavefront writes command n + 1 into the OTHER
  // slot, so no write ever races a read (both sides count commands: `epoch`).
  int epoch = 0;
  // first frame's raw targets
  double r0 = 0.0, r1 = 0.0;
  if (Ts > 0) {
    if (lane < (int)fstride) r0 = hs[lane];
    if (lane + 64 < (int)fstride) r1 = hs[lane + 64];
  }
  for (int t = 0; t < Ts; t++) {
    if (lane < (int)fstride) (sm + L.o.raw)[lane] = r0;
    if (lane + 64 < (int)fstride) (sm + L.o.raw)[lane + 64] = r1;
    // prefetch the next frame (nhuman*7 <= 128 doubles)
    if (t + 1 < Ts) {
      const double* nx = hs + (size_t)(t + 1) * fstride;
      if (lane < (int)fstride) r0 = nx[lane];
      if (lane + 64 < (int)fstride) r1 = nx[lane + 64];
    }
    WSYNC();
    int ns0 = 0, ns1 = 0;
    const size_t f = (size_t)s * T + t;
    if (stat == GMR_STATUS_OK) {
      preprocess_wave<NW>(L, sm, is_foot, human_root, prm[4], flags, lane, pr);
      if (tgt_out)     // the poses handed to task.set_target (motion_retarget.py:117-136) = scaled_human_data
        for (int i = lane; i < (int)fstride; i += 64) tgt_out[f * fstride + i] = (sm + L.o.tgt)[i];
      double last_E = -1.0;                          // the stage's last residual norm (at the current configuration)
      for (int stage = 0; stage < 2; stage++) {
        if (!(stage == 0 ? use0 : use1) || (flags & GMR_FLAG_EVAL_ONLY)) continue;
        const StageTabs tb = {si + L.o.i_task_body[stage], si + L.o.i_task_human[stage], si + L.o.i_pair_task[stage],
                              si + L.o.i_pair_dof[stage], si + L.o.i_pair_index[stage],
                              reinterpret_cast<const uint2*>(NW == 1 ? reinterpret_cast<const uint32_t*>(image) + L.g_items[stage]
                                                                     : sw + L.w_items[stage])};
        const int K = L.K[stage];
        if (h_stage != stage) {
          // structural zeros of H are never written by the schedule: clear when the pattern changes
          for (int i = lane; i < nv * L.o.ldh; i += 64) (sm + L.o.H)[i] = 0.0;
          h_stage = stage;
          WSYNC();
        }
        // every residual evaluation is preceded by a kick of the helpers: they turn the FK state into the body
        // Jacobians of this stage's (task, dof) pairs while this wavefront evaluates residuals and Jl^-1
        if (NW > 1 && !(stage == 1 && (use1 & 4) && last_E >= 0.0)) {   // (same pairs: the first stage's last kick still holds)
          if (lane == 0) { int* c = ctl + 2 * (epoch & 1); c[0] = CMD_JBODY; c[1] = stage; }
          epoch++;
          __syncthreads();
        }
        // (same task list in both tables: the first stage's last evaluation is this stage's first, and its residuals
        //  and log-map terms are still in LDS for the Jl^-1 phase)
        double curr = (stage == 1 && (use1 & 2) && last_E >= 0.0) ? last_E
                                                                 : errors_wave<NW>(L, sm, tb.task_body, tb.task_human, K, lane, pr);
        int nsol = 0, num_iter = 0;
        for (;;) {
          build_qp_main<NW>(L, sm, stage, tb, hinge_body, limited, ctl, epoch, prm[0], prm[1], prm[3], lane, pk_main[stage], pr);
          PROF_COUNT(pr, PH_NSOLVE);
          int rc;
          if (NW > 1) {   // the 4-wavefront shape is only launched for robots that decompose (gmr_abi.hip)
            PROF_COUNT(pr, PH_NFACT);
            // helpers joined after barrier B3
            rc = QP == QP_TREE_SMALL ? solve_qp_tree<7, 9, false, true>(L, sm, sw, si, 0, lane, tree_state, pr, rows_s)
                                     : solve_qp_tree<8, 10, false, false>(L, sm, sw, si, 0, lane, tree_state, pr, rows_l);
          } else if (QP == QP_TREE_SMALL) {   // one wavefront, the four limbs in its four 16-lane rows
            PROF_COUNT(pr, PH_NFACT);
            rc = solve_qp_tree<7, 9, true, true>(L, sm, sw, si, 0, lane, tree_state, pr, rows_r);
          } else {
            rc = solve_qp_regs<NVP, NW>(L, sm, lane, qp_state, pr);
          }
          if (rc != GMR_STATUS_OK) { stat = rc; break; }
          integrate_wave<NW>(L, sm, prm[5], lane, pr);
          fk_wave<NW>(L, sm, fkc, lane, pr);
          if (NW > 1) {
            if (lane == 0) { int* c = ctl + 2 * (epoch & 1); c[0] = CMD_JBODY; c[1] = stage; }
            epoch++;
            __syncthreads();
          }
          double next = errors_wave<NW>(L, sm, tb.task_body, tb.task_human, K, lane, pr);
          last_E = next;
          nsol++;
          if (nsol > 1) num_iter++;
          if (!(curr - next > prm[2] && num_iter < max_iter)) break;
          curr = next;
        }
        if (stage == 0) ns0 = nsol; else ns1 = nsol;
        if (stat != GMR_STATUS_OK) break;
      }
    }
    if (err_out && stat == GMR_STATUS_OK) {
      // error1() / error2() of the reference (motion_retarget.py:188-200): both tables' residual norms at the
      // configuration this frame ends with (only evaluated when the caller asks: the per-frame API)
      for (int stage = 0; stage < 2; stage++) {
        double E = 0.0;
        if (stage == 0 ? use0 : use1)
          E = errors_wave<NW>(L, sm, si + L.o.i_task_body[stage], si + L.o.i_task_human[stage], L.K[stage], lane, pr);
        if (lane == 0) err_out[2 * f + stage] = E;
      }
    }
    for (int i = lane; i < nq; i += 64) q_out[f * nq + i] = (sm + L.o.q)[i];
    if (lane == 0) { nsolve[2 * f] = ns0; nsolve[2 * f + 1] = ns1; }
    WSYNC();
  }
  if (lane == 0) status[s] = stat;
  if (NW > 1) {                           // release the helper waves
    if (lane == 0) ctl[2 * (epoch & 1)] = CMD_EXIT;
    __syncthreads();
  }
#ifdef GMR_IK_PROFILE
  pr.acc[PH_TICKS] = __builtin_amdgcn_s_memtime() - k_t0;
  pr.acc[PH_REALTIME] = __builtin_amdgcn_s_memrealtime() - k_r0;
  if (lane == 0 && prof_out)
    for (int i = 0; i < PH_COUNT; i++) prof_out[(size_t)s * PH_COUNT + i] = pr.acc[i];
#else
  (void)prof_out;
#endif
}

}  // namespace gmr

// host-side launchers used by gmr_abi.hip.  NW = 1: one wave per stream (throughput shape, many
// streams); NW = 4: one main wave + 3 helpers per stream (latency shape: fewer streams than the chip
// has SIMDs, the helpers share the two wide assembly phases).  The solver of an instance follows from the
// robot's decomposition: NW = 1 runs the tree solver in DPP rows (<7, 9> fits a row) or the dense one,
// NW = 4 (only launched for robots that decompose) the <7, 9> or the <8, 10> tree instance.
template <int NVP, int NW, int QP>
static hipError_t launch_nvp(const uint4* d_image, const gmr::IkLayout* L, const gmr::IkParams* P, int S, int T,
                             const double* d_q0, const double* d_human,
                             const int32_t* d_len, int flags, double* d_q_out, int32_t* d_nsolve, int32_t* d_status,
                             double* d_tgt_out, double* d_err_out, hipStream_t stream, unsigned long long* d_prof) {
  hipLaunchKernelGGL((gmr::ik_streams_kernel<NVP, NW, QP>), dim3(S), dim3(64 * NW), L->smem_bytes, stream, d_image,
                     gmr::IkLay<NVP, NW>(static_cast<const gmr::IkDims&>(*L)), *P, S, T, d_q0, d_human, d_len, flags, d_q_out, d_nsolve, d_status,
                     d_tgt_out, d_err_out, d_prof);
  return hipGetLastError();
}

#define GMR_ARGS d_image, L, P, S, T, d_q0, d_human, d_len, flags, d_q_out, d_nsolve, d_status, d_tgt_out, d_err_out, stream, d_prof
#define GMR_DISPATCH(NVP_)                                                                             \
  case NVP_:                                                                                           \
    if (L->nw == 1) return L->tree_small ? launch_nvp<NVP_, 1, gmr::QP_TREE_SMALL>(GMR_ARGS)           \
                                         : launch_nvp<NVP_, 1, gmr::QP_DENSE>(GMR_ARGS);               \
    return L->tree_small ? launch_nvp<NVP_, 4, gmr::QP_TREE_SMALL>(GMR_ARGS)                           \
                         : launch_nvp<NVP_, 4, gmr::QP_TREE>(GMR_ARGS);

extern "C" hipError_t gmr_launch_ik_streams(const uint4* d_image, const gmr::IkLayout* L, const gmr::IkParams* P,
                                            int S, int T, const double* d_q0, const double* d_human,
                                            const int32_t* d_len, int flags, double* d_q_out, int32_t* d_nsolve,
                                            int32_t* d_status, double* d_tgt_out, double* d_err_out,
                                            hipStream_t stream, unsigned long long* d_prof) {
  if (S <= 0 || T <= 0) return hipSuccess;
  switch (L->nvp) {
    GMR_DISPATCH(28)
    GMR_DISPATCH(32)
    GMR_DISPATCH(36)
    GMR_DISPATCH(48)
    default: return hipErrorInvalidValue;
  }
}

template <int NVP>
static const void* kernel_of(int nw, int tree_small) {
  if (nw == 1) return tree_small ? reinterpret_cast<const void*>(gmr::ik_streams_kernel<NVP, 1, gmr::QP_TREE_SMALL>)
                                 : reinterpret_cast<const void*>(gmr::ik_streams_kernel<NVP, 1, gmr::QP_DENSE>);
  return tree_small ? reinterpret_cast<const void*>(gmr::ik_streams_kernel<NVP, 4, gmr::QP_TREE_SMALL>)
                    : reinterpret_cast<const void*>(gmr::ik_streams_kernel<NVP, 4, gmr::QP_TREE>);
}

// opt the instance that (nvp, nw, tree_small) selects into `bytes` of dynamic LDS on the CURRENT device
extern "C" hipError_t gmr_ik_set_max_smem(int nvp, int nw, int tree_small, int bytes) {
  const void* f = nullptr;
  switch (nvp) {
    case 28: f = kernel_of<28>(nw, tree_small); break;
    case 32: f = kernel_of<32>(nw, tree_small); break;
    case 36: f = kernel_of<36>(nw, tree_small); break;
    case 48: f = kernel_of<48>(nw, tree_small); break;
    default: return hipErrorInvalidValue;
  }
  return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
