// gmr_ik_wide.hip -- the THROUGHPUT shape of the retargeting kernel (rows H2-H7 of SURVEY.md section 8a): one
// 64-lane wavefront per motion stream, 22 KB of LDS per stream and <= 256 registers, so that TWO wavefronts are
// resident per SIMD (seven streams per CU) where the kernel of gmr_ik.hip holds one (four streams per CU).
//
// Same algorithm, same lane mappings and the same arithmetic as the one-wavefront instance of gmr_ik.hip (see the
// header there); what differs is where things live (gmr_ik_wide_layout.h):
//   global image   everything that is the same for every stream -- joint-local transforms, limits, task and pair
//                  tables, tree tables, parameters, the H-assembly schedule -- read per phase with a few coalesced
//                  per-lane loads (the image is a few KB: vector-L1 / L2 resident, shared by the CU's streams)
//   LDS state      q, FK result, world hinge axes, targets; H in the block-arrowhead form the solver consumes
//                  (four 16 x 7 matrices [D_l; B_l] + the 9 x 9 trunk block), c, bounds, x
//   LDS scratch    residuals, -Jl^-1 blocks, weighted Jacobian columns; the solver's transposes and Schur parts
//                  alias them
// The box-QP is the tree-structured solver of gmr_ik_tree.h in its one-wavefront form (the four limbs in the four
// 16-lane DPP rows), reading H rows as contiguous 56-byte pieces and keeping the bound sets in LANE coordinates
// (bit = the lane that owns the variable), so that no dof tables are consulted inside a pivoting round.
//
// Used for every robot that decomposes into <= 4 limbs of <= 7 dofs and a trunk of <= 9 (all shipped robots) when a
// launch has more streams than the latency shape pays off for (gmr_abi.hip).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/gmr_hip.h"
#include "gmr_device_math.h"
#include "gmr_ik_prof.h"
#include "gmr_ik_wide_layout.h"

#ifndef GMR_WIDE_NO_SCHUR_MFMA
#define GMR_WIDE_SCHUR_MFMA 1
#endif
#ifndef GMR_WIDE_MIN_WAVES
#define GMR_WIDE_MIN_WAVES 3
#endif

namespace gmr {
namespace wide {

constexpr WideLds LD = wide_lds();
constexpr WideImg IM = wide_img();

// one wavefront per workgroup: LDS operations of a wave execute in order, the barrier degenerates to the wait
__device__ __forceinline__ void wsync() { __syncthreads(); }

template <class T>
__device__ __forceinline__ const T* img_at(const char* img, int byte_off) {
  return reinterpret_cast<const T*>(img + byte_off);
}

// value of lane `addr / 4` (per-lane index), all lanes active
__device__ __forceinline__ double bpermute_d(int addr, double v) {
  int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
  int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// FK: mj_kinematics semantics (App. A.3) by pointer jumping, lane = body (see fk_wave in gmr_ik.hip)
// ---------------------------------------------------------------------------------------------
template <class DimsRef>
__device__ __forceinline__ void fk_wide(DimsRef D, double* sm, const char* __restrict__ img, int lane, Prof& pr,
                                        bool root_is_unit = false) {
  PROF_BEGIN(pr);
  lane = fresh_lane(lane);     // (lane predicates of this phase are computed here and die with it: gmr_device_math.h)
  const int nb = D.nb;
  double* q = sm + LD.q;
  d3 pos = {0, 0, 0}, ax = {0, 0, 0};
  d4 quat = {1, 0, 0, 0};
  uint32_t hops = 0;
  int dep = 0, hinge = -1;
  if (lane < nb) {
    const uint2 ci = img_at<uint2>(img, IM.fki)[lane];
    hops = ci.x; dep = ci.y & 255u; hinge = (int)(ci.y >> 8) - 1;
    if (lane == 0) {
      quat = d4{q[3], q[4], q[5], q[6]};
      if (!root_is_unit) {          // (a q_out row read back by the next chunk went through here already: normalising
        quat = qnormalize(quat);    //  twice is not idempotent in the last bit)
        q[3] = quat.w; q[4] = quat.x; q[5] = quat.y; q[6] = quat.z;
      }
      pos = d3{q[0], q[1], q[2]};
    } else {
      const double* c = img_at<double>(img, IM.fkc) + 10 * lane;
      pos = d3{c[0], c[1], c[2]};
      quat = d4{c[3], c[4], c[5], c[6]};
      if (hinge >= 0) {
        ax = d3{c[7], c[8], c[9]};
        const double* sc = sm + LD.hsc + 2 * hinge;     // sin, cos of q[7 + h] / 2 (hinge_sincos / integrate)
        const double s = sc[0], cs = sc[1];
        quat = qmul(quat, d4{cs, ax.x * s, ax.y * s, ax.z * s});
      }
    }
  }
  // (the ancestor's transform comes out of its lane's registers as they were before the round: no staging through LDS)
  for (int r = 0; r < D.nhop; r++) {
    const int src = (int)(((hops >> (6 * r)) & 63u) << 2);                    // byte address of the source lane
    const d3 pa = {bpermute_d(src, pos.x), bpermute_d(src, pos.y), bpermute_d(src, pos.z)};
    const d4 qa = {bpermute_d(src, quat.w), bpermute_d(src, quat.x), bpermute_d(src, quat.y), bpermute_d(src, quat.z)};
    if (lane < nb && dep >= (1 << r)) {
      pos = pa + qrot(qa, pos);
      quat = qmul(qa, quat);
    }
  }
  if (lane < nb) {
    quat = qnormalize(quat);
    double* o = sm + LD.xa + 7 * lane;
    o[0] = pos.x; o[1] = pos.y; o[2] = pos.z; o[3] = quat.w; o[4] = quat.x; o[5] = quat.y; o[6] = quat.z;
    if (hinge >= 0) {
      d3 aw = qrot(quat, ax);
      double* xa = sm + LD.xaxis + 3 * lane;
      xa[0] = aw.x; xa[1] = aw.y; xa[2] = aw.z;
    }
  }
  wsync();
  PROF_END(pr, PH_FK);
}

// residuals of the stage's tasks and their unweighted norm (motion_retarget.py:188-200); lane = task
__device__ __forceinline__ double errors_wide(double* sm, uint32_t taskw, int K, int lane, Prof& pr) {
  PROF_BEGIN(pr);
  lane = fresh_lane(lane);     // (lane predicates of this phase are computed here and die with it: gmr_device_math.h)
  double ss = 0.0;
  if (lane < K) {
    const int b = taskw & 255u, h = taskw >> 8;
    const double* x = sm + LD.xa + 7 * b;
    const double* tg = sm + LD.tgt + 7 * h;
    double e[6], aux[5];
    se3_log_rel5(d3{x[0], x[1], x[2]}, d4{x[3], x[4], x[5], x[6]}, d3{tg[0], tg[1], tg[2]},
                 d4{tg[3], tg[4], tg[5], tg[6]}, e, aux);
    double* eo = sm + LD.e + 6 * lane;
#pragma unroll
    for (int r = 0; r < 6; r++) { eo[r] = e[r]; ss += e[r] * e[r]; }
    double* ao = sm + LD.eaux + 5 * lane;       // a, sin|w|, cos|w|, |w|, 1/|w|: reused by the Jl^-1 phase
#pragma unroll
    for (int r = 0; r < 5; r++) ao[r] = aux[r];
  }
  ss = row0_sum(ss);
  wsync();
  PROF_END(pr, PH_ERR);
  return sqrt(ss);
}

// (a) M_k = -Jl^-1(e_k), by COLUMN on three lanes per task (lane = task + 16 j, j < 3: se3_jlinv_col5, the latency kernel's
// form); returns the LM term mu.  (Round 2 measured this form here as "the same frames/s, 13 more registers, 8 of them in
// scratch"; since the lane predicates are no longer parked -- 229 registers -- it fits, and the phase issues a third of the
// instructions: 14 of 64 lanes were doing all of it.  GMR_WIDE_JLOG_ROWS restores the one-lane-per-task form.)
__device__ __forceinline__ double jlog_wide(double* sm, const char* __restrict__ img, int stage, int K, double lm_damping,
                                            int lane, Prof& pr) {
  PROF_BEGIN(pr);
  lane = fresh_lane(lane);     // (lane predicates of this phase are computed here and die with it: gmr_device_math.h)
  double mu = 0.0;
#ifdef GMR_WIDE_JLOG_ROWS
  if (lane < K) {
    const double* w = img_at<double>(img, IM.task[stage]) + 2 * lane;
    const double wp = w[0], wr = w[1];
    const double* e = sm + LD.e + 6 * lane;
    double ee[6];
#pragma unroll
    for (int r = 0; r < 6; r++) ee[r] = e[r];
    const double* ax = sm + LD.eaux + 5 * lane;
    const double aux[5] = {ax[0], ax[1], ax[2], ax[3], ax[4]};
    m3 A, B;
    se3_jlinv_aux5(ee, aux, A, B);
    double* M = sm + LD.M + 18 * lane;
#pragma unroll
    for (int i = 0; i < 9; i++) { M[i] = -A.a[i]; M[9 + i] = -B.a[i]; }
    double* wt = sm + LD.wts + 2 * lane;
    wt[0] = wp; wt[1] = wr;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      double v = (r < 3 ? wp : wr) * ee[r];
      mu += v * v;
    }
  }
#else
  const int k = lane & 15, j = lane >> 4;
  if (k < K && j < 3) {
    const double* e = sm + LD.e + 6 * k;
    double ee[6];
#pragma unroll
    for (int r = 0; r < 6; r++) ee[r] = e[r];
    const double* ax = sm + LD.eaux + 5 * k;
    const double aux[5] = {ax[0], ax[1], ax[2], ax[3], ax[4]};
    double Ac[3], Bc[3];
    se3_jlinv_col5(ee, aux, j, Ac, Bc);
    double* M = sm + LD.M + 18 * k + j;
#pragma unroll
    for (int i = 0; i < 3; i++) { M[3 * i] = -Ac[i]; M[9 + 3 * i] = -Bc[i]; }
    if (j == 0) {
      const double* w = img_at<double>(img, IM.task[stage]) + 2 * k;
      const double wp = w[0], wr = w[1];
      double* wt = sm + LD.wts + 2 * k;
      wt[0] = wp; wt[1] = wr;
#pragma unroll
      for (int r = 0; r < 6; r++) {
        double v = (r < 3 ? wp : wr) * ee[r];
        mu += v * v;
      }
    }
  }
#endif
  mu = lm_damping * row0_sum(mu);
  wsync();                                             // (eaux is dead from here: the column phase overwrites it)
  PROF_END(pr, PH_JLOG);
  return mu;
}

// (b) virtual lane = (task, dof) pair: weighted task-Jacobian column W_k (-Jl^-1(e_k)) J_body[:, d]
__device__ __forceinline__ void pairs_wide(double* sm, const char* __restrict__ img, int stage, int P, int lane) {
  lane = fresh_lane(lane);     // (lane predicates of this phase are computed here and die with it: gmr_device_math.h)
  double* Jw = sm + LD.Jw;
  double* cpart = sm + LD.cpart;
  const double* X = sm + LD.xa;
  const uint32_t* pw = img_at<uint32_t>(img, IM.pair[stage]);
  for (int p = lane; p < P; p += 64) {
    const uint32_t info = pw[p];                        // [3:0] task, [9:4] dof, [15:10] task body, [21:16] hinge body
    const int k = info & 15u, dof = (info >> 4) & 63u, b = (info >> 10) & 63u, c = (info >> 16) & 63u;
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
      const double* xa = sm + LD.xaxis + 3 * c;
      ang = d3{xa[0], xa[1], xa[2]};
      lin = cross(ang, pb - d3{X[7 * c], X[7 * c + 1], X[7 * c + 2]});
    }
    d3 jl = qrot_inv(qb, lin), ja = qrot_inv(qb, ang);   // body-frame Jacobian column
    const double* M = sm + LD.M + 18 * k;
    const double* e = sm + LD.e + 6 * k;
    const double wp = (sm + LD.wts)[2 * k], wr = (sm + LD.wts)[2 * k + 1];
    double* o = Jw + 6 * p;
    double cp = 0.0;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      double top = M[3 * r] * jl.x + M[3 * r + 1] * jl.y + M[3 * r + 2] * jl.z + M[9 + 3 * r] * ja.x +
                   M[9 + 3 * r + 1] * ja.y + M[9 + 3 * r + 2] * ja.z;
      double bot = M[3 * r] * ja.x + M[3 * r + 1] * ja.y + M[3 * r + 2] * ja.z;
      top *= wp; bot *= wr;
      o[r] = top;
      o[3 + r] = bot;
      cp += top * (wp * e[r]) + bot * (wr * e[3 + r]);
    }
    cpart[p] = cp;   // this column's contribution to c = sum_k (W J_k)^T (W e_k)
  }
  if (lane < 6) Jw[6 * WD_P + lane] = 0.0;   // the row the padding items of the H schedule read
  if (lane == 6) cpart[WD_P] = 0.0;          // and the share absent (task, dof) pairs gather
}

// (c) lane = dof: gather c; bounds of the limited hinges (mink ConfigurationLimit)
template <class DimsRef>
__device__ __forceinline__ void cvec_wide(DimsRef D, double* sm, const char* __restrict__ img, int stage,
                                          double limit_gain, int lane) {
  lane = fresh_lane(lane);     // (lane predicates of this phase are computed here and die with it: gmr_device_math.h)
  const double* cpart = sm + LD.cpart;
  if (lane < D.nv) {
    const uint32_t* ci = img_at<uint32_t>(img, IM.cidx[stage]) + lane;
    uint32_t w[WD_K / 2];
#pragma unroll
    for (int g = 0; g < WD_K / 2; g++) w[g] = ci[64 * g];
    const double2 lim = img_at<double2>(img, IM.lim)[lane];
    const uint32_t limited = img_at<uint32_t>(img, IM.limi)[lane];
    double cc = 0.0;
#pragma unroll
    for (int k = 0; k < WD_K; k++) {
      const uint32_t off = (k & 1) ? w[k / 2] >> 16 : w[k / 2] & 0xffffu;   // byte offset; absent pairs name the zero slot
      cc += *reinterpret_cast<const double*>(reinterpret_cast<const char*>(cpart) + off);
    }
    (sm + LD.c)[lane] = cc;
    double lo = -INFINITY, hi = INFINITY;
    if (limited) {
      const double th = (sm + LD.q)[7 + lane - 6];
      hi = limit_gain * (lim.y - th);
      lo = -limit_gain * (th - lim.x);
    }
    (sm + LD.lo)[lane] = lo;
    (sm + LD.hi)[lane] = hi;
  }
}

// (d) H: every lane sums the terms of the entries it owns (static schedule, streamed from the global image: the
// same words for every stream of the CU) and stores each entry once, directly where the solver reads it
struct __attribute__((aligned(16))) dd2 { double x, y; };   // Jw rows start on 16-byte boundaries: one b128 read per piece
__device__ __forceinline__ double dot6v(const double* a, const double* b) {
  const dd2* pa = reinterpret_cast<const dd2*>(a);
  const dd2* pb = reinterpret_cast<const dd2*>(b);
  dd2 a0 = pa[0], a1 = pa[1], a2 = pa[2], b0 = pb[0], b1 = pb[1], b2 = pb[2];
  return (a0.x * b0.x + a0.y * b0.y + a1.x * b1.x) + (a1.y * b1.y + a2.x * b2.x + a2.y * b2.y);
}

__device__ __forceinline__ void hacc_wide(double* sm, const char* __restrict__ img, int items_off, int ntrip, double diag,
                                          int lane) {
  lane = fresh_lane(lane);     // (lane predicates of this phase are computed here and die with it: gmr_device_math.h)
  double* __restrict__ H = sm + LD.H;
  char* __restrict__ Hb = reinterpret_cast<char*>(H);
  const char* __restrict__ Jb = reinterpret_cast<const char*>(sm + LD.Jw);
  const uint2* p = img_at<uint2>(img, items_off) + lane;
  uint2 n[4];
#pragma unroll
  for (int k = 0; k < 4; k++) n[k] = p[k * 64];
  const int dof = img_at<int>(img, IM.tree)[lane];
  // H takes the place of the residual / -Jl^-1 / c-share scratch (all consumed by now): structural zeros and the rows
  // of absent variables are never written by the schedule, so the block starts from zero
  for (int i = lane; i < WD_HN; i += 64) H[i] = 0.0;
  wsync();
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
    for (int k = 0; k < 4; k++)
      s[k] = dot6v(reinterpret_cast<const double*>(Jb + (w[k].x & 0xffffu)), reinterpret_cast<const double*>(Jb + (w[k].x >> 16)));
#pragma unroll
    for (int k = 0; k < 4; k++) {
      acc += s[k];
      if ((int)w[k].y < 0) {
        const uint32_t o1 = w[k].y & 0x1fffu, o2 = (w[k].y >> 13) & 0x1fffu;
        if (w[k].y & WD_ITEM_ADD) {     // half of a heavy entry: the cell was zeroed above, the two halves commute
          atomicAdd(reinterpret_cast<double*>(Hb + o1), acc);
          if (o2 != o1) atomicAdd(reinterpret_cast<double*>(Hb + o2), acc);
        } else {
          *reinterpret_cast<double*>(Hb + o1) = acc;
          *reinterpret_cast<double*>(Hb + o2) = acc;
        }
        acc = 0.0;
      }
    }
  }
  // the damping term: lane 16 g + a owns the diagonal of limb variable (g, a), lanes 7..15 the trunk's
  wsync();
  const int r = lane & 15;
  if (dof >= 0 && (r < WD_NL || lane < 16)) H[r < WD_NL ? 7 * lane + r : WD_HT + (WD_NT + 1) * (r - WD_NL)] += diag;
}

// ---------------------------------------------------------------------------------------------
// box-constrained strictly convex QP (App. A.6), tree-structured: see gmr_ik_tree.h for the algorithm.  This is its
// one-wavefront form on the compact H: row group g = lane >> 4 eliminates limb g (rows 0..6 of the group: D_g, rows
// 7..15: B_g and, redundantly in every group, the trunk).  Bound sets in lane coordinates: bit 16 g + a = limb
// variable (g, a), bit 7 + t = trunk variable t (its owner is the trunk lane of group 0).
// ---------------------------------------------------------------------------------------------
struct RowState { unsigned long long lower, upper; };

template <class DimsRef>
__device__ __forceinline__ int solve_rows(DimsRef D, double* sm, const char* __restrict__ img, const int lane_in0,
                                          RowState& bs, Prof& pr) {
  constexpr int NL = WD_NL, NT = WD_NT, NV = NL + NT, TLD = WD_LD;
  static_assert(NV == 16, "a limb's local matrix fills one 16-lane row");
  int lane_in = fresh_lane(lane_in0);
  int grp = lane_in >> 4, lane = lane_in & 15;
  const double* H = sm + LD.H;
  double* xs = sm + LD.x;
  const double* los = sm + LD.lo;
  const double* his = sm + LD.hi;
  double* Spart = sm + LD.spart;                  // [4][NT (NT + 1) / 2]: lower triangles of the Schur contributions
  double* rpart = sm + LD.rpart;                  // [4][NT]
  double* Tsh = sm + LD.tsh;                      // [NT][WD_TT]: the trunk factor, identical in all groups (benign same-value writes)
  double* xl = sm + LD.xl;                        // x by owner lane (multipliers)
  double* Lscr = sm + LD.lscr + grp * 16 * TLD;   // this group's transpose scratch
  unsigned long long* vset = reinterpret_cast<unsigned long long*>(sm + LD.vset);   // {to_lower, to_upper, release, flags} x 2
  const int* tree = img_at<int>(img, IM.tree);

  bool is_limb = lane < NL, is_trunk = !is_limb;
  int a = lane, t = lane - NL;
  const int dof = tree[lane_in];
  bool row = dof >= 0;                                        // this lane holds a real row
  bool own = row && (is_limb || grp == 0);                    // ... and reports the variable's violations
  int me = is_limb ? lane_in : lane;                          // the variable's bit in the bound sets
  const double lo = row ? los[dof] : 0.0, hi = row ? his[dof] : 0.0;
  const double ci = row ? (sm + LD.c)[dof] : 0.0;
  const double* LMrow = H + 7 * lane_in;                      // D_g row a (limb lanes) / B_g row t (trunk lanes)
  const double* Trow = H + WD_HT + NT * (is_trunk ? t : 0);
  double dual_tol = -1.0;                                     // 1e-13 (1 + max |c|): computed when a multiplier is first checked
  const double ptol_lo = 1e-12 * (1.0 + fabs(lo)), ptol_hi = 1e-12 * (1.0 + fabs(hi));
  const unsigned long long absent = __ballot(!row);           // rows without a variable (short limbs / trunk)

  int pcount = 3, ninf_best = 65;
  for (int it = 0; it < 100; it++) {
    PROF_BEGIN(pr);
    // (a round starts from a fresh copy of the lane id: its lane predicates are recomputed here -- one v_cmp each --
    //  instead of living in spilled scalar-register pairs across the whole frame loop)
    lane_in = fresh_lane(lane_in0);
    grp = lane_in >> 4; lane = lane_in & 15;
    is_limb = lane < NL; is_trunk = !is_limb;
    a = lane; t = lane - NL;
    row = dof >= 0;
    own = row && (is_limb || grp == 0);
    me = is_limb ? lane_in : lane;
    const unsigned long long fixedm = bs.lower | bs.upper;
    const unsigned long long gone = fixedm | absent;          // columns that are the identity in this round
    const unsigned cf = (unsigned)(gone >> (16 * grp)) & 0x7Fu;   // limb columns of this group
    const unsigned tf = (unsigned)(gone >> NL) & 0x1FFu;          // trunk columns (wave-uniform)
    const bool self_fixed = !row || ((fixedm >> me) & 1ull);
    const double xfix = !row ? 0.0 : (((bs.lower >> me) & 1ull) ? lo : (((bs.upper >> me) & 1ull) ? hi : 0.0));
    // ---- (1) local rows and right-hand side -----------------------------------------------------
    double r[NV];
#pragma unroll
    for (int m = 0; m < NV; m++) r[m] = 0.0;
    {
      double h[NL];
#pragma unroll
      for (int m = 0; m < NL; m++) h[m] = LMrow[m];
#pragma unroll
      for (int m = 0; m < NL; m++) {
        const bool cfixed = (cf >> m) & 1u;
        const bool keep = !self_fixed && !cfixed && (is_trunk || m <= a);
        double v = keep ? h[m] : 0.0;
        if (is_limb && m == a && (self_fixed || cfixed)) v = 1.0;   // fixed / padding limb row: identity
        r[m] = v;
      }
    }
    // -c_i - sum over fixed j of H_ij x_j (the bound value of j from the uniform sets)
    double rhs0 = row ? (self_fixed ? xfix : -ci) : 0.0;
    {
      // (one iteration ahead with the loads: the round trips of the fixed variables overlap; same terms, same order)
      auto term = [&](int j, double& coef, double& bj) {        // j: owner lane of a fixed variable (wave-uniform)
        const int dj = tree[j];
        bj = ((bs.lower >> j) & 1ull) ? los[dj] : his[dj];
        const int jr = j & 15, lj = j >> 4;
        if (jr < NL) coef = is_limb ? (grp == lj ? LMrow[jr] : 0.0) : H[7 * (16 * lj + lane) + jr];
        else coef = is_limb ? H[7 * (16 * grp + jr) + a] : Trow[jr - NL];
      };
      unsigned long long mm = fixedm;
      if (mm) {
        double coef, bj;
        term(__ffsll((long long)mm) - 1, coef, bj);
        mm &= mm - 1;
        while (mm) {
          double cn, bn;
          term(__ffsll((long long)mm) - 1, cn, bn);
          mm &= mm - 1;
          if (row && !self_fixed) rhs0 -= coef * bj;
          coef = cn; bj = bn;
        }
        if (row && !self_fixed) rhs0 -= coef * bj;
      }
    }
    double b = is_limb ? rhs0 : 0.0;
    PROF_END(pr, PH_KBUILD);
    PROF_BEGIN(pr);
    // ---- (2) eliminate the limb pivots (right-looking, forward substitution merged) --------------
    double mydinv = 1.0;
    bool bad = false;
    double dp = row_bcast_d(r[0], 0);
    double dinv = fast_rsqrt(dp);
#pragma unroll
    for (int p = 0; p < NL; p++) {
      bad = bad || !(dp > 0.0);
      const double rs = r[p] * dinv;                         // (row p holds the pivot itself: d_p / sqrt(d_p))
      double l = lane > p ? rs : 0.0;                        // column p of L_g (rows > p) and of Y_g
      r[p] = lane == p ? rs : l;
      if (lane == p) mydinv = dinv;
      double dinv_next = 1.0;
      if (p + 1 < NL) {
        r[p + 1] = fma(-l, row_bcast_d(l, p + 1), r[p + 1]);
        dp = row_bcast_d(r[p + 1], p + 1);
        dinv_next = fast_rsqrt(dp);
      }
      const double yp = row_bcast_d(b, p) * dinv;            // row p keeps its unscaled b (l = 0 there): scaled after the loop
      b = fma(-l, yp, b);
#ifdef GMR_WIDE_SCHUR_MFMA
#pragma unroll
      for (int k = (p + 1 < NL ? p + 2 : p + 1); k < NL; k++) r[k] = fma(-l, row_bcast_d(l, k), r[k]);   // limb columns only
#else
#pragma unroll
      for (int k = (p + 1 < NL ? p + 2 : p + 1); k < NV; k++) r[k] = fma(-l, row_bcast_d(l, k), r[k]);
#endif
      dinv = dinv_next;
    }
    if (is_limb) b *= mydinv;                                // y_p = b_p / sqrt(d_p): the value every later row was given
    PROF_END(pr, PH_CHOL);
    PROF_BEGIN(pr);
    // ---- (3) publish the Schur contribution; park L_g / Y_g for the transposed reads --------------
#pragma unroll
    for (int m = 0; m < NL; m++) Lscr[lane * TLD + m] = r[m];
    if (is_trunk) rpart[grp * NT + t] = b;
#ifdef GMR_WIDE_SCHUR_MFMA
    // The Schur contributions -Y_g Y_g^T (four 9 x 7 by 7 x 9 products) on the matrix cores.  As row-broadcast + FMA column
    // updates they are 63 instructions AND nine more live registers per lane (the trunk columns of r[]): 229 registers, two
    // wavefronts per SIMD.  v_mfma_f64_16x16x4_f64 wants A[i][k] in lane i + 16 k and B[k][j] in lane j + 16 k: with
    // A = Y_g and B = Y_g^T both operands are the SAME register, filled from the transposes just parked in LDS (lane l reads
    // Y_g[l & 15][4 s + (l >> 4)] for K-step s); D[row = (l >> 4) + 4 reg][col = l & 15] goes straight to the exchange
    // buffer.  8 LDS reads + 8 MFMAs; the kernel fits three wavefronts per SIMD.
    {
      typedef double d4v __attribute__((ext_vector_type(4)));
      wsync();
      const int mi = lane_in & 15, mk = lane_in >> 4;
      d4v acc[4];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const double* Lg = sm + LD.lscr + g * 16 * TLD + mi * TLD;
        const double a0 = Lg[mk];                                   // Y_g[mi][mk]
        const double a1 = (4 + mk < NL) ? Lg[4 + mk] : 0.0;         // Y_g[mi][4 + mk] (column 7 does not exist)
        d4v c = {0.0, 0.0, 0.0, 0.0};
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, c, 0, 0, 0);
        acc[g] = c;
      }
      // lane (mi, mk) holds D[mk + 4 ri][mi]: rows 4 .. 15 can belong to the trunk block (ri = 0: rows 0 .. 3 never do); one
      // predicate per ri, the four groups' stores under it
#pragma unroll
      for (int ri = 1; ri < 4; ri++) {
        const int rowi = mk + 4 * ri;
        if (mi >= NL && rowi >= NL && mi <= rowi) {
          double* sp = Spart + ((rowi - NL) * (rowi - NL + 1)) / 2 + (mi - NL);
#pragma unroll
          for (int g = 0; g < 4; g++) sp[g * WD_TRI] = -acc[g][ri];
        }
      }
    }
#else
    if (is_trunk) {                                           // row t of the contribution: columns u <= t only are ever read
      double* sp = Spart + grp * WD_TRI + (t * (t + 1)) / 2;
#pragma unroll
      for (int u = 0; u < NT; u++) if (u <= t) sp[u] = r[NL + u];
    }
#endif
    unsigned long long* vcur = vset + 4 * (it & 1);
    if (lane == 0 && bad) atomicOr(&vcur[3], 1ull);
    wsync();                                                                                 // B1
    // the other slot was last read before this point: clear it for the next round
    if (lane_in < 4) vset[4 * ((it + 1) & 1) + lane_in] = 0ull;
    PROF_END(pr, PH_SUBST);
    PROF_BEGIN(pr);
    // ---- (4) every group: trunk Schur complement, factor, solve (redundant, no exchange) -----------
    double bt = 0.0;
    bool tbad = false;
    {
      double s[NT];
      {
        const int tt = is_trunk ? t : 0;
        double hv[NT], sp[NT];
#pragma unroll
        for (int u = 0; u < NT; u++) {
          hv[u] = Trow[u];
          const double* q0 = Spart + (tt * (tt + 1)) / 2 + (u <= tt ? u : 0);       // (u > t: unused, any valid address)
          sp[u] = (q0[0] + q0[WD_TRI]) + (q0[2 * WD_TRI] + q0[3 * WD_TRI]);
        }
        const double rp = (rpart[tt] + rpart[NT + tt]) + (rpart[2 * NT + tt] + rpart[3 * NT + tt]);
        const bool live = is_trunk && row && !self_fixed;
#pragma unroll
        for (int u = 0; u < NT; u++) {
          const bool cfixed = (tf >> u) & 1u;                                  // wave-uniform
          double v = (live && !cfixed && u <= t) ? hv[u] + sp[u] : 0.0;
          if (is_trunk && u == t && !(live && !cfixed)) v = 1.0;
          s[u] = v;
        }
        bt = is_trunk ? (live ? rhs0 + rp : rhs0) : 0.0;
      }
      double tdinv = 1.0;
      double dq = row_bcast_d(s[0], NL);
      double dinv2 = fast_rsqrt(dq);
#pragma unroll
      for (int q = 0; q < NT; q++) {
        tbad = tbad || !(dq > 0.0);
        const double ss = s[q] * dinv2;
        double l = t > q ? ss : 0.0;
        s[q] = t == q ? ss : l;
        if (t == q) tdinv = dinv2;
        double dinv_next = 1.0;
        if (q + 1 < NT) {
          s[q + 1] = fma(-l, row_bcast_d(l, NL + q + 1), s[q + 1]);
          dq = row_bcast_d(s[q + 1], NL + q + 1);
          dinv_next = fast_rsqrt(dq);
        }
        const double yq = row_bcast_d(bt, NL + q) * dinv2;
        bt = fma(-l, yq, bt);
#pragma unroll
        for (int k = q + 2; k < NT; k++) s[k] = fma(-l, row_bcast_d(l, NL + k), s[k]);
        dinv2 = dinv_next;
      }
      // back substitution: L^T through this group's scratch (columns 7..15 of rows 7..15)
      if (is_trunk) {
#pragma unroll
        for (int u = 0; u < NT; u++) Tsh[t * WD_TT + u] = s[u];
      }
      wsync();
      // row t of L^T without its diagonal (rows below are zero, other lanes get zeros): x_q = bt_q / L_qq twice scaled --
      // once for y (forward), once here -- and row q is final when its step comes, so no step needs a select
      double lt[NT];
#pragma unroll
      for (int q = 0; q < NT; q++) lt[q] = (is_trunk && t != q) ? Tsh[q * WD_TT + t] : 0.0;
      bt *= tdinv;                                           // y
#pragma unroll
      for (int q = NT - 1; q >= 0; q--) {
        const double xq = row_bcast_d(bt * tdinv, NL + q);
        bt = fma(-lt[q], xq, bt);
      }
      bt *= tdinv;                                           // x
    }
    PROF_END(pr, PH_RATIO);
    PROF_BEGIN(pr);
    // ---- (5) limbs: y_g - Y_g^T x_T, then back substitution with L_g^T ----------------------------
    double x = bt;                                                                     // trunk lanes
    {
      double lt[NL];
#pragma unroll
      for (int m = 0; m < NL; m++) lt[m] = (is_limb && m != a) ? Lscr[m * TLD + a] : 0.0;   // column a of L_g, off-diagonal
      double bb = b;                                                                   // y_g (limb lanes)
#pragma unroll
      for (int u = 0; u < NT; u++) {
        const double xt = row_bcast_d(bt, NL + u);
        if (is_limb) bb = fma(-Lscr[(NL + u) * TLD + a], xt, bb);                      // Y_g[u][a]
      }
#pragma unroll
      for (int p = NL - 1; p >= 0; p--) {
        const double xp = row_bcast_d(bb * mydinv, p);
        bb = fma(-lt[p], xp, bb);                            // (rows >= p have lt[p] = 0: row p is final at its step)
      }
      if (is_limb) x = bb * mydinv;
    }
    // ---- (6) violated bounds (free set) / multipliers (fixed set): g = H x + c ---------------------
    if (fixedm != 0ull) {                                     // multipliers need the whole x, by owner lane
      if (dual_tol < 0.0) dual_tol = 1e-13 * (1.0 + rows3_max(lane_in < D.nv ? fabs((sm + LD.c)[lane_in]) : 0.0));   // (wave-uniform)
      if (is_limb || grp == 0) xl[lane_in] = row ? x : 0.0;
      wsync();                                                                               // B2
    }
    PROF_END(pr, PH_MULT);
    PROF_BEGIN(pr);
    int newst = 0;                                            // 0 none, 1 -> lower, 2 -> upper, 3 release
    if (own) {
      if (!self_fixed) {
        if (x < lo - ptol_lo) newst = 1;
        else if (x > hi + ptol_hi) newst = 2;
      } else {
        double g0 = ci, g1 = 0.0;
        if (is_limb) {                                        // row of a limb variable: D_g[a][:] and B_g[:][a]
#pragma unroll
          for (int m = 0; m < NL; m++) g0 = fma(LMrow[m], xl[16 * grp + m], g0);
#pragma unroll
          for (int u = 0; u < NT; u++) g1 = fma(H[7 * (16 * grp + NL + u) + a], xl[NL + u], g1);
        } else {                                              // row of a trunk variable: B_l[t][:] of every limb, T[t][:]
#pragma unroll
          for (int l = 0; l < 4; l++)
#pragma unroll
            for (int m = 0; m < NL; m++) g0 = fma(H[7 * (16 * l + lane) + m], xl[16 * l + m], g0);
#pragma unroll
          for (int u = 0; u < NT; u++) g1 = fma(Trow[u], xl[NL + u], g1);
        }
        const double g = g0 + g1;
        const bool at_lower = (bs.lower >> me) & 1ull;
        if (at_lower ? g < -dual_tol : g > dual_tol) newst = 3;
      }
    }
    // each violating owner lane sets its variable's bit in the round's set (LDS atomic OR: order-independent)
    if (newst != 0) atomicOr(&vcur[newst - 1], 1ull << me);
    if (lane == 0 && tbad) atomicOr(&vcur[3], 1ull);
    wsync();                                                                                 // B3
    PROF_END(pr, PH_IO);
    const unsigned long long to_lo = vcur[0], to_up = vcur[1], rel = vcur[2];
    if (vcur[3]) return GMR_STATUS_QP_FAILED;
    const unsigned long long all = to_lo | to_up | rel;
    if (all == 0ull) {
      if (own) xs[dof] = fmin(fmax(x, lo), hi);
      wsync();
      return GMR_STATUS_OK;
    }
    const int total = __popcll(all);
    unsigned long long sel = all;                             // block principal pivoting: exchange all
    if (total < ninf_best) { ninf_best = total; pcount = 3; }
    else if (pcount > 0) pcount--;
    else sel = 1ull << (63 - __clzll((long long)all));        // Murty: only the highest violated variable
    bs.lower = (bs.lower & ~(rel & sel)) | (to_lo & sel);
    bs.upper = (bs.upper & ~(rel & sel)) | (to_up & sel);
  }
  return GMR_STATUS_QP_MAXITER;
}

// (sin, cos) of every hinge's half angle for a configuration that did not come out of integrate_wide (q0)
template <class DimsRef>
__device__ __forceinline__ void hinge_sincos(DimsRef D, double* sm, int lane) {
  lane = fresh_lane(lane);     // (lane predicates of this phase are computed here and die with it: gmr_device_math.h)
  if (lane >= 6 && lane < D.nv) {
    double s, c;
    sincos_small(0.5 * (sm + LD.q)[7 + lane - 6], &s, &c);
    double* sc = sm + LD.hsc + 2 * (lane - 6);
    sc[0] = s; sc[1] = c;
  }
}

// mj_integratePos with v = dq/dt (App. A.7): lane 0 the free joint, lane 6+h hinge h (see integrate_wave)
template <class DimsRef>
__device__ __forceinline__ void integrate_wide(DimsRef D, double* sm, double dt, int lane, Prof& pr) {
  PROF_BEGIN(pr);
  lane = fresh_lane(lane);     // (lane predicates of this phase are computed here and die with it: gmr_device_math.h)
  double* q = sm + LD.q;
  const double* dq = sm + LD.x;
  const bool base = lane == 0, hinge = lane >= 6 && lane < D.nv;
  double half = 0.0, inv = 0.0;
  bool rotate = false;
  if (base) {
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
    double* sc = sm + LD.hsc + 2 * (lane - 6);
    sc[0] = s; sc[1] = c;
  }
  wsync();
  PROF_END(pr, PH_INTEG);
}

// target preprocessing (motion_retarget.py:203-270); lane = human body
template <class DimsRef>
__device__ __forceinline__ void preprocess_wide(DimsRef D, double* sm, const char* __restrict__ img, int human_root,
                                                double ground_offset, int flags, int lane, Prof& pr) {
  PROF_BEGIN(pr);
  lane = fresh_lane(lane);     // (lane predicates of this phase are computed here and die with it: gmr_device_math.h)
  const double* raw = sm + LD.raw;
  double* tgt = sm + LD.tgt;
  double z = INFINITY;
  d3 p = {0, 0, 0};
  d4 uq = {1, 0, 0, 0};
  const bool on = lane < D.nhum;
  if (on) {
    const double* cst = img_at<double>(img, IM.pre) + 8 * lane;
    const double* in = raw + 7 * lane;
    const double* rp = raw + 7 * human_root;
    const double sr = (img_at<double>(img, IM.pre) + 8 * human_root)[0];
    d3 srp = {sr * rp[0], sr * rp[1], sr * rp[2]};
    if (lane == human_root) p = srp;
    else {
      const double s = cst[0];
      p = d3{(in[0] - rp[0]) * s + srp.x, (in[1] - rp[1]) * s + srp.y, (in[2] - rp[2]) * s + srp.z};
    }
    d4 q = qnormalize(d4{in[3], in[4], in[5], in[6]});
    uq = qnormalize(qmul(q, qnormalize(d4{cst[4], cst[5], cst[6], cst[7]})));
    p = p + qrot(uq, d3{cst[1], cst[2], cst[3]});
    if (img_at<uint32_t>(img, IM.prei)[lane] && p.x == p.x) z = p.z;
  }
  if (flags & GMR_FLAG_OFFSET_TO_GROUND) {
    double lowest = row0_min(z);
    p.z = p.z - lowest + ground_offset;
  }
  if (on) {
    double* o = tgt + 7 * lane;
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = uq.w; o[4] = uq.x; o[5] = uq.y; o[6] = uq.z;
  }
  wsync();
  PROF_END(pr, PH_PRE);
}

// ---------------------------------------------------------------------------------------------
// the kernel
//
// Two dispatch modes of the same body:
//   direct  (Q.ring == nullptr)  workgroup s owns stream s for all of its frames;
//   queued                       a resident set of wavefronts serves (stream, chunk of Q.chunk frames) items from a FIFO
//                                in device memory.  A stream's next chunk is appended when its previous one is done (its
//                                state -- q is the last q_out row, the QP's bound sets, the status -- lives in global
//                                memory in between), so all streams advance together and the launch ends within about
//                                one chunk of its last stream instead of within one whole stream: with a few thousand
//                                resident wavefronts and streams of very different cost that tail was a fifth of the
//                                launch.  Results are bit-identical to the direct mode.
// Queue protocol: ticket i = atomicAdd(head); tickets >= nchunk (the total number of chunks, counted by the init kernel)
// end the wavefront; ring[i] != 0 publishes stream ring[i] - 1 (release store after the chunk's rows and state; the
// consumer's acquire fence follows its poll).  Entries [0, S) are the first chunks, written by the init kernel; push j
// lands in ring[S + j]; the ring has nchunk entries, so no slot is reused.  A ticket below nchunk always gets its
// entry: the pushes still missing belong to chunks held by wavefronts that are running.
// ---------------------------------------------------------------------------------------------
struct WideStreamState { unsigned long long lower, upper; int t_next, stat; };

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// What one launch retargets for ONE (robot, task set): everything the kernel body reads that is not in the image.  A
// plain launch passes one job by value (kernel arguments).  A GROUP launch (several robots in one scheduling domain:
// BASELINE.json configs[3], "one kernel with per-stream model index") keeps a table of jobs in
// device memory; streams are numbered globally (job j owns [first_j, first_j + S_j)), and a wavefront that takes an
// item loads its job -- lane l reads dword l, the fields are taken out with v_readlane, i.e. they are scalars exactly
// like kernel arguments.  Every wide-shape robot runs the SAME kernel instance (the layout constants WD_* are one
// class), so nothing else has to change per item.
struct WideJob {
  const char* img;
  const double* q0;
  const double* human;
  const int32_t* len;
  double* q_out;
  int32_t* nsolve;
  int32_t* status;
  double* tgt_out;
  double* err_out;
  WideDims D;
  int max_iter, human_root, use0, use1, S, T, first;
};
static_assert(sizeof(WideJob) <= 256 && sizeof(WideJob) % 4 == 0, "a job is read as one dword per lane");
constexpr int WD_MAX_JOBS = 8;
struct WideJobTable { WideJob job[WD_MAX_JOBS]; int njobs, total; };

struct WideQueue {
  unsigned* hdr;                 // [0] head (tickets), [1] tail (pushes, starts at the number of streams), [2] nchunk
  unsigned* ring;
  WideStreamState* state;
  int chunk;
  // group launches only: a WINDOW [t_begin, t_end) of every stream's frames (host pipeline: the copies of window w + 1 run under
  // the kernel of window w); `windowed` = the per-stream state persists in `state` from launch to launch
  int t_begin, t_end, windowed;
};

// job of global stream g (MULTI launches): lane l holds first_l, the job is the last one whose first stream is <= g
__device__ __forceinline__ int job_of(const WideJob* __restrict__ jobs, int njobs, int g, int lane) {
  const int first = lane < njobs ? jobs[lane].first : 0x7fffffff;
  return __popcll(__ballot(g >= first)) - 1;
}

// Queue workspace of a launch: per global stream its chunk count (summed into hdr[2]) and initial state, the ring's
// first entries; for group launches also the device copy of the job table (the by-value table is indexed dynamically
// here, i.e. from scratch: irrelevant in this one-shot kernel).  Q.ring == nullptr: only the table is copied.
__global__ void wide_queue_init(WideJobTable tab, WideJob* __restrict__ d_jobs, WideQueue Q, unsigned nring) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (d_jobs) {
    const unsigned nd = (unsigned)tab.njobs * (unsigned)(sizeof(WideJob) / 4);
    if (i < nd) reinterpret_cast<uint32_t*>(d_jobs)[i] = reinterpret_cast<const uint32_t*>(tab.job)[i];
  }
  if (!Q.ring) return;
  unsigned n = 0;
  if (i < (unsigned)tab.total) {
    int j = 0;
    while (j + 1 < tab.njobs && (int)i >= tab.job[j + 1].first) j++;
    const int T = tab.job[j].T;
    const int32_t* len = tab.job[j].len;
    const int Ts = len ? min(max(len[(int)i - tab.job[j].first], 0), T) : T;
    const int Tw = min(Ts, Q.t_end), t0 = Q.windowed ? min(Q.t_begin, Ts) : 0;      // this launch's share of the stream
    n = Tw > t0 ? (unsigned)((Tw - t0 + Q.chunk - 1) / Q.chunk) : 1u;             // nothing to do is one (empty) chunk
    if (!Q.windowed || Q.t_begin == 0) Q.state[i] = WideStreamState{0ull, 0ull, 0, GMR_STATUS_OK};
    // (a later window finds t_next where the previous one stopped: t_begin, or the stream's end)
  }
  if (i < nring) Q.ring[i] = i < (unsigned)tab.total ? i + 1u : 0u;
  for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(&Q.hdr[2], n);
  if (i == 0) Q.hdr[1] = (unsigned)tab.total;
}

// next item of the queue: false when the tickets are used up (the wavefront ends)
__device__ __forceinline__ bool queue_pop(const WideQueue& Q, const unsigned nchunk, const int lane, int& g, int& t0, int& stat,
                                          RowState& bounds) {
  unsigned ticket = 0;
  if (lane == 0) ticket = __hip_atomic_fetch_add(&Q.hdr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ticket = __builtin_amdgcn_readfirstlane(ticket);
  if (ticket >= nchunk) return false;
  unsigned v = 0;
  for (;;) {
    if (lane == 0) v = __hip_atomic_load(&Q.ring[ticket], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = __builtin_amdgcn_readfirstlane(v);
    if (v) break;
    __builtin_amdgcn_s_sleep(64);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the producer's q_out row and state are visible from here
  g = (int)v - 1;
  const WideStreamState st = Q.state[g];
  bounds.lower = uniform64(st.lower); bounds.upper = uniform64(st.upper);
  t0 = __builtin_amdgcn_readfirstlane(st.t_next);
  stat = __builtin_amdgcn_readfirstlane(st.stat);
  return true;
}

// Plain launch: ONE job, its fields are individual kernel arguments.  That matters: with ~100 SGPRs for ~350
// wave-uniform values the compiler re-loads a kernel ARGUMENT where it is used (s_load from the kernarg segment: no VALU
// slot), while a value it cannot re-load is parked in a VGPR lane and comes back through v_readlane.  Passing the job as
// one by-value struct (or reading it through a pointer to the kernarg segment) loses that property: measured, +500
// v_readlane instructions in the code object and -4 % frames/s.
__global__ __launch_bounds__(64, GMR_WIDE_MIN_WAVES) void ik_wide_kernel(
    const char* __restrict__ img, WideDims D, int max_iter, int human_root, int use0, int use1, int S, int T,
    const double* __restrict__ q0, const double* __restrict__ human, const int32_t* __restrict__ len, int flags,
    double* q_out, int32_t* __restrict__ nsolve, int32_t* __restrict__ status, double* __restrict__ tgt_out,
    double* __restrict__ err_out, WideQueue Q, unsigned long long* __restrict__ prof_out) {
  extern __shared__ __align__(16) double sm[];
  const int lane = threadIdx.x;
  const bool queued = Q.ring != nullptr;
  if (!queued && (int)blockIdx.x >= S) return;
  Prof pr;
#ifdef GMR_IK_PROFILE
  for (int i = 0; i < PH_COUNT; i++) pr.acc[i] = 0;
  const unsigned long long k_t0 = __builtin_amdgcn_s_memtime(), k_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int nq = D.nq, nhum = D.nhum;
  const double* prm = img_at<double>(img, IM.prm);   // damping, lm_damping, tol, limit_gain, ground_offset, dt
  const unsigned nchunk = queued ? Q.hdr[2] : 0u;
  const size_t fstride = (size_t)nhum * 7;
  for (;;) {
  int s = blockIdx.x, t0 = 0, stat = GMR_STATUS_OK;
  RowState bounds = {0ull, 0ull};
  if (queued) {
    unsigned ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(&Q.hdr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket >= nchunk) break;
    unsigned v = 0;
    for (;;) {
      if (lane == 0) v = __hip_atomic_load(&Q.ring[ticket], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v = __builtin_amdgcn_readfirstlane(v);
      if (v) break;
      __builtin_amdgcn_s_sleep(64);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the producer's q_out row and state are visible from here
    s = (int)v - 1;
    const WideStreamState st = Q.state[s];
    bounds.lower = uniform64(st.lower); bounds.upper = uniform64(st.upper);
    t0 = __builtin_amdgcn_readfirstlane(st.t_next);
    stat = __builtin_amdgcn_readfirstlane(st.stat);
  }
  const int gs = s;
#define GMR_DIMS_T const WideDims&
#define GMR_T_END T
#define GMR_WINDOWED false
#include "gmr_ik_wide_item.inc"
#undef GMR_WINDOWED
#undef GMR_T_END
#undef GMR_DIMS_T
  }
#ifdef GMR_IK_PROFILE
  pr.acc[PH_TICKS] = __builtin_amdgcn_s_memtime() - k_t0;
  pr.acc[PH_REALTIME] = __builtin_amdgcn_s_memrealtime() - k_r0;
  if (lane == 0 && prof_out && !queued)
    for (int i = 0; i < PH_COUNT; i++) prof_out[(size_t)blockIdx.x * PH_COUNT + i] = pr.acc[i];
#else
  (void)prof_out;
#endif
}

// Group launch: the job of an item comes from the table in device memory, by scalar loads through a constant-memory
// pointer (wave-uniform, like kernel arguments, but not re-loadable for free: this instance keeps a few more values in
// registers than the plain one).
__global__ __launch_bounds__(64, GMR_WIDE_MIN_WAVES) void ik_wide_group_kernel(const WideJob* __restrict__ jobs, int njobs,
                                                                               int total, int flags, WideQueue Q) {
  extern __shared__ __align__(16) double sm[];
  const int lane = threadIdx.x;
  const bool queued = Q.ring != nullptr;
  if (!queued && (int)blockIdx.x >= total) return;
  Prof pr;
#ifdef GMR_IK_PROFILE
  for (int i = 0; i < PH_COUNT; i++) pr.acc[i] = 0;
#endif
  const unsigned nchunk = queued ? Q.hdr[2] : 0u;
  for (;;) {
  int gs = blockIdx.x, t0 = 0, stat = GMR_STATUS_OK;
  RowState bounds = {0ull, 0ull};
  if (queued) {
    if (!queue_pop(Q, nchunk, lane, gs, t0, stat, bounds)) break;
  } else if (Q.windowed && Q.t_begin > 0) {           // a later window of a direct launch: continue from the parked state
    const WideStreamState st = Q.state[gs];
    bounds.lower = uniform64(st.lower); bounds.upper = uniform64(st.upper);
    t0 = __builtin_amdgcn_readfirstlane(st.t_next);
    stat = __builtin_amdgcn_readfirstlane(st.stat);
  }
  typedef const WideJob __attribute__((address_space(4))) CJob;
  const int j = __builtin_amdgcn_readfirstlane(job_of(jobs, njobs, gs, lane));
  CJob* cj = reinterpret_cast<CJob*>(reinterpret_cast<uintptr_t>(jobs + j));
  const WideDims __attribute__((address_space(4)))& D = cj->D;
  const char* __restrict__ img = cj->img;
  const int max_iter = cj->max_iter, human_root = cj->human_root, use0 = cj->use0, use1 = cj->use1, T = cj->T;
  const double* __restrict__ q0 = cj->q0;
  const double* __restrict__ human = cj->human;
  const int32_t* __restrict__ len = cj->len;
  double* q_out = cj->q_out;
  int32_t* __restrict__ nsolve = cj->nsolve;
  int32_t* __restrict__ status = cj->status;
  double* __restrict__ tgt_out = cj->tgt_out;
  double* __restrict__ err_out = cj->err_out;
  const int s = gs - cj->first;
  const int nq = D.nq, nhum = D.nhum;
  const double* prm = img_at<double>(img, IM.prm);
  const size_t fstride = (size_t)nhum * 7;
#define GMR_DIMS_T const WideDims __attribute__((address_space(4)))&
#define GMR_T_END Q.t_end
#define GMR_WINDOWED (Q.windowed != 0)
#include "gmr_ik_wide_item.inc"
#undef GMR_WINDOWED
#undef GMR_T_END
#undef GMR_DIMS_T
  }
}

}  // namespace wide
}  // namespace gmr

// ---------------------------------------------------------------------------------------------
// host side: the launcher used by gmr_abi.hip and the queue workspaces of a solver
// ---------------------------------------------------------------------------------------------
#include <map>
#include <mutex>

namespace {

// diagnostic (occupancy experiments): GMR_WIDE_LDS_PAD=<bytes> makes every launch ask for that much more LDS, so that fewer
// streams are resident per CU -- how throughput scales with resident wavefronts without touching the kernel
inline int wide_lds_launch_bytes() {
  static const int pad = [] { const char* e = getenv("GMR_WIDE_LDS_PAD"); return e ? std::max(0, std::min(atoi(e), 40 * 1024)) : 0; }();
  return gmr::WD_LDS_BYTES + pad;
}

struct QueueWs { char* base = nullptr; size_t bytes = 0; };
struct WidePool {
  std::mutex mu;
  std::map<hipStream_t, QueueWs> ws;       // one workspace per HIP stream: launches on one stream are ordered
  int slots = 0;                           // resident wavefronts of the device (the queued grid)
  int chunk = 4;                           // frames per queue item (measured at nine streams per CU: 1: 15.0, 2: 15.7, 4: 15.8, 8: 15.2 M frames/s at 16 384 x 16)
  bool chunk_auto = true;                  // nobody fixed it (GMR_IK_CHUNK, gmr_solver_set_dispatch): small launches halve it, below
  int min_streams_per_slot = 1;            // queued mode from slots * this + 1 streams (all resident: nothing to balance)
};

}  // namespace

extern "C" void* gmr_ik_wide_pool_create() {
  WidePool* p = new WidePool();
  int dev = 0, ncu = 0, nblk = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, reinterpret_cast<const void*>(gmr::wide::ik_wide_kernel), 64,
                                                   wide_lds_launch_bytes()) != hipSuccess)
    ncu = nblk = 0;
  // (the API assumes 512-byte LDS granules; the device hands out 1 280-byte ones: tools/micro/lds_occupancy.hip)
  nblk = std::min(nblk, 160 * 1024 / ((wide_lds_launch_bytes() + gmr::WD_LDS_GRANULE - 1) / gmr::WD_LDS_GRANULE * gmr::WD_LDS_GRANULE));
  p->slots = ncu * nblk;
  if (const char* e = getenv("GMR_IK_CHUNK")) { p->chunk = atoi(e); p->chunk_auto = false; }      // 0 = always direct
  if (const char* e = getenv("GMR_IK_QUEUE_MIN")) p->min_streams_per_slot = std::max(1, atoi(e));
  return p;
}

extern "C" void gmr_ik_wide_pool_set_chunk(void* pool, int chunk) {
  WidePool* p = static_cast<WidePool*>(pool);
  std::lock_guard<std::mutex> g(p->mu);
  p->chunk = chunk;
  p->chunk_auto = false;
}

extern "C" void gmr_ik_wide_pool_destroy(void* pool) {
  WidePool* p = static_cast<WidePool*>(pool);
  if (!p) return;
  for (auto& kv : p->ws) if (kv.second.base) (void)hipFree(kv.second.base);
  delete p;
}

// one job of a launch as gmr_abi.hip hands it over (device pointers)
struct gmr_wide_job_desc {
  const char* d_image; const gmr::WideLayout* L; const gmr::IkParams* P;
  int S, T;
  const double* d_q0; const double* d_human; const int32_t* d_len;
  double* d_q_out; int32_t* d_nsolve; int32_t* d_status; double* d_tgt_out; double* d_err_out;
};

// Launch njobs >= 1 jobs as ONE scheduling domain on `stream`: a single job runs the plain instance (its fields are
// kernel arguments), several jobs the group instance (job table in the workspace).  Queued dispatch engages when the
// streams of ALL jobs together outnumber the resident wavefronts.
// t_begin / t_end: with t_end > 0 the launch covers the frames [t_begin, t_end) of every stream only and runs the group
// instance; the per-stream state (QP bound sets, next frame, status) stays in the pool's workspace of `stream` for the next
// window, which must follow on the same stream with t_begin = this t_end (the first window has t_begin = 0).
extern "C" hipError_t gmr_launch_ik_wide_window(const gmr_wide_job_desc* jd, int njobs, int flags, hipStream_t stream,
                                                unsigned long long* d_prof, void* pool, int t_begin, int t_end);

extern "C" hipError_t gmr_launch_ik_wide_group(const gmr_wide_job_desc* jd, int njobs, int flags, hipStream_t stream,
                                               unsigned long long* d_prof, void* pool) {
  return gmr_launch_ik_wide_window(jd, njobs, flags, stream, d_prof, pool, 0, 0);
}

extern "C" hipError_t gmr_launch_ik_wide_window(const gmr_wide_job_desc* jd, int njobs, int flags, hipStream_t stream,
                                                unsigned long long* d_prof, void* pool, int t_begin, int t_end) {
  using namespace gmr::wide;
  if (njobs < 1 || njobs > WD_MAX_JOBS) return hipErrorInvalidValue;
  WidePool* p = static_cast<WidePool*>(pool);
  WideJobTable tab;
  memset(&tab, 0, sizeof tab);
  long long total = 0, nring = 0;
  int chunk = 0, maxT = 0, n = 0;
  bool chunk_auto = false;
  if (p) { std::lock_guard<std::mutex> g(p->mu); chunk = p->chunk; chunk_auto = p->chunk_auto; }
  for (int j = 0; j < njobs; j++) {
    if (jd[j].S <= 0 || jd[j].T <= 0) continue;
    WideJob& J = tab.job[n++];
    J.img = jd[j].d_image; J.q0 = jd[j].d_q0; J.human = jd[j].d_human; J.len = jd[j].d_len; J.q_out = jd[j].d_q_out;
    J.nsolve = jd[j].d_nsolve; J.status = jd[j].d_status; J.tgt_out = jd[j].d_tgt_out; J.err_out = jd[j].d_err_out;
    J.D = static_cast<const gmr::WideDims&>(*jd[j].L);
    J.max_iter = jd[j].P->max_iter; J.human_root = jd[j].P->human_root; J.use0 = jd[j].P->use0; J.use1 = jd[j].P->use1;
    J.S = jd[j].S; J.T = jd[j].T; J.first = (int)total;
    total += jd[j].S;
    maxT = std::max(maxT, jd[j].T);
  }
  if (n == 0) return hipSuccess;
  if (total > 0x7fffffffll) return hipErrorInvalidValue;
  tab.njobs = n; tab.total = (int)total;
  const bool windowed = t_end > 0;
  if (windowed && (t_begin < 0 || t_begin >= t_end)) return hipErrorInvalidValue;
  const bool multi = n > 1 || windowed;              // (windows are a feature of the group instance)
  const int span = windowed ? std::min(maxT, t_end) - t_begin : maxT;      // frames per stream in this launch, at most
  // the ring has one entry per chunk: very long jobs get longer chunks rather than a ring beyond 64 MB (more than 2^24
  // streams cannot be helped by longer chunks: the loop ends at chunk >= T and the launch below is a direct one)
  auto ring_entries = [&](int c) {
    long long r = 0;
    for (int j = 0; j < n; j++) {
      const int len = windowed ? std::max(0, std::min(tab.job[j].T, t_end) - t_begin) : tab.job[j].T;
      r += (long long)tab.job[j].S * std::max(1, (len + c - 1) / c);
    }
    return r;
  };
  // A launch that is only a few rounds of items deep ends on a ragged last round: 16 384 streams x 8 frames -- an eighth of
  // the 1M-frame batch, what one of eight GPUs gets -- is 14 rounds of 4-frame items on 2 304 resident wavefronts; with
  // 2-frame items 28 rounds, 13.64 -> 14.18 M frames/s (32 768 x 8: 14.46 -> 14.63; 131 072 x 8 prefers 4: 14.98 vs 14.65)
  if (chunk_auto && chunk == 4 && p && p->slots > 0 && ring_entries(4) < 24ll * p->slots) chunk = 2;
  while (chunk > 0 && chunk < span && ring_entries(chunk) > (1ll << 24)) chunk *= 2;
  // queued mode pays only when streams outnumber the resident wavefronts and have more than one chunk
  const bool queued = p && !d_prof && chunk > 0 && p->slots > 0 && span > chunk && total > (long long)p->slots * p->min_streams_per_slot;
  if (queued) nring = ring_entries(chunk);
  WideQueue Q{nullptr, nullptr, nullptr, 0, 0, 0, 0};
  Q.t_begin = windowed ? t_begin : 0;
  Q.t_end = windowed ? t_end : 0x7fffffff;
  Q.windowed = windowed ? 1 : 0;
  WideJob* d_jobs = nullptr;
  int grid = (int)total;
  if (queued || multi) {
    if (!p) return hipErrorInvalidValue;
    // workspace: [hdr 256 B | job table | per-stream state | ring]  (the state sits in front of the ring: its place must not
    // depend on a window's ring size)
    const size_t o_jobs = 256, o_state = 4096, o_ring = o_state + ((size_t)total * sizeof(WideStreamState) + 255) / 256 * 256;
    static_assert(256 + sizeof(WideJob) * WD_MAX_JOBS <= 4096, "job table must fit in front of the state");
    const size_t bytes = o_ring + (size_t)nring * 4 + 256;
    char* base = nullptr;
    {
      std::lock_guard<std::mutex> g(p->mu);
      QueueWs& w = p->ws[stream];
      if (w.bytes < bytes) {
        hipError_t e = hipSuccess;
        char* grown = nullptr;
        if ((e = hipMalloc((void**)&grown, bytes)) != hipSuccess) return e;
        if (w.base) {
          // a later window that needs a longer ring than the ones before it: the streams' state moves to the new workspace
          if (windowed && t_begin > 0 && w.bytes > o_state)
            e = hipMemcpyAsync(grown + o_state, w.base + o_state, std::min(o_ring, w.bytes) - o_state, hipMemcpyDeviceToDevice, stream);
          if (e == hipSuccess) e = hipStreamSynchronize(stream);
          if (e != hipSuccess) { (void)hipFree(grown); return e; }
          (void)hipFree(w.base);
        }
        w.base = grown;
        w.bytes = bytes;
      }
      base = w.base;
    }
    if (multi) d_jobs = reinterpret_cast<WideJob*>(base + o_jobs);
    Q.state = reinterpret_cast<WideStreamState*>(base + o_state);
    if (queued) {
      Q.hdr = reinterpret_cast<unsigned*>(base);
      Q.ring = reinterpret_cast<unsigned*>(base + o_ring);
      Q.chunk = chunk;
      hipError_t e = hipMemsetAsync(base, 0, 256, stream);
      if (e != hipSuccess) return e;
      grid = (int)std::min<long long>(total, p->slots);
    }
    const unsigned nthr = (unsigned)std::max<long long>(std::max<long long>(nring, queued ? total : 0), (long long)(n * sizeof(WideJob) / 4));
    hipLaunchKernelGGL(wide_queue_init, dim3((nthr + 255) / 256), dim3(256), 0, stream, tab, d_jobs, Q, (unsigned)nring);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (multi) {
    hipLaunchKernelGGL(ik_wide_group_kernel, dim3(grid), dim3(64), wide_lds_launch_bytes(), stream, d_jobs, n, (int)total, flags, Q);
  } else {
    const WideJob& J = tab.job[0];
    hipLaunchKernelGGL(ik_wide_kernel, dim3(grid), dim3(64), wide_lds_launch_bytes(), stream, J.img, J.D, J.max_iter, J.human_root,
                       J.use0, J.use1, J.S, J.T, J.q0, J.human, J.len, flags, J.q_out, J.nsolve, J.status, J.tgt_out, J.err_out, Q,
                       d_prof);
  }
  return hipGetLastError();
}

extern "C" hipError_t gmr_launch_ik_wide(const char* d_image, const gmr::WideLayout* L, const gmr::IkParams* P, int S, int T,
                                         const double* d_q0, const double* d_human, const int32_t* d_len, int flags,
                                         double* d_q_out, int32_t* d_nsolve, int32_t* d_status, double* d_tgt_out,
                                         double* d_err_out, hipStream_t stream, unsigned long long* d_prof, void* pool) {
  if (S <= 0 || T <= 0) return hipSuccess;
  const gmr_wide_job_desc jd{d_image, L, P, S, T, d_q0, d_human, d_len, d_q_out, d_nsolve, d_status, d_tgt_out, d_err_out};
  return gmr_launch_ik_wide_group(&jd, 1, flags, stream, d_prof, pool);
}

// the kernel's registers / LDS as the runtime sees them (occupancy reporting)
extern "C" hipError_t gmr_ik_wide_attributes(int* num_regs, int* lds_bytes, int* max_waves_per_cu) {
  hipFuncAttributes a;
  hipError_t e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(gmr::wide::ik_wide_kernel));
  if (e != hipSuccess) return e;
  if (num_regs) *num_regs = a.numRegs;
  if (lds_bytes) *lds_bytes = gmr::WD_LDS_BYTES;
  int nblk = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, reinterpret_cast<const void*>(gmr::wide::ik_wide_kernel), 64,
                                                   gmr::WD_LDS_BYTES);
  if (max_waves_per_cu) *max_waves_per_cu = nblk;
  return e;
}
