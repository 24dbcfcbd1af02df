// gmr_fk.hip -- post-hoc batched forward kinematics, float32 (rows H8-H9 of SURVEY.md section 8a).
//
// Replaces KinematicsModel.forward_kinematics (reference kinematics_model.py:213-246), which on the
// reference's "cuda:0" path is ~70 tiny ATen launches per body (~2.6k per call), by one kernel.
// This one IS bandwidth-shaped: 4*ndof + 28 B in, 12*nbody (+16*nbody) B out per frame.
//
// Lane mapping: one lane per FRAME, bodies visited in the tree's DFS order by the whole wave
// (wave-uniform control flow, no divergence).  In DFS order a body's first child follows it
// immediately, so the parent transform is usually still in registers; only bodies with two or more
// children are parked in LDS slots [slot][component][lane] (conflict-free; 2-3 slots for a humanoid)
// and reloaded when a later child comes up.  The serial order per frame is exactly the reference's
// (pos_j = pos_p + rot_p * t_j;
// rot_j = rot_p * (r_j * jr_j)), hence the same rounding sequence as its float32 loop.
// Stores: a lane's pieces (12 B of body j) are 456 B apart from its neighbour's, and ~100k frames are
// in flight chip-wide, more than L2 can hold until a line is complete -- written straight from
// registers they reach HBM as partial lines (measured 0.9 TB/s).  The positions-only variant (all the
// dataset scripts consume, smplx_to_robot_dataset.py:110,121) therefore stages the wave's 64 x 456 B
// = 29 KB output block in LDS in its final layout and streams it out with contiguous 16 B-per-lane
// stores, every line written once and whole.  With rotations requested the direct stores are kept.
// The dof row of a frame (116 B) is read element by element but stays L1-resident across the walk.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/gmr_hip.h"
#include "gmr_fk_tree.h"

// float32 arithmetic here mirrors torch eager ops (one rounding per operation): no FMA contraction,
// which also makes both template variants produce the same bits
#pragma clang fp contract(off)

namespace gmr {

struct f4 { float x, y, z, w; };

__device__ __forceinline__ f4 qmul_xyzw(f4 a, f4 b) {
  f4 r;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  return r;
}

// torch_utils.quat_rotate (reference torch_utils.py:65-75), same operation order
__device__ __forceinline__ void qrot_xyzw(f4 q, float vx, float vy, float vz, float& ox, float& oy, float& oz) {
  float s = 2.0f * q.w * q.w - 1.0f;
  float cx = q.y * vz - q.z * vy, cy = q.z * vx - q.x * vz, cz = q.x * vy - q.y * vx;
  float d = q.x * vx + q.y * vy + q.z * vz;
  ox = vx * s + cx * q.w * 2.0f + q.x * d * 2.0f;
  oy = vy * s + cy * q.w * 2.0f + q.y * d * 2.0f;
  oz = vz * s + cz * q.w * 2.0f + q.z * d * 2.0f;
}

// sin and cos of a joint half angle (|x| of a few radians): Cody-Waite reduction by pi/2 in three pieces, the Cephes
// single-precision kernels on [-pi/4, pi/4] (~1 ulp).  28 instructions instead of the 125 of the library call (which
// carries a large-argument path); explicit fmaf, so `fp contract(off)` does not change it.  The three-piece reduction is
// exact only while k * 1.5703125 is (|k| < 2^13): beyond |x| = 1000 rad -- forward_kinematics accepts any user angle,
// torch.sin / torch.cos are accurate for all of them -- and for NaN / Inf (where `(int)k` would be undefined) the
// library call is taken: a wave-rare branch.
__device__ __forceinline__ void sincosf_small(float x, float* sn, float* cs) {
  if (!(fabsf(x) <= 1000.0f)) { sincosf(x, sn, cs); return; }
  const float k = rintf(x * 0.636619772367581343f);
  float r = fmaf(-k, 1.5703125f, x);
  r = fmaf(-k, 4.837512969970703125e-4f, r);
  r = fmaf(-k, 7.54978995489188216e-8f, r);
  const float z = r * r;
  const float ps = fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f);
  const float pc = fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f);
  const float s = fmaf(r * z, ps, r);
  const float c = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  const int q = (int)k;
  const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
  *sn = (q & 2) ? -ss : ss;
  *cs = ((q + 1) & 2) ? -cc : cc;
}

// quat_rotate for a vector with known exact zeros (ZM bit k: component k is exactly 0): the terms of qrot_xyzw that survive,
// in its order -- a local translation is wave-uniform and most have one or two zero components (G1: 31 of 38 bodies)
template <bool ZA, bool ZB> __device__ __forceinline__ float diff_z(float a, float b) {
  if (ZA && ZB) return 0.0f;
  if (ZA) return -b;
  if (ZB) return a;
  return a - b;
}
template <bool ZA, bool ZB, bool ZC> __device__ __forceinline__ float sum3_z(float a, float b, float c) {
  if (ZA && ZB) return c;          // (callers never pass three zeros)
  if (ZA && ZC) return b;
  if (ZB && ZC) return a;
  if (ZA) return b + c;
  if (ZB) return a + c;
  if (ZC) return a + b;
  return a + b + c;
}
template <int ZM>
__device__ __forceinline__ void qrot_sparse(f4 q, float vx, float vy, float vz, float& ox, float& oy, float& oz) {
  constexpr bool zx = ZM & 1, zy = (ZM >> 1) & 1, zz = (ZM >> 2) & 1;
  if (zx && zy && zz) { ox = 0.0f; oy = 0.0f; oz = 0.0f; return; }
  const float s = 2.0f * q.w * q.w - 1.0f;
  const float cx = diff_z<zz, zy>(q.y * vz, q.z * vy), cy = diff_z<zx, zz>(q.z * vx, q.x * vz), cz = diff_z<zy, zx>(q.x * vy, q.y * vx);
  const float d = sum3_z<zx, zy, zz>(q.x * vx, q.y * vy, q.z * vz);
  ox = sum3_z<zx, zy && zz, false>(vx * s, cx * q.w * 2.0f, q.x * d * 2.0f);
  oy = sum3_z<zy, zz && zx, false>(vy * s, cy * q.w * 2.0f, q.y * d * 2.0f);
  oz = sum3_z<zz, zx && zy, false>(vz * s, cz * q.w * 2.0f, q.z * d * 2.0f);
}

// fminf without the two canonicalising v_max the compiler puts in front of llvm.minnum (one instruction per body
// instead of three; NaN handling as v_min_f32 in IEEE mode: a quiet NaN operand loses)
__device__ __forceinline__ float min1(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// a * b for b = (.., b_k, .., b_w) with the two other components exactly zero (K = 0, 1, 2: x, y, z): the terms of
// qmul_xyzw that survive, in its order -- the dropped ones are products with an exact zero, added to or subtracted from the
// running sum without changing it (only the sign of a zero RESULT can differ, and non-finite operands: 0 * inf)
template <int K>
__device__ __forceinline__ f4 qmul_axis(f4 a, float bk, float bw) {
  f4 r;
  if (K == 0) {
    r.x = a.w * bk + a.x * bw; r.y = a.y * bw + a.z * bk; r.z = a.z * bw - a.y * bk; r.w = a.w * bw - a.x * bk;
  } else if (K == 1) {
    r.x = a.x * bw - a.z * bk; r.y = a.w * bk + a.y * bw; r.z = a.x * bk + a.z * bw; r.w = a.w * bw - a.y * bk;
  } else {
    r.x = a.x * bw + a.y * bk; r.y = a.y * bw - a.x * bk; r.z = a.w * bk + a.z * bw; r.w = a.w * bw - a.z * bk;
  }
  return r;
}

// The same for a whole wavefront: when every lane's argument lies in [-0.785, 0.785] -- joint half angles of a humanoid in
// motion almost always do: |angle| <= 90 degrees -- the reduction (k = 0, r = x exactly) and the quadrant selects fall away:
// the same two polynomials on the same r, bit-identical, 14 instead of 30 instructions per hinge body.  One ballot decides.
__device__ __forceinline__ void sincosf_wave(float x, float* sn, float* cs) {
  if (__builtin_amdgcn_ballot_w64(!(fabsf(x) <= 0.785f)) != 0ull) { sincosf_small(x, sn, cs); return; }
  const float z = x * x;
  const float ps = fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f);
  const float pc = fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f);
  *sn = fmaf(x * z, ps, x);
  *cs = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
}

// One body of the walk: world rotation `rot` = prot * (r_j * joint(ang)) and the world offset of its origin
// R(prot) t_j (reference kinematics_model.py:213-246).  Everything in `cur` is wave-uniform (SGPRs), so the record's flags
// select code, not lanes: a local rotation that is exactly (0, 0, 0, 1) (meta bit 1: r_j * x = x) and a hinge axis that is
// exactly +-e_k (meta bits [3:2] = k + 1: the joint quaternion has two exact zeros) skip the products whose factor is an
// exact 0 or 1.  Every shipped robot's hinges are axis-aligned and three quarters of the bodies carry no local rotation:
// 186 -> ~115 vector instructions for such a body, in a walk that is bound by instruction issue.
__device__ __forceinline__ void fk_body(const FkBodyRec& cur, float ang, f4 prot, float& wx, float& wy, float& wz, f4& rot) {
  const f4 lr = {cur.r[0], cur.r[1], cur.r[2], cur.r[3]};
  const bool unit_lr = cur.meta & 2u;
  const unsigned kind = (cur.meta >> 2) & 3u;
  switch ((cur.next_park >> 16) & 7u) {     // which components of t_j are exactly zero (record flags, wave-uniform)
    case 1: qrot_sparse<1>(prot, cur.t[0], cur.t[1], cur.t[2], wx, wy, wz); break;
    case 2: qrot_sparse<2>(prot, cur.t[0], cur.t[1], cur.t[2], wx, wy, wz); break;
    case 3: qrot_sparse<3>(prot, cur.t[0], cur.t[1], cur.t[2], wx, wy, wz); break;
    case 4: qrot_sparse<4>(prot, cur.t[0], cur.t[1], cur.t[2], wx, wy, wz); break;
    case 5: qrot_sparse<5>(prot, cur.t[0], cur.t[1], cur.t[2], wx, wy, wz); break;
    case 6: qrot_sparse<6>(prot, cur.t[0], cur.t[1], cur.t[2], wx, wy, wz); break;
    case 7: qrot_sparse<7>(prot, cur.t[0], cur.t[1], cur.t[2], wx, wy, wz); break;
    default: qrot_xyzw(prot, cur.t[0], cur.t[1], cur.t[2], wx, wy, wz); break;
  }
  if (cur.meta & 1u) {
    // dof_to_rot: sin/cos of the float32 half angle; products and the normalisation in float64;
    // rounded to float32 on assignment (kinematics_model.py:21-36, torch_utils.py:353-359).
    // the record's axis is normalize(axis) (float64, computed once on the host).
    float th = ang / 2.0f;
    float sf, cf;
#ifdef GMR_FK_LIBM_SINCOS
    sincosf(th, &sf, &cf);
#else
    sincosf_wave(th, &sf, &cf);
#endif
    double s = (double)sf, c = (double)cf;
    if (kind) {
      const double qk = cur.axis[0] * s, qw = c;      // (the record of such a hinge carries its +-1.0 in axis[0])
      const double e = fma(qk, qk, fma(qw, qw, -1.0));
      const double rn = fma(e, fma(e, 0.375, -0.5), 1.0);
      const float jk = (float)(qk * rn), jw = (float)(qw * rn);
      if (unit_lr) {
        rot = kind == 1 ? qmul_axis<0>(prot, jk, jw) : (kind == 2 ? qmul_axis<1>(prot, jk, jw) : qmul_axis<2>(prot, jk, jw));
      } else {
        const f4 cr = kind == 1 ? qmul_axis<0>(lr, jk, jw) : (kind == 2 ? qmul_axis<1>(lr, jk, jw) : qmul_axis<2>(lr, jk, jw));
        rot = qmul_xyzw(prot, cr);
      }
      return;
    }
    double qx = cur.axis[0] * s, qy = cur.axis[1] * s, qz = cur.axis[2] * s, qw = c;
    // quat_unit in float64: x / |q|.  |q|^2 = 1 + e with |e| ~ 1e-7 (float32 sin / cos of one angle, a unit axis), so
    // 1 / |q| = 1 - e/2 + 3 e^2 / 8 to 1e-21: the quotient differs from x / sqrt(|q|^2) by < 1 ulp of float64 and
    // rounds to the same float32 (measured bit-equal with the rsqrt form on 2^20 random frames, tools/fk_bitcheck.py)
#ifdef GMR_FK_RSQRT_NORM
    double rn = rsqrt(fmax(qx * qx + qy * qy + qz * qz + qw * qw, 1e-18));
#else
    const double e = fma(qx, qx, fma(qy, qy, fma(qz, qz, fma(qw, qw, -1.0))));
    const double rn = fma(e, fma(e, 0.375, -0.5), 1.0);
#endif
    f4 jr = {(float)(qx * rn), (float)(qy * rn), (float)(qz * rn), (float)(qw * rn)};
    rot = qmul_xyzw(prot, unit_lr ? jr : qmul_xyzw(lr, jr));
    return;
  }  // no joint: r_j * (0,0,0,1) == r_j exactly
  rot = unit_lr ? prot : qmul_xyzw(prot, lr);
}

constexpr int FK_BLOCK = 64;  // one wave per block

template <bool STAGED>
__global__ __launch_bounds__(FK_BLOCK) void fk_batch_kernel(const FkTree* __restrict__ tree, int B,
                                                            const float* __restrict__ root_pos,
                                                            const float* __restrict__ root_rot,
                                                            const float* __restrict__ dof,
                                                            float* __restrict__ body_pos,
                                                            float* __restrict__ body_rot,
                                                            float* __restrict__ min_part) {
  // slots [nslot][SW][FK_BLOCK], then (STAGED) pos [FK_BLOCK][nb*3], rot [FK_BLOCK][nb*4].  A parked parent needs
  // SW = 7 floats (pos, rot) when results go straight to memory; when they are staged, the parent's position is
  // simply re-read from the staging area and only the rotation is parked (conflict-free columns): SW = 4
  extern __shared__ __align__(16) float fsm[];
  __shared__ float red[FK_BLOCK / 64];
  const int nb = tree->nbody, ndof = tree->ndof;
  const int tid = threadIdx.x;
  const long long f = (long long)blockIdx.x * FK_BLOCK + tid;
  const bool on = f < B;
  const long long fc = on ? f : 0;
  const int SW = STAGED ? 4 : 7;
  float* stk = fsm + tid;
  const int row = nb * 3;
  float* outb = fsm + tree->nslot * SW * FK_BLOCK;  // used when STAGED
  float* outr = outb + FK_BLOCK * row;             // used when STAGED and body_rot
  const int rrow = nb * 4;
  float zmin = INFINITY;
  float cpx, cpy, cpz;   // transform of the body visited last (the parent of a first child)
  f4 crot;
  if (STAGED && ndof > 0) {
    // Park dof d of this lane's frame in its staging row at the slot of the body it drives: the walk reads it there
    // right before it overwrites the slot with that body's position.  (Read per body during the walk, a lane's 116-B
    // row shares no line with its neighbours': 64 lines per load and 41 KB of lines per CU in flight between two uses
    // -> the vector L1 thrashes, 12x the L2 reads, 42 % of the wave's cycles waiting: profiles/r02_fk_*.)
    // Every lane reads ITS row, all loads issued back to back (independent: one trip to L2, each line fetched once
    // while its 29 users are in flight), and parks the values in its own staging row -- no cross-lane traffic.
    const float* drow0 = dof + fc * ndof;
    float* orow = outb + tid * row;
#pragma unroll 8
    for (int d = 0; d < ndof; d++) orow[3 * tree->dof_body[d]] = drow0[d];
  }
  {
    float px = root_pos[fc * 3], py = root_pos[fc * 3 + 1], pz = root_pos[fc * 3 + 2];
    f4 rot = {root_rot[fc * 4], root_rot[fc * 4 + 1], root_rot[fc * 4 + 2], root_rot[fc * 4 + 3]};
    cpx = px; cpy = py; cpz = pz; crot = rot;
    if (tree->save_slot[0] >= 0 && SW > 0) {
      float* sl = stk + tree->save_slot[0] * SW * FK_BLOCK;
      if (SW == 7) { sl[0] = px; sl[FK_BLOCK] = py; sl[2 * FK_BLOCK] = pz; sl += 3 * FK_BLOCK; }
      sl[0] = rot.x; sl[FK_BLOCK] = rot.y; sl[2 * FK_BLOCK] = rot.z; sl[3 * FK_BLOCK] = rot.w;
    }
    if (STAGED) {
      float* o = outb + tid * row;
      o[0] = px; o[1] = py; o[2] = pz;
      if (body_rot) *reinterpret_cast<float4*>(outr + tid * rrow) = make_float4(rot.x, rot.y, rot.z, rot.w);
      if (on) zmin = pz;
    } else if (on) {
      float* op = body_pos + f * nb * 3;
      op[0] = px; op[1] = py; op[2] = pz;
      if (body_rot) *reinterpret_cast<float4*>(body_rot + f * nb * 4) = make_float4(rot.x, rot.y, rot.z, rot.w);
      zmin = pz;
    }
  }
  const float* drow = dof + fc * ndof;
  FkBodyRec nxt = tree->rec[nb > 1 ? 1 : 0];
  float ang_nxt = STAGED ? (outb + tid * row)[3] : 0.0f;     // parked joint angle of body 1 (garbage if it has no joint: unused)
  for (int j = 1; j < nb; j++) {
    // wave-uniform tree data: ONE 64-byte scalar load per body, issued one body ahead; so is the parked joint angle
    const FkBodyRec cur = nxt;
    nxt = tree->rec[j + 1 < nb ? j + 1 : j];
    const float ang = ang_nxt;
    if (STAGED) ang_nxt = (outb + tid * row)[3 * (j + 1 < nb ? j + 1 : j)];
    const int src = (int)((cur.meta >> 8) & 255u) - 1, dst = (int)((cur.meta >> 16) & 255u) - 1;
    float ppx = cpx, ppy = cpy, ppz = cpz;
    f4 prot = crot;
    if (src >= 0) {   // wave-uniform: this body is not the first child of the body before it
      if (STAGED) {
        const int pj = (int)(cur.meta >> 24);
        const float* po = outb + tid * row + 3 * pj;
        ppx = po[0]; ppy = po[1]; ppz = po[2];
        const float* par = stk + src * 4 * FK_BLOCK;
        prot = f4{par[0], par[FK_BLOCK], par[2 * FK_BLOCK], par[3 * FK_BLOCK]};
      } else {
        const float* par = stk + src * 7 * FK_BLOCK;
        ppx = par[0]; ppy = par[FK_BLOCK]; ppz = par[2 * FK_BLOCK];
        prot = f4{par[3 * FK_BLOCK], par[4 * FK_BLOCK], par[5 * FK_BLOCK], par[6 * FK_BLOCK]};
      }
    }
    float wx, wy, wz;
    f4 rot;
    fk_body(cur, STAGED ? ang : ((cur.meta & 1u) ? drow[cur.dof_idx] : 0.0f), prot, wx, wy, wz, rot);
    float px = ppx + wx, py = ppy + wy, pz = ppz + wz;
    cpx = px; cpy = py; cpz = pz; crot = rot;
    if (dst >= 0 && SW > 0) {
      float* cur = stk + dst * SW * FK_BLOCK;
      if (SW == 7) { cur[0] = px; cur[FK_BLOCK] = py; cur[2 * FK_BLOCK] = pz; cur += 3 * FK_BLOCK; }
      cur[0] = rot.x; cur[FK_BLOCK] = rot.y; cur[2 * FK_BLOCK] = rot.z; cur[3 * FK_BLOCK] = rot.w;
    }
    if (STAGED) {
      float* o = outb + tid * row + 3 * j;
      o[0] = px; o[1] = py; o[2] = pz;
      if (body_rot) *reinterpret_cast<float4*>(outr + tid * rrow + 4 * j) = make_float4(rot.x, rot.y, rot.z, rot.w);   // one 16-B store: 2-way instead of 8-way bank conflicts
      if (on) zmin = fminf(zmin, pz);
    } else if (on) {
      float* op = body_pos + (f * nb + j) * 3;
      op[0] = px; op[1] = py; op[2] = pz;
      if (body_rot) *reinterpret_cast<float4*>(body_rot + (f * nb + j) * 4) = make_float4(rot.x, rot.y, rot.z, rot.w);
      zmin = fminf(zmin, pz);
    }
  }
  if (STAGED) {
    // the block's output is one contiguous range of body_pos: stream it out 16 B per lane
    __syncthreads();
    const long long f0 = (long long)blockIdx.x * FK_BLOCK;
    const long long nfr = (B - f0) < FK_BLOCK ? (B - f0) : FK_BLOCK;
    const int nfloat = (int)(nfr * row);
    float* gdst = body_pos + f0 * row;
    const int nvec = nfloat >> 2;
    {
      // four independent LDS reads in flight per trip (a read -> wait -> store loop costs a full LDS round trip per 16 B)
      const float4* sv = reinterpret_cast<const float4*>(outb);
      float4* dv = reinterpret_cast<float4*>(gdst);
      int i = tid;
      for (; i + 3 * FK_BLOCK < nvec; i += 4 * FK_BLOCK) {
        const float4 a0 = sv[i], a1 = sv[i + FK_BLOCK], a2 = sv[i + 2 * FK_BLOCK], a3 = sv[i + 3 * FK_BLOCK];
        dv[i] = a0; dv[i + FK_BLOCK] = a1; dv[i + 2 * FK_BLOCK] = a2; dv[i + 3 * FK_BLOCK] = a3;
      }
      for (; i < nvec; i += FK_BLOCK) dv[i] = sv[i];
    }
    for (int i = (nvec << 2) + tid; i < nfloat; i += FK_BLOCK) gdst[i] = outb[i];
    if (body_rot) {
      float4* rdst = reinterpret_cast<float4*>(body_rot + f0 * rrow);
      const int nrv = (int)(nfr * nb);
      const float4* sv = reinterpret_cast<const float4*>(outr);
      int i = tid;
      for (; i + 3 * FK_BLOCK < nrv; i += 4 * FK_BLOCK) {
        const float4 a0 = sv[i], a1 = sv[i + FK_BLOCK], a2 = sv[i + 2 * FK_BLOCK], a3 = sv[i + 3 * FK_BLOCK];
        rdst[i] = a0; rdst[i + FK_BLOCK] = a1; rdst[i + 2 * FK_BLOCK] = a2; rdst[i + 3 * FK_BLOCK] = a3;
      }
      for (; i < nrv; i += FK_BLOCK) rdst[i] = sv[i];
    }
  }
  if (min_part) {
    for (int off = 32; off > 0; off >>= 1) zmin = fminf(zmin, __shfl_xor(zmin, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = zmin;
    __syncthreads();
    if (tid == 0) {
      float m = red[0];
      for (int i = 1; i < FK_BLOCK / 64; i++) m = fminf(m, red[i]);
      min_part[blockIdx.x] = m;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The split walk: a block is still 64 frames (lane = frame) with its outputs staged in LDS, but up to four wavefronts
// walk it, each its part of the tree (FkTree::wrec): the chain of ancestors its subtrees hang from -- recomputed by
// every wavefront that needs it, a handful of bodies -- and then the subtrees themselves.  The staging area (456 B per
// frame for the G1) limits a CU to five blocks whatever the kernel does; with one wavefront per block that is 1.25
// wavefronts per SIMD walking 38 dependent bodies each (the wave issues 43 % of its cycles, profiles/r02_fk_*); with
// four it is 5 per SIMD walking ~12.  Parents that are not the body walked just before come from per-wavefront slots
// (position and rotation): no wavefront reads what another one wrote before the final barrier.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * FK_MAX_WAVES) void fk_split_kernel(const FkTree* __restrict__ tree, int B,
                                                                     const float* __restrict__ root_pos,
                                                                     const float* __restrict__ root_rot,
                                                                     const float* __restrict__ dof,
                                                                     float* __restrict__ body_pos,
                                                                     float* __restrict__ body_rot,
                                                                     float* __restrict__ min_part) {
  extern __shared__ __align__(16) float fsm[];     // slots [nslot_split][7][64], pos [64][nb*3], rot [64][nb*4]
  __shared__ float red[FK_MAX_WAVES];
  const int nb = tree->nbody, ndof = tree->ndof, nwave = tree->nwave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), tid = threadIdx.x & 63;
  const long long f = (long long)blockIdx.x * 64 + tid;
  const bool on = f < B;
  const long long fc = on ? f : 0;
  float* stk = fsm + tid;
  const int row = nb * 3, rrow = nb * 4;
  float* outb = fsm + tree->nslot_split * 7 * 64;
  float* outr = outb + 64 * row;
  float* xtra = outr + (body_rot ? 64 * rrow : 0) + tid;      // extra angle columns [k][lane] (FkTree::nextra)
  float* orow = outb + tid * row;
  // (the lane's staging rows as 32-bit indices into fsm: through the generic pointers the per-body address is a 64-bit
  //  multiply-add, a quarter-rate instruction)
  const unsigned orow_i = (unsigned)(tree->nslot_split * 7 * 64) + (unsigned)(tid * row);
  const unsigned rrow_i = (unsigned)(tree->nslot_split * 7 * 64 + 64 * row) + (unsigned)(tid * rrow);
  const float* drow = dof + fc * ndof;
  const int i0 = tree->wave_start[wave], i1 = tree->wave_start[wave + 1];
  float zmin = INFINITY;
  // Every joint angle this wavefront needs, read from the lane's dof row in ONE batch -- all loads issued back to back,
  // one wait -- and parked in LDS (FkTree::wave_park): the x slot of the body's own position in the lane's staging row, a
  // free y / z slot for an ancestor another wavefront stores, or an extra column.  The frame's root pose travels with the
  // same batch.  (The loop this replaces loaded, waited and stored angle by angle behind two dependent scalar loads
  // each -- the compiler could not batch them -- : a quarter of a block's lifetime was that serial chain.  Measured and
  // not kept: the block's 64 dof rows as ONE coalesced read by all wavefronts, scattered to the parking places, then a
  // barrier -- 0.230 against 0.207 ms: the per-lane loads are not what the block waits for, the extra barrier is felt.)
  const float rpx = root_pos[fc * 3], rpy = root_pos[fc * 3 + 1], rpz = root_pos[fc * 3 + 2];
  const f4 rrot = {root_rot[fc * 4], root_rot[fc * 4 + 1], root_rot[fc * 4 + 2], root_rot[fc * 4 + 3]};
  {
    const uint32_t* pk = tree->wave_park[wave];
    for (int c = 0; c < 32; c += 16) {
      uint32_t w[16];
#pragma unroll
      for (int k = 0; k < 16; k++) w[k] = pk[c + k];
      if (w[0] == 0xffffffffu) break;
      float a[16];
#pragma unroll
      for (int k = 0; k < 16; k++) a[k] = w[k] != 0xffffffffu ? drow[w[k] & 0xffffu] : 0.0f;
#pragma unroll
      for (int k = 0; k < 16; k++)
        if (w[k] != 0xffffffffu) *((w[k] >> 31) ? xtra + 64 * ((w[k] >> 16) & 0x7fffu) : orow + (w[k] >> 16)) = a[k];
    }
  }
  float cpx, cpy, cpz;
  f4 crot;
  float spx = 0.0f, spy = 0.0f, spz = 0.0f;        // the wavefront's first parked parent stays in registers (slot code 254):
  f4 srot = {0.0f, 0.0f, 0.0f, 1.0f};              // for the shipped trees no LDS slot is left, one more block fits a CU
  float ang_nxt = 0.0f;
  {
    const FkBodyRec r0 = tree->wrec[i0];           // body 0 opens every list
    if (r0.next_park >> 31) ang_nxt = *((r0.next_park & 0x40000000u) ? xtra + 64 * (r0.next_park & 0xffffu) : orow + (r0.next_park & 0xffffu));
    const float px = rpx, py = rpy, pz = rpz;
    const f4 rot = rrot;
    cpx = px; cpy = py; cpz = pz; crot = rot;
    const int dst = (int)((r0.meta >> 16) & 255u) - 1;
    if (dst == 254) { spx = px; spy = py; spz = pz; srot = rot; }
    else if (dst >= 0) {
      float* sl = stk + dst * 7 * 64;
      sl[0] = px; sl[64] = py; sl[128] = pz; sl[192] = rot.x; sl[256] = rot.y; sl[320] = rot.z; sl[384] = rot.w;
    }
    if (r0.meta & 16u) {
      orow[0] = px; orow[1] = py; orow[2] = pz;
      if (body_rot) *reinterpret_cast<float4*>(&fsm[rrow_i]) = make_float4(rot.x, rot.y, rot.z, rot.w);
      zmin = pz;                                   // (lanes beyond B walk frame 0 again: the minimum over all frames is the same)
    }
  }
  FkBodyRec nxt = tree->wrec[i0 + 1 < i1 ? i0 + 1 : i0];
#ifdef GMR_FK_EXP_NOWALK
  for (int i = i1; i < i1; i++) {
#else
  for (int i = i0 + 1; i < i1; i++) {
#endif
    const FkBodyRec cur = nxt;                     // one 64-byte scalar load per body, issued one body ahead
    // (the same record through the vector memory path -- a broadcast read straight into vector registers, which is
    //  where the packed FP32 instructions need it -- measured slower: 0.294 vs 0.241 ms)
    nxt = tree->wrec[i + 1 < i1 ? i + 1 : i];
    const int j = (int)(cur.meta >> 24);
    const int src = (int)((cur.meta >> 8) & 255u) - 1, dst = (int)((cur.meta >> 16) & 255u) - 1;
    const bool own = cur.meta & 16u;
    if (cur.meta & 32u) __syncthreads();           // shared trunk: the transforms this wavefront continues from are parked (fk_build_split)
    const float ang = ang_nxt;                     // parked angle of this body, read while the previous body was walked
    if (cur.next_park >> 31) ang_nxt = *((cur.next_park & 0x40000000u) ? xtra + 64 * (cur.next_park & 0xffffu) : orow + (cur.next_park & 0xffffu));
    float ppx = cpx, ppy = cpy, ppz = cpz;
    f4 prot = crot;
    if (src == 254) { ppx = spx; ppy = spy; ppz = spz; prot = srot; }
    else if (src >= 0) {                           // wave-uniform: the parent is not the body walked just before
      const float* par = stk + src * 7 * 64;
      ppx = par[0]; ppy = par[64]; ppz = par[128];
      prot = f4{par[192], par[256], par[320], par[384]};
    }
    float wx, wy, wz;
    f4 rot;
    fk_body(cur, ang, prot, wx, wy, wz, rot);
    const float px = ppx + wx, py = ppy + wy, pz = ppz + wz;
    cpx = px; cpy = py; cpz = pz; crot = rot;
    if (dst == 254) { spx = px; spy = py; spz = pz; srot = rot; }
    else if (dst >= 0) {
      float* sl = stk + dst * 7 * 64;
      sl[0] = px; sl[64] = py; sl[128] = pz; sl[192] = rot.x; sl[256] = rot.y; sl[320] = rot.z; sl[384] = rot.w;
    }
    if (const unsigned xd = (cur.meta >> 6) & 3u) {  // a parent other wavefronts continue from after the barrier
      float* sl = stk + (xd - 1) * 7 * 64;
      sl[0] = px; sl[64] = py; sl[128] = pz; sl[192] = rot.x; sl[256] = rot.y; sl[320] = rot.z; sl[384] = rot.w;
    }
    if (own) {
      const unsigned oi = orow_i + 3u * (unsigned)j;
      fsm[oi] = px; fsm[oi + 1] = py; fsm[oi + 2] = pz;
      if (body_rot) *reinterpret_cast<float4*>(&fsm[rrow_i + 4u * (unsigned)j]) = make_float4(rot.x, rot.y, rot.z, rot.w);   // one 16-B store: 2-way instead of 8-way bank conflicts
      zmin = min1(zmin, pz);
    }
  }
  if (tree->wave_tail_barrier[wave]) __syncthreads();   // (a wavefront whose list ended before the block's mid-walk barrier)
  __syncthreads();
#ifdef GMR_FK_EXP_NOFLUSH
  if (B > 0) { if (zmin == 12345.f) body_pos[0] = outb[threadIdx.x]; return; }
#endif
  // the block's output is one contiguous range of body_pos (and body_rot): all wavefronts stream it out, 16 B per lane
  const int nthr = 64 * nwave, t = threadIdx.x;
  const long long f0 = (long long)blockIdx.x * 64;
  const long long nfr = (B - f0) < 64 ? (B - f0) : 64;
  {
    const int nfloat = (int)(nfr * row), nvec = nfloat >> 2;
    float* gdst = body_pos + f0 * row;
    const float4* sv = reinterpret_cast<const float4*>(outb);
    float4* dv = reinterpret_cast<float4*>(gdst);
    int i = t;
    for (; i + 3 * nthr < nvec; i += 4 * nthr) {
      const float4 a0 = sv[i], a1 = sv[i + nthr], a2 = sv[i + 2 * nthr], a3 = sv[i + 3 * nthr];
      dv[i] = a0; dv[i + nthr] = a1; dv[i + 2 * nthr] = a2; dv[i + 3 * nthr] = a3;
    }
    for (; i < nvec; i += nthr) dv[i] = sv[i];
    for (int e = (nvec << 2) + t; e < nfloat; e += nthr) gdst[e] = outb[e];
  }
  if (body_rot) {
    float4* rdst = reinterpret_cast<float4*>(body_rot + f0 * rrow);
    const int nrv = (int)(nfr * nb);
    const float4* sv = reinterpret_cast<const float4*>(outr);
    int i = t;
    for (; i + 3 * nthr < nrv; i += 4 * nthr) {
      const float4 a0 = sv[i], a1 = sv[i + nthr], a2 = sv[i + 2 * nthr], a3 = sv[i + 3 * nthr];
      rdst[i] = a0; rdst[i + nthr] = a1; rdst[i + 2 * nthr] = a2; rdst[i + 3 * nthr] = a3;
    }
    for (; i < nrv; i += nthr) rdst[i] = sv[i];
  }
  if (min_part) {
    for (int off = 32; off > 0; off >>= 1) zmin = fminf(zmin, __shfl_xor(zmin, off, 64));
    if (tid == 0) red[wave] = zmin;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = red[0];
      for (int w = 1; w < nwave; w++) m = fminf(m, red[w]);
      min_part[blockIdx.x] = m;
    }
  }
}

__global__ __launch_bounds__(256) void min_reduce_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float z = INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) z = fminf(z, part[i]);
  for (int off = 32; off > 0; off >>= 1) z = fminf(z, __shfl_xor(z, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = z;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
}

// min over the frames [seg_start[g], seg_start[g + 1]) and all bodies of body_pos z: one block per segment (clip).
// The dataset scripts' height adjustment is per clip (smplx_to_robot_dataset.py:118-126); with many clips
// concatenated into one FK launch this is their reduction.  An empty segment yields +inf.
__global__ __launch_bounds__(256) void fk_segment_min_kernel(const float* __restrict__ body_pos, int nbody,
                                                              const int32_t* __restrict__ seg_start, float* __restrict__ seg_min) {
  __shared__ float red[4];
  const long long i0 = (long long)seg_start[blockIdx.x] * nbody, i1 = (long long)seg_start[blockIdx.x + 1] * nbody;
  float z = INFINITY;
  for (long long i = i0 + threadIdx.x; i < i1; i += 256) z = fminf(z, body_pos[3 * i + 2]);
  for (int off = 32; off > 0; off >>= 1) z = fminf(z, __shfl_xor(z, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = z;
  __syncthreads();
  if (threadIdx.x == 0) seg_min[blockIdx.x] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
}

}  // namespace gmr

extern "C" hipError_t gmr_launch_fk_segment_min(const float* d_body_pos, int nbody, const int32_t* d_seg_start, int nseg,
                                                float* d_seg_min, hipStream_t stream) {
  if (nseg <= 0) return hipSuccess;
  hipLaunchKernelGGL(gmr::fk_segment_min_kernel, dim3(nseg), dim3(256), 0, stream, d_body_pos, nbody, d_seg_start, d_seg_min);
  return hipGetLastError();
}

// blocks of a launch over B frames: one wave per 64 frames; also the size of the min-z partials
extern "C" int gmr_fk_blocks(int B) { return (B + gmr::FK_BLOCK - 1) / gmr::FK_BLOCK; }

// > 64 KB of dynamic LDS needs an opt-in per kernel and per DEVICE: done once for every device this process uses
static hipError_t fk_opt_in_large_lds() {
  static unsigned long long done_mask = 0;       // bit = device ordinal (benign if two threads race: the call is idempotent)
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 64 && ((done_mask >> dev) & 1ull)) return hipSuccess;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(gmr::fk_batch_kernel<true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gmr::fk_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024 - 1024);
  if (e == hipSuccess && dev < 64) __atomic_fetch_or(&done_mask, 1ull << dev, __ATOMIC_RELAXED);
  return e;
}

extern "C" hipError_t gmr_launch_fk_batch(const gmr::FkTree* d_tree, const gmr::FkTree* h_tree, int B,
                                          const float* d_root_pos, const float* d_root_rot, const float* d_dof,
                                          float* d_body_pos, float* d_body_rot, float* d_min_part, float* d_min_z,
                                          hipStream_t stream) {
  if (B <= 0) return hipSuccess;
  const int nbody = h_tree->nbody, nslot = h_tree->nslot;
  const int blocks = gmr_fk_blocks(B);
  size_t smem = (size_t)nslot * 7 * gmr::FK_BLOCK * sizeof(float);   // direct variant: parked (pos, rot)
  // 16-B aligned destinations: LDS-staged, fully coalesced output
  // (the block strides FK_BLOCK * nbody * 12 B and * 16 B are multiples of 16)
  const size_t stage_bytes = (size_t)gmr::FK_BLOCK * nbody * (d_body_rot ? 7 : 3) * sizeof(float);
  bool staged = (reinterpret_cast<uintptr_t>(d_body_pos) & 15u) == 0 && (reinterpret_cast<uintptr_t>(d_body_rot) & 15u) == 0 &&
                stage_bytes <= 160 * 1024 - 8192;
  if (staged) {
    // staged variant: parent positions are re-read from the staging area; only rotations are parked
    smem = (size_t)nslot * 4 * gmr::FK_BLOCK * sizeof(float) + stage_bytes;
    if (smem > 64 * 1024 && fk_opt_in_large_lds() != hipSuccess) {   // no opt-in on this device: the direct variant still works
      (void)hipGetLastError();
      staged = false;
      smem = (size_t)nslot * 7 * gmr::FK_BLOCK * sizeof(float);
    }
  }
  // several wavefronts per block when the tree splits (gmr_fk_create) and the slots of the split walk fit beside the staging area
  const size_t smem_split = (size_t)h_tree->nslot_split * 7 * 64 * sizeof(float) + stage_bytes + (size_t)h_tree->nextra * 64 * sizeof(float);
  if (staged && h_tree->nwave > 1 && smem_split <= 160 * 1024 - 8192 && (smem_split <= 64 * 1024 || fk_opt_in_large_lds() == hipSuccess))
    hipLaunchKernelGGL(gmr::fk_split_kernel, dim3(blocks), dim3(64 * h_tree->nwave), smem_split, stream, d_tree, B,
                       d_root_pos, d_root_rot, d_dof, d_body_pos, d_body_rot, d_min_z ? d_min_part : nullptr);
  else if (staged)
    hipLaunchKernelGGL(gmr::fk_batch_kernel<true>, dim3(blocks), dim3(gmr::FK_BLOCK), smem, stream, d_tree, B,
                       d_root_pos, d_root_rot, d_dof, d_body_pos, d_body_rot, d_min_z ? d_min_part : nullptr);
  else
    hipLaunchKernelGGL(gmr::fk_batch_kernel<false>, dim3(blocks), dim3(gmr::FK_BLOCK), smem, stream, d_tree, B,
                       d_root_pos, d_root_rot, d_dof, d_body_pos, d_body_rot, d_min_z ? d_min_part : nullptr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (d_min_z) {
    hipLaunchKernelGGL(gmr::min_reduce_kernel, dim3(1), dim3(256), 0, stream, d_min_part, blocks, d_min_z);
    e = hipGetLastError();
  }
  return e;
}
