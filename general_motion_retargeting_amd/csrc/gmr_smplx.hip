// gmr_smplx.hip -- SMPL-X frame extraction on the device: the step in
// front of the retargeting loop.  Replaces, for a whole clip in one launch each,
//   * the joints-only part of the body-model forward pass called at utils/smpl.py:12-34
//     (smplx_joints_kernel; the mesh, 10 475 vertices, is never needed by GMR), and
//   * get_smplx_data_offline_fast / get_smplx_data (utils/smpl.py:44-197): SLERP fps alignment of every
//     local joint rotation, linear interpolation of the joint positions, chain of global orientations
//     (smplx_align_kernel), writing the packed human[T'][nhuman][7] frames the IK kernel consumes.
//
// Mapping: lane = output frame.  The tree is walked in DFS order (host-built program); the transform of the
// ancestor at depth d sits in a lane-private LDS column stack[d][.][lane] (conflict-free), so LDS holds
// max_depth, not J, transforms.  Both kernels are streaming (read 2 pose rows + 2 joint rows, write nsel x 56 B
// per frame); the transcendental chain per joint makes them FP64-VALU bound well below the HBM roof, three
// orders of magnitude above the IK kernel's frame rate, so they are written for exactness, not tuned.
//
// Arithmetic follows the SciPy formulas the reference goes through (from_rotvec / as_rotvec small-angle
// series at 1e-3, from_quat normalisation, interp1d's float32 difference).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/gmr_hip.h"
#include "gmr_internal.h"

#define SX_MAX_JOINTS 64
#define SX_BLOCK 64

namespace gmr {

struct SmplxProg {
  int n;                         // joints visited (DFS order)
  int J;                         // joints per pose row
  int nrow;                      // rows written per frame
  int max_depth;                 // stack levels
  short joint[SX_MAX_JOINTS];    // joint index of step k
  short depth[SX_MAX_JOINTS];    // its depth (root = 0)
  short row[SX_MAX_JOINTS];      // output row or -1
  short parent[SX_MAX_JOINTS];   // parent JOINT index of step k (joints kernel: rest offsets)
  // joints kernel: the parent's transform is the previous step's (still in registers: load = -1) or parked in an LDS slot;
  // only joints with two or more visited children are parked (save >= 0).  SMPL-X: 3 slots instead of 11 stack levels
  short load[SX_MAX_JOINTS], save[SX_MAX_JOINTS];
  int nslot;
};

struct q4 { double x, y, z, w; };   // xyzw like SciPy

__device__ __forceinline__ q4 sx_from_rotvec(double vx, double vy, double vz) {
  const double a = sqrt(vx * vx + vy * vy + vz * vz);
  double scale;
  if (a <= 1e-3) {
    const double a2 = a * a;
    scale = 0.5 - a2 / 48.0 + a2 * a2 / 3840.0;
  } else {
    scale = sin(a / 2.0) / a;
  }
  return q4{scale * vx, scale * vy, scale * vz, cos(a / 2.0)};
}

__device__ __forceinline__ q4 sx_normalize(q4 q) {
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return q4{q.x / n, q.y / n, q.z / n, q.w / n};
}

__device__ __forceinline__ q4 sx_compose(q4 p, q4 q) {
  const double cx = p.y * q.z - p.z * q.y, cy = p.z * q.x - p.x * q.z, cz = p.x * q.y - p.y * q.x;
  q4 r;
  r.x = p.w * q.x + q.w * p.x + cx;
  r.y = p.w * q.y + q.w * p.y + cy;
  r.z = p.w * q.z + q.w * p.z + cz;
  r.w = p.w * q.w - p.x * q.x - p.y * q.y - p.z * q.z;
  return sx_normalize(r);
}

// slerp() of utils/smpl.py:76-107 followed by .as_rotvec() and from_rotvec() again (:140, :181-186)
__device__ __forceinline__ q4 sx_slerp_local(q4 q1, q4 q2, double t) {
  q1 = sx_normalize(q1);
  q2 = sx_normalize(q2);
  double dot = q1.x * q2.x + q1.y * q2.y + q1.z * q2.z + q1.w * q2.w;
  if (dot < 0.0) { q2 = q4{-q2.x, -q2.y, -q2.z, -q2.w}; dot = -dot; }
  q4 q;
  if (dot > 0.9995) {
    q = q4{q1.x + t * (q2.x - q1.x), q1.y + t * (q2.y - q1.y), q1.z + t * (q2.z - q1.z), q1.w + t * (q2.w - q1.w)};
  } else {
    const double th0 = acos(dot), th = th0 * t;
    const double st = sin(th), st0 = sin(th0);
    const double s0 = cos(th) - dot * st / st0, s1 = st / st0;
    q = q4{s0 * q1.x + s1 * q2.x, s0 * q1.y + s1 * q2.y, s0 * q1.z + s1 * q2.z, s0 * q1.w + s1 * q2.w};
  }
  q = sx_normalize(q);
  // as_rotvec
  if (q.w < 0) q = q4{-q.x, -q.y, -q.z, -q.w};
  const double a = 2.0 * atan2(sqrt(q.x * q.x + q.y * q.y + q.z * q.z), q.w);
  double scale;
  if (a <= 1e-3) {
    const double a2 = a * a;
    scale = 2.0 + a2 / 12.0 + 7.0 * a2 * a2 / 2880.0;
  } else {
    scale = a / sin(a / 2.0);
  }
  return sx_from_rotvec(scale * q.x, scale * q.y, scale * q.z);
}

#pragma clang fp contract(off)

// out[Nout][nrow][7]: pos xyz, quat wxyz.  ALIGN: target_time[Nout] (np.linspace(0, N-1, Nout)); else Nout == N.
// COMPACT: the inputs hold only what the kernel reads, FRAME-MINOR -- full_pose[P.n][3][N] in the order of the program's
// steps (the ancestor closure of the selection), joints[P.nrow][3][N] in output-row order.  A load instruction of the walk
// (one component of one joint, 64 consecutive output frames) then reads one short contiguous run of source frames, and
// the two frames of an interpolation pair sit next to each other in it: every fetched line is consumed by the loads that
// brought it in.  (Frame-major rows -- a lane's 240 B of poses touched joint by joint over the 200 us its libm chains
// take, 14 000 lanes per XCD -- kept 14 MB of lines open against 4 MB of L2 and fetched them two to three times:
// 6.3 x the algorithmic traffic in round 2, unchanged by compact frame-major rows in round 3.)
template <bool ALIGN, bool COMPACT>
__global__ __launch_bounds__(SX_BLOCK) void smplx_align_kernel(SmplxProg P, int N, int jstride,
                                                               const float* __restrict__ full_pose,
                                                               const float* __restrict__ joints, int Nout,
                                                               const double* __restrict__ target_time,
                                                               double* __restrict__ out) {
  extern __shared__ __align__(16) double stack[];   // [max_depth][4][SX_BLOCK]
  const int lane = threadIdx.x;
  const int o = blockIdx.x * SX_BLOCK + lane;
  if (o >= Nout) return;                             // no barriers in this kernel: columns are lane-private
  double t = 0.0, alpha = 0.0;
  int idx1 = o, idx2 = o, lo = o, hi = o;
  if (ALIGN) {
    t = target_time[o];
    idx1 = (int)floor(t);
    idx1 = min(max(idx1, 0), N - 1);
    idx2 = min(idx1 + 1, N - 1);
    alpha = t - (double)idx1;
    int ss = (int)ceil(t);
    ss = min(max(ss, 1), N - 1);
    lo = ss - 1; hi = ss;
  }
  // COMPACT with jstride < 0: the planes are those of ALL joints, indexed by joint (what the joints kernel reads and writes:
  // gmr_smplx_frames runs both steps without a gather in between)
  const bool byjoint = COMPACT && jstride < 0;
  // element (joint j, component c) of source frame f: frame-major rows, or frame-minor planes (COMPACT)
  const size_t pstep = COMPACT ? (size_t)N : 1, pj3 = COMPACT ? 3 * (size_t)N : 3;
  const float* p1 = full_pose + (COMPACT ? (size_t)idx1 : (size_t)idx1 * P.J * 3);
  const float* p2 = full_pose + (COMPACT ? (size_t)idx2 : (size_t)idx2 * P.J * 3);
  const float* jl = joints + (COMPACT ? (size_t)lo : (size_t)lo * jstride * 3);
  const float* jh = joints + (COMPACT ? (size_t)hi : (size_t)hi * jstride * 3);
  double* orow = out + (size_t)o * P.nrow * 7;
  for (int k = 0; k < P.n; k++) {
    const int j = (COMPACT && !byjoint) ? k : P.joint[k], d = P.depth[k];
    q4 ql;
    const float* a1 = p1 + (size_t)j * pj3;
    if (ALIGN) {
      const float* a2 = p2 + (size_t)j * pj3;
      q4 qa = sx_from_rotvec((double)a1[0], (double)a1[pstep], (double)a1[2 * pstep]);
      q4 qb = sx_from_rotvec((double)a2[0], (double)a2[pstep], (double)a2[2 * pstep]);
      ql = sx_slerp_local(qa, qb, alpha);
    } else {
      ql = sx_from_rotvec((double)a1[0], (double)a1[pstep], (double)a1[2 * pstep]);
    }
    q4 qg = ql;
    if (d > 0) {
      const double* s = stack + (size_t)(d - 1) * 4 * SX_BLOCK + lane;
      qg = sx_compose(q4{s[0], s[SX_BLOCK], s[2 * SX_BLOCK], s[3 * SX_BLOCK]}, ql);
    }
    {
      double* s = stack + (size_t)d * 4 * SX_BLOCK + lane;
      s[0] = qg.x; s[SX_BLOCK] = qg.y; s[2 * SX_BLOCK] = qg.z; s[3 * SX_BLOCK] = qg.w;
    }
    const int r = P.row[k];
    if (r >= 0) {
      double* w = orow + r * 7;
      const int jj = (COMPACT && !byjoint) ? r : j;                   // (compact joints: one row per output row)
#pragma unroll
      for (int c = 0; c < 3; c++) {
        if (ALIGN) {
          const float ylo = jl[(size_t)jj * pj3 + c * pstep], yhi = jh[(size_t)jj * pj3 + c * pstep];
          const float df = yhi - ylo;                                  // float32 difference (interp1d on a float32 y)
          const double slope = (double)df / (double)(hi - lo);
          w[c] = slope * (t - (double)lo) + (double)ylo;
        } else {
          w[c] = (double)jl[(size_t)jj * pj3 + c * pstep];
        }
      }
      w[3] = qg.w; w[4] = qg.x; w[5] = qg.y; w[6] = qg.z;
    }
  }
}

// Frame-major rows f32[N][C] <-> frame-minor planes f32[C][N] (C = J * 3 <= 256), 64 frames per block through one LDS tile.
// The joints kernel reads its poses and writes its joints as planes -- lane = frame, so every load / store instruction is
// one contiguous run; with frame-major rows a lane's 660-byte row stays open for the 200 us of its walk, and at eight or
// more blocks per CU the rows in flight (100 MB) no longer fit the L2: 2.2 GB fetched and 3.0 GB written for 0.69 + 0.69 GB
// (measured, profiles/r03_v6_*).  A block's 64 rows are ONE contiguous range of the row array (read / written flat, 16 bytes
// per lane), its piece of a plane 256 bytes.  (First version: generic 64 x 64 tiles, whose row side moved 256-byte pieces of
// 660-byte rows: 0.42 ms per 2^20 x 165 floats, 3.3 TB/s.)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <bool TO_PLANES>
__global__ __launch_bounds__(256) void rows_planes_kernel(const float* __restrict__ src, int N, int C, float* __restrict__ dst) {
  extern __shared__ float tile[];                      // [64][C + 1]
  const int r0 = blockIdx.x * 64, nr = min(64, N - r0), ld = C + 1, n = nr * C;
  const unsigned magic = (1u << 24) / (unsigned)C + 1u; // e / C for e < 2^14, C <= 256
  const int q = (threadIdx.x & 15) * 4, g = threadIdx.x >> 4;
  const float* rows_in = src + (size_t)r0 * C;          // TO_PLANES: flat rows in
  float* rows_out = dst + (size_t)r0 * C;               // else: flat rows out
  if (TO_PLANES) {
    for (int i = threadIdx.x * 4; i < n; i += 1024) {
      float v[4];
      if (i + 3 < n) { const f4u t = *reinterpret_cast<const f4u*>(rows_in + i); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
      else for (int k = 0; k < 4; k++) v[k] = i + k < n ? rows_in[i + k] : 0.0f;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const unsigned e = (unsigned)(i + k), r = (e * magic) >> 24;
        if ((int)e < n) tile[r * ld + (e - r * C)] = v[k];
      }
    }
  } else {
    for (int c = g; c < C; c += 16) {
      const float* p = src + (size_t)c * N + r0 + q;
      if (q + 3 < nr) { const f4u t = *reinterpret_cast<const f4u*>(p); tile[q * ld + c] = t.x; tile[(q + 1) * ld + c] = t.y; tile[(q + 2) * ld + c] = t.z; tile[(q + 3) * ld + c] = t.w; }
      else for (int k = 0; k < 4 && q + k < nr; k++) tile[(q + k) * ld + c] = p[k];
    }
  }
  __syncthreads();
  if (TO_PLANES) {
    for (int c = g; c < C; c += 16) {
      float* p = dst + (size_t)c * N + r0 + q;
      if (q + 3 < nr) { f4u t; t.x = tile[q * ld + c]; t.y = tile[(q + 1) * ld + c]; t.z = tile[(q + 2) * ld + c]; t.w = tile[(q + 3) * ld + c]; *reinterpret_cast<f4u*>(p) = t; }
      else for (int k = 0; k < 4 && q + k < nr; k++) p[k] = tile[(q + k) * ld + c];
    }
  } else {
    for (int i = threadIdx.x * 4; i < n; i += 1024) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const unsigned e = (unsigned)(i + k), r = (e * magic) >> 24;
        v[k] = (int)e < n ? tile[r * ld + (e - r * C)] : 0.0f;
      }
      if (i + 3 < n) { f4u t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3]; *reinterpret_cast<f4u*>(rows_out + i) = t; }
      else for (int k = 0; k < 4 && i + k < n; k++) rows_out[i + k] = v[k];
    }
  }
}

// joints from rest joints f64[J][3], poses, translations f32[N][3].  PLANES: poses and joints are frame-minor planes
// f32[J * 3][N] (the entry point transposes on either side); else frame-major rows f32[N][J][3]
template <bool PLANES>
__global__ __launch_bounds__(SX_BLOCK) void smplx_joints_kernel(SmplxProg P, int N, const double* __restrict__ j_rest,
                                                                const float* __restrict__ full_pose,
                                                                const float* __restrict__ transl,
                                                                float* __restrict__ joints) {
  // Parked transforms [nslot][12][SX_BLOCK] (R row-major, p), one column per lane.  A depth-indexed stack needs 11 levels for
  // SMPL-X -- 66 KB, two single-wavefront blocks per CU for a kernel whose time is the latency of its libm chains --; in DFS
  // order a joint's first child follows it immediately, so only joints with several children are parked: 3 slots, 18 KB.
  extern __shared__ __align__(16) double stack[];
  const int lane = threadIdx.x;
  const int n = blockIdx.x * SX_BLOCK + lane;
  if (n >= N) return;
  // element c of joint j of this lane's frame
  const float* pr = full_pose + (PLANES ? (size_t)n : (size_t)n * P.J * 3);
  float* jo = joints + (PLANES ? (size_t)n : (size_t)n * P.J * 3);
  const size_t es = PLANES ? (size_t)N : 1;
  const double tx = transl[(size_t)n * 3], ty = transl[(size_t)n * 3 + 1], tz = transl[(size_t)n * 3 + 2];
  double Rc[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pc[3] = {0, 0, 0};        // the transform of the previous step
  // the innermost parking slot (SMPL-X: a wrist while its fingers are walked, the head for jaw and eyes) is 24 registers
  // instead of 6 KB of LDS: 2 slots = 12 KB, twelve blocks per CU
  const int rslot = P.nslot - 1;
  double Rs[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, ps[3] = {0, 0, 0};
  // the pose of step k + 1 and its rest offset are requested while step k computes (a lane's pose row shares no line with its
  // neighbours': every read is a trip to L2 that nothing else would hide at two wavefronts per SIMD)
  float nvx = pr[(3 * P.joint[0]) * es], nvy = pr[(3 * P.joint[0] + 1) * es], nvz = pr[(3 * P.joint[0] + 2) * es];
  double nrel[3] = {j_rest[3 * P.joint[0]], j_rest[3 * P.joint[0] + 1], j_rest[3 * P.joint[0] + 2]};
  for (int k = 0; k < P.n; k++) {
    const int j = P.joint[k], d = P.depth[k];
    const double vx = nvx, vy = nvy, vz = nvz;
    const double rel[3] = {nrel[0], nrel[1], nrel[2]};      // step 0: the root's rest position
    if (k + 1 < P.n) {
      const int jn = P.joint[k + 1], pn = P.parent[k + 1];
      nvx = pr[(3 * jn) * es]; nvy = pr[(3 * jn + 1) * es]; nvz = pr[(3 * jn + 2) * es];
      nrel[0] = j_rest[3 * jn] - j_rest[3 * pn]; nrel[1] = j_rest[3 * jn + 1] - j_rest[3 * pn + 1]; nrel[2] = j_rest[3 * jn + 2] - j_rest[3 * pn + 2];
    }
    const double ax = vx + 1e-8, ay = vy + 1e-8, az = vz + 1e-8;
    const double ang = sqrt(ax * ax + ay * ay + az * az);
    const double x = vx / ang, y = vy / ang, z = vz / ang, s = sin(ang), c1 = 1.0 - cos(ang);
    const double K[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    double Rl[9];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) {
        double acc = 0;
#pragma unroll
        for (int m = 0; m < 3; m++) acc += K[a * 3 + m] * K[m * 3 + b];
        Rl[a * 3 + b] = (a == b ? 1.0 : 0.0) + s * K[a * 3 + b] + c1 * acc;
      }
    double Rg[9], pg[3];
    if (d == 0) {
#pragma unroll
      for (int i = 0; i < 9; i++) Rg[i] = Rl[i];
      pg[0] = rel[0]; pg[1] = rel[1]; pg[2] = rel[2];
    } else {
      double Rp[9], pp[3];
      const int ld = P.load[k];                       // (uniform)
      if (ld == rslot) {
#pragma unroll
        for (int i = 0; i < 9; i++) Rp[i] = Rs[i];
#pragma unroll
        for (int i = 0; i < 3; i++) pp[i] = ps[i];
      } else if (ld >= 0) {
        const double* sp = stack + (size_t)ld * 12 * SX_BLOCK + lane;
#pragma unroll
        for (int i = 0; i < 9; i++) Rp[i] = sp[i * SX_BLOCK];
#pragma unroll
        for (int i = 0; i < 3; i++) pp[i] = sp[(9 + i) * SX_BLOCK];
      } else {
#pragma unroll
        for (int i = 0; i < 9; i++) Rp[i] = Rc[i];
#pragma unroll
        for (int i = 0; i < 3; i++) pp[i] = pc[i];
      }
#pragma unroll
      for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int b = 0; b < 3; b++) {
          double acc = 0;
#pragma unroll
          for (int m = 0; m < 3; m++) acc += Rp[a * 3 + m] * Rl[m * 3 + b];
          Rg[a * 3 + b] = acc;
        }
        pg[a] = pp[a] + Rp[a * 3] * rel[0] + Rp[a * 3 + 1] * rel[1] + Rp[a * 3 + 2] * rel[2];
      }
    }
#pragma unroll
    for (int i = 0; i < 9; i++) Rc[i] = Rg[i];
#pragma unroll
    for (int i = 0; i < 3; i++) pc[i] = pg[i];
    const int sv = P.save[k];                         // (uniform)
    if (sv == rslot) {
#pragma unroll
      for (int i = 0; i < 9; i++) Rs[i] = Rg[i];
#pragma unroll
      for (int i = 0; i < 3; i++) ps[i] = pg[i];
    } else if (sv >= 0) {
      double* so = stack + (size_t)sv * 12 * SX_BLOCK + lane;
#pragma unroll
      for (int i = 0; i < 9; i++) so[i * SX_BLOCK] = Rg[i];
#pragma unroll
      for (int i = 0; i < 3; i++) so[(9 + i) * SX_BLOCK] = pg[i];
    }
    jo[(3 * j) * es] = (float)(pg[0] + tx);
    jo[(3 * j + 1) * es] = (float)(pg[1] + ty);
    jo[(3 * j + 2) * es] = (float)(pg[2] + tz);
  }
}

// DFS program over the joints in `keep` (all when empty); rows from `row_of` (-1 = not written)
static bool make_prog(int J, const int32_t* parents, const std::vector<char>& keep, const std::vector<int>& row_of,
                      int nrow, SmplxProg* P) {
  memset(P, 0, sizeof *P);
  P->J = J;
  P->nrow = nrow;
  std::vector<std::vector<int>> kids(J);
  int root = -1;
  for (int j = 0; j < J; j++) {
    if (parents[j] < 0) { if (root >= 0) return false; root = j; }
    else kids[parents[j]].push_back(j);
  }
  if (root < 0) return false;
  std::vector<std::pair<int, int>> st;     // (joint, depth)
  st.push_back({root, 0});
  int n = 0, md = 0;
  while (!st.empty()) {
    auto [j, d] = st.back();
    st.pop_back();
    if (!keep[j]) continue;
    P->joint[n] = (short)j; P->depth[n] = (short)d; P->row[n] = (short)row_of[j];
    P->parent[n] = (short)(parents[j] < 0 ? 0 : parents[j]);
    n++;
    md = std::max(md, d + 1);
    for (int c = (int)kids[j].size() - 1; c >= 0; c--) st.push_back({kids[j][c], d + 1});
  }
  P->n = n;
  P->max_depth = md;
  // parking slots of the joints kernel: slot of a joint = the number of parked ancestors above it
  std::vector<int> nkept(J, 0), parked_above(J, 0), slot_of(J, -1);
  for (int k = 0; k < n; k++) if (k > 0) nkept[P->parent[k]]++;
  int ns = 0;
  for (int k = 0; k < n; k++) {
    const int j = P->joint[k], p = P->parent[k];
    parked_above[j] = k == 0 ? 0 : parked_above[p] + (nkept[p] >= 2 ? 1 : 0);
    P->save[k] = -1;
    if (nkept[j] >= 2) { slot_of[j] = parked_above[j]; P->save[k] = (short)slot_of[j]; ns = std::max(ns, slot_of[j] + 1); }
    P->load[k] = (short)((k == 0 || P->joint[k - 1] == p) ? -1 : slot_of[p]);
    if (k > 0 && P->joint[k - 1] != p && slot_of[p] < 0) return false;      // (cannot happen in DFS preorder)
  }
  P->nslot = std::max(ns, 1);
  return true;
}

}  // namespace gmr

struct gmr_smplx {
  int J = 0, nsel = 0;
  gmr::SmplxProg all, sel;       // every joint (rows = joint index) / ancestor closure of the selection
  double* d_jrest = nullptr;     // staging of the host entry points
  char* ws = nullptr;            // their device workspace, grown on demand and kept (a hipMalloc / hipFree pair per clip costs
  size_t ws_bytes = 0;           //  more than the kernels of a short clip)
  std::mutex mu;                 // the host entry points of one handle run one at a time (they share d_jrest and ws)
  char* planes = nullptr;        // gmr_smplx_joints_dev: the transposed poses and joints of the launch in flight (one stream at
  size_t planes_bytes = 0;       //  a time per handle)
};

static hipError_t smplx_workspace(gmr_smplx* h, size_t bytes, char** out) {
  if (bytes > h->ws_bytes) {
    if (h->ws) (void)hipFree(h->ws);
    h->ws = nullptr; h->ws_bytes = 0;
    const size_t want = bytes + bytes / 4;
    hipError_t e = hipMalloc((void**)&h->ws, want);
    if (e != hipSuccess) return e;
    h->ws_bytes = want;
  }
  *out = h->ws;
  return hipSuccess;
}

extern "C" {

int gmr_smplx_create(int J, const int32_t* parents, int nsel, const int32_t* sel, gmr_smplx_t** out) {
  if (!parents || !out || J < 1 || J > SX_MAX_JOINTS) return gmr_fail(GMR_ERR_ARG, "gmr_smplx_create: 1 <= J <= %d", SX_MAX_JOINTS);
  if (nsel < 0 || nsel > J || (nsel > 0 && !sel)) return gmr_fail(GMR_ERR_ARG, "gmr_smplx_create: bad selection");
  for (int j = 0; j < J; j++)
    if (parents[j] >= j || (j > 0 && parents[j] < 0) || (j == 0 && parents[j] >= 0))
      return gmr_fail(GMR_ERR_ARG, "gmr_smplx_create: parents must precede children, joint 0 is the only root (joint %d)", j);
  gmr_smplx* h = new (std::nothrow) gmr_smplx;
  if (!h) return gmr_fail(GMR_ERR_ARG, "out of memory");
  h->J = J;
  h->nsel = nsel;
  std::vector<char> keep(J, 1);
  std::vector<int> row(J);
  for (int j = 0; j < J; j++) row[j] = j;
  bool ok = gmr::make_prog(J, parents, keep, row, J, &h->all);
  if (ok && nsel > 0) {
    std::fill(keep.begin(), keep.end(), 0);
    std::fill(row.begin(), row.end(), -1);
    for (int r = 0; r < nsel && ok; r++) {
      int j = sel[r];
      if (j < 0 || j >= J || row[j] >= 0) { ok = false; break; }
      row[j] = r;
      for (int a = j; a >= 0; a = parents[a]) keep[a] = 1;
    }
    ok = ok && gmr::make_prog(J, parents, keep, row, nsel, &h->sel);
  } else if (ok) {
    h->sel = h->all;
  }
  if (!ok) { delete h; return gmr_fail(GMR_ERR_ARG, "gmr_smplx_create: bad tree or duplicate / out-of-range selection"); }
  const int lds_align = h->all.max_depth * 4 * SX_BLOCK * 8, lds_joints = std::max(h->all.nslot - 1, 1) * 12 * SX_BLOCK * 8;
  if (lds_joints > 160 * 1024 - 1024) { delete h; return gmr_fail(GMR_ERR_ARG, "gmr_smplx_create: tree too deep (%d levels)", h->all.max_depth); }
  hipError_t e = hipSuccess;
  if (lds_align > 48 * 1024) {
    e = hipFuncSetAttribute((const void*)gmr::smplx_align_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_align);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gmr::smplx_align_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_align);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gmr::smplx_align_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_align);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gmr::smplx_align_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_align);
  }
  if (e == hipSuccess && lds_joints > 48 * 1024)
    {
    e = hipFuncSetAttribute((const void*)gmr::smplx_joints_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_joints);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gmr::smplx_joints_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_joints);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_jrest, (size_t)J * 3 * sizeof(double));
  if (e != hipSuccess) { delete h; return gmr_fail(GMR_ERR_HIP, "gmr_smplx_create: %s", hipGetErrorString(e)); }
  *out = h;
  return GMR_OK;
}

int gmr_smplx_destroy(gmr_smplx_t* h) {
  if (!h) return GMR_OK;
  (void)hipFree(h->d_jrest);
  if (h->ws) (void)hipFree(h->ws);
  if (h->planes) (void)hipFree(h->planes);
  delete h;
  return GMR_OK;
}

int gmr_smplx_rows(const gmr_smplx_t* h) { return h ? h->sel.nrow : 0; }

int gmr_smplx_joints_dev(gmr_smplx_t* h, int N, const double* d_j_rest, const float* d_full_pose, const float* d_transl,
                         float* d_joints, void* stream) {
  if (!h || N < 0 || !d_j_rest || !d_full_pose || !d_transl || !d_joints) return gmr_fail(GMR_ERR_ARG, "gmr_smplx_joints_dev: bad argument");
  if (N == 0) return GMR_OK;
  const int lds = std::max(h->all.nslot - 1, 1) * 12 * SX_BLOCK * 8;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((N + SX_BLOCK - 1) / SX_BLOCK), block(SX_BLOCK);
  static const bool rows = getenv("GMR_SMPLX_ROWS") != nullptr;       // A/B: the walk on frame-major rows
  if (rows) {
    hipLaunchKernelGGL(gmr::smplx_joints_kernel<false>, grid, block, lds, st, h->all, N, d_j_rest, d_full_pose, d_transl, d_joints);
  } else {
    const int C = h->J * 3;
    const size_t plane = ((size_t)N * C * sizeof(float) + 255) / 256 * 256;
    if (2 * plane > h->planes_bytes) {
      hipError_t e = hipStreamSynchronize(st);                        // (the launch in flight may still use the old block)
      if (e != hipSuccess) return gmr_fail(GMR_ERR_HIP, "gmr_smplx_joints_dev: %s", hipGetErrorString(e));
      if (h->planes) (void)hipFree(h->planes);
      h->planes = nullptr; h->planes_bytes = 0;
      if ((e = hipMalloc((void**)&h->planes, 2 * plane + plane / 2)) != hipSuccess) return gmr_fail(GMR_ERR_HIP, "gmr_smplx_joints_dev: %s", hipGetErrorString(e));
      h->planes_bytes = 2 * plane + plane / 2;
    }
    float* pose_t = reinterpret_cast<float*>(h->planes);
    float* joints_t = reinterpret_cast<float*>(h->planes + h->planes_bytes / 2 / 256 * 256);
    hipLaunchKernelGGL(gmr::rows_planes_kernel<true>, dim3((N + 63) / 64), dim3(256), 64 * (C + 1) * sizeof(float), st, d_full_pose, N, C, pose_t);
    hipLaunchKernelGGL(gmr::smplx_joints_kernel<true>, grid, block, lds, st, h->all, N, d_j_rest, pose_t, d_transl, joints_t);
    hipLaunchKernelGGL(gmr::rows_planes_kernel<false>, dim3((N + 63) / 64), dim3(256), 64 * (C + 1) * sizeof(float), st, joints_t, N, C, d_joints);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return gmr_fail(GMR_ERR_HIP, "smplx_joints_kernel: %s", hipGetErrorString(e));
  return GMR_OK;
}

static int smplx_align_launch(gmr_smplx_t* h, bool compact, int N, int jstride, const float* d_full_pose, const float* d_joints,
                              int Nout, const double* d_target_time, double* d_out, void* stream) {
  if (!h || N < 1 || Nout < 0 || (!compact && jstride < h->J) || !d_full_pose || !d_joints || !d_out)
    return gmr_fail(GMR_ERR_ARG, "gmr_smplx_align_dev: bad argument");
  if (!d_target_time && Nout != N) return gmr_fail(GMR_ERR_ARG, "gmr_smplx_align_dev: without target times Nout must equal N");
  if (d_target_time && N < 2) return gmr_fail(GMR_ERR_ARG, "gmr_smplx_align_dev: fps alignment needs at least two source frames");
  if (Nout == 0) return GMR_OK;
  const gmr::SmplxProg& P = h->sel;
  const int lds = P.max_depth * 4 * SX_BLOCK * 8;
  dim3 grid((Nout + SX_BLOCK - 1) / SX_BLOCK), block(SX_BLOCK);
#define GMR_SX_LAUNCH(A, C)                                                                                              \
  hipLaunchKernelGGL((gmr::smplx_align_kernel<A, C>), grid, block, lds, (hipStream_t)stream, P, N, jstride, d_full_pose, \
                     d_joints, Nout, d_target_time, d_out)
  if (d_target_time) { if (compact) GMR_SX_LAUNCH(true, true); else GMR_SX_LAUNCH(true, false); }
  else { if (compact) GMR_SX_LAUNCH(false, true); else GMR_SX_LAUNCH(false, false); }
#undef GMR_SX_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return gmr_fail(GMR_ERR_HIP, "smplx_align_kernel: %s", hipGetErrorString(e));
  return GMR_OK;
}

int gmr_smplx_align_dev(gmr_smplx_t* h, int N, int jstride, const float* d_full_pose, const float* d_joints, int Nout,
                        const double* d_target_time, double* d_out, void* stream) {
  return smplx_align_launch(h, false, N, jstride, d_full_pose, d_joints, Nout, d_target_time, d_out, stream);
}

// the joints the alignment reads, in the order the compact layout stores them: pose_joints[npose] = the ancestor closure of
// the selection in walk order, row_joints[nrow] = the joint of every output row
int gmr_smplx_compact_layout(const gmr_smplx_t* h, int32_t* pose_joints, int* npose, int32_t* row_joints, int* nrow) {
  if (!h) return gmr_fail(GMR_ERR_ARG, "null handle");
  const gmr::SmplxProg& P = h->sel;
  if (npose) *npose = P.n;
  if (nrow) *nrow = P.nrow;
  for (int k = 0; k < P.n; k++) {
    if (pose_joints) pose_joints[k] = P.joint[k];
    if (row_joints && P.row[k] >= 0) row_joints[P.row[k]] = P.joint[k];
  }
  return GMR_OK;
}

int gmr_smplx_align_compact_dev(gmr_smplx_t* h, int N, const float* d_pose_c, const float* d_joints_c, int Nout,
                                const double* d_target_time, double* d_out, void* stream) {
  return smplx_align_launch(h, true, N, 0, d_pose_c, d_joints_c, Nout, d_target_time, d_out, stream);
}

// host-buffer variants: copy, launch, synchronise
int gmr_smplx_joints(gmr_smplx_t* h, int N, const double* j_rest, const float* full_pose, const float* transl, float* joints) {
  if (!h || N < 0 || !j_rest || !full_pose || !transl || !joints) return gmr_fail(GMR_ERR_ARG, "gmr_smplx_joints: bad argument");
  if (N == 0) return GMR_OK;
  const size_t nb_pose = (size_t)N * h->J * 3 * sizeof(float), nb_tr = (size_t)N * 3 * sizeof(float);
  std::lock_guard<std::mutex> guard(h->mu);
  char* ws = nullptr;
  hipError_t e = smplx_workspace(h, 2 * nb_pose + nb_tr + 64, &ws);
  if (e != hipSuccess) return gmr_fail(GMR_ERR_HIP, "gmr_smplx_joints: %s", hipGetErrorString(e));
  float* d_pose = (float*)ws;
  float* d_j = (float*)(ws + nb_pose);
  float* d_tr = (float*)(ws + 2 * nb_pose);
  int rc = GMR_OK;
  if ((e = hipMemcpy(h->d_jrest, j_rest, (size_t)h->J * 3 * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(d_pose, full_pose, nb_pose, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(d_tr, transl, nb_tr, hipMemcpyHostToDevice)) != hipSuccess)
    rc = gmr_fail(GMR_ERR_HIP, "gmr_smplx_joints: %s", hipGetErrorString(e));
  if (rc == GMR_OK) rc = gmr_smplx_joints_dev(h, N, h->d_jrest, d_pose, d_tr, d_j, nullptr);
  if (rc == GMR_OK && (e = hipMemcpy(joints, d_j, nb_pose, hipMemcpyDeviceToHost)) != hipSuccess)
    rc = gmr_fail(GMR_ERR_HIP, "gmr_smplx_joints: %s", hipGetErrorString(e));
  return rc;
}

// Both steps for one clip, host buffers in and out, nothing but the packed frames coming back: poses up, transposed once,
// the body model's joints written as planes, the alignment reading pose and joint planes by joint index -- no joints on
// the host, no gather, no second upload (utils/smpl.py:12-41 + :109-197 in one call).  target_time == NULL: Nout == N.
int gmr_smplx_frames(gmr_smplx_t* h, int N, const double* j_rest, const float* full_pose, const float* transl, int Nout,
                     const double* target_time, double* out) {
  if (!h || N < 1 || Nout < 0 || !j_rest || !full_pose || !transl || !out) return gmr_fail(GMR_ERR_ARG, "gmr_smplx_frames: bad argument");
  if (!target_time && Nout != N) return gmr_fail(GMR_ERR_ARG, "gmr_smplx_frames: without target times Nout must equal N");
  if (target_time && N < 2) return gmr_fail(GMR_ERR_ARG, "gmr_smplx_frames: fps alignment needs at least two source frames");
  if (Nout == 0) return GMR_OK;
  const int C = h->J * 3;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t nb_pose = up((size_t)N * C * sizeof(float)), nb_tr = up((size_t)N * 3 * sizeof(float));
  const size_t nb_t = up((size_t)Nout * sizeof(double)), nb_out = up((size_t)Nout * h->sel.nrow * 7 * sizeof(double));
  std::lock_guard<std::mutex> guard(h->mu);
  char* ws = nullptr;
  hipError_t e = smplx_workspace(h, 3 * nb_pose + nb_tr + nb_t + nb_out, &ws);
  if (e != hipSuccess) return gmr_fail(GMR_ERR_HIP, "gmr_smplx_frames: %s", hipGetErrorString(e));
  float* d_pose = (float*)ws;
  float* pose_t = (float*)(ws + nb_pose);
  float* joints_t = (float*)(ws + 2 * nb_pose);
  float* d_tr = (float*)(ws + 3 * nb_pose);
  double* d_t = (double*)(ws + 3 * nb_pose + nb_tr);
  double* d_o = (double*)(ws + 3 * nb_pose + nb_tr + nb_t);
  if ((e = hipMemcpy(h->d_jrest, j_rest, (size_t)h->J * 3 * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(d_pose, full_pose, (size_t)N * C * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(d_tr, transl, (size_t)N * 3 * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess ||
      (target_time && (e = hipMemcpy(d_t, target_time, (size_t)Nout * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess))
    return gmr_fail(GMR_ERR_HIP, "gmr_smplx_frames: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(gmr::rows_planes_kernel<true>, dim3((N + 63) / 64), dim3(256), 64 * (C + 1) * sizeof(float), nullptr, d_pose, N, C, pose_t);
  hipLaunchKernelGGL(gmr::smplx_joints_kernel<true>, dim3((N + SX_BLOCK - 1) / SX_BLOCK), dim3(SX_BLOCK),
                     std::max(h->all.nslot - 1, 1) * 12 * SX_BLOCK * 8, nullptr, h->all, N, h->d_jrest, pose_t, d_tr, joints_t);
  if ((e = hipGetLastError()) != hipSuccess) return gmr_fail(GMR_ERR_HIP, "gmr_smplx_frames: %s", hipGetErrorString(e));
  int rc = smplx_align_launch(h, true, N, -1, pose_t, joints_t, Nout, target_time ? d_t : nullptr, d_o, nullptr);
  if (rc == GMR_OK && (e = hipMemcpy(out, d_o, (size_t)Nout * h->sel.nrow * 7 * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess)
    rc = gmr_fail(GMR_ERR_HIP, "gmr_smplx_frames: %s", hipGetErrorString(e));
  return rc;
}

int gmr_smplx_align(gmr_smplx_t* h, int N, int jstride, const float* full_pose, const float* joints, int Nout,
                    const double* target_time, double* out) {
  if (!h || N < 1 || Nout < 0 || jstride < (h ? h->J : 0) || !full_pose || !joints || !out)
    return gmr_fail(GMR_ERR_ARG, "gmr_smplx_align: bad argument");
  if (Nout == 0) return GMR_OK;
  // only what the walk reads crosses the bus and lies in device memory: the poses of the selection's ancestor closure, in walk
  // order, and the joints of the output rows (for the G1's 14 bodies: 20 + 14 of 55 + 55 joints)
  const gmr::SmplxProg& P = h->sel;
  const size_t np = (size_t)P.n, nr = (size_t)P.nrow;
  std::vector<float> pc((size_t)N * np * 3), jc((size_t)N * nr * 3);
  int rowj[SX_MAX_JOINTS];
  for (int k = 0; k < P.n; k++) if (P.row[k] >= 0) rowj[P.row[k]] = P.joint[k];
  for (int n = 0; n < N; n++) {                       // frame-minor planes [joint][component][frame]
    const float* ps = full_pose + (size_t)n * h->J * 3;
    const float* js = joints + (size_t)n * jstride * 3;
    for (size_t k = 0; k < np; k++) {
      const int j = P.joint[k];
      for (int c = 0; c < 3; c++) pc[(3 * k + c) * (size_t)N + n] = ps[3 * j + c];
    }
    for (size_t r = 0; r < nr; r++) {
      const int j = rowj[r];
      for (int c = 0; c < 3; c++) jc[(3 * r + c) * (size_t)N + n] = js[3 * j + c];
    }
  }
  const size_t nb_pose = pc.size() * sizeof(float), nb_j = jc.size() * sizeof(float);
  const size_t nb_t = (size_t)Nout * sizeof(double), nb_out = (size_t)Nout * h->sel.nrow * 7 * sizeof(double);
  auto up = [](size_t v) { return (v + 63) / 64 * 64; };
  std::lock_guard<std::mutex> guard(h->mu);
  char* ws = nullptr;
  hipError_t e = smplx_workspace(h, up(nb_pose) + up(nb_j) + up(nb_t) + up(nb_out), &ws);
  if (e != hipSuccess) return gmr_fail(GMR_ERR_HIP, "gmr_smplx_align: %s", hipGetErrorString(e));
  float* d_pose = (float*)ws;
  float* d_j = (float*)(ws + up(nb_pose));
  double* d_t = (double*)(ws + up(nb_pose) + up(nb_j));
  double* d_o = (double*)(ws + up(nb_pose) + up(nb_j) + up(nb_t));
  int rc = GMR_OK;
  if ((e = hipMemcpy(d_pose, pc.data(), nb_pose, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(d_j, jc.data(), nb_j, hipMemcpyHostToDevice)) != hipSuccess ||
      (target_time && (e = hipMemcpy(d_t, target_time, nb_t, hipMemcpyHostToDevice)) != hipSuccess))
    rc = gmr_fail(GMR_ERR_HIP, "gmr_smplx_align: %s", hipGetErrorString(e));
  if (rc == GMR_OK) rc = gmr_smplx_align_compact_dev(h, N, d_pose, d_j, Nout, target_time ? d_t : nullptr, d_o, nullptr);
  if (rc == GMR_OK && (e = hipMemcpy(out, d_o, nb_out, hipMemcpyDeviceToHost)) != hipSuccess)
    rc = gmr_fail(GMR_ERR_HIP, "gmr_smplx_align: %s", hipGetErrorString(e));
  return rc;
}

}  // extern "C"
