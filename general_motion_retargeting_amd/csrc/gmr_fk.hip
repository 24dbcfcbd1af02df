// gmr_fk.hip -- post-hoc batched forward kinematics, float32 (rows H8-H9 of SURVEY.md section 8a).
//
// Replaces KinematicsModel.forward_kinematics (reference kinematics_model.py:213-246), which on the
// reference's "cuda:0" path is ~70 tiny ATen launches per body (~2.6k per call), by one kernel.
// This one IS bandwidth-shaped: 4*ndof B in, 12*nbody (+16*nbody) B out per frame.
//
// Lane mapping: thread = (frame, body), bodies of a frame on consecutive lanes, frames of a block
// consecutive, so body_pos[f][b][0..2] / body_rot[f][b][0..3] stores are perfectly coalesced
// (consecutive lanes write consecutive 12 B / 16 B chunks).  Each lane first forms its own body's
// joint-composed local rotation (local_rot * axis_angle(dof)) into LDS, then walks root -> body
// reading its ancestors' local rotations from LDS.  The walk repeats, per lane, exactly the
// operation sequence of the reference's serial loop (pos_j = pos_p + rot_p * t_j;
// rot_j = rot_p * (r_j * jr_j)), so the redundancy changes no rounding.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/gmr_hip.h"
#include "gmr_fk_tree.h"

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

__global__ __launch_bounds__(256) void fk_batch_kernel(const FkTree* __restrict__ tree, int B,
                                                       const float* __restrict__ root_pos,
                                                       const float* __restrict__ root_rot,
                                                       const float* __restrict__ dof,
                                                       float* __restrict__ body_pos, float* __restrict__ body_rot,
                                                       float* __restrict__ min_part) {
  extern __shared__ __align__(16) float fsm[];
  const int nb = tree->nbody, ndof = tree->ndof, maxd = tree->maxd;
  const int fpb = 256 / nb;
  float* cr = fsm;                        // [fpb][nb][4]
  float* lt = cr + fpb * nb * 4;          // [nb][3]
  short* chain = reinterpret_cast<short*>(lt + nb * 3);  // [nb][maxd]
  short* depth = chain + nb * maxd;       // [nb]
  __shared__ float red[4];
  const int tid = threadIdx.x;
  for (int i = tid; i < nb * 3; i += 256) lt[i] = tree->local_t[i];
  for (int i = tid; i < nb * maxd; i += 256) chain[i] = tree->chain[i];
  for (int i = tid; i < nb; i += 256) depth[i] = tree->depth[i];
  const int fl = tid / nb, b = tid - fl * nb;
  const long long f = (long long)blockIdx.x * fpb + fl;
  const bool on = fl < fpb && f < B;
  if (on && b > 0) {
    // dof_to_rot: sin/cos of the float32 half angle; products and normalisation in float64;
    // rounded to float32 on assignment (kinematics_model.py:21-36, torch_utils.py:353-359)
    f4 jr = {0.f, 0.f, 0.f, 1.f};
    int di = tree->dof_idx[b];
    if (di >= 0) {
      float th = dof[f * ndof + di] / 2.0f;
      double s = (double)sinf(th), c = (double)cosf(th);
      double ax = tree->axis[3 * b], ay = tree->axis[3 * b + 1], az = tree->axis[3 * b + 2];
      double an = fmax(sqrt(ax * ax + ay * ay + az * az), 1e-9);
      double qx = ax / an * s, qy = ay / an * s, qz = az / an * s, qw = c;
      double qn = fmax(sqrt(qx * qx + qy * qy + qz * qz + qw * qw), 1e-9);
      jr = f4{(float)(qx / qn), (float)(qy / qn), (float)(qz / qn), (float)(qw / qn)};
    }
    f4 lr = {tree->local_r[4 * b], tree->local_r[4 * b + 1], tree->local_r[4 * b + 2], tree->local_r[4 * b + 3]};
    f4 c = qmul_xyzw(lr, jr);
    float* o = cr + (fl * nb + b) * 4;
    o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w;
  }
  __syncthreads();
  float z = INFINITY;
  if (on) {
    float px = root_pos[f * 3], py = root_pos[f * 3 + 1], pz = root_pos[f * 3 + 2];
    f4 rot = {root_rot[f * 4], root_rot[f * 4 + 1], root_rot[f * 4 + 2], root_rot[f * 4 + 3]};
    const int dep = depth[b];
    const short* ch = chain + b * maxd;
    for (int d = 1; d <= dep; d++) {
      int c = ch[d];
      float wx, wy, wz;
      qrot_xyzw(rot, lt[3 * c], lt[3 * c + 1], lt[3 * c + 2], wx, wy, wz);
      px = px + wx; py = py + wy; pz = pz + wz;
      const float* cc = cr + (fl * nb + c) * 4;
      rot = qmul_xyzw(rot, f4{cc[0], cc[1], cc[2], cc[3]});
    }
    float* op = body_pos + (f * nb + b) * 3;
    op[0] = px; op[1] = py; op[2] = pz;
    if (body_rot) {
      float4* orr = reinterpret_cast<float4*>(body_rot + (f * nb + b) * 4);
      *orr = make_float4(rot.x, rot.y, rot.z, rot.w);
    }
    z = pz;
  }
  if (min_part) {
    for (int off = 32; off > 0; off >>= 1) z = fminf(z, __shfl_xor(z, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = z;
    __syncthreads();
    if (tid == 0) min_part[blockIdx.x] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
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

}  // namespace gmr

extern "C" int gmr_fk_blocks(int nbody, int B) {
  int fpb = 256 / nbody;
  return (B + fpb - 1) / fpb;
}

extern "C" hipError_t gmr_launch_fk_batch(const gmr::FkTree* d_tree, int nbody, int maxd, int B,
                                          const float* d_root_pos, const float* d_root_rot, const float* d_dof,
                                          float* d_body_pos, float* d_body_rot, float* d_min_part, float* d_min_z,
                                          hipStream_t stream) {
  if (B <= 0) return hipSuccess;
  int fpb = 256 / nbody;
  int blocks = (B + fpb - 1) / fpb;
  size_t smem = (size_t)fpb * nbody * 4 * sizeof(float) + (size_t)nbody * 3 * sizeof(float) +
                (size_t)nbody * maxd * sizeof(short) + (size_t)nbody * sizeof(short);
  smem = (smem + 15) / 16 * 16;
  hipLaunchKernelGGL(gmr::fk_batch_kernel, dim3(blocks), dim3(256), smem, stream, d_tree, B, d_root_pos,
                     d_root_rot, d_dof, d_body_pos, d_body_rot, d_min_z ? d_min_part : nullptr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (d_min_z) {
    hipLaunchKernelGGL(gmr::min_reduce_kernel, dim3(1), dim3(256), 0, stream, d_min_part, blocks, d_min_z);
    e = hipGetLastError();
  }
  return e;
}
