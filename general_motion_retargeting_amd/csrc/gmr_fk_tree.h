// gmr_fk_tree.h -- device-side tree of the float32 post-hoc FK (reference KinematicsModel arrays).
#pragma once
#include <stdint.h>

namespace gmr {

constexpr int FK_MAX_BODIES = 64;
constexpr int FK_MAX_DEPTH = 24;

struct FkTree {
  int nbody, ndof, maxd, nslot;
  int32_t dof_idx[FK_MAX_BODIES];              // first dof of the body's joint or -1
  short depth[FK_MAX_BODIES];
  short load_slot[FK_MAX_BODIES];              // LDS slot holding the parent transform, or -1: parent == previous body
  short save_slot[FK_MAX_BODIES];              // LDS slot this body's transform is parked in (>= 2 children), or -1
  short parent[FK_MAX_BODIES];                 // parent body (0 for body 0): staged outputs are re-read as parent transforms
  short chain[FK_MAX_BODIES * FK_MAX_DEPTH];   // [nbody][maxd] packed with stride maxd
  float local_t[FK_MAX_BODIES * 3];
  float local_r[FK_MAX_BODIES * 4];            // xyzw, un-normalised (kinematics_model.py:119-123)
  double axis[FK_MAX_BODIES * 3];              // float64 hinge axis (kinematics_model.py:133-134)
};

}  // namespace gmr
