// gmr_fk_tree.h -- device-side tree of the float32 post-hoc FK (reference KinematicsModel arrays).
#pragma once
#include <stdint.h>

namespace gmr {

constexpr int FK_MAX_BODIES = 64;
constexpr int FK_MAX_DEPTH = 24;

// Everything the walk needs about one body, as ONE 64-byte record: a single s_load_dwordx16 per body, issued one
// body ahead (the per-field arrays cost four dependent scalar / vector round trips per body).
struct FkBodyRec {
  float t[3];                 // local translation
  uint32_t meta;              // [7:0] has a hinge, [15:8] load slot + 1 (0: parent = previous body), [23:16] save slot + 1, [31:24] parent
  float r[4];                 // local rotation xyzw, un-normalised
  double axis[3];             // normalised hinge axis (float64)
  int32_t dof_idx;            // first dof of the joint or -1
  uint32_t pad;
};
static_assert(sizeof(FkBodyRec) == 64, "one record = one s_load_dwordx16");

struct FkTree {
  int nbody, ndof, maxd, nslot;
  int32_t dof_idx[FK_MAX_BODIES];              // first dof of the body's joint or -1
  short dof_body[FK_MAX_BODIES];               // body whose joint dof d drives
  short depth[FK_MAX_BODIES];
  short load_slot[FK_MAX_BODIES];              // LDS slot holding the parent transform, or -1: parent == previous body
  short save_slot[FK_MAX_BODIES];              // LDS slot this body's transform is parked in (>= 2 children), or -1
  short parent[FK_MAX_BODIES];                 // parent body (0 for body 0): staged outputs are re-read as parent transforms
  short chain[FK_MAX_BODIES * FK_MAX_DEPTH];   // [nbody][maxd] packed with stride maxd
  float local_t[FK_MAX_BODIES * 3];
  float local_r[FK_MAX_BODIES * 4];            // xyzw, un-normalised (kinematics_model.py:119-123)
  double axis[FK_MAX_BODIES * 3];              // float64 hinge axis (kinematics_model.py:133-134)
  FkBodyRec rec[FK_MAX_BODIES];                // the same data, one record per body
};

}  // namespace gmr
