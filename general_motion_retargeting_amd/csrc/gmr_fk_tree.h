// gmr_fk_tree.h -- device-side tree of the float32 post-hoc FK (reference KinematicsModel arrays).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace gmr {

constexpr int FK_MAX_BODIES = 64;
constexpr int FK_MAX_DEPTH = 24;
constexpr int FK_MAX_WAVES = 4;

// Everything the walk needs about one body, as ONE 64-byte record: a single s_load_dwordx16 per body, issued one
// body ahead (the per-field arrays cost four dependent scalar / vector round trips per body).
struct FkBodyRec {
  float t[3];                 // local translation
  uint32_t meta;              // [0] has a hinge, [15:8] load slot + 1 (0: parent = previous body), [23:16] save slot + 1, [31:24] parent
                              // [1] r is exactly (0, 0, 0, 1), [3:2] = k + 1: the hinge axis is exactly +-e_k (0: any axis)
                              // split walk: [4] this wavefront stores the body, [5] workgroup barrier BEFORE this body (shared
                              // trunk), [7:6] also save the transform in block-wide LDS slot ([7:6] - 1) for other wavefronts
  float r[4];                 // local rotation xyzw, un-normalised
  double axis[3];             // normalised hinge axis (float64); meta[3:2] != 0: axis[0] = the +-1.0 of e_k
  int32_t dof_idx;            // first dof of the joint or -1
  uint32_t next_park;         // split walk: where the NEXT body of the wavefront's list finds its joint angle (read one body
                              // ahead) -- [31] that body has a joint, [30] extra column (index in [15:0]) instead of a float
                              // offset in the lane's staging row ([15:0]); every walk: [18:16] which components of THIS
                              // body's t are exactly zero
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
  // The split walk (fk_split_kernel): up to FK_MAX_WAVES wavefronts per block of 64 frames, each walking a part of the
  // tree -- the chain of ancestors its subtrees hang from (recomputed by every wavefront that needs it) and then its
  // own subtrees.  wrec[wave_start[w] .. wave_start[w + 1]) is wavefront w's list in body order; in these records
  // meta[4] = this wavefront stores the body's outputs, meta[15:8] / [23:16] = load / save slot + 1 in the block-wide
  // slot numbering (slots hold position and rotation), meta[31:24] = the body.
  int nwave, nslot_split;
  int wave_start[5];
  FkBodyRec wrec[2 * FK_MAX_BODIES];
  // the joint angles a wavefront needs, read from the frame's dof row in ONE batch before the walk and parked in LDS:
  // [15:0] dof, [30:16] float offset of the parking slot in the lane's staging row, or -- [31] set -- the index of an
  // extra column [index][lane] behind the staging area (nextra columns per block); 0xffffffff ends the list
  uint32_t wave_park[FK_MAX_WAVES][32];
  int nextra;
  // shared trunk (fk_build_split): the wavefront executes the block's one mid-walk barrier after its list instead of
  // before one of its bodies (record flag, meta bit 5)
  int wave_tail_barrier[FK_MAX_WAVES];
};

// Partition of the tree for the split walk (host side, gmr_fk_create).  `parent[b] < b`.  Returns the number of
// wavefronts (1: not worth splitting) and fills, per wavefront, the ascending list of bodies it walks.
inline int fk_split_tree(int nbody, const int* parent, int maxw, int lists[][FK_MAX_BODIES], int* nlist) {
  int size[FK_MAX_BODIES], nchild[FK_MAX_BODIES] = {0};
  for (int b = 0; b < nbody; b++) size[b] = 1;
  for (int b = nbody - 1; b >= 1; b--) { size[parent[b]] += size[b]; nchild[parent[b]]++; }
  bool trunk[FK_MAX_BODIES] = {false}, part[FK_MAX_BODIES] = {false};
  trunk[0] = true;
  for (int b = 1; b < nbody; b++) if (parent[b] == 0) part[b] = true;
  int bin_of[FK_MAX_BODIES];
  auto evaluate = [&](const bool* tr, const bool* pt, int* bins) {
    // longest-processing-time assignment of the parts; a bin also walks the trunk ancestors of its parts
    int order[FK_MAX_BODIES], n = 0;
    for (int b = 0; b < nbody; b++) if (pt[b]) order[n++] = b;
    for (int i = 1; i < n; i++) for (int j = i; j > 0 && size[order[j]] > size[order[j - 1]]; j--) { int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    int cost[FK_MAX_WAVES] = {0};
    bool anc[FK_MAX_WAVES][FK_MAX_BODIES] = {};
    for (int i = 0; i < n; i++) {
      int best = 0, bestc = 1 << 30;
      for (int w = 0; w < maxw; w++) {
        int extra = 0;
        for (int a = parent[order[i]]; a >= 0; a = a == 0 ? -1 : parent[a]) if (!anc[w][a]) extra++;
        if (cost[w] + extra + size[order[i]] < bestc) { bestc = cost[w] + extra + size[order[i]]; best = w; }
      }
      for (int a = parent[order[i]]; a >= 0; a = a == 0 ? -1 : parent[a]) anc[best][a] = true;
      cost[best] = bestc;
      bins[order[i]] = best;
    }
    (void)tr;
    int mx = 0;
    for (int w = 0; w < maxw; w++) mx = cost[w] > mx ? cost[w] : mx;
    return mx;
  };
  // open the largest part again and again (its top body joins the trunk, its children become parts) and keep the best
  // configuration seen: a chain like waist yaw -> roll -> torso improves nothing until the limbs below it separate
  int best = nbody > 1 ? evaluate(trunk, part, bin_of) : 1;
  {
    bool tr2[FK_MAX_BODIES], pt2[FK_MAX_BODIES];
    for (int b = 0; b < nbody; b++) { tr2[b] = trunk[b]; pt2[b] = part[b]; }
    for (int step = 0; step < nbody; step++) {
      int r = -1;
      for (int b = 1; b < nbody; b++) if (pt2[b] && nchild[b] > 0 && (r < 0 || size[b] > size[r])) r = b;
      if (r < 0) break;
      tr2[r] = true; pt2[r] = false;
      for (int b = r + 1; b < nbody; b++) if (parent[b] == r) pt2[b] = true;
      int bins2[FK_MAX_BODIES];
      const int c = evaluate(tr2, pt2, bins2);
      if (c < best) {
        best = c;
        for (int b = 0; b < nbody; b++) { trunk[b] = tr2[b]; part[b] = pt2[b]; bin_of[b] = bins2[b]; }
      }
    }
  }
  // a wavefront's list: ancestors of its parts and the parts' subtrees, ascending (parents come before children)
  int nw = 0;
  for (int w = 0; w < maxw; w++) {
    bool in[FK_MAX_BODIES] = {false};
    bool any = false;
    for (int b = 1; b < nbody; b++) {
      if (!(part[b] && bin_of[b] == w)) continue;
      any = true;
      for (int a = parent[b]; a >= 0; a = a == 0 ? -1 : parent[a]) in[a] = true;
      in[b] = true;
    }
    if (!any) continue;
    for (int b = 1; b < nbody; b++) if (!in[b] && !trunk[b] && in[parent[b]] && !part[b]) in[b] = true;   // descendants of a part
    nlist[nw] = 0;
    for (int b = 0; b < nbody; b++) if (in[b]) lists[nw][nlist[nw]++] = b;
    nw++;
  }
  if (nw == 0) { nlist[0] = 0; for (int b = 0; b < nbody; b++) lists[0][nlist[0]++] = b; nw = 1; }
  return nw;
}


// Records of the split walk (host side, gmr_fk_create): per-wavefront body lists with their own records, slots numbered
// block-wide, and the parking slots of the joint angles.  `t.rec`, `t.nbody` must be filled.  Returns nullptr or why
// the tree cannot be described (the caller fails); a tree that merely does not split gets nwave = 1.
inline const char* fk_build_split(FkTree& t, const int32_t* parent, int maxw) {
  const int nbody = t.nbody;
  int lists[FK_MAX_WAVES][FK_MAX_BODIES], nlist[FK_MAX_WAVES] = {0};
  int par[FK_MAX_BODIES];
  for (int b = 0; b < nbody; b++) par[b] = b == 0 ? -1 : parent[b];
  t.nwave = nbody >= 8 ? fk_split_tree(nbody, par, maxw, lists, nlist) : 1;
  if (t.nwave == 1) { nlist[0] = nbody; for (int b = 0; b < nbody; b++) lists[0][b] = b; }
  // Shared trunk.  fk_split_tree lets every wavefront recompute the chain of ancestors its subtrees hang from (for the G1:
  // the three waist bodies, walked by three wavefronts).  When all such shared bodies belong to ONE wavefront w0 and open
  // its list, they are walked by w0 alone: w0 parks the transforms other wavefronts need as parents in block-wide LDS slots,
  // the block meets at ONE barrier, and the others continue from those slots (38 + 2 instead of 47 body evaluations per
  // frame for the G1).  A wavefront without such a dependency places the barrier where w0 is expected to arrive.
  int shared_slot[FK_MAX_BODIES], barrier_at[FK_MAX_WAVES], extra_dst[FK_MAX_BODIES];
  for (int b = 0; b < nbody; b++) { shared_slot[b] = -1; extra_dst[b] = 0; }
  for (int w = 0; w < FK_MAX_WAVES; w++) { barrier_at[w] = -1; t.wave_tail_barrier[w] = 0; }
  int nshared = 0, share_owner = -1;
  if (t.nwave > 1 && !getenv("GMR_FK_NO_SHARE")) {
    int first[FK_MAX_BODIES], count[FK_MAX_BODIES];
    for (int b = 0; b < nbody; b++) { first[b] = -1; count[b] = 0; }
    for (int w = 0; w < t.nwave; w++) for (int i = 0; i < nlist[w]; i++) { const int b = lists[w][i]; if (first[b] < 0) first[b] = w; count[b]++; }
    int A[FK_MAX_BODIES], nA = 0, w0 = -1;
    bool ok = true;
    for (int b = 1; b < nbody; b++) if (count[b] > 1) { if (w0 < 0) w0 = first[b]; ok = ok && first[b] == w0; A[nA++] = b; }
    ok = ok && nA > 0 && w0 >= 0 && nlist[w0] > nA;
    for (int i = 0; ok && i < nA; i++) ok = lists[w0][1 + i] == A[i];            // the shared bodies open w0's list, in order
    bool inA[FK_MAX_BODIES] = {false};
    for (int i = 0; i < nA; i++) inA[A[i]] = true;
    // the transforms other wavefronts need as parents: at most 3 block-wide slots (two bits in the record)
    int need[FK_MAX_BODIES], nneed = 0;
    for (int w = 0; ok && w < t.nwave; w++) {
      if (w == w0) continue;
      for (int i = 0; i < nlist[w]; i++) {
        const int b = lists[w][i];
        if (b == 0 || inA[b]) continue;
        const int p = par[b];
        if (inA[p]) { bool seen = false; for (int k = 0; k < nneed; k++) seen = seen || need[k] == p; if (!seen) need[nneed++] = p; }
      }
    }
    ok = ok && nneed >= 1 && nneed <= 3;
    if (ok) {
      share_owner = w0;
      for (int k = 0; k < nneed; k++) { shared_slot[need[k]] = k; extra_dst[need[k]] = k + 1; }
      nshared = nneed;
      for (int w = 0; w < t.nwave; w++) {
        if (w == w0) { barrier_at[w] = 1 + nA; continue; }
        int m = 0;
        for (int i = 0; i < nlist[w]; i++) if (!inA[lists[w][i]]) lists[w][m++] = lists[w][i];      // drop the shared bodies
        nlist[w] = m;
        int at = -1;
        for (int i = 1; i < m && at < 0; i++) if (inA[par[lists[w][i]]]) at = i;                   // first body that hangs from the shared trunk
        // ... but not later than w0 is expected to arrive (everybody waits for the last one): the barrier may come
        // before the first dependent body, never after it
        const int when = 1 + nA < m ? 1 + nA : m;
        barrier_at[w] = (at >= 0 && at < when) ? at : when;
      }
    }
  }
  bool owned[FK_MAX_BODIES] = {false};
  int nrec = 0, nslot = nshared;
  for (int w = 0; w < t.nwave; w++) {
    t.wave_start[w] = nrec;
    const int n = nlist[w];
    int slot_of[FK_MAX_BODIES];
    for (int b = 0; b < nbody; b++) slot_of[b] = (share_owner >= 0 && w != share_owner) ? shared_slot[b] : -1;
    if (share_owner >= 0 && barrier_at[w] >= n) t.wave_tail_barrier[w] = 1;
    bool spare_used = false;                    // a wavefront's first parked parent lives in registers (slot code 254)
    for (int i = 1; i < n; i++) {               // a parent that is not the body walked just before is reloaded from a slot
      const int p = par[lists[w][i]];
      if (lists[w][i - 1] != p && slot_of[p] < 0) {
        if (!spare_used) { slot_of[p] = 254; spare_used = true; }
        else slot_of[p] = nslot++;
      }
    }
    for (int i = 0; i < n; i++) {
      const int b = lists[w][i];
      FkBodyRec r = t.rec[b];
      const int p = b == 0 ? 0 : par[b];
      const int src = (i == 0 || lists[w][i - 1] == p) ? -1 : slot_of[p];
      const bool own = !owned[b];
      owned[b] = true;
      // (a shared parent seeded slot_of[] of the other wavefronts: it is a LOAD slot there, never a save slot)
      const int dst = (share_owner >= 0 && w != share_owner && shared_slot[b] >= 0) ? -1 : slot_of[b];
      r.meta = (r.meta & 0xFu) | (own ? 16u : 0u) | ((uint32_t)(src + 1) << 8) | ((uint32_t)(dst + 1) << 16) | ((uint32_t)b << 24);
      if (share_owner >= 0 && i == barrier_at[w]) r.meta |= 32u;
      if (w == share_owner) r.meta |= (uint32_t)extra_dst[b] << 6;
      if (nrec >= 2 * FK_MAX_BODIES) return "too many records";
      t.wrec[nrec++] = r;
    }
  }
  t.wave_start[t.nwave] = nrec;
  for (int w = t.nwave + 1; w <= FK_MAX_WAVES; w++) t.wave_start[w] = nrec;
  t.nslot_split = nslot;
  // Where every joint angle of a wavefront's list is parked in the lane's staging row (read in one batch before the
  // walk): a body this wavefront stores -- the x slot of its own position (overwritten by that position right after
  // the angle was consumed); an ancestor another wavefront stores -- a y / z slot of a body this wavefront stores
  // LATER in its list (free until then), or, when those run out, an extra LDS column of the block.
  bool park_ok = true;
  memset(t.wave_park, 0xff, sizeof t.wave_park);
  t.nextra = 0;
  for (int w = 0; w < t.nwave && park_ok; w++) {
    const int i0 = t.wave_start[w], i1 = t.wave_start[w + 1];
    int park_of[2 * FK_MAX_BODIES];
    int npark = 0, next_free = i0, sub = 1;     // candidate: slot `sub` (1 = y, 2 = z) of the owned record `next_free`
    for (int i = i0; i < i1; i++) {
      const FkBodyRec& r = t.wrec[i];
      park_of[i - i0] = -1;
      if (!(r.meta & 1u)) continue;
      const int b = (int)(r.meta >> 24);
      int off = -1;
      if (r.meta & 16u) off = 3 * b;
      else {
        for (;;) {
          if (next_free <= i) { next_free = i + 1; sub = 1; }
          if (next_free >= i1) break;
          if (!(t.wrec[next_free].meta & 16u)) { next_free++; sub = 1; continue; }
          off = 3 * (int)(t.wrec[next_free].meta >> 24) + sub;
          if (++sub > 2) { next_free++; sub = 1; }
          break;
        }
      }
      if (npark >= 32 || r.dof_idx < 0 || r.dof_idx > 0xffff || off >= 0x8000) { park_ok = false; break; }
      const uint32_t code = off >= 0 ? (uint32_t)off : (0x8000u | (uint32_t)t.nextra++);      // (no free slot: an extra column)
      park_of[i - i0] = (int)code;
      t.wave_park[w][npark++] = (uint32_t)r.dof_idx | (code << 16);
    }
    for (int i = i0; i < i1 && park_ok; i++) {
      const int c = i + 1 < i1 ? park_of[i + 1 - i0] : -1;
      t.wrec[i].next_park = (t.wrec[i].next_park & 0x70000u) |      // ([18:16]: this body's zero translation components)
                            (c < 0 ? 0u : (0x80000000u | ((c & 0x8000) ? (0x40000000u | (uint32_t)(c & 0x7fff)) : (uint32_t)c)));
    }
  }
  if (!park_ok) t.nwave = 1;
  for (int b = 0; b < nbody; b++) if (!owned[b]) return "a body is not covered";
  return nullptr;
}

}  // namespace gmr
