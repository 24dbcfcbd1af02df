// gmr_ik_layout.h -- LDS carve-up of one IK stream and the static H-assembly schedule
// (shared by the host launcher and the kernel).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/gmr_types.h"

namespace gmr {

constexpr int IK_MAX_ZERO = 64;  // H cells of entries whose halves are added (gmr_ik_layout.h: make_ik_schedule)
constexpr int IK_MAX_HOPS = 5;  // pointer-jumping rounds of the FK: 2^5 = 32 > GMR_MAX_DEPTH

// rows of the dense register-resident factorisation are padded to one of these sizes = the SIZE CLASS of a
// robot.  Every LDS offset is a compile-time constant of the class (and of the launch shape), so that the
// kernel holds no layout in scalar registers and folds the offsets into the LDS instructions: with the ~95
// runtime offsets of a per-robot layout the wave-uniform state did not fit the 102 SGPRs and one instruction
// in six was an SGPR spill or reload (v_writelane / v_readlane through ten VGPRs, plus their s_nop hazards).
struct IkCaps { int nb, nh, k, p, nhum; };     // capacity: bodies, hinges, tasks / stage, (task, dof) pairs / stage, human bodies
constexpr IkCaps ik_caps(int nvp) {
  return nvp == 28 ? IkCaps{34, 22, 16, 160, 16}
       : nvp == 32 ? IkCaps{34, 26, 16, 160, 16}
       : nvp == 36 ? IkCaps{40, 30, 16, 160, 16}
                   : IkCaps{GMR_MAX_BODIES, GMR_MAX_HINGES, GMR_MAX_TASKS, GMR_MAX_PAIRS, GMR_MAX_HUMAN};
}

struct IkOffsets {
  IkCaps cap;
  int nvp;                       // the class: padded row count of the dense solver, row stride of pair_index
  int ldh;                       // row stride of H (odd: conflict-free column reads)
  // offsets in doubles
  int body_pos, body_quat, axis, range_lo, range_hi, scale, pos_off, quat_off;
  int wpos[2], wrot[2];
  int q, xa, xb, xaxis, raw, tgt, e, eaux, we, M, Jw, cpart, H, Kt, c, x, lo, hi, scal, tr_spart, tr_rpart;
  int params;                    // damping, lm_damping, tol, limit_gain, ground_offset, dt (read where used: not in SGPRs)
  int hsc;                       // (sin, cos) of every hinge's half angle, written where q changes, read by the FK
  int n_double;
  // offsets in shorts (after the doubles)
  int i_hop, i_depth, i_body_hinge, i_hinge_body, i_limited, i_is_foot, i_tree_limb, i_tree_trunk;
  int i_task_body[2], i_task_human[2], i_pair_task[2], i_pair_dof[2], i_pair_index[2];
  int n_short;                   // padded to a multiple of 8 (the words start 16-byte aligned)
  // offsets in 32-bit words (after the shorts); the H-assembly schedule (runtime size) comes last
  int w_ctl, w_tr_mask, w_tr_cnt, w_zero, w_items0;
  int fixed_bytes;               // bytes up to the schedule
};

constexpr int ik_max(int a, int b) { return a > b ? a : b; }

constexpr IkOffsets ik_offsets(int nvp, int nw) {
  IkOffsets L{};
  const IkCaps cp = ik_caps(nvp);
  L.cap = cp;
  L.nvp = nvp;
  L.ldh = nvp + 1;               // 29 / 33 / 37 / 49: odd
  int o = 0;
  L.body_pos = o; o += 3 * cp.nb; L.body_quat = o; o += 4 * cp.nb; L.axis = o; o += 3 * cp.nb;
  L.range_lo = o; o += cp.nh; L.range_hi = o; o += cp.nh;
  L.params = o; o += 6;
  L.scale = o; o += cp.nhum; L.pos_off = o; o += 3 * cp.nhum; L.quat_off = o; o += 4 * cp.nhum;
  for (int s = 0; s < 2; s++) { L.wpos[s] = o; o += cp.k; L.wrot[s] = o; o += cp.k; }
  L.q = o; o += 7 + cp.nh + 1;
  L.hsc = o; o += 2 * cp.nh;
  L.xa = o; o += 7 * cp.nb + 1;                    // FK result; the second buffer of the FK rounds aliases Jw
  L.xaxis = o; o += 3 * cp.nb;
  L.tgt = o; o += 7 * cp.nhum + 1;
  L.e = o; o += 6 * cp.k; L.eaux = o; o += 5 * cp.k; L.we = o; o += 6 * cp.k;
  L.M = o; o += ik_max(18 * cp.k, 7 * cp.nhum + 1);
  L.raw = L.M;                                     // the raw frame is consumed by the preprocess step, before any solve
  if (o & 1) o++;                                  // Jw rows (48 B) are read as three 16-B pieces
  L.Jw = o; o += ik_max(6 * (cp.p + 1), 7 * cp.nb + 1);   // row cap.p stays zero: what items without a term read
  L.cpart = o; o += cp.p + 1;                      // slot cap.p stays zero: the share of an absent (task, dof) pair
  L.xb = L.Jw;                                     // FK runs between solves, when the assembly scratch is dead
  // The transpose scratch of the QP solvers (dense: nvp x (nvp+1); tree: 4 x 18 x 19) is only live inside a
  // solve, when the assembly scratch [e, eaux, we, M, Jw, cpart] is dead: alias it there.
  // nw == 1: the one-wavefront tree solver (four limbs in the four 16-lane rows) parks 4 x 16 x 19 transposes
  // and exchanges its Schur parts (4 x 10 x 10 + 4 x 10) in the same dead region.
  {
    const int rows_scr = 4 * 16 * 19;
    const int need = nw == 4 ? ik_max(nvp * (nvp + 1), 4 * 18 * 19) : ik_max(nvp * (nvp + 1), rows_scr + 440);
    const int have = o - L.e;
    if (have < need) o += need - have;
    L.Kt = L.e;
    if (nw != 4) { L.tr_spart = L.Kt + rows_scr; L.tr_rpart = L.tr_spart + 400; }
  }
  if (o & 1) o++;
  L.H = o; o += nvp * L.ldh + 2;
  L.c = o; o += nvp; L.x = o; o += nvp; L.lo = o; o += nvp; L.hi = o; o += nvp; L.scal = o; o += 2;
  if (nw == 4) { L.tr_spart = o; o += 4 * 10 * 10; L.tr_rpart = o; o += 4 * 10; }
  if (o & 1) o++;                                  // every region starts 16-byte aligned
  L.n_double = o;
  int i = 0;
  L.i_hop = i; i += IK_MAX_HOPS * cp.nb; L.i_depth = i; i += cp.nb; L.i_body_hinge = i; i += cp.nb;
  L.i_hinge_body = i; i += cp.nh; L.i_limited = i; i += cp.nh; L.i_is_foot = i; i += cp.nhum;
  L.i_tree_limb = i; i += 4 * 8; L.i_tree_trunk = i; i += 10;
  for (int s = 0; s < 2; s++) {
    L.i_task_body[s] = i; i += cp.k; L.i_task_human[s] = i; i += cp.k;
    L.i_pair_task[s] = i; i += cp.p; L.i_pair_dof[s] = i; i += cp.p;
    L.i_pair_index[s] = i; i += cp.k * nvp;        // [task][dof], row stride nvp
  }
  L.n_short = (i + 7) / 8 * 8;
  int w = 0;
  L.w_ctl = w; w += 4;                             // two mailbox slots {command, stage}, alternating per command
  L.w_tr_mask = w; w += 16;
  L.w_tr_cnt = w; w += 4;
  L.w_zero = w; w += 2 * IK_MAX_ZERO;              // [stage][IK_MAX_ZERO] byte offsets of the H cells zeroed per solve
  if (w % 4) w += 4 - w % 4;                       // the schedule starts 16-byte aligned
  L.w_items0 = w;
  L.fixed_bytes = L.n_double * 8 + L.n_short * 2 + w * 4;
  return L;
}

// what varies per robot inside a class: passed by value to the kernel (the only layout state in SGPRs)
struct IkDims {
  int nb, nh, nq, nv, nhum, nhop, tree_ok;
  int tree_small;                // limbs <= 7 dofs and trunk <= 9: the <7, 9> instance of the tree solver applies
  int K[2], P[2], ntrip[2], nlanes, pair_lanes;
  int atomic_lanes[2], nzero[2];   // half entries added to H (helpers' schedule) and the H cells zeroed for them
  int w_items[2];                // NW > 1: word offset of a stage's schedule from the start of the words
  int g_items[2];                // NW == 1: the schedule stays in the global image at these word offsets; else -1
  int smem_bytes;
};

// host-side description of one (robot, task set, launch shape)
struct IkLayout : IkDims {
  int nvp, nw, maxd, tree_nt, nitem[2], image_bytes;
  IkOffsets o;
};

// device-side: the runtime dims + the class's compile-time offsets (L.o.xxx folds to a constant)
template <int NVP, int NW>
struct IkLay : IkDims {
  static constexpr IkOffsets o = ik_offsets(NVP, NW);
  IkLay() = default;
  explicit IkLay(const IkDims& d) : IkDims(d) {}
};

// rows of the dense register-resident factorisation are padded to one of these sizes
inline int ik_padded_nv(int nv) {
  if (nv <= 28) return 28;
  if (nv <= 32) return 32;
  if (nv <= 36) return 36;
  if (nv <= 48) return 48;   // generic upper size (GMR_MAX_DOF = 46); none of the shipped robots needs it
  return -1;
}

// size class of a (robot, task set): the smallest whose capacities hold it (48 = generic, GMR_MAX_* capacities)
inline int ik_size_class(const gmr_model_t& m, const gmr_taskset_t& ts) {
  for (int nvp : {28, 32, 36, 48}) {
    const IkCaps c = ik_caps(nvp);
    if (m.nv <= nvp && m.nbody <= c.nb && m.nhinge <= c.nh && ts.nhuman <= c.nhum && ts.ntask[0] <= c.k &&
        ts.ntask[1] <= c.k && ts.npair[0] <= c.p && ts.npair[1] <= c.p)
      return nvp;
  }
  return -1;
}

// ---------------------------------------------------------------------------------------------
// Static schedule of H = sum_k (W J_k)^T (W J_k): the Jacobian of task k is non-zero only on the
// dofs of its root->frame path, so H[i][j] only receives terms from tasks whose path holds both i
// and j.  Every (i >= j) entry with its list of (pair_a, pair_b) terms is owned by exactly ONE lane
// (longest-processing-time assignment), which sums the terms in a fixed order in a register and
// stores the entry once: no read-modify-write, no atomics, deterministic.
// item (64 bit), pre-scaled to LDS byte offsets so that a term costs two integer instructions beside its six FMAs:
//   lo  [15:0] byte offset of Jw row a, [31:16] of row b (an item without a term names the zero row `cap.p` twice)
//   hi  [14:0] byte offset of H[i][j], [29:15] of H[j][i], [30] diagonal entry (the damping term is added),
//       [31] last term of the entry
// ---------------------------------------------------------------------------------------------
constexpr uint64_t ik_item_nop(int zero_row) { return (uint64_t)(48u * (uint32_t)zero_row) * 0x10001ull; }

struct IkSchedule {
  int nlanes;                      // virtual lanes that share the assembly: 64 (one wave) or 192 (3 helpers)
  int pair_lanes;                  // lanes [0, pair_lanes) cooperate pairwise on one entry each
  std::vector<uint64_t> items[2];
  std::vector<int> istart[2];      // nlanes + 1 offsets
  // what the kernel reads: every lane padded to ntrip slots with no-op words, stored [slot][lane] so that
  // a wave reads consecutive words (conflict-free) and the loop trip count is wave-uniform
  int ntrip[2];
  int npaired[2];                  // entries (or half entries) split over a lane pair
  int atomic_lanes[2];             // the first atomic_lanes lanes close by adding to H (half entries): their cells are zeroed first
  std::vector<int> zero_off[2];    // byte offsets in H of those cells
  std::vector<uint64_t> padded[2];
};

inline IkSchedule make_ik_schedule(const gmr_model_t& m, const gmr_taskset_t& ts, int nlanes, int ldh, int zero_row) {
  IkSchedule sch;
  const uint64_t NOP = ik_item_nop(zero_row);
  sch.nlanes = nlanes;
  // With three helper wavefronts (192 lanes) the longest entries (all tasks meet in the base x base
  // block) bound the phase.  The first wavefront's 64 lanes therefore work in PAIRS: lanes 2i and 2i+1
  // each sum half of the terms of one of the 32 heaviest entries, the halves are added with one
  // shuffle (fixed order: even + odd) and the even lane stores.  sch.pair_lanes = 64 or 0.
  sch.pair_lanes = nlanes > 64 ? 64 : 0;
  const int nv = m.nv;
  for (int s = 0; s < 2; s++) {
    std::vector<std::vector<uint32_t>> terms((size_t)nv * nv);
    for (int k = 0; k < ts.ntask[s]; k++) {
      int c0 = ts.task_col0[s][k], n = ts.task_ncol[s][k];
      for (int a = 0; a < n; a++)
        for (int b = 0; b <= a; b++) {
          int da = ts.pair_dof[s][c0 + a], db = ts.pair_dof[s][c0 + b];
          terms[(size_t)da * nv + db].push_back(48u * (uint32_t)(c0 + a) | ((48u * (uint32_t)(c0 + b)) << 16));
        }
    }
    // an entry, or -- for the heaviest entries of the helpers' schedule -- one HALF of its terms [t0, t1): the two halves
    // are summed by two lane pairs and ADDED to H (LDS atomic add onto a zeroed cell; two addends commute exactly, so the
    // result does not depend on which pair comes first).  An entry of 14 terms then costs 4 slots instead of 7.
    struct Ent { int da, db, w, t0, t1, atomic; };
    auto dest = [&](const Ent& e) {
      return (uint64_t)((uint32_t)(8 * (e.da * ldh + e.db)) | ((uint32_t)(8 * (e.db * ldh + e.da)) << 15) |
                        (e.da == e.db && e.t0 == 0 ? 1u << 30 : 0u)) << 32;   // (the damping term rides on the first half)
    };
    std::vector<Ent> ents;
    for (int da = 0; da < nv; da++)
      for (int db = 0; db <= da; db++) {
        int w = (int)terms[(size_t)da * nv + db].size();
        if (w > 0 || da == db) ents.push_back({da, db, w, 0, w, 0});
      }
    std::stable_sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) { return x.w > y.w; });
    std::vector<std::vector<Ent>> per_lane(nlanes);
    std::vector<Ent> pair_ents;                   // one per lane pair of the first wavefront
    std::vector<Ent> singles;
    sch.atomic_lanes[s] = 0;
    sch.zero_off[s].clear();
    if (sch.pair_lanes) {
      const int npair = sch.pair_lanes / 2;
      size_t nheavy = 0;
      while (nheavy < ents.size() && ents[nheavy].w > 8) nheavy++;
      size_t next = 0;
      if (nheavy > 0 && 2 * (int)nheavy <= npair && 2 * (int)nheavy <= IK_MAX_ZERO) {   // split every heavy entry into two atomic halves
        for (; next < nheavy; next++) {
          const Ent& e = ents[next];
          const int h = (e.w + 1) / 2;
          pair_ents.push_back({e.da, e.db, h, 0, h, 1});
          pair_ents.push_back({e.da, e.db, e.w - h, h, e.w, 1});
          sch.zero_off[s].push_back(8 * (e.da * ldh + e.db));
          if (e.da != e.db) sch.zero_off[s].push_back(8 * (e.db * ldh + e.da));
        }
        sch.atomic_lanes[s] = 2 * (int)pair_ents.size();
      }
      for (; next < ents.size() && (int)pair_ents.size() < npair && ents[next].w >= 4; next++) pair_ents.push_back(ents[next]);
      for (; next < ents.size(); next++) singles.push_back(ents[next]);
    } else {
      singles = ents;
    }
    // single lanes: first-fit-decreasing into the smallest multiple of four slots that holds everything
    {
      const int l0 = sch.pair_lanes, nl1 = nlanes - l0;
      int total = 0;
      for (const Ent& e : singles) total += std::max(e.w, 1);
      for (int cap = std::max(4, ((total + nl1 - 1) / nl1 + 3) & ~3);; cap += 4) {
        for (int l = l0; l < nlanes; l++) per_lane[l].clear();
        std::vector<int> load(nlanes, 0);
        bool fits = true;
        for (const Ent& e : singles) {
          int lane = -1;
          for (int l = l0; l < nlanes && lane < 0; l++) if (load[l] + std::max(e.w, 1) <= cap) lane = l;
          if (lane < 0) { fits = false; break; }
          per_lane[lane].push_back(e);
          load[lane] += std::max(e.w, 1);
        }
        if (fits) break;
      }
    }
    sch.npaired[s] = (int)pair_ents.size();
    sch.items[s].clear();
    sch.istart[s].assign(nlanes + 1, 0);
    for (int l = 0; l < nlanes; l++) {
      sch.istart[s][l] = (int)sch.items[s].size();
      const bool paired = l < 2 * sch.npaired[s];
      if (paired) {
        const Ent& e = pair_ents[l >> 1];
        const auto& tt = terms[(size_t)e.da * nv + e.db];
        const size_t cnt = (size_t)(e.t1 - e.t0), half = (cnt + 1) / 2;   // both lanes get `half` slots
        const size_t lo2 = e.t0 + ((l & 1) ? half : 0), hi2 = (l & 1) ? (size_t)e.t1 : e.t0 + half;
        const uint64_t dd = dest(e);
        for (size_t i = 0; i < half; i++) {
          const bool have = lo2 + i < hi2;
          uint64_t w = (have ? (uint64_t)tt[lo2 + i] : NOP) | dd;
          if (i + 1 == half) w |= 1ull << 63;                 // both lanes close the entry in the same slot
          sch.items[s].push_back(w);
        }
        continue;
      }
      for (const Ent& e : per_lane[l]) {
        const auto& tt = terms[(size_t)e.da * nv + e.db];
        const uint64_t dd = dest(e);
        if (tt.empty()) sch.items[s].push_back(NOP | dd | (1ull << 63));
        for (size_t i = 0; i < tt.size(); i++)
          sch.items[s].push_back((uint64_t)tt[i] | dd | (i + 1 == tt.size() ? (1ull << 63) : 0ull));
      }
    }
    sch.istart[s][nlanes] = (int)sch.items[s].size();
    int nt = 0;
    for (int l = 0; l < nlanes; l++) nt = std::max(nt, sch.istart[s][l + 1] - sch.istart[s][l]);
    nt = (nt + 3) & ~3;                       // four slots per loop trip
    sch.ntrip[s] = nt;
    sch.padded[s].assign((size_t)nt * nlanes, NOP);
    for (int l = 0; l < nlanes; l++)
      for (int i = sch.istart[s][l]; i < sch.istart[s][l + 1]; i++)
        sch.padded[s][(size_t)(i - sch.istart[s][l]) * nlanes + l] = sch.items[s][i];
  }
  return sch;
}

// Limb / trunk decomposition of the velocity dofs for the tree solver (gmr_ik_tree.h): limbs are the
// maximal leaf chains below the last branching dof; everything above (floating base, waist) is the
// trunk.  At most 4 limbs of <= 8 dofs are kept (the longest); shorter extra chains (a head) join the
// trunk, which may hold <= 10 dofs.  limb[l][a] = dof or -1 (tip first), trunk[t] = dof or -1.
struct IkTree {
  bool ok;
  int nt;
  int limb[4][8];
  int trunk[10];
};

inline IkTree make_ik_tree(const gmr_model_t& m) {
  IkTree tr;
  tr.ok = false; tr.nt = 0;
  for (auto& l : tr.limb) for (int& d : l) d = -1;
  for (int& d : tr.trunk) d = -1;
  const int nv = m.nv;
  std::vector<int> pd(nv, -1), cc(nv, 0);
  for (int d = 1; d < 6; d++) pd[d] = d - 1;
  for (int h = 0; h < m.nhinge; h++) {
    int b = m.parent[m.hinge_body[h]], p = 5;
    while (b > 0) {
      if (m.body_hinge[b] >= 0) { p = 6 + m.body_hinge[b]; break; }
      b = m.parent[b];
    }
    pd[6 + h] = p;
  }
  for (int d = 1; d < nv; d++) cc[pd[d]]++;
  std::vector<char> intrunk(nv, 0);
  for (int d = 0; d < 6; d++) intrunk[d] = 1;   // the floating base always belongs to the trunk
  for (int d = 0; d < nv; d++)
    if (cc[d] >= 2) for (int x = d; x >= 0; x = pd[x]) intrunk[x] = 1;
  std::vector<std::vector<int>> limbs;
  for (int d = 0; d < nv; d++)
    if (cc[d] == 0 && !intrunk[d]) {
      std::vector<int> ch;
      for (int x = d; x >= 0 && !intrunk[x]; x = pd[x]) ch.push_back(x);
      limbs.push_back(ch);
    }
  std::stable_sort(limbs.begin(), limbs.end(), [](const std::vector<int>& a, const std::vector<int>& b) { return a.size() > b.size(); });
  std::vector<int> trunk;
  for (int d = 0; d < nv; d++) if (intrunk[d]) trunk.push_back(d);
  for (size_t l = 0; l < limbs.size(); l++) {
    if (l < 4 && limbs[l].size() <= 8) continue;
    for (int d : limbs[l]) trunk.push_back(d);     // surplus / over-long chains are solved densely in the trunk
  }
  if (trunk.size() > 10) return tr;
  for (size_t l = 0; l < limbs.size() && l < 4; l++) {
    if (limbs[l].size() > 8) continue;
    for (size_t a = 0; a < limbs[l].size(); a++) tr.limb[l][a] = limbs[l][a];
  }
  std::sort(trunk.begin(), trunk.end());
  for (size_t t = 0; t < trunk.size(); t++) tr.trunk[t] = trunk[t];
  tr.nt = (int)trunk.size();
  // every dof must be covered exactly once
  std::vector<int> seen(nv, 0);
  for (auto& l : tr.limb) for (int d : l) if (d >= 0) seen[d]++;
  for (int d : tr.trunk) if (d >= 0) seen[d]++;
  for (int d = 0; d < nv; d++) if (seen[d] != 1) return tr;
  tr.ok = true;
  return tr;
}

// the schedule of a robot's size class (its H row stride and zero row)
inline IkSchedule make_ik_schedule(const gmr_model_t& m, const gmr_taskset_t& ts, int nlanes) {
  const int c = ik_size_class(m, ts), nvp = c > 0 ? c : 48;
  return make_ik_schedule(m, ts, nlanes, nvp + 1, ik_caps(nvp).p);
}

inline IkLayout make_ik_layout(const gmr_model_t& m, const gmr_taskset_t& ts, const IkSchedule& sch, int nw) {
  IkLayout L{};
  L.nw = nw;
  const IkTree tree = make_ik_tree(m);
  L.tree_ok = tree.ok ? 1 : 0;
  L.tree_nt = tree.nt;
  {
    int maxlimb = 0;
    for (auto& l : tree.limb) { int n = 0; for (int d : l) n += d >= 0; maxlimb = std::max(maxlimb, n); }
    L.tree_small = (L.tree_ok && maxlimb <= 7 && tree.nt <= 9) ? 1 : 0;
  }
  L.nb = m.nbody; L.nh = m.nhinge; L.nq = m.nq; L.nv = m.nv; L.nhum = ts.nhuman;
  L.nvp = ik_size_class(m, ts);
  int maxd = 1;
  for (int b = 0; b < m.nbody; b++) if (m.depth[b] + 1 > maxd) maxd = m.depth[b] + 1;
  L.maxd = maxd;
  L.nhop = 0;
  while ((1 << L.nhop) < maxd) L.nhop++;
  for (int s = 0; s < 2; s++) { L.K[s] = ts.ntask[s]; L.P[s] = ts.npair[s]; L.nitem[s] = (int)sch.padded[s].size(); L.ntrip[s] = sch.ntrip[s]; }
  L.nlanes = sch.nlanes;
  L.pair_lanes = sch.pair_lanes;
  for (int s = 0; s < 2; s++) {
    L.atomic_lanes[s] = sch.atomic_lanes[s];
    L.nzero[s] = (int)sch.zero_off[s].size();      // (<= IK_MAX_ZERO: at most 16 split entries, two cells each)
  }
  L.o = ik_offsets(L.nvp > 0 ? L.nvp : 48, nw);
  // the schedule: in LDS after the fixed part (NW > 1), or only in the global image (NW == 1)
  int w = L.o.w_items0;
  for (int s = 0; s < 2; s++) {
    L.w_items[s] = nw == 1 ? 0 : w;
    if (nw != 1) w += (2 * L.nitem[s] + 3) / 4 * 4;            // 64-bit items
    L.g_items[s] = -1;
  }
  L.smem_bytes = L.o.n_double * 8 + L.o.n_short * 2 + w * 4;
  L.image_bytes = L.smem_bytes;                    // a multiple of 16
  if (nw == 1)
    for (int s = 0; s < 2; s++) { L.g_items[s] = L.image_bytes / 4; L.image_bytes += (L.nitem[s] * 8 + 15) / 16 * 16; }
  return L;
}

// scalar parameters of a solve, passed by value to the kernel
struct IkParams {
  double damping, lm_damping, tol, limit_gain, ground_offset, dt;
  int max_iter, human_root, use0, use1;
};

inline IkParams make_ik_params(const gmr_model_t& m, const gmr_taskset_t& ts) {
  IkParams p;
  p.damping = ts.damping; p.lm_damping = ts.lm_damping; p.tol = ts.tol; p.limit_gain = ts.limit_gain;
  p.ground_offset = ts.ground_offset; p.dt = m.timestep;
  p.max_iter = ts.max_iter; p.human_root = ts.human_root; p.use0 = ts.use_stage[0]; p.use1 = ts.use_stage[1];
  // Both tables name the same (robot body, human body) tasks in the same order -- every shipped config: the unweighted
  // residual norm the second stage starts from IS the one the first stage ended with (bit for bit), so the kernels
  // skip that evaluation.  Carried as bit 1 of use1.
  bool same = ts.use_stage[0] && ts.use_stage[1] && ts.ntask[0] == ts.ntask[1];
  for (int k = 0; same && k < ts.ntask[0]; k++)
    same = ts.task_body[0][k] == ts.task_body[1][k] && ts.task_human[0][k] == ts.task_human[1][k];
  if (same) p.use1 |= 2;
  // ... and the same (task, dof) pairs: the body Jacobians the helpers produced after the first stage's last FK are the
  // second stage's too (bit 2)
  bool same_pairs = same && ts.npair[0] == ts.npair[1];
  for (int i = 0; same_pairs && i < ts.npair[0]; i++)
    same_pairs = ts.pair_task[0][i] == ts.pair_task[1][i] && ts.pair_dof[0][i] == ts.pair_dof[1][i];
  if (same_pairs) p.use1 |= 4;
  return p;
}

// Host-built image of one stream's LDS: the constant regions filled, the state regions zero.  The
// kernel prologue is then one coalesced global -> LDS copy instead of dozens of scattered loads
// (matters for the per-frame entry point, where the prologue is paid on every call).
inline std::vector<char> make_ik_image(const gmr_model_t& m, const gmr_taskset_t& ts, const IkSchedule& sch,
                                       const IkLayout& L) {
  std::vector<char> img((size_t)L.image_bytes, 0);
  double* sm = reinterpret_cast<double*>(img.data());
  short* si = reinterpret_cast<short*>(sm + L.o.n_double);
  uint32_t* sw = reinterpret_cast<uint32_t*>(si + L.o.n_short);
  const int nb = L.nb, nv = L.nv;
  for (int i = 0; i < nb; i++) {
    for (int a = 0; a < 3; a++) sm[L.o.body_pos + 3 * i + a] = m.body_pos[i][a];
    for (int a = 0; a < 4; a++) sm[L.o.body_quat + 4 * i + a] = m.body_quat[i][a];
    int hh = m.body_hinge[i];
    for (int a = 0; a < 3; a++) sm[L.o.axis + 3 * i + a] = hh >= 0 ? m.hinge_axis[hh][a] : 0.0;
    int dep = m.depth[i];
    si[L.o.i_depth + i] = (short)dep;
    si[L.o.i_body_hinge + i] = (short)hh;
    for (int r = 0; r < L.nhop; r++)
      si[L.o.i_hop + r * L.o.cap.nb + i] = (short)(dep >= (1 << r) ? m.chain[i][dep - (1 << r)] : 0);
  }
  for (int i = 0; i < L.nh; i++) {
    sm[L.o.range_lo + i] = m.range_lo[i];
    sm[L.o.range_hi + i] = m.range_hi[i];
    si[L.o.i_hinge_body + i] = (short)m.hinge_body[i];
    si[L.o.i_limited + i] = (short)m.limited[i];
  }
  {
    const IkTree tree = make_ik_tree(m);
    for (int l = 0; l < 4; l++) for (int a2 = 0; a2 < 8; a2++) si[L.o.i_tree_limb + l * 8 + a2] = (short)tree.limb[l][a2];
    for (int t2 = 0; t2 < 10; t2++) si[L.o.i_tree_trunk + t2] = (short)tree.trunk[t2];
    for (int s = 0; s < 2; s++)
      for (size_t i = 0; i < sch.zero_off[s].size() && i < (size_t)IK_MAX_ZERO; i++) sw[L.o.w_zero + s * IK_MAX_ZERO + i] = (uint32_t)sch.zero_off[s][i];
    reinterpret_cast<int*>(sw + L.o.w_tr_cnt)[1] = -1;
    reinterpret_cast<int*>(sw + L.o.w_tr_cnt)[3] = -1;
  }
  {
    const double prm[6] = {ts.damping, ts.lm_damping, ts.tol, ts.limit_gain, ts.ground_offset, m.timestep};
    for (int i = 0; i < 6; i++) sm[L.o.params + i] = prm[i];
  }
  for (int i = 0; i < L.nhum; i++) {
    sm[L.o.scale + i] = ts.scale[i];
    for (int a = 0; a < 3; a++) sm[L.o.pos_off + 3 * i + a] = ts.pos_off[i][a];
    for (int a = 0; a < 4; a++) sm[L.o.quat_off + 4 * i + a] = ts.quat_off[i][a];
    si[L.o.i_is_foot + i] = (short)ts.is_foot[i];
  }
  for (int s = 0; s < 2; s++) {
    uint32_t* dst = L.g_items[s] >= 0 ? reinterpret_cast<uint32_t*>(img.data()) + L.g_items[s] : sw + L.w_items[s];
    std::memcpy(dst, sch.padded[s].data(), sch.padded[s].size() * 8);
    for (int k = 0; k < L.o.cap.k; k++)
      for (int d = 0; d < L.nvp; d++) si[L.o.i_pair_index[s] + k * L.nvp + d] = (short)(8 * L.o.cap.p);
    for (int k = 0; k < L.K[s]; k++) {
      si[L.o.i_task_body[s] + k] = (short)ts.task_body[s][k];
      si[L.o.i_task_human[s] + k] = (short)ts.task_human[s][k];
      sm[L.o.wpos[s] + k] = ts.w_pos[s][k];
      sm[L.o.wrot[s] + k] = ts.w_rot[s][k];
      // (byte offset of the pair's c share in cpart; absent pairs -- and, below, absent tasks -- name the zero slot)
      for (int d = 0; d < nv; d++) si[L.o.i_pair_index[s] + k * L.nvp + d] = (short)(8 * (ts.pair_index[s][k][d] >= 0 ? ts.pair_index[s][k][d] : L.o.cap.p));
    }
    // per (task, dof) pair, everything the Jacobian-column phase looks up, packed so that it is two
    // independent 16-bit reads instead of a chain of four: [3:0] task, [9:4] dof, [15:10] task body; hinge body
    for (int p = 0; p < L.P[s]; p++) {
      const int k = ts.pair_task[s][p], d = ts.pair_dof[s][p];
      si[L.o.i_pair_task[s] + p] = (short)(unsigned short)(k | (d << 4) | (ts.task_body[s][k] << 10));
      si[L.o.i_pair_dof[s] + p] = (short)(d >= 6 ? m.hinge_body[d - 6] : 0);
    }
  }
  return img;
}

}  // namespace gmr
