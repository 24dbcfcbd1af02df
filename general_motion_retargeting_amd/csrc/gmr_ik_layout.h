// gmr_ik_layout.h -- LDS carve-up of one IK stream and the static H-assembly schedule
// (shared by the host launcher and the kernel).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/gmr_types.h"

namespace gmr {

constexpr int IK_MAX_HOPS = 5;  // pointer-jumping rounds of the FK: 2^5 = 32 > GMR_MAX_DEPTH

struct IkLayout {
  // dimensions
  int nb, nh, nq, nv, nvp, nw, nhum, maxd, nhop, ldh;
  int tree_ok, tree_nt;          // limb/trunk decomposition usable by the 4-wavefront tree solver
  int K[2], P[2], nitem[2], ntrip[2], nlanes, pair_lanes;
  // offsets in doubles
  int body_pos, body_quat, axis, range_lo, range_hi, scale, pos_off, quat_off;
  int wpos[2], wrot[2];
  int q, xa, xb, xaxis, raw, tgt, e, eaux, we, M, Jw, cpart, H, Kt, c, x, lo, hi, scal, tr_spart, tr_rpart;
  int n_double;
  // offsets in 32-bit words (after the doubles): the H-assembly schedule
  int w_items[2], w_istart[2], w_ctl, w_tr_mask, w_tr_cnt;
  int n_word;
  // nw == 1: the schedule stays in the global image (read through the vector L1, identical for all streams)
  // at these 32-bit word offsets from the image start; -1 when it lives in LDS (w_items)
  int g_items[2], image_bytes;
  // offsets in shorts (after the words)
  int i_hop, i_depth, i_body_hinge, i_hinge_body, i_limited, i_is_foot, i_tree_limb, i_tree_trunk;
  int i_task_body[2], i_task_human[2], i_pair_task[2], i_pair_dof[2], i_pair_index[2];
  int n_short;
  int smem_bytes;
};

// rows of the dense register-resident factorisation are padded to one of these sizes
inline int ik_padded_nv(int nv) {
  if (nv <= 28) return 28;
  if (nv <= 32) return 32;
  if (nv <= 36) return 36;
  if (nv <= 48) return 48;   // generic upper size (GMR_MAX_DOF = 46); none of the shipped robots needs it
  return -1;
}

// ---------------------------------------------------------------------------------------------
// Static schedule of H = sum_k (W J_k)^T (W J_k): the Jacobian of task k is non-zero only on the
// dofs of its root->frame path, so H[i][j] only receives terms from tasks whose path holds both i
// and j.  Every (i >= j) entry with its list of (pair_a, pair_b) terms is owned by exactly ONE lane
// (longest-processing-time assignment), which sums the terms in a fixed order in a register and
// stores the entry once: no read-modify-write, no atomics, deterministic.
// item word: [8:0] pair a, [17:9] pair b, [23:18] dof i, [29:24] dof j, [30] entry has no term,
//            [31] last term of the entry
// ---------------------------------------------------------------------------------------------
constexpr uint32_t IK_ITEM_NOP = 1u << 30;   // contributes nothing (and, without bit 31, closes nothing)

struct IkSchedule {
  int nlanes;                      // virtual lanes that share the assembly: 64 (one wave) or 192 (3 helpers)
  int pair_lanes;                  // lanes [0, pair_lanes) cooperate pairwise on one entry each
  std::vector<uint32_t> items[2];
  std::vector<int> istart[2];      // nlanes + 1 offsets
  // what the kernel reads: every lane padded to ntrip slots with no-op words, stored [slot][lane] so that
  // a wave reads consecutive words (conflict-free) and the loop trip count is wave-uniform
  int ntrip[2];
  int npaired[2];                  // entries split over a lane pair
  std::vector<uint32_t> padded[2];
};

inline IkSchedule make_ik_schedule(const gmr_model_t& m, const gmr_taskset_t& ts, int nlanes) {
  IkSchedule sch;
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
          terms[(size_t)da * nv + db].push_back((uint32_t)(c0 + a) | ((uint32_t)(c0 + b) << 9));
        }
    }
    struct Ent { int da, db, w; };
    std::vector<Ent> ents;
    for (int da = 0; da < nv; da++)
      for (int db = 0; db <= da; db++) {
        int w = (int)terms[(size_t)da * nv + db].size();
        if (w > 0 || da == db) ents.push_back({da, db, w});
      }
    std::stable_sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) { return x.w > y.w; });
    std::vector<std::vector<Ent>> per_lane(nlanes);
    std::vector<int> load(nlanes, 0);
    size_t first_single = 0;
    if (sch.pair_lanes) {
      // ents is sorted by weight: the heaviest pair_lanes/2 entries go to the lane pairs (marked da|64)
      for (; first_single < ents.size() && (int)first_single < sch.pair_lanes / 2; first_single++) {
        const Ent& e = ents[first_single];
        if (e.w < 4) break;
        per_lane[2 * first_single].push_back(e);
        load[2 * first_single] = load[2 * first_single + 1] = 1 << 20;   // pair lanes take nothing else
      }
      for (int l = 2 * (int)first_single; l < sch.pair_lanes; l++) load[l] = 1 << 20;   // unused: the wavefront runs in pair mode
    }
    for (size_t ei = first_single; ei < ents.size(); ei++) {
      const Ent& e = ents[ei];
      int best = -1;
      for (int l = 0; l < nlanes; l++) if (load[l] < (1 << 20) && (best < 0 || load[l] < load[best])) best = l;
      per_lane[best].push_back(e);
      load[best] += std::max(e.w, 1) + 1;  // +1: the two stores of the entry
    }
    sch.npaired[s] = (int)first_single;
    sch.items[s].clear();
    sch.istart[s].assign(nlanes + 1, 0);
    for (int l = 0; l < nlanes; l++) {
      sch.istart[s][l] = (int)sch.items[s].size();
      const bool paired = l < 2 * sch.npaired[s];
      if (paired) {
        const Ent& e = per_lane[l & ~1][0];
        const auto& tt = terms[(size_t)e.da * nv + e.db];
        const size_t half = (tt.size() + 1) / 2;              // both lanes get `half` slots
        const size_t lo2 = (l & 1) ? half : 0, hi2 = (l & 1) ? tt.size() : half;
        uint32_t dd = ((uint32_t)e.da << 18) | ((uint32_t)e.db << 24);
        for (size_t i = 0; i < half; i++) {
          const bool have = lo2 + i < hi2;
          uint32_t w = (have ? tt[lo2 + i] : IK_ITEM_NOP) | dd;
          if (i + 1 == half) w |= 1u << 31;                   // both lanes close the entry in the same slot
          sch.items[s].push_back(w);
        }
        continue;
      }
      for (const Ent& e : per_lane[l]) {
        const auto& tt = terms[(size_t)e.da * nv + e.db];
        uint32_t dd = ((uint32_t)e.da << 18) | ((uint32_t)e.db << 24);
        if (tt.empty()) sch.items[s].push_back(dd | (1u << 30) | (1u << 31));
        for (size_t i = 0; i < tt.size(); i++)
          sch.items[s].push_back(tt[i] | dd | (i + 1 == tt.size() ? (1u << 31) : 0u));
      }
    }
    sch.istart[s][nlanes] = (int)sch.items[s].size();
    int nt = 0;
    for (int l = 0; l < nlanes; l++) nt = std::max(nt, sch.istart[s][l + 1] - sch.istart[s][l]);
    nt = nlanes == 64 ? (nt + 3) & ~3         // four slots per loop trip (items streamed from global memory)
                      : (nt + 1) & ~1;        // two slots per loop trip
    sch.ntrip[s] = nt;
    sch.padded[s].assign((size_t)nt * nlanes, IK_ITEM_NOP);
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

inline IkLayout make_ik_layout(const gmr_model_t& m, const gmr_taskset_t& ts, const IkSchedule& sch, int nw) {
  IkLayout L{};
  L.nw = nw;
  const IkTree tree = make_ik_tree(m);
  L.tree_ok = (nw == 4 && tree.ok) ? 1 : 0;
  L.tree_nt = tree.nt;
  L.nb = m.nbody; L.nh = m.nhinge; L.nq = m.nq; L.nv = m.nv; L.nhum = ts.nhuman;
  L.nvp = ik_padded_nv(m.nv);
  int maxd = 1;
  for (int b = 0; b < m.nbody; b++) if (m.depth[b] + 1 > maxd) maxd = m.depth[b] + 1;
  L.maxd = maxd;
  L.nhop = 0;
  while ((1 << L.nhop) < maxd) L.nhop++;
  L.ldh = (m.nv % 2 == 0) ? m.nv + 1 : m.nv + 2;  // odd row stride (in doubles): conflict-free column reads
  for (int s = 0; s < 2; s++) { L.K[s] = ts.ntask[s]; L.P[s] = ts.npair[s]; L.nitem[s] = (int)sch.padded[s].size(); L.ntrip[s] = sch.ntrip[s]; }
  L.nlanes = sch.nlanes;
  L.pair_lanes = sch.pair_lanes;
  int Kmax = std::max(L.K[0], L.K[1]);
  int Pmax = std::max(L.P[0], L.P[1]);
  int o = 0;
  auto D = [&](int n) { int r = o; o += n; return r; };
  L.body_pos = D(3 * L.nb); L.body_quat = D(4 * L.nb); L.axis = D(3 * L.nb);
  L.range_lo = D(L.nh); L.range_hi = D(L.nh);
  L.scale = D(L.nhum); L.pos_off = D(3 * L.nhum); L.quat_off = D(4 * L.nhum);
  for (int s = 0; s < 2; s++) { L.wpos[s] = D(L.K[s]); L.wrot[s] = D(L.K[s]); }
  L.q = D(L.nq + 1);
  L.xa = D(7 * L.nb + 1);                            // FK result; the second buffer of the FK rounds aliases Jw
  L.xaxis = D(3 * L.nb);
  L.tgt = D(7 * L.nhum + 1);
  L.e = D(6 * Kmax); L.eaux = D(3 * Kmax); L.we = D(6 * Kmax);
  L.M = D(std::max(18 * Kmax, 7 * L.nhum + 1));
  L.raw = L.M;                                // the raw frame is consumed by the preprocess step, before any solve
  if (o & 1) o++;                             // Jw rows (48 B) are read as three 16-B pieces
  L.Jw = D(std::max(6 * Pmax, 7 * L.nb + 1)); L.cpart = D(Pmax);
  L.xb = L.Jw;                                // FK runs between solves, when the assembly scratch is dead
  // The transpose scratch of the QP solvers (dense: nvp x (nvp+1); tree: 4 x 18 x 19) is only live inside a
  // solve, when the assembly scratch [e, eaux, we, M, Jw, cpart] is dead: alias it there (saves ~11 KB of
  // LDS per stream = one more resident workgroup per CU in the throughput shape).
  {
    const int need = std::max(L.nvp * (L.nvp + 1), nw == 4 ? 4 * 18 * 19 : 0);
    const int have = o - L.e;
    if (have < need) D(need - have);
    L.Kt = L.e;
  }
  if (o & 1) o++;
  L.H = D(L.nv * L.ldh + 2);
  L.c = D(L.nv); L.x = D(L.nv); L.lo = D(L.nv); L.hi = D(L.nv); L.scal = D(2);
  L.tr_spart = D(nw == 4 ? 4 * 10 * 10 : 0); L.tr_rpart = D(nw == 4 ? 4 * 10 : 0);
  L.n_double = o;
  int w = 0;
  auto W = [&](int n) { int r = w; w += n; return r; };
  for (int s = 0; s < 2; s++) { L.w_items[s] = nw == 1 ? 0 : W(L.nitem[s]); L.w_istart[s] = 0; L.g_items[s] = -1; }
  L.w_ctl = W(2);
  if (w % 2) w++;
  L.w_tr_mask = W(16); L.w_tr_cnt = W(4);
  if (w % 2) w++;
  L.n_word = w;
  int i = 0;
  auto I = [&](int n) { int r = i; i += n; return r; };
  L.i_hop = I(IK_MAX_HOPS * L.nb); L.i_depth = I(L.nb); L.i_body_hinge = I(L.nb);
  L.i_hinge_body = I(L.nh); L.i_limited = I(L.nh); L.i_is_foot = I(L.nhum);
  L.i_tree_limb = I(4 * 8); L.i_tree_trunk = I(10);
  for (int s = 0; s < 2; s++) {
    L.i_task_body[s] = I(L.K[s]); L.i_task_human[s] = I(L.K[s]);
    L.i_pair_task[s] = I(L.P[s]); L.i_pair_dof[s] = I(L.P[s]);
    L.i_pair_index[s] = I(GMR_MAX_TASKS * L.nv);
  }
  L.n_short = i;
  L.smem_bytes = (L.n_double * 8 + L.n_word * 4 + L.n_short * 2 + 15) / 16 * 16;
  L.image_bytes = (L.smem_bytes + 15) / 16 * 16;
  if (nw == 1)
    for (int s = 0; s < 2; s++) { L.g_items[s] = L.image_bytes / 4; L.image_bytes += (L.nitem[s] * 4 + 15) / 16 * 16; }
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
  return p;
}

// Host-built image of one stream's LDS: the constant regions filled, the state regions zero.  The
// kernel prologue is then one coalesced global -> LDS copy instead of dozens of scattered loads
// (matters for the per-frame entry point, where the prologue is paid on every call).
inline std::vector<char> make_ik_image(const gmr_model_t& m, const gmr_taskset_t& ts, const IkSchedule& sch,
                                       const IkLayout& L) {
  std::vector<char> img((size_t)L.image_bytes, 0);
  double* sm = reinterpret_cast<double*>(img.data());
  uint32_t* sw = reinterpret_cast<uint32_t*>(sm + L.n_double);
  short* si = reinterpret_cast<short*>(sw + L.n_word);
  const int nb = L.nb, nv = L.nv;
  for (int i = 0; i < nb; i++) {
    for (int a = 0; a < 3; a++) sm[L.body_pos + 3 * i + a] = m.body_pos[i][a];
    for (int a = 0; a < 4; a++) sm[L.body_quat + 4 * i + a] = m.body_quat[i][a];
    int hh = m.body_hinge[i];
    for (int a = 0; a < 3; a++) sm[L.axis + 3 * i + a] = hh >= 0 ? m.hinge_axis[hh][a] : 0.0;
    int dep = m.depth[i];
    si[L.i_depth + i] = (short)dep;
    si[L.i_body_hinge + i] = (short)hh;
    for (int r = 0; r < L.nhop; r++)
      si[L.i_hop + r * nb + i] = (short)(dep >= (1 << r) ? m.chain[i][dep - (1 << r)] : 0);
  }
  for (int i = 0; i < L.nh; i++) {
    sm[L.range_lo + i] = m.range_lo[i];
    sm[L.range_hi + i] = m.range_hi[i];
    si[L.i_hinge_body + i] = (short)m.hinge_body[i];
    si[L.i_limited + i] = (short)m.limited[i];
  }
  {
    const IkTree tree = make_ik_tree(m);
    for (int l = 0; l < 4; l++) for (int a2 = 0; a2 < 8; a2++) si[L.i_tree_limb + l * 8 + a2] = (short)tree.limb[l][a2];
    for (int t2 = 0; t2 < 10; t2++) si[L.i_tree_trunk + t2] = (short)tree.trunk[t2];
    reinterpret_cast<int*>(sw + L.w_tr_cnt)[1] = -1;
    reinterpret_cast<int*>(sw + L.w_tr_cnt)[3] = -1;
  }
  for (int i = 0; i < L.nhum; i++) {
    sm[L.scale + i] = ts.scale[i];
    for (int a = 0; a < 3; a++) sm[L.pos_off + 3 * i + a] = ts.pos_off[i][a];
    for (int a = 0; a < 4; a++) sm[L.quat_off + 4 * i + a] = ts.quat_off[i][a];
    si[L.i_is_foot + i] = (short)ts.is_foot[i];
  }
  for (int s = 0; s < 2; s++) {
    uint32_t* dst = L.g_items[s] >= 0 ? reinterpret_cast<uint32_t*>(img.data()) + L.g_items[s] : sw + L.w_items[s];
    for (size_t i = 0; i < sch.padded[s].size(); i++) dst[i] = sch.padded[s][i];
    for (int k = 0; k < L.K[s]; k++) {
      si[L.i_task_body[s] + k] = (short)ts.task_body[s][k];
      si[L.i_task_human[s] + k] = (short)ts.task_human[s][k];
      sm[L.wpos[s] + k] = ts.w_pos[s][k];
      sm[L.wrot[s] + k] = ts.w_rot[s][k];
      for (int d = 0; d < nv; d++) si[L.i_pair_index[s] + k * nv + d] = (short)ts.pair_index[s][k][d];
    }
    // per (task, dof) pair, everything the Jacobian-column phase looks up, packed so that it is two
    // independent 16-bit reads instead of a chain of four: [3:0] task, [9:4] dof, [15:10] task body; hinge body
    for (int p = 0; p < L.P[s]; p++) {
      const int k = ts.pair_task[s][p], d = ts.pair_dof[s][p];
      si[L.i_pair_task[s] + p] = (short)(unsigned short)(k | (d << 4) | (ts.task_body[s][k] << 10));
      si[L.i_pair_dof[s] + p] = (short)(d >= 6 ? m.hinge_body[d - 6] : 0);
    }
  }
  return img;
}

}  // namespace gmr
