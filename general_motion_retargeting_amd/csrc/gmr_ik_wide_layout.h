// gmr_ik_wide_layout.h -- the THROUGHPUT shape of the IK kernel (gmr_ik_wide.hip): one wavefront per stream with
// a per-stream LDS footprint small enough for two resident wavefronts per SIMD.
//
// Compared with the layout of gmr_ik_layout.h (latency shape, dense fallback):
//   * nothing that is the same for every stream lives in LDS: the robot's constants, the task tables, the tree
//     tables and the H schedule stay in ONE global image (this header builds it) and are read with coalesced
//     per-lane loads (vector L1 / L2 hits, shared by all streams of a CU) or scalar loads;
//   * H is stored in the block-arrowhead form the tree solver consumes (four 16 x 7 limb matrices [D_l; B_l] and
//     the 9 x 9 trunk block: 529 instead of 36 x 37 doubles), written there directly by the assembly schedule;
//   * the solver's input H is assembled into the space of the residual / -Jl^-1 scratch, the solver's own scratch
//     lives where the Jacobian columns were: every byte of the assembly is reused by the solve.
// 17.3 KB per stream instead of 37.5 KB: 9 instead of 4 streams per CU.
//
// Only robots that decompose into <= 4 limbs of <= 7 dofs and a trunk of <= 9 (every shipped robot) and fit the
// capacities below use this shape; others keep the one-wavefront kernel of gmr_ik.hip with the dense solver.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "gmr_ik_layout.h"

namespace gmr {

// capacities (one class: every shipped robot fits)
constexpr int WD_NB = 40;        // bodies
constexpr int WD_NH = 30;        // hinges (nv <= 36)
constexpr int WD_K = 16;         // tasks per stage
constexpr int WD_P = 160;        // (task, dof) pairs per stage
constexpr int WD_NHUM = 16;      // human bodies
constexpr int WD_NL = 7, WD_NT = 9;          // limb / trunk rows of the tree solver
constexpr int WD_LD = 9;                     // row stride of the solver's limb transposes (7 columns, odd stride)
constexpr int WD_TT = 10;                    // row stride of the trunk transpose (shared by the four row groups)
constexpr int WD_TRI = WD_NT * (WD_NT + 1) / 2;   // a Schur contribution is read as its lower triangle only

// ---- LDS (offsets in doubles) ---------------------------------------------------------------------------------
// Three regions are live at different times and share one block:
//   A = [e | wts | M] and C = [cpart]   residuals, weights, -Jl^-1 blocks (jlog -> column phase), column shares of c
//   B = [Jw]                            weighted Jacobian columns (column phase -> H assembly)
// H (the solver's input) is written by the assembly INTO A and C -- both are dead once the columns and c exist --
// and the solver's scratch lives in B, dead once H is assembled.  Nothing of a solve survives it except x.
struct WideLds {
  int q, hsc, xa, xaxis, tgt;                 // state
  int e, wts, M, Jw, cpart, eaux, raw, xb;    // assembly scratch (eaux aliases cpart, raw aliases M, xb aliases Jw)
  int lscr, tsh, spart, rpart, xl;            // solver scratch (aliases Jw)
  int H;                                      // compact H: LM[4][16][7] then T[9][9] (aliases e .. M and cpart)
  int c, x, lo, hi;
  int vset;                                   // 8 x u64: violation sets of the pivoting rounds (double-buffered)
  int n_double;
};
constexpr int WD_HT = 4 * 16 * 7;             // offset of T inside the compact H
constexpr int WD_HN = WD_HT + WD_NT * WD_NT;  // 529 doubles

constexpr WideLds wide_lds() {
  WideLds L{};
  int o = 0;
  L.q = o; o += 7 + WD_NH + 1;
  L.xa = o; o += 7 * WD_NB + 1;
  L.xaxis = o; o += 3 * WD_NB;
  L.tgt = o; o += 7 * WD_NHUM + 1;
  if (o & 1) o++;
  // regions A and C, contiguous: H must fit them
  L.e = o; o += 6 * WD_K;
  L.wts = o; o += 2 * WD_K;
  L.M = o; o += 18 * WD_K;
  L.raw = L.M;                                // consumed by the preprocess step, before any solve
  L.cpart = o; o += WD_P + 1;                   // slot WD_P is kept zero: the share of an absent (task, dof) pair
  L.eaux = L.cpart;                           // (a, sin, cos, t, 1/t) of the residual phase die before the columns are written
  L.H = L.e;
  if (o - L.e < WD_HN + 1) o = L.e + WD_HN + 1;
  if (o & 1) o++;                             // Jw rows (48 B) are read as three 16-B pieces
  // region B
  L.Jw = o; o += 6 * (WD_P + 1);                // row WD_P is kept zero: what the padding items of the schedule read
  L.xb = L.Jw;                                // FK runs between solves, when this region is dead
  // the solver's result and the half-angle sines / cosines live between a solve and the FK after it (integrate reads x and
  // writes hsc, FK reads hsc; hinge_sincos -> FK at the start of an item), when this region is dead as well
  L.hsc = L.Jw;
  L.x = L.Jw + 2 * WD_NH + 4;
  int t = L.Jw;
  L.lscr = t; t += 4 * 16 * WD_LD;
  L.tsh = t; t += WD_NT * WD_TT;
  L.spart = t; t += 4 * WD_TRI;
  L.rpart = t; t += 4 * WD_NT;
  L.xl = t; t += 64;
  if (t > o) o = t;
  if (o & 1) o++;
  L.c = o; o += 36; L.lo = o; o += 36; L.hi = o; o += 36;
  L.vset = o; o += 8;
  L.n_double = o;
  return L;
}
constexpr int WD_LDS_BYTES = wide_lds().n_double * 8;
static_assert(7 * WD_NB + 1 <= 6 * WD_P, "the second FK buffer must fit the Jw region");
static_assert(wide_lds().xl + 64 <= wide_lds().Jw + 6 * WD_P, "the solver scratch must fit the Jw region");
// LDS is handed out in granules of 1 280 bytes on gfx950 (measured: tools/micro/lds_occupancy.hip; the occupancy API says 512)
constexpr int WD_LDS_GRANULE = 1280;
static_assert((WD_LDS_BYTES + WD_LDS_GRANULE - 1) / WD_LDS_GRANULE * WD_LDS_GRANULE * 9 <= 160 * 1024, "nine streams per CU");
static_assert(2 * WD_NH + 4 + 36 <= 4 * 16 * WD_LD, "x and hsc must fit the head of the solver scratch");
static_assert(7 * WD_NHUM + 1 <= 18 * WD_K, "raw frame must fit the M region");
static_assert(5 * WD_K <= WD_P, "eaux must fit the cpart region");

// ---- global image (offsets in bytes; every section 16-byte aligned) ----------------------------------------------
// per-lane sections are indexed by the lane that consumes them, so that a phase starts with a few coalesced loads
struct WideImg {
  int fkc;        // f64 [WD_NB][10]    body_pos(3) body_quat(4) hinge axis(3)                  lane = body
  int fki;        // u32 [WD_NB][2]     {hop round r in bits [6r+5:6r], r < 5} {depth | (hinge+1) << 8}
  int pre;        // f64 [WD_NHUM][8]   scale, pos_off(3), quat_off(4)                         lane = human body
  int prei;       // u32 [WD_NHUM]      is_foot
  int task[2];    // f64 [WD_K][2] w_pos, w_rot ; then u32 [WD_K] body | human << 8              lane = task
  int taski[2];
  int pair[2];    // u32 [WD_P]         task | dof << 4 | task body << 10 | hinge body << 16   virtual lane = pair
  int cidx[2];    // u16 [WD_K][64]     byte offset of the c share of (task k, dof = lane) in cpart (absent: slot WD_P), as u32 [WD_K/2][64]
  int lim;        // f64 [64][2] range lo, hi ; then u32 [64] limited                             lane = dof
  int limi;
  int tree;       // i32 [64] dof of solver lane (16 l + row) or -1
  int prm;        // f64 [8]  damping, lm_damping, tol, limit_gain, ground_offset, dt (scalar loads where used)
  int items[2];   // u64 [ntrip][64]   (runtime size: last)
  int fixed_bytes;
};
constexpr int wd_up16(int x) { return (x + 15) / 16 * 16; }
constexpr WideImg wide_img() {
  WideImg I{};
  int o = 0;
  I.fkc = o; o += wd_up16(WD_NB * 10 * 8);
  I.fki = o; o += wd_up16(WD_NB * 2 * 4);
  I.pre = o; o += wd_up16(WD_NHUM * 8 * 8);
  I.prei = o; o += wd_up16(WD_NHUM * 4);
  for (int s = 0; s < 2; s++) {
    I.task[s] = o; o += wd_up16(WD_K * 2 * 8);
    I.taski[s] = o; o += wd_up16(WD_K * 4);
    I.pair[s] = o; o += wd_up16(WD_P * 4);
    I.cidx[s] = o; o += wd_up16(WD_K * 64 * 2);
  }
  I.lim = o; o += wd_up16(64 * 2 * 8);
  I.limi = o; o += wd_up16(64 * 4);
  I.tree = o; o += wd_up16(64 * 4);
  I.prm = o; o += 64;
  I.fixed_bytes = o;
  return I;
}

// what varies per robot: passed by value to the kernel
struct WideDims {
  int nb, nq, nv, nhum, nhop;
  int K[2], P[2], ntrip[2];
  int items[2];                   // byte offsets of the schedules in the image
};

struct WideLayout : WideDims {
  int ok;                         // the robot / task set fits this shape
  int image_bytes;
};

// H-assembly item (64 bit), everything pre-scaled to LDS byte offsets so that a slot costs two integer instructions
// beside its six FMAs.  lo: [15:0] byte offset of Jw row a, [31:16] of row b (padding items name the zero row WD_P
// twice).  hi: [12:0] first store offset in the compact H (bytes), [25:13] second store offset (the symmetric twin, or
// the same), [31] last term of the entry.  The damping term of the diagonal is added by the dof lanes afterwards.
constexpr uint32_t WD_ITEM_ADD = 1u << 30;   // (in the high word, with the close flag) this lane holds half of the entry: add, do not store
constexpr uint32_t WD_ITEM_NOP = (48u * WD_P) | ((48u * WD_P) << 16);
static_assert(48 * WD_P < 65536 && 8 * WD_HN < 8192, "item fields");

// position of a dof in the tree decomposition
struct WideLoc { int limb, idx; };            // limb = -1: trunk row idx

inline bool wide_fits(const gmr_model_t& m, const gmr_taskset_t& ts) {
  const IkTree tree = make_ik_tree(m);
  if (!tree.ok || tree.nt > WD_NT) return false;
  for (auto& l : tree.limb) { int n = 0; for (int d : l) n += d >= 0; if (n > WD_NL) return false; }
  if (m.nbody > WD_NB || m.nhinge > WD_NH || ts.nhuman > WD_NHUM) return false;
  for (int s = 0; s < 2; s++) if (ts.ntask[s] > WD_K || ts.npair[s] > WD_P) return false;
  int maxd = 1;
  for (int b = 0; b < m.nbody; b++) maxd = std::max(maxd, m.depth[b] + 1);
  return maxd <= 32;
}

// Where pair p of the task set lives in the kernel's tables (Jw row, c share, virtual lane of the pairs phase): base
// rotation columns first, then base translations, then hinges.  The pairs phase branches on that class; with the
// classes contiguous a 64-lane trip runs one or two of the three branches instead of all of them.  Sums are taken in
// task order whatever the slots are, so the order does not touch the arithmetic.
inline std::vector<int> wide_pair_slots(const gmr_taskset_t& ts, int s) {
  const int P = ts.npair[s];
  std::vector<int> order(P), slot(P);
  for (int p = 0; p < P; p++) order[p] = p;
  auto cls = [&](int p) { const int d = ts.pair_dof[s][p]; return d < 3 ? 1 : (d < 6 ? 0 : 2); };
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cls(x) < cls(y); });
  for (int i = 0; i < P; i++) slot[order[i]] = i;
  return slot;
}

// Static schedule of H = sum_k (W J_k)^T (W J_k) (see gmr_ik_layout.h) with the destinations expressed in the
// compact layout: every (i >= j) entry is owned by one lane (longest-processing-time assignment) which sums its
// terms in a fixed order and stores the entry once (twice for a symmetric twin).
inline bool make_wide_schedule_uncached(const gmr_model_t& m, const gmr_taskset_t& ts, std::vector<uint64_t> out[2], int ntrip[2]);

// The schedule depends on the STRUCTURE of the robot and the task tables only (which dofs each task sees), not on
// weights, offsets or the human's height: objects that differ in those alone -- one per AMASS subject in a dataset run --
// share one schedule (its conflict-aware ordering costs ~0.1 s to find).
inline bool make_wide_schedule(const gmr_model_t& m, const gmr_taskset_t& ts, std::vector<uint64_t> out[2], int ntrip[2]) {
  struct Cached { bool ok; std::vector<uint64_t> items[2]; int ntrip[2]; };
  static std::mutex mu;
  static std::map<std::string, Cached> cache;
  std::string key;
  auto add = [&](const void* p, size_t n) { key.append(reinterpret_cast<const char*>(p), n); };
  add(&m.nbody, sizeof m.nbody); add(&m.nv, sizeof m.nv); add(m.parent, sizeof m.parent); add(m.body_hinge, sizeof m.body_hinge);
  add(ts.ntask, sizeof ts.ntask); add(ts.npair, sizeof ts.npair); add(ts.task_col0, sizeof ts.task_col0);
  add(ts.task_ncol, sizeof ts.task_ncol); add(ts.pair_dof, sizeof ts.pair_dof); add(ts.pair_task, sizeof ts.pair_task);
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find(key);
    if (it != cache.end()) {
      for (int s = 0; s < 2; s++) { out[s] = it->second.items[s]; ntrip[s] = it->second.ntrip[s]; }
      return it->second.ok;
    }
  }
  Cached c;
  c.ok = make_wide_schedule_uncached(m, ts, c.items, c.ntrip);
  for (int s = 0; s < 2; s++) { out[s] = c.items[s]; ntrip[s] = c.ntrip[s]; }
  const bool ok = c.ok;
  std::lock_guard<std::mutex> g(mu);
  if (cache.size() < 64) cache.emplace(std::move(key), std::move(c));
  return ok;
}

inline bool make_wide_schedule_uncached(const gmr_model_t& m, const gmr_taskset_t& ts, std::vector<uint64_t> out[2], int ntrip[2]) {
  const int nv = m.nv;
  const IkTree tree = make_ik_tree(m);
  std::vector<WideLoc> loc(nv, WideLoc{-2, -1});
  for (int l = 0; l < 4; l++) for (int a = 0; a < 8; a++) if (tree.limb[l][a] >= 0) loc[tree.limb[l][a]] = {l, a};
  for (int t = 0; t < 10; t++) if (tree.trunk[t] >= 0) loc[tree.trunk[t]] = {-1, t};
  for (int s = 0; s < 2; s++) {
    std::vector<std::vector<uint32_t>> terms((size_t)nv * nv);
    const std::vector<int> slot = wide_pair_slots(ts, s);
    for (int k = 0; k < ts.ntask[s]; k++) {
      int c0 = ts.task_col0[s][k], n = ts.task_ncol[s][k];
      for (int a = 0; a < n; a++)
        for (int b = 0; b <= a; b++) {
          int da = ts.pair_dof[s][c0 + a], db = ts.pair_dof[s][c0 + b];
          terms[(size_t)da * nv + db].push_back(48u * (uint32_t)slot[c0 + a] | ((48u * (uint32_t)slot[c0 + b]) << 16));
        }
    }
    struct Ent { int da, db, w, first; uint32_t hi; };           // terms [first, first + w) of entry (da, db)
    std::vector<Ent> whole;
    for (int da = 0; da < nv; da++)
      for (int db = 0; db <= da; db++) {
        int w = (int)terms[(size_t)da * nv + db].size();
        if (w == 0) continue;                                    // the block is zero-filled before the schedule runs
        const WideLoc A = loc[da], B = loc[db];
        int o1, o2;
        if (A.limb >= 0 && B.limb >= 0) {
          if (A.limb != B.limb) return false;                    // two limbs never share a task path
          o1 = (16 * A.limb + A.idx) * 7 + B.idx; o2 = (16 * A.limb + B.idx) * 7 + A.idx;
        } else if (A.limb >= 0) { o1 = o2 = (16 * A.limb + 7 + B.idx) * 7 + A.idx; }
        else if (B.limb >= 0) { o1 = o2 = (16 * B.limb + 7 + A.idx) * 7 + B.idx; }
        else { o1 = WD_HT + A.idx * WD_NT + B.idx; o2 = WD_HT + B.idx * WD_NT + A.idx; }
        whole.push_back({da, db, w, 0, (uint32_t)(8 * o1) | ((uint32_t)(8 * o2) << 13)});
      }
    // A lane's cost is its number of terms (an entry's two stores ride on its last term).  The loop runs four slots per
    // trip, so what counts is the smallest multiple of four every lane fits in: first-fit-decreasing bin packing into 64
    // lanes of that capacity, the capacity raised until it works (G1: 1 014 terms -> 16 slots, the minimum).
    // An entry heavier than the mean load would alone set that capacity (G1's pruned first table: 666 terms, 10.4 per
    // lane, but 14 in each base-rotation entry): such entries are cut into two halves owned by two lanes, each ADDING
    // its sum to the zero-filled cell (WD_ITEM_ADD; two addends commute exactly) -- kept only when it saves a trip.
    auto pack = [&](const std::vector<Ent>& ents_in, std::vector<std::vector<Ent>>& per_lane) {
      std::vector<Ent> ents = ents_in;
      std::stable_sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) { return x.w > y.w; });
      int total = 0;
      for (const Ent& e : ents) total += e.w;
      for (int cap = std::max(4, ((total + 63) / 64 + 3) & ~3);; cap += 4) {
        per_lane.assign(64, {});
        std::vector<int> load(64, 0);
        bool fits = true;
        for (const Ent& e : ents) {
          int lane = -1;
          for (int l = 0; l < 64 && lane < 0; l++) if (load[l] + e.w <= cap) lane = l;
          if (lane < 0) { fits = false; break; }
          per_lane[lane].push_back(e);
          load[lane] += e.w;
        }
        if (fits) return cap;
      }
    };
    std::vector<std::vector<Ent>> per_lane, per_lane_cut;
    const int cap_whole = pack(whole, per_lane);
    {
      std::vector<Ent> cut;
      for (const Ent& e : whole) {
        if (e.w <= 8) { cut.push_back(e); continue; }
        const int h = (e.w + 1) / 2;
        cut.push_back({e.da, e.db, h, 0, e.hi | WD_ITEM_ADD});
        cut.push_back({e.da, e.db, e.w - h, h, e.hi | WD_ITEM_ADD});
      }
      if (pack(cut, per_lane_cut) < cap_whole) per_lane.swap(per_lane_cut);
    }
    // Which lane sums which entries is fixed by now (the packing above); what is still free -- the ORDER of a lane's
    // entries and which physical lane a list lives in -- decides the LDS bank conflicts of the row reads: a ds_read_b128
    // is served in four groups of 16 lanes (MI355X_MICROARCH.md, LDS), a 48-byte Jw row starts in one of 16 four-bank
    // windows, and two lanes of a group that read DIFFERENT rows of the same window serialise.  With the lists in packing
    // order the modelled reads take twice their conflict-free cycles (tools/micro/wide_conflicts.cpp); a deterministic
    // hill climb over (swap two entries of a lane | swap two lanes) removes most of that.  Sums are unchanged: an entry's
    // terms stay in their order, the two halves of a cut entry commute.
    {
      static const int group_lanes[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                             {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                             {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                             {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
      int grp_of[64];
      for (int g = 0; g < 4; g++) for (int k = 0; k < 16; k++) grp_of[group_lanes[g][k]] = g;
      int ntmax = 0;
      for (int l = 0; l < 64; l++) { int n = 0; for (const Ent& e : per_lane[l]) n += e.w; ntmax = std::max(ntmax, n); }
      ntmax = (ntmax + 3) & ~3;
      std::vector<std::vector<uint32_t>> rows(64);           // per lane: the lo words (two row offsets) of its slots
      auto fill = [&](int l) {
        rows[l].assign((size_t)ntmax, WD_ITEM_NOP);
        size_t i = 0;
        for (const Ent& e : per_lane[l]) {
          const auto& tt = terms[(size_t)e.da * nv + e.db];
          for (int t = e.first; t < e.first + e.w; t++) rows[l][i++] = tt[t];
        }
      };
      for (int l = 0; l < 64; l++) fill(l);
      auto group_cost = [&](int g) {
        long c = 0;
        for (int i = 0; i < ntmax; i++)
          for (int op = 0; op < 2; op++) {
            int addr[16][16], cnt[16] = {0};
            int mx = 1;
            for (int k = 0; k < 16; k++) {
              const uint32_t lo = rows[group_lanes[g][k]][i];
              const int a = (int)(op ? lo >> 16 : lo & 0xffffu), w = (a / 16) % 16;
              bool seen = false;
              for (int j = 0; j < cnt[w]; j++) seen = seen || addr[w][j] == a;
              if (!seen) { addr[w][cnt[w]++] = a; mx = std::max(mx, cnt[w]); }
            }
            c += mx - 1;
          }
        return c;
      };
      long cost[4];
      for (int g = 0; g < 4; g++) cost[g] = group_cost(g);
      uint32_t rng = 0x9e3779b9u + (uint32_t)s;
      auto rnd = [&](uint32_t n) { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return rng % n; };
      for (int iter = 0; iter < 40000 && cost[0] + cost[1] + cost[2] + cost[3] > 0; iter++) {
        if (rnd(4) == 0) {                                   // swap the lanes of two lists
          const int a = (int)rnd(64), b = (int)rnd(64);
          if (grp_of[a] == grp_of[b]) continue;
          std::swap(per_lane[a], per_lane[b]); std::swap(rows[a], rows[b]);
          const long ca = group_cost(grp_of[a]), cb = group_cost(grp_of[b]);
          if (ca + cb <= cost[grp_of[a]] + cost[grp_of[b]]) { cost[grp_of[a]] = ca; cost[grp_of[b]] = cb; }
          else { std::swap(per_lane[a], per_lane[b]); std::swap(rows[a], rows[b]); }
        } else {                                             // swap two entries of one lane
          const int l = (int)rnd(64);
          const size_t n = per_lane[l].size();
          if (n < 2) continue;
          const size_t x = rnd((uint32_t)n), y = rnd((uint32_t)n);
          if (x == y) continue;
          std::swap(per_lane[l][x], per_lane[l][y]);
          fill(l);
          const long c = group_cost(grp_of[l]);
          if (c <= cost[grp_of[l]]) cost[grp_of[l]] = c;
          else { std::swap(per_lane[l][x], per_lane[l][y]); fill(l); }
        }
      }
    }
    int nt = 0;
    std::vector<std::vector<uint64_t>> li(64);
    for (int l = 0; l < 64; l++) {
      for (const Ent& e : per_lane[l]) {
        const auto& tt = terms[(size_t)e.da * nv + e.db];
        for (int i = e.first; i < e.first + e.w; i++)
          li[l].push_back(((uint64_t)(i + 1 == e.first + e.w ? (e.hi | (1u << 31)) : (e.hi & ~WD_ITEM_ADD)) << 32) | tt[i]);
      }
      nt = std::max(nt, (int)li[l].size());
    }
    nt = (nt + 3) & ~3;                                           // four slots per loop trip
    ntrip[s] = nt;
    out[s].assign((size_t)nt * 64, (uint64_t)WD_ITEM_NOP);
    for (int l = 0; l < 64; l++)
      for (size_t i = 0; i < li[l].size(); i++) out[s][i * 64 + l] = li[l][i];
  }
  return true;
}

inline WideLayout make_wide_layout(const gmr_model_t& m, const gmr_taskset_t& ts, std::vector<char>* image) {
  WideLayout L{};
  L.ok = 0;
  if (!wide_fits(m, ts)) return L;
  std::vector<uint64_t> items[2];
  if (!make_wide_schedule(m, ts, items, L.ntrip)) return L;
  constexpr WideImg I = wide_img();
  L.nb = m.nbody; L.nq = m.nq; L.nv = m.nv; L.nhum = ts.nhuman;
  int maxd = 1;
  for (int b = 0; b < m.nbody; b++) maxd = std::max(maxd, m.depth[b] + 1);
  L.nhop = 0;
  while ((1 << L.nhop) < maxd) L.nhop++;
  int o = I.fixed_bytes;
  for (int s = 0; s < 2; s++) { L.K[s] = ts.ntask[s]; L.P[s] = ts.npair[s]; L.items[s] = o; o += wd_up16((int)items[s].size() * 8); }
  L.image_bytes = o;
  L.ok = 1;
  if (!image) return L;
  image->assign((size_t)o, 0);
  char* base = image->data();
  auto F = [&](int off) { return reinterpret_cast<double*>(base + off); };
  auto U = [&](int off) { return reinterpret_cast<uint32_t*>(base + off); };
  for (int b = 0; b < m.nbody; b++) {
    double* d = F(I.fkc) + 10 * b;
    for (int a = 0; a < 3; a++) d[a] = m.body_pos[b][a];
    for (int a = 0; a < 4; a++) d[3 + a] = m.body_quat[b][a];
    const int h = m.body_hinge[b];
    for (int a = 0; a < 3; a++) d[7 + a] = h >= 0 ? m.hinge_axis[h][a] : 0.0;
    const int dep = m.depth[b];
    uint32_t hops = 0;
    for (int r = 0; r < L.nhop; r++) hops |= (uint32_t)(dep >= (1 << r) ? m.chain[b][dep - (1 << r)] : 0) << (6 * r);
    U(I.fki)[2 * b] = hops;
    U(I.fki)[2 * b + 1] = (uint32_t)dep | ((uint32_t)(h + 1) << 8);
  }
  for (int i = 0; i < ts.nhuman; i++) {
    double* d = F(I.pre) + 8 * i;
    d[0] = ts.scale[i];
    for (int a = 0; a < 3; a++) d[1 + a] = ts.pos_off[i][a];
    for (int a = 0; a < 4; a++) d[4 + a] = ts.quat_off[i][a];
    U(I.prei)[i] = (uint32_t)ts.is_foot[i];
  }
  for (int s = 0; s < 2; s++) {
    for (int k = 0; k < ts.ntask[s]; k++) {
      F(I.task[s])[2 * k] = ts.w_pos[s][k];
      F(I.task[s])[2 * k + 1] = ts.w_rot[s][k];
      U(I.taski[s])[k] = (uint32_t)ts.task_body[s][k] | ((uint32_t)ts.task_human[s][k] << 8);
    }
    const std::vector<int> slot = wide_pair_slots(ts, s);
    for (int p = 0; p < ts.npair[s]; p++) {
      const int k = ts.pair_task[s][p], d = ts.pair_dof[s][p];
      U(I.pair[s])[slot[p]] = (uint32_t)k | ((uint32_t)d << 4) | ((uint32_t)ts.task_body[s][k] << 10) |
                        ((uint32_t)(d >= 6 ? m.hinge_body[d - 6] : 0) << 16);
    }
    uint16_t* ci = reinterpret_cast<uint16_t*>(base + I.cidx[s]);     // [k / 2][lane][k % 2]
    for (int k = 0; k < WD_K; k++)
      for (int d = 0; d < 64; d++) {
        int v = (k < ts.ntask[s] && d < m.nv) ? ts.pair_index[s][k][d] : -1;
        ci[((k / 2) * 64 + d) * 2 + (k % 2)] = (uint16_t)(8 * (v >= 0 ? slot[v] : WD_P));
      }
    std::memcpy(base + L.items[s], items[s].data(), items[s].size() * 8);
  }
  for (int d = 0; d < 64; d++) {
    const int h = d - 6;
    const bool lim = h >= 0 && h < m.nhinge && m.limited[h];
    F(I.lim)[2 * d] = lim ? m.range_lo[h] : 0.0;
    F(I.lim)[2 * d + 1] = lim ? m.range_hi[h] : 0.0;
    U(I.limi)[d] = lim ? 1u : 0u;
  }
  {
    const double prm[6] = {ts.damping, ts.lm_damping, ts.tol, ts.limit_gain, ts.ground_offset, m.timestep};
    for (int i = 0; i < 6; i++) F(I.prm)[i] = prm[i];
  }
  {
    const IkTree tree = make_ik_tree(m);
    int32_t* td = reinterpret_cast<int32_t*>(base + I.tree);
    for (int l = 0; l < 4; l++)
      for (int r = 0; r < 16; r++) td[16 * l + r] = r < WD_NL ? tree.limb[l][r] : tree.trunk[r - WD_NL];
  }
  return L;
}

}  // namespace gmr
