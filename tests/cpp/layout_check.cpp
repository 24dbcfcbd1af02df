// Host-side checks of the static schedules the IK kernel relies on (plain C++, no HIP):
// reads a packed (gmr_model_t, gmr_taskset_t) pair from a file and verifies
//   * the H-assembly schedule: every structurally non-zero (i >= j) entry and every diagonal entry is
//     owned by exactly one virtual lane, its terms are exactly the (task, pair_i, pair_j) triples,
//     the last-term flags close every entry, for 64 and 192 virtual lanes;
//   * the limb / trunk decomposition: every dof exactly once, limbs are ancestor chains, trunk is
//     closed under "parent of", sizes within the tree solver's static bounds;
//   * the LDS layout: regions do not overlap and fit the 160 KB of a CU.
// Prints a one-line summary; exit code 0 = all good.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#include "../../general_motion_retargeting_amd/csrc/gmr_ik_layout.h"
#include "../../general_motion_retargeting_amd/csrc/gmr_ik_wide_layout.h"

#define CHECK(c, ...) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s : ", #c); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  gmr_model_t m;
  gmr_taskset_t ts;
  if (std::fread(&m, sizeof m, 1, f) != 1 || std::fread(&ts, sizeof ts, 1, f) != 1) return 2;
  std::fclose(f);
  const int nv = m.nv;
  for (int nl : {64, 192}) {
    gmr::IkSchedule sch = gmr::make_ik_schedule(m, ts, nl);
    const int cls = gmr::ik_size_class(m, ts), nvp = cls > 0 ? cls : 48, ldh = nvp + 1, zero_row = gmr::ik_caps(nvp).p;
    for (int s = 0; s < 2; s++) {
      // expected terms per entry
      std::map<std::pair<int, int>, std::multiset<std::pair<int, int>>> want;
      for (int k = 0; k < ts.ntask[s]; k++) {
        int c0 = ts.task_col0[s][k], n = ts.task_ncol[s][k];
        for (int a = 0; a < n; a++)
          for (int b = 0; b <= a; b++)
            want[{ts.pair_dof[s][c0 + a], ts.pair_dof[s][c0 + b]}].insert({c0 + a, c0 + b});
      }
      for (int d = 0; d < nv; d++) want[{d, d}];   // diagonal always present
      std::map<std::pair<int, int>, std::multiset<std::pair<int, int>>> got;
      std::set<std::pair<int, int>> closed;
      std::map<std::pair<int, int>, int> halves;
      CHECK((int)sch.istart[s].size() == nl + 1, "istart size");
      CHECK(sch.istart[s][0] == 0 && sch.istart[s][nl] == (int)sch.items[s].size(), "istart ends");
      int maxload = 0;
      for (int l = 0; l < nl; l++) {
        CHECK(sch.istart[s][l] <= sch.istart[s][l + 1], "istart monotone");
        maxload = std::max(maxload, sch.istart[s][l + 1] - sch.istart[s][l]);
        bool open = false;
        std::pair<int, int> cur;
        for (int i = sch.istart[s][l]; i < sch.istart[s][l + 1]; i++) {
          const uint64_t w64 = sch.items[s][i];
          const uint32_t lo = (uint32_t)w64, hi = (uint32_t)(w64 >> 32);
          // destinations are byte offsets of H[i][j] and H[j][i] (row stride ldh doubles); rows byte offsets of Jw rows
          const int o1 = (int)(hi & 0x7fffu) / 8, o2 = (int)((hi >> 15) & 0x7fffu) / 8;
          CHECK((hi & 7u) == 0 && ((hi >> 15) & 7u) == 0, "store offsets must be multiples of 8");
          std::pair<int, int> e = {o1 / ldh, o1 % ldh};
          CHECK(o2 == e.second * ldh + e.first, "the second store is the transposed entry");
          CHECK(e.first >= e.second && e.first < nv, "entry range");
          const bool atomic_lane = l < sch.atomic_lanes[s];      // half entries, added to H by two lane pairs
          if (!atomic_lane) CHECK(((hi >> 30) & 1u) == (e.first == e.second ? 1u : 0u), "diagonal flag");
          else if ((hi >> 30) & 1u) { CHECK(e.first == e.second, "diagonal flag on an off-diagonal half"); }
          CHECK((lo & 0xffffu) % 48 == 0 && (lo >> 16) % 48 == 0, "row offsets must be multiples of 48");
          const int ra = (int)(lo & 0xffffu) / 48, rb = (int)(lo >> 16) / 48;
          const bool nop = ra == zero_row;
          CHECK(nop ? rb == zero_row : (ra < ts.npair[s] && rb < ts.npair[s]), "row range");
          const uint32_t w = (nop ? 1u << 30 : 0u) | (hi & (1u << 31)) | (uint32_t)ra | ((uint32_t)rb << 9);
          const bool pairlane = l < 2 * sch.npaired[s];
          if (open) CHECK(e == cur, "terms of one entry must be contiguous in one lane");
          else {
            if (!atomic_lane)
              CHECK(pairlane ? (!(l & 1) ? !closed.count(e) : closed.count(e) == 1) : !closed.count(e),
                    "entry (%d,%d) owned twice", e.first, e.second);
            else halves[e]++;                                     // counted below: four lanes (two pairs) per split entry
            cur = e; open = true;
          }
          if (!((w >> 30) & 1u)) got[e].insert({(int)(w & 511u), (int)((w >> 9) & 511u)});
          else got[e];
          if (w >> 31) { closed.insert(e); open = false; }
        }
        CHECK(!open, "lane %d ends inside an entry", l);
      }
      for (int i = 0; i < sch.npaired[s]; i++)   // the two halves of a pair close in the same slot
        CHECK(sch.istart[s][2 * i + 1] - sch.istart[s][2 * i] == sch.istart[s][2 * i + 2] - sch.istart[s][2 * i + 1], "pair %d halves differ in length", i);
      CHECK(sch.atomic_lanes[s] % 4 == 0 && sch.atomic_lanes[s] <= sch.pair_lanes, "atomic lanes");
      {
        std::set<int> zero(sch.zero_off[s].begin(), sch.zero_off[s].end());
        CHECK(zero.size() == sch.zero_off[s].size() && (int)zero.size() <= gmr::IK_MAX_ZERO, "zero list");
        for (auto& h : halves) {
          CHECK(h.second == 4, "split entry (%d,%d) seen on %d lanes", h.first.first, h.first.second, h.second);
          CHECK(zero.count(8 * (h.first.first * ldh + h.first.second)) && zero.count(8 * (h.first.second * ldh + h.first.first)),
                "cells of a split entry must be zeroed");
        }
        CHECK((int)halves.size() * 4 == sch.atomic_lanes[s], "atomic lanes = four per split entry");
      }
      CHECK(got == want, "stage %d: schedule terms differ from J^T J structure (%zu vs %zu entries)", s, got.size(), want.size());
      CHECK(closed.size() == want.size(), "unclosed entries");
      int total = (int)sch.items[s].size();
      // (lanes are filled up to the four slots of one loop trip)
      CHECK(maxload <= std::max(4, (total + nl - 1) / nl + ts.ntask[s] + 2), "schedule badly balanced: max %d of %d over %d lanes", maxload, total, nl);
    }
  }
  // tree decomposition
  gmr::IkTree tr = gmr::make_ik_tree(m);
  std::vector<int> pd(nv, -1);
  for (int d = 1; d < 6; d++) pd[d] = d - 1;
  for (int h = 0; h < m.nhinge; h++) {
    int b = m.parent[m.hinge_body[h]], p = 5;
    while (b > 0) { if (m.body_hinge[b] >= 0) { p = 6 + m.body_hinge[b]; break; } b = m.parent[b]; }
    pd[6 + h] = p;
  }
  int nlimb = 0;
  if (tr.ok) {
    std::vector<int> seen(nv, 0);
    std::set<int> trunk;
    for (int t = 0; t < 10; t++) if (tr.trunk[t] >= 0) { seen[tr.trunk[t]]++; trunk.insert(tr.trunk[t]); }
    CHECK((int)trunk.size() == tr.nt && tr.nt <= 10, "trunk size");
    for (int d = 0; d < 6; d++) CHECK(trunk.count(d), "floating base must be in the trunk");
    for (int l = 0; l < 4; l++) {
      int prev = -1, len = 0;
      for (int a = 0; a < 8; a++) {
        int d = tr.limb[l][a];
        if (d < 0) continue;
        seen[d]++; len++;
        if (prev >= 0) CHECK(pd[prev] == d, "limb %d is not an ancestor chain (tip first)", l);
        prev = d;
      }
      if (len) { nlimb++; CHECK(trunk.count(pd[prev]), "limb %d must hang off the trunk", l); }
    }
    for (int d = 0; d < nv; d++) CHECK(seen[d] == 1, "dof %d covered %d times", d, seen[d]);
    // no coupling between different limbs: no dof of one limb is an ancestor of a dof of another
  }
  // layouts
  int lds[2] = {0, 0}, nd[2] = {0, 0}, nwd[2] = {0, 0};
  for (int nw : {1, 4}) {
    gmr::IkSchedule sch = gmr::make_ik_schedule(m, ts, nw == 1 ? 64 : 192);
    gmr::IkLayout L = gmr::make_ik_layout(m, ts, sch, nw);
    CHECK(L.smem_bytes > 0 && L.smem_bytes <= 160 * 1024 - 1024, "LDS bytes %d", L.smem_bytes);
    CHECK(L.nvp >= nv && (L.nvp == 28 || L.nvp == 32 || L.nvp == 36 || L.nvp == 48), "nvp");
    CHECK(L.nb <= L.o.cap.nb && L.nh <= L.o.cap.nh && L.nhum <= L.o.cap.nhum && L.K[0] <= L.o.cap.k && L.P[0] <= L.o.cap.p && L.P[1] <= L.o.cap.p, "class capacities");
    CHECK((L.o.ldh & 1) == 1 && L.o.ldh > nv, "odd H row stride");
    std::vector<char> img = gmr::make_ik_image(m, ts, sch, L);
    CHECK((int)img.size() >= L.smem_bytes && img.size() % 16 == 0, "image size");
    CHECK(L.tree_ok == (tr.ok ? 1 : 0), "tree flag");
    {   // the QP transpose scratch may alias the assembly scratch, never H / c / x / lo / hi
      const int need = nw == 4 ? std::max(L.nvp * (L.nvp + 1), 4 * 18 * 19) : std::max(L.nvp * (L.nvp + 1), 4 * 16 * 19 + 440);
      CHECK(L.o.tr_spart >= L.o.Kt && L.o.tr_rpart + 40 <= (nw == 4 ? L.o.n_double : L.o.H), "Schur exchange area");
      CHECK(L.o.Kt >= L.o.e && L.o.Kt + need <= L.o.H, "Kt [%d,%d) overlaps H at %d", L.o.Kt, L.o.Kt + need, L.o.H);
      CHECK(L.o.H + nv * L.o.ldh <= L.o.c, "H overlaps c");
    }
    lds[nw == 4] = L.smem_bytes; nd[nw == 4] = L.o.n_double; nwd[nw == 4] = (L.smem_bytes - L.o.n_double * 8 - L.o.n_short * 2) / 4;
  }
  // the one-wavefront-per-stream kernel's tables: pair slots are a permutation with the three dof classes contiguous, and
  // the H schedule names exactly the expected (row, row) terms through those slots, closing every entry once
  if (gmr::wide_fits(m, ts)) {
    std::vector<uint64_t> items[2];
    int ntrip[2];
    CHECK(gmr::make_wide_schedule(m, ts, items, ntrip), "wide schedule");
    for (int s = 0; s < 2; s++) {
      const std::vector<int> slot = gmr::wide_pair_slots(ts, s);
      std::set<int> seen(slot.begin(), slot.end());
      CHECK((int)slot.size() == ts.npair[s] && (int)seen.size() == ts.npair[s] && (seen.empty() || (*seen.begin() == 0 && *seen.rbegin() == ts.npair[s] - 1)), "slots are a permutation");
      std::vector<int> cls(ts.npair[s]);
      for (int p = 0; p < ts.npair[s]; p++) cls[slot[p]] = ts.pair_dof[s][p] < 3 ? 1 : (ts.pair_dof[s][p] < 6 ? 0 : 2);
      for (int i = 1; i < ts.npair[s]; i++) CHECK(cls[i - 1] <= cls[i], "dof classes contiguous");
      std::multiset<std::pair<int, int>> want, got;
      std::set<std::pair<int, int>> ents;
      for (int k = 0; k < ts.ntask[s]; k++) {
        int c0 = ts.task_col0[s][k], n = ts.task_ncol[s][k];
        for (int a = 0; a < n; a++)
          for (int b = 0; b <= a; b++) { want.insert({slot[c0 + a], slot[c0 + b]}); ents.insert({ts.pair_dof[s][c0 + a], ts.pair_dof[s][c0 + b]}); }
      }
      CHECK(ntrip[s] % 4 == 0 && (int)items[s].size() == 64 * ntrip[s], "wide trips");
      int closes = 0, halves = 0;
      for (uint64_t w : items[s]) {
        const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
        if (lo == gmr::WD_ITEM_NOP) { CHECK(!(hi >> 31), "padding never stores"); continue; }
        CHECK((lo & 0xffffu) % 48 == 0 && (lo >> 16) % 48 == 0, "row offsets");
        got.insert({(int)(lo & 0xffffu) / 48, (int)(lo >> 16) / 48});
        closes += hi >> 31;
        halves += (hi >> 31) && (hi & gmr::WD_ITEM_ADD);
        CHECK((hi >> 31) || !(hi & gmr::WD_ITEM_ADD), "the add flag rides on a closing term");
      }
      CHECK(got == want, "wide H terms (stage %d)", s);
      // an entry is closed by one store, or by the two adds of its halves
      CHECK(halves % 2 == 0 && closes - halves / 2 == (int)ents.size(), "every entry closed once: %d (%d halves) of %d", closes, halves, (int)ents.size());
    }
  }
  // stage flags: bit 1 of use1 = both tables name the same tasks, bit 2 = and the same (task, dof) pairs
  const gmr::IkParams prm = gmr::make_ik_params(m, ts);
  bool same = ts.use_stage[0] && ts.use_stage[1] && ts.ntask[0] == ts.ntask[1];
  for (int k = 0; same && k < ts.ntask[0]; k++) same = ts.task_body[0][k] == ts.task_body[1][k] && ts.task_human[0][k] == ts.task_human[1][k];
  CHECK(((prm.use1 & 2) != 0) == same, "same-task flag");
  CHECK(!(prm.use1 & 4) || (prm.use1 & 2), "same pairs imply same tasks");
  CHECK((prm.use1 != 0) == (ts.use_stage[1] != 0) && prm.use0 == ts.use_stage[0], "stage switches");
  std::printf("ok nv=%d tree=%d limbs=%d trunk=%d lds1=%d (%d doubles, %d words) lds4=%d (%d doubles, %d words) use1=%d\n", nv, (int)tr.ok, nlimb, tr.nt,
              lds[0], nd[0], nwd[0], lds[1], nd[1], nwd[1], prm.use1);
  return 0;
}
