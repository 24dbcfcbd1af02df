// Host check of the tree partition behind fk_split_kernel (csrc/gmr_fk_tree.h: fk_split_tree): reads "nbody p0 p1 ..."
// from stdin, checks that every body is walked by some wavefront, that every list is closed under `parent`, ascending
// and opened by the root, and that the longest list is well below the whole tree for trees that branch.
#include <cstdio>
#include <cstdlib>
#include <set>

#include "../../general_motion_retargeting_amd/csrc/gmr_fk_tree.h"

#define CHECK(c, ...) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s : ", #c); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); return 1; } } while (0)

int main() {
  int nb = 0;
  if (std::scanf("%d", &nb) != 1 || nb < 1 || nb > gmr::FK_MAX_BODIES) return 2;
  int parent[gmr::FK_MAX_BODIES];
  for (int b = 0; b < nb; b++) if (std::scanf("%d", &parent[b]) != 1) return 2;
  for (int maxw = 1; maxw <= gmr::FK_MAX_WAVES; maxw++) {
    int lists[gmr::FK_MAX_WAVES][gmr::FK_MAX_BODIES], nlist[gmr::FK_MAX_WAVES] = {0};
    const int nw = gmr::fk_split_tree(nb, parent, maxw, lists, nlist);
    CHECK(nw >= 1 && nw <= maxw, "nw = %d", nw);
    std::set<int> seen;
    int longest = 0, total = 0;
    for (int w = 0; w < nw; w++) {
      CHECK(nlist[w] >= 1 && lists[w][0] == 0, "list %d must open with the root", w);
      std::set<int> in;
      for (int i = 0; i < nlist[w]; i++) {
        const int b = lists[w][i];
        CHECK(b >= 0 && b < nb, "body range");
        if (i) CHECK(b > lists[w][i - 1], "list %d not ascending", w);
        if (b) CHECK(in.count(parent[b]) == 1, "list %d: parent of %d missing", w, b);
        in.insert(b);
        seen.insert(b);
      }
      longest = nlist[w] > longest ? nlist[w] : longest;
      total += nlist[w];
    }
    CHECK((int)seen.size() == nb, "maxw %d: %zu of %d bodies covered", maxw, seen.size(), nb);
    CHECK(total <= 2 * gmr::FK_MAX_BODIES, "record capacity");
    if (maxw == 1) CHECK(longest == nb, "one wavefront walks the whole tree");
    std::printf("maxw=%d nw=%d longest=%d total=%d\n", maxw, nw, longest, total);
  }
  std::printf("ok\n");
  return 0;
}
