// gmr_ik_layout.h -- LDS carve-up of one IK stream (shared by host launcher and kernel).
#pragma once
#include <stdint.h>

#include "../../include/gmr_types.h"

namespace gmr {

struct IkLayout {
  // dimensions
  int nb, nh, nq, nv, nhum, maxd, ldh;
  int K[2], P[2];
  // offsets in doubles
  int body_pos, body_quat, axis, range_lo, range_hi, scale, pos_off, quat_off;
  int wpos[2], wrot[2];
  int q, lq, xpos, xquat, xaxis, raw, tgt, e, we, M, Jw, H, Kf, c, x, lo, hi;
  int n_double;
  // offsets in shorts (after the doubles)
  int i_chain, i_depth, i_body_hinge, i_hinge_body, i_limited, i_is_foot;
  int i_task_body[2], i_task_human[2], i_task_col0[2], i_task_ncol[2], i_pair_task[2], i_pair_dof[2],
      i_pair_index[2];
  int n_short;
  int smem_bytes;
};

inline IkLayout make_ik_layout(const gmr_model_t& m, const gmr_taskset_t& ts) {
  IkLayout L{};
  L.nb = m.nbody; L.nh = m.nhinge; L.nq = m.nq; L.nv = m.nv; L.nhum = ts.nhuman;
  int maxd = 1;
  for (int b = 0; b < m.nbody; b++) if (m.depth[b] + 1 > maxd) maxd = m.depth[b] + 1;
  L.maxd = maxd;
  L.ldh = (m.nv % 2 == 0) ? m.nv + 1 : m.nv + 2;  // odd row stride (in doubles): conflict-free column reads
  for (int s = 0; s < 2; s++) { L.K[s] = ts.ntask[s]; L.P[s] = ts.npair[s]; }
  int Kmax = L.K[0] > L.K[1] ? L.K[0] : L.K[1];
  int Pmax = L.P[0] > L.P[1] ? L.P[0] : L.P[1];
  int o = 0;
  auto D = [&](int n) { int r = o; o += n; return r; };
  L.body_pos = D(3 * L.nb); L.body_quat = D(4 * L.nb); L.axis = D(3 * L.nb);
  L.range_lo = D(L.nh); L.range_hi = D(L.nh);
  L.scale = D(L.nhum); L.pos_off = D(3 * L.nhum); L.quat_off = D(4 * L.nhum);
  for (int s = 0; s < 2; s++) { L.wpos[s] = D(L.K[s]); L.wrot[s] = D(L.K[s]); }
  L.q = D(L.nq + 1); L.lq = D(4 * L.nb); L.xpos = D(3 * L.nb); L.xquat = D(4 * L.nb); L.xaxis = D(3 * L.nb);
  L.raw = D(7 * L.nhum + 1); L.tgt = D(7 * L.nhum + 1);
  L.e = D(6 * Kmax); L.we = D(6 * Kmax); L.M = D(18 * Kmax); L.Jw = D(6 * Pmax);
  L.H = D(L.nv * L.ldh); L.Kf = D(L.nv * L.ldh);
  L.c = D(L.nv); L.x = D(L.nv); L.lo = D(L.nv); L.hi = D(L.nv);
  L.n_double = o;
  int i = 0;
  auto I = [&](int n) { int r = i; i += n; return r; };
  L.i_chain = I(L.nb * maxd); L.i_depth = I(L.nb); L.i_body_hinge = I(L.nb);
  L.i_hinge_body = I(L.nh); L.i_limited = I(L.nh); L.i_is_foot = I(L.nhum);
  for (int s = 0; s < 2; s++) {
    L.i_task_body[s] = I(L.K[s]); L.i_task_human[s] = I(L.K[s]);
    L.i_task_col0[s] = I(L.K[s]); L.i_task_ncol[s] = I(L.K[s]);
    L.i_pair_task[s] = I(L.P[s]); L.i_pair_dof[s] = I(L.P[s]);
    L.i_pair_index[s] = I(L.K[s] * L.nv);
  }
  L.n_short = i;
  L.smem_bytes = L.n_double * 8 + ((L.n_short * 2 + 15) / 16) * 16;
  return L;
}

}  // namespace gmr
