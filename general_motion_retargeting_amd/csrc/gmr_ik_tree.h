// gmr_ik_tree.h -- the box-QP of one IK solve, factorised along the kinematic tree by FOUR wavefronts.
//
// H = damping I + sum_k J_k^T W^2 J_k couples two dofs only if one is an ancestor of the other
// (a task's Jacobian lives on its root->frame path).  Ordered limbs-first, H is block-arrowhead:
//
//        [ D_1            B_1^T ]      D_l : dofs of limb l (a chain: dense, <= 8)
//    H = [      ...        ...  ]      T   : trunk dofs (floating base, waist, short appendages: <= 10)
//        [            D_4 B_4^T ]      B_l : trunk x limb coupling
//        [ B_1  ...  B_4   T    ]
//
// Each wavefront eliminates ONE limb (Cholesky of D_l, Y_l = B_l L_l^-T, its Schur contribution
// -Y_l Y_l^T and the forward-substituted right-hand side) with its rows in registers exactly like the
// dense solver, but on an 18-column local matrix: 108 instead of 630 (pivot, column) updates and 8
// instead of 36 pivots on the critical path, the four limbs side by side.  Every wavefront then sums,
// factors and solves the trunk Schur complement redundantly (no exchange of x_T), and the limbs
// back-substitute in parallel.  Bounds are handled by the same block principal pivoting as the dense solver
// (gmr_ik.hip) on wave-uniform bound masks kept identically in all wavefronts: two to three workgroup
// barriers per pivoting round (Schur parts; x when a bound is active; violation sets).
//
// Used by the latency shape (NW = 4) when the robot decomposes into <= 4 limbs of <= 8 dofs and a
// trunk of <= 10 (all 8 shipped robots do); other robots always run the 1-wavefront kernel (dense solver).
#pragma once

namespace gmr {

constexpr int TR_MAX_NL = 8;                     // capacity: limb rows per wavefront (LDS tables, scratch strides)
constexpr int TR_MAX_NT = 10;                    // capacity: trunk rows
constexpr int TR_LD = TR_MAX_NL + TR_MAX_NT + 1; // row stride of the LDS transpose scratch

// Bound sets of the QP, identical in every wavefront (wave-uniform registers, carried from solve to
// solve for the warm start): bit d of `lower` / `upper` = dof d sits on its lower / upper bound.
struct TreeState { unsigned long long lower, upper; };

// All four wavefronts call this together.  Returns GMR_STATUS_* (the same value in every wavefront);
// the solution is left in sm[L.o.x].  Three workgroup barriers per pivoting round (two when no bound is
// active): the trunk system is summed, factorised and solved redundantly by every wavefront, so
// the only exchanges are the limbs' Schur contributions, the solution x and the violation sets.
// TR_NL / TR_NT: rows actually eliminated (limbs <= TR_NL dofs, trunk <= TR_NT): the pivots are unrolled, so a
// robot with 7-dof limbs and a 9-dof trunk (every shipped one) runs the <7, 9> instance: 16 instead of 18 pivots.
//
// ROWS = true: the same algorithm inside ONE wavefront (the 1-wavefront launch shape).  The four limbs occupy
// the four 16-lane DPP rows of the wave (TR_NL + TR_NT <= 16: 7 limb rows + 9 trunk rows), "wavefront" becomes
// "row", v_readlane broadcasts become DPP row broadcasts (one instruction pair serves all four limbs; the
// result stays in a VGPR), workgroup barriers become LDS fences.  Replaces the dense 36-pivot factorisation
// of the throughput shape for robots that fit.
// What a wavefront reads from the decomposition tables for every solve: the dof of its row and the (wave-uniform) dofs
// of its local matrix' columns.  They never change during a launch: read once (tree_rows), kept in registers.
template <int TR_NV>
struct TreeRows { int dof; int cdof[TR_NV]; };

template <int TR_NL, int TR_NT, bool ROWS, class LT>
__device__ __forceinline__ TreeRows<TR_NL + TR_NT> tree_rows(const LT& L, const short* si, int wave_in, int lane_in) {
  const int wave = ROWS ? (lane_in >> 4) : wave_in, lane = ROWS ? (lane_in & 15) : lane_in;
  const short* limb = si + L.o.i_tree_limb + wave * TR_MAX_NL;
  const short* trunk = si + L.o.i_tree_trunk;
  TreeRows<TR_NL + TR_NT> R;
  R.dof = lane < TR_NL ? limb[lane] : (lane < TR_NL + TR_NT ? trunk[lane - TR_NL] : -1);
#pragma unroll
  for (int m = 0; m < TR_NL; m++) R.cdof[m] = limb[m];
#pragma unroll
  for (int u = 0; u < TR_NT; u++) R.cdof[TR_NL + u] = trunk[u];
  return R;
}

template <int TR_NL, int TR_NT, bool ROWS, bool DPPB, class LT>
__device__ __forceinline__ int solve_qp_tree(const LT& L, double* sm, uint32_t* sw, const short* si, int wave_in,
                                             int lane_in, TreeState& bs, Prof& pr, const TreeRows<TR_NL + TR_NT>& rows) {
  constexpr int TR_NV = TR_NL + TR_NT;             // local matrix order
  static_assert(!ROWS || TR_NV <= 16, "a limb's local matrix must fit one 16-lane row");
  const int wave = ROWS ? (lane_in >> 4) : wave_in;        // which limb this wavefront / row eliminates
  const int lane0 = ROWS ? (lane_in & 15) : lane_in;       // row of the local matrix
  // Every phase below starts from a FRESH copy of the row index (fresh_lane, gmr_device_math.h): its lane predicates
  // (row == pivot, row > pivot, ...) are recomputed where they are used and die with the phase, instead of being
  // computed once per kernel and parked in (spilled) SGPR pairs.
#define TR_ROW()                                                                                   \
  const int lane = fresh_lane(lane0);                                                              \
  const bool is_limb = lane < TR_NL, is_trunk = lane >= TR_NL && lane < TR_NV;                     \
  const int a = lane, t = lane - TR_NL;                                                            \
  (void)is_limb; (void)is_trunk; (void)a; (void)t;
  const int lane = lane0;
  static_assert(!DPPB || TR_NV <= 16, "DPP row broadcasts need the local matrix in one 16-lane row");
#define TR_BCAST(v, k) ((ROWS || DPPB) ? row_bcast_d((v), (k)) : readlane_d((v), (k)))
#define TR_SYNC() do { if (ROWS) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); else __syncthreads(); } while (0)
  const int n = L.nv, ldh = L.o.ldh;
  const double* H = sm + L.o.H;
  double* xs = sm + L.o.x;
  const double* los = sm + L.o.lo;
  const double* his = sm + L.o.hi;
  double* Spart = sm + L.o.tr_spart;                           // [4][TR_NT][TR_NT]
  double* rpart = sm + L.o.tr_rpart;                           // [4][TR_NT]
  double* Lscr = sm + L.o.Kt + wave * (ROWS ? 16 : TR_MAX_NL + TR_MAX_NT) * TR_LD;   // this wavefront's / row's transpose scratch
  // violation sets of a round, double-buffered: {to_lower, to_upper, release, flags} x 2
  unsigned long long* vset = reinterpret_cast<unsigned long long*>(sw + L.o.w_tr_mask);

  const bool is_limb = lane < TR_NL, is_trunk = lane >= TR_NL && lane < TR_NV;
  const int a = lane, t = lane - TR_NL;
  const int dof = rows.dof;                                  // dof of limb row a / trunk row t, or -1
  const bool row = dof >= 0;                                 // this lane holds a real row
  const bool own = row && (is_limb || wave == 0);            // ... and reports the variable's violations
  const double lo = row ? los[dof] : 0.0, hi = row ? his[dof] : 0.0;
  const double ci = row ? (sm + L.o.c)[dof] : 0.0;
  const double* Hrow = H + (row ? dof : 0) * ldh;
  double dual_tol = -1.0;                                      // 1e-13 (1 + max |c|): computed when a multiplier is first checked
  const double ptol_lo = 1e-12 * (1.0 + fabs(lo)), ptol_hi = 1e-12 * (1.0 + fabs(hi));
  // column dofs of the local matrix (wave-uniform): limb columns then trunk columns
  const int* cdof = rows.cdof;

  int pcount = 3, ninf_best = 65;
  for (int it = 0; it < 100; it++) {
    PROF_BEGIN(pr);
    const unsigned long long fixedm = bs.lower | bs.upper;
    const bool self_fixed = !row || ((fixedm >> dof) & 1ull);
    const double xfix = !row ? 0.0 : (((bs.lower >> dof) & 1ull) ? lo : (((bs.upper >> dof) & 1ull) ? hi : 0.0));
    // ---- (1) local rows and right-hand side -----------------------------------------------------
    double r[TR_NV];
#pragma unroll
    for (int m = 0; m < TR_NV; m++) r[m] = 0.0;
    double rhs0, b;
    {
    TR_ROW()
#pragma unroll
    for (int m = 0; m < TR_NL; m++) {
      const int cd = cdof[m];                                 // wave-uniform
      const bool cfixed = cd < 0 || ((fixedm >> cd) & 1ull);
      // limb row a keeps columns m <= a; trunk rows keep all limb columns (B_l)
      const bool keep = row && !self_fixed && !cfixed && (is_trunk || (is_limb && m <= a));
      double h = (row && cd >= 0) ? Hrow[cd] : 0.0;
      double v = keep ? h : 0.0;
      if (is_limb && m == a && (self_fixed || cfixed)) v = 1.0;   // fixed / padding limb row: identity
      r[m] = v;
    }
    // -c_i - sum over fixed j of H_ij x_j (the bound value of j from the uniform sets)
    rhs0 = row ? (self_fixed ? xfix : -ci) : 0.0;
    if (row && !self_fixed) {
      // (same terms in the same order, one iteration ahead with the loads: the LDS round trips of the fixed variables
      //  overlap instead of adding up -- the stream that sets the batch's time has five of them in every solve)
      unsigned long long mm = fixedm;
      if (mm) {
        int j = __ffsll((long long)mm) - 1;
        mm &= mm - 1;
        double h = Hrow[j], bv = (((bs.lower >> j) & 1ull) ? los : his)[j];
        while (mm) {
          const int jn = __ffsll((long long)mm) - 1;
          mm &= mm - 1;
          const double hn = Hrow[jn], bn = (((bs.lower >> jn) & 1ull) ? los : his)[jn];
          rhs0 -= h * bv;
          h = hn; bv = bn;
        }
        rhs0 -= h * bv;
      }
    }
    b = is_limb ? rhs0 : 0.0;
    }
    PROF_END(pr, PH_KBUILD);
    PROF_BEGIN(pr);
    // ---- (2) eliminate the limb pivots (right-looking, forward substitution merged) --------------
    // The next pivot's column is updated first and its reciprocal square root started at once, so
    // that the Newton steps overlap the remaining (independent) column updates of this pivot.
    double mydinv = 1.0;
    bool bad = false;
    double dp = TR_BCAST(r[0], 0);
    double dinv = fast_rsqrt(dp);
#pragma unroll
    for (int p = 0; p < TR_NL; p++) {
      const int lane = fresh_lane(lane0);                    // (lane > p), (lane == p): computed here, dead after this pivot
      bad = bad || !(dp > 0.0);
      const double rs = r[p] * dinv;                         // (row p holds the pivot itself: d_p / sqrt(d_p))
      double l = lane > p ? rs : 0.0;                        // column p of L_l (rows > p) and of Y_l
      r[p] = lane == p ? rs : l;
      if (lane == p) mydinv = dinv;
      double dinv_next = 1.0;
      if (p + 1 < TR_NL) {
        r[p + 1] = fma(-l, TR_BCAST(l, p + 1), r[p + 1]);
        dp = TR_BCAST(r[p + 1], p + 1);
        dinv_next = fast_rsqrt(dp);
      }
      const double yp = TR_BCAST(b, p) * dinv;               // row p keeps its unscaled b (l = 0 there): scaled after the loop
      b = fma(-l, yp, b);
#pragma unroll
      for (int k = (p + 1 < TR_NL ? p + 2 : p + 1); k < TR_NV; k++) r[k] = fma(-l, TR_BCAST(l, k), r[k]);
      dinv = dinv_next;
    }
    if (fresh_lane(lane0) < TR_NL) b *= mydinv;              // y_p = b_p / sqrt(d_p): the value every later row was given
    PROF_END(pr, PH_CHOL);
    PROF_BEGIN(pr);
    // ---- (3) publish the Schur contribution; park L_l / Y_l for the transposed reads --------------
    unsigned long long* vcur = vset + 4 * (it & 1);
    {
    TR_ROW()
    if (is_trunk) {
#pragma unroll
      for (int u = 0; u < TR_NT; u++) Spart[(wave * TR_MAX_NT + t) * TR_MAX_NT + u] = r[TR_NL + u];
      rpart[wave * TR_MAX_NT + t] = b;
    }
    if (lane < TR_NV) {
#pragma unroll
      for (int m = 0; m < TR_NL; m++) Lscr[lane * TR_LD + m] = r[m];
    }
    if (lane == 0 && bad) atomicOr(&vcur[3], 1ull);
    TR_SYNC();                                                                               // B1
    // the other slot was last read before this barrier: clear it for the next round
    if (wave == 0 && lane < 4) vset[4 * ((it + 1) & 1) + lane] = 0ull;
    }
    PROF_END(pr, PH_SUBST);
    PROF_BEGIN(pr);
    // ---- (4) every wavefront: trunk Schur complement, factor, solve (redundant, no exchange) -------
    double bt = 0.0;
    bool tbad = false;
    {
      double s[TR_NT];
#pragma unroll
      for (int u = 0; u < TR_NT; u++) s[u] = 0.0;
      {
        TR_ROW()
        // all loads first (clamped addresses, no branches), then the masks
        const int tt = is_trunk ? t : 0;
        double hv[TR_NT], sp[TR_NT];
#pragma unroll
        for (int u = 0; u < TR_NT; u++) {
          const int cd = cdof[TR_NL + u];
          hv[u] = Hrow[cd >= 0 ? cd : 0];
          const double* q0 = Spart + tt * TR_MAX_NT + u;
          sp[u] = (q0[0] + q0[TR_MAX_NT * TR_MAX_NT]) + (q0[2 * TR_MAX_NT * TR_MAX_NT] + q0[3 * TR_MAX_NT * TR_MAX_NT]);
        }
        const double rp = (rpart[tt] + rpart[TR_MAX_NT + tt]) + (rpart[2 * TR_MAX_NT + tt] + rpart[3 * TR_MAX_NT + tt]);
        const bool live = is_trunk && row && !self_fixed;
#pragma unroll
        for (int u = 0; u < TR_NT; u++) {
          const int cd = cdof[TR_NL + u];
          const bool cfixed = cd < 0 || ((fixedm >> cd) & 1ull);           // wave-uniform
          double v = (live && !cfixed && u <= t) ? hv[u] + sp[u] : 0.0;
          if (is_trunk && u == t && !(live && !cfixed)) v = 1.0;
          s[u] = v;
        }
        bt = is_trunk ? (live ? rhs0 + rp : rhs0) : 0.0;
      }
      double tdinv = 1.0;
      double dq = TR_BCAST(s[0], TR_NL);
      double dinv = fast_rsqrt(dq);
#pragma unroll
      for (int q = 0; q < TR_NT; q++) {
        const int t = fresh_lane(lane0) - TR_NL;             // (t > q), (t == q): computed here, dead after this pivot
        tbad = tbad || !(dq > 0.0);
        const double ss = s[q] * dinv;
        double l = t > q ? ss : 0.0;
        s[q] = t == q ? ss : l;
        if (t == q) tdinv = dinv;
        double dinv_next = 1.0;
        if (q + 1 < TR_NT) {
          s[q + 1] = fma(-l, TR_BCAST(l, TR_NL + q + 1), s[q + 1]);
          dq = TR_BCAST(s[q + 1], TR_NL + q + 1);
          dinv_next = fast_rsqrt(dq);
        }
        const double yq = TR_BCAST(bt, TR_NL + q) * dinv;
        bt = fma(-l, yq, bt);
#pragma unroll
        for (int k = q + 2; k < TR_NT; k++) s[k] = fma(-l, TR_BCAST(l, TR_NL + k), s[k]);
        dinv = dinv_next;
      }
      // back substitution: L^T through this wavefront's scratch (columns 8..17 of rows 8..17 are free)
      double* Tscr = Lscr + TR_NL;
      double lt[TR_NT];
      {
        TR_ROW()
        if (is_trunk) {
#pragma unroll
          for (int u = 0; u < TR_NT; u++) Tscr[lane * TR_LD + u] = s[u];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#pragma unroll
        // row t of L^T without its diagonal (the rows below are zero, rows outside the trunk get zeros): row q is final
        // when its step comes, so no step needs a select
        for (int q = 0; q < TR_NT; q++) lt[q] = (is_trunk && t != q) ? Tscr[(TR_NL + q) * TR_LD + t] : 0.0;
      }
      bt *= tdinv;                                           // y (rows kept their unscaled right-hand side)
#pragma unroll
      for (int q = TR_NT - 1; q >= 0; q--) {
        const double xq = TR_BCAST(bt * tdinv, TR_NL + q);
        bt = fma(-lt[q], xq, bt);
      }
      bt *= tdinv;                                           // x
    }
    PROF_END(pr, PH_RATIO);
    PROF_BEGIN(pr);
    // ---- (5) limbs: y_l - Y_l^T x_T, then back substitution with L_l^T ----------------------------
    double x = bt;                                                                     // trunk lanes
    {
      double lt[TR_NL];
      double bb = b;                                                                   // y_l (limb lanes)
      {
        TR_ROW()
#pragma unroll
        for (int m = 0; m < TR_NL; m++) lt[m] = (is_limb && m != a) ? Lscr[m * TR_LD + a] : 0.0;   // column a of L_l, off-diagonal
#pragma unroll
        for (int u = 0; u < TR_NT; u++) {
          const double xt = TR_BCAST(bt, TR_NL + u);
          if (is_limb) bb = fma(-Lscr[(TR_NL + u) * TR_LD + a], xt, bb);               // Y_l[u][a]
        }
      }
#pragma unroll
      for (int p = TR_NL - 1; p >= 0; p--) {
        const double xp = TR_BCAST(bb * mydinv, p);
        bb = fma(-lt[p], xp, bb);                            // (rows >= p have lt[p] = 0: row p is final at its step)
      }
      if (fresh_lane(lane0) < TR_NL) x = bb * mydinv;
    }
    // ---- (6) violated bounds (free set) / multipliers (fixed set): g = H x + c ---------------------
    if (fixedm != 0ull) {                                     // multipliers need the whole x
      if (own) xs[dof] = x;
      if (dual_tol < 0.0) {                                   // (wave-uniform; most solves never fix a variable)
        const int li = fresh_lane(lane_in);
        dual_tol = 1e-13 * (1.0 + rows3_max(li < n ? fabs((sm + L.o.c)[li]) : 0.0));
      }
      TR_SYNC();                                                                             // B2
    }
    PROF_END(pr, PH_MULT);
    PROF_BEGIN(pr);
    int newst = 0;                                            // 0 none, 1 -> lower, 2 -> upper, 3 release
    if (own) {
      if (!self_fixed) {
        if (x < lo - ptol_lo) newst = 1;
        else if (x > hi + ptol_hi) newst = 2;
      } else {
        double g0 = ci, g1 = 0.0;
        // (eight products per trip with all their loads in flight: the stream that sets the batch's time sits on joint
        //  limits in nearly every frame, so this row product is on the headline's critical path)
        double g2 = 0.0, g3 = 0.0;
        int j = 0;
        for (; j + 7 < n; j += 8) {
          const double h0 = Hrow[j], h1 = Hrow[j + 1], h2 = Hrow[j + 2], h3 = Hrow[j + 3], h4 = Hrow[j + 4], h5 = Hrow[j + 5],
                       h6 = Hrow[j + 6], h7 = Hrow[j + 7];
          const double x0 = xs[j], x1 = xs[j + 1], x2 = xs[j + 2], x3 = xs[j + 3], x4 = xs[j + 4], x5 = xs[j + 5],
                       x6 = xs[j + 6], x7 = xs[j + 7];
          g0 = fma(h0, x0, g0); g1 = fma(h1, x1, g1); g2 = fma(h2, x2, g2); g3 = fma(h3, x3, g3);
          g0 = fma(h4, x4, g0); g1 = fma(h5, x5, g1); g2 = fma(h6, x6, g2); g3 = fma(h7, x7, g3);
        }
        for (; j < n; j++) g0 = fma(Hrow[j], xs[j], g0);
        const double g = (g0 + g1) + (g2 + g3);
        const bool at_lower = (bs.lower >> dof) & 1ull;
        if (at_lower ? g < -dual_tol : g > dual_tol) newst = 3;
      }
    }
    // each violating owner lane sets its dof's bit in the round's set (LDS atomic OR: order-independent)
    if (newst != 0) atomicOr(&vcur[newst - 1], 1ull << dof);
    if (fresh_lane(lane0) == 0 && tbad) atomicOr(&vcur[3], 1ull);
    // With no variable fixed nobody reads xs in this round (no multipliers): a lane without a violation stores its
    // result now, and if the round turns out to be the last one, B3 has already published it -- one barrier less per
    // solve on the common path.  (A violating lane's round is not the last; its store would be overwritten anyway.)
    const bool early = fixedm == 0ull;                        // wave- and workgroup-uniform
    if (early && own && newst == 0) xs[dof] = fmin(fmax(x, lo), hi);
    TR_SYNC();                                                                               // B3
    PROF_END(pr, PH_IO);
    const unsigned long long to_lo = vcur[0], to_up = vcur[1], rel = vcur[2];
    if (vcur[3]) return GMR_STATUS_QP_FAILED;
    const unsigned long long all = to_lo | to_up | rel;
    if (all == 0ull) {
      if (!early) {
        if (own) xs[dof] = fmin(fmax(x, lo), hi);
        TR_SYNC();
      }
      return GMR_STATUS_OK;
    }
    const int total = __popcll(all);
    unsigned long long sel = all;                             // block principal pivoting: exchange all
    if (total < ninf_best) { ninf_best = total; pcount = 3; }
    else if (pcount > 0) pcount--;
    else sel = 1ull << (63 - __clzll((long long)all));        // Murty: only the highest violated variable
    bs.lower = (bs.lower & ~(rel & sel)) | (to_lo & sel);
    bs.upper = (bs.upper & ~(rel & sel)) | (to_up & sel);
  }
  return GMR_STATUS_QP_MAXITER;
#undef TR_BCAST
#undef TR_SYNC
#undef TR_ROW
}

}  // namespace gmr
