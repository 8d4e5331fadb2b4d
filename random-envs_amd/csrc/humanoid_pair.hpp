// humanoid_pair.hpp -- the humanoid's forward dynamics over TWO LANES PER ENVIRONMENT (lanes 2e and 2e + 1 of a wave hold
// env e).  humanoid.xml is a trunk (torso, lwaist, pelvis: the free root + 3 abdomen hinges = dofs 0..8) with a right and a
// left side hanging off it (leg: thigh / shin / foot, 4 hinges; arm: upper / lower arm, 3 hinges).  Lane 0 owns the right
// side, lane 1 the left side; the trunk is replicated bit for bit in both.  Each lane therefore works on a LOCAL tree of
// 8 bodies and 16 dofs instead of 13 and 23:
//   * kinematics, com, RNE, CRB run over the local tree; whatever a side contributes to the trunk (subtree forces, composite
//     inertias, the Schur complement of the L^T D L factorisation, the back-substitution of a right-hand side) is summed
//     across the pair with one DPP exchange per value (lane ^ 1, no LDS).  a + b is commutative, so both lanes hold
//     identical trunk bits afterwards and the replicated trunk never drifts apart.
//   * the packed tree-sparse mass matrix of the local tree has 115 entries (trunk block 45 + own side 70) instead of 185;
//     constraint rows are stored as 16 local columns per lane.
//   * model constants of the side bodies are the SAME code with per-lane selects between the right and the left constant
//     (the model is not exactly mirror symmetric: right_hip_y armature .008 vs .01, left_knee stiffness 1, humanoid.xml:46,59,62).
//   * collision: both lanes hold the same candidate mask; the narrow phase takes candidates two at a time (lane 0 the
//     first, lane 1 the second), hits are queued in table order in the env's LDS column (shared by the pair); every lane
//     builds its own 16 columns of every row.
//   * the dual matrix A = J M^-1 J^T + R sits in the env's LDS column as before; row dot products are split over the pair.
// Row order, the PGS sweep sequence and every formula are those of humanoid_engine.hpp (which the reset / forward kernels
// still use, one env per lane): results agree to rounding, and the same oracle tests gate both.
//
// `P` is the lane policy: side(), xchg(), any(), col(), sync().  On the device these are threadIdx.x & 1, a DPP quad
// permute, a wave ballot, the env's LDS column and a wave-level fence; the test harness runs the two lanes of a pair as
// two host threads in lock step (tests/host_harness/humanoid_pair_host.cpp).
#pragma once
#include "humanoid_engine.hpp"

#if defined(REX_KTIME) && defined(__HIP_DEVICE_COMPILE__)
namespace rex { extern __device__ unsigned long long g_ktime[24 + 72]; }   // [24 + lvl]: wave-evaluations per sweep level, [40 + lvl]: their sweep cycles
#endif
namespace rex {
namespace hum {
namespace pr {

constexpr int LB = 8;    // local bodies: 0 torso, 1 lwaist, 2 pelvis | 3 thigh, 4 shin, 5 foot, 6 upper arm, 7 lower arm
constexpr int LD = 16;   // local dofs: 0..8 trunk | 9..12 hip x, z, y, knee | 13..15 shoulder 1, 2, elbow
constexpr int LQ = 17;   // local qpos: free joint 7, then hinge of local dof d at d + 1
constexpr int LU = 10;   // local controls: the hinges 6..15
constexpr int LG = 11;   // local geoms: torso1, head, uwaist, lwaist, butt | thigh, shin, foot, upper arm, lower arm, hand
constexpr int kLParent[LD] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 5, 13, 14};
constexpr int kLDofBody[LD] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 3, 3, 3, 4, 6, 6, 7};
constexpr int kLBodyParent[LB] = {-1, 0, 1, 2, 3, 4, 0, 6};
constexpr int kLBodyDofAdr[LB] = {0, 6, 8, 9, 12, 0, 13, 15};
constexpr int kLBodyDofNum[LB] = {6, 2, 1, 3, 1, 0, 2, 1};
constexpr int kLGeomBody[LG] = {0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 7};
// local -> global index of the right (lane 0) and the left (lane 1) side
constexpr int gbR(int lb) { return lb < 3 ? lb + 1 : (lb < 6 ? 4 + (lb - 3) : 10 + (lb - 6)); }
constexpr int gbL(int lb) { return lb < 3 ? lb + 1 : (lb < 6 ? 7 + (lb - 3) : 12 + (lb - 6)); }
constexpr int gdR(int ld) { return ld < 9 ? ld : (ld < 13 ? 9 + (ld - 9) : 17 + (ld - 13)); }
constexpr int gdL(int ld) { return ld < 9 ? ld : (ld < 13 ? 13 + (ld - 9) : 20 + (ld - 13)); }
constexpr int ggR(int lg) { return lg < 5 ? lg + 1 : (lg < 8 ? 6 + (lg - 5) : 12 + (lg - 8)); }
constexpr int ggL(int lg) { return lg < 5 ? lg + 1 : (lg < 8 ? 9 + (lg - 5) : 15 + (lg - 8)); }
constexpr bool tables_ok() {
  for (int ld = 0; ld < LD; ld++) {
    for (int s = 0; s < 2; s++) {
      const int g = s ? gdL(ld) : gdR(ld), pl = kLParent[ld], pg = pl < 0 ? -1 : (s ? gdL(pl) : gdR(pl));
      if (kDofParent[g] != pg) return false;
      if (kDofBody[g] != (s ? gbL(kLDofBody[ld]) : gbR(kLDofBody[ld]))) return false;
    }
  }
  for (int lb = 0; lb < LB; lb++) for (int s = 0; s < 2; s++) {
    const int g = s ? gbL(lb) : gbR(lb), pl = kLBodyParent[lb], pg = pl < 0 ? 0 : (s ? gbL(pl) : gbR(pl));
    if (kBodyParent[g] != pg || kBodyDofNum[g] != kLBodyDofNum[lb]) return false;
    if (kLBodyDofNum[lb] && kBodyDofAdr[g] != (s ? gdL(kLBodyDofAdr[lb]) : gdR(kLBodyDofAdr[lb]))) return false;
  }
  for (int lg = 0; lg < LG; lg++) for (int s = 0; s < 2; s++) {
    const int g = s ? ggL(lg) : ggR(lg);
    if (kGeomBody[g] != (s ? gbL(kLGeomBody[lg]) : gbR(kLGeomBody[lg]))) return false;
    if (lg >= 5 && ggL(lg) != ggR(lg) + 3) return false;
  }
  return true;
}
static_assert(tables_ok(), "local (per-side) tree tables disagree with the global tree of humanoid_engine.hpp");

// packed tree-sparse matrix of the LOCAL dof tree
constexpr int ldepth(int d) { int n = 0; while (kLParent[d] >= 0) { d = kLParent[d]; n++; } return n; }
constexpr int lrow(int i) { int o = 0; for (int k = 0; k < i; k++) o += ldepth(k) + 1; return o; }
constexpr int LNNZ = lrow(LD);                       // 115 = 45 (trunk block) + 46 (leg rows) + 24 (arm rows)
constexpr int lidx(int i, int j) { return lrow(i) + ldepth(j); }   // valid when j is an ancestor-or-self of i
constexpr int TNNZ = lrow(9);                        // 45: the trunk block is the head of the packed array
static_assert(LNNZ == 115 && TNNZ == 45, "packed local mass matrix");
template <int I, class F> REX_HD void lfor_anc(F&& f) { if constexpr (kLParent[I] >= 0) { f(IC<kLParent[I]>{}); lfor_anc<kLParent[I]>(f); } }
template <int I, class F> REX_HD void lfor_anc_self(F&& f) { f(IC<I>{}); lfor_anc<I>(f); }

template <class T> REX_HD T sel(bool left, T r, T l) { return left ? l : r; }

// bits of a GLOBAL dof mask (PairRec::mask1 / mask2) that fall on this lane's 16 local dofs
REX_HD int local_mask(int mask, bool left) {
  return (mask & 0x1FF) | (((mask >> (left ? 13 : 9)) & 0xF) << 9) | (((mask >> (left ? 20 : 17)) & 0x7) << 13);
}

template <class T> struct PLane { T mass[LB]; T damping[LD]; };   // the randomised part of the model, local view
template <class T> struct PFactor { T a[LNNZ]; };

template <class T>
struct PSmooth {
  T xmat[LB][9], xipos[LB][3];
  T an[LD][3], ax[LD][3];          // joint anchors / axes of the local dofs (registers: compile-time indices only)
  T com[3];
  T cinert[LB][10], cvel[LB][6], cdof[LD][6];
};

template <class T>
struct PKin {
  T qfrc_smooth[LD], qacc_smooth[LD];
  int ncon, nefc, overflow;
#if defined(REX_KTIME)
  unsigned long long tacc[HT_SLOTS];
#endif
};

// observation inputs of the LAST evaluation (random_humanoid.py:193-204), local view
template <class T> struct PObs { T cinert[LB][10], cvel[LB][6], xipos_x[LB], act[LD]; };

// runtime-indexed per-lane arrays (HIP scratch): constraint rows as 16 local columns.  Both lanes of a pair hold the same
// row count; R / aref / force are replicated.
template <class T>
struct PScratch {
  T J[MAXEFC][LD], MiJ[MAXEFC][LD], R[MAXEFC], aref[MAXEFC], Adiag[MAXEFC], force[MAXEFC];
};

// the env's LDS column: geom poses + hit queue during the collision phase, then the dual PGS working set
constexpr int PGEO = 0;                                  // (g - 1) * 6 + {pos 0..2, axis 3..5}, g = 1..17
constexpr int PHITQ = (NGEOM - 1) * 6, PHITQ_WORDS = 8, PHITQ_MAX = 8;
constexpr int PAIR_WORDS = DUAL_WORDS;                   // 292 words per env; 4 blocks of 32 envs fill a CU's 160 KB
static_assert(PHITQ + PHITQ_WORDS * PHITQ_MAX <= PAIR_WORDS, "geom poses + hit queue must fit the LDS column");

// sum of a pair-partial: own + partner's (identical bits in both lanes)
template <class T, class P> REX_HD T psum(const P& p, T x) { return x + p.xchg(x); }

// dot product over the local dofs of two vectors whose trunk part is replicated: trunk + (own side + partner's side)
template <class T, class P>
REX_HD T pdot(const P& p, const T (&a)[LD], const T (&b)[LD]) {
  T tr = 0, sd = 0;
  static_for<0, 9>([&](auto KK) { constexpr int k = KK; tr += a[k] * b[k]; });
  static_for<9, LD>([&](auto KK) { constexpr int k = KK; sd += a[k] * b[k]; });
  return tr + psum(p, sd);
}

// ---- kinematics ([3P] mj_kinematics over the local tree) -----------------------------------------------------------------
template <class T, class P>
REX_HD void kinematics(const P& p, const Model<T>& m, const T (&ql)[LQ], PSmooth<T>& K) {
  const bool left = p.side() != 0;
  T* const col = p.col();
  T xp[LB][3], xq[LB][4];
  static_for<0, LB>([&](auto BB) {
    constexpr int lb = BB, par = kLBodyParent[lb], bR = gbR(lb), bL = gbL(lb);
    T xpos[3], xquat[4], t[3], R[9];
    if constexpr (lb == 0) {   // free joint of the torso
      for (int k = 0; k < 3; k++) xpos[k] = ql[k];
      for (int k = 0; k < 4; k++) xquat[k] = ql[3 + k];
      qnorm(xquat); q2mat(R, xquat);
      for (int k = 0; k < 3; k++) for (int x = 0; x < 3; x++) {
        K.an[k][x] = xpos[x]; K.ax[k][x] = (x == k) ? T(1) : T(0);
        K.an[3 + k][x] = xpos[x]; K.ax[3 + k][x] = R[3 * x + k];
      }
    } else {
      T bpos[3];
      for (int k = 0; k < 3; k++) bpos[k] = lb < 3 ? m.body_pos[bR][k] : sel(left, m.body_pos[bR][k], m.body_pos[bL][k]);
      mulv(t, K.xmat[par], bpos);
      for (int k = 0; k < 3; k++) xpos[k] = xp[par][k] + t[k];
      if constexpr (lb < 3) qmul(xquat, xq[par], m.body_quat[bR]);
      else for (int k = 0; k < 4; k++) xquat[k] = xq[par][k];   // the side bodies have no orientation offset (check_pair_model)
      static_for<0, kLBodyDofNum[lb]>([&](auto JJ) {
        constexpr int ld = kLBodyDofAdr[lb] + JJ, jR = gdR(ld) - 5, jL = gdL(ld) - 5;
        T jpos[3], jax[3];
        for (int k = 0; k < 3; k++) { jpos[k] = sel(left, m.jnt_pos[jR][k], m.jnt_pos[jL][k]); jax[k] = sel(left, m.jnt_axis[jR][k], m.jnt_axis[jL][k]); }
        q2mat(R, xquat);
        T anchor[3], axis[3];
        mulv(t, R, jpos); for (int k = 0; k < 3; k++) anchor[k] = xpos[k] + t[k];
        mulv(axis, R, jax);
        T sn, cs; hsincos(T(0.5) * ql[ld + 1], sn, cs);           // qpos0 of every hinge is 0
        T qj[4] = {cs, jax[0] * sn, jax[1] * sn, jax[2] * sn}, nq[4];
        qmul(nq, xquat, qj); for (int k = 0; k < 4; k++) xquat[k] = nq[k];
        qnorm(xquat); q2mat(R, xquat);
        mulv(t, R, jpos); for (int k = 0; k < 3; k++) xpos[k] = anchor[k] - t[k];
        for (int k = 0; k < 3; k++) { K.an[ld][k] = anchor[k]; K.ax[ld][k] = axis[k]; }
      });
    }
    for (int k = 0; k < 3; k++) xp[lb][k] = xpos[k];
    for (int k = 0; k < 4; k++) xq[lb][k] = xquat[k];
    q2mat(K.xmat[lb], xquat);
    T ip[3];
    for (int k = 0; k < 3; k++) ip[k] = lb < 3 ? m.body_ipos[bR][k] : sel(left, m.body_ipos[bR][k], m.body_ipos[bL][k]);
    mulv(t, K.xmat[lb], ip); for (int k = 0; k < 3; k++) K.xipos[lb][k] = xpos[k] + t[k];
  });
  // geom poses into the env's LDS column: the trunk geoms by the right lane, the side geoms by their owner
  static_for<0, LG>([&](auto GG) {
    constexpr int lg = GG, lb = kLGeomBody[lg], gR = ggR(lg), gL = ggL(lg);
    T gp[3], ga[3], t[3], a[3];
    for (int k = 0; k < 3; k++) { gp[k] = lg < 5 ? m.geom_pos[gR][k] : sel(left, m.geom_pos[gR][k], m.geom_pos[gL][k]);
                                  ga[k] = lg < 5 ? m.geom_axis[gR][k] : sel(left, m.geom_axis[gR][k], m.geom_axis[gL][k]); }
    mulv(t, K.xmat[lb], gp); mulv(a, K.xmat[lb], ga);
    if constexpr (lg < 5) {
      if (!left) for (int k = 0; k < 3; k++) { col[PGEO + (gR - 1) * 6 + k] = xp[lb][k] + t[k]; col[PGEO + (gR - 1) * 6 + 3 + k] = a[k]; }
    } else {
      T* const g = col + PGEO + (gR - 1 + (left ? 3 : 0)) * 6;     // ggL = ggR + 3
      for (int k = 0; k < 3; k++) { g[k] = xp[lb][k] + t[k]; g[3 + k] = a[k]; }
    }
  });
  p.sync();
}

// ---- [3P] mj_comPos: reference point, cinert, cdof ----------------------------------------------------------------------
template <class T, class P>
REX_HD void com_pos(const P& p, const Model<T>& m, const PLane<T>& L, PSmooth<T>& K) {
  const bool left = p.side() != 0;
  T sc[3] = {0, 0, 0};
  static_for<0, LB>([&](auto BB) { constexpr int lb = BB;
    const T ms = (lb < 3 && left) ? T(0) : L.mass[lb];             // the trunk counts once
    for (int k = 0; k < 3; k++) sc[k] += ms * K.xipos[lb][k]; });
  for (int k = 0; k < 3; k++) K.com[k] = psum(p, sc[k]) / m.subtreemass_root;   // compile-time subtree mass (Q4-style staleness)
  static_for<0, LB>([&](auto BB) {
    constexpr int lb = BB, bR = gbR(lb), bL = gbL(lb);
    const T* R = K.xmat[lb];
    T I[6];
    for (int k = 0; k < 6; k++) I[k] = lb < 3 ? m.body_inertia[bR][k] : sel(left, m.body_inertia[bR][k], m.body_inertia[bL][k]);
    T Ib[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}, RI[9], Iw[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T a = 0; for (int k = 0; k < 3; k++) a += R[3 * i + k] * Ib[3 * k + j]; RI[3 * i + j] = a; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T a = 0; for (int k = 0; k < 3; k++) a += RI[3 * i + k] * R[3 * j + k]; Iw[3 * i + j] = a; }
    T d[3] = {K.xipos[lb][0] - K.com[0], K.xipos[lb][1] - K.com[1], K.xipos[lb][2] - K.com[2]}, ms = L.mass[lb];
    T* c = K.cinert[lb];
    c[0] = Iw[0] + ms * (d[1] * d[1] + d[2] * d[2]); c[1] = Iw[4] + ms * (d[0] * d[0] + d[2] * d[2]); c[2] = Iw[8] + ms * (d[0] * d[0] + d[1] * d[1]);
    c[3] = Iw[1] - ms * d[0] * d[1]; c[4] = Iw[2] - ms * d[0] * d[2]; c[5] = Iw[5] - ms * d[1] * d[2];
    c[6] = ms * d[0]; c[7] = ms * d[1]; c[8] = ms * d[2]; c[9] = ms;
  });
  static_for<0, LD>([&](auto II) {
    constexpr int i = II;
    if constexpr (i < 3) { for (int k = 0; k < 3; k++) { K.cdof[i][k] = 0; K.cdof[i][3 + k] = K.ax[i][k]; } }
    else {
      T off[3] = {K.com[0] - K.an[i][0], K.com[1] - K.an[i][1], K.com[2] - K.an[i][2]}, t[3];
      cross3(t, K.ax[i], off);
      for (int k = 0; k < 3; k++) { K.cdof[i][k] = K.ax[i][k]; K.cdof[i][3 + k] = t[k]; }
    }
  });
}

// ---- [3P] mj_comVel + mj_rne (flg_acc = 0) ---------------------------------------------------------------------------------
template <class T, class P>
REX_HD void com_vel_rne(const P& p, const Model<T>& m, const T (&vl)[LD], PSmooth<T>& K, T (&qfrc_bias)[LD]) {
  T cacc[LB][6], cfrc[LB][6], cdofdot[LD][6];
  static_for<0, LB>([&](auto BB) {
    constexpr int lb = BB, par = kLBodyParent[lb], da = kLBodyDofAdr[lb], nd = kLBodyDofNum[lb];
    T v[6], a[6];
    if constexpr (lb == 0) {   // parent = world: zero velocity, acceleration -gravity (the world accelerates upwards)
      for (int k = 0; k < 6; k++) { v[k] = 0; a[k] = 0; }
      a[5] = m.gravity;
      for (int i = 0; i < 3; i++) { for (int k = 0; k < 6; k++) { cdofdot[i][k] = 0; v[k] += K.cdof[i][k] * vl[i]; } }
      for (int i = 3; i < 6; i++) cross_motion(cdofdot[i], v, K.cdof[i]);
      for (int i = 3; i < 6; i++) for (int k = 0; k < 6; k++) v[k] += K.cdof[i][k] * vl[i];
    } else {
      for (int k = 0; k < 6; k++) { v[k] = K.cvel[par][k]; a[k] = cacc[par][k]; }
      static_for<0, nd>([&](auto JJ) { constexpr int i = da + JJ; cross_motion(cdofdot[i], v, K.cdof[i]); for (int k = 0; k < 6; k++) v[k] += K.cdof[i][k] * vl[i]; });
    }
    static_for<0, nd>([&](auto JJ) { constexpr int i = da + JJ; for (int k = 0; k < 6; k++) a[k] += cdofdot[i][k] * vl[i]; });
    for (int k = 0; k < 6; k++) { K.cvel[lb][k] = v[k]; cacc[lb][k] = a[k]; }
    T Ia[6], Iv[6], t[6];
    mul_inert(Ia, K.cinert[lb], a); mul_inert(Iv, K.cinert[lb], v); cross_force(t, v, Iv);
    for (int k = 0; k < 6; k++) cfrc[lb][k] = Ia[k] + t[k];
  });
  // backward pass: the side chains, then what BOTH sides hand to the pelvis (legs) and the torso (arms), then the trunk
  for (int k = 0; k < 6; k++) { cfrc[4][k] += cfrc[5][k]; cfrc[3][k] += cfrc[4][k]; cfrc[6][k] += cfrc[7][k]; }
  for (int k = 0; k < 6; k++) {
    const T legs = psum(p, cfrc[3][k]), arms = psum(p, cfrc[6][k]);
    cfrc[2][k] += legs; cfrc[1][k] += cfrc[2][k]; cfrc[0][k] += cfrc[1][k] + arms;
  }
  static_for<0, LD>([&](auto II) { constexpr int i = II; T a = 0; for (int k = 0; k < 6; k++) a += K.cdof[i][k] * cfrc[kLDofBody[i]][k]; qfrc_bias[i] = a; });
}

// ---- [3P] mj_crb -> packed local M ---------------------------------------------------------------------------------------
template <class T, class P>
REX_HD void crb(const P& p, const Model<T>& m, const PSmooth<T>& K, PFactor<T>& F) {
  const bool left = p.side() != 0;
  T c[LB][10];
  static_for<0, LB>([&](auto BB) { constexpr int lb = BB; for (int k = 0; k < 10; k++) c[lb][k] = K.cinert[lb][k]; });
  for (int k = 0; k < 10; k++) { c[4][k] += c[5][k]; c[3][k] += c[4][k]; c[6][k] += c[7][k]; }
  for (int k = 0; k < 10; k++) {
    const T legs = psum(p, c[3][k]), arms = psum(p, c[6][k]);
    c[2][k] += legs; c[1][k] += c[2][k]; c[0][k] += c[1][k] + arms;
  }
  static_for<0, LD>([&](auto II) {
    constexpr int i = II;
    T buf[6]; mul_inert(buf, c[kLDofBody[i]], K.cdof[i]);
    lfor_anc_self<i>([&](auto JJ) {
      constexpr int j = JJ;
      T a = 0; for (int k = 0; k < 6; k++) a += K.cdof[j][k] * buf[k];
      if constexpr (i == j) a += i < 9 ? m.dof_armature[i] : sel(left, m.dof_armature[gdR(i)], m.dof_armature[gdL(i)]);
      F.a[lidx(i, j)] = a;
    });
  });
}

// ---- sparse L^T D L of the local tree ([3P] mj_factorM): the side dofs first, each lane its own; their Schur complement on the
// trunk block is summed over the pair; then the trunk, replicated -------------------------------------------------------------
template <class T, class P>
REX_HD void factor(const P& p, PFactor<T>& F) {
  T acc[TNNZ];
  static_for<0, TNNZ>([&](auto KK) { acc[KK] = T(0); });
  static_rfor<9, LD>([&](auto KK) {
    constexpr int k = KK;
    const T inv = rcp_t(F.a[lidx(k, k)]);
    lfor_anc<k>([&](auto II) {
      constexpr int i = II;
      const T a = F.a[lidx(k, i)] * inv;
      lfor_anc_self<i>([&](auto JJ) { constexpr int j = JJ;
        if constexpr (i < 9) acc[lidx(i, j)] -= a * F.a[lidx(k, j)];     // trunk block: accumulated apart, combined below
        else F.a[lidx(i, j)] -= a * F.a[lidx(k, j)]; });
      F.a[lidx(k, i)] = a;
    });
    F.a[lidx(k, k)] = inv;
  });
  static_for<0, TNNZ>([&](auto KK) { constexpr int k = KK; F.a[k] += psum(p, acc[k]); });
  static_rfor<0, 9>([&](auto KK) {
    constexpr int k = KK;
    const T inv = rcp_t(F.a[lidx(k, k)]);
    lfor_anc<k>([&](auto II) {
      constexpr int i = II;
      const T a = F.a[lidx(k, i)] * inv;
      lfor_anc_self<i>([&](auto JJ) { constexpr int j = JJ; F.a[lidx(i, j)] -= a * F.a[lidx(k, j)]; });
      F.a[lidx(k, i)] = a;
    });
    F.a[lidx(k, k)] = inv;
  });
}
// x <- L^-T x ([3P] mj_solveM2's half): the side dofs push into the trunk entries; the pushes of both sides are summed
template <int NR, class T, class P>
REX_HD void solve_back(const P& p, const PFactor<T>& F, T (&x)[NR][LD]) {
  T acc[NR][9];
  for (int r = 0; r < NR; r++) for (int k = 0; k < 9; k++) acc[r][k] = T(0);
  static_rfor<9, LD>([&](auto KK) { constexpr int k = KK; lfor_anc<k>([&](auto II) { constexpr int i = II;
    const T l = F.a[lidx(k, i)];
    for (int r = 0; r < NR; r++) { if constexpr (i < 9) acc[r][i] -= l * x[r][k]; else x[r][i] -= l * x[r][k]; } }); });
  for (int r = 0; r < NR; r++) static_for<0, 9>([&](auto KK) { constexpr int k = KK; x[r][k] += psum(p, acc[r][k]); });
  static_rfor<0, 9>([&](auto KK) { constexpr int k = KK; lfor_anc<k>([&](auto II) { constexpr int i = II;
    const T l = F.a[lidx(k, i)];
    for (int r = 0; r < NR; r++) x[r][i] -= l * x[r][k]; }); });
}
// x <- L^-1 x: the trunk first (replicated), the side rows read it -- no exchange
template <class T>
REX_HD void solve_fwd(const PFactor<T>& F, T (&x)[LD]) {
  static_for<0, LD>([&](auto KK) { constexpr int k = KK; lfor_anc<k>([&](auto II) { constexpr int i = II; x[k] -= F.a[lidx(k, i)] * x[i]; }); });
}
template <class T, class P>
REX_HD void solve(const P& p, const PFactor<T>& F, T (&x)[LD]) {
  T xr[1][LD];
  static_for<0, LD>([&](auto KK) { xr[0][KK] = x[KK]; });
  solve_back<1>(p, F, xr);
  static_for<0, LD>([&](auto KK) { constexpr int k = KK; x[k] = xr[0][k] * F.a[lidx(k, k)]; });
  solve_fwd(F, x);
}

// ---- Jacobian columns of a world point on the local dofs: out[d][i] = dir_d . (sign_i * axis_i x (p - anchor_i)) --------------
template <int NDIR, class T, class P>
REX_HD void jac_dirs(const P& p, const PSmooth<T>& K, int lm1, int lm2, const T* pt, const T* dirs, T (&out)[NDIR][LD]) {
  const int diff = lm1 ^ lm2;
  static_for<0, 3>([&](auto II) {
    constexpr int i = II;
    const T sign = T(((lm2 >> i) & 1) - ((lm1 >> i) & 1));
    for (int d = 0; d < NDIR; d++) out[d][i] = sign * dirs[3 * d + i];
  });
  auto limb = [&](auto LO, auto HI) {
    constexpr int lo = LO, hi = HI;
    for (int d = 0; d < NDIR; d++) for (int i = lo; i < hi; i++) out[d][i] = 0;
    if (p.any((diff & (((1 << (hi - lo)) - 1) << lo)) != 0)) {
      static_for<lo, hi>([&](auto II) {
        constexpr int i = II;
        const T sign = T(((lm2 >> i) & 1) - ((lm1 >> i) & 1));
        T r[3] = {pt[0] - K.an[i][0], pt[1] - K.an[i][1], pt[2] - K.an[i][2]}, c[3];
        cross3(c, K.ax[i], r);
        for (int d = 0; d < NDIR; d++) out[d][i] = sign * dot3(dirs + 3 * d, c);
      });
    }
  };
  limb(IC<3>{}, IC<6>{}); limb(IC<6>{}, IC<9>{}); limb(IC<9>{}, IC<13>{}); limb(IC<13>{}, IC<LD>{});
}

// ---- hinge-limit rows ([3P] mj_instantiateLimit), MuJoCo's joint order: abdomen z, y, x | right leg | left leg | right arm | left arm
template <class T, class P>
REX_HD void limit_rows(const P& p, const Model<T>& m, const T (&ql)[LQ], const T (&vl)[LD], PKin<T>& K, PScratch<T>& s) {
  const bool left = p.side() != 0;
  // every lane tests the trunk hinges (replicated) and its own 7; R / aref of the partner's active rows come over the pair
  T Rr[10], ar[10], sg[10]; unsigned act = 0;
  static_for<6, LD>([&](auto DD) {
    constexpr int ld = DD, jR = gdR(ld) - 5, jL = gdL(ld) - 5, k = ld - 6;
    const T lo = ld < 9 ? m.jnt_lo[jR] : sel(left, m.jnt_lo[jR], m.jnt_lo[jL]), hi = ld < 9 ? m.jnt_hi[jR] : sel(left, m.jnt_hi[jR], m.jnt_hi[jL]);
    const T iw = ld < 9 ? m.dof_invw[gdR(ld)] : sel(left, m.dof_invw[gdR(ld)], m.dof_invw[gdL(ld)]);
    const T val = ql[ld + 1], dlo = val - lo, dhi = hi - val;
    const bool low = dlo < T(0);
    const T dist = low ? dlo : dhi;
    sg[k] = low ? T(1) : T(-1);
    const T vel = sg[k] * vl[ld];
    const T imp = impedance3(m, habs(dist));
    Rr[k] = hmax(T(1e-15), (T(1) - imp) * iw * rcp_t(imp));
    ar[k] = -m.B * vel - m.K * imp * dist;
    act |= dist < T(0) ? (1u << k) : 0u;
  });
  const unsigned oact = p.xchg(act);
  T oR[7], oa[7];
  static_for<0, 7>([&](auto KK) { constexpr int k = KK; oR[k] = p.xchg(Rr[3 + k]); oa[k] = p.xchg(ar[3 + k]); });
  int ne = 0;
  auto emit = [&](bool on, int own_col, T sign, T Rv, T av) {     // own_col < 0: the row lives on the partner's side (zeros here)
    if (on && ne < MAXEFC) {
      for (int c = 0; c < LD; c++) s.J[ne][c] = (c == own_col) ? sign : T(0);
      s.R[ne] = Rv; s.aref[ne] = av; ne++;
    }
  };
  static_for<0, 3>([&](auto KK) { constexpr int k = KK; emit((act >> k) & 1u, 6 + k, sg[k], Rr[k], ar[k]); });
  auto side_rows = [&](auto LO, auto N, bool rows_are_left) {
    constexpr int lo = LO, n = N;     // local side dofs lo .. lo + n - 1 (k = ld - 6)
    const bool mine = rows_are_left == left;
    static_for<0, n>([&](auto JJ) {
      constexpr int ld = lo + JJ, k = ld - 6;
      const bool on = mine ? ((act >> k) & 1u) : ((oact >> k) & 1u);
      emit(on, mine ? ld : -1, sg[k], mine ? Rr[k] : oR[k - 3], mine ? ar[k] : oa[k - 3]);
    });
  };
  side_rows(IC<9>{}, IC<4>{}, false); side_rows(IC<9>{}, IC<4>{}, true);      // right leg, left leg
  side_rows(IC<13>{}, IC<3>{}, false); side_rows(IC<13>{}, IC<3>{}, true);    // right arm, left arm
  K.nefc = ne;
}

// ---- a detected contact -> its constraint rows (16 local columns per lane) ------------------------------------------------------
template <class T, class P>
REX_HD void add_contact(const P& p, PKin<T>& K, PScratch<T>& s, const PSmooth<T>& S, const Model<T>& m, const T (&vl)[LD], const PairRec<T>& pr,
                        T dist, const T* pos, const T* normal, const T* yaxis) {
  if (K.ncon >= MAXCON) { K.overflow = 1; return; }
  K.ncon++;
  const bool left = p.side() != 0;
  T f[9];
  for (int k = 0; k < 3; k++) { f[k] = normal[k]; f[3 + k] = yaxis ? yaxis[k] : T(0); f[6 + k] = 0; }
  make_frame(f);
  if (!(dist < m.margin)) return;
  int ne = K.nefc;
  const T tran = pr.tran, mu = pr.mu;
  const T imp = impedance3(m, habs(dist - m.margin)), kterm = m.K * imp * (dist - m.margin);
  const int lm1 = local_mask(pr.mask1, left), lm2 = local_mask(pr.mask2, left);
  auto rowvel = [&](const T (&row)[LD]) {    // J qvel: the trunk part is replicated, the side parts are summed over the pair
    T tr = 0, sd = 0;
    static_for<0, 9>([&](auto KK) { constexpr int k = KK; tr += row[k] * vl[k]; });
    static_for<9, LD>([&](auto KK) { constexpr int k = KK; sd += row[k] * vl[k]; });
    return tr + psum(p, sd);
  };
  if (pr.dim == 1) {
    if (ne >= MAXEFC) { K.overflow = 1; return; }
    T jn[1][LD];
    jac_dirs<1>(p, S, lm1, lm2, pos, f, jn);
    for (int k = 0; k < LD; k++) s.J[ne][k] = jn[0][k];
    const T vel = rowvel(jn[0]);
    s.R[ne] = hmax(T(1e-15), (T(1) - imp) * tran * rcp_t(imp));
    s.aref[ne] = -m.B * vel - kterm;
    ne++;
  } else {
    if (ne + 4 > MAXEFC) { K.overflow = 1; return; }
    T jf[3][LD];
    jac_dirs<3>(p, S, lm1, lm2, pos, f, jf);
    const T R1 = hmax(T(1e-15), (T(1) - imp) * (tran + mu * mu * tran) * rcp_t(imp)), Rpy = T(2) * mu * mu * R1;
    for (int t = 1; t <= 2; t++) {
      for (int sg = 1; sg >= -1; sg -= 2) {
        T row[LD];
        for (int k = 0; k < LD; k++) { row[k] = jf[0][k] + sg * mu * jf[t][k]; s.J[ne][k] = row[k]; }
        const T vel = rowvel(row);
        s.R[ne] = Rpy; s.aref[ne] = -m.B * vel - kterm;
        ne++;
      }
    }
  }
  K.nefc = ne;
}

// narrow phase of one candidate pair: humanoid_engine.hpp::collide_pair reading the geom poses of the env's LDS column
template <class T>
REX_HD void collide_pair_col(const Model<T>& m, const T* col, const PairRec<T>& pr, Hits<T>& h) {
  const int t1 = pr.t1, t2 = pr.t2;
  h.n = 0;
  for (int k = 0; k < 2; k++) { h.dist[k] = 0; for (int x = 0; x < 3; x++) { h.pos[k][x] = 0; h.normal[k][x] = 0; } }
  T p1[3], a1[3], p2[3], a2[3];
  for (int k = 0; k < 3; k++) { p2[k] = col[PGEO + (pr.g2 - 1) * 6 + k]; a2[k] = col[PGEO + (pr.g2 - 1) * 6 + 3 + k]; }
  const T r2 = pr.r2, l2 = pr.l2;
  if (t1 == G_PLANE) {
    if (t2 == G_SPHERE) plane_sphere(h, m, p2, r2);
    else { T c[3]; for (int sg = 1; sg >= -1; sg -= 2) { for (int k = 0; k < 3; k++) c[k] = p2[k] + a2[k] * (sg * l2); plane_sphere(h, m, c, r2); } }
    return;
  }
  for (int k = 0; k < 3; k++) { p1[k] = col[PGEO + (pr.g1 - 1) * 6 + k]; a1[k] = col[PGEO + (pr.g1 - 1) * 6 + 3 + k]; }
  const T r1 = pr.r1, l1 = pr.l1;
  T d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  T c1[3] = {p1[0], p1[1], p1[2]}, c2[3] = {p2[0], p2[1], p2[2]};
  bool single = true;
  if (t1 == G_SPHERE && t2 == G_CAPSULE) {
    T x = -(d[0] * a2[0] + d[1] * a2[1] + d[2] * a2[2]);
    x = hmin(hmax(x, -l2), l2);
    for (int k = 0; k < 3; k++) c2[k] = p2[k] + a2[k] * x;
  } else if (t1 == G_CAPSULE) {
    T dif[3] = {-d[0], -d[1], -d[2]};
    T ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif), det = ma * mc - mb * mb;
    if (habs(det) >= T(1e-15)) {
      const T idet = rcp_t(det), imc = rcp_t(mc), ima = rcp_t(ma);
      T x1 = (mc * u - mb * v) * idet, x2 = (ma * v - mb * u) * idet;
      if (x1 > l1) { x1 = l1; x2 = (v - mb * l1) * imc; } else if (x1 < -l1) { x1 = -l1; x2 = (v + mb * l1) * imc; }
      if (x2 > l2) { x2 = l2; x1 = (u - mb * l2) * ima; } else if (x2 < -l2) { x2 = -l2; x1 = (u + mb * l2) * ima; }
      if (x1 > l1) x1 = l1; else if (x1 < -l1) x1 = -l1;
      for (int k = 0; k < 3; k++) { c1[k] = p1[k] + a1[k] * x1; c2[k] = p2[k] + a2[k] * x2; }
    } else {
      single = false;
      for (int sg = -1; sg <= 1 && h.n < 2; sg += 2) {
        T e1[3], t[3]; for (int k = 0; k < 3; k++) { e1[k] = p1[k] + a1[k] * sg * l1; t[k] = e1[k] - p2[k]; }
        T x2 = dot3(t, a2);
        if (x2 >= -l2 && x2 <= l2) { T e2[3]; for (int k = 0; k < 3; k++) e2[k] = p2[k] + a2[k] * x2; sphere_sphere(h, m, e1, r1, e2, r2); }
      }
      for (int sg = -1; sg <= 1 && h.n < 2; sg += 2) {
        T e2[3], t[3]; for (int k = 0; k < 3; k++) { e2[k] = p2[k] + a2[k] * sg * l2; t[k] = e2[k] - p1[k]; }
        T x1 = dot3(t, a1);
        if (x1 >= -l1 && x1 <= l1) { T e1[3]; for (int k = 0; k < 3; k++) e1[k] = p1[k] + a1[k] * x1; sphere_sphere(h, m, e1, r1, e2, r2); }
      }
    }
  }
  if (single) sphere_sphere(h, m, c1, r1, c2, r2);
}

// ---- [3P] mj_collision: broad phase (both lanes: the same straight-line code over the 126 pairs, geom centres read once from
// the LDS column), narrow phase two candidates per trip, rows slot by slot -----------------------------------------------------
template <class T, class P>
REX_HD void collide(const P& p, const Model<T>& m, const T (&vl)[LD], const PSmooth<T>& S, PKin<T>& K, PScratch<T>& s) {
  K.ncon = 0;
  const bool left = p.side() != 0;
  T* const col = p.col();
  REX_HSTAMP(c0);
  T gp[NGEOM][3], ga[NGEOM][3];
  static_for<1, NGEOM>([&](auto GG) { constexpr int g = GG; for (int k = 0; k < 3; k++) gp[g][k] = col[PGEO + (g - 1) * 6 + k]; });
  static_for<1, NGEOM>([&](auto GG) { constexpr int g = GG;
    if constexpr (kGeomType[g] == G_CAPSULE) for (int k = 0; k < 3; k++) ga[g][k] = col[PGEO + (g - 1) * 6 + 3 + k]; });
  unsigned cand[(MAXPAIR + 31) / 32] = {};
  T margin = m.margin, bnd[NGEOM]; opaque(margin);
  static_for<1, NGEOM>([&](auto GG) { constexpr int g = GG; bnd[g] = m.geom_bound[g]; opaque(bnd[g]); });
  const T slack = margin + T(1e-4);
  static_for<0, kPairs.n>([&](auto PP) {     // (the tests of humanoid_engine.hpp::collide, see there)
    constexpr int q = PP, g1 = kPairs.g1[q], g2 = kPairs.g2[q], t1 = kGeomType[g1], t2 = kGeomType[g2];
    bool keep;
    if constexpr (t1 == G_PLANE) {
#if !defined(REX_NO_SECOND_CULL)
      if constexpr (t2 == G_CAPSULE) keep = !(gp[g2][2] - T(kGeomHalfUB[g2]) * habs(ga[g2][2]) - T(kGeomRadUB[g2]) > slack);
      else
#endif
      keep = !(gp[g2][2] - bnd[g2] > margin);
    } else {
      const T d[3] = {gp[g2][0] - gp[g1][0], gp[g2][1] - gp[g1][1], gp[g2][2] - gp[g1][2]}, reach = bnd[g1] + bnd[g2] + margin, dd = dot3(d, d);
      T worst = dd - reach * reach;
#if !defined(REX_NO_SECOND_CULL)
      if constexpr (t1 == G_CAPSULE && t2 == G_CAPSULE) {
        const T rs = T(kGeomRadUB[g1] + kGeomRadUB[g2]) + slack;
        const T c = dot3(ga[g1], ga[g2]), sn = fast_sqrt(hmax(T(0), T(1) - c * c)), p1 = dot3(d, ga[g1]), p2 = dot3(d, ga[g2]);
        const T e1 = rs + T(kGeomHalfUB[g2]) * sn, e2 = rs + T(kGeomHalfUB[g1]) * sn;
        worst = hmax(worst, hmax((dd - p1 * p1) - e1 * e1, (dd - p2 * p2) - e2 * e2));
      } else if constexpr (t1 == G_CAPSULE || t2 == G_CAPSULE) {
        constexpr int gc = t1 == G_CAPSULE ? g1 : g2;
        const T rs = T(kGeomRadUB[g1] + kGeomRadUB[g2]) + slack, pc = dot3(d, ga[gc]);
        worst = hmax(worst, (dd - pc * pc) - rs * rs);
      }
#endif
      keep = !(worst > T(0));
    }
    cand[q >> 5] |= keep ? (1u << (q & 31)) : 0u;
  });
  REX_HSTAMP(c1); REX_HACC(K, HT_BROAD, c0, c1);
  unsigned w0 = cand[0], w1 = cand[1], w2 = cand[2], w3 = cand[3];
  static_assert((kPairs.n + 31) / 32 == 4, "pair mask words");
  auto pop = [&]() -> int {
    if ((w0 | w1 | w2 | w3) == 0u) return -1;
    const unsigned wsel = w0 ? w0 : (w1 ? w1 : (w2 ? w2 : w3));
    const int base = w0 ? 0 : (w1 ? 32 : (w2 ? 64 : 96));
    const unsigned cleared = wsel & (wsel - 1u);
    if (w0) w0 = cleared; else if (w1) w1 = cleared; else if (w2) w2 = cleared; else w3 = cleared;
    return base + __builtin_ctz(wsel);
  };
  // Both lanes hold the same mask and pop the same two candidates per trip; lane 0 takes the first, lane 1 the second.  The
  // queue keeps table order: lane 0's hits, then lane 1's.
  int nq = 0;
  bool more = true;
  int pa = pop(), pb = pop();
  while (more) {
    for (;;) {
      const bool go = pa >= 0 && nq <= PHITQ_MAX - 4;
      if (!p.any(go)) break;
      REX_HSTAMP(n0);
      if (go) {
        const int mine = left ? pb : pa;
        Hits<T> h; h.n = 0;
        for (int k = 0; k < 2; k++) { h.dist[k] = 0; for (int x = 0; x < 3; x++) { h.pos[k][x] = 0; h.normal[k][x] = 0; } }
        if (mine >= 0) { const PairRec<T> pr = m.pair[mine]; collide_pair_col(m, col, pr, h); }
        const int other = (int)p.xchg((unsigned)h.n);
        const int at = nq + (left ? other : 0);
        if (h.n > 0) { T* q = col + PHITQ + PHITQ_WORDS * at; q[0] = h.dist[0]; for (int k = 0; k < 3; k++) { q[1 + k] = h.pos[0][k]; q[4 + k] = h.normal[0][k]; } q[7] = T(mine); }
        if (h.n > 1) { T* q = col + PHITQ + PHITQ_WORDS * (at + 1); q[0] = h.dist[1]; for (int k = 0; k < 3; k++) { q[1 + k] = h.pos[1][k]; q[4 + k] = h.normal[1][k]; } q[7] = T(mine); }
        nq += h.n + other;
        pa = pop(); pb = pop();
      }
      REX_HSTAMP(n1); REX_HACC(K, HT_PAIR, n0, n1); REX_HCNT(K, HC_PAIR_CALLS, 1);
    }
    p.sync();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
    for (int slot = 0; slot < PHITQ_MAX; slot++) {
      const bool mine = slot < nq;
      if (!p.any(mine)) break;
      REX_HSTAMP(r0);
      if (mine) {
        const T* q = col + PHITQ + PHITQ_WORDS * slot;
        const T hd = q[0], hp[3] = {q[1], q[2], q[3]}, hn[3] = {q[4], q[5], q[6]};
        const PairRec<T> pr = m.pair[(int)q[7]];
        T yaxis[3];
        for (int k = 0; k < 3; k++) yaxis[k] = col[PGEO + (pr.g2 - 1) * 6 + 3 + k];
        const bool has_y = pr.t1 == G_PLANE && pr.t2 == G_CAPSULE;
        add_contact(p, K, s, S, m, vl, pr, hd, hp, hn, has_y ? yaxis : (const T*)nullptr);
      }
      REX_HSTAMP(r1); REX_HACC(K, HT_ROWS, r0, r1); REX_HCNT(K, HC_ROW_CALLS, 1);
    }
    p.sync();
    nq = 0;
    more = p.any(pa >= 0);
  }
  REX_HSTAMP(c2); REX_HACC(K, HT_NARROW_LOOP, c1, c2);
}

// ---- [3P] mj_solPGS with the rows in scratch (more rows than the LDS column holds: pile-ups) ---------------------------------
template <class T, class P>
REX_HD int solve_pgs(const P& p, const Model<T>& m, const PFactor<T>& F, const PKin<T>& K, PScratch<T>& s, T (&qacc)[LD]) {
  for (int k = 0; k < LD; k++) qacc[k] = K.qacc_smooth[k];
  for (int i = 0; i < K.nefc; i++) {
    T x[LD], jr[LD];
    for (int k = 0; k < LD; k++) { jr[k] = s.J[i][k]; x[k] = jr[k]; }
    solve(p, F, x);
    const T a = s.R[i] + pdot(p, jr, x);
    for (int k = 0; k < LD; k++) s.MiJ[i][k] = x[k];
    s.Adiag[i] = a; s.force[i] = 0;   // warmstart disabled (humanoid.xml:11)
  }
  const T scale = T(1) / (m.meaninertia * T(NV));
  int it = 0;
  const int n = K.nefc;
  T jn[LD], mn[LD], sn[4];
  auto fetch = [&](int i) { for (int k = 0; k < LD; k++) { jn[k] = s.J[i][k]; mn[k] = s.MiJ[i][k]; } sn[0] = s.R[i]; sn[1] = s.aref[i]; sn[2] = s.Adiag[i]; sn[3] = s.force[i]; };
  for (; it < m.iterations; it++) {
    T improvement = 0;
    fetch(0);
    for (int i = 0; i < n; i++) {
      T jr[LD], mr[LD];
      for (int k = 0; k < LD; k++) { jr[k] = jn[k]; mr[k] = mn[k]; }
      const T Ri = sn[0], arefi = sn[1], Ad = sn[2], old = sn[3];
      fetch(i + 1 < n ? i + 1 : i);
      const T res = Ri * old - arefi + pdot(p, jr, qacc);
      const T nf = hmax(T(0), old - res / Ad), df = nf - old;
      s.force[i] = nf;
      if (df != T(0)) for (int k = 0; k < LD; k++) qacc[k] += mr[k] * df;
      improvement -= T(0.5) * df * df * Ad + df * res;
    }
    if (improvement * scale < m.tolerance) { it++; break; }
  }
  return it;
}

// ---- pgs_sweeps_sq for the 16-row size over the pair (device, fp32): 256 matrix entries do not fit the VGPRs one lane has left during
// the sweeps (they camp in AGPRs: a move per use), half of them do.  Lane 0 keeps res[0 .. NP/2), lane 1 res[NP/2 .. NP), so an update
// of f_i pushes only HALF of row i per lane; the lane that owns res_i computes the new force, the partner gets it with one DPP move.
// Every residual sees the same sequence of multiply-adds as in the replicated form: bit-identical results.
#if defined(__HIP_DEVICE_COMPILE__)
template <int NC, class P>
__device__ __forceinline__ int pgs_sweeps_sq_pair(const P& p, const Model<float>& m, const float* col, float (&f)[DUAL_NMAX]) {
  constexpr int NP = sq_stride(NC), BOFF = NP * NP, DOFF = BOFF + NP, NH = NP / 2;
  static_assert(NH % 2 == 0 && NC <= DUAL_NMAX, "half rows are read as 8-byte words");
  typedef pgs_v2f v2f;
  const bool hi = p.side() != 0;
  const float* const colh = col + (hi ? NH : 0);       // this lane's half of every row (8-byte aligned: NH is even)
  const float scale = 1.0f / (m.meaninertia * float(NV));
  v2f rp[NH / 2]; float fv[NC];
  static_for<0, NH / 2>([&](auto QQ) { constexpr int q = QQ; rp[q] = *(const v2f*)(colh + BOFF + 2 * q); });
  static_for<0, NC>([&](auto II) { fv[II] = 0.0f; });
  float improvement = 0;
  auto sweep = [&](auto CHECK) {
    static_for<0, NC>([&](auto II) {
      constexpr int i = II;
      constexpr bool ownhi = i >= NH;                  // the lane that holds res_i (and A_ii in its half of row i)
      constexpr int il = i - (ownhi ? NH : 0);
      v2f a[NH / 2];
      static_for<0, NH / 2>([&](auto QQ) { constexpr int q = QQ; a[q] = *(const v2f*)(colh + i * NP + 2 * q); });
      const float di = col[DOFF + i];
      const float res = (il & 1) ? rp[il / 2].y : rp[il / 2].x;
      const float old = fv[i], nf_mine = __builtin_fmaxf(0.0f, old - res * di);
      const float nf_other = p.xchg(nf_mine);
      const bool owner = hi == ownhi;
      const float nf = owner ? nf_mine : nf_other, df = nf - old;
      fv[i] = nf;
      if constexpr (decltype(CHECK)::value) {
        const float aii = (il & 1) ? a[il / 2].y : a[il / 2].x;
        const float term = df * (0.5f * df * aii + res);
        improvement -= owner ? term : 0.0f;
      }
      const v2f dfp = {df, df};
      static_for<0, NH / 2>([&](auto QQ) { constexpr int q = QQ; rp[q] = __builtin_elementwise_fma(a[q], dfp, rp[q]); });
    });
  };
  int it = 0;
  while (it < m.iterations) {
    int nun = m.iterations - 1 - it; nun = nun < PGS_CHECK - 1 ? nun : PGS_CHECK - 1;
#pragma unroll 1
    for (int u = 0; u < nun; u++) { sweep(std::false_type{}); it++; }
    improvement = 0;
    sweep(std::true_type{}); it++;
    improvement += p.xchg(improvement);                // (commutative: the same bits in both lanes)
    if (improvement * scale < m.tolerance) break;
  }
  static_for<0, NC>([&](auto II) { f[II] = fv[II]; });
  return it;
}
#endif
template <int NC, class T, class P>
REX_HD int pgs_sweeps_sq_split(const P& p, const Model<T>& m, const T* col, T (&f)[DUAL_NMAX]) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(REX_NO_PK) && !defined(REX_NO_SPLIT_SWEEPS)
  if constexpr (sizeof(T) == 4) return pgs_sweeps_sq_pair<NC>(p, m, col, f);
  else
#endif
  return pgs_sweeps_sq<NC>(m, col, f);
}

// ---- 17 .. 21 rows over the pair: the PADDED lower triangle -- row r as r / 4 + 1 quads of four words at seg_off(r), b behind it, 1 / A_ii
// formed from the diagonal -- with every row segment read ONCE per sweep: when row i is due, its segment A_i,0..i gives
//   (1) the lower part of res_i:  sum_{j <= i} A_ij f_j   (f_j already this sweep's for j < i), and, once f_i is known,
//   (2) the push of A_ij f_i into u_j, j <= i:            u_j collects sum_{k > j} A_kj f_k for row j's NEXT turn (A is symmetric),
// so res_i = b_i + (1) + u_i - A_ii f_i (u_i also holds row i's own push of the previous sweep; f_i is still the old one in (1)).
// The packed triangle's other half -- column i of the later rows, scattered words -- is never read.  Over the pair: lane h takes
// words 2h, 2h + 1 of every quad (8-byte reads at a lane-constant offset, the same instruction in both lanes), keeps the forces and
// the u of ITS columns as packed pairs, and the two partial sums of a row meet in one DPP add.  21 rows: 66 eight-byte reads + 132
// v_pk_fma + ~13 instructions per row instead of the packed dot-product form's 222 scattered reads + 252 v_pk_fma + re-pairing moves
// per lane (pgs_sweeps<21>: 837 instructions per sweep, the slowest waves of a launch).
constexpr int seg_off(int r) { return 4 * ((r >> 2) + 1) * (2 * (r >> 2) + (r & 3)); }   // rows 4a .. 4a + 3 hold a + 1 quads each
constexpr int SEG_B = seg_off(DUAL_NMAX);
static_assert(seg_off(4) == 16 && SEG_B == 264 && SEG_B + DUAL_NMAX <= DUAL_WORDS, "padded triangle + b must fit the LDS column");
#if defined(__HIP_DEVICE_COMPILE__)
template <int NC, class P>
__device__ __forceinline__ int pgs_sweeps_seg_pair(const P& p, const Model<float>& m, const float* col, float (&f)[DUAL_NMAX]) {
  typedef pgs_v2f v2f;
  constexpr int NQ = (NC + 3) / 4;                       // quads of the longest row
  const bool hi = p.side() != 0;
  const float* const colh = col + (hi ? 2 : 0);          // this lane's two words of every quad
  const float scale = 1.0f / (m.meaninertia * float(NV));
  v2f fh[NQ], u[NQ], bh[NQ]; float fv[NC], dd[NC], aii[NC];
  static_for<0, NQ>([&](auto QQ) { constexpr int q = QQ; fh[q] = v2f{0.0f, 0.0f}; u[q] = v2f{0.0f, 0.0f}; bh[q] = *(const v2f*)(colh + SEG_B + 4 * q); });
  static_for<0, NC>([&](auto II) { constexpr int i = II; fv[i] = 0.0f; aii[i] = col[seg_off(i) + i]; dd[i] = aii[i] != 0.0f ? rcp_t(aii[i]) : 0.0f; });
  float improvement = 0;
  auto sweep = [&](auto CHECK) {
    static_for<0, NC>([&](auto II) {
      constexpr int i = II, Q = i / 4 + 1, qi = i / 4;
      constexpr bool ownhi = ((i >> 1) & 1) != 0;        // the lane whose words hold column i
      v2f a[Q];
      static_for<0, Q>([&](auto QQ) { constexpr int q = QQ; a[q] = *(const v2f*)(colh + seg_off(i) + 4 * q); });
      v2f acc = {0.0f, 0.0f};
      static_for<0, Q>([&](auto QQ) { constexpr int q = QQ; acc = __builtin_elementwise_fma(a[q], fh[q], acc); });
      const bool owner = hi == ownhi;
      const float old = fv[i];
      const float mine = ((i & 1) ? bh[qi].y : bh[qi].x) + ((i & 1) ? u[qi].y : u[qi].x) - aii[i] * old;   // b_i + u_i - A_ii f_i: the owner's
      const float t = (acc.x + acc.y) + (owner ? mine : 0.0f);
      const float res = t + p.xchg(t);                   // (commutative: the same bits in both lanes)
      const float nf = __builtin_fmaxf(0.0f, old - res * dd[i]), df = nf - old;
      fv[i] = nf;
      if constexpr (decltype(CHECK)::value) improvement -= df * (0.5f * df * aii[i] + res);
      if constexpr (i & 1) { fh[qi].y = owner ? nf : fh[qi].y; u[qi].y = owner ? 0.0f : u[qi].y; }
      else { fh[qi].x = owner ? nf : fh[qi].x; u[qi].x = owner ? 0.0f : u[qi].x; }
      const v2f nfp = {nf, nf};
      static_for<0, Q>([&](auto QQ) { constexpr int q = QQ; u[q] = __builtin_elementwise_fma(a[q], nfp, u[q]); });
    });
  };
  int it = 0;
  while (it < m.iterations) {
    int nun = m.iterations - 1 - it; nun = nun < PGS_CHECK - 1 ? nun : PGS_CHECK - 1;
#pragma unroll 1
    for (int k = 0; k < nun; k++) { sweep(std::false_type{}); it++; }
    improvement = 0;
    sweep(std::true_type{}); it++;
    if (improvement * scale < m.tolerance) break;
  }
  static_for<0, NC>([&](auto II) { f[II] = fv[II]; });
  static_for<NC, DUAL_NMAX>([&](auto II) { f[II] = 0.0f; });
  return it;
}
#endif
// the same Gauss-Seidel sweeps on the padded triangle, plain form (host builds; fp64 pins layout and algorithm)
template <int NC, class T, class P>
REX_HD int pgs_sweeps_seg(const P& p, const Model<T>& m, const T* col, T (&f)[DUAL_NMAX]) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(REX_NO_PK)
  if constexpr (sizeof(T) == 4) return pgs_sweeps_seg_pair<NC>(p, m, col, f);
  else
#endif
  {
    auto A = [&](int i, int j) { return i >= j ? col[seg_off(i) + j] : col[seg_off(j) + i]; };
    const T scale = T(1) / (m.meaninertia * T(NV));
    int it = 0;
    for (; it < m.iterations; it++) {
      T improvement = 0;
      for (int i = 0; i < NC; i++) {
        T res = col[SEG_B + i];
        for (int j = 0; j < NC; j++) res += A(i, j) * f[j];
        const T a = A(i, i), di = a != T(0) ? rcp_t(a) : T(0);
        const T old = f[i], nf = hmax(T(0), old - res * di), df = nf - old;
        f[i] = nf;
        improvement -= df * (T(0.5) * df * a + res);
      }
      if (pgs_checked<T>(it, m.iterations) && improvement * scale < m.tolerance) { it++; break; }
    }
    return it;
  }
}

// ---- the dual PGS (A = J M^-1 J^T + R in the env's LDS column; humanoid_engine.hpp::solve_pgs_dual) with the row algebra
// over the pair: every lane back-substitutes its own 16 columns, dot products are partial sums exchanged once ------------------
template <class T, class P>
REX_HD int solve_pgs_dual(const P& p, const Model<T>& m, const PFactor<T>& F, PKin<T>& K, PScratch<T>& s, T (&qacc)[LD]) {
  const int n = K.nefc;
  T* const col = (T*)__builtin_assume_aligned(p.col(), 16);
  REX_HSTAMP(p0);
  int lvl = 0;
  static_for<0, 8>([&](auto LL) { constexpr int thr[8] = {4, 8, 10, 12, 14, 16, 18, 21}; if (p.any(n > thr[LL])) lvl = LL + 1; });
#if defined(__HIP_DEVICE_COMPILE__)
  lvl = __builtin_amdgcn_readfirstlane(lvl);
#endif
  const bool sq = lvl <= 5;   // square rows up to 16 rows, the padded triangle (seg_off) above
  const int stride = lvl == 0 ? 4 : lvl == 1 ? 8 : lvl <= 3 ? 12 : 16;
  const int boff = sq ? stride * stride : SEG_B, doff = boff + stride;
  static_for<0, DUAL_WORDS>([&](auto KK) { col[KK] = T(0); });   // (both lanes write the same zeros; padding must read as zero)
  for (int j0 = 0; j0 < n; j0 += 4) {
    T y[4][LD], Rr[4], ar[4];
    bool ok[4]; int jj[4];
    static_for<0, 4>([&](auto TT) { constexpr int t = TT; ok[t] = j0 + t < n; jj[t] = ok[t] ? j0 + t : j0; });
    static_for<0, 4>([&](auto TT) { constexpr int t = TT; for (int k = 0; k < LD; k++) y[t][k] = s.J[jj[t]][k]; Rr[t] = s.R[jj[t]]; ar[t] = s.aref[jj[t]]; });
    static_for<0, 4>([&](auto TT) { constexpr int t = TT; const T b = pdot(p, y[t], K.qacc_smooth) - ar[t]; if (ok[t]) col[boff + j0 + t] = b; });
    solve_back<4>(p, F, y);
    auto rowp = [&](int r) -> T* { return col + (sq ? r * stride : seg_off(r)); };
    auto put = [&](int r, int c, T v) { rowp(r)[c] = v; if (sq && c != r) col[c * stride + r] = v; };
    static_for<0, 4>([&](auto UU) {
      constexpr int u = UU;
      T z[LD];
      static_for<0, LD>([&](auto KK) { constexpr int k = KK; z[k] = y[u][k] * F.a[lidx(k, k)]; });
      if (ok[u]) for (int k = 0; k < LD; k++) s.J[j0 + u][k] = z[k];
      const T d = Rr[u] + pdot(p, z, y[u]);
      if (ok[u]) { rowp(j0 + u)[j0 + u] = d; if (sq) col[doff + j0 + u] = rcp_t(d); }
      static_for<u + 1, 4>([&](auto TT) { constexpr int t = TT; const T v = pdot(p, z, y[t]); if (ok[t]) put(j0 + t, j0 + u, v); });
    });
    T ra[LD], rb[LD];
    auto fetch2 = [&](int i, T (&u)[LD], T (&v)[LD]) {
      const int i0 = i < j0 ? i : 0, i1 = i + 1 < j0 ? i + 1 : i0;
      for (int k = 0; k < LD; k++) { u[k] = s.J[i0][k]; v[k] = s.J[i1][k]; }
    };
    auto use1 = [&](int i, T (&u)[LD]) {
      T a[4];
      static_for<0, 4>([&](auto TT) { constexpr int t = TT; a[t] = pdot(p, u, y[t]); });
      if (i < j0) { put(j0, i, a[0]); if (ok[1]) put(j0 + 1, i, a[1]); if (ok[2]) put(j0 + 2, i, a[2]); if (ok[3]) put(j0 + 3, i, a[3]); }
    };
    for (int i = 0; i < j0; i += 2) {
      fetch2(i, ra, rb);
      use1(i, ra); use1(i + 1, rb);
    }
  }
  REX_HSTAMP(p1); REX_HACC(K, HT_BUILD_A, p0, p1);
  int it;
  T f[DUAL_NMAX];
  static_for<0, DUAL_NMAX>([&](auto II) { f[II] = T(0); });
  switch (lvl) {
    case 0: it = pgs_sweeps_sq<4>(m, col, f); break;
    case 1: it = pgs_sweeps_sq<8>(m, col, f); break;
    case 2: it = pgs_sweeps_sq<10>(m, col, f); break;
    case 3: it = pgs_sweeps_sq<12>(m, col, f); break;
    case 4: it = pgs_sweeps_sq<14>(m, col, f); break;
    case 5: it = pgs_sweeps_sq_split<16>(p, m, col, f); break;
    case 6: it = pgs_sweeps_seg<18>(p, m, col, f); break;
    default: it = pgs_sweeps_seg<DUAL_NMAX>(p, m, col, f); break;
  }
  REX_HSTAMP(p2); REX_HACC(K, HT_SWEEPS, p1, p2); REX_HCNT(K, HC_SWEEPS, it);
#if defined(REX_KTIME) && defined(__HIP_DEVICE_COMPILE__)
  { int nm = n; for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(nm, off); nm = o > nm ? o : nm; }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&g_ktime[24 + lvl], 1ull); atomicAdd(&g_ktime[40 + lvl], p2 - p1); atomicAdd(&g_ktime[56 + lvl], p1 - p0);
                                   atomicAdd(&g_ktime[72 + (nm < 23 ? nm : 23)], 1ull); } }   // [72 + n]: wave-evaluations by their largest row count
#endif
  T x[LD];
  for (int k = 0; k < LD; k++) x[k] = 0;
  static_for<0, DUAL_NMAX / 3>([&](auto CC) {
    constexpr int i0 = 3 * CC;
    if (p.any(i0 < n)) {
      T r[3][LD], fi[3];
      static_for<0, 3>([&](auto RR) { constexpr int r_ = RR, i = i0 + r_; const int ii = i < n ? i : 0; fi[r_] = i < n ? f[i] : T(0);
        if (i < n) s.force[i] = fi[r_];
        for (int k = 0; k < LD; k++) r[r_][k] = s.J[ii][k]; });
      for (int k = 0; k < LD; k++) x[k] += r[0][k] * fi[0] + r[1][k] * fi[1] + r[2][k] * fi[2];
    }
  });
  solve_fwd(F, x);
  for (int k = 0; k < LD; k++) qacc[k] = K.qacc_smooth[k] + x[k];
  REX_HSTAMP(p3); REX_HACC(K, HT_QACC, p2, p3);
  return it;
}

// ---- [3P] mj_forward ------------------------------------------------------------------------------------------------------------
// cl: the local controls (hinge of local dof d at d - 6), already clamped?  no: raw, clamped here (ctrlrange +-0.4, humanoid.xml:6)
template <class T, class P>
REX_HD int forward(const P& p, const Model<T>& m_in, const PLane<T>& L, const T (&ql)[LQ], const T (&vl)[LD], const T (&cl)[LU], PKin<T>& K,
                   PScratch<T>& s, T (&qacc)[LD], PObs<T>* park) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(REX_HOIST_MODEL)
  int zidx = 0; asm volatile("" : "+s"(zidx));
  const Model<T>& m = (&m_in)[zidx];
#else
  const Model<T>& m = m_in;
#endif
  const bool left = p.side() != 0;
  K.overflow = 0;
  REX_HSTAMP(t0);
  PFactor<T> F;
  {
    PSmooth<T> S;
    kinematics(p, m, ql, S);
    REX_FENCE(); REX_HSTAMP(t3); REX_HACC(K, HT_SMOOTH, t0, t3);
    limit_rows(p, m, ql, vl, K, s);
    REX_HSTAMP(t3b); REX_HACC(K, HT_LIMITS, t3, t3b);
    collide(p, m, vl, S, K, s);
    REX_FENCE(); REX_HSTAMP(t4);
    com_pos(p, m, L, S);
    T qfrc_bias[LD], act[LD];
    com_vel_rne(p, m, vl, S, qfrc_bias);
    static_for<0, 6>([&](auto II) { act[II] = T(0); });
    static_for<6, LD>([&](auto DD) {   // motor of the hinge at local dof d (humanoid.xml:106-122): gear * clamp(ctrl)
      constexpr int ld = DD;
      constexpr int uR = ld == 6 ? 1 : ld == 7 ? 0 : ld == 8 ? 2 : ld < 13 ? 3 + (ld - 9) : 11 + (ld - 13);
      constexpr int uL = ld < 9 ? uR : ld < 13 ? 7 + (ld - 9) : 14 + (ld - 13);
      static_assert(kActDof[uR] == gdR(ld) && kActDof[uL] == gdL(ld), "actuator table");
      const T gear = ld < 9 ? m.act_gear[uR] : sel(left, m.act_gear[uR], m.act_gear[uL]);
      act[ld] = gear * hmin(hmax(cl[ld - 6], T(-0.4)), T(0.4));
    });
    static_for<0, LD>([&](auto II) { constexpr int i = II; K.qfrc_smooth[i] = -L.damping[i] * vl[i] - qfrc_bias[i] + act[i]; });
    static_for<6, LD>([&](auto DD) { constexpr int ld = DD;   // joint springs, springref 0
      const T st = ld < 9 ? m.jnt_stiff[gdR(ld) - 5] : sel(left, m.jnt_stiff[gdR(ld) - 5], m.jnt_stiff[gdL(ld) - 5]);
      K.qfrc_smooth[ld] -= st * ql[ld + 1]; });
    crb(p, m, S, F);
    if (park) {
      static_for<0, LB>([&](auto BB) { constexpr int b = BB; for (int k = 0; k < 10; k++) park->cinert[b][k] = S.cinert[b][k]; for (int k = 0; k < 6; k++) park->cvel[b][k] = S.cvel[b][k]; park->xipos_x[b] = S.xipos[b][0]; });
      static_for<0, LD>([&](auto II) { park->act[II] = act[II]; });
    }
    REX_FENCE(); REX_HSTAMP(t4e); REX_HACC(K, HT_SMOOTH, t4, t4e);
  }
  REX_FENCE(); REX_HSTAMP(t5);
  factor(p, F);
  for (int i = 0; i < LD; i++) K.qacc_smooth[i] = K.qfrc_smooth[i];
  solve(p, F, K.qacc_smooth);
  REX_FENCE(); REX_HSTAMP(t7); REX_HACC(K, HT_FACTOR, t5, t7); REX_HCNT(K, HC_EVALS, 1); REX_HCNT(K, HC_NEFC, K.nefc);
  int it = 0;
  if (K.nefc == 0) { for (int i = 0; i < LD; i++) qacc[i] = K.qacc_smooth[i]; }
  else it = K.nefc <= DUAL_NMAX ? solve_pgs_dual(p, m, F, K, s, qacc) : solve_pgs(p, m, F, K, s, qacc);
  REX_HSTAMP(t8); REX_HACC(K, HT_FORWARD, t0, t8);
  return it;
}

// [3P] mj_integratePos on the local coordinates
template <class T>
REX_HD void integrate_pos(T (&ql)[LQ], const T (&vl)[LD], T h) {
  for (int k = 0; k < 3; k++) ql[k] += h * vl[k];
  T w[3] = {vl[3], vl[4], vl[5]}, n = hsqrt(dot3(w, w));
  if (n * h > T(1e-15)) {
    T sn, cs; hsincos(T(0.5) * n * h, sn, cs);
    const T sn_n = sn * rcp_t(n);
    T dq[4] = {cs, w[0] * sn_n, w[1] * sn_n, w[2] * sn_n}, r[4];
    qmul(r, &ql[3], dq); qnorm(r);
    for (int k = 0; k < 4; k++) ql[3 + k] = r[k];
  }
  for (int k = 6; k < LD; k++) ql[k + 1] += h * vl[k];
}

// One mj_step with RK4 ([3P] mj_RungeKutta, N = 4)
template <class T, class P>
REX_HD void substep(const P& p, const Model<T>& m, const PLane<T>& L, T (&ql)[LQ], T (&vl)[LD], const T (&cl)[LU], PKin<T>& K, PScratch<T>& s, PObs<T>* park) {
  const T h = m.timestep;
  T q0[LQ], v0[LD], dq[LD], dv[LD];
  static_for<0, LQ>([&](auto KK) { q0[KK] = ql[KK]; });
  static_for<0, LD>([&](auto KK) { v0[KK] = vl[KK]; dq[KK] = 0; dv[KK] = 0; });
  for (int stage = 0; stage < 4; stage++) {
    T acc[LD];
    forward(p, m, L, ql, vl, cl, K, s, acc, stage == 3 ? park : (PObs<T>*)nullptr);
    const T w = (stage == 0 || stage == 3) ? T(1.0 / 6) : T(1.0 / 3), c = stage == 2 ? h : T(0.5) * h;
    static_for<0, LD>([&](auto KK) { constexpr int k = KK; dq[k] += w * vl[k]; dv[k] += w * acc[k]; });
    if (stage < 3) {
      T vs[LD]; for (int k = 0; k < LD; k++) vs[k] = vl[k];
      static_for<0, LQ>([&](auto KK) { ql[KK] = q0[KK]; });
      integrate_pos(ql, vs, c);
      static_for<0, LD>([&](auto KK) { constexpr int k = KK; vl[k] = v0[k] + c * acc[k]; });
    } else {
      static_for<0, LQ>([&](auto KK) { ql[KK] = q0[KK]; });
      static_for<0, LD>([&](auto KK) { constexpr int k = KK; vl[k] = v0[k] + h * dv[k]; });
      integrate_pos(ql, dq, h);
    }
  }
}

// RandomHumanoidEnv.step (random_humanoid.py:161-216), local view.  xipos_x: data.xipos[:, 0] of this lane's 8 bodies left by the
// previous forward (mass_center() "before"), replaced by this step's.  asq: sum of squares of ALL 17 raw actions.
template <class T, class P>
REX_HD void env_step(const P& p, const Model<T>& m, const PLane<T>& L, T (&ql)[LQ], T (&vl)[LD], const T (&cl)[LU], T asq, T (&xipos_x)[LB],
                     PKin<T>& K, PScratch<T>& s, PObs<T>& park, T& reward, bool& done, T* terms = nullptr) {
  const bool left = p.side() != 0;
  T mt = 0, s0 = 0, s1 = 0;
  static_for<0, LB>([&](auto BB) { constexpr int b = BB; const T ms = (b < 3 && left) ? T(0) : L.mass[b]; mt += ms; s0 += ms * xipos_x[b]; });
  mt = psum(p, mt); s0 = psum(p, s0);
  for (int f = 0; f < 5; f++) substep(p, m, L, ql, vl, cl, K, s, f == 4 ? &park : (PObs<T>*)nullptr);   // frame_skip 5 (:41)
  static_for<0, LB>([&](auto BB) { constexpr int b = BB; xipos_x[b] = park.xipos_x[b]; const T ms = (b < 3 && left) ? T(0) : L.mass[b]; s1 += ms * xipos_x[b]; });
  s1 = psum(p, s1);
  const T dt = m.timestep * T(5);
  reward = T(1.25) * (s1 / mt - s0 / mt) / dt - T(0.1) * asq - T(0) /* cfrc_ext = 0, SURVEY Q15 */ + T(5);
  if (terms) { terms[0] = T(1.25) * (s1 / mt - s0 / mt) / dt; terms[1] = -T(0.1) * asq; terms[2] = T(5); terms[3] = -T(0); }
  done = (ql[2] < T(1.0)) || (ql[2] > T(2.0));
}

// _get_obs (random_humanoid.py:193-204): qpos[2:], qvel, cinert, cvel, qfrc_actuator, cfrc_ext (= 0).  The right lane
// writes the world / trunk rows and its side, the left lane its side; obs(IC<row on the right lane>, IC<row on the left lane>, value).
template <class T, class P, class Sink>
REX_HD void emit_obs(const P& p, const T (&ql)[LQ], const T (&vl)[LD], const PObs<T>& o, Sink&& obs) {
  const bool left = p.side() != 0;
  if (!left) {
    static_for<2, 10>([&](auto KK) { constexpr int k = KK; obs(IC<k - 2>{}, IC<k - 2>{}, ql[k]); });
    static_for<0, 9>([&](auto KK) { constexpr int k = KK; obs(IC<22 + k>{}, IC<22 + k>{}, vl[k]); });
    static_for<0, 10>([&](auto KK) { constexpr int k = KK; obs(IC<45 + k>{}, IC<45 + k>{}, T(0)); });      // world body
    static_for<0, 6>([&](auto KK) { constexpr int k = KK; obs(IC<185 + k>{}, IC<185 + k>{}, T(0)); });
    static_for<0, 3>([&](auto BB) { constexpr int lb = BB, b = lb + 1;
      static_for<0, 10>([&](auto KK) { constexpr int k = KK; obs(IC<45 + 10 * b + k>{}, IC<45 + 10 * b + k>{}, o.cinert[lb][k]); });
      static_for<0, 6>([&](auto KK) { constexpr int k = KK; obs(IC<185 + 6 * b + k>{}, IC<185 + 6 * b + k>{}, o.cvel[lb][k]); }); });
    static_for<0, 9>([&](auto KK) { constexpr int k = KK; obs(IC<269 + k>{}, IC<269 + k>{}, o.act[k]); });
  }
  static_for<9, LD>([&](auto DD) { constexpr int ld = DD, dR = gdR(ld), dL = gdL(ld);
    obs(IC<dR + 1 - 2>{}, IC<dL + 1 - 2>{}, ql[ld + 1]); obs(IC<22 + dR>{}, IC<22 + dL>{}, vl[ld]); obs(IC<269 + dR>{}, IC<269 + dL>{}, o.act[ld]); });
  static_for<3, LB>([&](auto BB) { constexpr int lb = BB, bR = gbR(lb), bL = gbL(lb);
    static_for<0, 10>([&](auto KK) { constexpr int k = KK; obs(IC<45 + 10 * bR + k>{}, IC<45 + 10 * bL + k>{}, o.cinert[lb][k]); });
    static_for<0, 6>([&](auto KK) { constexpr int k = KK; obs(IC<185 + 6 * bR + k>{}, IC<185 + 6 * bL + k>{}, o.cvel[lb][k]); }); });
  static_for<0, 42>([&](auto KK) { constexpr int k = KK; obs(IC<292 + k>{}, IC<334 + k>{}, T(0)); });         // cfrc_ext block, half each
}

// the side bodies carry no orientation offset (kinematics() skips their quaternion product); hinge qpos0 are zero
template <class T>
inline bool check_pair_model(const Model<T>& m) {
  bool ok = true;
  for (int b = 4; b < NBODY; b++) ok = ok && m.body_quat[b][0] == T(1) && m.body_quat[b][1] == T(0) && m.body_quat[b][2] == T(0) && m.body_quat[b][3] == T(0);
  for (int k = 7; k < NQ; k++) ok = ok && m.qpos0[k] == T(0);
  return ok;
}

}  // namespace pr
}  // namespace hum
}  // namespace rex
