// humanoid_engine.hpp -- forward dynamics + PGS constraint solve + RK4 for the 3-D humanoid
// (random_envs/jinja/assets/humanoid.xml: free root + 17 hinges, nv = 23, 13 bodies, 17 collision
// geoms + floor, PGS capped at 50 sweeps).  One environment per lane; unlike the planar trees the
// per-env working set (M 23x23, J and M^-1 J^T for up to 64 rows) does not fit in registers, so the
// arrays below are runtime-indexed per-lane arrays (HIP scratch, lane-interleaved => every access of
// a wave is one coalesced 256-byte segment).  Algorithms follow MuJoCo's own com-based spatial
// formulation ([3P] mj_comPos / mj_crb / mj_comVel / mj_rne), which makes this file independent of
// the oracle (world-frame Jacobian sums + Newton-Euler) it is tested against.
//
// Host + device code (REX_HD): tests compile it for the CPU in fp64 / fp32.
#pragma once
#include <math.h>

#include "planar_spec.hpp"   // REX_HD

namespace rex {
namespace hum {

constexpr int NBODY = 14;   // incl. world
constexpr int NJNT = 18;    // free + 17 hinges
constexpr int NQ = 24, NV = 23, NU = 17;
constexpr int NGEOM = 18;   // floor + 17 body geoms
constexpr int MAXPAIR = 128;
constexpr int MAXCON = 24;  // contacts kept per evaluation
constexpr int MAXEFC = 64;  // constraint rows kept per evaluation
constexpr int NXI = 30, NOBS = 376;
// dual-space PGS working set kept in LDS on the device (one column per lane): packed lower triangle of
// A = J M^-1 J^T + diag(R) for up to DUAL_NMAX rows, then force, b = J qacc_smooth - aref
constexpr int DUAL_NMAX = 21;
constexpr int DUAL_F = DUAL_NMAX * (DUAL_NMAX + 1) / 2, DUAL_B = DUAL_F + DUAL_NMAX, DUAL_WORDS = DUAL_B + DUAL_NMAX;   // 273 words per lane

enum { G_PLANE = 0, G_SPHERE = 2, G_CAPSULE = 3 };

// Compile-time dof tree of humanoid.xml (free root 0-5, abdomen 6-8, right leg 9-12, left leg 13-16, right arm 17-19,
// left arm 20-22).  The mass-matrix code below is generated from these tables (static indices => registers, no
// pointer chasing through the model); build_model() derives the same tables from the XML transcription and
// check_topology() compares the two.
constexpr int kDofParent[NV] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 8, 13, 14, 15, 5, 17, 18, 5, 20, 21};
constexpr int kDofBody[NV] = {1, 1, 1, 1, 1, 1, 2, 2, 3, 4, 4, 4, 5, 7, 7, 7, 8, 10, 10, 11, 12, 12, 13};
constexpr int kBodyParent[NBODY] = {0, 0, 1, 2, 3, 4, 5, 3, 7, 8, 1, 10, 1, 12};
constexpr int dof_depth(int d) { int n = 0; while (kDofParent[d] >= 0) { d = kDofParent[d]; n++; } return n; }
constexpr int m_row(int i) { int o = 0; for (int k = 0; k < i; k++) o += dof_depth(k) + 1; return o; }
constexpr int MNNZ = m_row(NV);   // 185 = entries (i, j) with j an ancestor-or-self dof of i
constexpr int midx(int i, int j) { return m_row(i) + dof_depth(j); }   // valid when j is an ancestor-or-self of i
constexpr bool dof_is_anc_or_self(int j, int i) { while (i >= 0) { if (i == j) return true; i = kDofParent[i]; } return false; }
template <int I, class F> REX_HD void for_anc(F&& f) {          // f(IC<j>) for every proper ancestor dof j of I, nearest first
  if constexpr (kDofParent[I] >= 0) { f(IC<kDofParent[I]>{}); for_anc<kDofParent[I]>(f); }
}
template <int I, class F> REX_HD void for_anc_self(F&& f) { f(IC<I>{}); for_anc<I>(f); }

// Compiled model (uniform across a batch; lives in __constant__ memory on the device)
template <class T>
struct Model {
  int body_parent[NBODY], body_jntadr[NBODY], body_jntnum[NBODY], body_dofadr[NBODY], body_dofnum[NBODY];
  T body_pos[NBODY][3], body_quat[NBODY][4], body_ipos[NBODY][3];
  T body_inertia[NBODY][6];     // xx yy zz xy xz yz about the COM, body axes (NOMINAL)
  T body_mass0[NBODY];          // nominal masses
  T subtreemass_root;           // compile-time subtree mass of the torso (not refreshed by set_task)
  T body_invw[NBODY][2], dof_invw[NV];
  int jnt_body[NJNT], jnt_qadr[NJNT], jnt_dadr[NJNT];
  T jnt_pos[NJNT][3], jnt_axis[NJNT][3], jnt_lo[NJNT], jnt_hi[NJNT], jnt_stiff[NJNT];
  int dof_body[NV], dof_parent[NV];
  T dof_armature[NV], dof_damping0[NV];
  int geom_type[NGEOM], geom_body[NGEOM];
  T geom_pos[NGEOM][3], geom_axis[NGEOM][3], geom_rad[NGEOM], geom_half[NGEOM];
  int npair, pair_g1[MAXPAIR], pair_g2[MAXPAIR], pair_dim[MAXPAIR];
  T pair_mu[MAXPAIR];
  int act_dof[NU]; T act_gear[NU];
  T qpos0[NQ];
  // solver constants: contacts and limits share solref (.02,1); solimp = MuJoCo default (.9,.95,.001)
  T K, B, dmin, dmax, width, margin, timestep, gravity, meaninertia, tolerance;
  int iterations;
};

template <class T> REX_HD T hsqrt(T a);
template <> REX_HD float hsqrt<float>(float a) { return sqrtf(a); }
template <> REX_HD double hsqrt<double>(double a) { return sqrt(a); }
template <class T> REX_HD T habs(T a) { return a < T(0) ? -a : a; }
template <class T> REX_HD T hmax(T a, T b) { return a > b ? a : b; }
template <class T> REX_HD T hmin(T a, T b) { return a < b ? a : b; }
REX_HD void hsincos(float a, float& s, float& c) {
#if defined(__HIP_DEVICE_COMPILE__)
  sincosf(a, &s, &c);
#else
  s = sinf(a); c = cosf(a);
#endif
}
REX_HD void hsincos(double a, double& s, double& c) { s = sin(a); c = cos(a); }

template <class T> REX_HD T dot3(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class T> REX_HD void cross3(T* r, const T* a, const T* b) {
  T x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
template <class T> REX_HD void qmul(T* r, const T* a, const T* b) {
  T w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  T x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  T y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  T z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
template <class T> REX_HD void qnorm(T* q) {
  T n = hsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < T(1e-15)) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else { T i = T(1) / n; q[0] *= i; q[1] *= i; q[2] *= i; q[3] *= i; }
}
template <class T> REX_HD void q2mat(T* m, const T* q) {
  T w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
template <class T> REX_HD void mulv(T* r, const T* m, const T* v) {
  T x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2], z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
// spatial motion cross product  (w;v) x (w2;v2) = (w x w2 ; w x v2 + v x w2)
template <class T> REX_HD void cross_motion(T* r, const T* a, const T* b) {
  T t1[3], t2[3], t3[3];
  cross3(t1, a, b); cross3(t2, a, b + 3); cross3(t3, a + 3, b);
  r[0] = t1[0]; r[1] = t1[1]; r[2] = t1[2]; r[3] = t2[0] + t3[0]; r[4] = t2[1] + t3[1]; r[5] = t2[2] + t3[2];
}
// spatial force cross product  (w;v) x* (n;f) = (w x n + v x f ; w x f)
template <class T> REX_HD void cross_force(T* r, const T* a, const T* f) {
  T t1[3], t2[3], t3[3];
  cross3(t1, a, f); cross3(t2, a + 3, f + 3); cross3(t3, a, f + 3);
  r[0] = t1[0] + t2[0]; r[1] = t1[1] + t2[1]; r[2] = t1[2] + t2[2]; r[3] = t3[0]; r[4] = t3[1]; r[5] = t3[2];
}
// cinert (10: Ixx Iyy Izz Ixy Ixz Iyz, m*off(3), m) times a motion vector -> force vector ([3P] mju_mulInertVec)
template <class T> REX_HD void mul_inert(T* r, const T* I, const T* v) {
  r[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] - I[8] * v[4] + I[7] * v[5];
  r[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] + I[8] * v[3] - I[6] * v[5];
  r[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] - I[7] * v[3] + I[6] * v[4];
  r[3] = I[8] * v[1] - I[7] * v[2] + I[9] * v[3];
  r[4] = I[6] * v[2] - I[8] * v[0] + I[9] * v[4];
  r[5] = I[7] * v[0] - I[6] * v[1] + I[9] * v[5];
}

template <class T>
struct Lane {   // randomised part of the model (xi)
  T mass[NBODY];      // body_mass (world = 0)
  T damping[NV];      // dof_damping
};

template <class T>
struct Scratch {   // per-lane working set of one forward evaluation
  T xpos[NBODY][3], xmat[NBODY][9], xquat[NBODY][4], xipos[NBODY][3];
  T com[3];                       // subtree COM of the root (MuJoCo's reference point)
  T cinert[NBODY][10], cvel[NBODY][6], cdof[NV][6], cdofdot[NV][6];
  T anchor[NV][3], axis[NV][3];   // world frame joint anchor / axis per dof
  T gpos[NGEOM][3], gaxis[NGEOM][3];   // world pose of every geom (computed once per evaluation)
  T qfrc_bias[NV], qfrc_smooth[NV], qfrc_actuator[NV], qacc_smooth[NV];
  // contacts
  int ncon; T cpos[MAXCON][3], cframe[MAXCON][9], cdist[MAXCON], cmu[MAXCON]; int cdim[MAXCON], cb1[MAXCON], cb2[MAXCON];
  // constraint rows
  int nefc; T J[MAXEFC][NV], MiJ[MAXEFC][NV], R[MAXEFC], aref[MAXEFC], Adiag[MAXEFC], force[MAXEFC];
  int overflow;
#if !defined(__HIP_DEVICE_COMPILE__)
  T dual_host[DUAL_WORDS];
#endif
};

// per-lane dual-PGS workspace: LDS on the device (word k of lane l at k * blockDim.x + l: conflict-free), plain array on the host
#if defined(__HIP_DEVICE_COMPILE__)
extern __shared__ float hum_lds[];
template <class T> REX_HD T& dual(Scratch<T>&, int k) { static_assert(sizeof(T) == 4, "device path is fp32"); return hum_lds[k * blockDim.x + threadIdx.x]; }
#else
template <class T> REX_HD T& dual(Scratch<T>& s, int k) { return s.dual_host[k]; }
#endif

// Tree-sparse joint-space inertia, packed (midx).  After factor(): L^T D L in place with the diagonal holding 1/D.
// Every index into it is a compile-time constant, so on the device it lives in registers (AGPRs as overflow).
template <class T>
struct MassFactor {
  T a[MNNZ];
  REX_HD T get(int i, int j) const { return dof_is_anc_or_self(j, i) ? a[midx(i, j)] : (dof_is_anc_or_self(i, j) ? a[midx(j, i)] : T(0)); }   // tests
};

template <class T>
REX_HD void kinematics(const Model<T>& m, const T* qpos, Scratch<T>& s) {
  for (int k = 0; k < 3; k++) s.xpos[0][k] = 0;
  s.xquat[0][0] = 1; s.xquat[0][1] = s.xquat[0][2] = s.xquat[0][3] = 0;
  q2mat(s.xmat[0], s.xquat[0]);
  for (int k = 0; k < 3; k++) s.xipos[0][k] = 0;
  for (int b = 1; b < NBODY; b++) {
    int p = m.body_parent[b];
    T xpos[3], xquat[4], t[3], R[9];
    mulv(t, s.xmat[p], m.body_pos[b]);
    for (int k = 0; k < 3; k++) xpos[k] = s.xpos[p][k] + t[k];
    qmul(xquat, s.xquat[p], m.body_quat[b]);
    for (int jj = 0; jj < m.body_jntnum[b]; jj++) {
      int j = m.body_jntadr[b] + jj, qa = m.jnt_qadr[j], da = m.jnt_dadr[j];
      if (j == 0) {   // free joint of the torso
        for (int k = 0; k < 3; k++) xpos[k] = qpos[k];
        for (int k = 0; k < 4; k++) xquat[k] = qpos[3 + k];
        qnorm(xquat); q2mat(R, xquat);
        for (int k = 0; k < 3; k++) {
          for (int x = 0; x < 3; x++) { s.axis[k][x] = (x == k) ? T(1) : T(0); s.anchor[k][x] = xpos[x]; s.axis[3 + k][x] = R[3 * x + k]; s.anchor[3 + k][x] = xpos[x]; }
        }
      } else {
        q2mat(R, xquat);
        T anchor[3], axis[3];
        mulv(t, R, m.jnt_pos[j]); for (int k = 0; k < 3; k++) anchor[k] = xpos[k] + t[k];
        mulv(axis, R, m.jnt_axis[j]);
        T ang = qpos[qa] - m.qpos0[qa], sn, cs; hsincos(T(0.5) * ang, sn, cs);
        T ql[4] = {cs, m.jnt_axis[j][0] * sn, m.jnt_axis[j][1] * sn, m.jnt_axis[j][2] * sn}, nq[4];
        qmul(nq, xquat, ql); for (int k = 0; k < 4; k++) xquat[k] = nq[k];
        qnorm(xquat); q2mat(R, xquat);
        mulv(t, R, m.jnt_pos[j]); for (int k = 0; k < 3; k++) xpos[k] = anchor[k] - t[k];
        for (int k = 0; k < 3; k++) { s.axis[da][k] = axis[k]; s.anchor[da][k] = anchor[k]; }
      }
    }
    for (int k = 0; k < 3; k++) s.xpos[b][k] = xpos[k];
    for (int k = 0; k < 4; k++) s.xquat[b][k] = xquat[k];
    q2mat(s.xmat[b], xquat);
    mulv(t, s.xmat[b], m.body_ipos[b]); for (int k = 0; k < 3; k++) s.xipos[b][k] = xpos[k] + t[k];
  }
}

// [3P] mj_comPos: reference point, cinert, cdof
template <class T>
REX_HD void com_pos(const Model<T>& m, const Lane<T>& L, Scratch<T>& s) {
  T sc[3] = {0, 0, 0};
  for (int b = 1; b < NBODY; b++) for (int k = 0; k < 3; k++) sc[k] += L.mass[b] * s.xipos[b][k];
  for (int k = 0; k < 3; k++) s.com[k] = sc[k] / m.subtreemass_root;   // compile-time subtree mass (Q4-style staleness)
  for (int k = 0; k < 10; k++) s.cinert[0][k] = 0;
  for (int b = 1; b < NBODY; b++) {
    const T* R = s.xmat[b]; const T* I = m.body_inertia[b];
    T Ib[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}, RI[9], Iw[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T a = 0; for (int k = 0; k < 3; k++) a += R[3 * i + k] * Ib[3 * k + j]; RI[3 * i + j] = a; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T a = 0; for (int k = 0; k < 3; k++) a += RI[3 * i + k] * R[3 * j + k]; Iw[3 * i + j] = a; }
    T d[3] = {s.xipos[b][0] - s.com[0], s.xipos[b][1] - s.com[1], s.xipos[b][2] - s.com[2]}, ms = L.mass[b];
    T* c = s.cinert[b];
    c[0] = Iw[0] + ms * (d[1] * d[1] + d[2] * d[2]); c[1] = Iw[4] + ms * (d[0] * d[0] + d[2] * d[2]); c[2] = Iw[8] + ms * (d[0] * d[0] + d[1] * d[1]);
    c[3] = Iw[1] - ms * d[0] * d[1]; c[4] = Iw[2] - ms * d[0] * d[2]; c[5] = Iw[5] - ms * d[1] * d[2];
    c[6] = ms * d[0]; c[7] = ms * d[1]; c[8] = ms * d[2]; c[9] = ms;
  }
  for (int i = 0; i < NV; i++) {
    if (i < 3) { for (int k = 0; k < 3; k++) { s.cdof[i][k] = 0; s.cdof[i][3 + k] = s.axis[i][k]; } }
    else {
      T off[3] = {s.com[0] - s.anchor[i][0], s.com[1] - s.anchor[i][1], s.com[2] - s.anchor[i][2]}, t[3];
      cross3(t, s.axis[i], off);
      for (int k = 0; k < 3; k++) { s.cdof[i][k] = s.axis[i][k]; s.cdof[i][3 + k] = t[k]; }
    }
  }
}

// [3P] mj_crb: composite rigid body -> M (packed tree-sparse lower triangle)
template <class T>
REX_HD void crb(const Model<T>& m, const Scratch<T>& s, MassFactor<T>& F) {
  T crbI[NBODY][10];
  static_for<0, NBODY>([&](auto BB) { constexpr int b = BB; for (int k = 0; k < 10; k++) crbI[b][k] = s.cinert[b][k]; });
  static_rfor<2, NBODY>([&](auto BB) { constexpr int b = BB, p = kBodyParent[b]; for (int k = 0; k < 10; k++) crbI[p][k] += crbI[b][k]; });
  static_for<0, NV>([&](auto II) {
    constexpr int i = II;
    T ci[6], buf[6];
    for (int k = 0; k < 6; k++) ci[k] = s.cdof[i][k];
    mul_inert(buf, crbI[kDofBody[i]], ci);
    for_anc_self<i>([&](auto JJ) {
      constexpr int j = JJ;
      T a = 0; for (int k = 0; k < 6; k++) a += s.cdof[j][k] * buf[k];
      F.a[midx(i, j)] = (i == j) ? a + m.dof_armature[i] : a;
    });
  });
}

// [3P] mj_comVel + mj_rne (flg_acc = 0): bias forces incl. gravity
template <class T>
REX_HD void com_vel_rne(const Model<T>& m, const Lane<T>& L, const T* qvel, Scratch<T>& s) {
  T cacc[NBODY][6], cfrc[NBODY][6];
  for (int k = 0; k < 6; k++) { s.cvel[0][k] = 0; cacc[0][k] = 0; }
  cacc[0][5] = m.gravity;   // -gravity: world accelerates upwards
  for (int b = 1; b < NBODY; b++) {
    int p = m.body_parent[b];
    T v[6], a[6];
    for (int k = 0; k < 6; k++) { v[k] = s.cvel[p][k]; a[k] = cacc[p][k]; }
    int da = m.body_dofadr[b], nd = m.body_dofnum[b];
    if (b == 1) {   // free joint: translations have cdofdot = 0; the three rotations use the velocity after the translations
      for (int i = 0; i < 3; i++) { for (int k = 0; k < 6; k++) { s.cdofdot[i][k] = 0; v[k] += s.cdof[i][k] * qvel[i]; } }
      for (int i = 3; i < 6; i++) cross_motion(s.cdofdot[i], v, s.cdof[i]);
      for (int i = 3; i < 6; i++) for (int k = 0; k < 6; k++) v[k] += s.cdof[i][k] * qvel[i];
    } else {
      for (int jj = 0; jj < nd; jj++) { int i = da + jj; cross_motion(s.cdofdot[i], v, s.cdof[i]); for (int k = 0; k < 6; k++) v[k] += s.cdof[i][k] * qvel[i]; }
    }
    for (int jj = 0; jj < nd; jj++) { int i = da + jj; for (int k = 0; k < 6; k++) a[k] += s.cdofdot[i][k] * qvel[i]; }
    for (int k = 0; k < 6; k++) { s.cvel[b][k] = v[k]; cacc[b][k] = a[k]; }
    T Ia[6], Iv[6], t[6];
    mul_inert(Ia, s.cinert[b], a); mul_inert(Iv, s.cinert[b], v); cross_force(t, v, Iv);
    for (int k = 0; k < 6; k++) cfrc[b][k] = Ia[k] + t[k];
  }
  for (int k = 0; k < 6; k++) cfrc[0][k] = 0;
  for (int b = NBODY - 1; b > 0; b--) { int p = m.body_parent[b]; for (int k = 0; k < 6; k++) cfrc[p][k] += cfrc[b][k]; }
  for (int i = 0; i < NV; i++) { T a = 0; const T* f = cfrc[m.dof_body[i]]; for (int k = 0; k < 6; k++) a += s.cdof[i][k] * f[k]; s.qfrc_bias[i] = a; }
}

// sparse L^T D L in place ([3P] mj_factorM): (k,k) <- 1/D_k, (k,i) <- L_ki for the ancestor dofs i of k
template <class T>
REX_HD void factor(MassFactor<T>& F) {
  static_rfor<0, NV>([&](auto KK) {
    constexpr int k = KK;
    const T inv = T(1) / F.a[midx(k, k)];
    for_anc<k>([&](auto II) {
      constexpr int i = II;
      const T a = F.a[midx(k, i)] * inv;
      for_anc_self<i>([&](auto JJ) { constexpr int j = JJ; F.a[midx(i, j)] -= a * F.a[midx(k, j)]; });
      F.a[midx(k, i)] = a;
    });
    F.a[midx(k, k)] = inv;
  });
}
template <class T>
REX_HD void solve(const MassFactor<T>& F, T (&x)[NV]) {
  static_rfor<0, NV>([&](auto KK) { constexpr int k = KK; for_anc<k>([&](auto II) { constexpr int i = II; x[i] -= F.a[midx(k, i)] * x[k]; }); });
  static_for<0, NV>([&](auto KK) { constexpr int k = KK; x[k] *= F.a[midx(k, k)]; });
  static_for<0, NV>([&](auto KK) { constexpr int k = KK; for_anc<k>([&](auto II) { constexpr int i = II; x[k] -= F.a[midx(k, i)] * x[i]; }); });
}

// ---- collision ([3P] engine_collision_primitive) -----------------------------------------------------
template <class T>
REX_HD void geom_pose(const Model<T>& m, const Scratch<T>& s, int g, T* pos, T* axis) {
  int b = m.geom_body[g]; T t[3];
  mulv(t, s.xmat[b], m.geom_pos[g]); for (int k = 0; k < 3; k++) pos[k] = s.xpos[b][k] + t[k];
  mulv(axis, s.xmat[b], m.geom_axis[g]);
}
template <class T>
REX_HD void make_frame(T* f) {   // [3P] mju_makeFrame
  T n = hsqrt(dot3(f, f)); for (int k = 0; k < 3; k++) f[k] /= n;
  if (hsqrt(dot3(f + 3, f + 3)) < T(0.5)) { f[3] = f[4] = f[5] = 0; if (f[1] < T(0.5) && f[1] > T(-0.5)) f[4] = 1; else f[5] = 1; }
  T d = dot3(f, f + 3); for (int k = 0; k < 3; k++) f[3 + k] -= d * f[k];
  T n2 = hsqrt(dot3(f + 3, f + 3));
  if (n2 < T(1e-15)) { f[3] = 1; f[4] = 0; f[5] = 0; } else for (int k = 0; k < 3; k++) f[3 + k] /= n2;
  cross3(f + 6, f, f + 3);
}
template <class T>
REX_HD void add_contact(Scratch<T>& s, const Model<T>& m, int p, T dist, const T* pos, const T* normal, const T* yaxis) {
  if (s.ncon >= MAXCON) { s.overflow = 1; return; }
  int c = s.ncon++;
  s.cdist[c] = dist; s.cdim[c] = m.pair_dim[p]; s.cmu[c] = m.pair_mu[p];
  s.cb1[c] = m.geom_body[m.pair_g1[p]]; s.cb2[c] = m.geom_body[m.pair_g2[p]];
  for (int k = 0; k < 3; k++) { s.cpos[c][k] = pos[k]; s.cframe[c][k] = normal[k]; s.cframe[c][3 + k] = yaxis ? yaxis[k] : T(0); s.cframe[c][6 + k] = 0; }
  make_frame(s.cframe[c]);
}
template <class T>
REX_HD void sphere_sphere(Scratch<T>& s, const Model<T>& m, int p, const T* c1, T r1, const T* c2, T r2) {
  T d[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
  T len = hsqrt(dot3(d, d)), dist = len - r1 - r2;
  if (dist > m.margin) return;
  T n[3] = {1, 0, 0};
  if (len >= T(1e-15)) { n[0] = d[0] / len; n[1] = d[1] / len; n[2] = d[2] / len; }
  T pos[3]; for (int k = 0; k < 3; k++) pos[k] = c1[k] + n[k] * (r1 + T(0.5) * dist);
  add_contact(s, m, p, dist, pos, n, (const T*)nullptr);
}
template <class T>
REX_HD void plane_sphere(Scratch<T>& s, const Model<T>& m, int p, const T* c, T r, const T* yaxis) {
  T n[3] = {0, 0, 1};                    // the floor: z = 0, normal +z (humanoid.xml:28)
  T dist = c[2] - r;
  if (dist > m.margin) return;
  T pos[3] = {c[0], c[1], c[2] - (r + T(0.5) * dist)};
  add_contact(s, m, p, dist, pos, n, yaxis);
}
template <class T>
REX_HD void collide(const Model<T>& m, Scratch<T>& s) {
  s.ncon = 0;
  for (int g = 1; g < NGEOM; g++) geom_pose(m, s, g, s.gpos[g], s.gaxis[g]);   // 17 poses instead of 2 per pair (126 pairs)
  for (int p = 0; p < m.npair; p++) {
    int g1 = m.pair_g1[p], g2 = m.pair_g2[p], t1 = m.geom_type[g1], t2 = m.geom_type[g2];
    T p1[3], a1[3], p2[3], a2[3];
    for (int k = 0; k < 3; k++) { p2[k] = s.gpos[g2][k]; a2[k] = s.gaxis[g2][k]; }
    T r2 = m.geom_rad[g2], l2 = m.geom_half[g2];
    if (t1 == G_PLANE) {
      if (p2[2] - r2 - l2 > m.margin) continue;                       // bounding sphere above the floor
      if (t2 == G_SPHERE) plane_sphere(s, m, p, p2, r2, (const T*)nullptr);
      else {   // [3P] mjc_PlaneCapsule: the two end spheres, frame y-axis along the capsule
        T c[3];
        for (int k = 0; k < 3; k++) c[k] = p2[k] + a2[k] * l2; plane_sphere(s, m, p, c, r2, a2);
        for (int k = 0; k < 3; k++) c[k] = p2[k] - a2[k] * l2; plane_sphere(s, m, p, c, r2, a2);
      }
      continue;
    }
    for (int k = 0; k < 3; k++) { p1[k] = s.gpos[g1][k]; a1[k] = s.gaxis[g1][k]; }
    T r1 = m.geom_rad[g1], l1 = m.geom_half[g1];
    T d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, reach = r1 + l1 + r2 + l2 + m.margin;
    if (dot3(d, d) > reach * reach) continue;                          // bounding spheres
    if (t1 == G_SPHERE && t2 == G_SPHERE) sphere_sphere(s, m, p, p1, r1, p2, r2);
    else if (t1 == G_SPHERE && t2 == G_CAPSULE) {
      T x = -(d[0] * a2[0] + d[1] * a2[1] + d[2] * a2[2]);   // (p1 - p2).a2
      x = hmin(hmax(x, -l2), l2);
      T c2[3] = {p2[0] + a2[0] * x, p2[1] + a2[1] * x, p2[2] + a2[2] * x};
      sphere_sphere(s, m, p, p1, r1, c2, r2);
    } else {   // capsule-capsule ([3P] mjc_CapsuleCapsule)
      T dif[3] = {-d[0], -d[1], -d[2]};   // p1 - p2
      T ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif), det = ma * mc - mb * mb;
      if (habs(det) >= T(1e-15)) {
        T x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
        if (x1 > l1) { x1 = l1; x2 = (v - mb * l1) / mc; } else if (x1 < -l1) { x1 = -l1; x2 = (v + mb * l1) / mc; }
        if (x2 > l2) { x2 = l2; x1 = (u - mb * l2) / ma; } else if (x2 < -l2) { x2 = -l2; x1 = (u + mb * l2) / ma; }
        if (x1 > l1) x1 = l1; else if (x1 < -l1) x1 = -l1;
        T c1[3], c2[3]; for (int k = 0; k < 3; k++) { c1[k] = p1[k] + a1[k] * x1; c2[k] = p2[k] + a2[k] * x2; }
        sphere_sphere(s, m, p, c1, r1, c2, r2);
      } else {   // parallel axes: end points of 1 against 2, then of 2 against 1 (<= 2 contacts)
        int n0 = s.ncon;
        for (int sg = -1; sg <= 1 && s.ncon - n0 < 2; sg += 2) {
          T c1[3], t[3]; for (int k = 0; k < 3; k++) { c1[k] = p1[k] + a1[k] * sg * l1; t[k] = c1[k] - p2[k]; }
          T x2 = dot3(t, a2);
          if (x2 >= -l2 && x2 <= l2) { T c2[3]; for (int k = 0; k < 3; k++) c2[k] = p2[k] + a2[k] * x2; sphere_sphere(s, m, p, c1, r1, c2, r2); }
        }
        for (int sg = -1; sg <= 1 && s.ncon - n0 < 2; sg += 2) {
          T c2[3], t[3]; for (int k = 0; k < 3; k++) { c2[k] = p2[k] + a2[k] * sg * l2; t[k] = c2[k] - p1[k]; }
          T x1 = dot3(t, a1);
          if (x1 >= -l1 && x1 <= l1) { T c1[3]; for (int k = 0; k < 3; k++) c1[k] = p1[k] + a1[k] * x1; sphere_sphere(s, m, p, c1, r1, c2, r2); }
        }
      }
    }
  }
}

template <class T>
REX_HD T impedance3(const Model<T>& m, T x_abs) {   // power 2, midpoint .5
  T x = x_abs / m.width;
  T y = x < T(0.5) ? T(2) * x * x : T(1) - T(2) * (T(1) - x) * (T(1) - x);
  T imp = m.dmin + y * (m.dmax - m.dmin);
  return x >= T(1) ? m.dmax : imp;
}

// translational Jacobian row of a world point on body b projected on direction n:  out[dof] += sign * n . Jp
template <class T>
REX_HD void jac_dir(const Model<T>& m, const Scratch<T>& s, int b, const T* p, const T* n, T sign, T* out) {
  for (; b > 0; b = m.body_parent[b]) {
    for (int jj = 0; jj < m.body_dofnum[b]; jj++) {
      int i = m.body_dofadr[b] + jj;
      if (i < 3) out[i] += sign * n[i];
      else { T r[3] = {p[0] - s.anchor[i][0], p[1] - s.anchor[i][1], p[2] - s.anchor[i][2]}, t[3]; cross3(t, s.axis[i], r); out[i] += sign * dot3(n, t); }
    }
  }
}

// [3P] mj_makeConstraint + mj_diagApprox + mj_makeImpedance + mj_referenceConstraint
template <class T>
REX_HD void make_constraints(const Model<T>& m, const T* qpos, const T* qvel, Scratch<T>& s) {
  int ne = 0;
  for (int j = 1; j < NJNT; j++) {   // hinge limits (every hinge of the humanoid is limited, humanoid.xml:4)
    T val = qpos[m.jnt_qadr[j]];
    for (int side = -1; side <= 1; side += 2) {
      T dist = side * ((side < 0 ? m.jnt_lo[j] : m.jnt_hi[j]) - val);
      if (dist < T(0) && ne < MAXEFC) {
        for (int k = 0; k < NV; k++) s.J[ne][k] = 0;
        int d = m.jnt_dadr[j]; s.J[ne][d] = T(-side);
        T imp = impedance3(m, habs(dist));
        s.R[ne] = hmax(T(1e-15), (T(1) - imp) * m.dof_invw[d] / imp);
        s.aref[ne] = -m.B * (T(-side) * qvel[d]) - m.K * imp * dist;
        ne++;
      }
    }
  }
  for (int c = 0; c < s.ncon; c++) {
    if (!(s.cdist[c] < m.margin)) continue;
    int b1 = s.cb1[c], b2 = s.cb2[c];
    T tran = m.body_invw[b1][0] + m.body_invw[b2][0];
    T imp = impedance3(m, habs(s.cdist[c] - m.margin));
    T kterm = m.K * imp * (s.cdist[c] - m.margin);
    if (s.cdim[c] == 1) {
      if (ne >= MAXEFC) { s.overflow = 1; break; }
      for (int k = 0; k < NV; k++) s.J[ne][k] = 0;
      jac_dir(m, s, b2, s.cpos[c], s.cframe[c], T(1), s.J[ne]); jac_dir(m, s, b1, s.cpos[c], s.cframe[c], T(-1), s.J[ne]);
      s.R[ne] = hmax(T(1e-15), (T(1) - imp) * tran / imp);
      T vel = 0; for (int k = 0; k < NV; k++) vel += s.J[ne][k] * qvel[k];
      s.aref[ne] = -m.B * vel - kterm;
      ne++;
    } else {   // condim 3, pyramidal: n + mu t1, n - mu t1, n + mu t2, n - mu t2
      if (ne + 4 > MAXEFC) { s.overflow = 1; break; }
      T jn[NV], jt[NV]; T mu = s.cmu[c];
      for (int k = 0; k < NV; k++) jn[k] = 0;
      jac_dir(m, s, b2, s.cpos[c], s.cframe[c], T(1), jn); jac_dir(m, s, b1, s.cpos[c], s.cframe[c], T(-1), jn);
      T R1 = hmax(T(1e-15), (T(1) - imp) * (tran + mu * mu * tran) / imp), Rpy = T(2) * mu * mu * R1;
      for (int t = 1; t <= 2; t++) {
        for (int k = 0; k < NV; k++) jt[k] = 0;
        jac_dir(m, s, b2, s.cpos[c], s.cframe[c] + 3 * t, T(1), jt); jac_dir(m, s, b1, s.cpos[c], s.cframe[c] + 3 * t, T(-1), jt);
        for (int sg = 1; sg >= -1; sg -= 2) {
          T vel = 0;
          for (int k = 0; k < NV; k++) { T v = jn[k] + sg * mu * jt[k]; s.J[ne][k] = v; vel += v * qvel[k]; }
          s.R[ne] = Rpy; s.aref[ne] = -m.B * vel - kterm;
          ne++;
        }
      }
    }
  }
  s.nefc = ne;
}

// [3P] mj_solPGS on the dual, with qacc carried along: res_i = J_i qacc - aref_i + R_i f_i
template <class T>
REX_HD int solve_pgs(const Model<T>& m, const MassFactor<T>& F, Scratch<T>& s, T* qacc) {
  for (int k = 0; k < NV; k++) qacc[k] = s.qacc_smooth[k];
  for (int i = 0; i < s.nefc; i++) {
    T x[NV], jr[NV];
    for (int k = 0; k < NV; k++) { jr[k] = s.J[i][k]; x[k] = jr[k]; }
    solve(F, x);
    T a = s.R[i];
    for (int k = 0; k < NV; k++) { s.MiJ[i][k] = x[k]; a += jr[k] * x[k]; }
    s.Adiag[i] = a; s.force[i] = 0;   // warmstart disabled (humanoid.xml:11)
  }
  const T scale = T(1) / (m.meaninertia * T(NV));
  int it = 0;
  for (; it < m.iterations; it++) {
    T improvement = 0;
    for (int i = 0; i < s.nefc; i++) {
      T res = s.R[i] * s.force[i] - s.aref[i];
      for (int k = 0; k < NV; k++) res += s.J[i][k] * qacc[k];
      T old = s.force[i], nf = hmax(T(0), old - res / s.Adiag[i]), df = nf - old;
      s.force[i] = nf;
      if (df != T(0)) for (int k = 0; k < NV; k++) qacc[k] += s.MiJ[i][k] * df;
      improvement -= T(0.5) * df * df * s.Adiag[i] + df * res;
    }
    if (improvement * scale < m.tolerance) { it++; break; }
  }
  return it;
}

#ifndef REX_STAMP
#define REX_STAMP(var) ((void)0)
#define REX_TACC(slot, t0, t1) ((void)0)
#endif

// The same Gauss-Seidel sweeps on the dual: res_i = sum_j A_ij f_j + b_i with A = J M^-1 J^T + diag(R),
// b = J qacc_smooth - aref ([3P] mj_solPGS works on exactly this matrix).  A, f and b sit in the lane's LDS column, so a
// sweep costs n^2 LDS reads instead of 2 n nv reads of J / M^-1 J^T rows from scratch (which miss every cache level);
// J is read once per row pair to build A and once more for qacc = qacc_smooth + M^-1 J^T f.
template <class T>
REX_HD int solve_pgs_dual(const Model<T>& m, const MassFactor<T>& F, Scratch<T>& s, T* qacc) {
  const int n = s.nefc;
  T qs[NV];
  REX_STAMP(p0);
  for (int k = 0; k < NV; k++) qs[k] = s.qacc_smooth[k];
  for (int j = 0; j < n; j++) {
    T x[NV], jr[NV];
    for (int k = 0; k < NV; k++) { jr[k] = s.J[j][k]; x[k] = jr[k]; }
    solve(F, x);
    T b = -s.aref[j]; for (int k = 0; k < NV; k++) b += jr[k] * qs[k];
    dual(s, DUAL_B + j) = b; dual(s, DUAL_F + j) = T(0);   // warmstart disabled (humanoid.xml:11)
    const int tj = j * (j + 1) / 2;
    for (int i = 0; i < j; i++) { T a = 0; for (int k = 0; k < NV; k++) a += s.J[i][k] * x[k]; dual(s, tj + i) = a; }
    T a = s.R[j]; for (int k = 0; k < NV; k++) a += jr[k] * x[k];
    dual(s, tj + j) = a;
  }
  REX_STAMP(p1); REX_TACC(20, p0, p1);
  const T scale = T(1) / (m.meaninertia * T(NV));
  int it = 0;
  for (; it < m.iterations; it++) {
    T improvement = 0;
    for (int i = 0; i < n; i++) {
      const int ti = i * (i + 1) / 2;
      T res = dual(s, DUAL_B + i);
      for (int j = 0; j <= i; j++) res += dual(s, ti + j) * dual(s, DUAL_F + j);
      for (int j = i + 1; j < n; j++) res += dual(s, j * (j + 1) / 2 + i) * dual(s, DUAL_F + j);
      const T ad = dual(s, ti + i), old = dual(s, DUAL_F + i), nf = hmax(T(0), old - res / ad), df = nf - old;
      dual(s, DUAL_F + i) = nf;
      improvement -= T(0.5) * df * df * ad + df * res;
    }
    if (improvement * scale < m.tolerance) { it++; break; }
  }
  REX_STAMP(p2); REX_TACC(21, p1, p2);
  T x[NV];
  for (int k = 0; k < NV; k++) x[k] = 0;
  for (int i = 0; i < n; i++) { const T f = dual(s, DUAL_F + i); s.force[i] = f; if (f != T(0)) for (int k = 0; k < NV; k++) x[k] += s.J[i][k] * f; }
  solve(F, x);
  for (int k = 0; k < NV; k++) qacc[k] = qs[k] + x[k];
  REX_STAMP(p3); REX_TACC(22, p2, p3);
  return it;
}

template <class T>
struct ForwardOut { T xipos_x[NBODY]; };

// [3P] mj_forward
template <class T>
REX_HD int forward(const Model<T>& m, const Lane<T>& L, const T* qpos, const T* qvel, const T* ctrl, Scratch<T>& s, T* qacc) {
  s.overflow = 0;
  REX_STAMP(t0);
  kinematics(m, qpos, s);
  REX_STAMP(t1); REX_TACC(8, t0, t1);
  com_pos(m, L, s);
  REX_STAMP(t2); REX_TACC(9, t1, t2);
  com_vel_rne(m, L, qvel, s);
  for (int i = 0; i < NV; i++) s.qfrc_actuator[i] = 0;
  for (int u = 0; u < NU; u++) { T c = hmin(hmax(ctrl[u], T(-0.4)), T(0.4)); s.qfrc_actuator[m.act_dof[u]] += m.act_gear[u] * c; }   // ctrlrange, humanoid.xml:6
  for (int i = 0; i < NV; i++) s.qfrc_smooth[i] = -L.damping[i] * qvel[i] - s.qfrc_bias[i] + s.qfrc_actuator[i];
  for (int j = 1; j < NJNT; j++) s.qfrc_smooth[m.jnt_dadr[j]] -= m.jnt_stiff[j] * qpos[m.jnt_qadr[j]];   // springref 0
  REX_STAMP(t3); REX_TACC(11, t2, t3);
  collide(m, s);
  REX_STAMP(t4); REX_TACC(12, t3, t4);
  make_constraints(m, qpos, qvel, s);
  REX_STAMP(t5); REX_TACC(13, t4, t5);
  // the mass matrix and its factor are built last so that their 185 registers are live only from here on
  MassFactor<T> F;
  crb(m, s, F);
  REX_STAMP(t6); REX_TACC(10, t5, t6);
  factor(F);
  { T x[NV]; for (int i = 0; i < NV; i++) x[i] = s.qfrc_smooth[i]; solve(F, x); for (int i = 0; i < NV; i++) s.qacc_smooth[i] = x[i]; }
  REX_STAMP(t7); REX_TACC(14, t6, t7);
  if (s.nefc == 0) { for (int i = 0; i < NV; i++) qacc[i] = s.qacc_smooth[i]; return 0; }
  int it = s.nefc <= DUAL_NMAX ? solve_pgs_dual(m, F, s, qacc) : solve_pgs(m, F, s, qacc);   // the scratch-row variant only for rare pile-ups
  REX_STAMP(t8); REX_TACC(15, t7, t8); REX_TACC(16, t0, t8);
#if defined(REX_KTIME) && defined(__HIP_DEVICE_COMPILE__)
  if ((threadIdx.x & 63) == 0) { atomicAdd(&g_ktime[17], 1ull); atomicAdd(&g_ktime[18], (unsigned long long)s.nefc); atomicAdd(&g_ktime[19], (unsigned long long)it); }
#endif
  return it;
}

// [3P] mj_integratePos
template <class T>
REX_HD void integrate_pos(T* qpos, const T* qvel, T h) {
  for (int k = 0; k < 3; k++) qpos[k] += h * qvel[k];
  T w[3] = {qvel[3], qvel[4], qvel[5]}, n = hsqrt(dot3(w, w));
  if (n * h > T(1e-15)) {
    T sn, cs; hsincos(T(0.5) * n * h, sn, cs);
    T dq[4] = {cs, w[0] / n * sn, w[1] / n * sn, w[2] / n * sn}, r[4];
    qmul(r, qpos + 3, dq); qnorm(r);
    for (int k = 0; k < 4; k++) qpos[3 + k] = r[k];
  }
  for (int k = 0; k < 17; k++) qpos[7 + k] += h * qvel[6 + k];
}

// One mj_step with RK4 ([3P] mj_RungeKutta, N = 4).  `s` keeps the quantities of the LAST forward
// evaluation (stage 4), which is what the reference's observation / reward read (random_humanoid.py:161-216).
template <class T>
REX_HD void substep(const Model<T>& m, const Lane<T>& L, T* qpos, T* qvel, const T* ctrl, Scratch<T>& s) {
  const T h = m.timestep;
  T q0[NQ], v0[NV], dq[NV], dv[NV], acc[NV];
  for (int k = 0; k < NQ; k++) q0[k] = qpos[k];
  for (int k = 0; k < NV; k++) { v0[k] = qvel[k]; dq[k] = 0; dv[k] = 0; }
  for (int stage = 0; stage < 4; stage++) {
    forward(m, L, qpos, qvel, ctrl, s, acc);
    const T w = (stage == 0 || stage == 3) ? T(1.0 / 6) : T(1.0 / 3), c = stage == 2 ? h : T(0.5) * h;
    for (int k = 0; k < NV; k++) { dq[k] += w * qvel[k]; dv[k] += w * acc[k]; }
    if (stage < 3) {
      T vs[NV]; for (int k = 0; k < NV; k++) vs[k] = qvel[k];
      for (int k = 0; k < NQ; k++) qpos[k] = q0[k];
      integrate_pos(qpos, vs, c);
      for (int k = 0; k < NV; k++) qvel[k] = v0[k] + c * acc[k];
    } else {
      for (int k = 0; k < NQ; k++) qpos[k] = q0[k];
      for (int k = 0; k < NV; k++) qvel[k] = v0[k] + h * dv[k];
      integrate_pos(qpos, dq, h);
    }
  }
}

// RandomHumanoidEnv.step + _get_obs (random_humanoid.py:161-216), noise-free.
//   xipos_x: in = data.xipos[:,0] left by the previous forward (mass_center() reads it before do_simulation);
//            out = the same after this step (stage-4 forward of the last mj_step).
//   obs(k, value) is called for k = 0..375 in order.
template <class T, class ObsSink>
REX_HD void env_step(const Model<T>& m, const Lane<T>& L, T* qpos, T* qvel, const T* action, T* xipos_x, Scratch<T>& s,
                     T& reward, bool& done, ObsSink&& obs) {
  T mt = 0, s0 = 0, s1 = 0, asq = 0;
  for (int b = 0; b < NBODY; b++) { mt += L.mass[b]; s0 += L.mass[b] * xipos_x[b]; }
  for (int u = 0; u < NU; u++) asq += action[u] * action[u];       // data.ctrl holds the raw action (:167)
  for (int f = 0; f < 5; f++) substep(m, L, qpos, qvel, action, s);   // frame_skip 5 (:41)
  for (int b = 0; b < NBODY; b++) { xipos_x[b] = s.xipos[b][0]; s1 += L.mass[b] * s.xipos[b][0]; }
  const T dt = m.timestep * T(5);
  reward = T(1.25) * (s1 / mt - s0 / mt) / dt - T(0.1) * asq - T(0) /* cfrc_ext = 0, SURVEY Q15 */ + T(5);
  done = (qpos[2] < T(1.0)) || (qpos[2] > T(2.0));                 // :173
  int c = 0;
  for (int k = 2; k < NQ; k++) obs(c++, qpos[k]);
  for (int k = 0; k < NV; k++) obs(c++, qvel[k]);
  for (int b = 0; b < NBODY; b++) for (int k = 0; k < 10; k++) obs(c++, s.cinert[b][k]);
  for (int b = 0; b < NBODY; b++) for (int k = 0; k < 6; k++) obs(c++, s.cvel[b][k]);
  for (int k = 0; k < NV; k++) obs(c++, s.qfrc_actuator[k]);
  for (int k = 0; k < 84; k++) obs(c++, T(0));
}

// observation right after set_state / reset: sim.forward() at the given state (jinja_mujoco_env.py:146-154)
template <class T, class ObsSink>
REX_HD void env_reset_obs(const Model<T>& m, const Lane<T>& L, const T* qpos, const T* qvel, T* xipos_x, Scratch<T>& s, ObsSink&& obs) {
  T ctrl[NU], acc[NV];
  for (int u = 0; u < NU; u++) ctrl[u] = 0;                        // sim.reset() zeroes data.ctrl
  forward(m, L, qpos, qvel, ctrl, s, acc);
  for (int b = 0; b < NBODY; b++) xipos_x[b] = s.xipos[b][0];
  int c = 0;
  for (int k = 2; k < NQ; k++) obs(c++, qpos[k]);
  for (int k = 0; k < NV; k++) obs(c++, qvel[k]);
  for (int b = 0; b < NBODY; b++) for (int k = 0; k < 10; k++) obs(c++, s.cinert[b][k]);
  for (int b = 0; b < NBODY; b++) for (int k = 0; k < 6; k++) obs(c++, s.cvel[b][k]);
  for (int k = 0; k < NV; k++) obs(c++, s.qfrc_actuator[k]);
  for (int k = 0; k < 84; k++) obs(c++, T(0));
}

}  // namespace hum
}  // namespace rex
