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
#include <type_traits>

#include "planar_spec.hpp"   // REX_HD

// Diagnostic build (-DREX_KTIME): s_memtime deltas summed in per-lane registers (Kin::tacc) and flushed once per kernel by
// the caller (wave maximum per slot), so the probes do not perturb what they measure.
#if defined(REX_KTIME) && defined(__HIP_DEVICE_COMPILE__)
#define REX_HSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define REX_HACC(K, slot, t0, t1) ((K).tacc[slot] += (t1) - (t0))
#define REX_HCNT(K, slot, v) ((K).tacc[slot] += (unsigned long long)(v))
#else
#define REX_HSTAMP(var) ((void)0)
#define REX_HACC(K, slot, t0, t1) ((void)0)
#define REX_HCNT(K, slot, v) ((void)0)
#endif
enum { HT_SMOOTH = 0, HT_LIMITS, HT_BROAD, HT_NARROW_LOOP, HT_PAIR, HT_ROWS, HT_FACTOR, HT_BUILD_A, HT_SWEEPS, HT_QACC, HT_FORWARD,
       HC_EVALS, HC_PAIR_CALLS, HC_ROW_CALLS, HC_SWEEPS, HC_NEFC, HT_SLOTS };

namespace rex {
namespace hum {

constexpr int NBODY = 14;   // incl. world
constexpr int NJNT = 18;    // free + 17 hinges
constexpr int NQ = 24, NV = 23, NU = 17;
constexpr int NGEOM = 18;   // floor + 17 body geoms
constexpr int MAXPAIR = 128;
// Row storage is sized to what the MODEL can produce, so nothing is ever dropped in practice (the reference's MuJoCo keeps
// every row: its own caps, nconmax 100 / njmax 500 by default, are out of this model's reach as well): 17 limit rows +
// 33 floor contacts (16 capsules x 2 ends + the head sphere) x 4 pyramid rows + a few capsule-capsule rows.  A lying,
// crumpled humanoid (tests/test_gpu_humanoid.py::test_pile_up_states) reaches 26 contacts / 108 rows.  Rows beyond the
// arrays would still be dropped and counted (`overflow`); the tests require the counter to stay 0.
constexpr int MAXCON = 64;   // contacts per evaluation (a counter on the device; the contact log exists on the host only)
constexpr int MAXEFC = 192;  // constraint rows per evaluation
constexpr int NXI = 30, NOBS = 376;
enum { G_PLANE = 0, G_SPHERE = 2, G_CAPSULE = 3 };
// dual-space PGS working set kept in LDS on the device (one contiguous column per lane): A = J M^-1 J^T + diag(R) for up to
// DUAL_NMAX rows (0.09% of the evaluations of a random-policy batch have more than 16 rows, none more than 21) -- square and
// row-major up to 16 rows (pgs_sweeps_sq), the packed lower triangle above --, then b = J qacc_smooth - aref and 1 / A_ii
constexpr int DUAL_NMAX = 21;
constexpr int tri(int i) { return i * (i + 1) / 2; }
constexpr int DUAL_B = tri(DUAL_NMAX), DUAL_DI = DUAL_B + DUAL_NMAX;
constexpr int DUAL_WORDS = 292;   // >= DUAL_DI + DUAL_NMAX and the geometry overlay; 4 blocks of 32 lanes fill a CU's 160 KB
static_assert(DUAL_WORDS >= DUAL_DI + DUAL_NMAX && DUAL_WORDS % 8 == 4, "LDS column layout");


// Compile-time dof tree of humanoid.xml (free root 0-5, abdomen 6-8, right leg 9-12, left leg 13-16, right arm 17-19,
// left arm 20-22).  The mass-matrix code below is generated from these tables (static indices => registers, no
// pointer chasing through the model); build_model() derives the same tables from the XML transcription and
// check_topology() compares the two.
constexpr int kDofParent[NV] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 8, 13, 14, 15, 5, 17, 18, 5, 20, 21};
constexpr int kDofBody[NV] = {1, 1, 1, 1, 1, 1, 2, 2, 3, 4, 4, 4, 5, 7, 7, 7, 8, 10, 10, 11, 12, 12, 13};
constexpr int kBodyParent[NBODY] = {0, 0, 1, 2, 3, 4, 5, 3, 7, 8, 1, 10, 1, 12};
constexpr int kBodyDofAdr[NBODY] = {0, 0, 6, 8, 9, 12, 0, 13, 16, 0, 17, 19, 20, 22};   // hinge dof d belongs to joint d - 5, qpos d + 1
constexpr int kBodyDofNum[NBODY] = {0, 6, 2, 1, 3, 1, 0, 3, 1, 0, 2, 1, 2, 1};
constexpr int kGeomBody[NGEOM] = {0, 1, 1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 11, 12, 13, 13};
constexpr int kGeomType[NGEOM] = {0, 3, 2, 3, 3, 3, 3, 3, 2, 3, 3, 2, 3, 3, 2, 3, 3, 2};   // G_PLANE 0, G_SPHERE 2, G_CAPSULE 3
// Upper bounds (rounded up in the 6th decimal) of the geoms' radii and capsule half-lengths, humanoid.xml:33-88: literals in
// the broad phase's second test, which only has to be conservative.  check_topology() holds them against the compiled model.
constexpr float kGeomRadUB[NGEOM] = {0.f, .070001f, .090001f, .060001f, .060001f, .090001f, .060001f, .049001f, .075001f, .060001f, .049001f, .075001f,
                                     .040001f, .031001f, .040001f, .040001f, .031001f, .040001f};
constexpr float kGeomHalfUB[NGEOM] = {0.f, .070001f, 0.f, .060001f, .060001f, .070001f, .170075f, .150001f, 0.f, .170075f, .150001f, 0.f,
                                      .138566f, .138566f, 0.f, .138566f, .138566f, 0.f};
// candidate geom pairs in MuJoCo's order ([3P] mj_collision): body pairs ascending, geoms of body 1 x geoms of body 2,
// no pair inside one weld group (the feet have no joint: they belong to the shins) or between parent and child groups
struct PairTable { int n; int g1[MAXPAIR], g2[MAXPAIR]; };
constexpr PairTable make_pair_table() {
  PairTable t{};
  int weld[NBODY] = {};
  for (int b = 1; b < NBODY; b++) weld[b] = kBodyDofNum[b] ? b : weld[kBodyParent[b]];
  for (int b1 = 0; b1 < NBODY; b1++) for (int b2 = b1 + 1; b2 < NBODY; b2++) {
    const int w1 = weld[b1], w2 = weld[b2], wp1 = weld[kBodyParent[w1]], wp2 = weld[kBodyParent[w2]];
    if (w1 == w2) continue;
    if (w1 != 0 && w2 != 0 && (w1 == wp2 || w2 == wp1)) continue;
    for (int g1 = 0; g1 < NGEOM; g1++) for (int g2 = 0; g2 < NGEOM; g2++) {
      if (kGeomBody[g1] != b1 || kGeomBody[g2] != b2) continue;
      int a = g1, b = g2; if (kGeomType[a] > kGeomType[b]) { int x = a; a = b; b = x; }
      t.g1[t.n] = a; t.g2[t.n] = b; t.n++;
    }
  }
  return t;
}
constexpr PairTable kPairs = make_pair_table();
constexpr int kActDof[NU] = {7, 6, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22};   // motors humanoid.xml:106-122
constexpr int body_dof_mask(int b) {   // dofs that move body b
  int mask = 0;
  for (; b > 0; b = kBodyParent[b]) for (int k = 0; k < kBodyDofNum[b]; k++) mask |= 1 << (kBodyDofAdr[b] + k);
  return mask;
}
constexpr int dof_depth(int d) { int n = 0; while (kDofParent[d] >= 0) { d = kDofParent[d]; n++; } return n; }
constexpr int m_row(int i) { int o = 0; for (int k = 0; k < i; k++) o += dof_depth(k) + 1; return o; }
constexpr int MNNZ = m_row(NV);   // 185 = entries (i, j) with j an ancestor-or-self dof of i
constexpr int midx(int i, int j) { return m_row(i) + dof_depth(j); }   // valid when j is an ancestor-or-self of i
constexpr bool dof_is_anc_or_self(int j, int i) { while (i >= 0) { if (i == j) return true; i = kDofParent[i]; } return false; }
template <int I, class F> REX_HD void for_anc(F&& f) {          // f(IC<j>) for every proper ancestor dof j of I, nearest first
  if constexpr (kDofParent[I] >= 0) { f(IC<kDofParent[I]>{}); for_anc<kDofParent[I]>(f); }
}
template <int I, class F> REX_HD void for_anc_self(F&& f) { f(IC<I>{}); for_anc<I>(f); }

// everything the pair loop needs about one candidate geom pair, as one record (a single wave-uniform load)
template <class T>
struct alignas(64) PairRec { int g1, g2, t1, t2, dim, mask1, mask2, b1, b2; T mu, r1, l1, r2, l2, tran; int pad; };

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
  T geom_bound[NGEOM];          // bounding-sphere radius: rad + half length
  int npair, pair_g1[MAXPAIR], pair_g2[MAXPAIR], pair_dim[MAXPAIR];
  T pair_mu[MAXPAIR];
  PairRec<T> pair[MAXPAIR];     // the same pairs, packed for the device loop (fill_pair_records)
  int act_dof[NU]; T act_gear[NU];
  T qpos0[NQ];
  // solver constants: contacts and limits share solref (.02,1); solimp = MuJoCo default (.9,.95,.001)
  T K, B, dmin, dmax, width, margin, timestep, gravity, meaninertia, tolerance;
  int iterations;
};

template <class T> REX_HD T hsqrt(T a);
template <> REX_HD float hsqrt<float>(float a) { return sqrtf(a); }
template <> REX_HD double hsqrt<double>(double a) { return sqrt(a); }
template <class T> REX_HD T fast_sqrt(T a) { return hsqrt(a); }   // where a bound, not a result, is computed
#if defined(__HIP_DEVICE_COMPILE__)
template <> REX_HD float fast_sqrt<float>(float a) { return __builtin_amdgcn_sqrtf(a); }   // one v_sqrt_f32 (1 ulp), no denormal fix-up
#endif
template <class T> REX_HD T habs(T a) { return a < T(0) ? -a : a; }
template <class T> REX_HD T hmax(T a, T b) { return a > b ? a : b; }
template <class T> REX_HD T hmin(T a, T b) { return a < b ? a : b; }
REX_HD void hsincos(float a, float& s, float& c) {
#if defined(REX_LIBM_SINCOS) && defined(__HIP_DEVICE_COMPILE__)
  sincosf(a, &s, &c);
#elif defined(REX_LIBM_SINCOS)
  s = sinf(a); c = cosf(a);
#else
  sincos_poly(a, s, c);
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
  if (n < T(1e-15)) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else { T i = rcp_t(n); q[0] *= i; q[1] *= i; q[2] *= i; q[3] *= i; }
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

// Smooth-dynamics working set of one evaluation (kinematics, then -- after the collision phase, forward() -- com -> RNE ->
// CRB).  Every access uses compile-time indices, so it lives in registers; only the body frames (xmat, xipos) are alive across
// the collision phase, and what the observation needs later (cinert, cvel, xipos_x, qfrc_actuator of the LAST evaluation,
// random_humanoid.py:193-204) is parked in Scratch, not kept live across the solver.
template <class T>
struct Smooth {
  T xmat[NBODY][9], xipos[NBODY][3];
  T com[3];                       // subtree COM of the root (MuJoCo's reference point)
  T cinert[NBODY][10], cvel[NBODY][6], cdof[NV][6];
};

// what survives the smooth phase inside one evaluation (registers)
template <class T>
struct Kin {
  T qfrc_smooth[NV], qacc_smooth[NV];
  int ncon, nefc, overflow;       // contacts / constraint rows of this evaluation; overflow: some were dropped (MAXCON / MAXEFC)
#if defined(REX_KTIME)
  unsigned long long tacc[HT_SLOTS];
#endif
};

// host stand-in for the lane's LDS column (empty for the fp32 device lanes)
template <class T> struct HostColumn { alignas(16) T w[DUAL_WORDS]; };
#if defined(__HIP_DEVICE_COMPILE__)
template <> struct HostColumn<float> {};
#endif

template <class T>
struct Scratch {   // runtime-indexed per-lane arrays (HIP scratch): contacts and constraint rows
#if !defined(__HIP_DEVICE_COMPILE__)
  T cpos[MAXCON][3], cdist[MAXCON]; int cdim[MAXCON], cb1[MAXCON], cb2[MAXCON];   // contact log: host builds only (tests); the engine never reads it
#endif
  // (the dual path, <= DUAL_NMAX rows, overwrites row j of J with z_j = D^-1 L^-T J_j^T while it builds A; MiJ / Adiag are
  // the scratch-row PGS's, used for more rows than that)
  T J[MAXEFC][NV], MiJ[MAXEFC][NV], R[MAXEFC], aref[MAXEFC], Adiag[MAXEFC], force[MAXEFC];
  // parked between phases (plain stores / loads at fixed offsets; keeps them out of the register file while the solver runs)
  T obs_cinert[NBODY][10], obs_cvel[NBODY][6], obs_xipos_x[NBODY], obs_qfrc_actuator[NV];   // observation inputs
  T rk_q0[NQ], rk_v0[NV], rk_dq[NV], rk_dv[NV];                                              // RK4 accumulators
  HostColumn<T> col;
};

// Per-lane LDS column (DUAL_WORDS contiguous words); a plain array on the host.  It is used
// twice per evaluation: first for the runtime-indexed geometry (geom poses for the pair loop, joint anchors / axes
// for the contact Jacobians), then -- once the rows exist -- for the dual PGS working set.
#if defined(__HIP_DEVICE_COMPILE__)
extern __shared__ __attribute__((aligned(16))) float hum_lds[];
REX_HD float& dual(Scratch<float>&, int k) { return hum_lds[threadIdx.x * DUAL_WORDS + k]; }
inline double& dual(Scratch<double>& s, int k) { return s.col.w[k]; }   // the fp64 model compiler (host code seen by the device pass)
#else
template <class T> REX_HD T& dual(Scratch<T>& s, int k) { return s.col.w[k]; }
#endif
constexpr int GEO_GEOM = 0;                        // (g - 1) * 6 + {pos 0..2, axis 3..5}, g = 1..17
constexpr int GEO_DOF = (NGEOM - 1) * 6;           // i * 6 + {anchor 0..2, axis 3..5}, i = 0..22
constexpr int HITQ_BASE = GEO_DOF + NV * 6, HITQ_WORDS = 8, HITQ_MAX = 6;   // queued narrow-phase hits: dist, pos, normal, pair index
static_assert(HITQ_BASE + HITQ_WORDS * HITQ_MAX <= DUAL_WORDS, "geometry overlay + hit queue must fit the LDS column");

// Tree-sparse joint-space inertia, packed (midx).  After factor(): L^T D L in place with the diagonal holding 1/D.
// Every index into it is a compile-time constant, so on the device it lives in registers (AGPRs as overflow).
template <class T>
struct MassFactor {
  T a[MNNZ];
  REX_HD T get(int i, int j) const { return dof_is_anc_or_self(j, i) ? a[midx(i, j)] : (dof_is_anc_or_self(i, j) ? a[midx(j, i)] : T(0)); }   // tests
};

// [3P] mj_kinematics, unrolled over the fixed tree.  Leaves xmat / xipos in K and the runtime-indexed geometry
// (geom poses, dof anchors / axes) in the lane's LDS column.
template <class T>
REX_HD void kinematics(const Model<T>& m, const T* qpos, Smooth<T>& K, Scratch<T>& s) {
  T xp[NBODY][3], xq[NBODY][4];
  for (int k = 0; k < 3; k++) { xp[0][k] = 0; K.xipos[0][k] = 0; }
  xq[0][0] = 1; xq[0][1] = xq[0][2] = xq[0][3] = 0;
  q2mat(K.xmat[0], xq[0]);
  static_for<1, NBODY>([&](auto BB) {
    constexpr int b = BB, p = kBodyParent[b];
    T xpos[3], xquat[4], t[3], R[9];
    if constexpr (b == 1) {   // free joint of the torso
      for (int k = 0; k < 3; k++) xpos[k] = qpos[k];
      for (int k = 0; k < 4; k++) xquat[k] = qpos[3 + k];
      qnorm(xquat); q2mat(R, xquat);
      for (int k = 0; k < 3; k++) for (int x = 0; x < 3; x++) {
        dual(s, GEO_DOF + k * 6 + x) = xpos[x]; dual(s, GEO_DOF + k * 6 + 3 + x) = (x == k) ? T(1) : T(0);
        dual(s, GEO_DOF + (3 + k) * 6 + x) = xpos[x]; dual(s, GEO_DOF + (3 + k) * 6 + 3 + x) = R[3 * x + k];
      }
    } else {
      mulv(t, K.xmat[p], m.body_pos[b]);
      for (int k = 0; k < 3; k++) xpos[k] = xp[p][k] + t[k];
      qmul(xquat, xq[p], m.body_quat[b]);
      static_for<0, kBodyDofNum[b]>([&](auto JJ) {
        constexpr int da = kBodyDofAdr[b] + JJ, j = da - 5, qa = da + 1;
        q2mat(R, xquat);
        T anchor[3], axis[3];
        mulv(t, R, m.jnt_pos[j]); for (int k = 0; k < 3; k++) anchor[k] = xpos[k] + t[k];
        mulv(axis, R, m.jnt_axis[j]);
        T ang = qpos[qa] - m.qpos0[qa], sn, cs; hsincos(T(0.5) * ang, sn, cs);
        T ql[4] = {cs, m.jnt_axis[j][0] * sn, m.jnt_axis[j][1] * sn, m.jnt_axis[j][2] * sn}, nq[4];
        qmul(nq, xquat, ql); for (int k = 0; k < 4; k++) xquat[k] = nq[k];
        qnorm(xquat); q2mat(R, xquat);
        mulv(t, R, m.jnt_pos[j]); for (int k = 0; k < 3; k++) xpos[k] = anchor[k] - t[k];
        for (int k = 0; k < 3; k++) { dual(s, GEO_DOF + da * 6 + k) = anchor[k]; dual(s, GEO_DOF + da * 6 + 3 + k) = axis[k]; }
      });
    }
    for (int k = 0; k < 3; k++) xp[b][k] = xpos[k];
    for (int k = 0; k < 4; k++) xq[b][k] = xquat[k];
    q2mat(K.xmat[b], xquat);
    mulv(t, K.xmat[b], m.body_ipos[b]); for (int k = 0; k < 3; k++) K.xipos[b][k] = xpos[k] + t[k];
  });
  static_for<1, NGEOM>([&](auto GG) {   // geom poses for the pair loop
    constexpr int g = GG, b = kGeomBody[g];
    T t[3], a[3];
    mulv(t, K.xmat[b], m.geom_pos[g]); mulv(a, K.xmat[b], m.geom_axis[g]);
    for (int k = 0; k < 3; k++) { dual(s, GEO_GEOM + (g - 1) * 6 + k) = xp[b][k] + t[k]; dual(s, GEO_GEOM + (g - 1) * 6 + 3 + k) = a[k]; }
  });
}

// [3P] mj_comPos: reference point, cinert, cdof
template <class T>
REX_HD void com_pos(const Model<T>& m, const Lane<T>& L, Smooth<T>& K, Scratch<T>& s) {
  T sc[3] = {0, 0, 0};
  static_for<1, NBODY>([&](auto BB) { constexpr int b = BB; for (int k = 0; k < 3; k++) sc[k] += L.mass[b] * K.xipos[b][k]; });
  for (int k = 0; k < 3; k++) K.com[k] = sc[k] / m.subtreemass_root;   // compile-time subtree mass (Q4-style staleness)
  for (int k = 0; k < 10; k++) K.cinert[0][k] = 0;
  static_for<1, NBODY>([&](auto BB) {
    constexpr int b = BB;
    const T* R = K.xmat[b]; const T* I = m.body_inertia[b];
    T Ib[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}, RI[9], Iw[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T a = 0; for (int k = 0; k < 3; k++) a += R[3 * i + k] * Ib[3 * k + j]; RI[3 * i + j] = a; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T a = 0; for (int k = 0; k < 3; k++) a += RI[3 * i + k] * R[3 * j + k]; Iw[3 * i + j] = a; }
    T d[3] = {K.xipos[b][0] - K.com[0], K.xipos[b][1] - K.com[1], K.xipos[b][2] - K.com[2]}, ms = L.mass[b];
    T* c = K.cinert[b];
    c[0] = Iw[0] + ms * (d[1] * d[1] + d[2] * d[2]); c[1] = Iw[4] + ms * (d[0] * d[0] + d[2] * d[2]); c[2] = Iw[8] + ms * (d[0] * d[0] + d[1] * d[1]);
    c[3] = Iw[1] - ms * d[0] * d[1]; c[4] = Iw[2] - ms * d[0] * d[2]; c[5] = Iw[5] - ms * d[1] * d[2];
    c[6] = ms * d[0]; c[7] = ms * d[1]; c[8] = ms * d[2]; c[9] = ms;
  });
  static_for<0, NV>([&](auto II) {
    constexpr int i = II;
    T anchor[3], axis[3];
    for (int k = 0; k < 3; k++) { anchor[k] = dual(s, GEO_DOF + i * 6 + k); axis[k] = dual(s, GEO_DOF + i * 6 + 3 + k); }
    if constexpr (i < 3) { for (int k = 0; k < 3; k++) { K.cdof[i][k] = 0; K.cdof[i][3 + k] = axis[k]; } }
    else {
      T off[3] = {K.com[0] - anchor[0], K.com[1] - anchor[1], K.com[2] - anchor[2]}, t[3];
      cross3(t, axis, off);
      for (int k = 0; k < 3; k++) { K.cdof[i][k] = axis[k]; K.cdof[i][3 + k] = t[k]; }
    }
  });
}

// [3P] mj_crb: composite rigid body -> M (packed tree-sparse lower triangle)
template <class T>
REX_HD void crb(const Model<T>& m, const Smooth<T>& K, MassFactor<T>& F) {
  T crbI[NBODY][10];
  static_for<0, NBODY>([&](auto BB) { constexpr int b = BB; for (int k = 0; k < 10; k++) crbI[b][k] = K.cinert[b][k]; });
  static_rfor<2, NBODY>([&](auto BB) { constexpr int b = BB, p = kBodyParent[b]; for (int k = 0; k < 10; k++) crbI[p][k] += crbI[b][k]; });
  static_for<0, NV>([&](auto II) {
    constexpr int i = II;
    T buf[6]; mul_inert(buf, crbI[kDofBody[i]], K.cdof[i]);
    for_anc_self<i>([&](auto JJ) {
      constexpr int j = JJ;
      T a = 0; for (int k = 0; k < 6; k++) a += K.cdof[j][k] * buf[k];
      F.a[midx(i, j)] = (i == j) ? a + m.dof_armature[i] : a;
    });
  });
}

// [3P] mj_comVel + mj_rne (flg_acc = 0): bias forces incl. gravity
template <class T>
REX_HD void com_vel_rne(const Model<T>& m, const Lane<T>& L, const T* qvel, Smooth<T>& K, T (&qfrc_bias)[NV]) {
  T cacc[NBODY][6], cfrc[NBODY][6], cdofdot[NV][6];
  for (int k = 0; k < 6; k++) { K.cvel[0][k] = 0; cacc[0][k] = 0; cfrc[0][k] = 0; }
  cacc[0][5] = m.gravity;   // -gravity: world accelerates upwards
  static_for<1, NBODY>([&](auto BB) {
    constexpr int b = BB, p = kBodyParent[b], da = kBodyDofAdr[b], nd = kBodyDofNum[b];
    T v[6], a[6];
    for (int k = 0; k < 6; k++) { v[k] = K.cvel[p][k]; a[k] = cacc[p][k]; }
    if constexpr (b == 1) {   // free joint: translations have cdofdot = 0; the three rotations use the velocity after the translations
      for (int i = 0; i < 3; i++) { for (int k = 0; k < 6; k++) { cdofdot[i][k] = 0; v[k] += K.cdof[i][k] * qvel[i]; } }
      for (int i = 3; i < 6; i++) cross_motion(cdofdot[i], v, K.cdof[i]);
      for (int i = 3; i < 6; i++) for (int k = 0; k < 6; k++) v[k] += K.cdof[i][k] * qvel[i];
    } else {
      static_for<0, nd>([&](auto JJ) { constexpr int i = da + JJ; cross_motion(cdofdot[i], v, K.cdof[i]); for (int k = 0; k < 6; k++) v[k] += K.cdof[i][k] * qvel[i]; });
    }
    static_for<0, nd>([&](auto JJ) { constexpr int i = da + JJ; for (int k = 0; k < 6; k++) a[k] += cdofdot[i][k] * qvel[i]; });
    for (int k = 0; k < 6; k++) { K.cvel[b][k] = v[k]; cacc[b][k] = a[k]; }
    T Ia[6], Iv[6], t[6];
    mul_inert(Ia, K.cinert[b], a); mul_inert(Iv, K.cinert[b], v); cross_force(t, v, Iv);
    for (int k = 0; k < 6; k++) cfrc[b][k] = Ia[k] + t[k];
  });
  static_rfor<1, NBODY>([&](auto BB) { constexpr int b = BB, p = kBodyParent[b]; for (int k = 0; k < 6; k++) cfrc[p][k] += cfrc[b][k]; });
  static_for<0, NV>([&](auto II) { constexpr int i = II; T a = 0; for (int k = 0; k < 6; k++) a += K.cdof[i][k] * cfrc[kDofBody[i]][k]; qfrc_bias[i] = a; });
}

// sparse L^T D L in place ([3P] mj_factorM): (k,k) <- 1/D_k, (k,i) <- L_ki for the ancestor dofs i of k
template <class T>
REX_HD void factor(MassFactor<T>& F) {
  static_rfor<0, NV>([&](auto KK) {
    constexpr int k = KK;
    const T inv = rcp_t(F.a[midx(k, k)]);
    for_anc<k>([&](auto II) {
      constexpr int i = II;
      const T a = F.a[midx(k, i)] * inv;
      for_anc_self<i>([&](auto JJ) { constexpr int j = JJ; F.a[midx(i, j)] -= a * F.a[midx(k, j)]; });
      F.a[midx(k, i)] = a;
    });
    F.a[midx(k, k)] = inv;
  });
}
// M^-1 = L^-1 D^-1 L^-T in three passes ([3P] mj_solveM); the halves are used on their own by the dual build ([3P] mj_solveM2)
template <class T>
REX_HD void solve_back(const MassFactor<T>& F, T (&x)[NV]) {   // x <- L^-T x
  static_rfor<0, NV>([&](auto KK) { constexpr int k = KK; for_anc<k>([&](auto II) { constexpr int i = II; x[i] -= F.a[midx(k, i)] * x[k]; }); });
}
// four right-hand sides at once: every factor entry is fetched once for four multiply-adds (most of the factor sits in
// AGPRs while the rows occupy the VGPRs: an entry costs a move per use)
template <class T>
REX_HD void solve_back4(const MassFactor<T>& F, T (&x0)[NV], T (&x1)[NV], T (&x2)[NV], T (&x3)[NV]) {
  static_rfor<0, NV>([&](auto KK) { constexpr int k = KK; for_anc<k>([&](auto II) { constexpr int i = II;
    const T l = F.a[midx(k, i)]; x0[i] -= l * x0[k]; x1[i] -= l * x1[k]; x2[i] -= l * x2[k]; x3[i] -= l * x3[k]; }); });
}
template <class T>
REX_HD void solve_fwd(const MassFactor<T>& F, T (&x)[NV]) {    // x <- L^-1 x
  static_for<0, NV>([&](auto KK) { constexpr int k = KK; for_anc<k>([&](auto II) { constexpr int i = II; x[k] -= F.a[midx(k, i)] * x[i]; }); });
}
template <class T>
REX_HD void solve(const MassFactor<T>& F, T (&x)[NV]) {
  solve_back(F, x);
  static_for<0, NV>([&](auto KK) { constexpr int k = KK; x[k] *= F.a[midx(k, k)]; });
  solve_fwd(F, x);
}

// ---- collision ([3P] engine_collision_primitive) -----------------------------------------------------
template <class T>
REX_HD void make_frame(T* f) {   // [3P] mju_makeFrame
  const T in = rcp_t(hsqrt(dot3(f, f))); for (int k = 0; k < 3; k++) f[k] *= in;
  if (hsqrt(dot3(f + 3, f + 3)) < T(0.5)) { f[3] = f[4] = f[5] = 0; if (f[1] < T(0.5) && f[1] > T(-0.5)) f[4] = 1; else f[5] = 1; }
  T d = dot3(f, f + 3); for (int k = 0; k < 3; k++) f[3 + k] -= d * f[k];
  T n2 = hsqrt(dot3(f + 3, f + 3));
  if (n2 < T(1e-15)) { f[3] = 1; f[4] = 0; f[5] = 0; } else { const T in2 = rcp_t(n2); for (int k = 0; k < 3; k++) f[3 + k] *= in2; }
  cross3(f + 6, f, f + 3);
}
template <class T>
REX_HD T impedance3(const Model<T>& m, T x_abs) {   // power 2, midpoint .5
  T x = x_abs * rcp_t(m.width);
  T y = x < T(0.5) ? T(2) * x * x : T(1) - T(2) * (T(1) - x) * (T(1) - x);
  T imp = m.dmin + y * (m.dmax - m.dmin);
  return x >= T(1) ? m.dmax : imp;
}

// Translational Jacobian of a world point p, body-2 chain minus body-1 chain (dof masks), projected on NDIR directions
// at once: out[d][i] = dir_d . (sign_i * axis_i x (p - anchor_i)), compile-time dof indices.
template <int NDIR, class T>
REX_HD void jac_dirs(Scratch<T>& s, int mask1, int mask2, const T* p, const T* dirs, T (&out)[NDIR][NV]) {
  const int diff = mask1 ^ mask2;   // dofs that move exactly one of the two bodies
  static_for<0, 3>([&](auto II) {   // root translations: unit columns
    constexpr int i = II;
    const T sign = T(((mask2 >> i) & 1) - ((mask1 >> i) & 1));
    for (int d = 0; d < NDIR; d++) out[d][i] = sign * dirs[3 * d + i];
  });
  // The hinge dofs by limb (root rotations, abdomen, the legs, the arms): a limb no lane of the wave needs is skipped, and a
  // limb that is needed reads all its anchors / axes in ONE batch of LDS reads before the first cross product (a skip test
  // and a read-wait-compute round trip per dof was 20 serialised LDS latencies per contact).
  auto limb = [&](auto LO, auto HI) {
    constexpr int lo = LO, hi = HI, n = hi - lo;
    for (int d = 0; d < NDIR; d++) for (int i = lo; i < hi; i++) out[d][i] = 0;
    if (REX_WAVE_ANY((diff & (((1 << n) - 1) << lo)) != 0)) {
      T an[n][3], ax[n][3];
      static_for<0, n>([&](auto JJ) { constexpr int j = JJ, i = lo + j; for (int k = 0; k < 3; k++) { an[j][k] = dual(s, GEO_DOF + i * 6 + k); ax[j][k] = dual(s, GEO_DOF + i * 6 + 3 + k); } });
      static_for<0, n>([&](auto JJ) {
        constexpr int j = JJ, i = lo + j;
        const T sign = T(((mask2 >> i) & 1) - ((mask1 >> i) & 1));
        T r[3] = {p[0] - an[j][0], p[1] - an[j][1], p[2] - an[j][2]}, col[3];
        cross3(col, ax[j], r);
        for (int d = 0; d < NDIR; d++) out[d][i] = sign * dot3(dirs + 3 * d, col);
      });
    }
  };
  limb(IC<3>{}, IC<6>{}); limb(IC<6>{}, IC<9>{}); limb(IC<9>{}, IC<13>{}); limb(IC<13>{}, IC<17>{}); limb(IC<17>{}, IC<20>{}); limb(IC<20>{}, IC<23>{});
}

// hinge-limit rows ([3P] mj_instantiateLimit; every hinge of the humanoid is limited, humanoid.xml:4).  They precede the
// contact rows in MuJoCo's row order.
template <class T>
REX_HD void limit_rows(const Model<T>& m, const T* qpos, const T* qvel, Kin<T>& K, Scratch<T>& s) {
  // Both sides of every hinge are tested in one straight-line pass (a joint cannot be beyond both limits); the 17 branches
  // below then work on registers.  Tested inside the branches, every joint angle -- spilled by then -- came back from scratch
  // with a wait of its own: 34 serialised memory round trips per evaluation.  (Computing impedance, R and aref up front as
  // well keeps too much alive across the branches and is slower.)
  T dist[NJNT], vel[NJNT], jsign[NJNT];
  static_for<1, NJNT>([&](auto JJ) {
    constexpr int j = JJ, d = j + 5;
    const T val = qpos[j + 6], dlo = val - m.jnt_lo[j], dhi = m.jnt_hi[j] - val;   // side -1: dist = val - lo; side +1: dist = hi - val
    const bool low = dlo < T(0);
    dist[j] = low ? dlo : dhi; jsign[j] = low ? T(1) : T(-1);                      // J_row = -side e_d
    vel[j] = jsign[j] * qvel[d];
  });
  int ne = 0;
  static_for<1, NJNT>([&](auto JJ) {
    constexpr int j = JJ, d = j + 5;
    if (dist[j] < T(0) && ne < MAXEFC) {
      for (int k = 0; k < NV; k++) s.J[ne][k] = (k == d) ? jsign[j] : T(0);
      T imp = impedance3(m, habs(dist[j]));
      s.R[ne] = hmax(T(1e-15), (T(1) - imp) * m.dof_invw[d] * rcp_t(imp));
      s.aref[ne] = -m.B * vel[j] - m.K * imp * dist[j];
      ne++;
    }
  });
  K.nefc = ne;
}

// A detected contact: recorded (diagnostics / tests) and, if inside the margin, turned into its constraint rows right away
// while position and frame are still in registers ([3P] mj_instantiateContact + mj_diagApprox + mj_makeImpedance +
// mj_referenceConstraint; pyramidal cone: n + mu t1, n - mu t1, n + mu t2, n - mu t2).
template <class T>
REX_HD void add_contact(Kin<T>& K, Scratch<T>& s, const Model<T>& m, const T* qvel, const PairRec<T>& pr, T dist, const T* pos, const T* normal, const T* yaxis) {
  if (K.ncon >= MAXCON) { K.overflow = 1; return; }
  int c = K.ncon++;
#if !defined(__HIP_DEVICE_COMPILE__)
  s.cdist[c] = dist; s.cdim[c] = pr.dim; s.cb1[c] = pr.b1; s.cb2[c] = pr.b2;
  for (int k = 0; k < 3; k++) s.cpos[c][k] = pos[k];
#else
  (void)c;
#endif
  T f[9];
  for (int k = 0; k < 3; k++) { f[k] = normal[k]; f[3 + k] = yaxis ? yaxis[k] : T(0); f[6 + k] = 0; }
  make_frame(f);
  if (!(dist < m.margin)) return;
  int ne = K.nefc;
  const T tran = pr.tran, mu = pr.mu;
  const T imp = impedance3(m, habs(dist - m.margin)), kterm = m.K * imp * (dist - m.margin);
  if (pr.dim == 1) {
    if (ne >= MAXEFC) { K.overflow = 1; return; }
    T jn[1][NV];
    jac_dirs<1>(s, pr.mask1, pr.mask2, pos, f, jn);
    T vel = 0; for (int k = 0; k < NV; k++) { s.J[ne][k] = jn[0][k]; vel += jn[0][k] * qvel[k]; }
    s.R[ne] = hmax(T(1e-15), (T(1) - imp) * tran * rcp_t(imp));
    s.aref[ne] = -m.B * vel - kterm;
    ne++;
  } else {
    if (ne + 4 > MAXEFC) { K.overflow = 1; return; }
    T jf[3][NV];
    jac_dirs<3>(s, pr.mask1, pr.mask2, pos, f, jf);
    const T R1 = hmax(T(1e-15), (T(1) - imp) * (tran + mu * mu * tran) * rcp_t(imp)), Rpy = T(2) * mu * mu * R1;
    for (int t = 1; t <= 2; t++) {
      for (int sg = 1; sg >= -1; sg -= 2) {
        T vel = 0;
        for (int k = 0; k < NV; k++) { T v = jf[0][k] + sg * mu * jf[t][k]; s.J[ne][k] = v; vel += v * qvel[k]; }
        s.R[ne] = Rpy; s.aref[ne] = -m.B * vel - kterm;
        ne++;
      }
    }
  }
  K.nefc = ne;
}
// what the narrow phase found for one pair: at most two contacts (plane-capsule, parallel capsules)
template <class T>
struct Hits { int n; T dist[2], pos[2][3], normal[2][3]; };

template <class T>
REX_HD void sphere_sphere(Hits<T>& h, const Model<T>& m, const T* c1, T r1, const T* c2, T r2) {
  T d[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
  T len = hsqrt(dot3(d, d)), dist = len - r1 - r2;
  if (dist > m.margin || h.n >= 2) return;
  T n[3] = {1, 0, 0};
  if (len >= T(1e-15)) { const T il = rcp_t(len); n[0] = d[0] * il; n[1] = d[1] * il; n[2] = d[2] * il; }
  // value selects into both slots, not `if (first) h.pos[0] = .. else h.pos[1] = ..`: LLVM sinks the two stores into one
  // store through a selected ADDRESS, which puts the whole Hits record into scratch -- every hit then went out and came back
  // through memory, one wait per field, before it reached the LDS queue
  const bool first = h.n == 0; h.n++;
  for (int x = 0; x < 3; x++) {
    const T px = c1[x] + n[x] * (r1 + T(0.5) * dist);
    h.pos[0][x] = first ? px : h.pos[0][x]; h.normal[0][x] = first ? n[x] : h.normal[0][x];
    h.pos[1][x] = first ? h.pos[1][x] : px; h.normal[1][x] = first ? h.normal[1][x] : n[x];
  }
  h.dist[0] = first ? dist : h.dist[0]; h.dist[1] = first ? h.dist[1] : dist;
}
template <class T>
REX_HD void plane_sphere(Hits<T>& h, const Model<T>& m, const T* c, T r) {
  T dist = c[2] - r;                     // the floor: z = 0, normal +z (humanoid.xml:28)
  if (dist > m.margin || h.n >= 2) return;
  const bool first = h.n == 0; h.n++;
  const T pz = c[2] - (r + T(0.5) * dist), pp[3] = {c[0], c[1], pz}, nn[3] = {T(0), T(0), T(1)};
  for (int x = 0; x < 3; x++) {   // (value selects: see sphere_sphere)
    h.pos[0][x] = first ? pp[x] : h.pos[0][x]; h.normal[0][x] = first ? nn[x] : h.normal[0][x];
    h.pos[1][x] = first ? h.pos[1][x] : pp[x]; h.normal[1][x] = first ? h.normal[1][x] : nn[x];
  }
  h.dist[0] = first ? dist : h.dist[0]; h.dist[1] = first ? h.dist[1] : dist;
}

// narrow phase of one candidate pair ([3P] engine_collision_primitive); yaxis = frame hint of the contacts (capsule axis for
// plane-capsule, else none)
template <class T>
REX_HD void collide_pair(const Model<T>& m, Scratch<T>& s, const PairRec<T>& pr, Hits<T>& h, T (&yaxis)[3], bool& has_y) {
  const int t1 = pr.t1, t2 = pr.t2;
  h.n = 0; has_y = false;
  for (int k = 0; k < 2; k++) { h.dist[k] = 0; for (int x = 0; x < 3; x++) { h.pos[k][x] = 0; h.normal[k][x] = 0; } }
  T p1[3], a1[3], p2[3], a2[3];
  for (int k = 0; k < 3; k++) { p2[k] = dual(s, GEO_GEOM + (pr.g2 - 1) * 6 + k); a2[k] = dual(s, GEO_GEOM + (pr.g2 - 1) * 6 + 3 + k); yaxis[k] = a2[k]; }
  const T r2 = pr.r2, l2 = pr.l2;
  if (t1 == G_PLANE) {
    if (t2 == G_SPHERE) plane_sphere(h, m, p2, r2);
    else {   // [3P] mjc_PlaneCapsule: the two end spheres, frame y-axis along the capsule
      T c[3]; has_y = true;
      for (int sg = 1; sg >= -1; sg -= 2) { for (int k = 0; k < 3; k++) c[k] = p2[k] + a2[k] * (sg * l2); plane_sphere(h, m, c, r2); }
    }
    return;
  }
  for (int k = 0; k < 3; k++) { p1[k] = dual(s, GEO_GEOM + (pr.g1 - 1) * 6 + k); a1[k] = dual(s, GEO_GEOM + (pr.g1 - 1) * 6 + 3 + k); }
  const T r1 = pr.r1, l1 = pr.l1;
  T d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  T c1[3] = {p1[0], p1[1], p1[2]}, c2[3] = {p2[0], p2[1], p2[2]};   // the two sphere centres the pair reduces to
  bool single = true;
  if (t1 == G_SPHERE && t2 == G_CAPSULE) {
    T x = -(d[0] * a2[0] + d[1] * a2[1] + d[2] * a2[2]);   // (p1 - p2).a2
    x = hmin(hmax(x, -l2), l2);
    for (int k = 0; k < 3; k++) c2[k] = p2[k] + a2[k] * x;
  } else if (t1 == G_CAPSULE) {   // capsule-capsule ([3P] mjc_CapsuleCapsule)
    T dif[3] = {-d[0], -d[1], -d[2]};   // p1 - p2
    T ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif), det = ma * mc - mb * mb;
    if (habs(det) >= T(1e-15)) {
      const T idet = rcp_t(det), imc = rcp_t(mc), ima = rcp_t(ma);
      T x1 = (mc * u - mb * v) * idet, x2 = (ma * v - mb * u) * idet;
      if (x1 > l1) { x1 = l1; x2 = (v - mb * l1) * imc; } else if (x1 < -l1) { x1 = -l1; x2 = (v + mb * l1) * imc; }
      if (x2 > l2) { x2 = l2; x1 = (u - mb * l2) * ima; } else if (x2 < -l2) { x2 = -l2; x1 = (u + mb * l2) * ima; }
      if (x1 > l1) x1 = l1; else if (x1 < -l1) x1 = -l1;
      for (int k = 0; k < 3; k++) { c1[k] = p1[k] + a1[k] * x1; c2[k] = p2[k] + a2[k] * x2; }
    } else {   // parallel axes: end points of 1 against 2, then of 2 against 1 (<= 2 contacts)
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

// [3P] mj_collision.  Broad phase: every candidate pair of the compile-time table against its bounding spheres, straight-
// line code on the geom centres (read once from the LDS column) that leaves one bit per pair.  Narrow phase + constraint
// rows: only for the pairs some lane of the wave kept, in table order (the PGS row order depends on it).
template <class T>
REX_HD void collide(const Model<T>& m, const T* qvel, Kin<T>& K, Scratch<T>& s) {
  K.ncon = 0;
  REX_HSTAMP(c0);
  T gp[NGEOM][3];
  static_for<1, NGEOM>([&](auto GG) { constexpr int g = GG; for (int k = 0; k < 3; k++) gp[g][k] = dual(s, GEO_GEOM + (g - 1) * 6 + k); });
  unsigned cand[(MAXPAIR + 31) / 32] = {};
  // the 109 thresholds (bound1 + bound2 + margin)^2 are constants of the model: left alone, LICM hoists them out of the RK4 /
  // frame-skip loops and the register allocator then reloads each one from scratch right before its compare (109 dependent
  // ~500-cycle round trips per evaluation were measured).  Opaque copies of the margin and the 17 bounds keep them two adds and a multiply at the use.
  T margin = m.margin, bnd[NGEOM]; opaque(margin);
  static_for<1, NGEOM>([&](auto GG) { constexpr int g = GG; bnd[g] = m.geom_bound[g]; opaque(bnd[g]); });
  // Second test, fused with the first: bounding spheres of two long capsules side by side (the thighs, the shins: candidates
  // in 85-100 % of all states, in contact in 0.2-1 %) always overlap.  A lower bound of the segment-segment distance that needs
  // no closest points: seen along axis 1 (an orthogonal projection never lengthens a distance) segment 1 is a point and
  // segment 2 has half-length l2 sin(theta), so dist >= |d_perp1| - l2 sin(theta); the same along axis 2.  A pair is dropped
  // only if a bound exceeds r1 + r2 + margin by 0.1 mm (rounding is 1e-7 here; sizes enter as compile-time upper bounds), so
  // no contact the narrow phase would report is lost -- it removes 5.5 of 6.5 candidates per state and takes the busiest lane
  // of a wave from ~19 narrow-phase trips to ~6.
  T ga[NGEOM][3];
  static_for<1, NGEOM>([&](auto GG) { constexpr int g = GG;
    if constexpr (kGeomType[g] == G_CAPSULE) for (int k = 0; k < 3; k++) ga[g][k] = dual(s, GEO_GEOM + (g - 1) * 6 + 3 + k); });
  const T slack = margin + T(1e-4);
  static_for<0, kPairs.n>([&](auto PP) {
    constexpr int p = PP, g1 = kPairs.g1[p], g2 = kPairs.g2[p], t1 = kGeomType[g1], t2 = kGeomType[g2];
    bool keep;
    if constexpr (t1 == G_PLANE) {
#if !defined(REX_NO_SECOND_CULL)   // (tests build the harness both ways: the second test must never change a result)
      if constexpr (t2 == G_CAPSULE) keep = !(gp[g2][2] - T(kGeomHalfUB[g2]) * habs(ga[g2][2]) - T(kGeomRadUB[g2]) > slack);   // lowest point of the capsule: what the narrow phase tests
      else
#endif
      keep = !(gp[g2][2] - bnd[g2] > margin);   // bounding sphere above the floor
    } else {
      const T d[3] = {gp[g2][0] - gp[g1][0], gp[g2][1] - gp[g1][1], gp[g2][2] - gp[g1][2]}, reach = bnd[g1] + bnd[g2] + margin, dd = dot3(d, d);
      T worst = dd - reach * reach;   // > 0: bounding spheres apart
#if !defined(REX_NO_SECOND_CULL)
      if constexpr (t1 == G_CAPSULE && t2 == G_CAPSULE) {
        const T rs = T(kGeomRadUB[g1] + kGeomRadUB[g2]) + slack;
        const T c = dot3(ga[g1], ga[g2]), sn = fast_sqrt(hmax(T(0), T(1) - c * c)), p1 = dot3(d, ga[g1]), p2 = dot3(d, ga[g2]);
        const T e1 = rs + T(kGeomHalfUB[g2]) * sn, e2 = rs + T(kGeomHalfUB[g1]) * sn;
        worst = hmax(worst, hmax((dd - p1 * p1) - e1 * e1, (dd - p2 * p2) - e2 * e2));
      } else if constexpr (t1 == G_CAPSULE || t2 == G_CAPSULE) {   // a sphere and a capsule: distance of the centre from the capsule's axis line
        constexpr int gc = t1 == G_CAPSULE ? g1 : g2;
        const T rs = T(kGeomRadUB[g1] + kGeomRadUB[g2]) + slack, pc = dot3(d, ga[gc]);
        worst = hmax(worst, (dd - pc * pc) - rs * rs);
      }
#endif
      keep = !(worst > T(0));
    }
    cand[p >> 5] |= keep ? (1u << (p & 31)) : 0u;
  });
  REX_HSTAMP(c1); REX_HACC(K, HT_BROAD, c0, c1);
  // Narrow phase per LANE, not per pair: every lane walks its own candidate bits in table order (the PGS row order depends on
  // it) with its own pair record, so a wave needs as many trips as its busiest lane has candidates (~10) instead of one per
  // pair of the union over its lanes (~42).  Hits are queued in the free tail of the LDS column and turned into rows slot by
  // slot afterwards -- again one trip per queue slot, not per contact of the union -- by the ONE inlined copy of add_contact.
  unsigned w0 = cand[0], w1 = cand[1], w2 = cand[2], w3 = cand[3];
  static_assert((kPairs.n + 31) / 32 == 4, "pair mask words");
  T* const col = &dual(s, 0);
  int nq = 0;
  // the lane's next candidate (table order) and its pair record: the record is gathered from the constant table one
  // candidate AHEAD, so that its latency (a vector load per lane, L2 at best) runs behind the current pair's narrow phase
  auto pop = [&]() -> int {
    if ((w0 | w1 | w2 | w3) == 0u) return -1;
    const unsigned wsel = w0 ? w0 : (w1 ? w1 : (w2 ? w2 : w3));
    const int base = w0 ? 0 : (w1 ? 32 : (w2 ? 64 : 96));
    const unsigned cleared = wsel & (wsel - 1u);
    if (w0) w0 = cleared; else if (w1) w1 = cleared; else if (w2) w2 = cleared; else w3 = cleared;
    return base + __builtin_ctz(wsel);
  };
  int p = pop();
  PairRec<T> pr = m.pair[p < 0 ? 0 : p];
  bool more = true;
  while (more) {
    for (;;) {   // gather: lanes with candidates left and room for two more hits
      const bool go = p >= 0 && nq <= HITQ_MAX - 2;
      if (!REX_WAVE_ANY(go)) break;
      REX_HSTAMP(n0);
      if (go) {
        const int pn = pop();
        const PairRec<T> nx = m.pair[pn < 0 ? 0 : pn];
        Hits<T> h; T yaxis[3]; bool has_y;
        collide_pair(m, s, pr, h, yaxis, has_y);
        if (h.n > 0) { T* q = col + HITQ_BASE + HITQ_WORDS * nq; q[0] = h.dist[0]; for (int k = 0; k < 3; k++) { q[1 + k] = h.pos[0][k]; q[4 + k] = h.normal[0][k]; } q[7] = T(p); nq++; }
        if (h.n > 1) { T* q = col + HITQ_BASE + HITQ_WORDS * nq; q[0] = h.dist[1]; for (int k = 0; k < 3; k++) { q[1 + k] = h.pos[1][k]; q[4 + k] = h.normal[1][k]; } q[7] = T(p); nq++; }
        p = pn; pr = nx;
      }
      REX_HSTAMP(n1); REX_HACC(K, HT_PAIR, n0, n1); REX_HCNT(K, HC_PAIR_CALLS, 1);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
    for (int slot = 0; slot < HITQ_MAX; slot++) {   // rows, slot by slot
      const bool mine = slot < nq;
      if (!REX_WAVE_ANY(mine)) break;
      REX_HSTAMP(r0);
      if (mine) {
        const T* q = col + HITQ_BASE + HITQ_WORDS * slot;
        const T hd = q[0], hp[3] = {q[1], q[2], q[3]}, hn[3] = {q[4], q[5], q[6]};
        const PairRec<T> pr = m.pair[(int)q[7]];
        T yaxis[3];
        for (int k = 0; k < 3; k++) yaxis[k] = dual(s, GEO_GEOM + (pr.g2 - 1) * 6 + 3 + k);
        const bool has_y = pr.t1 == G_PLANE && pr.t2 == G_CAPSULE;   // [3P] mjc_PlaneCapsule: frame y-axis along the capsule
        add_contact(K, s, m, qvel, pr, hd, hp, hn, has_y ? yaxis : (const T*)nullptr);
      }
      REX_HSTAMP(r1); REX_HACC(K, HT_ROWS, r0, r1); REX_HCNT(K, HC_ROW_CALLS, 1);
    }
    nq = 0;
    more = REX_WAVE_ANY(p >= 0);
  }
  REX_HSTAMP(c2); REX_HACC(K, HT_NARROW_LOOP, c1, c2);
}

// [3P] mj_solPGS on the dual, with qacc carried along: res_i = J_i qacc - aref_i + R_i f_i.  Rows stay in scratch: only
// used when an evaluation has more rows than the LDS column holds (pile-ups).
template <class T>
REX_HD int solve_pgs(const Model<T>& m, const MassFactor<T>& F, const Kin<T>& K, Scratch<T>& s, T* qacc) {
  for (int k = 0; k < NV; k++) qacc[k] = K.qacc_smooth[k];
  for (int i = 0; i < K.nefc; i++) {
    T x[NV], jr[NV];
    for (int k = 0; k < NV; k++) { jr[k] = s.J[i][k]; x[k] = jr[k]; }
    solve(F, x);
    T a = s.R[i];
    for (int k = 0; k < NV; k++) { s.MiJ[i][k] = x[k]; a += jr[k] * x[k]; }
    s.Adiag[i] = a; s.force[i] = 0;   // warmstart disabled (humanoid.xml:11)
  }
  const T scale = T(1) / (m.meaninertia * T(NV));
  int it = 0;
  const int n = K.nefc;
  // the rows live in scratch: row i + 1 (J, M^-1 J^T and its four scalars) is fetched while row i is updated -- unpipelined, every
  // row update was two dependent memory round trips, which is all this path (lying, crumpled bodies: 22 .. 108 rows) waits for
  T jn[NV], mn[NV], sn[4];
  auto fetch = [&](int i) { for (int k = 0; k < NV; k++) { jn[k] = s.J[i][k]; mn[k] = s.MiJ[i][k]; } sn[0] = s.R[i]; sn[1] = s.aref[i]; sn[2] = s.Adiag[i]; sn[3] = s.force[i]; };
  for (; it < m.iterations; it++) {
    T improvement = 0;
    fetch(0);
    for (int i = 0; i < n; i++) {
      T jr[NV], mr[NV];
      for (int k = 0; k < NV; k++) { jr[k] = jn[k]; mr[k] = mn[k]; }
      const T Ri = sn[0], arefi = sn[1], Ad = sn[2], old = sn[3];
      fetch(i + 1 < n ? i + 1 : i);   // (the force of row i + 1 is not written by row i)
      T res = Ri * old - arefi;
      for (int k = 0; k < NV; k++) res += jr[k] * qacc[k];
      const T nf = hmax(T(0), old - res / Ad), df = nf - old;
      s.force[i] = nf;
      if (df != T(0)) for (int k = 0; k < NV; k++) qacc[k] += mr[k] * df;
      improvement -= T(0.5) * df * df * Ad + df * res;
    }
    if (improvement * scale < m.tolerance) { it++; break; }
  }
  return it;
}

// phase boundary: keeps the machine scheduler from interleaving two phases of forward() (which only lengthens live
// ranges: 739 -> spilled VGPRs without it)
#if defined(__HIP_DEVICE_COMPILE__)
#define REX_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define REX_FENCE() ((void)0)
#endif

// keeps a batch of LDS reads together: all of them are issued, waited for once and held in registers from here on
#if defined(__HIP_DEVICE_COMPILE__) && !defined(REX_NOPIN)
#define REX_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define REX_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#else
#define REX_PIN4(a, b, c, d) ((void)0)
#define REX_PIN2(a, b) ((void)0)
#endif

template <int Q, int NP, class T>
REX_HD void pin_all(T (&a)[NP]) { if constexpr (Q < NP) { REX_PIN4(a[Q], a[Q + 1], a[Q + 2], a[Q + 3]); pin_all<Q + 4>(a); } }

// Gauss-Seidel sweeps over rows / columns 0..NC-1 of the lane's packed A (NC a multiple of 4, or DUAL_NMAX).  The forces
// live in registers and every index is a compile-time constant: a row update is one batch of LDS reads at fixed offsets
// (no address arithmetic, no LDS write) followed by four short FMA chains.  Rows beyond the largest row count of the wave
// are skipped; padding inside the range is zero, which makes its update a no-op.
#ifndef REX_PGS_CHECK
#define REX_PGS_CHECK 10
#endif
constexpr int PGS_CHECK = REX_PGS_CHECK;   // the device sweeps evaluate the stopping criterion every PGS_CHECK-th sweep (pgs_sweeps_sq)

// How often a sweep evaluates the stopping criterion: every PGS_CHECK-th sweep in the fp32 device code (pgs_sweeps_sq has the argument), every
// sweep ([3P] mj_solPGS) in the host instantiations -- unless the harness is built with -DREX_PGS_CHECK_HOST, which gives the fp32 host code the
// DEVICE's schedule so a host test exercises exactly the sweep counts the GPU runs (tests/test_humanoid_host.py).  REX_PGS_CHECK itself is a
// compile-time define: it reaches the product only through the hipcc flags, which __graft_entry__ hashes into librex_hip.so.digest.
template <class T> constexpr int pgs_check_period() {
#if defined(__HIP_DEVICE_COMPILE__) || defined(REX_PGS_CHECK_HOST)
  return sizeof(T) == 4 ? PGS_CHECK : 1;
#else
  return 1;
#endif
}
// is the criterion evaluated after sweep `it` (0-based) of at most `iters`?  (groups of K - 1 unchecked sweeps + one checked, the last sweep always checked)
template <class T> REX_HD bool pgs_checked(int it, int iters) { constexpr int K = pgs_check_period<T>(); return K == 1 || (it + 1) % K == 0 || it + 1 == iters; }
template <int NC, int I, class T>
REX_HD void pgs_load_row(const T* col, T (&a)[(NC + 3) / 4 * 4 + 2]) {   // row I of the packed A, then b_I and 1 / A_II
  constexpr int NP = (NC + 3) / 4 * 4;
  static_for<0, NP>([&](auto JJ) { constexpr int j = JJ; a[j] = j < NC ? col[j <= I ? tri(I) + j : tri(j) + I] : T(0); });
  a[NP] = col[DUAL_B + I]; a[NP + 1] = col[DUAL_DI + I];
}
template <int Q, int N, class T>
REX_HD void pin_row(T (&a)[N]) {
  if constexpr (Q + 4 <= N) { REX_PIN4(a[Q], a[Q + 1], a[Q + 2], a[Q + 3]); pin_row<Q + 4>(a); }
  else if constexpr (Q + 2 <= N) { REX_PIN2(a[Q], a[Q + 1]); pin_row<Q + 2>(a); }
}

template <int NC, class T>
REX_HD int pgs_sweeps(const Model<T>& m, const T* col, int n, T (&f)[DUAL_NMAX]) {
  constexpr int NP = (NC + 3) / 4 * 4;
  const T scale = T(1) / (m.meaninertia * T(NV));
  // Software pipeline over the rows: the reads of row i + 1 (they do not depend on the forces) are issued before row i is
  // consumed, so the LDS latency hides behind the FMA chains and the projection of the previous row.  All NC rows run (no
  // branch between rows: the whole sweep is one basic block); rows >= n are zero padding and leave f unchanged.
  T buf[NP + 2];
  pgs_load_row<NC, 0>(col, buf);
  int it = 0;
  T improvement = 0;
  auto sweep = [&](auto CHECK) {
    static_for<0, NC>([&](auto II) {
      constexpr int i = II, nxt = (i + 1) % NC;
      T a[NP + 2];
      static_for<0, NP + 2>([&](auto KK) { a[KK] = buf[KK]; });
      pgs_load_row<NC, nxt>(col, buf);   // next row (row 0 of the next sweep after the last one): in flight while this one is used
      pin_row<0>(a);
      T res;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(REX_NO_PK)
      if constexpr (sizeof(T) == 4) {   // two packed chains: v_pk_fma_f32 does two of the row's products per instruction
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f acc0 = {a[NP], 0.0f}, acc1 = {0.0f, 0.0f};
        static_for<0, NP / 4>([&](auto QQ) {
          constexpr int q = 4 * QQ;
          const v2f a01 = {a[q], a[q + 1]}, f01 = {f[q], q + 1 < NC ? f[q + 1 < DUAL_NMAX ? q + 1 : 0] : 0.0f};
          const v2f a23 = {a[q + 2], a[q + 3]}, f23 = {q + 2 < NC ? f[q + 2 < DUAL_NMAX ? q + 2 : 0] : 0.0f, q + 3 < NC ? f[q + 3 < DUAL_NMAX ? q + 3 : 0] : 0.0f};
          acc0 = __builtin_elementwise_fma(a01, f01, acc0); acc1 = __builtin_elementwise_fma(a23, f23, acc1);
        });
        const v2f t = acc0 + acc1;
        res = t.x + t.y;
      } else
#endif
      {
        T r0 = a[NP], r1 = 0, r2 = 0, r3 = 0;
        static_for<0, NP / 4>([&](auto QQ) {
          constexpr int q = 4 * QQ;
          r0 += a[q] * f[q]; if constexpr (q + 1 < NC) r1 += a[q + 1] * f[q + 1]; if constexpr (q + 2 < NC) r2 += a[q + 2] * f[q + 2]; if constexpr (q + 3 < NC) r3 += a[q + 3] * f[q + 3];
        });
        res = (r0 + r1) + (r2 + r3);
      }
      const T old = f[i], nf = hmax(T(0), old - res * a[NP + 1]), df = nf - old;
      f[i] = nf;
      if constexpr (decltype(CHECK)::value) improvement -= df * (T(0.5) * df * a[i] + res);   // cost decrease of this update ([3P] mj_solPGS), three instructions
    });
  };
  // the stopping criterion in every K-th sweep only on the device (pgs_sweeps_sq has the argument); every sweep on the host
  constexpr int K = pgs_check_period<T>();
  while (it < m.iterations) {
    int nun = m.iterations - 1 - it; nun = nun < K - 1 ? nun : K - 1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int u = 0; u < nun; u++) { sweep(std::false_type{}); it++; }
    improvement = 0;
    sweep(std::true_type{}); it++;
    if (improvement * scale < m.tolerance) break;
  }
  return it;
}

// The same sweeps for NC <= 16 with A stored SQUARE, row-major with row stride NP = NC rounded up to a multiple of 4, then
// b[NP] and 1/A_ii[NP] (NP^2 + 2 NP <= 288 words).  A row is NP / 4 contiguous 16-byte LDS reads that land in aligned register
// pairs -- what v_pk_fma_f32 wants -- instead of NC scattered words of the packed triangle that have to be re-paired with
// moves.
constexpr int sq_stride(int nc) { return (nc + 3) / 4 * 4; }
#if defined(__HIP_DEVICE_COMPILE__)
typedef float pgs_v2f __attribute__((ext_vector_type(2)));
typedef float pgs_v4f __attribute__((ext_vector_type(4)));
template <int Q, int N> __device__ __forceinline__ void pin_v4(pgs_v4f (&a)[N]) { if constexpr (Q < N) { asm volatile("" : "+v"(a[Q])); pin_v4<Q + 1>(a); } }
__device__ __forceinline__ void pin_v2(pgs_v2f& x) { asm volatile("" : "+v"(x)); }
#endif
template <int NC, class T>
REX_HD int pgs_sweeps_sq(const Model<T>& m, const T* col, T (&f)[DUAL_NMAX]) {
  constexpr int NP = sq_stride(NC), BOFF = NP * NP, DOFF = BOFF + NP;
  static_assert(DOFF + NP <= DUAL_WORDS && NC <= DUAL_NMAX, "square layout must fit the LDS column");
  const T scale = T(1) / (m.meaninertia * T(NV));
  int it = 0;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(REX_NO_PK)
  if constexpr (sizeof(T) == 4) {
    typedef pgs_v2f v2f; typedef pgs_v4f v4f;
    // Residual form: res_r = b_r + sum_j A_rj f_j is kept current for every row; an update of f_k by df pushes A_rk df into
    // all of them (column k = row k, A is symmetric).  The same multiply-adds as a row dot product per update, but they are
    // independent of one another: the dependent chain from one row's update to the next is fma, max, sub, pk_fma instead of
    // a whole dot product and its reduction -- what counts for a lone wave on its SIMD.
    v2f rp[NP / 2]; float fv[NC];
    static_for<0, NP / 4>([&](auto QQ) { constexpr int q = QQ; const v4f b4 = *(const v4f*)(col + BOFF + 4 * q); rp[2 * q] = v2f{b4.x, b4.y}; rp[2 * q + 1] = v2f{b4.z, b4.w}; });
    static_for<0, NC>([&](auto II) { fv[II] = 0.0f; });
    v4f buf[NP / 4]; float dnext;
    auto load_row = [&](auto II, v4f (&a)[NP / 4], float& d) {
      constexpr int i = II;
      static_for<0, NP / 4>([&](auto QQ) { constexpr int q = QQ; a[q] = *(const v4f*)(col + i * NP + 4 * q); });
      d = col[DOFF + i];
    };
    load_row(IC<0>{}, buf, dnext);
    // [3P] mj_solPGS stops after the first sweep whose cost decrease falls under the tolerance.  Computing the decrease costs four
    // of a row update's instructions and -- worse -- keeps the matrix from living in registers across the sweeps: without it a
    // 12-row sweep is 125 instructions (6 v_pk_fma + fma, max, sub per row, no LDS read, no move) instead of 257.  A wave runs
    // until its LAST lane stops (one that hits the 50-sweep cap in most waves), so the criterion is evaluated in every
    // PGS_CHECK-th sweep only: a lane that would have stopped after sweep k stops after sweep PGS_CHECK * ceil(k / PGS_CHECK) <= 50
    // instead -- up to PGS_CHECK - 1 more sweeps of a system already converged to the tolerance (1e-8 of the cost scale: far
    // below fp32 rounding of the result; the unchecked sweeps cost half, so even those lanes do not pay), never more than
    // MuJoCo's cap.  Humanoid step kernel 2.17 -> 1.99 ms (every 5th sweep, square sizes) -> 1.91 (packed sizes too) -> 1.86 (10th).
    float improvement = 0;
    auto sweep = [&](auto CHECK) {
      static_for<0, NC>([&](auto II) {
        constexpr int i = II, nxt = (i + 1) % NC;
        v4f a[NP / 4]; const float di = dnext;
        static_for<0, NP / 4>([&](auto QQ) { a[QQ] = buf[QQ]; });
        load_row(IC<nxt>{}, buf, dnext);   // next row in flight while this one is consumed
        const float res = (i & 1) ? rp[i / 2].y : rp[i / 2].x;
        const float old = fv[i], nf = __builtin_fmaxf(0.0f, old - res * di), df = nf - old;   // a NaN residual (non-finite state: the lane is flagged) gives 0
        fv[i] = nf;
        if constexpr (decltype(CHECK)::value) { const float aii = a[i / 4][i & 3]; improvement -= df * (0.5f * df * aii + res); }
        const v2f dfp = {df, df};
        static_for<0, NP / 4>([&](auto QQ) {
          constexpr int q = QQ;
          rp[2 * q] = __builtin_elementwise_fma(v2f{a[q].x, a[q].y}, dfp, rp[2 * q]);
          rp[2 * q + 1] = __builtin_elementwise_fma(v2f{a[q].z, a[q].w}, dfp, rp[2 * q + 1]);
        });
      });
    };
    while (it < m.iterations) {
      int nun = m.iterations - 1 - it; nun = nun < PGS_CHECK - 1 ? nun : PGS_CHECK - 1;
#pragma unroll 1
      for (int u = 0; u < nun; u++) { sweep(std::false_type{}); it++; }
      improvement = 0;
      sweep(std::true_type{}); it++;
      if (improvement * scale < m.tolerance) break;
    }
    static_for<0, NC>([&](auto II) { f[II] = fv[II]; });
    return it;
  } else
#endif
  {
    for (; it < m.iterations; it++) {
      T improvement = 0;
      for (int i = 0; i < NC; i++) {
        T res = col[BOFF + i];
        for (int j = 0; j < NC; j++) res += col[i * NP + j] * f[j];
        const T old = f[i], nf = hmax(T(0), old - res * col[DOFF + i]), df = nf - old;
        f[i] = nf;
        improvement -= df * (T(0.5) * df * col[i * NP + i] + res);
      }
      if (pgs_checked<T>(it, m.iterations) && improvement * scale < m.tolerance) { it++; break; }
    }
    return it;
  }
}

// dot product of two nv-vectors; on the device (fp32) eleven v_pk_fma_f32 + one fma instead of 23 fma
template <class T>
REX_HD T dot_nv(const T (&a)[NV], const T (&b)[NV]) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(REX_NO_PK)
  if constexpr (sizeof(T) == 4) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f acc0 = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};
    static_for<0, NV / 4>([&](auto QQ) { constexpr int q = 4 * QQ;
      acc0 = __builtin_elementwise_fma(v2f{a[q], a[q + 1]}, v2f{b[q], b[q + 1]}, acc0);
      acc1 = __builtin_elementwise_fma(v2f{a[q + 2], a[q + 3]}, v2f{b[q + 2], b[q + 3]}, acc1); });
    acc0 = __builtin_elementwise_fma(v2f{a[20], a[21]}, v2f{b[20], b[21]}, acc0);
    const v2f t = acc0 + acc1;
    return (t.x + t.y) + a[22] * b[22];
  } else
#endif
  { T r = 0; for (int k = 0; k < NV; k++) r += a[k] * b[k]; return r; }
}
static_assert(NV == 23, "dot_nv is written for nv = 23");

// The same Gauss-Seidel sweeps on the dual: res_i = sum_j A_ij f_j + b_i with A = J M^-1 J^T + diag(R),
// b = J qacc_smooth - aref ([3P] mj_solPGS works on exactly this matrix).  A, f and b sit in the lane's LDS column, so a
// sweep costs n^2 LDS reads instead of 2 n nv reads of J / M^-1 J^T rows from scratch (which miss every cache level);
// J is read once per row pair to build A and once more for qacc = qacc_smooth + M^-1 J^T f.
template <class T>
REX_HD int solve_pgs_dual(const Model<T>& m, const MassFactor<T>& F, Kin<T>& K, Scratch<T>& s, T* qacc) {
  const int n = K.nefc;
  T* const col = (T*)__builtin_assume_aligned(&dual(s, 0), 16);
  REX_HSTAMP(p0);
  // Sweeps run over the smallest of eight fixed sizes that holds every lane of the wave (the launch ends with its slowest wave
  // and sweeps cost 50 x NC rows: sizes in steps of two above 8).  Sizes up to 16 keep A square (pgs_sweeps_sq), the two largest
  // the packed lower triangle; the level is wave-uniform, so the layout is known before A is built.
  int lvl = 0;
  static_for<0, 8>([&](auto LL) { constexpr int thr[8] = {4, 8, 10, 12, 14, 16, 18, 21}; if (REX_WAVE_ANY(n > thr[LL])) lvl = LL + 1; });
#if defined(__HIP_DEVICE_COMPILE__)
  lvl = __builtin_amdgcn_readfirstlane(lvl);
#endif
  const bool sq = lvl <= 5;
  const int stride = lvl == 0 ? 4 : lvl == 1 ? 8 : lvl <= 3 ? 12 : 16;          // sq_stride of the level's NC
  const int boff = sq ? stride * stride : DUAL_B, doff = sq ? boff + stride : DUAL_DI;
  static_for<0, DUAL_WORDS>([&](auto KK) { col[KK] = T(0); });   // padding rows / columns must read as zero
  // A = J M^-1 J^T = Y D^-1 Y^T with Y^T = L^-T J^T ([3P] mj_solveM2): only the backward half of the solve per row, four rows
  // per trip (one pass over the factor for all four).  Row j of the scratch array is overwritten with z_j = D^-1 y_j: the
  // dots with later rows and the final M^-1 J^T f = L^-1 sum_j f_j z_j need nothing else (J_j is dead once b_j is formed).
  // Every earlier row is read once per group of four, two rows per read batch, a batch ahead of its dot products.
  for (int j0 = 0; j0 < n; j0 += 4) {
    T y0[NV], y1[NV], y2[NV], y3[NV], Rr[4];
    bool ok[4]; int jj[4];
    static_for<0, 4>([&](auto TT) { constexpr int t = TT; ok[t] = j0 + t < n; jj[t] = ok[t] ? j0 + t : j0; });   // rows past the lane's count duplicate the group's first row and are dropped
    { T ar[4];
      for (int k = 0; k < NV; k++) { y0[k] = s.J[jj[0]][k]; y1[k] = s.J[jj[1]][k]; y2[k] = s.J[jj[2]][k]; y3[k] = s.J[jj[3]][k]; }
      static_for<0, 4>([&](auto TT) { constexpr int t = TT; Rr[t] = s.R[jj[t]]; ar[t] = s.aref[jj[t]]; });
      pin_row<0>(y0); pin_row<0>(y1); pin_row<0>(y2); pin_row<0>(y3);
      const T b0 = dot_nv(y0, K.qacc_smooth) - ar[0], b1 = dot_nv(y1, K.qacc_smooth) - ar[1], b2 = dot_nv(y2, K.qacc_smooth) - ar[2], b3 = dot_nv(y3, K.qacc_smooth) - ar[3];
      col[boff + j0] = b0; if (ok[1]) col[boff + j0 + 1] = b1; if (ok[2]) col[boff + j0 + 2] = b2; if (ok[3]) col[boff + j0 + 3] = b3; }
    solve_back4(F, y0, y1, y2, y3);
    auto rowp = [&](int r) -> T* { return col + (sq ? r * stride : tri(r)); };   // row r of A, columns 0..r
    auto put = [&](int r, int c, T v) { rowp(r)[c] = v; if (sq && c != r) col[c * stride + r] = v; };   // (r, c) with c <= r, and its mirror in the square layout
    // entries inside the group: z_u = D^-1 y_u is formed, stored (row j0 + u of the scratch array) and dotted with y_t, t >= u
    auto within = [&](auto UU, T (&yu)[NV]) {
      constexpr int u = UU;
      T z[NV];
      static_for<0, NV>([&](auto KK) { constexpr int k = KK; z[k] = yu[k] * F.a[midx(k, k)]; });
      if (ok[u]) {
        for (int k = 0; k < NV; k++) s.J[j0 + u][k] = z[k];
        const T d = Rr[u] + dot_nv(z, yu);
        rowp(j0 + u)[j0 + u] = d; col[doff + j0 + u] = rcp_t(d);
      }
      if constexpr (u < 1) { const T v = dot_nv(z, y1); if (ok[1]) put(j0 + 1, j0 + u, v); }
      if constexpr (u < 2) { const T v = dot_nv(z, y2); if (ok[2]) put(j0 + 2, j0 + u, v); }
      if constexpr (u < 3) { const T v = dot_nv(z, y3); if (ok[3]) put(j0 + 3, j0 + u, v); }
    };
    within(IC<0>{}, y0); within(IC<1>{}, y1); within(IC<2>{}, y2); within(IC<3>{}, y3);
    // earlier rows (already z rows): two per batch, the next batch in flight while this one is dotted with the four y
    T ra[NV], rb[NV], rc[NV], rd[NV];
    auto fetch2 = [&](int i, T (&u)[NV], T (&v)[NV]) {
      const int i0 = i < j0 ? i : 0, i1 = i + 1 < j0 ? i + 1 : i0;
      for (int k = 0; k < NV; k++) { u[k] = s.J[i0][k]; v[k] = s.J[i1][k]; }
    };
    auto use1 = [&](int i, T (&u)[NV]) {
      const T a0 = dot_nv(u, y0), a1 = dot_nv(u, y1), a2 = dot_nv(u, y2), a3 = dot_nv(u, y3);
      if (i < j0) { put(j0, i, a0); if (ok[1]) put(j0 + 1, i, a1); if (ok[2]) put(j0 + 2, i, a2); if (ok[3]) put(j0 + 3, i, a3); }
    };
    fetch2(0, ra, rb);
    for (int i = 0; i < j0; i += 4) {
      fetch2(i + 2, rc, rd);
      pin_row<0>(ra); pin_row<0>(rb); use1(i, ra); use1(i + 1, rb);
      fetch2(i + 4, ra, rb);
      pin_row<0>(rc); pin_row<0>(rd); use1(i + 2, rc); use1(i + 3, rd);
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
    case 5: it = pgs_sweeps_sq<16>(m, col, f); break;
    case 6: it = pgs_sweeps<18>(m, col, n, f); break;
    default: it = pgs_sweeps<DUAL_NMAX>(m, col, n, f); break;
  }
  REX_HSTAMP(p2); REX_HACC(K, HT_SWEEPS, p1, p2); REX_HCNT(K, HC_SWEEPS, it);
  T x[NV];
  for (int k = 0; k < NV; k++) x[k] = 0;
  static_for<0, DUAL_NMAX / 3>([&](auto CC) {   // sum_j f_j z_j, three rows per trip (reads batched; rows >= n re-read row 0 with f = 0)
    constexpr int i0 = 3 * CC;
    if (REX_WAVE_ANY(i0 < n)) {
      T r[3][NV], fi[3];
      static_for<0, 3>([&](auto RR) { constexpr int r_ = RR, i = i0 + r_; const int ii = i < n ? i : 0; fi[r_] = i < n ? f[i] : T(0);
        if (i < n) s.force[i] = fi[r_];
        for (int k = 0; k < NV; k++) r[r_][k] = s.J[ii][k]; });
      pin_row<0>(r[0]); pin_row<0>(r[1]); pin_row<0>(r[2]);
      for (int k = 0; k < NV; k++) x[k] += r[0][k] * fi[0] + r[1][k] * fi[1] + r[2][k] * fi[2];
    }
  });
  solve_fwd(F, x);
  for (int k = 0; k < NV; k++) qacc[k] = K.qacc_smooth[k] + x[k];
  REX_HSTAMP(p3); REX_HACC(K, HT_QACC, p2, p3);
  return it;
}

// [3P] mj_forward
template <class T>
REX_HD int forward(const Model<T>& m_in, const Lane<T>& L, const T* qpos, const T* qvel, const T* ctrl, Kin<T>& K, Scratch<T>& s, T* qacc, bool keep_obs = true) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(REX_HOIST_MODEL)
  // The model's ~500 constants are read through an opaque (zero) index every evaluation: read through a loop-invariant address
  // LICM hoists every scalar load to the kernel's entry, where the SGPRs cannot hold them -- they end up in VGPR lanes and
  // come back one v_readlane at a time (1 400 of the smooth phase's 7 000 instructions).
  int zidx = 0; asm volatile("" : "+s"(zidx));
  const Model<T>& m = (&m_in)[zidx];
#else
  const Model<T>& m = m_in;
#endif
  K.overflow = 0;
  REX_HSTAMP(t0);
  MassFactor<T> F;
  {
    // Order: kinematics (geometry into the LDS column) -> limit rows -> collision -> the rest of the smooth dynamics -> M.
    // Only the body frames (xmat, xipos: 168 values) are alive across the collision phase this way; with the whole smooth
    // phase first it was cinert + cdof (278) -- the compiler had sunk crb() behind the collision phase anyway -- or M (185).
    Smooth<T> S;
    kinematics(m, qpos, S, s);
    REX_FENCE(); REX_HSTAMP(t3); REX_HACC(K, HT_SMOOTH, t0, t3);
    limit_rows(m, qpos, qvel, K, s);
    REX_HSTAMP(t3b); REX_HACC(K, HT_LIMITS, t3, t3b);
    collide(m, qvel, K, s);
    REX_FENCE(); REX_HSTAMP(t4);
    com_pos(m, L, S, s);   // (reads the joint anchors / axes kinematics left in the LDS column: before the dual overwrites it)
    T qfrc_bias[NV], act[NV];
    com_vel_rne(m, L, qvel, S, qfrc_bias);
    static_for<0, NV>([&](auto II) { act[II] = 0; });
    static_for<0, NU>([&](auto UU) {   // motor u drives dof kActDof[u]; ctrlrange +-0.4 (humanoid.xml:6)
      constexpr int u = UU; T c = hmin(hmax(ctrl[u], T(-0.4)), T(0.4)); act[kActDof[u]] += m.act_gear[u] * c; });
    static_for<0, NV>([&](auto II) { constexpr int i = II; K.qfrc_smooth[i] = -L.damping[i] * qvel[i] - qfrc_bias[i] + act[i]; });
    static_for<1, NJNT>([&](auto JJ) { constexpr int j = JJ; K.qfrc_smooth[j + 5] -= m.jnt_stiff[j] * qpos[j + 6]; });   // springref 0
    crb(m, S, F);
    // observation inputs: stored, not kept -- and only by the evaluation the observation is taken from (the last of an env
    // step: 261 words per lane that 19 of 20 evaluations would write for nothing)
    if (keep_obs) {
    static_for<0, NBODY>([&](auto BB) { constexpr int b = BB; for (int k = 0; k < 10; k++) s.obs_cinert[b][k] = S.cinert[b][k]; for (int k = 0; k < 6; k++) s.obs_cvel[b][k] = S.cvel[b][k]; s.obs_xipos_x[b] = S.xipos[b][0]; });
    static_for<0, NV>([&](auto II) { s.obs_qfrc_actuator[II] = act[II]; });
    }
    REX_FENCE(); REX_HSTAMP(t4e); REX_HACC(K, HT_SMOOTH, t4, t4e);   // (HT_SMOOTH: kinematics + everything from com to M)
  }
  REX_FENCE(); REX_HSTAMP(t5);
  factor(F);
  for (int i = 0; i < NV; i++) K.qacc_smooth[i] = K.qfrc_smooth[i];
  solve(F, K.qacc_smooth);
  REX_FENCE(); REX_HSTAMP(t7); REX_HACC(K, HT_FACTOR, t5, t7); REX_HCNT(K, HC_EVALS, 1); REX_HCNT(K, HC_NEFC, K.nefc);
  int it = 0;
  if (K.nefc == 0) { for (int i = 0; i < NV; i++) qacc[i] = K.qacc_smooth[i]; }
  else it = K.nefc <= DUAL_NMAX ? solve_pgs_dual(m, F, K, s, qacc) : solve_pgs(m, F, K, s, qacc);   // the scratch-row variant only for rare pile-ups
  REX_HSTAMP(t8); REX_HACC(K, HT_FORWARD, t0, t8);
  return it;
}

// [3P] mj_integratePos
template <class T>
REX_HD void integrate_pos(T* qpos, const T* qvel, T h) {
  for (int k = 0; k < 3; k++) qpos[k] += h * qvel[k];
  T w[3] = {qvel[3], qvel[4], qvel[5]}, n = hsqrt(dot3(w, w));
  if (n * h > T(1e-15)) {
    T sn, cs; hsincos(T(0.5) * n * h, sn, cs);
    const T sn_n = sn * rcp_t(n);
    T dq[4] = {cs, w[0] * sn_n, w[1] * sn_n, w[2] * sn_n}, r[4];
    qmul(r, qpos + 3, dq); qnorm(r);
    for (int k = 0; k < 4; k++) qpos[3 + k] = r[k];
  }
  for (int k = 0; k < 17; k++) qpos[7 + k] += h * qvel[6 + k];
}

// One mj_step with RK4 ([3P] mj_RungeKutta, N = 4).  The stage accumulators are parked in Scratch while forward() runs.
template <class T>
REX_HD void substep(const Model<T>& m, const Lane<T>& L, T* qpos, T* qvel, const T* ctrl, Kin<T>& K, Scratch<T>& s, bool last_frame = true) {
  const T h = m.timestep;
  static_for<0, NQ>([&](auto KK) { s.rk_q0[KK] = qpos[KK]; });
  static_for<0, NV>([&](auto KK) { s.rk_v0[KK] = qvel[KK]; s.rk_dq[KK] = 0; s.rk_dv[KK] = 0; });
  for (int stage = 0; stage < 4; stage++) {
    T acc[NV];
    forward(m, L, qpos, qvel, ctrl, K, s, acc, last_frame && stage == 3);   // mj_step ends with stage 4's forward: what _get_obs reads
    const T w = (stage == 0 || stage == 3) ? T(1.0 / 6) : T(1.0 / 3), c = stage == 2 ? h : T(0.5) * h;
    T dq[NV], dv[NV];
    static_for<0, NV>([&](auto KK) { constexpr int k = KK; dq[k] = s.rk_dq[k] + w * qvel[k]; dv[k] = s.rk_dv[k] + w * acc[k]; });
    if (stage < 3) {
      static_for<0, NV>([&](auto KK) { constexpr int k = KK; s.rk_dq[k] = dq[k]; s.rk_dv[k] = dv[k]; });
      T vs[NV]; for (int k = 0; k < NV; k++) vs[k] = qvel[k];
      static_for<0, NQ>([&](auto KK) { qpos[KK] = s.rk_q0[KK]; });
      integrate_pos(qpos, vs, c);
      static_for<0, NV>([&](auto KK) { constexpr int k = KK; qvel[k] = s.rk_v0[k] + c * acc[k]; });
    } else {
      static_for<0, NQ>([&](auto KK) { qpos[KK] = s.rk_q0[KK]; });
      static_for<0, NV>([&](auto KK) { constexpr int k = KK; qvel[k] = s.rk_v0[k] + h * dv[k]; });
      integrate_pos(qpos, dq, h);
    }
  }
}

// _get_obs (random_humanoid.py:193-204): qpos[2:], qvel, cinert, cvel, qfrc_actuator, cfrc_ext (= 0)
template <class T, class ObsSink>
REX_HD void emit_obs(const T* qpos, const T* qvel, const Scratch<T>& s, ObsSink&& obs) {
  static_for<2, NQ>([&](auto KK) { constexpr int k = KK; obs(k - 2, qpos[k]); });
  static_for<0, NV>([&](auto KK) { constexpr int k = KK; obs(22 + k, qvel[k]); });
  static_for<0, NBODY>([&](auto BB) { constexpr int b = BB; static_for<0, 10>([&](auto KK) { constexpr int k = KK; obs(45 + 10 * b + k, s.obs_cinert[b][k]); }); });
  static_for<0, NBODY>([&](auto BB) { constexpr int b = BB; static_for<0, 6>([&](auto KK) { constexpr int k = KK; obs(185 + 6 * b + k, s.obs_cvel[b][k]); }); });
  static_for<0, NV>([&](auto KK) { constexpr int k = KK; obs(269 + k, s.obs_qfrc_actuator[k]); });
  for (int k = 0; k < 84; k++) obs(292 + k, T(0));
}

// RandomHumanoidEnv.step + _get_obs (random_humanoid.py:161-216), noise-free.
//   xipos_x: in = data.xipos[:,0] left by the previous forward (mass_center() reads it before do_simulation);
//            out = the same after this step (stage-4 forward of the last mj_step).
//   obs(k, value) is called for k = 0..375 in order.
template <class T, class ObsSink>
REX_HD void env_step(const Model<T>& m, const Lane<T>& L, T* qpos, T* qvel, const T* action, T* xipos_x, Kin<T>& K, Scratch<T>& s,
                     T& reward, bool& done, ObsSink&& obs, T* terms = nullptr) {
  T mt = 0, s0 = 0, s1 = 0, asq = 0;
  for (int b = 0; b < NBODY; b++) { mt += L.mass[b]; s0 += L.mass[b] * xipos_x[b]; }
  for (int u = 0; u < NU; u++) asq += action[u] * action[u];       // data.ctrl holds the raw action (:167)
  for (int f = 0; f < 5; f++) substep(m, L, qpos, qvel, action, K, s, f == 4);   // frame_skip 5 (:41)
  static_for<0, NBODY>([&](auto BB) { constexpr int b = BB; xipos_x[b] = s.obs_xipos_x[b]; s1 += L.mass[b] * xipos_x[b]; });
  const T dt = m.timestep * T(5);
  reward = T(1.25) * (s1 / mt - s0 / mt) / dt - T(0.1) * asq - T(0) /* cfrc_ext = 0, SURVEY Q15 */ + T(5);
  if (terms) { terms[0] = T(1.25) * (s1 / mt - s0 / mt) / dt; terms[1] = -T(0.1) * asq; terms[2] = T(5); terms[3] = -T(0); }   // info dict :182-187
  done = (qpos[2] < T(1.0)) || (qpos[2] > T(2.0));                 // :173
  emit_obs(qpos, qvel, s, obs);
}

// observation right after set_state / reset: sim.forward() at the given state (jinja_mujoco_env.py:146-154).  Of that forward
// the observation reads cinert, cvel, xipos and qfrc_actuator only (random_humanoid.py:193-204; data.ctrl is zero after
// sim.reset(), so qfrc_actuator = 0): kinematics -> com -> velocities, in forward()'s order and arithmetic -- no collision, no
// mass matrix, no solver (the reset launch of every step used to pay a whole evaluation for them).
template <class T, class ObsSink>
REX_HD void env_reset_obs(const Model<T>& m, const Lane<T>& L, const T* qpos, const T* qvel, T* xipos_x, Kin<T>& K, Scratch<T>& s, ObsSink&& obs) {
  K.overflow = 0; K.ncon = 0; K.nefc = 0;
  Smooth<T> S;
  kinematics(m, qpos, S, s);
  com_pos(m, L, S, s);
  T qfrc_bias[NV];
  com_vel_rne(m, L, qvel, S, qfrc_bias);
  static_for<0, NBODY>([&](auto BB) { constexpr int b = BB; for (int k = 0; k < 10; k++) s.obs_cinert[b][k] = S.cinert[b][k]; for (int k = 0; k < 6; k++) s.obs_cvel[b][k] = S.cvel[b][k]; s.obs_xipos_x[b] = S.xipos[b][0]; });
  static_for<0, NV>([&](auto II) { s.obs_qfrc_actuator[II] = T(0); });
  static_for<0, NBODY>([&](auto BB) { constexpr int b = BB; xipos_x[b] = s.obs_xipos_x[b]; });
  emit_obs(qpos, qvel, s, obs);
}

}  // namespace hum
}  // namespace rex
