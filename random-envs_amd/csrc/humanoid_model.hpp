// humanoid_model.hpp -- the product's own "model compiler" for random_envs/jinja/assets/humanoid.xml
// (replaces load_model_from_xml + MjSim, jinja_mujoco_env.py:92-97).  Host-side, double precision;
// the result is converted to float and uploaded to __constant__ memory once per process.
// MuJoCo-2.1.0 compile rules restated ([3P], provenance in oracle/mjo_core.c): capsule mass
// 1000*pi*r^2*(L+r), inertiafromgeom, parent-child / same-weld-body collision filtering,
// dof_invweight0 = diag(M^-1), body_invweight0 = mean diag of J M^-1 J^T at qpos0.
#pragma once
#include <string.h>

#include "humanoid_engine.hpp"

namespace rex {
namespace hum {

struct BodyDef { int parent; double pos[3]; double quat[4]; };
struct JointDef { int body; double armature, axis[3], damping, pos[3], lo, hi, stiffness; };
struct GeomDef { int body; int type; double a[3], b[3], radius; };   // capsule: fromto a->b; sphere: centre a

inline void build_model(Model<double>& m) {
  memset(&m, 0, sizeof m);
  const double D2R = 3.14159265358979323846 / 180.0, PI = 3.14159265358979323846;
  // bodies: humanoid.xml:30-91 (0 = world)
  static const BodyDef B[NBODY] = {
      {0, {0, 0, 0}, {1, 0, 0, 0}},
      {0, {0, 0, 1.4}, {1, 0, 0, 0}},                 // 1 torso :30
      {1, {-.01, 0, -0.260}, {1.000, 0, -0.002, 0}},  // 2 lwaist :36
      {2, {0, 0, -0.165}, {1.000, 0, -0.002, 0}},     // 3 pelvis :40
      {3, {0, -0.1, -0.04}, {1, 0, 0, 0}},            // 4 right_thigh :43
      {4, {0, 0.01, -0.403}, {1, 0, 0, 0}},           // 5 right_shin :48
      {5, {0, 0, -0.45}, {1, 0, 0, 0}},               // 6 right_foot :51
      {3, {0, 0.1, -0.04}, {1, 0, 0, 0}},             // 7 left_thigh :56
      {7, {0, -0.01, -0.403}, {1, 0, 0, 0}},          // 8 left_shin :61
      {8, {0, 0, -0.45}, {1, 0, 0, 0}},               // 9 left_foot :64
      {1, {0, -0.17, 0.06}, {1, 0, 0, 0}},            // 10 right_upper_arm :70
      {10, {.18, -.18, -.18}, {1, 0, 0, 0}},          // 11 right_lower_arm :74
      {1, {0, 0.17, 0.06}, {1, 0, 0, 0}},             // 12 left_upper_arm :81
      {12, {.18, .18, -.18}, {1, 0, 0, 0}}};          // 13 left_lower_arm :85
  // hinges in XML order (joint 0 is the free root :32); default joint: armature 1 damping 1 limited (:4)
  static const JointDef J[17] = {
      {2, 0.02, {0, 0, 1}, 5, {0, 0, 0.065}, -45, 45, 20},     // abdomen_z :38
      {2, 0.02, {0, 1, 0}, 5, {0, 0, 0.065}, -75, 30, 10},     // abdomen_y :39
      {3, 0.02, {1, 0, 0}, 5, {0, 0, 0.1}, -35, 35, 10},       // abdomen_x :41
      {4, 0.01, {1, 0, 0}, 5, {0, 0, 0}, -25, 5, 10},          // right_hip_x :44
      {4, 0.01, {0, 0, 1}, 5, {0, 0, 0}, -60, 35, 10},         // right_hip_z :45
      {4, 0.0080, {0, 1, 0}, 5, {0, 0, 0}, -110, 20, 20},      // right_hip_y :46
      {5, 0.0060, {0, -1, 0}, 1, {0, 0, .02}, -160, -2, 0},    // right_knee :49
      {7, 0.01, {-1, 0, 0}, 5, {0, 0, 0}, -25, 5, 10},         // left_hip_x :57
      {7, 0.01, {0, 0, -1}, 5, {0, 0, 0}, -60, 35, 10},        // left_hip_z :58
      {7, 0.01, {0, 1, 0}, 5, {0, 0, 0}, -110, 20, 20},        // left_hip_y :59
      {8, 0.0060, {0, -1, 0}, 1, {0, 0, .02}, -160, -2, 1},    // left_knee :62
      {10, 0.0068, {2, 1, 1}, 1, {0, 0, 0}, -85, 60, 1},       // right_shoulder1 :71
      {10, 0.0051, {0, -1, 1}, 1, {0, 0, 0}, -85, 60, 1},      // right_shoulder2 :72
      {11, 0.0028, {0, -1, 1}, 1, {0, 0, 0}, -90, 50, 0},      // right_elbow :75
      {12, 0.0068, {2, -1, 1}, 1, {0, 0, 0}, -60, 85, 1},      // left_shoulder1 :82
      {12, 0.0051, {0, 1, 1}, 1, {0, 0, 0}, -60, 85, 1},       // left_shoulder2 :83
      {13, 0.0028, {0, -1, -1}, 1, {0, 0, 0}, -90, 50, 0}};    // left_elbow :86
  static const GeomDef G[NGEOM] = {
      {0, G_PLANE, {0, 0, 0}, {0, 0, 0}, 0},                                       // floor :28
      {1, G_CAPSULE, {0, -.07, 0}, {0, .07, 0}, 0.07},                             // torso1 :33
      {1, G_SPHERE, {0, 0, .19}, {0, 0, 0}, 0.09},                                 // head :34
      {1, G_CAPSULE, {-.01, -.06, -.12}, {-.01, .06, -.12}, 0.06},                 // uwaist :35
      {2, G_CAPSULE, {0, -.06, 0}, {0, .06, 0}, 0.06},                             // lwaist :37
      {3, G_CAPSULE, {-.02, -.07, 0}, {-.02, .07, 0}, 0.09},                       // butt :42
      {4, G_CAPSULE, {0, 0, 0}, {0, 0.01, -.34}, 0.06},                            // right_thigh1 :47
      {5, G_CAPSULE, {0, 0, 0}, {0, 0, -.3}, 0.049},                               // right_shin1 :50
      {6, G_SPHERE, {0, 0, 0.1}, {0, 0, 0}, 0.075},                                // right_foot :52
      {7, G_CAPSULE, {0, 0, 0}, {0, -0.01, -.34}, 0.06},                           // left_thigh1 :60
      {8, G_CAPSULE, {0, 0, 0}, {0, 0, -.3}, 0.049},                               // left_shin1 :63
      {9, G_SPHERE, {0, 0, 0.1}, {0, 0, 0}, 0.075},                                // left_foot :65
      {10, G_CAPSULE, {0, 0, 0}, {.16, -.16, -.16}, 0.04},                         // right_uarm1 :73
      {11, G_CAPSULE, {.01, .01, .01}, {.17, .17, .17}, 0.031},                    // right_larm :76
      {11, G_SPHERE, {.18, .18, .18}, {0, 0, 0}, 0.04},                            // right_hand :77
      {12, G_CAPSULE, {0, 0, 0}, {.16, .16, -.16}, 0.04},                          // left_uarm1 :84
      {13, G_CAPSULE, {.01, -.01, .01}, {.17, -.17, .17}, 0.031},                  // left_larm :87
      {13, G_SPHERE, {.18, -.18, .18}, {0, 0, 0}, 0.04}};                          // left_hand :88
  // motors :106-122 (joint index into J, gear)
  static const int MJ[NU] = {1, 0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  static const double MG[NU] = {100, 100, 100, 100, 100, 300, 200, 100, 100, 300, 200, 25, 25, 25, 25, 25, 25};

  for (int b = 0; b < NBODY; b++) {
    m.body_parent[b] = B[b].parent; m.body_jntadr[b] = -1; m.body_dofadr[b] = -1;
    for (int k = 0; k < 3; k++) m.body_pos[b][k] = B[b].pos[k];
    double q[4] = {B[b].quat[0], B[b].quat[1], B[b].quat[2], B[b].quat[3]}; qnorm(q);
    for (int k = 0; k < 4; k++) m.body_quat[b][k] = q[k];
  }
  // joints / dofs
  m.jnt_body[0] = 1; m.jnt_qadr[0] = 0; m.jnt_dadr[0] = 0; m.body_jntadr[1] = 0; m.body_jntnum[1] = 1; m.body_dofadr[1] = 0; m.body_dofnum[1] = 6;
  for (int d = 0; d < 6; d++) { m.dof_body[d] = 1; m.dof_parent[d] = d - 1; m.dof_armature[d] = 0; m.dof_damping0[d] = 0; }
  for (int j = 0; j < 17; j++) {
    int jj = j + 1, b = J[j].body, d = 6 + j;
    m.jnt_body[jj] = b; m.jnt_qadr[jj] = 7 + j; m.jnt_dadr[jj] = d;
    if (m.body_jntadr[b] < 0) { m.body_jntadr[b] = jj; m.body_dofadr[b] = d; }
    m.body_jntnum[b]++; m.body_dofnum[b]++;
    double n = sqrt(J[j].axis[0] * J[j].axis[0] + J[j].axis[1] * J[j].axis[1] + J[j].axis[2] * J[j].axis[2]);
    for (int k = 0; k < 3; k++) { m.jnt_pos[jj][k] = J[j].pos[k]; m.jnt_axis[jj][k] = J[j].axis[k] / n; }
    m.jnt_lo[jj] = J[j].lo * D2R; m.jnt_hi[jj] = J[j].hi * D2R; m.jnt_stiff[jj] = J[j].stiffness;
    m.dof_body[d] = b; m.dof_armature[d] = J[j].armature; m.dof_damping0[d] = J[j].damping;
  }
  // dof tree: previous dof of the same body, else last dof of the nearest ancestor that has dofs
  for (int d = 6; d < NV; d++) {
    int b = m.dof_body[d];
    if (d > m.body_dofadr[b]) { m.dof_parent[d] = d - 1; continue; }
    int p = m.body_parent[b];
    while (p > 0 && m.body_dofnum[p] == 0) p = m.body_parent[p];
    m.dof_parent[d] = p > 0 ? m.body_dofadr[p] + m.body_dofnum[p] - 1 : -1;
  }
  // geoms + inertiafromgeom
  double gmass[NGEOM], ginert[NGEOM][9];
  for (int g = 0; g < NGEOM; g++) {
    m.geom_type[g] = G[g].type; m.geom_body[g] = G[g].body; m.geom_rad[g] = G[g].radius;
    gmass[g] = 0; for (int k = 0; k < 9; k++) ginert[g][k] = 0;
    if (G[g].type == G_SPHERE) {
      for (int k = 0; k < 3; k++) { m.geom_pos[g][k] = G[g].a[k]; m.geom_axis[g][k] = k == 2; }
      double r = G[g].radius; gmass[g] = 1000 * 4.0 * PI * r * r * r / 3.0;
      for (int k = 0; k < 3; k++) ginert[g][4 * k] = 2 * gmass[g] * r * r / 5;
    } else if (G[g].type == G_CAPSULE) {
      double v[3], L = 0;
      for (int k = 0; k < 3; k++) { v[k] = G[g].a[k] - G[g].b[k]; L += v[k] * v[k]; m.geom_pos[g][k] = 0.5 * (G[g].a[k] + G[g].b[k]); }
      L = sqrt(L);
      for (int k = 0; k < 3; k++) { v[k] /= L; m.geom_axis[g][k] = v[k]; }
      m.geom_half[g] = L / 2;
      double r = G[g].radius;
      gmass[g] = 1000 * PI * r * r * (L + r);                       // 2.1.0 capsule volume
      double ms = gmass[g] * r / (L + r), mc = gmass[g] - ms;
      double Iperp = mc * (3 * r * r + L * L) / 12 + 2 * ms * r * r / 5 + ms * L * (3 * r + 2 * L) / 8;
      double Iax = mc * r * r / 2 + 2 * ms * r * r / 5;
      for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) ginert[g][3 * i + k] = Iperp * ((i == k) - v[i] * v[k]) + Iax * v[i] * v[k];
    } else { m.geom_axis[g][2] = 1; }
  }
  for (int b = 1; b < NBODY; b++) {
    double M = 0, c[3] = {0, 0, 0};
    for (int g = 1; g < NGEOM; g++) if (G[g].body == b) { M += gmass[g]; for (int k = 0; k < 3; k++) c[k] += gmass[g] * m.geom_pos[g][k]; }
    for (int k = 0; k < 3; k++) c[k] /= M;
    double I[9] = {0};
    for (int g = 1; g < NGEOM; g++) if (G[g].body == b) {
      double d[3] = {m.geom_pos[g][0] - c[0], m.geom_pos[g][1] - c[1], m.geom_pos[g][2] - c[2]}, dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) I[3 * i + k] += ginert[g][3 * i + k] + gmass[g] * ((i == k) * dd - d[i] * d[k]);
    }
    m.body_mass0[b] = M; for (int k = 0; k < 3; k++) m.body_ipos[b][k] = c[k];
    m.body_inertia[b][0] = I[0]; m.body_inertia[b][1] = I[4]; m.body_inertia[b][2] = I[8];
    m.body_inertia[b][3] = I[1]; m.body_inertia[b][4] = I[2]; m.body_inertia[b][5] = I[5];
  }
  m.subtreemass_root = 0; for (int b = 1; b < NBODY; b++) m.subtreemass_root += m.body_mass0[b];
  // collision pairs: body pairs ascending, geoms of body1 x geoms of body2; filters of [3P] mj_collision
  int weld[NBODY]; weld[0] = 0;
  for (int b = 1; b < NBODY; b++) weld[b] = m.body_jntnum[b] ? b : weld[m.body_parent[b]];
  m.npair = 0;
  for (int b1 = 0; b1 < NBODY; b1++) for (int b2 = b1 + 1; b2 < NBODY; b2++) {
    int w1 = weld[b1], w2 = weld[b2], wp1 = weld[m.body_parent[w1]], wp2 = weld[m.body_parent[w2]];
    if (w1 == w2) continue;
    if (w1 != 0 && w2 != 0 && (w1 == wp2 || w2 == wp1)) continue;
    for (int g1 = 0; g1 < NGEOM; g1++) for (int g2 = 0; g2 < NGEOM; g2++) {
      if (G[g1].body != b1 || G[g2].body != b2) continue;
      int a = g1, b = g2; if (G[a].type > G[b].type) { int t = a; a = b; b = t; }
      int p = m.npair++;
      m.pair_g1[p] = a; m.pair_g2[p] = b;
      m.pair_dim[p] = G[a].type == G_PLANE ? 3 : 1;      // max(condim): floor 3 (:28), bodies 1 (:5)
      m.pair_mu[p] = 1.0;                                // max(friction[0]): floor 1, default geom 1
    }
  }
  for (int u = 0; u < NU; u++) { m.act_dof[u] = 6 + MJ[u]; m.act_gear[u] = MG[u]; }
  for (int k = 0; k < NQ; k++) m.qpos0[k] = 0;
  m.qpos0[2] = 1.4; m.qpos0[3] = 1;
  // option: RK4, PGS, 50 iterations, dt 0.003 (:9); geom margin 0.001 (:5); default solref (.02,1) solimp (.9,.95,.001)
  m.timestep = 0.003; m.gravity = 9.81; m.iterations = 50; m.tolerance = 1e-8; m.margin = 0.001;
  m.dmin = 0.9; m.dmax = 0.95; m.width = 0.001;
  double tc = 0.02 > 2 * m.timestep ? 0.02 : 2 * m.timestep;
  m.K = 1 / (m.dmax * m.dmax * tc * tc); m.B = 2 / (m.dmax * tc);
  // mj_setConst at qpos0
  Lane<double> L; L.mass[0] = 0; for (int b = 1; b < NBODY; b++) L.mass[b] = m.body_mass0[b];
  for (int d = 0; d < NV; d++) L.damping[d] = m.dof_damping0[d];
  Scratch<double>* s = new Scratch<double>();
  Smooth<double>* K = new Smooth<double>();
  kinematics(m, m.qpos0, *K, *s); com_pos(m, L, *K, *s);
  MassFactor<double> F; crb(m, *K, F);
  double tr = 0; for (int d = 0; d < NV; d++) tr += F.get(d, d);
  m.meaninertia = tr / NV;
  factor(F);
  double Minv[NV][NV];
  for (int d = 0; d < NV; d++) { double e[NV] = {0}; e[d] = 1; solve(F, e); for (int k = 0; k < NV; k++) Minv[k][d] = e[k]; }
  for (int d = 0; d < NV; d++) m.dof_invw[d] = Minv[d][d];
  for (int g = 0; g < 2; g++) { double a = (m.dof_invw[3 * g] + m.dof_invw[3 * g + 1] + m.dof_invw[3 * g + 2]) / 3; for (int k = 0; k < 3; k++) m.dof_invw[3 * g + k] = a; }
  for (int b = 1; b < NBODY; b++) {
    double A[6];
    const int mask = body_dof_mask(b);
    for (int r = 0; r < 6; r++) {
      double row[1][NV], e[3] = {0, 0, 0}; e[r % 3] = 1;
      if (r < 3) jac_dirs<1>(*s, 0, mask, K->xipos[b], e, row);
      else for (int i = 0; i < NV; i++) row[0][i] = (i >= 3 && ((mask >> i) & 1)) ? dual(*s, GEO_DOF + i * 6 + 3 + (r - 3)) : 0.0;
      double a = 0; for (int i = 0; i < NV; i++) for (int k = 0; k < NV; k++) a += row[0][i] * Minv[i][k] * row[0][k];
      A[r] = a;
    }
    m.body_invw[b][0] = (A[0] + A[1] + A[2]) / 3; m.body_invw[b][1] = (A[3] + A[4] + A[5]) / 3;
  }
  delete s; delete K;
  for (int g = 0; g < NGEOM; g++) m.geom_bound[g] = m.geom_rad[g] + m.geom_half[g];
  // packed pair records for the device loop
  for (int p = 0; p < m.npair; p++) {
    PairRec<double>& r = m.pair[p];
    const int g1 = m.pair_g1[p], g2 = m.pair_g2[p], b1 = m.geom_body[g1], b2 = m.geom_body[g2];
    r.g1 = g1; r.g2 = g2; r.t1 = m.geom_type[g1]; r.t2 = m.geom_type[g2]; r.dim = m.pair_dim[p];
    r.mask1 = body_dof_mask(b1); r.mask2 = body_dof_mask(b2); r.b1 = b1; r.b2 = b2; r.pad = 0;
    r.mu = m.pair_mu[p]; r.r1 = m.geom_rad[g1]; r.l1 = m.geom_half[g1]; r.r2 = m.geom_rad[g2]; r.l2 = m.geom_half[g2];
    r.tran = m.body_invw[b1][0] + m.body_invw[b2][0];
  }
}

// the engine's compile-time tables (humanoid_engine.hpp) against the tables derived from the XML transcription above
inline bool check_topology(const Model<double>& m) {
  bool ok = true;
  for (int d = 0; d < NV; d++) ok = ok && m.dof_parent[d] == kDofParent[d] && m.dof_body[d] == kDofBody[d];
  for (int b = 0; b < NBODY; b++) {
    ok = ok && m.body_parent[b] == kBodyParent[b] && m.body_dofnum[b] == kBodyDofNum[b];
    if (m.body_dofnum[b] > 0) ok = ok && m.body_dofadr[b] == kBodyDofAdr[b];
  }
  for (int j = 1; j < NJNT; j++) ok = ok && m.jnt_dadr[j] == j + 5 && m.jnt_qadr[j] == j + 6;
  for (int g = 0; g < NGEOM; g++) ok = ok && m.geom_body[g] == kGeomBody[g] && m.geom_type[g] == kGeomType[g];
  for (int g = 1; g < NGEOM; g++)   // the broad phase's literal size bounds: never below the model's sizes, and tight
    ok = ok && double(kGeomRadUB[g]) >= m.geom_rad[g] && double(kGeomRadUB[g]) - m.geom_rad[g] < 1e-5
            && double(kGeomHalfUB[g]) >= m.geom_half[g] && double(kGeomHalfUB[g]) - m.geom_half[g] < 1e-5;
  ok = ok && m.npair == kPairs.n;
  for (int p = 0; p < kPairs.n && p < m.npair; p++) ok = ok && m.pair_g1[p] == kPairs.g1[p] && m.pair_g2[p] == kPairs.g2[p];
  for (int u = 0; u < NU; u++) ok = ok && m.act_dof[u] == kActDof[u];
  return ok;
}

template <class T>
inline void convert_model(const Model<double>& a, Model<T>& b) {
#define CP(f) do { const double* s_ = reinterpret_cast<const double*>(&a.f); T* d_ = reinterpret_cast<T*>(&b.f); \
    for (size_t k_ = 0; k_ < sizeof(a.f) / sizeof(double); k_++) d_[k_] = T(s_[k_]); } while (0)
#define CI(f) memcpy(&b.f, &a.f, sizeof(a.f))
  CI(body_parent); CI(body_jntadr); CI(body_jntnum); CI(body_dofadr); CI(body_dofnum);
  CP(body_pos); CP(body_quat); CP(body_ipos); CP(body_inertia); CP(body_mass0); CP(subtreemass_root); CP(body_invw); CP(dof_invw);
  CI(jnt_body); CI(jnt_qadr); CI(jnt_dadr); CP(jnt_pos); CP(jnt_axis); CP(jnt_lo); CP(jnt_hi); CP(jnt_stiff);
  CI(dof_body); CI(dof_parent); CP(dof_armature); CP(dof_damping0);
  CI(geom_type); CI(geom_body); CP(geom_pos); CP(geom_axis); CP(geom_rad); CP(geom_half); CP(geom_bound);
  CI(npair); CI(pair_g1); CI(pair_g2); CI(pair_dim); CP(pair_mu); CI(act_dof); CP(act_gear); CP(qpos0);
  CP(K); CP(B); CP(dmin); CP(dmax); CP(width); CP(margin); CP(timestep); CP(gravity); CP(meaninertia); CP(tolerance);
  CI(iterations);
  for (int p = 0; p < MAXPAIR; p++) {
    const PairRec<double>& x = a.pair[p]; PairRec<T>& y = b.pair[p];
    y.g1 = x.g1; y.g2 = x.g2; y.t1 = x.t1; y.t2 = x.t2; y.dim = x.dim; y.mask1 = x.mask1; y.mask2 = x.mask2; y.b1 = x.b1; y.b2 = x.b2;
    y.mu = T(x.mu); y.r1 = T(x.r1); y.l1 = T(x.l1); y.r2 = T(x.r2); y.l2 = T(x.l2); y.tran = T(x.tran);
    y.pad = 0;
  }
#undef CP
#undef CI
}

}  // namespace hum
}  // namespace rex
