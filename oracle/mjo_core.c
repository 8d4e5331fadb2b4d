/*
 * mjo_core.c -- ORACLE (test infrastructure, NOT product code).  See mjo.h.
 *
 * Generic fp64 restatement of MuJoCo 2.1.0's forward dynamics + mj_step for small kinematic
 * trees (free / slide / hinge joints; plane / sphere / capsule geoms; motors on joints).
 * The call site in the reference is `self.sim.step()` (jinja_mujoco_env.py:170-173); the
 * algorithm itself lives in libmujoco210 (third party, not under /root/reference), so every
 * function below cites the MuJoCo pipeline stage it restates ("[3P] mj_xxx") rather than a
 * reference file:line.  Parity with real mujoco-py is UNPINNED (see mjo.h).
 *
 * Deliberate choice: the rigid-body terms (M, bias) are computed with plain world-frame
 * Jacobians / Newton-Euler instead of MuJoCo's subtree-COM spatial algebra.  M(q) and c(q,v)
 * are properties of the physical model, not of the algorithm, so this is a restatement of the
 * same quantities and gives an independent check of the planar reduction used by the kernels.
 */
#include "mjo.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999
#define PI 3.14159265358979323846

/* ------------------------------------------------------------------ small math helpers -- */
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static void copy3(double* r, const double* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static void zero3(double* r) { r[0] = r[1] = r[2] = 0; }
static void add3(double* r, const double* a, const double* b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static void sub3(double* r, const double* a, const double* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static void addscl3(double* r, const double* a, double s) { r[0] += a[0] * s; r[1] += a[1] * s; r[2] += a[2] * s; }
static double norm3(const double* a) { return sqrt(dot3(a, a)); }
/* [3P] mju_normalize3: returns the norm; vectors shorter than MINVAL become (1,0,0) */
static double normalize3(double* a) {
  double n = norm3(a);
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; } else { a[0] /= n; a[1] /= n; a[2] /= n; }
  return n;
}
static void q_mul(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void q_normalize(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
static void q_to_mat(double* m, const double* q) { /* row-major 3x3 */
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z);         m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z);         m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y);         m[7] = 2 * (y * z + w * x);         m[8] = w * w - x * x - y * y + z * z;
}
static void mat_vec(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void q_axisangle(double* q, const double* axis, double ang) {
  double s = sin(ang / 2);
  q[0] = cos(ang / 2); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
/* quaternion that rotates (0,0,1) onto vec ([3P] mjuu_z2quat) */
static void z2quat(double* q, const double* vec_in) {
  double vec[3] = {vec_in[0], vec_in[1], vec_in[2]}, z[3] = {0, 0, 1}, ax[3];
  normalize3(vec);
  cross3(ax, z, vec);
  double s = norm3(ax);
  if (s < 1e-10) { ax[0] = 1; ax[1] = 0; ax[2] = 0; } else { ax[0] /= s; ax[1] /= s; ax[2] /= s; }
  double ang = atan2(s, vec[2]);
  q_axisangle(q, ax, ang);
}
/* dense symmetric positive definite solve helpers (n <= MJO_MAXV) */
static int chol_factor(double* L, const double* A, int n, int ld) {
  for (int i = 0; i < n; i++) {
    for (int j = 0; j <= i; j++) {
      double s = A[i * ld + j];
      for (int k = 0; k < j; k++) s -= L[i * ld + k] * L[j * ld + k];
      if (i == j) { if (s < MINVAL) s = MINVAL; L[i * ld + i] = sqrt(s); }
      else L[i * ld + j] = s / L[j * ld + j];
    }
  }
  return 0;
}
static void chol_solve(const double* L, double* x, int n, int ld) {
  for (int i = 0; i < n; i++) { double s = x[i]; for (int k = 0; k < i; k++) s -= L[i * ld + k] * x[k]; x[i] = s / L[i * ld + i]; }
  for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= L[k * ld + i] * x[k]; x[i] = s / L[i * ld + i]; }
}

/* ------------------------------------------------------------------ model building ------ */
typedef struct { /* joint defaults of a <default><joint .../> block */
  double armature, damping, stiffness; int limited;
  double solreflimit[2], solimplimit[5], margin;
} JntDef;
typedef struct { /* geom defaults */
  int contype, conaffinity, condim; double friction[3], margin, gap, solref[2], solimp[5], solmix, density;
} GeomDef;

static void jntdef_init(JntDef* j) { /* MuJoCo global defaults [3P] */
  j->armature = 0; j->damping = 0; j->stiffness = 0; j->limited = 0; j->margin = 0;
  j->solreflimit[0] = 0.02; j->solreflimit[1] = 1;
  double si[5] = {0.9, 0.95, 0.001, 0.5, 2}; memcpy(j->solimplimit, si, sizeof si);
}
static void geomdef_init(GeomDef* g) {
  g->contype = 1; g->conaffinity = 1; g->condim = 3; g->margin = 0; g->gap = 0; g->solmix = 1; g->density = 1000;
  g->friction[0] = 1; g->friction[1] = 0.005; g->friction[2] = 0.0001;
  g->solref[0] = 0.02; g->solref[1] = 1;
  double si[5] = {0.9, 0.95, 0.001, 0.5, 2}; memcpy(g->solimp, si, sizeof si);
}

static void model_init(mjoModel* m) {
  memset(m, 0, sizeof *m);
  m->timestep = 0.002; m->gravity[2] = -9.81; m->tolerance = 1e-8; m->impratio = 1;
  m->integrator = MJO_INT_EULER; m->solver = MJO_SOL_NEWTON; m->iterations = 100;
  /* world body */
  m->nbody = 1; m->body_parent[0] = 0; m->body_quat[0][0] = 1; m->body_jntadr[0] = -1; m->body_dofadr[0] = -1;
}
static int add_body(mjoModel* m, int parent, const double* pos, const double* quat) {
  int b = m->nbody++;
  m->body_parent[b] = parent; copy3(m->body_pos[b], pos);
  if (quat) { memcpy(m->body_quat[b], quat, 4 * sizeof(double)); q_normalize(m->body_quat[b]); }
  else { m->body_quat[b][0] = 1; }
  m->body_jntadr[b] = -1; m->body_jntnum[b] = 0; m->body_dofadr[b] = -1; m->body_dofnum[b] = 0;
  return b;
}
/* range in radians (or metres); joints must be added in body order */
static int add_joint(mjoModel* m, int body, int type, const double* pos, const double* axis, double ref,
                     const JntDef* def, int limited, double lo, double hi) {
  int j = m->njnt++;
  m->jnt_type[j] = type; m->jnt_body[j] = body; m->jnt_qposadr[j] = m->nq; m->jnt_dofadr[j] = m->nv;
  if (m->body_jntadr[body] < 0) { m->body_jntadr[body] = j; m->body_dofadr[body] = m->nv; }
  m->body_jntnum[body]++;
  if (pos) copy3(m->jnt_pos[j], pos);
  if (axis) { copy3(m->jnt_axis[j], axis); normalize3(m->jnt_axis[j]); } else m->jnt_axis[j][2] = 1;
  m->jnt_ref[j] = ref; m->jnt_springref[j] = 0;
  m->jnt_limited[j] = limited; m->jnt_range[j][0] = lo; m->jnt_range[j][1] = hi;
  m->jnt_stiffness[j] = def->stiffness; m->jnt_margin[j] = def->margin;
  memcpy(m->jnt_solref[j], def->solreflimit, sizeof def->solreflimit);
  memcpy(m->jnt_solimp[j], def->solimplimit, sizeof def->solimplimit);
  int nq = (type == MJO_JNT_FREE) ? 7 : 1, nv = (type == MJO_JNT_FREE) ? 6 : 1;
  for (int k = 0; k < nv; k++) {
    int d = m->nv + k;
    m->dof_body[d] = body; m->dof_jnt[d] = j; m->dof_armature[d] = def->armature; m->dof_damping[d] = def->damping;
  }
  if (type == MJO_JNT_FREE) {
    /* qpos0 filled in compile from the body pose */
  } else {
    m->qpos0[m->nq] = ref;
  }
  m->nq += nq; m->nv += nv; m->body_dofnum[body] += nv;
  return j;
}
static int add_geom_common(mjoModel* m, int body, int type, const GeomDef* def) {
  int g = m->ngeom++;
  m->geom_type[g] = type; m->geom_body[g] = body;
  m->geom_contype[g] = def->contype; m->geom_conaffinity[g] = def->conaffinity; m->geom_condim[g] = def->condim;
  memcpy(m->geom_friction[g], def->friction, sizeof def->friction);
  m->geom_margin[g] = def->margin; m->geom_gap[g] = def->gap; m->geom_solmix[g] = def->solmix;
  memcpy(m->geom_solref[g], def->solref, sizeof def->solref);
  memcpy(m->geom_solimp[g], def->solimp, sizeof def->solimp);
  m->geom_density[g] = def->density; m->geom_quat[g][0] = 1;
  return g;
}
static int add_plane(mjoModel* m, const GeomDef* def) {
  return add_geom_common(m, 0, MJO_GEOM_PLANE, def);
}
/* capsule given by its two end points in the BODY frame ([3P] fromto handling of the MJCF compiler:
 * pos = midpoint, size[1] = half length, z axis of the geom along from - to) */
static int add_capsule_fromto(mjoModel* m, int body, const double* from, const double* to, double radius,
                              const GeomDef* def) {
  int g = add_geom_common(m, body, MJO_GEOM_CAPSULE, def);
  double vec[3]; sub3(vec, from, to);
  m->geom_size[g][0] = radius; m->geom_size[g][1] = norm3(vec) / 2;
  for (int k = 0; k < 3; k++) m->geom_pos[g][k] = 0.5 * (from[k] + to[k]);
  z2quat(m->geom_quat[g], vec);
  return g;
}
/* capsule given by pos / axisangle about y / half length (half_cheetah.xml style) */
static int add_capsule_pos(mjoModel* m, int body, const double* pos, const double* quat, double radius,
                           double half, const GeomDef* def) {
  int g = add_geom_common(m, body, MJO_GEOM_CAPSULE, def);
  m->geom_size[g][0] = radius; m->geom_size[g][1] = half; copy3(m->geom_pos[g], pos);
  if (quat) memcpy(m->geom_quat[g], quat, 4 * sizeof(double));
  return g;
}
static int add_sphere(mjoModel* m, int body, const double* pos, double radius, const GeomDef* def) {
  int g = add_geom_common(m, body, MJO_GEOM_SPHERE, def);
  m->geom_size[g][0] = radius; copy3(m->geom_pos[g], pos);
  return g;
}
static int add_motor(mjoModel* m, int jnt, double gear, double lo, double hi) {
  int u = m->nu++;
  m->act_dof[u] = m->jnt_dofadr[jnt]; m->act_gear[u] = gear; m->act_ctrllimited[u] = 1;
  m->act_ctrlrange[u][0] = lo; m->act_ctrlrange[u][1] = hi;
  return u;
}

/* [3P] mj_contactParam: mixing rule for a dynamically generated geom pair (equal priority) */
static void mix_pair_params(const mjoModel* m, int g1, int g2, int p, mjoModel* out) {
  out->pair_dim[p] = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
  out->pair_margin[p] = fmax(m->geom_margin[g1], m->geom_margin[g2]);
  out->pair_gap[p] = fmax(m->geom_gap[g1], m->geom_gap[g2]);
  double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2], mix;
  if (s1 >= MINVAL && s2 >= MINVAL) mix = s1 / (s1 + s2);
  else if (s1 < MINVAL && s2 < MINVAL) mix = 0.5;
  else if (s1 < MINVAL) mix = 0.0; else mix = 1.0;
  if (m->geom_solref[g1][0] > 0 && m->geom_solref[g2][0] > 0)
    for (int k = 0; k < 2; k++) out->pair_solref[p][k] = mix * m->geom_solref[g1][k] + (1 - mix) * m->geom_solref[g2][k];
  else
    for (int k = 0; k < 2; k++) out->pair_solref[p][k] = fmin(m->geom_solref[g1][k], m->geom_solref[g2][k]);
  for (int k = 0; k < 5; k++) out->pair_solimp[p][k] = mix * m->geom_solimp[g1][k] + (1 - mix) * m->geom_solimp[g2][k];
  double f[3];
  for (int k = 0; k < 3; k++) f[k] = fmax(m->geom_friction[g1][k], m->geom_friction[g2][k]);
  out->pair_friction[p][0] = f[0]; out->pair_friction[p][1] = f[0]; out->pair_friction[p][2] = f[1];
  out->pair_friction[p][3] = f[2]; out->pair_friction[p][4] = f[2];
}
/* explicit <pair>: parameters not given in the XML default to the geom mixing rule */
static int add_pair(mjoModel* m, int g1, int g2, int condim, const double* friction5) {
  int p = m->npair++;
  /* collision functions take the lower geom type first (plane before capsule) */
  if (m->geom_type[g1] > m->geom_type[g2]) { int t = g1; g1 = g2; g2 = t; }
  mix_pair_params(m, g1, g2, p, m);
  m->pair_geom1[p] = g1; m->pair_geom2[p] = g2; m->pair_explicit[p] = 1;
  if (condim > 0) m->pair_dim[p] = condim;
  if (friction5) memcpy(m->pair_friction[p], friction5, 5 * sizeof(double));
  return p;
}

static void mjo_kinematics(const mjoModel* m, mjoData* d);
static void mjo_mass_matrix(const mjoModel* m, mjoData* d);

/* capsule volume as computed by the MuJoCo 2.1.0 binary that mujoco-py 2.1 binds: pi*r^2*(L + r)
 * (NOT the exact pi*r^2*L + 4/3*pi*r^3 of later releases).  Evidence: the body masses every
 * mujoco-py-era gym model reports (Hopper [3.53429174, 3.92699082, 2.71433605, 5.0893801],
 * Humanoid thigh 4.52555626, shin 2.63249442) equal 1000*pi*r^2*(L+r) for the capsules in
 * hopper.xml / humanoid.xml; SURVEY.md Q16.  Recalled public constants, not verifiable here. */
static double capsule_volume_210(double r, double half) { return PI * r * r * (2 * half + r); }

static void geom_mass_inertia(const mjoModel* m, int g, double* mass, double* diag) {
  double r = m->geom_size[g][0], rho = m->geom_density[g];
  if (m->geom_type[g] == MJO_GEOM_SPHERE) {
    *mass = rho * 4.0 * PI * r * r * r / 3.0;
    diag[0] = diag[1] = diag[2] = 2 * (*mass) * r * r / 5;
  } else if (m->geom_type[g] == MJO_GEOM_CAPSULE) {
    double h = 2 * m->geom_size[g][1];
    *mass = rho * capsule_volume_210(r, m->geom_size[g][1]);
    /* The 2.1.0 capsule inertia formula is not published; this restatement splits the (2.1.0)
     * mass between cylinder (pi r^2 h) and end caps (pi r^3, the same cap volume the mass uses)
     * and applies the textbook cylinder + two-hemisphere inertias.  PARITY HAZARD, see DESIGN.md. */
    double ms = (*mass) * r / (h + r), mc = (*mass) - ms;
    diag[0] = diag[1] = mc * (3 * r * r + h * h) / 12 + 2 * ms * r * r / 5 + ms * h * (3 * r + 2 * h) / 8;
    diag[2] = mc * r * r / 2 + 2 * ms * r * r / 5;
  } else { *mass = 0; diag[0] = diag[1] = diag[2] = 0; }
}

/* [3P] MJCF compiler, inertiafromgeom="true": body mass / COM / inertia from its geoms */
static void compile_inertia(mjoModel* m, double settotalmass) {
  for (int b = 1; b < m->nbody; b++) {
    double M = 0, com[3] = {0, 0, 0};
    for (int g = 0; g < m->ngeom; g++) if (m->geom_body[g] == b) {
      double mg, dg[3]; geom_mass_inertia(m, g, &mg, dg);
      M += mg; addscl3(com, m->geom_pos[g], mg);
    }
    if (M > 0) { com[0] /= M; com[1] /= M; com[2] /= M; }
    double I[9] = {0};
    for (int g = 0; g < m->ngeom; g++) if (m->geom_body[g] == b) {
      double mg, dg[3], R[9]; geom_mass_inertia(m, g, &mg, dg);
      q_to_mat(R, m->geom_quat[g]);
      double dd[3]; sub3(dd, m->geom_pos[g], com);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += R[i * 3 + k] * dg[k] * R[j * 3 + k];
        s += mg * ((i == j ? dot3(dd, dd) : 0) - dd[i] * dd[j]);
        I[i * 3 + j] += s;
      }
    }
    m->body_mass[b] = M; copy3(m->body_ipos[b], com); memcpy(m->body_inertia[b], I, sizeof I);
  }
  if (settotalmass > 0) { /* [3P] compiler settotalmass: scale all masses and inertias */
    double tot = 0; for (int b = 1; b < m->nbody; b++) tot += m->body_mass[b];
    double s = settotalmass / tot;
    for (int b = 1; b < m->nbody; b++) { m->body_mass[b] *= s; for (int k = 0; k < 9; k++) m->body_inertia[b][k] *= s; }
  }
}

static void compile_pairs(mjoModel* m) {
  /* weld ids: a body without joints is welded to its parent ([3P] body_weldid) */
  m->body_weldid[0] = 0;
  for (int b = 1; b < m->nbody; b++) m->body_weldid[b] = m->body_jntnum[b] ? b : m->body_weldid[m->body_parent[b]];
  /* [3P] mj_collision walks body pairs sorted by (body1 << 16) + body2 and, inside a pair, geoms of body1 x
   * geoms of body2: the contact (hence PGS row) order follows from that */
  for (int bb1 = 0; bb1 < m->nbody; bb1++) for (int bb2 = bb1 + 1; bb2 < m->nbody; bb2++)
  for (int g1 = 0; g1 < m->ngeom; g1++) for (int g2 = 0; g2 < m->ngeom; g2++) {
    if (m->geom_body[g1] != bb1 || m->geom_body[g2] != bb2) continue;
    int a = g1, b = g2;
    if (m->geom_type[a] > m->geom_type[b]) { int t = a; a = b; b = t; }
    int b1 = m->geom_body[a], b2 = m->geom_body[b];
    int w1 = m->body_weldid[b1], w2 = m->body_weldid[b2];
    int wp1 = m->body_weldid[m->body_parent[w1]], wp2 = m->body_weldid[m->body_parent[w2]];
    /* [3P] mj_collision filters: same weld body; parent-child unless one side is the world;
     * contype/conaffinity bit test */
    if (w1 == w2) continue;
    if (w1 != 0 && w2 != 0 && (w1 == wp2 || w2 == wp1)) continue;
    if (!((m->geom_contype[a] & m->geom_conaffinity[b]) || (m->geom_contype[b] & m->geom_conaffinity[a]))) continue;
    if (m->geom_type[a] == MJO_GEOM_PLANE && m->geom_type[b] == MJO_GEOM_PLANE) continue;
    int dup = 0; /* an explicit <pair> for the same geoms takes precedence */
    for (int p = 0; p < m->npair; p++) if (m->pair_explicit[p] && m->pair_geom1[p] == a && m->pair_geom2[p] == b) dup = 1;
    if (dup) continue;
    int p = m->npair++;
    if (p >= MJO_MAXPAIR) { fprintf(stderr, "mjo: too many pairs\n"); abort(); }
    m->pair_geom1[p] = a; m->pair_geom2[p] = b; m->pair_explicit[p] = 0;
    mix_pair_params(m, a, b, p, m);
  }
}

/* [3P] mj_setConst / set0: quantities evaluated once at qpos0 */
static void compile_setconst(mjoModel* m) {
  mjoData* d = (mjoData*)calloc(1, sizeof(mjoData));
  /* subtree masses */
  for (int b = 0; b < m->nbody; b++) m->body_subtreemass[b] = m->body_mass[b];
  for (int b = m->nbody - 1; b > 0; b--) m->body_subtreemass[m->body_parent[b]] += m->body_subtreemass[b];
  /* qpos0 of free joints from the body pose */
  for (int j = 0; j < m->njnt; j++) if (m->jnt_type[j] == MJO_JNT_FREE) {
    int b = m->jnt_body[j], a = m->jnt_qposadr[j];
    copy3(m->qpos0 + a, m->body_pos[b]); memcpy(m->qpos0 + a + 3, m->body_quat[b], 4 * sizeof(double));
  }
  memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
  mjo_kinematics(m, d);
  mjo_mass_matrix(m, d);
  int nv = m->nv;
  double L[MJO_MAXV * MJO_MAXV], Minv[MJO_MAXV * MJO_MAXV];
  chol_factor(L, d->qM, nv, MJO_MAXV);
  for (int i = 0; i < nv; i++) {
    double e[MJO_MAXV] = {0}; e[i] = 1; chol_solve(L, e, nv, MJO_MAXV);
    for (int k = 0; k < nv; k++) Minv[k * MJO_MAXV + i] = e[k];
  }
  double tr = 0; for (int i = 0; i < nv; i++) tr += d->qM[i * MJO_MAXV + i];
  m->meaninertia = nv ? tr / nv : 1;
  /* dof_invweight0: diagonal of M^-1; free-joint translation / rotation triplets are averaged */
  for (int i = 0; i < nv; i++) m->dof_invweight0[i] = Minv[i * MJO_MAXV + i];
  for (int j = 0; j < m->njnt; j++) if (m->jnt_type[j] == MJO_JNT_FREE) {
    int a = m->jnt_dofadr[j];
    for (int g = 0; g < 2; g++) {
      double s = (m->dof_invweight0[a + 3 * g] + m->dof_invweight0[a + 3 * g + 1] + m->dof_invweight0[a + 3 * g + 2]) / 3;
      for (int k = 0; k < 3; k++) m->dof_invweight0[a + 3 * g + k] = s;
    }
  }
  /* body_invweight0: mean diagonal of J M^-1 J^T for the translational / rotational body-COM Jacobian */
  for (int b = 1; b < m->nbody; b++) {
    double J[6][MJO_MAXV]; memset(J, 0, sizeof J);
    int has = 0;
    for (int c = b; c > 0; c = m->body_parent[c])
      for (int k = 0; k < m->body_dofnum[c]; k++) {
        int dd = m->body_dofadr[c] + k; has = 1;
        if (d->dof_kind[dd] == 0) { for (int x = 0; x < 3; x++) J[x][dd] = d->dof_axis[dd][x]; }
        else {
          double r[3], v[3]; sub3(r, d->xipos[b], d->dof_anchor[dd]); cross3(v, d->dof_axis[dd], r);
          for (int x = 0; x < 3; x++) { J[x][dd] = v[x]; J[3 + x][dd] = d->dof_axis[dd][x]; }
        }
      }
    if (!has) { m->body_invweight0[b][0] = m->body_invweight0[b][1] = 0; continue; }
    double A[6];
    for (int r = 0; r < 6; r++) {
      double s = 0;
      for (int i = 0; i < nv; i++) for (int k = 0; k < nv; k++) s += J[r][i] * Minv[i * MJO_MAXV + k] * J[r][k];
      A[r] = s;
    }
    m->body_invweight0[b][0] = fmax(MINVAL, (A[0] + A[1] + A[2]) / 3);
    m->body_invweight0[b][1] = fmax(MINVAL, (A[3] + A[4] + A[5]) / 3);
  }
  free(d);
}

static void compile(mjoModel* m, double settotalmass) {
  compile_inertia(m, settotalmass);
  compile_pairs(m);
  compile_setconst(m);
}

/* ------------------------------------------------------------------ kinematics ---------- */
/* [3P] mj_kinematics: body / inertial / geom frames and joint axes+anchors in the world frame */
static void mjo_kinematics(const mjoModel* m, mjoData* d) {
  d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0; zero3(d->xpos[0]);
  q_to_mat(d->xmat[0], d->xquat[0]); zero3(d->xipos[0]);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parent[b];
    double xpos[3], xquat[4], t[3];
    mat_vec(t, d->xmat[p], m->body_pos[b]); add3(xpos, d->xpos[p], t);
    q_mul(xquat, d->xquat[p], m->body_quat[b]);
    for (int jj = 0; jj < m->body_jntnum[b]; jj++) {
      int j = m->body_jntadr[b] + jj, qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == MJO_JNT_FREE) {
        copy3(xpos, d->qpos + qa); memcpy(xquat, d->qpos + qa + 3, 4 * sizeof(double)); q_normalize(xquat);
        double R[9]; q_to_mat(R, xquat);
        for (int k = 0; k < 3; k++) { /* translations along world axes, rotations about body axes */
          d->dof_kind[da + k] = 0; zero3(d->dof_axis[da + k]); d->dof_axis[da + k][k] = 1; copy3(d->dof_anchor[da + k], xpos);
          d->dof_kind[da + 3 + k] = 1; d->dof_axis[da + 3 + k][0] = R[k]; d->dof_axis[da + 3 + k][1] = R[3 + k];
          d->dof_axis[da + 3 + k][2] = R[6 + k]; copy3(d->dof_anchor[da + 3 + k], xpos);
        }
      } else {
        double R[9], anchor[3], axis[3]; q_to_mat(R, xquat);
        mat_vec(t, R, m->jnt_pos[j]); add3(anchor, xpos, t);
        mat_vec(axis, R, m->jnt_axis[j]);
        double q = d->qpos[qa] - m->qpos0[qa];
        if (m->jnt_type[j] == MJO_JNT_SLIDE) {
          addscl3(xpos, axis, q); d->dof_kind[da] = 0;
        } else {
          double ql[4], nq[4]; q_axisangle(ql, m->jnt_axis[j], q); q_mul(nq, xquat, ql); memcpy(xquat, nq, sizeof nq);
          q_normalize(xquat);
          q_to_mat(R, xquat); mat_vec(t, R, m->jnt_pos[j]); sub3(xpos, anchor, t); /* off-centre rotation */
          d->dof_kind[da] = 1;
        }
        copy3(d->dof_axis[da], axis); copy3(d->dof_anchor[da], anchor);
      }
    }
    copy3(d->xpos[b], xpos); memcpy(d->xquat[b], xquat, sizeof xquat); q_to_mat(d->xmat[b], xquat);
    mat_vec(t, d->xmat[b], m->body_ipos[b]); add3(d->xipos[b], xpos, t);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_body[g]; double t[3], q[4];
    mat_vec(t, d->xmat[b], m->geom_pos[g]); add3(d->geom_xpos[g], d->xpos[b], t);
    q_mul(q, d->xquat[b], m->geom_quat[g]); q_to_mat(d->geom_xmat[g], q);
  }
}

/* point Jacobian of a body-fixed point `p` (world coords) of body b: jacp[3][nv], jacr[3][nv] */
static void jac_point(const mjoModel* m, const mjoData* d, int b, const double* p, double jacp[3][MJO_MAXV],
                      double jacr[3][MJO_MAXV]) {
  for (int x = 0; x < 3; x++) for (int i = 0; i < m->nv; i++) { jacp[x][i] = 0; if (jacr) jacr[x][i] = 0; }
  for (int c = b; c > 0; c = m->body_parent[c])
    for (int k = 0; k < m->body_dofnum[c]; k++) {
      int dd = m->body_dofadr[c] + k;
      if (d->dof_kind[dd] == 0) { for (int x = 0; x < 3; x++) jacp[x][dd] = d->dof_axis[dd][x]; }
      else {
        double r[3], v[3]; sub3(r, p, d->dof_anchor[dd]); cross3(v, d->dof_axis[dd], r);
        for (int x = 0; x < 3; x++) { jacp[x][dd] = v[x]; if (jacr) jacr[x][dd] = d->dof_axis[dd][x]; }
      }
    }
}

/* [3P] mj_crb result (joint-space inertia M incl. armature), computed as sum_b J_b^T I_b J_b */
static void mjo_mass_matrix(const mjoModel* m, mjoData* d) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) for (int k = 0; k < nv; k++) d->qM[i * MJO_MAXV + k] = 0;
  for (int b = 1; b < m->nbody; b++) {
    double jp[3][MJO_MAXV], jr[3][MJO_MAXV], Iw[9], RI[9];
    jac_point(m, d, b, d->xipos[b], jp, jr);
    const double* R = d->xmat[b];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += R[i * 3 + k] * m->body_inertia[b][k * 3 + j]; RI[i * 3 + j] = s; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += RI[i * 3 + k] * R[j * 3 + k]; Iw[i * 3 + j] = s; }
    for (int i = 0; i < nv; i++) for (int k = 0; k < nv; k++) {
      double s = 0;
      for (int x = 0; x < 3; x++) s += m->body_mass[b] * jp[x][i] * jp[x][k];
      for (int x = 0; x < 3; x++) for (int y = 0; y < 3; y++) s += jr[x][i] * Iw[x * 3 + y] * jr[y][k];
      d->qM[i * MJO_MAXV + k] += s;
    }
  }
  for (int i = 0; i < nv; i++) d->qM[i * MJO_MAXV + i] += m->dof_armature[i];
}

/* [3P] mj_comVel + mj_rne(flg_acc=0): bias forces c(q,v) incl. gravity, via world-frame Newton-Euler.
 * Spatial vectors are (angular; linear-at-world-origin). */
static void mjo_bias(const mjoModel* m, mjoData* d) {
  double S[MJO_MAXV][6], acc[MJO_MAXBODY][6], frc[MJO_MAXBODY][6];
  int nv = m->nv;
  for (int i = 0; i < nv; i++) {
    if (d->dof_kind[i] == 0) { zero3(S[i]); copy3(S[i] + 3, d->dof_axis[i]); }
    else { copy3(S[i], d->dof_axis[i]); cross3(S[i] + 3, d->dof_anchor[i], d->dof_axis[i]); }
  }
  memset(d->bvel[0], 0, sizeof d->bvel[0]);
  zero3(acc[0]); acc[0][3] = -m->gravity[0]; acc[0][4] = -m->gravity[1]; acc[0][5] = -m->gravity[2];
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parent[b];
    double v[6], a[6]; memcpy(v, d->bvel[p], sizeof v); memcpy(a, acc[p], sizeof a);
    for (int jj = 0; jj < m->body_jntnum[b]; jj++) {
      int j = m->body_jntadr[b] + jj, da = m->jnt_dofadr[j];
      int nd = (m->jnt_type[j] == MJO_JNT_FREE) ? 6 : 1;
      for (int k = 0; k < nd; k++) {
        int i = da + k;
        /* Sdot = v x S (motion cross product) with v the velocity of the frame S is fixed in.
         * free joint: translations are world-fixed (Sdot = 0); the three rotations all use the
         * velocity after the translations ([3P] mj_comVel, case mjJNT_FREE/BALL). */
        double vv[6];
        if (nd == 6) { if (k < 3) memset(vv, 0, sizeof vv); else { memcpy(vv, d->bvel[p], sizeof vv); for (int t = 0; t < 3; t++) for (int x = 0; x < 6; x++) vv[x] += S[da + t][x] * d->qvel[da + t]; } }
        else memcpy(vv, v, sizeof vv);
        double sd[6], t1[3], t2[3];
        cross3(sd, vv, S[i]);                       /* w x s_w */
        cross3(t1, vv, S[i] + 3); cross3(t2, vv + 3, S[i]); add3(sd + 3, t1, t2); /* w x s_v + v x s_w */
        for (int x = 0; x < 6; x++) a[x] += sd[x] * d->qvel[i];
        for (int x = 0; x < 6; x++) v[x] += S[i][x] * d->qvel[i];
      }
    }
    memcpy(d->bvel[b], v, sizeof v); memcpy(acc[b], a, sizeof a);
    /* force on body b: F = m a_c, N = I alpha + w x I w, expressed as spatial force about the origin */
    const double* w = v; const double* vo = v + 3; const double* al = a; const double* ao = a + 3;
    const double* c = d->xipos[b];
    double ac[3], t[3], t3[3];
    cross3(t, w, vo); add3(ac, ao, t);            /* classical accel of the point at the origin */
    cross3(t, al, c); add3(ac, ac, t);
    cross3(t, w, c); cross3(t3, w, t); add3(ac, ac, t3);
    double F[3] = {m->body_mass[b] * ac[0], m->body_mass[b] * ac[1], m->body_mass[b] * ac[2]};
    const double* R = d->xmat[b]; double Iw[9], RI[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += R[i * 3 + k] * m->body_inertia[b][k * 3 + j]; RI[i * 3 + j] = s; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += RI[i * 3 + k] * R[j * 3 + k]; Iw[i * 3 + j] = s; }
    double Ia[3], Iww[3], N[3];
    mat_vec(Ia, Iw, al); mat_vec(Iww, Iw, w); cross3(t, w, Iww); add3(N, Ia, t);
    cross3(t, c, F); add3(frc[b], N, t); copy3(frc[b] + 3, F);
  }
  for (int i = 0; i < nv; i++) d->qfrc_bias[i] = 0;
  memset(frc[0], 0, sizeof frc[0]);
  for (int b = m->nbody - 1; b > 0; b--) {
    for (int k = 0; k < m->body_dofnum[b]; k++) {
      int i = m->body_dofadr[b] + k; double s = 0;
      for (int x = 0; x < 6; x++) s += S[i][x] * frc[b][x];
      d->qfrc_bias[i] = s;
    }
    int p = m->body_parent[b];
    for (int x = 0; x < 6; x++) frc[p][x] += frc[b][x];
  }
}

/* ------------------------------------------------------------------ collision ----------- */
/* [3P] mju_makeFrame: complete a contact frame from its normal (row 0) and an optional row 1 */
static void make_frame(double* f) {
  normalize3(f);
  if (norm3(f + 3) < 0.5) { zero3(f + 3); if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1; }
  double s = dot3(f, f + 3); addscl3(f + 3, f, -s); normalize3(f + 3);
  cross3(f + 6, f, f + 3);
}
static int plane_sphere(const double* ppos, const double* pn, const double* c, double r, double margin, mjoContact* con) {
  double t[3]; sub3(t, c, ppos);
  double dist = dot3(t, pn) - r;
  if (dist > margin) return 0;
  con->dist = dist;
  copy3(con->pos, c); addscl3(con->pos, pn, -(r + dist / 2));
  memset(con->frame, 0, sizeof con->frame); copy3(con->frame, pn);
  return 1;
}
static int sphere_sphere(const double* c1, double r1, const double* c2, double r2, double margin, mjoContact* con) {
  double dif[3]; sub3(dif, c2, c1);
  double len = norm3(dif), dist = len - r1 - r2;
  if (dist > margin) return 0;
  double n[3] = {1, 0, 0};
  if (len >= MINVAL) { n[0] = dif[0] / len; n[1] = dif[1] / len; n[2] = dif[2] / len; }
  con->dist = dist; copy3(con->pos, c1); addscl3(con->pos, n, r1 + dist / 2);
  memset(con->frame, 0, sizeof con->frame); copy3(con->frame, n);
  return 1;
}
/* [3P] mjc_PlaneCapsule: the two end spheres, tangent frame aligned with the capsule axis */
static int plane_capsule(const mjoModel* m, const mjoData* d, int g1, int g2, double margin, mjoContact* con) {
  const double* pm = d->geom_xmat[g1]; double pn[3] = {pm[2], pm[5], pm[8]};
  const double* cm = d->geom_xmat[g2]; double ax[3] = {cm[2], cm[5], cm[8]};
  double seg[3] = {ax[0] * m->geom_size[g2][1], ax[1] * m->geom_size[g2][1], ax[2] * m->geom_size[g2][1]};
  int n = 0; double c[3];
  add3(c, d->geom_xpos[g2], seg);
  if (plane_sphere(d->geom_xpos[g1], pn, c, m->geom_size[g2][0], margin, con + n)) { copy3(con[n].frame + 3, ax); n++; }
  sub3(c, d->geom_xpos[g2], seg);
  if (plane_sphere(d->geom_xpos[g1], pn, c, m->geom_size[g2][0], margin, con + n)) { copy3(con[n].frame + 3, ax); n++; }
  return n;
}
/* [3P] mjc_CapsuleCapsule: closest points of the two segments, then sphere-sphere */
static int capsule_capsule(const mjoModel* m, const mjoData* d, int g1, int g2, double margin, mjoContact* con) {
  const double* m1 = d->geom_xmat[g1]; const double* m2 = d->geom_xmat[g2];
  double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
  const double* p1 = d->geom_xpos[g1]; const double* p2 = d->geom_xpos[g2];
  double l1 = m->geom_size[g1][1], l2 = m->geom_size[g2][1], r1 = m->geom_size[g1][0], r2 = m->geom_size[g2][0];
  double dif[3]; sub3(dif, p1, p2);
  double ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2);
  double u = -dot3(a1, dif), v = dot3(a2, dif), det = ma * mc - mb * mb;
  if (fabs(det) >= MINVAL) {
    double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > l1) { x1 = l1; x2 = (v - mb * l1) / mc; } else if (x1 < -l1) { x1 = -l1; x2 = (v + mb * l1) / mc; }
    if (x2 > l2) { x2 = l2; x1 = (u - mb * l2) / ma; } else if (x2 < -l2) { x2 = -l2; x1 = (u + mb * l2) / ma; }
    if (x1 > l1) x1 = l1; else if (x1 < -l1) x1 = -l1;
    double c1[3], c2[3]; copy3(c1, p1); addscl3(c1, a1, x1); copy3(c2, p2); addscl3(c2, a2, x2);
    return sphere_sphere(c1, r1, c2, r2, margin, con);
  }
  /* parallel axes: test the end points of segment 1 against segment 2 and vice versa (<= 2 contacts) */
  int n = 0;
  for (int s = -1; s <= 1 && n < 2; s += 2) {
    double c1[3]; copy3(c1, p1); addscl3(c1, a1, s * l1);
    double t[3]; sub3(t, c1, p2); double x2 = dot3(t, a2);
    if (x2 >= -l2 && x2 <= l2) { double c2[3]; copy3(c2, p2); addscl3(c2, a2, x2); n += sphere_sphere(c1, r1, c2, r2, margin, con + n); }
  }
  for (int s = -1; s <= 1 && n < 2; s += 2) {
    double c2[3]; copy3(c2, p2); addscl3(c2, a2, s * l2);
    double t[3]; sub3(t, c2, p1); double x1 = dot3(t, a1);
    if (x1 >= -l1 && x1 <= l1) { double c1[3]; copy3(c1, p1); addscl3(c1, a1, x1); n += sphere_sphere(c1, r1, c2, r2, margin, con + n); }
  }
  return n;
}
static int sphere_capsule(const mjoModel* m, const mjoData* d, int g1, int g2, double margin, mjoContact* con) {
  const double* m2 = d->geom_xmat[g2]; double ax[3] = {m2[2], m2[5], m2[8]};
  double t[3]; sub3(t, d->geom_xpos[g1], d->geom_xpos[g2]);
  double x = dot3(t, ax), l = m->geom_size[g2][1];
  if (x > l) x = l; else if (x < -l) x = -l;
  double c2[3]; copy3(c2, d->geom_xpos[g2]); addscl3(c2, ax, x);
  return sphere_sphere(d->geom_xpos[g1], m->geom_size[g1][0], c2, m->geom_size[g2][0], margin, con);
}

/* [3P] mj_collision narrow phase over the compiled pair list */
static void mjo_collision(const mjoModel* m, mjoData* d) {
  d->ncon = 0;
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    mjoContact con[2]; int n = 0; double margin = m->pair_margin[p];
    if (t1 == MJO_GEOM_PLANE && t2 == MJO_GEOM_CAPSULE) n = plane_capsule(m, d, g1, g2, margin, con);
    else if (t1 == MJO_GEOM_PLANE && t2 == MJO_GEOM_SPHERE) {
      const double* pm = d->geom_xmat[g1]; double pn[3] = {pm[2], pm[5], pm[8]};
      n = plane_sphere(d->geom_xpos[g1], pn, d->geom_xpos[g2], m->geom_size[g2][0], margin, con);
    } else if (t1 == MJO_GEOM_SPHERE && t2 == MJO_GEOM_SPHERE)
      n = sphere_sphere(d->geom_xpos[g1], m->geom_size[g1][0], d->geom_xpos[g2], m->geom_size[g2][0], margin, con);
    else if (t1 == MJO_GEOM_SPHERE && t2 == MJO_GEOM_CAPSULE) n = sphere_capsule(m, d, g1, g2, margin, con);
    else if (t1 == MJO_GEOM_CAPSULE && t2 == MJO_GEOM_CAPSULE) n = capsule_capsule(m, d, g1, g2, margin, con);
    for (int k = 0; k < n; k++) {
      if (d->ncon >= MJO_MAXCON) { fprintf(stderr, "mjo: contact overflow\n"); abort(); }
      mjoContact* c = d->contact + d->ncon++;
      *c = con[k]; make_frame(c->frame);
      c->geom1 = g1; c->geom2 = g2; c->dim = m->pair_dim[p];
      c->includemargin = m->pair_margin[p] - m->pair_gap[p];
      memcpy(c->friction, m->pair_friction[p], sizeof c->friction);
      memcpy(c->solref, m->pair_solref[p], sizeof c->solref);
      memcpy(c->solimp, m->pair_solimp[p], sizeof c->solimp);
      c->efc_address = -1;
    }
  }
}

/* ------------------------------------------------------------------ constraints --------- */
static double* efc_row(mjoData* d, int i) { return d->efc_J + (size_t)i * MJO_MAXV; }

/* [3P] mj_makeConstraint: joint-limit rows, then contact rows (pyramidal cones) */
static void mjo_make_constraint(const mjoModel* m, mjoData* d, double solref[][2], double solimp[][5]) {
  int nv = m->nv, ne = 0;
  /* [3P] mj_instantiateLimit, hinge / slide */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j] || m->jnt_type[j] == MJO_JNT_FREE) continue;
    double value = d->qpos[m->jnt_qposadr[j]];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[j][(side + 1) / 2] - value);
      if (dist < m->jnt_margin[j]) {
        double* J = efc_row(d, ne); for (int i = 0; i < nv; i++) J[i] = 0;
        J[m->jnt_dofadr[j]] = -side;
        d->efc_pos[ne] = dist; d->efc_margin[ne] = m->jnt_margin[j]; d->efc_type[ne] = 0; d->efc_id[ne] = j;
        memcpy(solref[ne], m->jnt_solref[j], 2 * sizeof(double)); memcpy(solimp[ne], m->jnt_solimp[j], 5 * sizeof(double));
        ne++;
      }
    }
  }
  d->nlimit = ne;
  /* [3P] mj_instantiateContact, pyramidal */
  for (int c = 0; c < d->ncon; c++) {
    mjoContact* con = d->contact + c;
    if (con->dist >= con->includemargin) continue; /* within margin but not "included" (gap) */
    int b1 = m->geom_body[con->geom1], b2 = m->geom_body[con->geom2];
    double j1[3][MJO_MAXV], j2[3][MJO_MAXV], jc[3][MJO_MAXV];
    jac_point(m, d, b1, con->pos, j1, NULL); jac_point(m, d, b2, con->pos, j2, NULL);
    for (int r = 0; r < 3; r++) for (int i = 0; i < nv; i++) {
      double s = 0; for (int x = 0; x < 3; x++) s += con->frame[r * 3 + x] * (j2[x][i] - j1[x][i]);
      jc[r][i] = s;
    }
    con->efc_address = ne;
    if (ne + 4 >= MJO_MAXEFC) { fprintf(stderr, "mjo: efc overflow\n"); abort(); }
    if (con->dim == 1) {
      double* J = efc_row(d, ne); for (int i = 0; i < nv; i++) J[i] = jc[0][i];
      d->efc_pos[ne] = con->dist; d->efc_margin[ne] = con->includemargin; d->efc_type[ne] = 1; d->efc_id[ne] = c;
      memcpy(solref[ne], con->solref, 2 * sizeof(double)); memcpy(solimp[ne], con->solimp, 5 * sizeof(double));
      ne++;
    } else { /* dim 3: edges n +/- mu_k t_k */
      for (int k = 1; k < con->dim; k++) for (int s = 1; s >= -1; s -= 2) {
        double* J = efc_row(d, ne);
        for (int i = 0; i < nv; i++) J[i] = jc[0][i] + s * con->friction[k - 1] * jc[k][i];
        d->efc_pos[ne] = con->dist; d->efc_margin[ne] = con->includemargin; d->efc_type[ne] = 2; d->efc_id[ne] = c;
        memcpy(solref[ne], con->solref, 2 * sizeof(double)); memcpy(solimp[ne], con->solimp, 5 * sizeof(double));
        ne++;
      }
    }
  }
  d->nefc = ne;
}

/* [3P] mj_diagApprox */
static void mjo_diag_approx(const mjoModel* m, mjoData* d) {
  for (int i = 0; i < d->nefc; i++) {
    if (d->efc_type[i] == 0) { d->efc_diagApprox[i] = m->dof_invweight0[m->jnt_dofadr[d->efc_id[i]]]; continue; }
    const mjoContact* con = d->contact + d->efc_id[i];
    int b1 = m->geom_body[con->geom1], b2 = m->geom_body[con->geom2];
    double tran = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
    double rot = m->body_invweight0[b1][1] + m->body_invweight0[b2][1];
    if (d->efc_type[i] == 1) { d->efc_diagApprox[i] = tran; continue; }
    int k = (i - con->efc_address) / 2; /* friction dimension of this pyramid edge */
    double fri = con->friction[k];
    d->efc_diagApprox[i] = tran + fri * fri * (k < 2 ? tran : rot);
  }
}

/* [3P] getimpedance: sigmoid impedance d(r) of the solimp parameters */
static double impedance(const double* solimp, double pos, double margin) {
  double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  if (dmin == dmax || width <= MINVAL) return 0.5 * (dmin + dmax);
  double x = fabs((pos - margin) / width);
  if (x >= 1) return dmax;
  if (x == 0) return dmin;
  double y;
  if (power == 1) y = x;
  else if (x <= mid) { double a = 1 / pow(mid, power - 1); y = a * pow(x, power); }
  else { double b = 1 / pow(1 - mid, power - 1); y = 1 - b * pow(1 - x, power); }
  return dmin + y * (dmax - dmin);
}

/* [3P] mj_makeImpedance + mj_referenceConstraint: KBIP, R, D, aref */
static void mjo_make_impedance(const mjoModel* m, mjoData* d, double solref[][2], double solimp[][5]) {
  int nv = m->nv;
  for (int i = 0; i < d->nefc; i++) {
    double sr[2] = {solref[i][0], solref[i][1]}, si[5]; memcpy(si, solimp[i], sizeof si);
    /* [3P] getsolparam: refsafe and solimp clamps */
    if (sr[0] > 0) sr[0] = fmax(sr[0], 2 * m->timestep);
    si[0] = fmin(fmax(si[0], MINIMP), MAXIMP); si[1] = fmin(fmax(si[1], MINIMP), MAXIMP);
    si[2] = fmax(0, si[2]); si[3] = fmin(fmax(si[3], MINIMP), MAXIMP); si[4] = fmax(1, si[4]);
    double imp = impedance(si, d->efc_pos[i], d->efc_margin[i]);
    double K, B;
    if (sr[0] > 0) { K = 1 / fmax(MINVAL, si[1] * si[1] * sr[0] * sr[0] * sr[1] * sr[1]); B = 2 / fmax(MINVAL, si[1] * sr[0]); }
    else { K = -sr[0] / fmax(MINVAL, si[1] * si[1]); B = -sr[1] / fmax(MINVAL, si[1]); }
    d->efc_KBIP[i][0] = K; d->efc_KBIP[i][1] = B; d->efc_KBIP[i][2] = imp; d->efc_KBIP[i][3] = 0;
    d->efc_R[i] = fmax(MINVAL, (1 - imp) * d->efc_diagApprox[i] / imp);
  }
  /* pyramidal contacts: all edges share Rpy = 2 mu^2 R(first edge), mu = friction[0]/sqrt(impratio) */
  for (int c = 0; c < d->ncon; c++) {
    const mjoContact* con = d->contact + c;
    if (con->dim > 1 && con->efc_address >= 0) {
      double mu = con->friction[0] * sqrt(1 / m->impratio);
      double Rpy = 2 * mu * mu * d->efc_R[con->efc_address];
      for (int k = 0; k < 2 * (con->dim - 1); k++) d->efc_R[con->efc_address + k] = Rpy;
    }
  }
  for (int i = 0; i < d->nefc; i++) {
    d->efc_D[i] = 1 / d->efc_R[i];
    double vel = 0; const double* J = efc_row(d, i);
    for (int k = 0; k < nv; k++) vel += J[k] * d->qvel[k];
    d->efc_vel[i] = vel;
    d->efc_aref[i] = -d->efc_KBIP[i][1] * vel - d->efc_KBIP[i][0] * d->efc_KBIP[i][2] * (d->efc_pos[i] - d->efc_margin[i]);
  }
}

/* ------------------------------------------------------------------ solvers ------------- */
/* cost and derivatives along qacc + alpha*search for the primal problem ([3P] engine_solver Newton).
 * All rows here are inequality rows (limits, contacts): s(jar) = 0.5*D*min(0,jar)^2. */
static void line_eval(const mjoData* d, int nefc, const double* jar, const double* jv, double qg0, double qg1, double qg2,
                      double a, double* c, double* d1, double* d2) {
  double cost = a * a * qg2 + a * qg1 + qg0, g = 2 * a * qg2 + qg1, h = 2 * qg2;
  for (int i = 0; i < nefc; i++) {
    double x = jar[i] + a * jv[i];
    if (x < 0) { cost += 0.5 * d->efc_D[i] * x * x; g += d->efc_D[i] * x * jv[i]; h += d->efc_D[i] * jv[i] * jv[i]; }
  }
  *c = cost; *d1 = g; *d2 = h;
}

static void mjo_solve_newton(const mjoModel* m, mjoData* d) {
  int nv = m->nv, ne = d->nefc;
  double Ma[MJO_MAXV], grad[MJO_MAXV], search[MJO_MAXV], Mv[MJO_MAXV];
  double* jar = (double*)malloc(sizeof(double) * (2 * ne + 1));
  double* jv = jar + ne;
  double H[MJO_MAXV * MJO_MAXV], L[MJO_MAXV * MJO_MAXV];
  memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv); /* warmstart disabled in all four XMLs */
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  /* m->tolerance is the model's 1e-8 by default ([3P] opt.tolerance); tests tighten it so that the
   * oracle returns the minimiser itself. */
  double tol = m->tolerance;
  d->solver_iter = 0;
  for (int iter = 0; iter < 200; iter++) {
    for (int i = 0; i < nv; i++) { double s = 0; for (int k = 0; k < nv; k++) s += d->qM[i * MJO_MAXV + k] * d->qacc[k]; Ma[i] = s; }
    for (int i = 0; i < ne; i++) { const double* J = efc_row(d, i); double s = -d->efc_aref[i]; for (int k = 0; k < nv; k++) s += J[k] * d->qacc[k]; jar[i] = s; }
    for (int i = 0; i < nv; i++) grad[i] = Ma[i] - d->qfrc_smooth[i];
    for (int i = 0; i < nv; i++) for (int k = 0; k < nv; k++) H[i * MJO_MAXV + k] = d->qM[i * MJO_MAXV + k];
    for (int i = 0; i < ne; i++) {
      double f = jar[i] < 0 ? -d->efc_D[i] * jar[i] : 0; d->efc_force[i] = f;
      if (jar[i] < 0) {
        const double* J = efc_row(d, i);
        for (int k = 0; k < nv; k++) grad[k] -= J[k] * f;
        for (int a = 0; a < nv; a++) if (J[a] != 0) for (int b = 0; b < nv; b++) H[a * MJO_MAXV + b] += d->efc_D[i] * J[a] * J[b];
      }
    }
    double gn = 0; for (int i = 0; i < nv; i++) gn += grad[i] * grad[i];
    gn = sqrt(gn);
    if (gn * scale < tol) break;
    chol_factor(L, H, nv, MJO_MAXV);
    for (int i = 0; i < nv; i++) search[i] = -grad[i];
    chol_solve(L, search, nv, MJO_MAXV);
    for (int i = 0; i < nv; i++) { double s = 0; for (int k = 0; k < nv; k++) s += d->qM[i * MJO_MAXV + k] * search[k]; Mv[i] = s; }
    for (int i = 0; i < ne; i++) { const double* J = efc_row(d, i); double s = 0; for (int k = 0; k < nv; k++) s += J[k] * search[k]; jv[i] = s; }
    /* Gauss term along the line: 0.5*(a - a0)^T M (a - a0) */
    double qg0 = 0, qg1 = 0, qg2 = 0;
    for (int i = 0; i < nv; i++) {
      qg0 += 0.5 * (Ma[i] - d->qfrc_smooth[i]) * (d->qacc[i] - d->qacc_smooth[i]);
      qg1 += search[i] * (Ma[i] - d->qfrc_smooth[i]);
      qg2 += 0.5 * search[i] * Mv[i];
    }
    /* exact 1-D minimisation: phi' is piecewise linear and increasing; safeguarded Newton */
    double a = 0, c, g, h, lo = 0, hi = -1;
    line_eval(d, ne, jar, jv, qg0, qg1, qg2, 0, &c, &g, &h);
    for (int ls = 0; ls < 100; ls++) {
      if (g < 0) lo = a; else hi = a;
      double an = a - g / h;
      if (hi >= 0 && (an <= lo || an >= hi)) an = 0.5 * (lo + hi);
      if (an < lo) an = lo;
      double prev = a; a = an;
      line_eval(d, ne, jar, jv, qg0, qg1, qg2, a, &c, &g, &h);
      if (fabs(g) <= 1e-15 * (1 + fabs(qg1)) || fabs(a - prev) <= 1e-16 * (1 + fabs(a))) break;
    }
    double amax = 0, smax = 0;
    for (int i = 0; i < nv; i++) { d->qacc[i] += a * search[i]; amax = fmax(amax, fabs(d->qacc[i])); smax = fmax(smax, fabs(a * search[i])); }
    d->solver_iter++;
    if (smax <= 1e-15 * (1 + amax)) break; /* stagnation at rounding level */
  }
  for (int i = 0; i < ne; i++) { const double* J = efc_row(d, i); double s = -d->efc_aref[i]; for (int k = 0; k < nv; k++) s += J[k] * d->qacc[k]; d->efc_force[i] = s < 0 ? -d->efc_D[i] * s : 0; }
  free(jar);
}

/* [3P] PGS solver (dual): A = J M^-1 J^T + R, Gauss-Seidel sweeps in row order with per-row
 * projection f_i >= 0; pyramidal contact edges are ordinary non-negative rows. Capped at
 * m->iterations sweeps with the model tolerance (humanoid.xml:9: PGS, 50). */
static void mjo_solve_pgs(const mjoModel* m, mjoData* d) {
  int nv = m->nv, ne = d->nefc;
  double L[MJO_MAXV * MJO_MAXV];
  chol_factor(L, d->qM, nv, MJO_MAXV);
  double* A = (double*)malloc(sizeof(double) * ((size_t)ne * ne + (size_t)ne * nv + 2 * ne));
  double* MiJt = A + (size_t)ne * ne; double* b = MiJt + (size_t)ne * nv; double* f = b + ne;
  for (int i = 0; i < ne; i++) {
    double* x = MiJt + (size_t)i * nv; const double* J = efc_row(d, i);
    double tmp[MJO_MAXV]; memcpy(tmp, J, sizeof(double) * nv); chol_solve(L, tmp, nv, MJO_MAXV); memcpy(x, tmp, sizeof(double) * nv);
  }
  for (int i = 0; i < ne; i++) {
    const double* J = efc_row(d, i);
    for (int k = 0; k < ne; k++) { const double* x = MiJt + (size_t)k * nv; double s = 0; for (int t = 0; t < nv; t++) s += J[t] * x[t]; A[(size_t)i * ne + k] = s; }
    A[(size_t)i * ne + i] += d->efc_R[i];
    double s = -d->efc_aref[i]; for (int t = 0; t < nv; t++) s += J[t] * d->qacc_smooth[t];
    b[i] = s; f[i] = 0; /* warmstart disabled: start from zero force */
  }
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  d->solver_iter = 0;
  for (int iter = 0; iter < m->iterations; iter++) {
    double improvement = 0;
    for (int i = 0; i < ne; i++) {
      double res = b[i]; for (int k = 0; k < ne; k++) res += A[(size_t)i * ne + k] * f[k];
      double old = f[i], nf = old - res / A[(size_t)i * ne + i];
      if (nf < 0) nf = 0;
      double df = nf - old; f[i] = nf;
      improvement -= 0.5 * df * df * A[(size_t)i * ne + i] + df * res; /* decrease of the dual cost */
    }
    d->solver_iter++;
    if (improvement * scale < m->tolerance) break;
  }
  for (int t = 0; t < nv; t++) d->qacc[t] = d->qacc_smooth[t];
  for (int i = 0; i < ne; i++) { d->efc_force[i] = f[i]; const double* x = MiJt + (size_t)i * nv; for (int t = 0; t < nv; t++) d->qacc[t] += x[t] * f[i]; }
  free(A);
}

/* ------------------------------------------------------------------ forward + step ------ */
void mjo_forward(const mjoModel* m, mjoData* d) {
  int nv = m->nv;
  /* [3P] mj_fwdPosition */
  mjo_kinematics(m, d);
  mjo_mass_matrix(m, d);
  mjo_collision(m, d);
  static __thread double solref[MJO_MAXEFC][2], solimp[MJO_MAXEFC][5];
  mjo_make_constraint(m, d, solref, solimp);
  /* [3P] mj_fwdVelocity: passive forces, bias */
  for (int i = 0; i < nv; i++) d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i];
  for (int j = 0; j < m->njnt; j++) if (m->jnt_type[j] != MJO_JNT_FREE && m->jnt_stiffness[j] != 0) {
    int qa = m->jnt_qposadr[j];
    d->qfrc_passive[m->jnt_dofadr[j]] -= m->jnt_stiffness[j] * (d->qpos[qa] - m->jnt_springref[j]);
  }
  mjo_bias(m, d);
  mjo_diag_approx(m, d);
  mjo_make_impedance(m, d, solref, solimp);
  /* [3P] mj_fwdActuation: motors, ctrl clamped to ctrlrange, moment = gear */
  for (int i = 0; i < nv; i++) d->qfrc_actuator[i] = 0;
  for (int u = 0; u < m->nu; u++) {
    double c = d->ctrl[u];
    if (m->act_ctrllimited[u]) c = fmin(fmax(c, m->act_ctrlrange[u][0]), m->act_ctrlrange[u][1]);
    d->qfrc_actuator[m->act_dof[u]] += m->act_gear[u] * c;
  }
  /* [3P] mj_fwdAcceleration */
  double L[MJO_MAXV * MJO_MAXV];
  for (int i = 0; i < nv; i++) { d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i]; d->qacc_smooth[i] = d->qfrc_smooth[i]; }
  chol_factor(L, d->qM, nv, MJO_MAXV); chol_solve(L, d->qacc_smooth, nv, MJO_MAXV);
  /* [3P] mj_fwdConstraint */
  if (d->nefc == 0) { memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv); d->solver_iter = 0; }
  else if (m->solver == MJO_SOL_PGS) mjo_solve_pgs(m, d);
  else mjo_solve_newton(m, d);
  for (int i = 0; i < nv; i++) d->qfrc_constraint[i] = 0;
  for (int r = 0; r < d->nefc; r++) { const double* J = efc_row(d, r); for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += J[i] * d->efc_force[r]; }
}

/* [3P] mj_integratePos: qpos += h*qvel with quaternion update for free joints */
static void integrate_pos(const mjoModel* m, double* qpos, const double* qvel, double h) {
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == MJO_JNT_FREE) {
      for (int k = 0; k < 3; k++) qpos[qa + k] += h * qvel[da + k];
      double w[3] = {qvel[da + 3], qvel[da + 4], qvel[da + 5]};
      double n = norm3(w);
      if (n * h > MINVAL) { /* [3P] mju_quatIntegrate: q <- q * axisangle(w/|w|, |w| h) (w in body frame) */
        double ax[3] = {w[0] / n, w[1] / n, w[2] / n}, dq[4], r[4];
        q_axisangle(dq, ax, n * h); q_mul(r, qpos + qa + 3, dq); q_normalize(r); memcpy(qpos + qa + 3, r, sizeof r);
      }
    } else qpos[qa] += h * qvel[da];
  }
}

/* [3P] mj_Euler: semi-implicit Euler with joint damping treated implicitly */
static void mjo_euler(const mjoModel* m, mjoData* d) {
  int nv = m->nv; double h = m->timestep, qacc[MJO_MAXV];
  int damp = 0; for (int i = 0; i < nv; i++) if (m->dof_damping[i] > 0) damp = 1;
  if (damp) {
    double A[MJO_MAXV * MJO_MAXV], L[MJO_MAXV * MJO_MAXV];
    for (int i = 0; i < nv; i++) for (int k = 0; k < nv; k++) A[i * MJO_MAXV + k] = d->qM[i * MJO_MAXV + k];
    for (int i = 0; i < nv; i++) { A[i * MJO_MAXV + i] += h * m->dof_damping[i]; qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i]; }
    chol_factor(L, A, nv, MJO_MAXV); chol_solve(L, qacc, nv, MJO_MAXV);
  } else memcpy(qacc, d->qacc, sizeof(double) * nv);
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
  integrate_pos(m, d->qpos, d->qvel, h);
  d->time += h;
}

/* [3P] mj_RungeKutta(N=4): classic RK4, full mj_forward (collision + solver) at every stage */
static void mjo_rk4(const mjoModel* m, mjoData* d) {
  static const double A[3][3] = {{0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 1}}, B[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
  int nv = m->nv, nq = m->nq; double h = m->timestep;
  double X0q[MJO_MAXQ], X0v[MJO_MAXV], Fv[4][MJO_MAXV], Fa[4][MJO_MAXV], dv[MJO_MAXV], da[MJO_MAXV];
  double t0 = d->time;
  memcpy(X0q, d->qpos, sizeof(double) * nq); memcpy(X0v, d->qvel, sizeof(double) * nv);
  memcpy(Fv[0], d->qvel, sizeof(double) * nv); memcpy(Fa[0], d->qacc, sizeof(double) * nv);
  for (int i = 1; i < 4; i++) {
    for (int k = 0; k < nv; k++) { dv[k] = 0; da[k] = 0; for (int j = 0; j < i; j++) { dv[k] += A[i - 1][j] * Fv[j][k]; da[k] += A[i - 1][j] * Fa[j][k]; } }
    memcpy(d->qpos, X0q, sizeof(double) * nq); integrate_pos(m, d->qpos, dv, h);
    for (int k = 0; k < nv; k++) d->qvel[k] = X0v[k] + h * da[k];
    double c = 0; for (int j = 0; j < 3; j++) c += A[i - 1][j];
    d->time = t0 + c * h;
    mjo_forward(m, d);
    memcpy(Fv[i], d->qvel, sizeof(double) * nv); memcpy(Fa[i], d->qacc, sizeof(double) * nv);
  }
  for (int k = 0; k < nv; k++) { dv[k] = 0; da[k] = 0; for (int j = 0; j < 4; j++) { dv[k] += B[j] * Fv[j][k]; da[k] += B[j] * Fa[j][k]; } }
  memcpy(d->qpos, X0q, sizeof(double) * nq);
  for (int k = 0; k < nv; k++) d->qvel[k] = X0v[k] + h * da[k];
  integrate_pos(m, d->qpos, dv, h);
  d->time = t0 + h;
}

/* [3P] mj_step */
void mjo_step(const mjoModel* m, mjoData* d) {
  mjo_forward(m, d);
  if (m->integrator == MJO_INT_RK4) mjo_rk4(m, d); else mjo_euler(m, d);
}

void mjo_reset_data(const mjoModel* m, mjoData* d) {
  memset(d, 0, sizeof *d);
  memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
}

void mjo_energy(const mjoModel* m, mjoData* d) {
  mjo_kinematics(m, d); mjo_mass_matrix(m, d);
  double pe = 0, ke = 0;
  for (int b = 1; b < m->nbody; b++) pe -= m->body_mass[b] * dot3(m->gravity, d->xipos[b]);
  for (int j = 0; j < m->njnt; j++) if (m->jnt_type[j] != MJO_JNT_FREE) {
    double x = d->qpos[m->jnt_qposadr[j]] - m->jnt_springref[j]; pe += 0.5 * m->jnt_stiffness[j] * x * x;
  }
  for (int i = 0; i < m->nv; i++) for (int k = 0; k < m->nv; k++) ke += 0.5 * d->qvel[i] * d->qM[i * MJO_MAXV + k] * d->qvel[k];
  d->energy[0] = pe; d->energy[1] = ke;
}

/* ======================================================================================== */
/*  Model builders: the reference's MJCF templates restated as code                          */
/* ======================================================================================== */
#define DEG (PI / 180.0)

/* random_envs/jinja/assets/hopper.xml (coordinate="global": every pos below is the XML's global
 * value minus the parent body's global pos; all body orientations are identity). */
int mjo_build_hopper(mjoModel* m, const double* size) {
  static const double def[4] = {.4, .45, .5, .39}; /* random_hopper.py:18 */
  const double* s = size ? size : def;
  model_init(m);
  m->timestep = 0.002; m->integrator = MJO_INT_RK4;             /* hopper.xml:17 */
  JntDef jd; jntdef_init(&jd); jd.armature = 1; jd.damping = 1; jd.limited = 1; /* hopper.xml:4 */
  JntDef root = jd; root.armature = 0; root.damping = 0; root.limited = 0; root.stiffness = 0; /* :29-31 */
  GeomDef gd; geomdef_init(&gd); gd.conaffinity = 1; gd.condim = 1; gd.contype = 1; gd.margin = 0.001; /* :5 */
  gd.solimp[0] = .8; gd.solimp[1] = .8; gd.solimp[2] = .01; gd.solref[0] = .02; gd.solref[1] = 1;
  GeomDef fl = gd; fl.condim = 3;                               /* floor, hopper.xml:26 */
  add_plane(m, &fl);
  double zt = s[0] / 2 + s[1] + s[2] + 0.1;                     /* torso body z, :27 */
  double X[3] = {1, 0, 0}, Z[3] = {0, 0, 1}, Y[3] = {0, 1, 0}, NY[3] = {0, -1, 0}, O[3] = {0, 0, 0};
  double p[3] = {0, 0, zt};
  int torso = add_body(m, 0, p, NULL);
  add_joint(m, torso, MJO_JNT_SLIDE, O, X, 0, &root, 0, 0, 0);
  add_joint(m, torso, MJO_JNT_SLIDE, O, Z, 1.25, &root, 0, 0, 0); /* ref="1.25", :30 */
  add_joint(m, torso, MJO_JNT_HINGE, O, Y, 0, &root, 0, 0, 0);    /* pos = body pos, :31 */
  GeomDef g09 = gd; g09.friction[0] = 0.9; GeomDef g20 = gd; g20.friction[0] = 2.0;
  double f[3] = {0, 0, s[0] + s[1] + s[2] + 0.1 - zt}, t[3] = {0, 0, s[1] + s[2] + 0.1 - zt};
  add_capsule_fromto(m, torso, f, t, 0.05, &g09);               /* :32 */
  /* thigh: body pos global "0 0 1.05" (literal), joint at z = s1+s2+0.1 */
  double zthigh = 1.05; double pth[3] = {0, 0, zthigh - zt};
  int thigh = add_body(m, torso, pth, NULL);
  double jp[3] = {0, 0, s[1] + s[2] + 0.1 - zthigh};
  int jthigh = add_joint(m, thigh, MJO_JNT_HINGE, jp, NY, 0, &jd, 1, -150 * DEG, 0); /* :34 */
  double f2[3] = {0, 0, s[1] + s[2] + 0.1 - zthigh}, t2[3] = {0, 0, s[2] + 0.1 - zthigh};
  add_capsule_fromto(m, thigh, f2, t2, 0.05, &g09);             /* :35 */
  double zleg = 0.35; double pl[3] = {0, 0, zleg - zthigh};
  int leg = add_body(m, thigh, pl, NULL);
  double jp3[3] = {0, 0, s[2] + 0.1 - zleg};
  int jleg = add_joint(m, leg, MJO_JNT_HINGE, jp3, NY, 0, &jd, 1, -150 * DEG, 0); /* :37 */
  double f3[3] = {0, 0, s[2] + 0.1 - zleg}, t3[3] = {0, 0, 0.1 - zleg};
  add_capsule_fromto(m, leg, f3, t3, 0.04, &g09);               /* :38 */
  double xf = s[3] / 6, zf = 0.1; double pf[3] = {xf, 0, zf - zleg};
  int foot = add_body(m, leg, pf, NULL);
  double jp4[3] = {0 - xf, 0, 0.1 - zf};
  int jfoot = add_joint(m, foot, MJO_JNT_HINGE, jp4, NY, 0, &jd, 1, -45 * DEG, 45 * DEG); /* :40 */
  double f4[3] = {-s[3] / 3 - xf, 0, 0.1 - zf}, t4[3] = {s[3] * 2 / 3 - xf, 0, 0.1 - zf};
  add_capsule_fromto(m, foot, f4, t4, 0.06, &g20);              /* :41 */
  add_motor(m, jthigh, 200, -1, 1); add_motor(m, jleg, 200, -1, 1); add_motor(m, jfoot, 200, -1, 1); /* :48-50 */
  compile(m, 0);
  return 0;
}

/* random_envs/jinja/assets/walker2d.xml */
int mjo_build_walker2d(mjoModel* m, const double* size) {
  static const double def[4] = {.4, .45, .6, .2}; /* random_walker2d.py:21 */
  const double* s = size ? size : def;
  model_init(m);
  m->timestep = 0.002; m->integrator = MJO_INT_RK4;             /* walker2d.xml:18 */
  JntDef jd; jntdef_init(&jd); jd.armature = 0.01; jd.damping = .1; jd.limited = 1; /* :4 */
  JntDef root = jd; root.armature = 0; root.damping = 0; root.limited = 0;
  GeomDef gd; geomdef_init(&gd); gd.conaffinity = 0; gd.condim = 3; gd.contype = 1; gd.density = 1000; /* :5 */
  gd.friction[0] = .7; gd.friction[1] = .1; gd.friction[2] = .1;
  GeomDef fl = gd; fl.conaffinity = 1;                          /* floor :24 */
  int floor = add_plane(m, &fl);
  double X[3] = {1, 0, 0}, Z[3] = {0, 0, 1}, Y[3] = {0, 1, 0}, NY[3] = {0, -1, 0}, O[3] = {0, 0, 0};
  double zt = 1.25; double p[3] = {0, 0, zt};                   /* :25 */
  int torso = add_body(m, 0, p, NULL);
  add_joint(m, torso, MJO_JNT_SLIDE, O, X, 0, &root, 0, 0, 0);
  add_joint(m, torso, MJO_JNT_SLIDE, O, Z, 1.25, &root, 0, 0, 0);
  add_joint(m, torso, MJO_JNT_HINGE, O, Y, 0, &root, 0, 0, 0);  /* pos "0 0 1.25" global = body origin */
  GeomDef g09 = gd; g09.friction[0] = 0.9; GeomDef g19 = gd; g19.friction[0] = 1.9;
  double f[3] = {0, 0, s[1] + s[2] + s[0] - zt}, t[3] = {0, 0, s[1] + s[2] - zt};
  add_capsule_fromto(m, torso, f, t, 0.05, &g09);               /* :30 */
  int footg[2], jn[6];
  for (int side = 0; side < 2; side++) {
    double zthigh = s[1] + s[2]; double pth[3] = {0, 0, zthigh - zt}; /* :31 / :44 */
    int thigh = add_body(m, torso, pth, NULL);
    double jp[3] = {0, 0, 0};
    jn[3 * side] = add_joint(m, thigh, MJO_JNT_HINGE, jp, NY, 0, &jd, 1, -150 * DEG, 0);
    double f2[3] = {0, 0, 0}, t2[3] = {0, 0, s[2] - zthigh};
    add_capsule_fromto(m, thigh, f2, t2, 0.05, &g09);
    double zleg = 0.35; double pl[3] = {0, 0, zleg - zthigh};   /* literal "0 0 0.35" :34 */
    int leg = add_body(m, thigh, pl, NULL);
    double jp3[3] = {0, 0, s[2] - zleg};
    jn[3 * side + 1] = add_joint(m, leg, MJO_JNT_HINGE, jp3, NY, 0, &jd, 1, -150 * DEG, 0);
    double f3[3] = {0, 0, s[2] - zleg}, t3[3] = {0, 0, 0.1 - zleg};
    add_capsule_fromto(m, leg, f3, t3, 0.04, &g09);
    /* foot body pos="0.2/2 0 0.1" (SURVEY Q17): the body frame does not enter the dynamics
     * (inertiafromgeom + global joint/geom coordinates); use x = 0.1 */
    double xf = 0.1, zf = 0.1; double pf[3] = {xf, 0, zf - zleg};
    int foot = add_body(m, leg, pf, NULL);
    double jp4[3] = {0 - xf, 0, 0.1 - zf};
    jn[3 * side + 2] = add_joint(m, foot, MJO_JNT_HINGE, jp4, NY, 0, &jd, 1, -45 * DEG, 45 * DEG);
    double f4[3] = {-0.0 - xf, 0, 0.1 - zf}, t4[3] = {s[3] - xf, 0, 0.1 - zf};
    footg[side] = add_capsule_fromto(m, foot, f4, t4, 0.06, side ? &g19 : &g09);
  }
  /* six motors, gear 100 (walker2d.xml:60-65) */
  for (int k = 0; k < 6; k++) add_motor(m, jn[k], 100, -1, 1);
  double fr[5] = {0.9, 0.9, .1, .1, .1}, fl2[5] = {1.9, 1.9, .1, .1, .1};
  add_pair(m, footg[0], floor, 3, fr); add_pair(m, footg[1], floor, 3, fl2); /* :70-71 */
  compile(m, 0);
  return 0;
}

/* random_envs/jinja/assets/half_cheetah.xml (coordinate="local", angles in radian) */
int mjo_build_halfcheetah(mjoModel* m, const double* size) {
  static const double def[8] = {1., .15, .145, .15, .094, .133, .106, .07}; /* random_half_cheetah.py:19 */
  const double* s = size ? size : def;
  model_init(m);
  m->timestep = 0.01; m->integrator = MJO_INT_EULER; m->gravity[2] = -9.81; /* :72 */
  JntDef jd; jntdef_init(&jd); jd.armature = .1; jd.damping = .01; jd.limited = 1; jd.stiffness = 8; /* :56 */
  jd.solimplimit[0] = 0; jd.solimplimit[1] = .8; jd.solimplimit[2] = .03; jd.solreflimit[0] = .02; jd.solreflimit[1] = 1;
  JntDef root = jd; root.armature = 0; root.damping = 0; root.limited = 0; root.stiffness = 0;
  GeomDef gd; geomdef_init(&gd); gd.conaffinity = 0; gd.condim = 3; gd.contype = 1; /* :57 */
  gd.friction[0] = .4; gd.friction[1] = .1; gd.friction[2] = .1;
  gd.solimp[0] = 0; gd.solimp[1] = .8; gd.solimp[2] = .01; gd.solref[0] = .02; gd.solref[1] = 1;
  GeomDef fl = gd; fl.conaffinity = 1;                          /* :85 */
  int floor = add_plane(m, &fl);
  double X[3] = {1, 0, 0}, Z[3] = {0, 0, 1}, Y[3] = {0, 1, 0}, O[3] = {0, 0, 0};
  double tl = s[0], head_angle = 0.87, head = s[1];
  double bth_a = -3.8, bth = s[2], bsh_a = -2.03, bsh = s[3], bft_a = -0.27, bft = s[4];
  double fth_a = 0.52, fth = s[5], fsh_a = -0.6, fsh = s[6], fft_a = -0.6, fft = s[7];
  double p[3] = {0, 0, .7};
  int torso = add_body(m, 0, p, NULL);                          /* :86 */
  add_joint(m, torso, MJO_JNT_SLIDE, O, X, 0, &root, 0, 0, 0);
  add_joint(m, torso, MJO_JNT_SLIDE, O, Z, 0, &root, 0, 0, 0);
  add_joint(m, torso, MJO_JNT_HINGE, O, Y, 0, &root, 0, 0, 0);
  double f[3] = {-tl / 2, 0, 0}, t[3] = {tl / 2, 0, 0};
  add_capsule_fromto(m, torso, f, t, 0.046, &gd);               /* :91 */
  double q[4], gp[3];
  q_axisangle(q, Y, head_angle); gp[0] = tl / 2 + head * cos(head_angle); gp[1] = 0; gp[2] = head * cos(head_angle);
  add_capsule_pos(m, torso, gp, q, 0.046, head, &gd);           /* :92 (z uses cos, as the template does) */
  struct { double ang, len, stiff, damp, lo, hi; } J[6] = {
      {bth_a, bth, 240, 6, -.52, 1.05}, {bsh_a, bsh, 180, 4.5, -.785, .785}, {bft_a, bft, 120, 3, -.4, .785},
      {fth_a, fth, 180, 4.5, -1, .7},   {fsh_a, fsh, 120, 3, -1.2, .87},      {fft_a, fft, 60, 1.5, -.5, .5}};
  int jn[6], footg[2];
  /* back leg :93-104 */
  double pb[3] = {-tl / 2, 0, 0};
  int bthigh = add_body(m, torso, pb, NULL);
  JntDef j0 = jd; j0.damping = J[0].damp; j0.stiffness = J[0].stiff;
  jn[0] = add_joint(m, bthigh, MJO_JNT_HINGE, O, Y, 0, &j0, 1, J[0].lo, J[0].hi);
  q_axisangle(q, Y, bth_a); gp[0] = bth * sin(bth_a); gp[1] = 0; gp[2] = bth * cos(bth_a);
  add_capsule_pos(m, bthigh, gp, q, 0.046, bth, &gd);
  double pbs[3] = {2 * bth * sin(bth_a), 0, 2 * bth * cos(bth_a)};
  int bshin = add_body(m, bthigh, pbs, NULL);
  JntDef j1 = jd; j1.damping = J[1].damp; j1.stiffness = J[1].stiff;
  jn[1] = add_joint(m, bshin, MJO_JNT_HINGE, O, Y, 0, &j1, 1, J[1].lo, J[1].hi);
  q_axisangle(q, Y, bsh_a); gp[0] = bsh * sin(bsh_a); gp[2] = bsh * cos(bsh_a);
  add_capsule_pos(m, bshin, gp, q, 0.046, bsh, &gd);
  double pbf[3] = {2 * bsh * sin(bsh_a), 0, 2 * bsh * cos(bsh_a)};
  int bfoot = add_body(m, bshin, pbf, NULL);
  JntDef j2 = jd; j2.damping = J[2].damp; j2.stiffness = J[2].stiff;
  jn[2] = add_joint(m, bfoot, MJO_JNT_HINGE, O, Y, 0, &j2, 1, J[2].lo, J[2].hi);
  q_axisangle(q, Y, bft_a); gp[0] = sin(-bft_a) * bft; gp[2] = -bft;
  footg[0] = add_capsule_pos(m, bfoot, gp, q, 0.046, bft, &gd);
  /* front leg :105-117 */
  double pf[3] = {tl / 2, 0, 0};
  int fthigh = add_body(m, torso, pf, NULL);
  JntDef j3 = jd; j3.damping = J[3].damp; j3.stiffness = J[3].stiff;
  jn[3] = add_joint(m, fthigh, MJO_JNT_HINGE, O, Y, 0, &j3, 1, J[3].lo, J[3].hi);
  q_axisangle(q, Y, fth_a); gp[0] = fth * sin(-fth_a); gp[2] = -fth * cos(fth_a);
  add_capsule_pos(m, fthigh, gp, q, 0.046, fth, &gd);
  double pfs[3] = {2 * fth * sin(-fth_a), 0, -2 * fth * cos(fth_a)};
  int fshin = add_body(m, fthigh, pfs, NULL);
  JntDef j4 = jd; j4.damping = J[4].damp; j4.stiffness = J[4].stiff;
  jn[4] = add_joint(m, fshin, MJO_JNT_HINGE, O, Y, 0, &j4, 1, J[4].lo, J[4].hi);
  q_axisangle(q, Y, fsh_a); gp[0] = fsh * sin(-fsh_a); gp[2] = -fsh * cos(fsh_a);
  add_capsule_pos(m, fshin, gp, q, 0.046, fsh, &gd);
  double pff[3] = {2 * fsh * sin(-fsh_a), 0, -2 * fsh * cos(fsh_a)};
  int ffoot = add_body(m, fshin, pff, NULL);
  JntDef j5 = jd; j5.damping = J[5].damp; j5.stiffness = J[5].stiff;
  jn[5] = add_joint(m, ffoot, MJO_JNT_HINGE, O, Y, 0, &j5, 1, J[5].lo, J[5].hi);
  q_axisangle(q, Y, fft_a); gp[0] = sin(-fft_a) * fft * 9 / 8; gp[2] = -fft;
  footg[1] = add_capsule_pos(m, ffoot, gp, q, 0.046, fft, &gd);
  static const double gear[6] = {120, 90, 60, 120, 60, 30};     /* :121-126 */
  for (int k = 0; k < 6; k++) add_motor(m, jn[k], gear[k], -1, 1);
  double fr[5] = {.4, .4, .1, .1, .1};                          /* class foot-floor :63-65 */
  add_pair(m, footg[0], floor, 3, fr); add_pair(m, footg[1], floor, 3, fr); /* :129-132 */
  compile(m, 14.0);                                             /* settotalmass="14" :54 */
  return 0;
}


/* random_envs/jinja/assets/humanoid.xml (local coordinates, angles in degrees, inertiafromgeom) */
int mjo_build_humanoid(mjoModel* m) {
  model_init(m);
  m->timestep = 0.003; m->integrator = MJO_INT_RK4; m->solver = MJO_SOL_PGS; m->iterations = 50;   /* humanoid.xml:9 */
  JntDef jd; jntdef_init(&jd); jd.armature = 1; jd.damping = 1; jd.limited = 1;                      /* :4 */
  JntDef root = jd; root.armature = 0; root.damping = 0; root.limited = 0; root.stiffness = 0;         /* :32 */
  GeomDef gd; geomdef_init(&gd); gd.conaffinity = 1; gd.condim = 1; gd.contype = 1; gd.margin = 0.001; /* :5 */
  GeomDef fl = gd; fl.condim = 3; fl.friction[0] = 1; fl.friction[1] = .1; fl.friction[2] = .1;        /* :28 */
  add_plane(m, &fl);
  double O[3] = {0, 0, 0};
  struct J { const char* name; double arm, ax[3], damp, pos[3], lo, hi, stiff; };
#define JNT(body, A, X, Y, Z, D, PX, PY, PZ, LO, HI, K) do { JntDef t = jd; t.armature = A; t.damping = D; t.stiffness = K; \
    double ax_[3] = {X, Y, Z}, ps_[3] = {PX, PY, PZ}; jn[nj++] = add_joint(m, body, MJO_JNT_HINGE, ps_, ax_, 0, &t, 1, (LO) * DEG, (HI) * DEG); } while (0)
  int jn[17], nj = 0;
  double p[3], f[3], t[3], q[4];
  p[0] = 0; p[1] = 0; p[2] = 1.4;
  int torso = add_body(m, 0, p, NULL);                                                                  /* :30 */
  add_joint(m, torso, MJO_JNT_FREE, O, NULL, 0, &root, 0, 0, 0);                                        /* :32 */
#define V3(v, a, b, c) do { v[0] = a; v[1] = b; v[2] = c; } while (0)
  V3(f, 0, -.07, 0); V3(t, 0, .07, 0); add_capsule_fromto(m, torso, f, t, 0.07, &gd);                   /* :33 */
  V3(p, 0, 0, .19); add_sphere(m, torso, p, .09, &gd);                                                  /* :34 */
  V3(f, -.01, -.06, -.12); V3(t, -.01, .06, -.12); add_capsule_fromto(m, torso, f, t, 0.06, &gd);       /* :35 */
  q[0] = 1.0; q[1] = 0; q[2] = -0.002; q[3] = 0;
  V3(p, -.01, 0, -0.260); int lwaist = add_body(m, torso, p, q);                                        /* :36 */
  V3(f, 0, -.06, 0); V3(t, 0, .06, 0); add_capsule_fromto(m, lwaist, f, t, 0.06, &gd);                  /* :37 */
  JNT(lwaist, 0.02, 0, 0, 1, 5, 0, 0, 0.065, -45, 45, 20);                                              /* abdomen_z :38 */
  JNT(lwaist, 0.02, 0, 1, 0, 5, 0, 0, 0.065, -75, 30, 10);                                              /* abdomen_y :39 */
  V3(p, 0, 0, -0.165); int pelvis = add_body(m, lwaist, p, q);                                          /* :40 */
  JNT(pelvis, 0.02, 1, 0, 0, 5, 0, 0, 0.1, -35, 35, 10);                                                /* abdomen_x :41 */
  V3(f, -.02, -.07, 0); V3(t, -.02, .07, 0); add_capsule_fromto(m, pelvis, f, t, 0.09, &gd);            /* butt :42 */
  /* right leg :43-56 */
  V3(p, 0, -0.1, -0.04); int rthigh = add_body(m, pelvis, p, NULL);
  JNT(rthigh, 0.01, 1, 0, 0, 5, 0, 0, 0, -25, 5, 10);                                                   /* right_hip_x */
  JNT(rthigh, 0.01, 0, 0, 1, 5, 0, 0, 0, -60, 35, 10);                                                  /* right_hip_z */
  JNT(rthigh, 0.0080, 0, 1, 0, 5, 0, 0, 0, -110, 20, 20);                                               /* right_hip_y */
  V3(f, 0, 0, 0); V3(t, 0, 0.01, -.34); add_capsule_fromto(m, rthigh, f, t, 0.06, &gd);
  V3(p, 0, 0.01, -0.403); int rshin = add_body(m, rthigh, p, NULL);
  JNT(rshin, 0.0060, 0, -1, 0, 1, 0, 0, .02, -160, -2, 0);                                              /* right_knee (default damping 1) */
  V3(f, 0, 0, 0); V3(t, 0, 0, -.3); add_capsule_fromto(m, rshin, f, t, 0.049, &gd);
  V3(p, 0, 0, -0.45); int rfoot = add_body(m, rshin, p, NULL);
  V3(p, 0, 0, 0.1); add_sphere(m, rfoot, p, 0.075, &gd);
  /* left leg :57-70 */
  V3(p, 0, 0.1, -0.04); int lthigh = add_body(m, pelvis, p, NULL);
  JNT(lthigh, 0.01, -1, 0, 0, 5, 0, 0, 0, -25, 5, 10);                                                  /* left_hip_x */
  JNT(lthigh, 0.01, 0, 0, -1, 5, 0, 0, 0, -60, 35, 10);                                                 /* left_hip_z */
  JNT(lthigh, 0.01, 0, 1, 0, 5, 0, 0, 0, -110, 20, 20);                                                 /* left_hip_y */
  V3(f, 0, 0, 0); V3(t, 0, -0.01, -.34); add_capsule_fromto(m, lthigh, f, t, 0.06, &gd);
  V3(p, 0, -0.01, -0.403); int lshin = add_body(m, lthigh, p, NULL);
  JNT(lshin, 0.0060, 0, -1, 0, 1, 0, 0, .02, -160, -2, 1);                                              /* left_knee stiffness 1 */
  V3(f, 0, 0, 0); V3(t, 0, 0, -.3); add_capsule_fromto(m, lshin, f, t, 0.049, &gd);
  V3(p, 0, 0, -0.45); int lfoot = add_body(m, lshin, p, NULL);
  V3(p, 0, 0, 0.1); add_sphere(m, lfoot, p, 0.075, &gd);
  /* arms :72-91 */
  V3(p, 0, -0.17, 0.06); int ruarm = add_body(m, torso, p, NULL);
  JNT(ruarm, 0.0068, 2, 1, 1, 1, 0, 0, 0, -85, 60, 1);                                                  /* right_shoulder1 */
  JNT(ruarm, 0.0051, 0, -1, 1, 1, 0, 0, 0, -85, 60, 1);                                                 /* right_shoulder2 */
  V3(f, 0, 0, 0); V3(t, .16, -.16, -.16); add_capsule_fromto(m, ruarm, f, t, 0.04, &gd);
  V3(p, .18, -.18, -.18); int rlarm = add_body(m, ruarm, p, NULL);
  JNT(rlarm, 0.0028, 0, -1, 1, 1, 0, 0, 0, -90, 50, 0);                                                 /* right_elbow */
  V3(f, .01, .01, .01); V3(t, .17, .17, .17); add_capsule_fromto(m, rlarm, f, t, 0.031, &gd);
  V3(p, .18, .18, .18); add_sphere(m, rlarm, p, 0.04, &gd);
  V3(p, 0, 0.17, 0.06); int luarm = add_body(m, torso, p, NULL);
  JNT(luarm, 0.0068, 2, -1, 1, 1, 0, 0, 0, -60, 85, 1);                                                 /* left_shoulder1 */
  JNT(luarm, 0.0051, 0, 1, 1, 1, 0, 0, 0, -60, 85, 1);                                                  /* left_shoulder2 */
  V3(f, 0, 0, 0); V3(t, .16, .16, -.16); add_capsule_fromto(m, luarm, f, t, 0.04, &gd);
  V3(p, .18, .18, -.18); int llarm = add_body(m, luarm, p, NULL);
  JNT(llarm, 0.0028, 0, -1, -1, 1, 0, 0, 0, -90, 50, 0);                                                /* left_elbow */
  V3(f, .01, -.01, .01); V3(t, .17, -.17, .17); add_capsule_fromto(m, llarm, f, t, 0.031, &gd);
  V3(p, .18, -.18, .18); add_sphere(m, llarm, p, 0.04, &gd);
  /* motors :106-122 -- note the order: abdomen_y, abdomen_z, abdomen_x, ... */
  static const int order[17] = {1, 0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  static const double gear[17] = {100, 100, 100, 100, 100, 300, 200, 100, 100, 300, 200, 25, 25, 25, 25, 25, 25};
  for (int k = 0; k < 17; k++) add_motor(m, jn[order[k]], gear[k], -0.4, 0.4);                          /* ctrlrange :6 */
  compile(m, 0);
  return 0;
#undef JNT
#undef V3
}

/* [3P] mj_comPos / mj_comVel quantities entering the humanoid observation: cinert (rotational inertia about
 * the root's subtree COM in world axes, mass*offset, mass) and cvel (spatial velocity referred to that point).
 * subtree COM uses the CURRENT body_mass but the compile-time body_subtreemass (not refreshed by set_task). */
void mjo_com_quantities(const mjoModel* m, mjoData* d) {
  double sc[3] = {0, 0, 0};
  for (int b = 1; b < m->nbody; b++) addscl3(sc, d->xipos[b], m->body_mass[b]);
  double sm = m->body_subtreemass[1];      /* root body = torso (body 1) */
  if (sm < MINVAL) copy3(sc, d->xipos[1]); else { sc[0] /= sm; sc[1] /= sm; sc[2] /= sm; }
  copy3(d->subtree_com_root, sc);
  memset(d->cinert[0], 0, sizeof d->cinert[0]); memset(d->cvel[0], 0, sizeof d->cvel[0]);
  for (int b = 1; b < m->nbody; b++) {
    const double* R = d->xmat[b]; double Iw[9], RI[9], dif[3], mass = m->body_mass[b];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += R[i * 3 + k] * m->body_inertia[b][k * 3 + j]; RI[i * 3 + j] = s; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += RI[i * 3 + k] * R[j * 3 + k]; Iw[i * 3 + j] = s; }
    sub3(dif, d->xipos[b], sc);
    double* c = d->cinert[b];
    c[0] = Iw[0] + mass * (dif[1] * dif[1] + dif[2] * dif[2]);
    c[1] = Iw[4] + mass * (dif[0] * dif[0] + dif[2] * dif[2]);
    c[2] = Iw[8] + mass * (dif[0] * dif[0] + dif[1] * dif[1]);
    c[3] = Iw[1] - mass * dif[0] * dif[1];
    c[4] = Iw[2] - mass * dif[0] * dif[2];
    c[5] = Iw[5] - mass * dif[1] * dif[2];
    c[6] = mass * dif[0]; c[7] = mass * dif[1]; c[8] = mass * dif[2]; c[9] = mass;
    /* bvel = (w; v at world origin)  ->  v at sc = v_O + w x sc */
    double t[3]; cross3(t, d->bvel[b], sc);
    copy3(d->cvel[b], d->bvel[b]); add3(d->cvel[b] + 3, d->bvel[b] + 3, t);
  }
}
