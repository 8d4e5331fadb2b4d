// humanoid_host.cpp -- TEST HARNESS ONLY (never loaded by the product): host instantiation of the
// humanoid kernels' math (random-envs_amd/csrc/humanoid_engine.hpp) in fp32 / fp64.
#include <cstring>
#include "../../random-envs_amd/csrc/humanoid_model.hpp"

using namespace rex::hum;

template <class T> struct Ctx { Model<T> m; Kin<T> k; Scratch<T> s; };
template <class T> static Ctx<T>* ctx() {
  static Ctx<T>* c = nullptr;
  if (!c) { c = new Ctx<T>(); Model<double> md; build_model(md); convert_model(md, c->m); }
  return c;
}
template <class T> static void lane_from_xi(const Model<T>& m, const double* xi, size_t stride, Lane<T>& L) {
  L.mass[0] = 0;
  for (int b = 1; b < NBODY; b++) L.mass[b] = xi ? T(xi[(size_t)(b - 1) * stride]) : m.body_mass0[b];
  for (int d = 0; d < 6; d++) L.damping[d] = 0;
  for (int d = 6; d < NV; d++) L.damping[d] = xi ? T(xi[(size_t)(13 + d - 6) * stride]) : m.dof_damping0[d];
}

template <class T>
static void run_step(int n, const double* qpos, const double* qvel, const double* act, const double* xi, const double* xprev,
                     double* qpos_out, double* qvel_out, double* obs, double* reward, unsigned char* done, double* xout, int* overflow) {
  Ctx<T>* c = ctx<T>();
  for (int i = 0; i < n; i++) {
    Lane<T> L; lane_from_xi(c->m, xi + i, n, L);
    T q[NQ], v[NV], a[NU], xp[NBODY];
    for (int k = 0; k < NQ; k++) q[k] = T(qpos[(size_t)k * n + i]);
    for (int k = 0; k < NV; k++) v[k] = T(qvel[(size_t)k * n + i]);
    for (int k = 0; k < NU; k++) a[k] = T(act[(size_t)k * n + i]);
    if (xprev && xprev[i] == xprev[i]) for (int b = 0; b < NBODY; b++) xp[b] = T(xprev[(size_t)b * n + i]);
    else env_reset_obs(c->m, L, q, v, xp, c->k, c->s, [](int, T) {});
    T r; bool d;
    env_step(c->m, L, q, v, a, xp, c->k, c->s, r, d, [&](int k, T val) { obs[(size_t)k * n + i] = double(val); });
    for (int k = 0; k < NQ; k++) qpos_out[(size_t)k * n + i] = double(q[k]);
    for (int k = 0; k < NV; k++) qvel_out[(size_t)k * n + i] = double(v[k]);
    reward[i] = double(r); done[i] = d; if (overflow) overflow[i] = c->k.overflow;
    if (xout) for (int b = 0; b < NBODY; b++) xout[(size_t)b * n + i] = double(xp[b]);
  }
}

template <class T>
static void run_forward(const double* qpos, const double* qvel, const double* ctrl, const double* xi, double* qacc, double* M, int* info) {
  Ctx<T>* c = ctx<T>();
  Lane<T> L; lane_from_xi(c->m, xi, 1, L);
  T q[NQ], v[NV], a[NU], acc[NV];
  for (int k = 0; k < NQ; k++) q[k] = T(qpos[k]);
  for (int k = 0; k < NV; k++) v[k] = T(qvel[k]);
  for (int k = 0; k < NU; k++) a[k] = T(ctrl[k]);
  // M before factorisation
  { Smooth<T> S; kinematics(c->m, q, S, c->s); com_pos(c->m, L, S, c->s);
    MassFactor<T> F; crb(c->m, S, F);
    for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) M[i * NV + j] = double(F.get(i, j)); }
  int it = forward(c->m, L, q, v, a, c->k, c->s, acc);
  for (int k = 0; k < NV; k++) qacc[k] = double(acc[k]);
  info[0] = c->k.ncon; info[1] = c->k.nefc; info[2] = it; info[3] = c->k.overflow;
}

extern "C" {
int hh_constants(double* mass, double* invw_body, double* invw_dof, double* ipos, double* inertia, int* npair) {
  Ctx<double>* c = ctx<double>();
  for (int b = 0; b < NBODY; b++) { mass[b] = c->m.body_mass0[b]; invw_body[2 * b] = c->m.body_invw[b][0]; invw_body[2 * b + 1] = c->m.body_invw[b][1];
    for (int k = 0; k < 3; k++) ipos[3 * b + k] = c->m.body_ipos[b][k]; for (int k = 0; k < 6; k++) inertia[6 * b + k] = c->m.body_inertia[b][k]; }
  for (int d = 0; d < NV; d++) invw_dof[d] = c->m.dof_invw[d];
  *npair = c->m.npair;
  return 0;
}
int hh_step(int f32, int n, const double* qpos, const double* qvel, const double* act, const double* xi, const double* xprev,
            double* qpos_out, double* qvel_out, double* obs, double* reward, unsigned char* done, double* xout, int* overflow) {
  if (f32) run_step<float>(n, qpos, qvel, act, xi, xprev, qpos_out, qvel_out, obs, reward, done, xout, overflow);
  else run_step<double>(n, qpos, qvel, act, xi, xprev, qpos_out, qvel_out, obs, reward, done, xout, overflow);
  return 0;
}
int hh_check_topology() { Model<double> md; build_model(md); return check_topology(md) ? 1 : 0; }
int hh_forward(int f32, const double* qpos, const double* qvel, const double* ctrl, const double* xi, double* qacc, double* M, int* info) {
  if (f32) run_forward<float>(qpos, qvel, ctrl, xi, qacc, M, info); else run_forward<double>(qpos, qvel, ctrl, xi, qacc, M, info);
  return 0;
}
}
extern "C" int hh_contacts(double* out, int maxn) {   // contacts of the last hh_forward<double>
  Ctx<double>* c = ctx<double>(); int n = c->k.ncon < maxn ? c->k.ncon : maxn;
  for (int i = 0; i < n; i++) { out[5 * i] = c->s.cb1[i]; out[5 * i + 1] = c->s.cb2[i]; out[5 * i + 2] = c->s.cdist[i]; out[5 * i + 3] = c->s.cdim[i]; out[5 * i + 4] = c->s.cpos[i][2]; }
  return n;
}
