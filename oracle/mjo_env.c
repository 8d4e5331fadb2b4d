/*
 * mjo_env.c -- ORACLE (test infrastructure, NOT product code).  See mjo.h.
 *
 * Env-level restatement of the reference's task-definition files: xi <-> model scatter
 * (get_task/set_task), reward / done / observation formulas, plus batch helpers for ctypes and a
 * pthread-partitioned rollout used as bench.py's `cpu_baseline` ("port").
 */
#include "mjo.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static const double HOPPER_SIZE[4] = {.4, .45, .5, .39};       /* random_hopper.py:18 */
static const double WALKER_SIZE[4] = {.4, .45, .6, .2};        /* random_walker2d.py:21 */

int mjo_env_init(mjoEnv* e, int kind, int variant) {
  memset(e, 0, sizeof *e);
  e->kind = kind; e->variant = variant; e->endless = 0;
  switch (kind) {
    case MJO_ENV_HOPPER:      /* random_hopper.py:30 frame_skip 4; obs 11; xi = 4 masses */
      memcpy(e->size, HOPPER_SIZE, sizeof HOPPER_SIZE); mjo_build_hopper(&e->model, e->size);
      e->frame_skip = 4; e->obs_dim = 11; e->task_dim = variant ? 3 : 4; e->act_dim = 3;
      /* Unmodeled: torso mass frozen at 0.8x nominal (random_hopper_unmodeled.py:24-26) */
      if (variant) e->model.body_mass[1] *= 0.8;
      break;
    case MJO_ENV_WALKER2D:    /* random_walker2d.py:32; xi = 7 masses + 4 lengths + 2 frictions */
      memcpy(e->size, WALKER_SIZE, sizeof WALKER_SIZE);
      if (variant) e->size[0] *= 0.8;   /* torso length frozen at 0.8x (random_walker2d_unmodeled.py:25-27) */
      mjo_build_walker2d(&e->model, e->size);
      e->frame_skip = 4; e->obs_dim = 17; e->task_dim = variant ? 9 : 13; e->act_dim = 6;
      /* masses 1..3 at 0.8x until the first set_task rebuilds the model (:33-36) */
      if (variant) for (int b = 1; b <= 3; b++) e->model.body_mass[b] *= 0.8;
      break;
    case MJO_ENV_HALFCHEETAH: /* random_half_cheetah.py:33; xi = 7 masses + 1 friction */
      mjo_build_halfcheetah(&e->model, NULL);
      e->frame_skip = 5; e->obs_dim = 17; e->task_dim = variant ? 5 : 8; e->act_dim = 6;
      if (variant) for (int b = 1; b <= 3; b++) e->model.body_mass[b] *= 0.8;   /* random_half_cheetah_unmodeled.py:28-31 */
      break;
    case MJO_ENV_HUMANOID:    /* random_humanoid.py:41 frame_skip 5; obs 376; xi = 13 masses + 17 dampings */
      mjo_build_humanoid(&e->model);
      e->frame_skip = 5; e->obs_dim = 376; e->task_dim = variant ? 23 : 30; e->act_dim = 17;
      if (variant) {   /* random_humanoid_unmodeled.py:40-50 */
        for (int b = 1; b <= 4; b++) e->model.body_mass[b] *= 0.8;
        for (int d = 6; d <= 8; d++) e->model.dof_damping[d] *= 0.8;
      }
      break;
    default: return -1;
  }
  mjo_reset_data(&e->model, &e->data);
  return 0;
}

/* solver tolerance override used by the parity tests (<=0: keep the model's 1e-8) */
static double g_tolerance = 0;
void mjo_set_tolerance(double tol) { g_tolerance = tol; }

void mjo_env_get_task(const mjoEnv* e, double* xi) {
  const mjoModel* m = &e->model;
  if (e->variant) {   /* Unmodeled ids: the frozen prefix is not part of xi */
    switch (e->kind) {
      case MJO_ENV_HOPPER:      /* random_hopper_unmodeled.py:71-73 */
        for (int i = 0; i < 3; i++) xi[i] = m->body_mass[2 + i];
        break;
      case MJO_ENV_HALFCHEETAH: /* random_half_cheetah_unmodeled.py:82-86 */
        for (int i = 0; i < 4; i++) xi[i] = m->body_mass[4 + i];
        xi[4] = m->pair_friction[0][0]; break;
      case MJO_ENV_WALKER2D:    /* random_walker2d_unmodeled.py:104-107 */
        for (int i = 0; i < 4; i++) xi[i] = m->body_mass[4 + i];
        for (int i = 0; i < 3; i++) xi[4 + i] = e->size[1 + i];
        xi[7] = m->pair_friction[0][0]; xi[8] = m->pair_friction[1][0]; break;
      case MJO_ENV_HUMANOID:    /* random_humanoid_unmodeled.py:167-170 */
        for (int i = 0; i < 9; i++) xi[i] = m->body_mass[5 + i];
        for (int i = 0; i < 14; i++) xi[9 + i] = m->dof_damping[9 + i];
        break;
    }
    return;
  }
  switch (e->kind) {
    case MJO_ENV_HOPPER:      /* random_hopper.py:75-77 */
      for (int i = 0; i < 4; i++) xi[i] = m->body_mass[1 + i];
      break;
    case MJO_ENV_HALFCHEETAH: /* random_half_cheetah.py:85-89 */
      for (int i = 0; i < 7; i++) xi[i] = m->body_mass[1 + i];
      xi[7] = m->pair_friction[0][0]; break;
    case MJO_ENV_WALKER2D:    /* random_walker2d.py:101-104 */
      for (int i = 0; i < 7; i++) xi[i] = m->body_mass[1 + i];
      for (int i = 0; i < 4; i++) xi[7 + i] = e->size[i];
      xi[11] = m->pair_friction[0][0]; xi[12] = m->pair_friction[1][0]; break;
    case MJO_ENV_HUMANOID:    /* random_humanoid.py:151-154 */
      for (int i = 0; i < 13; i++) xi[i] = m->body_mass[1 + i];
      for (int i = 0; i < 17; i++) xi[13 + i] = m->dof_damping[6 + i];
      break;
  }
}

void mjo_env_set_task(mjoEnv* e, const double* xi) {
  mjoModel* m = &e->model;
  if (e->variant) {
    switch (e->kind) {
      case MJO_ENV_HOPPER:      /* random_hopper_unmodeled.py:75-76 */
        for (int i = 0; i < 3; i++) m->body_mass[2 + i] = xi[i];
        break;
      case MJO_ENV_HALFCHEETAH: /* random_half_cheetah_unmodeled.py:94-95 */
        for (int i = 0; i < 4; i++) m->body_mass[4 + i] = xi[i];
        for (int p = 0; p < 2; p++) { m->pair_friction[p][0] = xi[4]; m->pair_friction[p][1] = xi[4]; } break;
      case MJO_ENV_WALKER2D:    /* random_walker2d_unmodeled.py:109-116: the rebuild drops the 0.8x mass scaling */
        for (int i = 0; i < 3; i++) e->size[1 + i] = xi[4 + i];
        mjo_build_walker2d(m, e->size);
        for (int i = 0; i < 4; i++) m->body_mass[4 + i] = xi[i];
        m->pair_friction[0][0] = xi[7]; m->pair_friction[0][1] = xi[7];
        m->pair_friction[1][0] = xi[8]; m->pair_friction[1][1] = xi[8];
        mjo_reset_data(m, &e->data);
        break;
      case MJO_ENV_HUMANOID:    /* random_humanoid_unmodeled.py:172-174 */
        for (int i = 0; i < 9; i++) m->body_mass[5 + i] = xi[i];
        for (int i = 0; i < 14; i++) m->dof_damping[9 + i] = xi[9 + i];
        break;
    }
    return;
  }
  switch (e->kind) {
    case MJO_ENV_HOPPER:      /* random_hopper.py:79-80: body_mass[1:] = task (inertia, invweight0 untouched) */
      for (int i = 0; i < 4; i++) m->body_mass[1 + i] = xi[i];
      break;
    case MJO_ENV_HALFCHEETAH: /* random_half_cheetah.py:97-98: pair_friction[0:2,0:2] = task[-1] */
      for (int i = 0; i < 7; i++) m->body_mass[1 + i] = xi[i];
      for (int p = 0; p < 2; p++) { m->pair_friction[p][0] = xi[7]; m->pair_friction[p][1] = xi[7]; } break;
    case MJO_ENV_WALKER2D: {  /* random_walker2d.py:106-113: rebuild from lengths, then masses, then frictions */
      double qpos[MJO_MAXQ], qvel[MJO_MAXV];
      memcpy(qpos, e->data.qpos, sizeof qpos); memcpy(qvel, e->data.qvel, sizeof qvel);
      for (int i = 0; i < 4; i++) e->size[i] = xi[7 + i];
      mjo_build_walker2d(m, e->size);
      for (int i = 0; i < 7; i++) m->body_mass[1 + i] = xi[i];
      m->pair_friction[0][0] = xi[11]; m->pair_friction[0][1] = xi[11];
      m->pair_friction[1][0] = xi[12]; m->pair_friction[1][1] = xi[12];
      /* build_model creates a fresh MjSim (jinja_mujoco_env.py:92-97): state back to qpos0 */
      mjo_reset_data(m, &e->data);
      (void)qpos; (void)qvel;
      break; }
    case MJO_ENV_HUMANOID:    /* random_humanoid.py:156-158 */
      for (int i = 0; i < 13; i++) m->body_mass[1 + i] = xi[i];
      for (int i = 0; i < 17; i++) m->dof_damping[6 + i] = xi[13 + i];
      break;
  }
}

void mjo_env_set_state(mjoEnv* e, const double* qpos, const double* qvel) {
  /* MujocoEnv.set_state (jinja_mujoco_env.py:146-154) */
  memcpy(e->data.qpos, qpos, sizeof(double) * e->model.nq);
  memcpy(e->data.qvel, qvel, sizeof(double) * e->model.nv);
  e->xipos_valid = 0;   /* set_state runs sim.forward(): done lazily by the humanoid step */
}

static void humanoid_obs(mjoEnv* e, double* obs) {   /* random_humanoid.py:190-216 (noise-free) */
  mjoData* d = &e->data; int c = 0;
  mjo_com_quantities(&e->model, d);
  for (int k = 2; k < 24; k++) obs[c++] = d->qpos[k];
  for (int k = 0; k < 23; k++) obs[c++] = d->qvel[k];
  for (int b = 0; b < 14; b++) for (int k = 0; k < 10; k++) obs[c++] = d->cinert[b][k];
  for (int b = 0; b < 14; b++) for (int k = 0; k < 6; k++) obs[c++] = d->cvel[b][k];
  for (int k = 0; k < 23; k++) obs[c++] = d->qfrc_actuator[k];
  for (int k = 0; k < 84; k++) obs[c++] = 0.0;   /* cfrc_ext, SURVEY Q15 */
}

void mjo_env_obs(const mjoEnv* e, double* obs) {
  if (e->kind == MJO_ENV_HUMANOID) { humanoid_obs((mjoEnv*)e, obs); return; }
  /* _get_obs: concat(qpos[1:], qvel) -- random_hopper.py:100-110, random_half_cheetah.py:112-121,
   * random_walker2d.py:133-142 (noise is added by the caller, it is not part of the oracle) */
  int nq = e->model.nq, nv = e->model.nv;
  for (int i = 1; i < nq; i++) obs[i - 1] = e->data.qpos[i];
  for (int i = 0; i < nv; i++) obs[nq - 1 + i] = e->data.qvel[i];
}

int mjo_env_step(mjoEnv* e, const double* a, double* obs, double* reward) {
  mjoModel* m = &e->model; mjoData* d = &e->data;
  if (g_tolerance > 0) m->tolerance = g_tolerance;
  if (e->kind == MJO_ENV_HUMANOID) {   /* random_humanoid.py:161-188 */
    if (!e->xipos_valid) { mjo_forward(m, d); for (int b = 0; b < m->nbody; b++) e->xipos_x_prev[b] = d->xipos[b][0]; e->xipos_valid = 1; }
    double s0 = 0, s1 = 0, mt = 0, asq = 0;
    for (int b = 0; b < m->nbody; b++) { s0 += m->body_mass[b] * e->xipos_x_prev[b]; mt += m->body_mass[b]; }
    for (int u = 0; u < m->nu; u++) { d->ctrl[u] = a[u]; asq += a[u] * a[u]; }
    for (int k = 0; k < e->frame_skip; k++) mjo_step(m, d);
    for (int b = 0; b < m->nbody; b++) { e->xipos_x_prev[b] = d->xipos[b][0]; s1 += m->body_mass[b] * d->xipos[b][0]; }
    *reward = 1.25 * (s1 / mt - s0 / mt) / (m->timestep * e->frame_skip) - 0.1 * asq + 5.0;
    int dn = (d->qpos[2] < 1.0) || (d->qpos[2] > 2.0);
    if (e->endless) dn = 0;
    if (obs) humanoid_obs(e, obs);
    return dn;
  }
  double posbefore = d->qpos[0];
  /* do_simulation (jinja_mujoco_env.py:170-173) */
  for (int u = 0; u < m->nu; u++) d->ctrl[u] = a[u];
  for (int k = 0; k < e->frame_skip; k++) mjo_step(m, d);
  double dt = m->timestep * e->frame_skip;
  double posafter = d->qpos[0], height = d->qpos[1], ang = d->qpos[2];
  double asq = 0; for (int u = 0; u < m->nu; u++) asq += a[u] * a[u];
  int done = 0;
  switch (e->kind) {
    case MJO_ENV_HOPPER: {    /* random_hopper.py:83-98 */
      *reward = (posafter - posbefore) / dt + 1.0 - 1e-3 * asq;
      int ok = 1;
      for (int i = 0; i < m->nq; i++) if (!isfinite(d->qpos[i])) ok = 0;
      for (int i = 0; i < m->nv; i++) if (!isfinite(d->qvel[i])) ok = 0;
      for (int i = 2; i < m->nq; i++) if (!(fabs(d->qpos[i]) < 100)) ok = 0;
      for (int i = 0; i < m->nv; i++) if (!(fabs(d->qvel[i]) < 100)) ok = 0;
      done = !(ok && height > .7 && fabs(ang) < .2);
      break; }
    case MJO_ENV_WALKER2D:    /* random_walker2d.py:116-131 */
      *reward = (posafter - posbefore) / dt + 1.0 - 1e-3 * asq;
      done = !(height > 0.8 && height < 2.0 && ang > -1.0 && ang < 1.0);
      break;
    case MJO_ENV_HALFCHEETAH: /* random_half_cheetah.py:101-110 */
      *reward = -0.1 * asq + (posafter - posbefore) / dt;
      done = 0; break;
  }
  if (e->endless) done = 0;
  if (obs) mjo_env_obs(e, obs);
  return done;
}

/* ------------------------------------------------------------------ batch helpers ------- */
typedef struct {
  int kind, variant, n, lo, hi, steps;
  const double *qpos, *qvel, *action, *xi;
  double *qpos_out, *qvel_out, *obs_out, *reward_out; unsigned char* done_out;
  const double *qpos_reset, *qvel_reset;   /* auto-reset rollouts (bench.py's cpu_baseline): state a finished lane restarts from */
  long long* resets_out;
} BatchJob;

static void* batch_worker(void* arg) {
  BatchJob* j = (BatchJob*)arg;
  mjoEnv* e = (mjoEnv*)malloc(sizeof(mjoEnv));
  mjo_env_init(e, j->kind, j->variant);
  int nq = e->model.nq, nv = e->model.nv, nu = e->act_dim, nx = e->task_dim, n = j->n;
  double q[MJO_MAXQ], v[MJO_MAXV], a[MJO_MAXU], xi[32], last_xi[32], obs[384];
  int have_xi = 0;
  for (int i = j->lo; i < j->hi; i++) {
    for (int k = 0; k < nx; k++) xi[k] = j->xi[(size_t)k * n + i];
    /* Walker2d set_task recompiles the model: skip when xi is unchanged */
    /* a NaN xi keeps the freshly constructed model (used to test the Unmodeled ids before any set_task) */
    if (xi[0] == xi[0] && (!have_xi || memcmp(xi, last_xi, sizeof(double) * nx))) { mjo_env_set_task(e, xi); memcpy(last_xi, xi, sizeof(double) * nx); have_xi = 1; }
    for (int k = 0; k < nq; k++) q[k] = j->qpos[(size_t)k * n + i];
    for (int k = 0; k < nv; k++) v[k] = j->qvel[(size_t)k * n + i];
    mjo_env_set_state(e, q, v);
    double rsum = 0; int done = 0; long long nreset = 0;
    for (int s = 0; s < j->steps; s++) {
      for (int k = 0; k < nu; k++) a[k] = j->action[((size_t)s * nu + k) * n + i];
      double r; done = mjo_env_step(e, a, obs, &r); rsum += r;
      if (done && j->qpos_reset) {   /* the vectorised env's auto-reset: reset_model() state drawn by the caller */
        for (int k = 0; k < nq; k++) q[k] = j->qpos_reset[(size_t)k * n + i];
        for (int k = 0; k < nv; k++) v[k] = j->qvel_reset[(size_t)k * n + i];
        mjo_env_set_state(e, q, v); nreset++;
      }
    }
    if (j->resets_out) j->resets_out[i] = nreset;
    for (int k = 0; k < nq; k++) j->qpos_out[(size_t)k * n + i] = e->data.qpos[k];
    for (int k = 0; k < nv; k++) j->qvel_out[(size_t)k * n + i] = e->data.qvel[k];
    if (j->obs_out) for (int k = 0; k < e->obs_dim; k++) j->obs_out[(size_t)k * n + i] = obs[k];
    if (j->reward_out) j->reward_out[i] = rsum;
    if (j->done_out) j->done_out[i] = (unsigned char)done;
  }
  free(e);
  return NULL;
}

static int run_batch(BatchJob* proto, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > proto->n) nthreads = proto->n > 0 ? proto->n : 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
  BatchJob* jobs = (BatchJob*)malloc(sizeof(BatchJob) * nthreads);
  for (int t = 0; t < nthreads; t++) {
    jobs[t] = *proto;
    jobs[t].lo = (int)((long long)proto->n * t / nthreads); jobs[t].hi = (int)((long long)proto->n * (t + 1) / nthreads);
    if (nthreads == 1) batch_worker(&jobs[t]); else pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
  }
  if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  free(th); free(jobs);
  return 0;
}

int mjo_batch_step(int kind, int variant, int n, const double* qpos, const double* qvel, const double* action,
                   const double* xi, double* qpos_out, double* qvel_out, double* obs_out, double* reward_out,
                   unsigned char* done_out, int nthreads) {
  BatchJob j = {kind, variant, n, 0, n, 1, qpos, qvel, action, xi, qpos_out, qvel_out, obs_out, reward_out, done_out, NULL, NULL, NULL};
  return run_batch(&j, nthreads);
}

int mjo_batch_rollout(int kind, int variant, int n, int steps, const double* qpos, const double* qvel,
                      const double* actions, const double* xi, double* qpos_out, double* qvel_out,
                      double* reward_sum, int nthreads) {
  BatchJob j = {kind, variant, n, 0, n, steps, qpos, qvel, actions, xi, qpos_out, qvel_out, NULL, reward_sum, NULL, NULL, NULL, NULL};
  return run_batch(&j, nthreads);
}

/* the same rollout with the batched env's auto-reset: a lane whose step returns done restarts from (qpos_reset, qvel_reset)[lane]
 * (reset_model() states drawn by the caller) -- what the GPU leg of bench.py does inside its timed region */
int mjo_batch_rollout_autoreset(int kind, int variant, int n, int steps, const double* qpos, const double* qvel,
                                const double* actions, const double* xi, const double* qpos_reset, const double* qvel_reset,
                                double* qpos_out, double* qvel_out, double* reward_sum, long long* resets_out, int nthreads) {
  BatchJob j = {kind, variant, n, 0, n, steps, qpos, qvel, actions, xi, qpos_out, qvel_out, NULL, reward_sum, NULL, qpos_reset, qvel_reset, resets_out};
  return run_batch(&j, nthreads);
}

int mjo_probe_forward(int kind, const double* qpos, const double* qvel, const double* action, const double* xi,
                      double* qacc, double* qacc_smooth, double* qM, double* qfrc_bias, int* ncon, int* nefc,
                      double* efc_force, int* niter) {
  mjoEnv* e = (mjoEnv*)malloc(sizeof(mjoEnv));
  if (mjo_env_init(e, kind, 0)) { free(e); return -1; }
  if (xi) mjo_env_set_task(e, xi);
  mjo_env_set_state(e, qpos, qvel);
  int nv = e->model.nv;
  for (int u = 0; u < e->model.nu; u++) e->data.ctrl[u] = action ? action[u] : 0;
  if (g_tolerance > 0) e->model.tolerance = g_tolerance;
  mjo_forward(&e->model, &e->data);
  for (int i = 0; i < nv; i++) { qacc[i] = e->data.qacc[i]; qacc_smooth[i] = e->data.qacc_smooth[i]; qfrc_bias[i] = e->data.qfrc_bias[i];
    for (int k = 0; k < nv; k++) qM[i * nv + k] = e->data.qM[i * MJO_MAXV + k]; }
  *ncon = e->data.ncon; *nefc = e->data.nefc; *niter = e->data.solver_iter;
  if (efc_force) for (int i = 0; i < e->data.nefc; i++) efc_force[i] = e->data.efc_force[i];
  free(e);
  return 0;
}

int mjo_model_constants(int kind, const double* size, double* body_mass, double* body_inertia, double* body_ipos,
                        double* body_invweight0, double* dof_invweight0, double* qpos0, int* dims) {
  mjoModel* m = (mjoModel*)malloc(sizeof(mjoModel));
  switch (kind) {
    case MJO_ENV_HOPPER: mjo_build_hopper(m, size); break;
    case MJO_ENV_WALKER2D: mjo_build_walker2d(m, size); break;
    case MJO_ENV_HALFCHEETAH: mjo_build_halfcheetah(m, size); break;
    case MJO_ENV_HUMANOID: mjo_build_humanoid(m); break;
    default: free(m); return -1;
  }
  for (int b = 0; b < m->nbody; b++) {
    body_mass[b] = m->body_mass[b];
    for (int k = 0; k < 9; k++) body_inertia[b * 9 + k] = m->body_inertia[b][k];
    for (int k = 0; k < 3; k++) body_ipos[b * 3 + k] = m->body_ipos[b][k];
    body_invweight0[b * 2] = m->body_invweight0[b][0]; body_invweight0[b * 2 + 1] = m->body_invweight0[b][1];
  }
  for (int i = 0; i < m->nv; i++) dof_invweight0[i] = m->dof_invweight0[i];
  for (int i = 0; i < m->nq; i++) qpos0[i] = m->qpos0[i];
  dims[0] = m->nbody; dims[1] = m->nq; dims[2] = m->nv; dims[3] = m->ngeom; dims[4] = m->nu; dims[5] = m->npair;
  free(m);
  return 0;
}

/* self-consistency probe: conservative variant of a model (no damping, no limits, no contacts,
 * no actuation); returns total energy before / after `steps` mj_steps. RK4 -> drift O(h^4). */
int mjo_test_energy_drift(int kind, int steps, const double* qpos, const double* qvel, int keep_springs,
                          double* e0, double* e1) {
  mjoEnv* e = (mjoEnv*)malloc(sizeof(mjoEnv));
  if (mjo_env_init(e, kind, 0)) { free(e); return -1; }
  mjoModel* m = &e->model;
  for (int i = 0; i < m->nv; i++) m->dof_damping[i] = 0;
  for (int j = 0; j < m->njnt; j++) { m->jnt_limited[j] = 0; if (!keep_springs) m->jnt_stiffness[j] = 0; }
  m->npair = 0; m->integrator = MJO_INT_RK4;
  mjo_env_set_state(e, qpos, qvel);
  mjo_energy(m, &e->data); *e0 = e->data.energy[0] + e->data.energy[1];
  for (int s = 0; s < steps; s++) mjo_step(m, &e->data);
  mjo_energy(m, &e->data); *e1 = e->data.energy[0] + e->data.energy[1];
  free(e);
  return 0;
}

/* contact list of one state (debug / tests): out rows = (geom1, geom2, dist, pos x,y,z, normal x,y,z, dim) */
int mjo_probe_contacts(int kind, const double* qpos, const double* qvel, const double* xi, double* out, int max_rows) {
  mjoEnv* e = (mjoEnv*)malloc(sizeof(mjoEnv));
  if (mjo_env_init(e, kind, 0)) { free(e); return -1; }
  if (xi) mjo_env_set_task(e, xi);
  mjo_env_set_state(e, qpos, qvel);
  mjo_forward(&e->model, &e->data);
  int n = e->data.ncon < max_rows ? e->data.ncon : max_rows;
  for (int c = 0; c < n; c++) {
    const mjoContact* k = e->data.contact + c; double* r = out + 10 * c;
    r[0] = k->geom1; r[1] = k->geom2; r[2] = k->dist; r[3] = k->pos[0]; r[4] = k->pos[1]; r[5] = k->pos[2];
    r[6] = k->frame[0]; r[7] = k->frame[1]; r[8] = k->frame[2]; r[9] = k->dim;
  }
  free(e);
  return n;
}

/* world-frame dump of a compiled model at qpos0 (tests compare it with tests/golden/mjcf_tables.json):
 * jnt rows: type, anchor(3), axis(3), limited, lo, hi, armature, damping, stiffness, qpos0  (14)
 * geom rows: type, p0(3), p1(3), radius, friction0, condim, margin, body                   (12)
 * act rows: joint, gear, lo, hi (4);  opt: timestep, integrator, solver, iterations, gravity_z (5) */
int mjo_model_dump(int kind, const double* size, double* jnt, double* geom, double* act, double* opt, int* dims) {
  mjoModel* m = (mjoModel*)malloc(sizeof(mjoModel)); mjoData* d = (mjoData*)malloc(sizeof(mjoData));
  switch (kind) {
    case MJO_ENV_HOPPER: mjo_build_hopper(m, size); break;
    case MJO_ENV_WALKER2D: mjo_build_walker2d(m, size); break;
    case MJO_ENV_HALFCHEETAH: mjo_build_halfcheetah(m, size); break;
    case MJO_ENV_HUMANOID: mjo_build_humanoid(m); break;
    default: free(m); free(d); return -1;
  }
  mjo_reset_data(m, d); mjo_forward(m, d);
  for (int j = 0; j < m->njnt; j++) {
    double* r = jnt + 14 * j; int da = m->jnt_dofadr[j] + (m->jnt_type[j] == MJO_JNT_FREE ? 3 : 0);
    r[0] = m->jnt_type[j];
    for (int k = 0; k < 3; k++) { r[1 + k] = d->dof_anchor[da][k]; r[4 + k] = m->jnt_type[j] == MJO_JNT_FREE ? 0 : d->dof_axis[da][k]; }
    r[7] = m->jnt_limited[j]; r[8] = m->jnt_range[j][0]; r[9] = m->jnt_range[j][1];
    r[10] = m->dof_armature[m->jnt_dofadr[j]]; r[11] = m->dof_damping[m->jnt_dofadr[j]]; r[12] = m->jnt_stiffness[j];
    r[13] = m->qpos0[m->jnt_qposadr[j]];
  }
  for (int g = 0; g < m->ngeom; g++) {
    double* r = geom + 12 * g; const double* X = d->geom_xmat[g]; double ax[3] = {X[2], X[5], X[8]}, h = m->geom_size[g][1];
    r[0] = m->geom_type[g];
    for (int k = 0; k < 3; k++) { double hh = m->geom_type[g] == MJO_GEOM_CAPSULE ? h : 0; r[1 + k] = d->geom_xpos[g][k] + ax[k] * hh; r[4 + k] = d->geom_xpos[g][k] - ax[k] * hh; }
    r[7] = m->geom_size[g][0]; r[8] = m->geom_friction[g][0]; r[9] = m->geom_condim[g]; r[10] = m->geom_margin[g]; r[11] = m->geom_body[g];
  }
  for (int u = 0; u < m->nu; u++) { double* r = act + 4 * u; r[0] = m->dof_jnt[m->act_dof[u]]; r[1] = m->act_gear[u]; r[2] = m->act_ctrlrange[u][0]; r[3] = m->act_ctrlrange[u][1]; }
  opt[0] = m->timestep; opt[1] = m->integrator; opt[2] = m->solver; opt[3] = m->iterations; opt[4] = m->gravity[2];
  dims[0] = m->nbody; dims[1] = m->njnt; dims[2] = m->ngeom; dims[3] = m->nu; dims[4] = m->npair;
  for (int p = 0; p < m->npair && p < 8; p++) { opt[5 + 3 * p] = m->pair_explicit[p]; opt[6 + 3 * p] = m->pair_friction[p][0]; opt[7 + 3 * p] = m->pair_dim[p]; }
  free(m); free(d);
  return 0;
}
