/*
 * mjo_humanoid.c -- ORACLE (test infrastructure, NOT product code).  See mjo.h.
 * Env-level restatement of random_envs/jinja/random_humanoid.py: xi <-> model scatter, reward with
 * mass_center(), done, 376-dim observation.  Data-staleness of the reference is reproduced: every
 * `data.*` field read by step()/_get_obs() other than qpos/qvel comes from the LAST mj_forward, which
 * inside mj_step+RK4 is the 4th stage evaluation (not the integrated end state).
 */
#include "mjo.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define NB 14 /* bodies incl. world */

/* mass_center (random_humanoid.py:22-25): sum(m * xipos_x) / sum(m) with the CURRENT body_mass */
static double mass_center_x(const mjoModel* m, const double* xipos_x) {
  double s = 0, mt = 0;
  for (int b = 0; b < NB; b++) { s += m->body_mass[b] * xipos_x[b]; mt += m->body_mass[b]; }
  return s / mt;
}

typedef struct {
  int n, lo, hi;
  const double *qpos, *qvel, *action, *xi, *xprev;
  double *qpos_out, *qvel_out, *obs_out, *reward_out, *xout; unsigned char* done_out;
} HJob;

static double g_tol = 0;
void mjo_humanoid_set_tolerance(double t) { g_tol = t; }

static void* hworker(void* arg) {
  HJob* j = (HJob*)arg; int n = j->n;
  mjoModel* m = (mjoModel*)malloc(sizeof(mjoModel)); mjoData* d = (mjoData*)malloc(sizeof(mjoData));
  mjo_build_humanoid(m);
  if (g_tol > 0) m->tolerance = g_tol;
  for (int i = j->lo; i < j->hi; i++) {
    /* set_task (random_humanoid.py:156-158): body_mass[1:] = xi[:13]; dof_damping[6:] = xi[13:] */
    for (int k = 0; k < 13; k++) m->body_mass[1 + k] = j->xi[(size_t)k * n + i];
    for (int k = 0; k < 17; k++) m->dof_damping[6 + k] = j->xi[(size_t)(13 + k) * n + i];
    mjo_reset_data(m, d);
    for (int k = 0; k < 24; k++) d->qpos[k] = j->qpos[(size_t)k * n + i];
    for (int k = 0; k < 23; k++) d->qvel[k] = j->qvel[(size_t)k * n + i];
    double xprev[NB];
    if (j->xprev && j->xprev[i] == j->xprev[i]) { for (int b = 0; b < NB; b++) xprev[b] = j->xprev[(size_t)b * n + i]; }
    else { mjo_forward(m, d); for (int b = 0; b < NB; b++) xprev[b] = d->xipos[b][0]; }   /* set_state -> sim.forward() */
    double pos_before = mass_center_x(m, xprev);                       /* :162 */
    double a[17], asq = 0;
    for (int u = 0; u < 17; u++) { a[u] = j->action[(size_t)u * n + i]; d->ctrl[u] = a[u]; asq += a[u] * a[u]; }
    for (int s = 0; s < 5; s++) mjo_step(m, d);                        /* do_simulation, frame_skip 5 (:41) */
    double xnow[NB]; for (int b = 0; b < NB; b++) xnow[b] = d->xipos[b][0];   /* left by the last RK4 stage forward */
    double pos_after = mass_center_x(m, xnow);
    double dt = m->timestep * 5;
    /* data.ctrl holds the raw action (mujoco-py does not clamp data.ctrl in place) :167 */
    double reward = 1.25 * (pos_after - pos_before) / dt - 0.1 * asq - 0.0 /* cfrc_ext == 0, SURVEY Q15 */ + 5.0;
    int done = (d->qpos[2] < 1.0) || (d->qpos[2] > 2.0);               /* :173 */
    /* _get_obs :190-216 */
    mjo_com_quantities(m, d);
    double* o = j->obs_out; int c = 0;
    if (o) {
      for (int k = 2; k < 24; k++) o[(size_t)(c++) * n + i] = d->qpos[k];
      for (int k = 0; k < 23; k++) o[(size_t)(c++) * n + i] = d->qvel[k];
      for (int b = 0; b < NB; b++) for (int k = 0; k < 10; k++) o[(size_t)(c++) * n + i] = d->cinert[b][k];
      for (int b = 0; b < NB; b++) for (int k = 0; k < 6; k++) o[(size_t)(c++) * n + i] = d->cvel[b][k];
      for (int k = 0; k < 23; k++) o[(size_t)(c++) * n + i] = d->qfrc_actuator[k];
      for (int k = 0; k < 84; k++) o[(size_t)(c++) * n + i] = 0.0;     /* cfrc_ext: not computed without sensors [3P], Q15 */
    }
    for (int k = 0; k < 24; k++) j->qpos_out[(size_t)k * n + i] = d->qpos[k];
    for (int k = 0; k < 23; k++) j->qvel_out[(size_t)k * n + i] = d->qvel[k];
    if (j->reward_out) j->reward_out[i] = reward;
    if (j->done_out) j->done_out[i] = (unsigned char)done;
    if (j->xout) for (int b = 0; b < NB; b++) j->xout[(size_t)b * n + i] = xnow[b];
  }
  free(m); free(d);
  return NULL;
}

int mjo_humanoid_batch_step(int n, const double* qpos, const double* qvel, const double* action, const double* xi,
                            const double* xipos_x_prev, double* qpos_out, double* qvel_out, double* obs_out,
                            double* reward_out, unsigned char* done_out, double* xipos_x_out, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > n) nthreads = n > 0 ? n : 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
  HJob* jobs = (HJob*)malloc(sizeof(HJob) * nthreads);
  for (int t = 0; t < nthreads; t++) {
    HJob j = {n, (int)((long long)n * t / nthreads), (int)((long long)n * (t + 1) / nthreads), qpos, qvel, action, xi,
              xipos_x_prev, qpos_out, qvel_out, obs_out, reward_out, xipos_x_out, done_out};
    jobs[t] = j;
    if (nthreads == 1) hworker(&jobs[t]); else pthread_create(&th[t], NULL, hworker, &jobs[t]);
  }
  if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  free(th); free(jobs);
  return 0;
}

/* Observation returned by reset() / after set_state (random_humanoid.py:219-234, jinja_mujoco_env.py:146-154):
 * set_state(qpos, qvel) runs sim.forward() with the masses / dampings in force at that moment (`xi` here -- under
 * dr_training these are the PREVIOUS episode's, because set_random_task comes after set_state: SURVEY Q10) and with
 * data.ctrl = 0 (sim.reset()); _get_obs then reads qpos, qvel and the data.* fields that forward left behind. */
int mjo_humanoid_batch_reset_obs(int n, const double* qpos, const double* qvel, const double* xi, double* obs_out,
                                 double* xipos_x_out) {
  mjoModel* m = (mjoModel*)malloc(sizeof(mjoModel)); mjoData* d = (mjoData*)malloc(sizeof(mjoData));
  mjo_build_humanoid(m);
  if (g_tol > 0) m->tolerance = g_tol;
  for (int i = 0; i < n; i++) {
    for (int k = 0; k < 13; k++) m->body_mass[1 + k] = xi[(size_t)k * n + i];
    for (int k = 0; k < 17; k++) m->dof_damping[6 + k] = xi[(size_t)(13 + k) * n + i];
    mjo_reset_data(m, d);                                              /* sim.reset(): ctrl = 0 */
    for (int k = 0; k < 24; k++) d->qpos[k] = qpos[(size_t)k * n + i];
    for (int k = 0; k < 23; k++) d->qvel[k] = qvel[(size_t)k * n + i];
    mjo_forward(m, d);                                                 /* set_state -> sim.forward() */
    mjo_com_quantities(m, d);
    int c = 0;
    for (int k = 2; k < 24; k++) obs_out[(size_t)(c++) * n + i] = d->qpos[k];
    for (int k = 0; k < 23; k++) obs_out[(size_t)(c++) * n + i] = d->qvel[k];
    for (int b = 0; b < NB; b++) for (int k = 0; k < 10; k++) obs_out[(size_t)(c++) * n + i] = d->cinert[b][k];
    for (int b = 0; b < NB; b++) for (int k = 0; k < 6; k++) obs_out[(size_t)(c++) * n + i] = d->cvel[b][k];
    for (int k = 0; k < 23; k++) obs_out[(size_t)(c++) * n + i] = d->qfrc_actuator[k];
    for (int k = 0; k < 84; k++) obs_out[(size_t)(c++) * n + i] = 0.0; /* cfrc_ext, Q15 */
    if (xipos_x_out) for (int b = 0; b < NB; b++) xipos_x_out[(size_t)b * n + i] = d->xipos[b][0];
  }
  free(m); free(d);
  return 0;
}

/* compiled constants + one forward probe for tests */
int mjo_humanoid_probe(const double* qpos, const double* qvel, const double* ctrl, const double* xi, double* body_mass,
                       double* qacc, double* qM, int* ncon, int* nefc, int* niter, double* efc_force, double* contacts, int max_con) {
  mjoModel* m = (mjoModel*)malloc(sizeof(mjoModel)); mjoData* d = (mjoData*)malloc(sizeof(mjoData));
  mjo_build_humanoid(m);
  if (g_tol > 0) m->tolerance = g_tol;
  if (body_mass) for (int b = 0; b < NB; b++) body_mass[b] = m->body_mass[b];
  if (xi) { for (int k = 0; k < 13; k++) m->body_mass[1 + k] = xi[k]; for (int k = 0; k < 17; k++) m->dof_damping[6 + k] = xi[13 + k]; }
  mjo_reset_data(m, d);
  if (qpos) for (int k = 0; k < 24; k++) d->qpos[k] = qpos[k];
  if (qvel) for (int k = 0; k < 23; k++) d->qvel[k] = qvel[k];
  if (ctrl) for (int u = 0; u < 17; u++) d->ctrl[u] = ctrl[u];
  mjo_forward(m, d);
  if (qacc) for (int k = 0; k < 23; k++) qacc[k] = d->qacc[k];
  if (qM) for (int i = 0; i < 23; i++) for (int k = 0; k < 23; k++) qM[i * 23 + k] = d->qM[i * MJO_MAXV + k];
  *ncon = d->ncon; *nefc = d->nefc; *niter = d->solver_iter;
  if (efc_force) for (int i = 0; i < d->nefc; i++) efc_force[i] = d->efc_force[i];
  if (contacts) for (int c = 0; c < d->ncon && c < max_con; c++) { contacts[4 * c] = d->contact[c].geom1; contacts[4 * c + 1] = d->contact[c].geom2; contacts[4 * c + 2] = d->contact[c].dist; contacts[4 * c + 3] = d->contact[c].dim; }
  int np = m->npair;
  free(m); free(d);
  return np;
}
