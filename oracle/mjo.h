/*
 * mjo.h -- ORACLE (test infrastructure, NOT product code).
 *
 * fp64 CPU restatement of the physics the reference reaches through `self.sim.step()`
 * (random_envs/jinja/jinja_mujoco_env.py:170-173): MuJoCo 2.1.0's mj_step for the fixed
 * kinematic trees described by random_envs/jinja/assets/{hopper,walker2d,half_cheetah,
 * humanoid}.xml.  MuJoCo itself is a third-party, un-vendored dependency of the reference
 * (requirements.txt:3 `mujoco-py<2.2,>=2.1` -> libmujoco210), absent from this image, and the
 * reference holds no numeric test for this path: PARITY WITH REAL mujoco-py IS UNPINNED.
 * What is restated here is MuJoCo's published "Computation" chapter (soft constraints with
 * solref/solimp impedance, pyramidal friction cones, regularisation R from invweight0,
 * primal Newton solver, RK4 / semi-implicit Euler with implicit joint damping) applied to the
 * constants in the reference's XML files.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may link or call this.
 */
#ifndef MJO_H_
#define MJO_H_

#ifdef __cplusplus
extern "C" {
#endif

#define MJO_MAXBODY 16
#define MJO_MAXJNT 24
#define MJO_MAXQ 26
#define MJO_MAXV 24
#define MJO_MAXGEOM 24
#define MJO_MAXPAIR 256
#define MJO_MAXU 20
#define MJO_MAXCON 96
#define MJO_MAXEFC 320

enum { MJO_JNT_FREE = 0, MJO_JNT_SLIDE = 2, MJO_JNT_HINGE = 3 };
enum { MJO_GEOM_PLANE = 0, MJO_GEOM_SPHERE = 2, MJO_GEOM_CAPSULE = 3 };
enum { MJO_INT_EULER = 0, MJO_INT_RK4 = 1 };
enum { MJO_SOL_PGS = 0, MJO_SOL_NEWTON = 2 };
enum { MJO_ENV_CARTPOLE = 0, MJO_ENV_HOPPER = 1, MJO_ENV_HALFCHEETAH = 2, MJO_ENV_WALKER2D = 3,
       MJO_ENV_HUMANOID = 4 };

typedef struct mjoModel {
  int nbody, njnt, nq, nv, ngeom, nu, npair;
  /* options */
  double timestep, gravity[3], tolerance, impratio;
  int integrator, solver, iterations;
  /* bodies (0 = world) */
  int body_parent[MJO_MAXBODY], body_jntadr[MJO_MAXBODY], body_jntnum[MJO_MAXBODY];
  int body_dofadr[MJO_MAXBODY], body_dofnum[MJO_MAXBODY], body_weldid[MJO_MAXBODY];
  double body_pos[MJO_MAXBODY][3], body_quat[MJO_MAXBODY][4];
  double body_ipos[MJO_MAXBODY][3];         /* COM in body frame */
  double body_inertia[MJO_MAXBODY][9];      /* full 3x3 about COM, body frame axes */
  double body_mass[MJO_MAXBODY], body_subtreemass[MJO_MAXBODY];
  double body_invweight0[MJO_MAXBODY][2];
  /* joints */
  int jnt_type[MJO_MAXJNT], jnt_body[MJO_MAXJNT], jnt_qposadr[MJO_MAXJNT], jnt_dofadr[MJO_MAXJNT];
  int jnt_limited[MJO_MAXJNT];
  double jnt_pos[MJO_MAXJNT][3], jnt_axis[MJO_MAXJNT][3], jnt_range[MJO_MAXJNT][2];
  double jnt_stiffness[MJO_MAXJNT], jnt_springref[MJO_MAXJNT], jnt_ref[MJO_MAXJNT];
  double jnt_margin[MJO_MAXJNT], jnt_solref[MJO_MAXJNT][2], jnt_solimp[MJO_MAXJNT][5];
  double qpos0[MJO_MAXQ];
  /* dofs */
  int dof_body[MJO_MAXV], dof_jnt[MJO_MAXV];
  double dof_armature[MJO_MAXV], dof_damping[MJO_MAXV], dof_invweight0[MJO_MAXV];
  /* geoms */
  int geom_type[MJO_MAXGEOM], geom_body[MJO_MAXGEOM], geom_contype[MJO_MAXGEOM];
  int geom_conaffinity[MJO_MAXGEOM], geom_condim[MJO_MAXGEOM];
  double geom_size[MJO_MAXGEOM][3], geom_pos[MJO_MAXGEOM][3], geom_quat[MJO_MAXGEOM][4];
  double geom_friction[MJO_MAXGEOM][3], geom_margin[MJO_MAXGEOM], geom_gap[MJO_MAXGEOM];
  double geom_solref[MJO_MAXGEOM][2], geom_solimp[MJO_MAXGEOM][5], geom_solmix[MJO_MAXGEOM];
  double geom_density[MJO_MAXGEOM];
  /* collision pairs after compile (explicit <pair>s first, then filtered dynamic pairs) */
  int pair_geom1[MJO_MAXPAIR], pair_geom2[MJO_MAXPAIR], pair_dim[MJO_MAXPAIR];
  int pair_explicit[MJO_MAXPAIR];
  double pair_friction[MJO_MAXPAIR][5], pair_solref[MJO_MAXPAIR][2], pair_solimp[MJO_MAXPAIR][5];
  double pair_margin[MJO_MAXPAIR], pair_gap[MJO_MAXPAIR];
  /* actuators: motors on joints */
  int act_dof[MJO_MAXU];
  double act_gear[MJO_MAXU], act_ctrlrange[MJO_MAXU][2];
  int act_ctrllimited[MJO_MAXU];
  double meaninertia;
} mjoModel;

typedef struct mjoContact {
  double dist, pos[3], frame[9], includemargin, friction[5], solref[2], solimp[5];
  int dim, geom1, geom2, efc_address;
} mjoContact;

typedef struct mjoData {
  double time;
  double qpos[MJO_MAXQ], qvel[MJO_MAXV], ctrl[MJO_MAXU];
  double qacc[MJO_MAXV], qacc_smooth[MJO_MAXV], qfrc_smooth[MJO_MAXV];
  double qfrc_bias[MJO_MAXV], qfrc_passive[MJO_MAXV], qfrc_actuator[MJO_MAXV];
  double qfrc_constraint[MJO_MAXV];
  double qM[MJO_MAXV * MJO_MAXV];
  /* kinematics */
  double xpos[MJO_MAXBODY][3], xquat[MJO_MAXBODY][4], xmat[MJO_MAXBODY][9];
  double xipos[MJO_MAXBODY][3];
  double geom_xpos[MJO_MAXGEOM][3], geom_xmat[MJO_MAXGEOM][9];
  double dof_axis[MJO_MAXV][3], dof_anchor[MJO_MAXV][3]; /* world frame */
  int dof_kind[MJO_MAXV];                                  /* 0 translation, 1 rotation */
  /* body spatial velocity about world origin (ang; lin) */
  double bvel[MJO_MAXBODY][6];
  /* contacts / constraints */
  int ncon, nefc, nlimit;
  mjoContact contact[MJO_MAXCON];
  double efc_J[MJO_MAXEFC * MJO_MAXV], efc_pos[MJO_MAXEFC], efc_margin[MJO_MAXEFC];
  double efc_R[MJO_MAXEFC], efc_D[MJO_MAXEFC], efc_aref[MJO_MAXEFC], efc_force[MJO_MAXEFC];
  double efc_diagApprox[MJO_MAXEFC], efc_KBIP[MJO_MAXEFC][4], efc_vel[MJO_MAXEFC];
  int efc_type[MJO_MAXEFC]; /* 0 limit, 1 frictionless contact, 2 pyramidal contact row */
  int efc_id[MJO_MAXEFC];
  int solver_iter;
  double energy[2];
  /* com-based quantities of the humanoid observation ([3P] mj_comPos / mj_comVel): subtree COM of the
   * root body, per-body cinert (10) and cvel (6) */
  double subtree_com_root[3];
  double cinert[MJO_MAXBODY][10], cvel[MJO_MAXBODY][6];
} mjoData;

/* ---- model builders: restate the reference's MJCF templates ---------------------------- */
/* size: the Jinja `size` list (random_hopper.py:18, random_walker2d.py:21, random_half_cheetah.py:19);
 * NULL = reference defaults. */
int mjo_build_hopper(mjoModel* m, const double* size);
int mjo_build_walker2d(mjoModel* m, const double* size);
int mjo_build_halfcheetah(mjoModel* m, const double* size);
int mjo_build_humanoid(mjoModel* m);

void mjo_reset_data(const mjoModel* m, mjoData* d);
void mjo_forward(const mjoModel* m, mjoData* d);
void mjo_step(const mjoModel* m, mjoData* d);
/* total mechanical energy (potential, kinetic) for self-consistency tests */
void mjo_energy(const mjoModel* m, mjoData* d);
void mjo_com_quantities(const mjoModel* m, mjoData* d);

/* ---- env-level oracle: one reference env object (task definition files) ---------------- */
typedef struct mjoEnv {
  int kind, variant;
  int frame_skip, obs_dim, task_dim, act_dim;
  int endless;
  mjoModel model;
  mjoData data;
  double size[8];
  double xipos_x_prev[MJO_MAXBODY];   /* humanoid: data.xipos[:,0] of the last mj_forward */
  int xipos_valid;
} mjoEnv;

int mjo_env_init(mjoEnv* e, int kind, int variant);
/* get_task / set_task of the task files (e.g. random_hopper.py:75-80). Walker2d rebuilds the
 * model from the xi lengths (random_walker2d.py:106-113). */
void mjo_env_set_task(mjoEnv* e, const double* xi);
void mjo_env_get_task(const mjoEnv* e, double* xi);
void mjo_env_set_state(mjoEnv* e, const double* qpos, const double* qvel);
/* step(a): reward/done/obs formulas of random_hopper.py:83-110 etc. Returns done. */
int mjo_env_step(mjoEnv* e, const double* action, double* obs, double* reward);
void mjo_env_obs(const mjoEnv* e, double* obs);

/* ---- batch helpers for ctypes (SoA in / SoA out, double) ------------------------------- */
/* one independent step from (qpos,qvel,action,xi) for n envs; arrays are [dim][n].
 * nthreads > 1 partitions envs over pthreads (used by bench.py's cpu_baseline). */
int mjo_batch_step(int kind, int variant, int n, const double* qpos, const double* qvel,
                   const double* action, const double* xi, double* qpos_out, double* qvel_out,
                   double* obs_out, double* reward_out, unsigned char* done_out, int nthreads);
/* rollout of `steps` env-steps per env with given per-step actions [steps][act][n] (no reset);
 * returns final state; used for CPU baseline timing. */
int mjo_batch_rollout(int kind, int variant, int n, int steps, const double* qpos, const double* qvel,
                      const double* actions, const double* xi, double* qpos_out, double* qvel_out,
                      double* reward_sum, int nthreads);
int mjo_batch_rollout_autoreset(int kind, int variant, int n, int steps, const double* qpos, const double* qvel,
                                const double* actions, const double* xi, const double* qpos_reset, const double* qvel_reset,
                                double* qpos_out, double* qvel_out, double* reward_sum, long long* resets_out, int nthreads);
/* forward-dynamics probe: returns qacc, qacc_smooth, M, bias, nefc, ncon for one state */
int mjo_probe_forward(int kind, const double* qpos, const double* qvel, const double* action,
                      const double* xi, double* qacc, double* qacc_smooth, double* qM,
                      double* qfrc_bias, int* ncon, int* nefc, double* efc_force, int* niter);
/* compiled model constants for tests / documentation */
int mjo_model_constants(int kind, const double* size, double* body_mass, double* body_inertia,
                        double* body_ipos, double* body_invweight0, double* dof_invweight0,
                        double* qpos0, int* dims);

/* parity tests tighten the Newton tolerance (default: the model's 1e-8) */
void mjo_set_tolerance(double tol);

/* Humanoid env step (random_humanoid.py:161-216). xipos_x_prev [14][n]: data.xipos[:,0] left by the previous
 * mj_forward (mass_center() reads it BEFORE do_simulation); a NaN entry means "state was just set"
 * (set_state runs sim.forward(), jinja_mujoco_env.py:146-154).  obs_out [376][n]. */
int mjo_humanoid_batch_step(int n, const double* qpos, const double* qvel, const double* action, const double* xi,
                            const double* xipos_x_prev, double* qpos_out, double* qvel_out, double* obs_out,
                            double* reward_out, unsigned char* done_out, double* xipos_x_out, int nthreads);

/* Humanoid reset / set_state observation (random_humanoid.py:219-234): forward at (qpos, qvel) with the task `xi`
 * in force (the OLD one under dr_training) and ctrl = 0.  obs_out [376][n], xipos_x_out [14][n] or NULL. */
int mjo_humanoid_batch_reset_obs(int n, const double* qpos, const double* qvel, const double* xi, double* obs_out,
                                 double* xipos_x_out);

/* CartPole closed-form step (random_envs/random_cartpole.py:172-224) */
void mjo_cartpole_step(const double* state, int action, const double* xi, double* next_state,
                       double* reward, int* done, int* steps_beyond_done);
int mjo_cartpole_batch_step(int n, const double* state, const int* action, const double* xi,
                            double* next_state, double* reward, unsigned char* done);

#ifdef __cplusplus
}
#endif
#endif
