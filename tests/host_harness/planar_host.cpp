// planar_host.cpp -- TEST HARNESS ONLY (never loaded by the product).
// Host instantiation of the kernels' math (random-envs_amd/csrc/planar_engine.hpp) in fp32 and
// fp64 so the CPU test-suite can compare the exact code the GPU runs against the independent
// 3-D oracle (oracle/mjo_core.c) without a GPU.  The product path has no CPU fallback: the
// Python package only ever calls librex_hip.so.
#include <cstring>
#include "../../random-envs_amd/csrc/planar_model.hpp"

using namespace rex;

static int g_fast = 1;   // SolParams::fast of every call below: 1 = allow the feet-only straight-line solver, 0 = general path only
static int g_ls_max = -1, g_ls_free = -1;   // overrides of SolParams::ls_max / ls_free (-1: the model's defaults)
extern "C" void ph_set_ls(int ls_max, int ls_free) { g_ls_max = ls_max; g_ls_free = ls_free; }
static int g_last_mode = -1;   // SolveStats::mode of the last ph_forward
extern "C" void ph_set_fast(int f) { g_fast = f; }
static int g_rolled = 0;   // the GEN flag of forward(): 0 = unrolled general instantiations, 1 = the rolled ROW-list solver (the 256-register hopper kernel,
                           // planar_engine.hpp::solve_newton_rolled), 2 = the LIST solver of the two-lanes-per-env kernels (solve_newton_list; one lane walks both ends here)
extern "C" void ph_set_rolled(int r) { g_rolled = r; }
extern "C" int ph_last_mode() { return g_last_mode; }

template <class T, class S>
static void run_step(int n, int nsub, const double* qpos, const double* qvel, const double* act, const double* xi,
                     const double* size, double* qpos_out, double* qvel_out, int* capped) {
  PlanarGeom<T, S> G; SolParams<T> sp; T nominal[S::NB]; T sz[8];
  for (int k = 0; k < S::NSIZE; k++) sz[k] = size ? T(size[k]) : T(S::default_size[k]);
  derive_model<T, S>(sz, G, nominal, sp);
  for (int i = 0; i < n; i++) {
    T q[S::NV], v[S::NV], c[S::NU], x[S::NXI];
    for (int k = 0; k < S::NV; k++) { q[k] = T(qpos[(size_t)k * n + i]); v[k] = T(qvel[(size_t)k * n + i]); }
    for (int k = 0; k < S::NU; k++) c[k] = T(act[(size_t)k * n + i]);
    for (int k = 0; k < S::NXI; k++) x[k] = T(xi[(size_t)k * n + i]);
    if (S::KIND == 3) {  // walker: geometry from the xi lengths
      T s4[4] = {x[7], x[8], x[9], x[10]};
      derive_model<T, S>(s4, G, nominal, sp);
    }
    sp.fast = g_fast; if (g_ls_max >= 0) sp.ls_max = g_ls_max; if (g_ls_free >= 0) sp.ls_free = g_ls_free;
    LaneParams<T, S> P; lane_params(S{}, x, P);
    bool cap = false; T acc[S::NV];
    for (int k = 0; k < S::NV; k++) acc[k] = T(0);
    for (int s = 0; s < nsub; s++) cap |= g_rolled == 2 ? substep<T, S, false, 2>(q, v, c, G, P, sp, acc, s > 0)
                                            : (g_rolled ? substep<T, S, false, 1>(q, v, c, G, P, sp, acc, s > 0) : substep<T, S>(q, v, c, G, P, sp, acc, s > 0));
    for (int k = 0; k < S::NV; k++) { qpos_out[(size_t)k * n + i] = double(q[k]); qvel_out[(size_t)k * n + i] = double(v[k]); }
    if (capped) capped[i] = cap;
  }
}

template <class T, class S>
static void run_forward(const double* qpos, const double* qvel, const double* act, const double* xi, const double* size,
                        double* qacc, double* Mout, int* iters) {
  PlanarGeom<T, S> G; SolParams<T> sp; T nominal[S::NB]; T sz[8];
  for (int k = 0; k < S::NSIZE; k++) sz[k] = size ? T(size[k]) : T(S::default_size[k]);
  T q[S::NV], v[S::NV], c[S::NU], x[S::NXI], a[S::NV], M[S::NV][S::NV];
  for (int k = 0; k < S::NV; k++) { q[k] = T(qpos[k]); v[k] = T(qvel[k]); }
  for (int k = 0; k < S::NU; k++) c[k] = T(act[k]);
  for (int k = 0; k < S::NXI; k++) x[k] = T(xi[k]);
  if (S::KIND == 3) { sz[0] = x[7]; sz[1] = x[8]; sz[2] = x[9]; sz[3] = x[10]; }
  derive_model<T, S>(sz, G, nominal, sp);
  sp.fast = g_fast; if (g_ls_max >= 0) sp.ls_max = g_ls_max; if (g_ls_free >= 0) sp.ls_free = g_ls_free;
  LaneParams<T, S> P; lane_params(S{}, x, P);
  for (int i = 0; i < S::NV; i++) for (int j = 0; j < S::NV; j++) M[i][j] = T(0);
  SolveStats st = g_rolled == 2 ? forward<T, S, false, 2>(q, v, c, G, P, sp, a, M) : (g_rolled ? forward<T, S, false, 1>(q, v, c, G, P, sp, a, M) : forward<T, S>(q, v, c, G, P, sp, a, M));
  for (int k = 0; k < S::NV; k++) qacc[k] = double(a[k]);
  for (int i = 0; i < S::NV; i++) for (int j = 0; j < S::NV; j++) Mout[i * S::NV + j] = double(j <= i ? M[i][j] : M[j][i]);
  *iters = st.iters; g_last_mode = st.mode;
}

template <class T, class S>
static void run_constants(const double* size, double* mass, double* iyy, double* tran, double* dofw, double* sol) {
  PlanarGeom<T, S> G; SolParams<T> sp; T nominal[S::NB]; T sz[8];
  for (int k = 0; k < S::NSIZE; k++) sz[k] = size ? T(size[k]) : T(S::default_size[k]);
  derive_model<T, S>(sz, G, nominal, sp);
  for (int b = 0; b < S::NB; b++) { mass[b] = double(nominal[b]); iyy[b] = double(G.iyy[b]); tran[b] = double(G.tran_invw[b]); dofw[b] = double(G.dof_invw[b]); }
  sol[0] = double(sp.con_K); sol[1] = double(sp.con_B); sol[2] = double(sp.lim_K); sol[3] = double(sp.lim_B); sol[4] = double(sp.meaninertia);
}

#define DISPATCH(FN, ...)                                                              \
  do {                                                                                 \
    if (kind == 1) { if (f32) FN<float, HopperSpec>(__VA_ARGS__); else FN<double, HopperSpec>(__VA_ARGS__); }            \
    else if (kind == 2) { if (f32) FN<float, HalfCheetahSpec>(__VA_ARGS__); else FN<double, HalfCheetahSpec>(__VA_ARGS__); } \
    else if (kind == 3) { if (f32) FN<float, Walker2dSpec>(__VA_ARGS__); else FN<double, Walker2dSpec>(__VA_ARGS__); }      \
    else return -1;                                                                    \
  } while (0)

extern "C" {
int ph_step(int kind, int f32, int n, int nsub, const double* qpos, const double* qvel, const double* act, const double* xi,
            const double* size, double* qpos_out, double* qvel_out, int* capped) {
  DISPATCH(run_step, n, nsub, qpos, qvel, act, xi, size, qpos_out, qvel_out, capped);
  return 0;
}
int ph_forward(int kind, int f32, const double* qpos, const double* qvel, const double* act, const double* xi,
               const double* size, double* qacc, double* M, int* iters) {
  DISPATCH(run_forward, qpos, qvel, act, xi, size, qacc, M, iters);
  return 0;
}
int ph_constants(int kind, int f32, const double* size, double* mass, double* iyy, double* tran, double* dofw, double* sol) {
  DISPATCH(run_constants, size, mass, iyy, tran, dofw, sol);
  return 0;
}
}

// walker compact-geometry table check: expand(compact(derive(size))) must reproduce derive(size) field by field
extern "C" int ph_walker_compact_check(const double* size, const double* nominal_size, double* max_err) {
  using S = Walker2dSpec;
  PlanarGeom<double, S> G, U, E; SolParams<double> sp; double nominal[S::NB];
  double s1[4] = {size[0], size[1], size[2], size[3]}, s0[4] = {nominal_size[0], nominal_size[1], nominal_size[2], nominal_size[3]};
  derive_model<double, S>(s1, G, nominal, sp); derive_model<double, S>(s0, U, nominal, sp);
  double c[kWalkerCompact]; walker_compact_from_geom(G, c);
  walker_expand(U, [&](int k) { return c[k]; }, E);
  const double* a = reinterpret_cast<const double*>(&G); const double* b = reinterpret_cast<const double*>(&E);
  double e = 0; int worst = -1;
  for (int f = 0; f < 105; f++) { double d = fabs(a[f] - b[f]); if (d > e) { e = d; worst = f; } }
  *max_err = e;
  return worst;
}

