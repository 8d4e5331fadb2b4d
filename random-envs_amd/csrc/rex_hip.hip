// rex_hip.hip -- HIP kernels (gfx950 / MI355X) + the C-ABI of include/rex.h.
//
// Execution model: ONE ENVIRONMENT PER LANE, 64-lane workgroups (one wavefront each) so a batch
// of B envs is B/64 independent waves spread over the 256 CUs.  State is SoA in HBM
// (qpos[nq][B], qvel[nv][B], xi[dim][B], ...): lane i touches element i of every row, so every
// global access of a wave is one contiguous 256-byte segment.  The whole per-env solve
// (composite-inertia M, L^T D L, pyramidal contact rows, Newton) lives in VGPRs
// (planar_engine.hpp); hopper / half-cheetah model constants arrive as kernel arguments
// (scalar registers), walker2d's per-env geometry as SoA rows.  No LDS, no MFMA: these are
// tiny per-instance solves, not dense contractions.
//
// reset()-time xi sampling and init-state noise use rocRAND's Philox4x32-10 device API with
// subsequence = GLOBAL env index (handle env_offset + lane) and offset = f(episode, t): results
// do not depend on how a batch is sharded over GPUs and no RNG state is stored.
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/rex.h"
#include "planar_model.hpp"
#include "humanoid_model.hpp"
#include "humanoid_pair.hpp"

using namespace rex;

// -DREX_ONLY_KIND=<rex_env_kind>: tuning builds that compile ONE chain's kernels (seconds instead of minutes);
// the other kinds then fail in rex_create with REX_ERR_UNSUPPORTED.  The product build defines nothing.
#ifdef REX_ONLY_KIND
#define REX_EN_CARTPOLE (REX_ONLY_KIND == 0)
#define REX_EN_HOPPER (REX_ONLY_KIND == 1)
#define REX_EN_HALFCHEETAH (REX_ONLY_KIND == 2)
#define REX_EN_WALKER2D (REX_ONLY_KIND == 3)
#define REX_EN_HUMANOID (REX_ONLY_KIND == 4)
#else
#define REX_EN_CARTPOLE 1
#define REX_EN_HOPPER 1
#define REX_EN_HALFCHEETAH 1
#define REX_EN_WALKER2D 1
#define REX_EN_HUMANOID 1
#endif

#if defined(REX_KTIME)
namespace rex { __device__ unsigned long long g_ktime[24 + 72]; }   // 0..7 planar phases, 8..23 humanoid phases, 24.. histogram of humanoid row counts
extern "C" int rex_debug_ktime(unsigned long long* out) {   // diagnostic build only (not in rex.h)
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rex::g_ktime), sizeof(unsigned long long) * 96) != hipSuccess) return -1;
  unsigned long long z[96] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(rex::g_ktime), z, sizeof z); return 0; }
#endif
#if defined(REX_WAVETIME)
// diagnostic build only: cycles every wave of the last planar / humanoid step launch spent in its substeps (the kernel time at
// B = 32 768 is the SLOWEST wave's, not the average)
namespace rex { __device__ unsigned long long g_wavetime[8192]; __device__ unsigned long long g_waveinfo[8192][8]; __device__ unsigned long long g_wavehum[1024][16];
                __device__ unsigned long long g_wavephase[8192][4];
                __device__ unsigned long long g_waveplace[8192][4];   // 100 MHz clock at entry and exit, HW_ID, XCC_ID: where and when each wave ran
                }
#endif
#if defined(REX_WAVETIME) || defined(REX_PHASES)
namespace rex { __device__ unsigned long long g_evalphase[8192][16]; }   // forward(): kinematics, mass+bias, detect, dispatch+self, rows+solve, pass 1, pass 2, H, ldl+solve, phi', update, correction
extern "C" int rex_debug_evalphase(unsigned long long* out, int n) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rex::g_evalphase), sizeof(unsigned long long) * 16 * (n < 8192 ? n : 8192)) != hipSuccess) return -1;
  static unsigned long long z[8192][16]; return hipMemcpyToSymbol(HIP_SYMBOL(rex::g_evalphase), z, sizeof z) == hipSuccess ? 0 : -1; }
#endif
#if defined(REX_WAVETIME)
// g_wavephase: planar step kernel, cycles entry -> state loaded -> substeps done -> outputs stored -> fused reset done
extern "C" int rex_debug_wavephase(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rex::g_wavephase), sizeof(unsigned long long) * 4 * (n < 8192 ? n : 8192)) == hipSuccess ? 0 : -1; }
extern "C" int rex_debug_waveplace(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rex::g_waveplace), sizeof(unsigned long long) * 4 * (n < 8192 ? n : 8192)) == hipSuccess ? 0 : -1; }
extern "C" int rex_debug_wavehum(unsigned long long* out) {   // humanoid: per-wave phase accumulators of the last launch (-DREX_KTIME -DREX_WAVETIME)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rex::g_wavehum), sizeof(unsigned long long) * 1024 * 16) == hipSuccess ? 0 : -1; }
extern "C" int rex_debug_waveinfo(unsigned long long* out, int n) {   // n waves x 8 counters, then zeroed
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rex::g_waveinfo), sizeof(unsigned long long) * 8 * (n < 8192 ? n : 8192)) != hipSuccess) return -1;
  static unsigned long long z[8192][8]; return hipMemcpyToSymbol(HIP_SYMBOL(rex::g_waveinfo), z, sizeof z) == hipSuccess ? 0 : -1; }
extern "C" int rex_debug_wavetime(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rex::g_wavetime), sizeof(unsigned long long) * (n < 8192 ? n : 8192)) == hipSuccess ? 0 : -1; }
#endif
#if defined(REX_KSTATS)
namespace rex { __device__ unsigned long long g_kstats[8]; }
extern "C" int rex_debug_kstats(unsigned long long* out) {   // diagnostic build only (not in rex.h)
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rex::g_kstats), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(rex::g_kstats), z, sizeof z); return 0; }
#endif

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int set_err(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
  return code;
}
#define HIP_TRY(x)                                                                            \
  do { hipError_t e_ = (x); if (e_ != hipSuccess) return set_err(REX_ERR_HIP, "%s failed: %s", #x, hipGetErrorString(e_)); } while (0)

extern "C" const char* rex_last_error(void) { return g_err; }
extern "C" const char* rex_version(void) { return "rex-hip 0.1 (gfx950)"; }

// First statement of every entry point that enqueues work, copies or synchronises on behalf of a handle: the handle's device becomes the
// calling thread's current device (one process may drive one handle per GPU: SURVEY.md 8(b) "Threading").  tests/test_abi.py parses this file
// and fails on an exported function that takes a handle and launches / copies without it.
#define REX_ENTER(h, fn)                                                    \
  do { if (!(h)) return set_err(REX_ERR_ARG, fn ": null handle"); HIP_TRY(hipSetDevice((h)->device)); } while (0)

// Environment knobs.  They select among the product's own launch shapes and solver schedules (every choice converges to the same
// minimiser) and exist for A/B measurements and for the parity tests that hold every shape to the oracle.  A stray variable
// must not change what a production process runs, so a knob is honoured only when REX_ALLOW_TUNING=1 is set beside it and rex_create
// REFUSES (REX_ERR_STATE) a handle when a knob is set without it.  Knobs that change the physics (no floor contacts, a PGS sweep cap)
// exist only in -DREX_TUNING builds, which build() never produces.
static const char* const kKnobs[] = {"REX_LANES", "REX_PAIR", "REX_ROLLED", "REX_HUM_PAIR", "REX_HUM_FUSED_RESET", "REX_FUSED_DERIVE",
                                     "REX_FAST", "REX_LS_MAX", "REX_LS_FREE", "REX_WARM", "REX_CORR",
                                     "REX_DIAG_NOCONTACT", "REX_HUM_ITERS"};
static bool tuning_allowed() { const char* e = getenv("REX_ALLOW_TUNING"); return e && atoi(e) == 1; }
static const char* stray_knob() {   // a knob set without REX_ALLOW_TUNING=1, or null
  if (tuning_allowed()) return nullptr;
  for (const char* k : kKnobs) if (getenv(k)) return k;
  return nullptr;
}
static const char* knob(const char* name) { return tuning_allowed() ? getenv(name) : nullptr; }

// ------------------------------------------------------------------------------------------
// DR distribution block (device-visible copy of RandomEnv's min/max/mean/stdev/cov state,
// random_env.py:102-127)
// ------------------------------------------------------------------------------------------
constexpr int MAX_XI = 32;
struct DRParams {
  int type;                 // rex_dr_type
  int dim;
  float a[MAX_XI];          // uniform: lo   | truncnorm/gaussian: mean | fullgaussian: mean (normalised space)
  float b[MAX_XI];          // uniform: hi   | truncnorm/gaussian: std
  float lower[MAX_XI];      // get_task_lower_bound(i)
  float lo[MAX_XI], hi[MAX_XI];   // fullgaussian: search bounds for denormalisation
  int map[MAX_XI];          // task index -> row of the kernels' full xi block (identity for the regular ids;
                            // the Unmodeled ids randomise a suffix only)
  const float* chol;        // fullgaussian: lower Cholesky factor of cov, row-major [dim][MAX_XI], DEVICE memory
};

// Philox offsets per episode: [0, 256) init-state noise, [256, 512) xi draws, STEP_BASE + t * STEP_STRIDE the
// observation noise of step t.  2^32 offsets per episode keep the step regions of consecutive episodes disjoint for
// 2^26 steps (time_limit off / endless episodes run far past 500 steps); the Philox counter is 64-bit + 64-bit subsequence.
constexpr unsigned long long EP_STRIDE = 1ull << 32;
constexpr unsigned long long STEP_BASE = 512, STEP_STRIDE = 64;
constexpr unsigned long long SAMPLE_SEED_SALT = 0x9E3779B97F4A7C15ull;   // rex_sample_task: a stream family of its own

// truncated standard normal on [-2, 2] by inverse CDF (the method scipy.stats.truncnorm.rvs uses)
__device__ __forceinline__ float truncnorm2(float u) {
  const float Fa = 0.022750131948179195f, Fb = 0.9772498680518208f;   // Phi(-2), Phi(2)
  float p = Fa + u * (Fb - Fa);
  float x = normcdfinvf(p);
  return fminf(fmaxf(x, -2.0f), 2.0f);
}

// RandomEnv.sample_task (random_env.py:148-203), one lane = one env.  Cold path (reset only): runtime dimension,
// rolled loops, every draw stored straight to its xi row (no per-lane array => the kernel needs no scratch).
__device__ void sample_task(const DRParams& dr, unsigned long long seed, unsigned long long subseq, unsigned long long offset,
                            float* __restrict__ xi_rows, size_t B, unsigned i, unsigned long long* counters) {
  const int d = dr.dim;
  rocrand_state_philox4x32_10 st;
  rocrand_init(seed, subseq, offset, &st);
  if (dr.type == REX_DR_UNIFORM) {           // :150-151  U(min, max) per dim
    for (int k = 0; k < d; k++) { float u = rocrand_uniform(&st); (xi_rows + (size_t)dr.map[k] * B)[i] = dr.a[k] + (dr.b[k] - dr.a[k]) * (1.0f - u); }
  } else if (dr.type == REX_DR_TRUNCNORM) {  // :153-171 (intended semantics; the reference raises NameError)
    for (int k = 0; k < d; k++) {
      float lb = dr.lower[k];
      float obs = dr.a[k] + dr.b[k] * truncnorm2(rocrand_uniform(&st));
      // `attempts` 1,2 keep a redraw; the third redraw is overwritten by lower_bound (:162-167)
      for (int att = 0; att < 2 && obs < lb; att++) obs = dr.a[k] + dr.b[k] * truncnorm2(rocrand_uniform(&st));
      if (obs < lb) obs = lb;
      (xi_rows + (size_t)dr.map[k] * B)[i] = obs;
    }
  } else if (dr.type == REX_DR_GAUSSIAN) {   // :173-190: redraw while < 0.1, raise after the 3rd failure
    for (int k = 0; k < d; k++) {
      float obs = dr.a[k] + dr.b[k] * rocrand_normal(&st);
      for (int att = 0; att < 2 && obs < 0.1f; att++) obs = dr.a[k] + dr.b[k] * rocrand_normal(&st);
      if (obs < 0.1f) { obs = 0.1f; atomicAdd(counters + 1, 1ull); }   // a device lane cannot raise: clamp + count
      (xi_rows + (size_t)dr.map[k] * B)[i] = obs;
    }
  } else if (dr.type == REX_DR_FULLGAUSSIAN) {  // :192-198: MVN in normalised [0,4]^d, clip, denormalise (:205-220)
    // x_k = mean_k + sum_{j<=k} L_kj z_j: the z stream is replayed from the counter for every k (no z[] array)
    for (int k = 0; k < d; k++) {
      rocrand_state_philox4x32_10 sz; rocrand_init(seed, subseq, offset, &sz);
      float acc = dr.a[k];
      for (int j = 0; j <= k; j++) acc += dr.chol[k * MAX_XI + j] * rocrand_normal(&sz);
      acc = fminf(fmaxf(acc, 0.0f), 4.0f);
      (xi_rows + (size_t)dr.map[k] * B)[i] = acc * (dr.hi[k] - dr.lo[k]) * 0.25f + dr.lo[k];
    }
  }
}

// ------------------------------------------------------------------------------------------
// device-side state of one handle
// ------------------------------------------------------------------------------------------
struct DevState {
  float* qpos; float* qvel; float* xi;     // SoA rows of length B
  float* geom;                             // walker2d: per-env PlanarGeom rows [NGEOMF][B]; else null
  float* aux;                              // humanoid: data.xipos[:,0] of the last forward, [14][B]; else null
  int* t; unsigned* episode; unsigned char* done;
  unsigned long long* counters;            // [4]
  long long B, env_offset;
  unsigned long long seed;
};

struct StepFlags {
  int endless, noisy, time_limit, max_steps;
  float noise_std;
  float* info;   // optional per-term reward rows [n_info][B] (random_half_cheetah.py:110, random_humanoid.py:182-187); null = off
  int readonly;  // rex_replay: state comes from the caller's buffers and nothing of the handle is written (no t / done / state stores, no reset)
};

// ------------------------------------------------------------------------------------------
// CartPole (random_envs/random_cartpole.py:172-229).  qpos = (x, theta), qvel = (x_dot, theta_dot).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) cartpole_step_kernel(DevState s, StepFlags fl, const int* __restrict__ action,
                                                           float* __restrict__ obs, float* __restrict__ reward,
                                                           unsigned char* __restrict__ done_out, unsigned char* __restrict__ trunc_out,
                                                           float* __restrict__ term_obs) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;   // 32-bit lane offset + uniform (SGPR) row bases
  if (i >= s.B) return;
  const long long B = s.B;
  float x = s.qpos[i], th = s.qpos[B + i], xd = s.qvel[i], thd = s.qvel[B + i];
  float g = s.xi[i], mc = s.xi[B + i], mp = s.xi[2 * B + i], l = s.xi[3 * B + i];
  float total = mp + mc;                                  // set_task :166
  const float pml = 0.1f * 0.5f;                          // :79, not refreshed by set_task
  float force = action[i] == 1 ? 10.0f : -10.0f;          // :80,178
  float st, ct; sincosf(th, &st, &ct);
  float temp = (force + pml * thd * thd * st) / total;    // :184
  float thacc = (g * st - ct * temp) / (l * (4.0f / 3.0f - mp * ct * ct / total));   // :185
  float xacc = temp - pml * thacc * ct / total;           // :186
  const float tau = 0.02f;
  x = x + tau * xd; xd = xd + tau * xacc; th = th + tau * thd; thd = thd + tau * thacc;   // :188-192
  s.qpos[i] = x; s.qpos[B + i] = th; s.qvel[i] = xd; s.qvel[B + i] = thd;
  const float th_thr = 12.0f * 2.0f * 3.14159265358979323846f / 360.0f, x_thr = 2.4f;   // :84-85
  bool was_done = s.done[i] != 0;                          // steps_beyond_done bookkeeping :208-222
  bool dn = (x < -x_thr) || (x > x_thr) || (th < -th_thr) || (th > th_thr);
  float r = (!dn) ? 1.0f : (was_done ? 0.0f : 1.0f);
  int t = s.t[i] + 1; s.t[i] = t;
  bool trunc = fl.time_limit && t >= fl.max_steps && !dn;
  bool d = dn || trunc;
  s.done[i] = (unsigned char)((dn || was_done) ? 1 : 0) | (unsigned char)(d ? 2 : 0);
  obs[i] = x; obs[B + i] = xd; obs[2 * B + i] = th; obs[3 * B + i] = thd;   // np.array(self.state) :224
  if (term_obs) { term_obs[i] = x; term_obs[B + i] = xd; term_obs[2 * B + i] = th; term_obs[3 * B + i] = thd; }
  reward[i] = r; done_out[i] = d ? 1 : 0;
  if (trunc_out) trunc_out[i] = trunc ? 1 : 0;
}

// reset(): state ~ U(-0.05, 0.05)^4 (random_cartpole.py:226-229). `resample` = set_random_task.
__global__ void __launch_bounds__(64) cartpole_reset_kernel(DevState s, DRParams dr, int resample, int reset_state,
                                                            const unsigned char* __restrict__ mask, int mask_bit,
                                                            float* __restrict__ obs) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;   // 32-bit lane offset + uniform (SGPR) row bases
  if (i >= s.B) return;
  if (mask && !(mask[i] & mask_bit)) return;
  const long long B = s.B;
  unsigned ep = s.episode[i] + 1; s.episode[i] = ep;
  rocrand_state_philox4x32_10 st;
  rocrand_init(s.seed, (unsigned long long)(s.env_offset + i), (unsigned long long)ep * EP_STRIDE, &st);
  if (reset_state) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = -0.05f + 0.1f * (1.0f - rocrand_uniform(&st));
    s.qpos[i] = v[0]; s.qvel[i] = v[1]; s.qpos[B + i] = v[2]; s.qvel[B + i] = v[3];
    s.t[i] = 0; s.done[i] = 0;
    if (obs) { obs[i] = v[0]; obs[B + i] = v[1]; obs[2 * B + i] = v[2]; obs[3 * B + i] = v[3]; }
  }
  if (resample && dr.type != REX_DR_NONE) {
    sample_task(dr, s.seed, (unsigned long long)(s.env_offset + i), (unsigned long long)ep * EP_STRIDE + 256, s.xi, (size_t)B, i, s.counters);
  }
}

// ------------------------------------------------------------------------------------------
// planar MuJoCo-style envs
// ------------------------------------------------------------------------------------------
template <class S> constexpr int geom_floats() { return sizeof(PlanarGeom<float, S>) / sizeof(float); }

template <class S>
__device__ __forceinline__ void load_geom(const DevState& s, unsigned i, const PlanarGeom<float, S>& uniform,
                                          PlanarGeom<float, S>& G) {
  if constexpr (S::KIND == 3) {   // walker2d: geometry is a function of the xi lengths (25 distinct values per env)
    walker_expand(uniform, [&](int k) { return (s.geom + (size_t)k * s.B)[i]; }, G);
  } else {
    G = uniform;
  }
}

// observation: concat(qpos[1:], qvel) (random_hopper.py:100-110, random_half_cheetah.py:112-121,
// random_walker2d.py:133-142) + optional N(0, noise_var)
template <class S>
__device__ __forceinline__ void write_obs(const float (&q)[S::NV], const float (&v)[S::NV], float* __restrict__ obs,
                                          long long B, unsigned i, bool noisy, float noise_std,
                                          rocrand_state_philox4x32_10* st) {
  static_for<0, S::NOBS>([&](auto KK) {
    constexpr int k = KK;
    float o = k < S::NV - 1 ? q[k + 1] : v[k - (S::NV - 1)];
    if (noisy) o += noise_std * rocrand_normal(st);
    (obs + (size_t)k * B)[i] = o;
  });
}

template <class S>
__device__ __forceinline__ void planar_reset_lane(const DevState& s, const StepFlags& fl, const DRParams& dr, int resample,
                                                  int reset_state, unsigned i, float* __restrict__ obs);
// walker2d: the per-env model constants of lane i from its xi lengths (what build_model() does inside
// RandomWalker2dEnv.set_task, random_walker2d.py:106-113)
__device__ __forceinline__ void walker_derive_lane(const DevState& s, unsigned i, int refresh_frozen_masses);
__device__ __attribute__((noinline)) void walker_derive_call(const DevState& s, unsigned i, int refresh_frozen_masses);
// `resample` argument of the step kernel: bit 0 = set_random_task at reset, bit 1 = walker2d: re-derive the lane's geometry from its
// new xi lengths right there (the auto-reset under DR used to cost a reset launch and a derive launch behind every step),
// bit 2 = the Unmodeled id's frozen masses follow the new lengths
constexpr int RS_RESAMPLE = 1, RS_DERIVE = 2, RS_REFRESH = 4;

// Register budget of the planar step kernel: waves per SIMD the allocator must leave room for (512 registers per lane and
// SIMD: 1 wave -> 512, 2 -> 256, 3 -> 168, 4 -> 128).  A lone wave issues one VALU instruction per 4 cycles, the SIMD one
// per 2: the step kernel is VALU-issue bound (PMC: 1.0 quad-cycle per VALU instruction), so two narrower co-resident waves
// beat one wide one as long as the live state fits.
#ifndef REX_STEP_WAVES
#define REX_STEP_WAVES 1
#endif
#define REX_STEP_OCC __attribute__((amdgpu_waves_per_eu(REX_STEP_WAVES, REX_STEP_WAVES)))

// PAIR: two lanes per environment (lane 2i and 2i + 1 both hold env i; planar_spec.hpp "two lanes per environment"):
// the launch has 2 B lanes in 64-lane blocks = 32 envs per wave, exactly the envs-per-wave of the 32-lane 1-lane-per-env
// launch, but the wave is full and the per-slot work of the feet-only solver is split over the two lanes.
// ROLLED: the general solver instantiation as runtime loops over a row list in scratch (planar_engine.hpp::solve_newton_rolled): the kernel
// then fits 256 registers and is built for TWO waves per SIMD -- hopper handles with more full one-lane-per-env waves than the GPU has SIMDs (rex_create).
template <class S, bool PAIR, bool ROLLED = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ROLLED ? 2 : REX_STEP_WAVES, ROLLED ? 2 : REX_STEP_WAVES)))
planar_step_kernel(DevState s, StepFlags fl, PlanarGeom<float, S> ugeom,
                                                         SolParams<float> sp, const float* __restrict__ action,
                                                         float* __restrict__ obs, float* __restrict__ reward,
                                                         unsigned char* __restrict__ done_out,
                                                         unsigned char* __restrict__ trunc_out, float* __restrict__ term_obs,
                                                         DRParams dr, int fused_reset, int resample) {
#if defined(REX_WAVETIME)
  const unsigned long long tp0 = __builtin_amdgcn_s_memtime();
  const unsigned long long tw0 = __builtin_amdgcn_s_memrealtime();
#endif
  // Narrow blocks (pair_lanes_for: 32 / 16 lanes) touch 64 / 32 bytes of every SoA row, so 2 / 4 neighbouring blocks share each 128-byte line.  Blocks are
  // dealt round-robin over the 8 XCDs (b and b + 8 share one: MI355X_MICROARCH.md, workgroup dispatch), each with its own L2: in launch order
  // the sharers sit on different XCDs and every line is fetched 2 / 4 times (C4: 4.4x the algorithmic bytes).  Transposed, block b works on the
  // env group (b % 8) * (blocks / 8) + b / 8: neighbours in memory are neighbours on one XCD.  The groups themselves -- which envs share a wave --
  // do not change, so neither does any result.
  unsigned blk = blockIdx.x;
  if constexpr (PAIR) { if (blockDim.x < 64u && (gridDim.x & 7u) == 0u) blk = (blk & 7u) * (gridDim.x >> 3) + (blk >> 3); }
  const unsigned i = (blk * blockDim.x + threadIdx.x) >> (PAIR ? 1 : 0);   // 32-bit lane offset + uniform (SGPR) row bases
  if (i >= s.B) return;   // (both lanes of a pair leave together: i is the same)
  const long long B = s.B;
  float q[S::NV], v[S::NV], ctrl[S::NU], xi[S::NXI];
  static_for<0, S::NV>([&](auto KK) { constexpr int k = KK; q[k] = (s.qpos + (size_t)k * B)[i]; v[k] = (s.qvel + (size_t)k * B)[i]; });
  static_for<0, S::NU>([&](auto KK) { constexpr int k = KK; ctrl[k] = (action + (size_t)k * B)[i]; });
  static_for<0, S::NXI>([&](auto KK) { constexpr int k = KK; xi[k] = (s.xi + (size_t)k * B)[i]; });
  PlanarGeom<float, S> G; load_geom<S>(s, i, ugeom, G);
  LaneParams<float, S> P; lane_params(S{}, xi, P);
  // the dynamics are invariant to the root x translation: integrate the step from x = 0 so the
  // forward-progress reward (posafter - posbefore)/dt keeps full fp32 resolution far from the origin
#if defined(REX_KTIME) || defined(REX_WAVETIME)
  unsigned long long tk0 = __builtin_amdgcn_s_memtime();
#endif
  const float x_before = q[0];
  q[0] = 0.0f;
  bool capped = false;
  float acc[S::NV];
  static_for<0, S::NV>([&](auto KK) { acc[KK] = 0.0f; });
  // which code the general solver modes run: the LIST solver in the two-lanes-per-env kernels (per-unit data in a column of LDS, one column per
  // lane: 9 - 20 KB per wave), the rolled row list in the hopper's two-waves-per-SIMD kernel, the unrolled per-slot instantiations otherwise
  constexpr int GEN = PAIR ? 2 : (ROLLED ? 1 : 0);
  float* slot_col = nullptr;
  if constexpr (GEN == 2) {
    __shared__ float slot_lds[SlotMem<float, S, PAIR>::WORDS];
    slot_col = slot_lds + threadIdx.x;
  }
#pragma unroll 1
  for (int f = 0; f < S::FRAME_SKIP; f++) capped |= substep<float, S, PAIR, GEN>(q, v, ctrl, G, P, sp, acc, f > 0, slot_col);   // do_simulation, jinja_mujoco_env.py:170-173
  if (PAIR && (threadIdx.x & 1u)) return;   // the even lane of a pair writes the results and runs the fused reset
  // the output addresses are formed from an opaque copy of the lane index: formed from `i`, the compiler computes all of them
  // next to the loads at the top, carries them through the solver, spills them and reloads each with a wait of its own
  unsigned io = i; asm volatile("" : "+v"(io));
#if defined(REX_KTIME)
  if ((threadIdx.x & 63) == 0) atomicAdd(&g_ktime[5], __builtin_amdgcn_s_memtime() - tk0);
#endif
#if defined(REX_WAVETIME)
  const unsigned long long tk1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) g_wavetime[blockIdx.x & 8191] = tk1 - tk0;
#endif
  const float dx = q[0];
  q[0] = x_before + dx;
  // reward / done
  float asq = 0.0f;
  static_for<0, S::NU>([&](auto KK) { asq += ctrl[KK] * ctrl[KK]; });
  const float dt = float(S::TIMESTEP * S::FRAME_SKIP);
  float r = dx / dt + S::ALIVE - S::CTRL_COST * asq;
  bool finite = true;
  static_for<0, S::NV>([&](auto KK) { constexpr int k = KK; finite = finite && isfinite(q[k]) && isfinite(v[k]); });
  bool dn = false;
  if constexpr (S::KIND == 1) {          // random_hopper.py:92
    bool small = true;
    static_for<2, S::NV>([&](auto KK) { constexpr int k = KK; small = small && fabsf(q[k]) < 100.0f; });
    static_for<0, S::NV>([&](auto KK) { constexpr int k = KK; small = small && fabsf(v[k]) < 100.0f; });
    dn = !(finite && small && q[1] > 0.7f && fabsf(q[2]) < 0.2f);
  } else if constexpr (S::KIND == 3) {   // random_walker2d.py:124-125
    dn = !(q[1] > 0.8f && q[1] < 2.0f && q[2] > -1.0f && q[2] < 1.0f);
  } else {                               // random_half_cheetah.py:108
    dn = false;
  }
  if (fl.endless) dn = false;            // random_hopper.py:95-96
  if (!fl.readonly) {   // (rex_replay: nothing of the handle is written, its counters included)
    if (!finite) atomicAdd(s.counters + 0, 1ull);
    if (capped && threadIdx.x == 0) atomicAdd(s.counters + 2, 1ull);
  }
  int t = s.t[io] + 1;
  bool trunc = fl.time_limit && t >= fl.max_steps && !dn && !fl.readonly;     // gym TimeLimit
  bool d = dn || trunc;
  if (!fl.readonly) {
    s.t[io] = t;
    static_for<0, S::NV>([&](auto KK) { constexpr int k = KK; (s.qpos + (size_t)k * B)[io] = q[k]; (s.qvel + (size_t)k * B)[io] = v[k]; });
    s.done[io] = d ? 2 : 0;
  }
  rocrand_state_philox4x32_10 st;
  if (fl.noisy) rocrand_init(s.seed, (unsigned long long)(s.env_offset + io),
                             (unsigned long long)s.episode[io] * EP_STRIDE + STEP_BASE + (unsigned long long)t * STEP_STRIDE, &st);
  write_obs<S>(q, v, obs, B, io, fl.noisy != 0, fl.noise_std, &st);
  if (term_obs) static_for<0, S::NOBS>([&](auto KK) { constexpr int k = KK; (term_obs + (size_t)k * B)[io] = (obs + (size_t)k * B)[io]; });
  reward[io] = r; done_out[io] = d ? 1 : 0;
  if (trunc_out) trunc_out[io] = trunc ? 1 : 0;
  if (fl.info) { fl.info[io] = dx / dt; (fl.info + (size_t)B)[io] = -S::CTRL_COST * asq; }   // info: reward_run, reward_ctrl (random_half_cheetah.py:105-110)
  // auto-reset fused into the step launch: finished lanes restart here (saves the masked reset launch and
  // the kernel boundary, ~10 % of a hopper step at B = 32768)
#if defined(REX_WAVETIME)
  const unsigned long long tr0 = __builtin_amdgcn_s_memtime();
#endif
  if (fused_reset && d) {
    planar_reset_lane<S>(s, fl, dr, resample & RS_RESAMPLE, 1, io, obs);
    if constexpr (S::KIND == 3) {
      if (resample & RS_DERIVE) {
        if constexpr (PAIR) walker_derive_lane(s, io, (resample & RS_REFRESH) ? 1 : 0);
        else walker_derive_call(s, io, (resample & RS_REFRESH) ? 1 : 0);   // (one lane per env: inlined it spills the step's own state; as a call only this branch pays)
      }
    }
  }
#if defined(REX_WAVETIME)
  if ((threadIdx.x & 63) == 0) { g_waveinfo[blockIdx.x & 8191][1] += __builtin_amdgcn_s_memtime() - tr0; }   // slot 1 ("iters", unused): cycles in the fused reset
  if ((threadIdx.x & 63) == 0) { unsigned long long* ph = g_wavephase[blockIdx.x & 8191]; ph[0] = tk0 - tp0; ph[1] = tk1 - tk0; ph[2] = tr0 - tk1; ph[3] = __builtin_amdgcn_s_memtime() - tr0; }
  if ((threadIdx.x & 63) == 0) { unsigned long long* pl = g_waveplace[blockIdx.x & 8191]; pl[0] = tw0; pl[1] = __builtin_amdgcn_s_memrealtime();
    pl[2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); pl[3] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20); }   // HW_REG_HW_ID, HW_REG_XCC_ID
#endif
}

// reset_model (random_hopper.py:112-120, random_half_cheetah.py:123-131, random_walker2d.py:144-153)
// + set_random_task (random_env.py:37-39) for one lane.
template <class S>
__device__ __forceinline__ void planar_reset_lane(const DevState& s, const StepFlags& fl, const DRParams& dr, int resample,
                                                  int reset_state, unsigned i, float* __restrict__ obs) {
  const long long B = s.B;
  unsigned ep = s.episode[i] + 1; s.episode[i] = ep;
#if defined(REX_DIAG_CHEAP_RESET)   // timing diagnostics only: what the RNG work of the fused reset costs the step kernel
  if (reset_state) {
    static_for<0, S::NV>([&](auto KK) { constexpr int k = KK; (s.qpos + (size_t)k * B)[i] = (k == 1 && S::KIND != 2) ? 1.25f : 0.0f; (s.qvel + (size_t)k * B)[i] = 0.0f; });
    s.t[i] = 0; s.done[i] = 0;
  }
  return;
#endif
  if (reset_state) {
    rocrand_state_philox4x32_10 st;
    rocrand_init(s.seed, (unsigned long long)(s.env_offset + i), (unsigned long long)ep * EP_STRIDE, &st);
    float q[S::NV], v[S::NV];
    const float c = S::INIT_NOISE;
    static_for<0, S::NV>([&](auto KK) { constexpr int k = KK;
      q[k] = c * (2.0f * (1.0f - rocrand_uniform(&st)) - 1.0f);            // init_qpos + U(-c, c)
      if constexpr (S::KIND == 2) v[k] = 0.1f * rocrand_normal(&st);       // random_half_cheetah.py:125
      else v[k] = c * (2.0f * (1.0f - rocrand_uniform(&st)) - 1.0f);
    });
    if constexpr (S::KIND != 2) q[1] += 1.25f;                             // init_qpos[1] = 1.25 (ref, hopper.xml:30)
    static_for<0, S::NV>([&](auto KK) { constexpr int k = KK; (s.qpos + (size_t)k * B)[i] = q[k]; (s.qvel + (size_t)k * B)[i] = v[k]; });
    s.t[i] = 0; s.done[i] = 0;
    if (obs) {
      rocrand_state_philox4x32_10 st2;
      if (fl.noisy) rocrand_init(s.seed, (unsigned long long)(s.env_offset + i), (unsigned long long)ep * EP_STRIDE + STEP_BASE, &st2);
      write_obs<S>(q, v, obs, B, i, fl.noisy != 0, fl.noise_std, &st2);
    }
  }
  if (resample && dr.type != REX_DR_NONE) {
    // separate stream region so the xi draw does not depend on reset_state
    sample_task(dr, s.seed, (unsigned long long)(s.env_offset + i), (unsigned long long)ep * EP_STRIDE + 256, s.xi, (size_t)B, i, s.counters);
  }
}

// pending_bit: walker2d's auto-reset under DR -- the lane's geometry has to follow its NEW xi lengths, which is the derive
// launch behind this one; the auto-reset mask is s.done itself and reset_lane clears it, so the reset leaves this bit for
// walker_derive_kernel to find (and clear).
template <class S>
__global__ void __launch_bounds__(64) planar_reset_kernel(DevState s, StepFlags fl, DRParams dr, int resample, int reset_state,
                                                          const unsigned char* __restrict__ mask, int mask_bit,
                                                          float* __restrict__ obs, int pending_bit) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.B) return;
  if (mask && !(mask[i] & mask_bit)) return;
  planar_reset_lane<S>(s, fl, dr, resample, reset_state, i, obs);
  if (pending_bit) s.done[i] = (unsigned char)pending_bit;
}

#if REX_EN_WALKER2D
// walker2d: re-derive the per-env model constants from the xi lengths for the masked lanes
// (replaces build_model() inside RandomWalker2dEnv.set_task, random_walker2d.py:106-113).
__global__ void __launch_bounds__(64) walker_derive_kernel(DevState s, const unsigned char* mask, int mask_bit,
                                                           int refresh_frozen_masses, int clear_pending) {
  using S = Walker2dSpec;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;   // 32-bit lane offset + uniform (SGPR) row bases
  if (i >= s.B) return;
  if (mask && !(mask[i] & mask_bit)) return;
  if (clear_pending) s.done[i] = 0;   // (mask is s.done: the pending bit planar_reset_kernel left)
  walker_derive_lane(s, i, refresh_frozen_masses);
}
__device__ __forceinline__ void walker_derive_lane(const DevState& s, unsigned i, int refresh_frozen_masses) {
  using S = Walker2dSpec;
  double size[4];
  for (int k = 0; k < 4; k++) size[k] = (double)s.xi[(long long)(7 + k) * s.B + i];
  PlanarGeom<double, S> G; SolParams<double> sp; double nominal[S::NB];
  derive_model<double, S>(size, G, nominal, sp);
  double c[kWalkerCompact];
  walker_compact_from_geom(G, c);
  for (int k = 0; k < kWalkerCompact; k++) (s.geom + (size_t)k * s.B)[i] = (float)c[k];
  // RandomWalker2dUnmodeled.set_task rebuilds the model and rewrites body_mass[4:] only, so the frozen
  // masses 1..3 become the geometry-derived ones of the new lengths (random_walker2d_unmodeled.py:109-116)
  if (refresh_frozen_masses) for (int b = 0; b < 3; b++) (s.xi + (size_t)b * s.B)[i] = (float)nominal[b];
}
__device__ __attribute__((noinline)) void walker_derive_call(const DevState& s, unsigned i, int refresh_frozen_masses) { walker_derive_lane(s, i, refresh_frozen_masses); }
#else
__device__ __forceinline__ void walker_derive_lane(const DevState&, unsigned, int) {}
__device__ __forceinline__ void walker_derive_call(const DevState&, unsigned, int) {}

#endif

template <class S>
__global__ void __launch_bounds__(64) planar_obs_kernel(DevState s, float* __restrict__ obs) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;   // 32-bit lane offset + uniform (SGPR) row bases
  if (i >= s.B) return;
  float q[S::NV], v[S::NV];
  static_for<0, S::NV>([&](auto KK) { constexpr int k = KK; q[k] = (s.qpos + (size_t)k * s.B)[i]; v[k] = (s.qvel + (size_t)k * s.B)[i]; });
  write_obs<S>(q, v, obs, s.B, i, false, 0.0f, nullptr);
}
__global__ void __launch_bounds__(64) cartpole_obs_kernel(DevState s, float* __restrict__ obs) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;   // 32-bit lane offset + uniform (SGPR) row bases
  if (i >= s.B) return;
  const long long B = s.B;
  obs[i] = s.qpos[i]; obs[B + i] = s.qvel[i]; obs[2 * B + i] = s.qpos[B + i]; obs[3 * B + i] = s.qvel[B + i];
}
__global__ void fill_rows_kernel(float* dst, const float* vals, int nrows, long long B) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  for (int k = 0; k < nrows; k++) dst[(long long)k * B + i] = vals[k];
}

// ------------------------------------------------------------------------------------------
// Humanoid (random_envs/jinja/random_humanoid.py).  One env per lane; the per-lane working set of a
// forward evaluation (hum::Scratch, ~20 KB: M 23x23, J and M^-1 J^T for up to 64 rows, contact list)
// lives in HIP scratch memory, lane-interleaved so every access of a wave is one coalesced segment.
// The compiled model is uniform and sits in __constant__ memory.
// ------------------------------------------------------------------------------------------
#if REX_EN_HUMANOID
__constant__ hum::Model<float> c_hum;

__device__ __forceinline__ void hum_lane(const DevState& s, unsigned i, hum::Lane<float>& L) {
  // set_task (random_humanoid.py:156-158): body_mass[1:] = xi[:13]; dof_damping[6:] = xi[13:]
  L.mass[0] = 0.0f;
  for (int k = 0; k < 13; k++) L.mass[1 + k] = (s.xi + (size_t)k * s.B)[i];
  for (int d = 0; d < 6; d++) L.damping[d] = 0.0f;
  for (int k = 0; k < 17; k++) L.damping[6 + k] = (s.xi + (size_t)(13 + k) * s.B)[i];
}

__global__ void __launch_bounds__(64) humanoid_step_kernel(DevState s, StepFlags fl, const float* __restrict__ action,
                                                           float* __restrict__ obs, float* __restrict__ reward,
                                                           unsigned char* __restrict__ done_out, unsigned char* __restrict__ trunc_out,
                                                           float* __restrict__ term_obs) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.B) return;
  const size_t B = (size_t)s.B;
  hum::Lane<float> L; hum_lane(s, i, L);
  float q[hum::NQ], v[hum::NV], a[hum::NU], xp[hum::NBODY];
  for (int k = 0; k < hum::NQ; k++) q[k] = (s.qpos + k * B)[i];
  for (int k = 0; k < hum::NV; k++) v[k] = (s.qvel + k * B)[i];
  for (int k = 0; k < hum::NU; k++) a[k] = (action + k * B)[i];
  for (int b = 0; b < hum::NBODY; b++) xp[b] = (s.aux + b * B)[i];
  hum::Kin<float> kn; hum::Scratch<float> sc;
#if defined(REX_KTIME)
  for (int k = 0; k < HT_SLOTS; k++) kn.tacc[k] = 0;
#endif
  float r; bool dn;
  rocrand_state_philox4x32_10 st;
  int t = s.t[i] + 1;
  if (fl.noisy) rocrand_init(s.seed, (unsigned long long)(s.env_offset + i),
                             (unsigned long long)s.episode[i] * EP_STRIDE + STEP_BASE + (unsigned long long)t * STEP_STRIDE, &st);
#if defined(REX_WAVETIME)
  const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
#endif
  float terms[4];
  hum::env_step(c_hum, L, q, v, a, xp, kn, sc, r, dn, [&](int k, float val) {
    // noise only on the qpos / qvel slices (random_humanoid.py:193-204)
    if (fl.noisy && k < 45) val += fl.noise_std * rocrand_normal(&st);
    (obs + k * B)[i] = val;
    if (term_obs) (term_obs + k * B)[i] = val;
  }, terms);
  if (fl.info) for (int k = 0; k < 4; k++) (fl.info + k * B)[i] = terms[k];   // reward_linvel, _quadctrl, _alive, _impact (random_humanoid.py:182-187)
#if defined(REX_WAVETIME)
  if ((threadIdx.x & 63) == 0) g_wavetime[blockIdx.x & 8191] = __builtin_amdgcn_s_memtime() - tk0;
#endif
#if defined(REX_KTIME)
  for (int k = 0; k < HT_SLOTS; k++) {   // one flush per wave and kernel: the wave maximum of every accumulator
    unsigned long long v = kn.tacc[k];
    for (int off = 32; off > 0; off >>= 1) { unsigned long long o = __shfl_xor(v, off); v = o > v ? o : v; }
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_ktime[8 + k], v);
#if defined(REX_WAVETIME)
    if ((threadIdx.x & 63) == 0 && k < 16) g_wavehum[blockIdx.x & 1023][k] = v;
#endif
  }
#endif
  bool finite = true;
  for (int k = 0; k < hum::NQ; k++) finite = finite && isfinite(q[k]);
  for (int k = 0; k < hum::NV; k++) finite = finite && isfinite(v[k]);
  if (!finite) dn = true;                                           // a diverged lane ends its episode
  if (fl.endless && finite) dn = false;
  bool trunc = fl.time_limit && t >= fl.max_steps && !dn && !fl.readonly;
  bool d = dn || trunc;
  if (!fl.readonly) {   // (rex_replay: nothing of the handle is written, its counters included)
    if (!finite) atomicAdd(s.counters + 0, 1ull);
    if (kn.overflow) atomicAdd(s.counters + 3, 1ull);
    s.t[i] = t;
    for (int k = 0; k < hum::NQ; k++) (s.qpos + k * B)[i] = q[k];
    for (int k = 0; k < hum::NV; k++) (s.qvel + k * B)[i] = v[k];
    for (int b = 0; b < hum::NBODY; b++) (s.aux + b * B)[i] = xp[b];
    s.done[i] = d ? 2 : 0;
  }
  reward[i] = r; done_out[i] = d ? 1 : 0;
  if (trunc_out) trunc_out[i] = trunc ? 1 : 0;
}

// ---- the step kernel over TWO LANES PER ENVIRONMENT (humanoid_pair.hpp): lanes 2e / 2e + 1 of a 64-lane block hold env e, the right
// lane the trunk + right leg / arm, the left lane the trunk (replicated) + left leg / arm; 32 envs per wave, every lane active.
struct DevPair {
#if defined(__HIP_DEVICE_COMPILE__)
  __device__ __forceinline__ int side() const { return (int)(threadIdx.x & 1u); }
  __device__ __forceinline__ float xchg(float x) const { return pair_xchg(x); }
  __device__ __forceinline__ unsigned xchg(unsigned x) const { return pair_xchg(x); }
  __device__ __forceinline__ bool any(bool b) const { return REX_WAVE_ANY(b); }
  __device__ __forceinline__ float* col() const { return hum::hum_lds + (threadIdx.x >> 1) * hum::pr::PAIR_WORDS; }
  // LDS hand-over between the two lanes of a pair: same wave, LDS operations of a wave execute in order, so only the COMPILER has
  // to be kept from moving a read of the partner's words above the partner's (= this instruction's) write
  __device__ __forceinline__ void sync() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#else   // (the host pass only parses the kernel body)
  __device__ int side() const { return 0; }
  __device__ float xchg(float x) const { return x; }
  __device__ unsigned xchg(unsigned x) const { return x; }
  __device__ bool any(bool b) const { return b; }
  __device__ float* col() const { return nullptr; }
  __device__ void sync() const {}
#endif
};

__global__ void __launch_bounds__(64) humanoid_pair_step_kernel(DevState s, StepFlags fl, const float* __restrict__ action,
                                                                float* __restrict__ obs, float* __restrict__ reward,
                                                                unsigned char* __restrict__ done_out, unsigned char* __restrict__ trunc_out,
                                                                float* __restrict__ term_obs, DRParams dr, int fused_reset, int resample) {
  namespace pr = hum::pr;
  const unsigned lane = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned i = lane >> 1;
  if (i >= s.B) return;   // (both lanes of a pair leave together)
  const bool left = (lane & 1u) != 0u;
  const size_t B = (size_t)s.B;
  const DevPair p;
  // set_task (random_humanoid.py:156-158): body_mass[1:] = xi[:13]; dof_damping[6:] = xi[13:] -- this lane's 8 bodies / 16 dofs
  pr::PLane<float> L;
  static_for<0, pr::LB>([&](auto BB) { constexpr int lb = BB; L.mass[lb] = (s.xi + (size_t)((left ? pr::gbL(lb) : pr::gbR(lb)) - 1) * B)[i]; });
  static_for<0, pr::LD>([&](auto DD) { constexpr int ld = DD;
    if constexpr (ld < 6) L.damping[ld] = 0.0f; else L.damping[ld] = (s.xi + (size_t)(13 + (left ? pr::gdL(ld) : pr::gdR(ld)) - 6) * B)[i]; });
  float ql[pr::LQ], vl[pr::LD], cl[pr::LU], xp[pr::LB];
  static_for<0, 7>([&](auto KK) { constexpr int k = KK; ql[k] = (s.qpos + (size_t)k * B)[i]; });
  static_for<6, pr::LD>([&](auto DD) { constexpr int ld = DD; ql[ld + 1] = (s.qpos + (size_t)((left ? pr::gdL(ld) : pr::gdR(ld)) + 1) * B)[i]; });
  static_for<0, pr::LD>([&](auto DD) { constexpr int ld = DD; vl[ld] = (s.qvel + (size_t)(left ? pr::gdL(ld) : pr::gdR(ld)) * B)[i]; });
  static_for<6, pr::LD>([&](auto DD) { constexpr int ld = DD;     // data.ctrl holds the raw action (:167); motor u drives dof kActDof[u]
    constexpr int uR = ld == 6 ? 1 : ld == 7 ? 0 : ld == 8 ? 2 : ld < 13 ? 3 + (ld - 9) : 11 + (ld - 13);
    constexpr int uL = ld < 9 ? uR : ld < 13 ? 7 + (ld - 9) : 14 + (ld - 13);
    cl[ld - 6] = (action + (size_t)(left ? uL : uR) * B)[i]; });
  static_for<0, pr::LB>([&](auto BB) { constexpr int lb = BB; xp[lb] = (s.aux + (size_t)(left ? pr::gbL(lb) : pr::gbR(lb)) * B)[i]; });
  float asq_side = 0.0f, asq = 0.0f;
  static_for<0, 3>([&](auto KK) { asq += cl[KK] * cl[KK]; });
  static_for<3, pr::LU>([&](auto KK) { asq_side += cl[KK] * cl[KK]; });
  asq += pr::psum(p, asq_side);
  pr::PKin<float> kn; pr::PScratch<float> sc; pr::PObs<float> park;
#if defined(REX_KTIME)
  for (int k = 0; k < HT_SLOTS; k++) kn.tacc[k] = 0;
#endif
  const int t = s.t[i] + 1;
#if defined(REX_WAVETIME)
  const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
#endif
  float r, terms[4]; bool dn;
  pr::env_step(p, c_hum, L, ql, vl, cl, asq, xp, kn, sc, park, r, dn, terms);
#if defined(REX_WAVETIME)
  if ((threadIdx.x & 63) == 0) g_wavetime[blockIdx.x & 8191] = __builtin_amdgcn_s_memtime() - tk0;
#endif
#if defined(REX_KTIME)
  for (int k = 0; k < HT_SLOTS; k++) {   // one flush per wave and kernel: the wave maximum of every accumulator
    unsigned long long v = kn.tacc[k];
    for (int off = 32; off > 0; off >>= 1) { unsigned long long o = __shfl_xor(v, off); v = o > v ? o : v; }
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_ktime[8 + k], v);
#if defined(REX_WAVETIME)
    if ((threadIdx.x & 63) == 0 && k < 16) g_wavehum[blockIdx.x & 1023][k] = v;
#endif
  }
#endif
  // observation (random_humanoid.py:193-204); noise only on the qpos / qvel slices: the 45 draws in row order, as one lane per env made them
  float nz[45];
  if (fl.noisy) {
    rocrand_state_philox4x32_10 st;
    rocrand_init(s.seed, (unsigned long long)(s.env_offset + i), (unsigned long long)s.episode[i] * EP_STRIDE + STEP_BASE + (unsigned long long)t * STEP_STRIDE, &st);
    for (int k = 0; k < 45; k++) nz[k] = fl.noise_std * rocrand_normal(&st);
  }
  pr::emit_obs(p, ql, vl, park, [&](auto RR, auto RL, float val) {
    constexpr int rr = RR, rl = RL;
    if constexpr (rr < 45 && rl < 45) { if (fl.noisy) val += left ? nz[rl] : nz[rr]; }
    const size_t row = left ? (size_t)rl : (size_t)rr;
    (obs + row * B)[i] = val;
    if (term_obs) (term_obs + row * B)[i] = val;
  });
  bool finite = true;
  static_for<0, pr::LQ>([&](auto KK) { finite = finite && isfinite(ql[KK]); });
  static_for<0, pr::LD>([&](auto KK) { finite = finite && isfinite(vl[KK]); });
  finite = finite && (p.xchg(finite ? 1u : 0u) != 0u);
  if (!finite) dn = true;                                           // a diverged lane ends its episode
  if (fl.endless && finite) dn = false;
  const bool trunc = fl.time_limit && t >= fl.max_steps && !dn && !fl.readonly;
  const bool d = dn || trunc;
  if (!fl.readonly) {   // (rex_replay: nothing of the handle is written, its counters included)
    static_for<9, pr::LD>([&](auto DD) { constexpr int ld = DD; const size_t g = left ? pr::gdL(ld) : pr::gdR(ld);
      (s.qpos + (g + 1) * B)[i] = ql[ld + 1]; (s.qvel + g * B)[i] = vl[ld]; });
    static_for<3, pr::LB>([&](auto BB) { constexpr int lb = BB; (s.aux + (size_t)(left ? pr::gbL(lb) : pr::gbR(lb)) * B)[i] = xp[lb]; });
  }
  if (!left) {
    if (!fl.readonly) {
      if (!finite) atomicAdd(s.counters + 0, 1ull);
      if (kn.overflow) atomicAdd(s.counters + 3, 1ull);
      s.t[i] = t;
      static_for<0, 10>([&](auto KK) { constexpr int k = KK; (s.qpos + (size_t)k * B)[i] = ql[k]; });
      static_for<0, 9>([&](auto KK) { constexpr int k = KK; (s.qvel + (size_t)k * B)[i] = vl[k]; });
      static_for<0, 3>([&](auto BB) { constexpr int lb = BB; (s.aux + (size_t)(lb + 1) * B)[i] = xp[lb]; });
      s.aux[i] = 0.0f;                                                // world body
      s.done[i] = d ? 2 : 0;
    }
    if (fl.info) for (int k = 0; k < 4; k++) (fl.info + k * B)[i] = terms[k];   // reward_linvel, _quadctrl, _alive, _impact (random_humanoid.py:182-187)
    reward[i] = r; done_out[i] = d ? 1 : 0;
    if (trunc_out) trunc_out[i] = trunc ? 1 : 0;
  }
  // Auto-reset fused into the step launch (the masked reset launch behind every step was 80 us of a 1.77 ms step): a finished env
  // restarts here, both lanes of its pair.  reset_model (random_humanoid.py:219-234) exactly as humanoid_reset_kernel does it -- the same
  // Philox streams and draw order (q 0..23, then v 0..22), set_state -> sim.forward() with the masses in force, THEN
  // set_random_task -- with the forward's kinematics / com / velocities over the pair's local trees.
  if (fused_reset && d) {
    const unsigned ep = s.episode[i] + 1;
    rocrand_state_philox4x32_10 st;
    rocrand_init(s.seed, (unsigned long long)(s.env_offset + i), (unsigned long long)ep * EP_STRIDE, &st);
    static_for<0, hum::NQ>([&](auto KK) { constexpr int k = KK;
      const float val = c_hum.qpos0[k] + 0.01f * (2.0f * (1.0f - rocrand_uniform(&st)) - 1.0f);
      if constexpr (k < 10) ql[k] = val;                              // free joint + the three abdomen hinges: replicated
      else static_for<9, pr::LD>([&](auto DD) { constexpr int ld = DD;
        if constexpr (pr::gdR(ld) + 1 == k) ql[ld + 1] = left ? ql[ld + 1] : val;
        if constexpr (pr::gdL(ld) + 1 == k) ql[ld + 1] = left ? val : ql[ld + 1]; }); });
    static_for<0, hum::NV>([&](auto KK) { constexpr int k = KK;
      const float val = 0.01f * (2.0f * (1.0f - rocrand_uniform(&st)) - 1.0f);
      if constexpr (k < 9) vl[k] = val;
      else static_for<9, pr::LD>([&](auto DD) { constexpr int ld = DD;
        if constexpr (pr::gdR(ld) == k) vl[ld] = left ? vl[ld] : val;
        if constexpr (pr::gdL(ld) == k) vl[ld] = left ? val : vl[ld]; }); });
    {
      pr::PSmooth<float> S;
      pr::kinematics(p, c_hum, ql, S);
      pr::com_pos(p, c_hum, L, S);
      float qb[pr::LD];
      pr::com_vel_rne(p, c_hum, vl, S, qb);
      static_for<0, pr::LB>([&](auto BB) { constexpr int b = BB; for (int k = 0; k < 10; k++) park.cinert[b][k] = S.cinert[b][k]; for (int k = 0; k < 6; k++) park.cvel[b][k] = S.cvel[b][k]; xp[b] = S.xipos[b][0]; });
      static_for<0, pr::LD>([&](auto II) { park.act[II] = 0.0f; });     // sim.reset() zeroes data.ctrl
    }
    if (fl.noisy) {
      rocrand_state_philox4x32_10 st2;
      rocrand_init(s.seed, (unsigned long long)(s.env_offset + i), (unsigned long long)ep * EP_STRIDE + STEP_BASE, &st2);
      for (int k = 0; k < 45; k++) nz[k] = fl.noise_std * rocrand_normal(&st2);
    }
    pr::emit_obs(p, ql, vl, park, [&](auto RR, auto RL, float val) {
      constexpr int rr = RR, rl = RL;
      if constexpr (rr < 45 && rl < 45) { if (fl.noisy) val += left ? nz[rl] : nz[rr]; }
      (obs + (left ? (size_t)rl : (size_t)rr) * B)[i] = val;
    });
    static_for<9, pr::LD>([&](auto DD) { constexpr int ld = DD; const size_t g = left ? pr::gdL(ld) : pr::gdR(ld);
      (s.qpos + (g + 1) * B)[i] = ql[ld + 1]; (s.qvel + g * B)[i] = vl[ld]; });
    static_for<3, pr::LB>([&](auto BB) { constexpr int lb = BB; (s.aux + (size_t)(left ? pr::gbL(lb) : pr::gbR(lb)) * B)[i] = xp[lb]; });
    if (!left) {
      s.episode[i] = ep;
      static_for<0, 10>([&](auto KK) { constexpr int k = KK; (s.qpos + (size_t)k * B)[i] = ql[k]; });
      static_for<0, 9>([&](auto KK) { constexpr int k = KK; (s.qvel + (size_t)k * B)[i] = vl[k]; });
      static_for<0, 3>([&](auto BB) { constexpr int lb = BB; (s.aux + (size_t)(lb + 1) * B)[i] = xp[lb]; });
      s.t[i] = 0; s.done[i] = 0;
      if (resample && dr.type != REX_DR_NONE)
        sample_task(dr, s.seed, (unsigned long long)(s.env_offset + i), (unsigned long long)ep * EP_STRIDE + 256, s.xi, B, i, s.counters);
    }
  }
}

// reset_model (random_humanoid.py:219-234): init noise U(-.01,.01) on all of qpos (incl. the quaternion) and qvel,
// set_state -> sim.forward() with the CURRENT task, THEN set_random_task (the cinert block of the
// returned observation is computed with the previous episode's masses).
__global__ void __launch_bounds__(64) humanoid_reset_kernel(DevState s, StepFlags fl, DRParams dr, int resample, int reset_state,
                                                            const unsigned char* __restrict__ mask, int mask_bit,
                                                            float* __restrict__ obs) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.B) return;
  if (mask && !(mask[i] & mask_bit)) return;
  const size_t B = (size_t)s.B;
  unsigned ep = s.episode[i] + 1; s.episode[i] = ep;
  rocrand_state_philox4x32_10 st;
  rocrand_init(s.seed, (unsigned long long)(s.env_offset + i), (unsigned long long)ep * EP_STRIDE, &st);
  if (reset_state) {
    float q[hum::NQ], v[hum::NV], xp[hum::NBODY];
    for (int k = 0; k < hum::NQ; k++) q[k] = c_hum.qpos0[k] + 0.01f * (2.0f * (1.0f - rocrand_uniform(&st)) - 1.0f);
    for (int k = 0; k < hum::NV; k++) v[k] = 0.01f * (2.0f * (1.0f - rocrand_uniform(&st)) - 1.0f);
    hum::Lane<float> L; hum_lane(s, i, L);
    hum::Kin<float> kn; hum::Scratch<float> sc;
    rocrand_state_philox4x32_10 st2;
    if (fl.noisy) rocrand_init(s.seed, (unsigned long long)(s.env_offset + i), (unsigned long long)ep * EP_STRIDE + STEP_BASE, &st2);
    hum::env_reset_obs(c_hum, L, q, v, xp, kn, sc, [&](int k, float val) {
      if (fl.noisy && k < 45) val += fl.noise_std * rocrand_normal(&st2);
      if (obs) (obs + k * B)[i] = val;
    });
    for (int k = 0; k < hum::NQ; k++) (s.qpos + k * B)[i] = q[k];
    for (int k = 0; k < hum::NV; k++) (s.qvel + k * B)[i] = v[k];
    for (int b = 0; b < hum::NBODY; b++) (s.aux + b * B)[i] = xp[b];
    s.t[i] = 0; s.done[i] = 0;
  }
  if (resample && dr.type != REX_DR_NONE) {
    sample_task(dr, s.seed, (unsigned long long)(s.env_offset + i), (unsigned long long)ep * EP_STRIDE + 256, s.xi, B, i, s.counters);
  }
}

// set_state / get_obs: sim.forward() at the stored state (jinja_mujoco_env.py:146-154)
__global__ void __launch_bounds__(64) humanoid_forward_kernel(DevState s, float* __restrict__ obs) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.B) return;
  const size_t B = (size_t)s.B;
  float q[hum::NQ], v[hum::NV], xp[hum::NBODY];
  for (int k = 0; k < hum::NQ; k++) q[k] = (s.qpos + k * B)[i];
  for (int k = 0; k < hum::NV; k++) v[k] = (s.qvel + k * B)[i];
  hum::Lane<float> L; hum_lane(s, i, L);
  hum::Kin<float> kn; hum::Scratch<float> sc;
  hum::env_reset_obs(c_hum, L, q, v, xp, kn, sc, [&](int k, float val) { if (obs) (obs + k * B)[i] = val; });
  for (int b = 0; b < hum::NBODY; b++) (s.aux + b * B)[i] = xp[b];
}

#endif  // REX_EN_HUMANOID

// ------------------------------------------------------------------------------------------
// host-side handle
// ------------------------------------------------------------------------------------------
struct rex_env {
  int kind = 0, variant = 0, device = 0;
  long long B = 0, env_offset = 0;
  unsigned long long seed = 0;
  rex_dims dims{};
  DevState dev{};
  DRParams dr{};
  StepFlags flags{};
  int dr_training = 0, autoreset = 1;
  int64_t step_count = 0;
  // host-derived constants
  PlanarGeom<float, HopperSpec> g_hopper{};
  PlanarGeom<float, HalfCheetahSpec> g_cheetah{};
  PlanarGeom<float, Walker2dSpec> g_walker{};
  SolParams<float> sp{};
  float nominal_xi[MAX_XI] = {0};   // FULL xi block of the kernels
  int full_dim = 0;                 // rows of the full xi block (dims.task_dim = rows exposed as the task)
  float* d_scratch = nullptr;   // MAX_XI floats
  float* d_chol = nullptr;      // MAX_XI*MAX_XI floats (fullgaussian Cholesky factor)
  int lanes = 32;               // lanes per workgroup of the one-lane-per-env launches, fixed at create time (lanes_for)
  int pair_lanes = 64;          // lanes per workgroup of the two-lanes-per-env planar step (pair_lanes_for): narrower waves while they all still get a SIMD
  int pair = 1;                 // planar chains: two lanes per env up to 32 envs x SIMDs, one lane per env past that (REX_PAIR overrides)
  int hum_pair = 1;             // humanoid step: two lanes per env (humanoid_pair_step_kernel; REX_HUM_PAIR=0: one env per lane)
  int rolled = 0;               // hopper, one lane per env: the 256-register step kernel (rolled general solver, two waves per SIMD); REX_ROLLED overrides
  int hum_fused_reset = 1;      // humanoid pair step: finished envs restart inside the step launch (REX_HUM_FUSED_RESET=0: the masked reset launch)
  int fused_derive = 1;         // walker2d: the auto-reset under DR re-derives the lane's geometry inside the step kernel (REX_FUSED_DERIVE=0: reset + derive launches)
  // timing: event pool created by rex_enable_timing, used as a ring by rex_step (no allocation in the step path)
  int timing = 0;
  std::vector<hipEvent_t> ev0, ev1;
  size_t ev_n = 0;              // launches recorded since the last enable / read
  unsigned long long launches = 0;   // rex_step calls since the last enable (sampling phase)
  // rex_replay scratch (allocated by the first replay that needs it): the full xi block the Unmodeled ids' reduced task is
  // scattered into, and walker2d's per-env geometry rows derived from the CALLER's xi lengths
  float* rp_xi = nullptr; float* rp_rows = nullptr; int* d_map = nullptr;   // rp_rows: walker2d geometry rows / humanoid xipos rows
};
constexpr size_t EV_POOL = 8192;

static int fill_dims(int kind, int variant, rex_dims* d) {
  memset(d, 0, sizeof *d);
  d->max_episode_steps = 500;                 // every gym.envs.register call, e.g. random_hopper.py:155-166
  int rc = -1;
  switch (kind) {
    case REX_CARTPOLE:    d->nq = 2; d->nv = 2; d->act_dim = 1; d->obs_dim = 4; d->task_dim = 4; d->frame_skip = 1; d->n_info = 0;
                          d->discrete_action = 1; d->dt = 0.02f; d->act_low = 0; d->act_high = 1; rc = 0; break;
    case REX_HOPPER:      d->nq = 6; d->nv = 6; d->act_dim = 3; d->obs_dim = 11; d->task_dim = 4; d->frame_skip = 4; d->n_info = 2;
                          d->dt = 0.008f; d->act_low = -1; d->act_high = 1; rc = 0; break;
    case REX_HALFCHEETAH: d->nq = 9; d->nv = 9; d->act_dim = 6; d->obs_dim = 17; d->task_dim = 8; d->frame_skip = 5; d->n_info = 2;
                          d->dt = 0.05f; d->act_low = -1; d->act_high = 1; rc = 0; break;
    case REX_WALKER2D:    d->nq = 9; d->nv = 9; d->act_dim = 6; d->obs_dim = 17; d->task_dim = 13; d->frame_skip = 4; d->n_info = 2;
                          d->dt = 0.008f; d->act_low = -1; d->act_high = 1; rc = 0; break;
    case REX_HUMANOID:    d->nq = 24; d->nv = 23; d->act_dim = 17; d->obs_dim = 376; d->task_dim = 30; d->frame_skip = 5; d->n_info = 4; d->n_aux = 14;
                          d->dt = 0.015f; d->act_low = -0.4f; d->act_high = 0.4f; rc = 0; break;   // humanoid.xml:6,9; random_humanoid.py:41
    default: return -1;
  }
  if (rc == 0 && variant) {
    if (variant != 1 || kind == REX_CARTPOLE) return -1;
    d->task_dim = kind == REX_HOPPER ? 3 : (kind == REX_HALFCHEETAH ? 5 : (kind == REX_WALKER2D ? 9 : 23));
  }
  return rc;
}
// Unmodeled ids: a prefix of xi is frozen and leaves the task vector
// (random_hopper_unmodeled.py:28-30, random_half_cheetah_unmodeled.py:33-36, random_walker2d_unmodeled.py:38-41)
static int variant_task_dim(int kind, int variant, int full) {
  if (!variant) return full;
  switch (kind) { case REX_HOPPER: return 3; case REX_HALFCHEETAH: return 5; case REX_WALKER2D: return 9; case REX_HUMANOID: return 23; default: return -1; }
}
static void variant_map(int kind, int variant, int full, int* map) {
  if (!variant) { for (int k = 0; k < full; k++) map[k] = k; return; }
  switch (kind) {
    case REX_HOPPER: for (int k = 0; k < 3; k++) map[k] = 1 + k; break;                 // thigh, leg, foot masses
    case REX_HALFCHEETAH: for (int k = 0; k < 5; k++) map[k] = 3 + k; break;            // bfoot..ffoot masses, friction
    case REX_WALKER2D: { const int m[9] = {3, 4, 5, 6, 8, 9, 10, 11, 12}; for (int k = 0; k < 9; k++) map[k] = m[k]; break; }
    case REX_HUMANOID:   // body_mass[5:] and dof_damping[9:] (random_humanoid_unmodeled.py:52-53,167-174)
      for (int k = 0; k < 9; k++) map[k] = 4 + k;
      for (int k = 0; k < 14; k++) map[9 + k] = 16 + k;
      break;
  }
}

extern "C" int rex_get_dims(int env_kind, int variant, rex_dims* out) {
  if (!out) return set_err(REX_ERR_ARG, "rex_get_dims: null out");
  if (fill_dims(env_kind, variant, out)) return set_err(REX_ERR_ARG, "unknown env kind %d / variant %d", env_kind, variant);
  return REX_OK;
}

template <class T, class S>
static void to_float_geom(const PlanarGeom<double, S>& g, PlanarGeom<float, S>& o) {
  const double* src = reinterpret_cast<const double*>(&g); float* dst = reinterpret_cast<float*>(&o);
  for (int k = 0; k < geom_floats<S>(); k++) dst[k] = (float)src[k];
}
static void sp_to_float(const SolParams<double>& a, SolParams<float>& b) {
  b.con_K = (float)a.con_K; b.con_B = (float)a.con_B; b.con_dmin = (float)a.con_dmin; b.con_dmax = (float)a.con_dmax;
  b.con_width = (float)a.con_width; b.con_margin = (float)a.con_margin; b.lim_K = (float)a.lim_K; b.lim_B = (float)a.lim_B;
  b.lim_dmin = (float)a.lim_dmin; b.lim_dmax = (float)a.lim_dmax; b.lim_width = (float)a.lim_width; b.meaninertia = (float)a.meaninertia;
  b.ls_max = a.ls_max; b.warm = a.warm; b.fast = a.fast; b.ls_free = a.ls_free; b.corr = a.corr;
}

template <class S>
static void host_derive(rex_env* h, PlanarGeom<float, S>& out, const double* size_override = nullptr) {
  PlanarGeom<double, S> G; SolParams<double> sp; double nominal[S::NB]; double size[8];
  for (int k = 0; k < S::NSIZE; k++) size[k] = size_override ? size_override[k] : S::default_size[k];
  derive_model<double, S>(size, G, nominal, sp);
  to_float_geom<double, S>(G, out); sp_to_float(sp, h->sp);
  for (int b = 0; b < S::NB; b++) h->nominal_xi[b] = (float)nominal[b];
}

// Launch shape by batch (rex_create; measured on MI355X, DESIGN.md section 6.1 "launch shape by batch").  The step kernels are latency-bound
// (one wave per SIMD, ~9 cycles per dependent VALU instruction against a 2-cycle issue), so as long as the GPU has a SIMD for every wave what
// matters is the time of ONE wave, and work is spread thin: two lanes per env (`pair`, 32 envs per wave) up to 32 envs x SIMDs (MI355X: 32 768 envs),
// 32-lane blocks for the one-lane-per-env kernels.  Past that a SIMD has several waves to run one after the other and what matters is the work
// per env: one lane per env in full 64-lane waves (the pair split costs 1.5-1.9x the instructions per env: cheetah 65 536 envs 309 -> 528 M
// env-steps/s, hopper 419 -> 646 M, walker 157 -> 215 M), and for the hopper past 64 envs x SIMDs the 256-register kernel whose waves share a
// SIMD two at a time (`rolled`: 2^20 envs 866 -> 1 422 M).  The humanoid stays on two lanes per env at every size (65 536 envs: 20.5 M against 13.4 M).
static int simds_of(int device_id) {
  int cus = 256;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || cus <= 0) cus = 256;
  return 4 * cus;
}
static int lanes_for(long long B, int simds) {
  const char* e = knob("REX_LANES");
  if (e && atoi(e) > 0) return atoi(e);
  return B > 32ll * simds ? 64 : 32;
}
// Two lanes per env: a step launch costs ONE wave's latency while every wave has a SIMD to itself, and a wave pays for the slowest of its envs in
// every Newton pass -- so a walker2d / half-cheetah batch that leaves SIMDs idle is spread over them in narrower waves (16 384 envs: 32-lane
// blocks = 16 envs per wave, 8 192: 16 lanes): fewer envs to wait for per pass (walker2d: 26.5 -> 21.0 passes per wave-step at 16 lanes).  Two
// limits, both measured (profiles/HISTORY.md, round 4; profiles/waveplace_probe.py):
//  - a CU with fewer than 64 ACTIVE lanes on it runs the same instruction stream slower (walker2d, cycles per Newton pass: 12.5 k with one 64-lane
//    wave on the CU, 13.1 k with four 16-lane waves, 14.2 k with one 32-lane wave, 18.8 k with two 16-lane waves -- same clock, same pass
//    counts, one wave per SIMD in every case), so below 8 envs per SIMD (8 192 envs) the blocks stay 64 lanes wide;
//  - the hopper's waves gain nothing from being narrow (its slowest wave is set by the feet-only passes every env runs): always 64 lanes.
static int pair_lanes_for(int kind, long long B, int simds) {
  const char* e = knob("REX_LANES");
  if (e && atoi(e) > 0) return atoi(e);
  if (kind == REX_HOPPER || B < 8ll * simds) return 64;
  int L = 64;
  while (L > 16 && (4 * B + L - 1) / L <= (long long)simds) L /= 2;   // halve while the halved blocks still number <= SIMDs
  return L;
}
// dynamic LDS of the humanoid kernels: one dual-PGS column (hum::DUAL_WORDS floats) per lane
// (read once in rex_create and cached in the handle: grid, block and dynamic-LDS size always agree)
static size_t hum_lds_bytes(const rex_env* h) { return sizeof(float) * hum::DUAL_WORDS * (size_t)h->lanes; }
static unsigned grid_for(const rex_env* h) { return (unsigned)((h->B + h->lanes - 1) / h->lanes); }
static unsigned lanes_of(const rex_env* h) { return (unsigned)h->lanes; }

constexpr int DERIVE_PENDING_BIT = 4;   // in DevState::done: reset under DR, geometry not yet re-derived (walker2d auto-reset)
static int launch_walker_derive(rex_env* h, const unsigned char* mask, int bit, hipStream_t st, int task_changed) {
#if REX_EN_WALKER2D
  const bool auto_mask = mask == h->dev.done;   // the auto-reset path: the reset launch replaced the done bit by the pending bit
  hipLaunchKernelGGL(walker_derive_kernel, dim3(grid_for(h)), dim3(lanes_of(h)), 0, st, h->dev, mask, auto_mask ? DERIVE_PENDING_BIT : bit,
                     (h->variant && task_changed) ? 1 : 0, auto_mask ? 1 : 0);
  HIP_TRY(hipGetLastError());
#endif
  return REX_OK;
}

// the humanoid's compiled model: built once per process (magic static: thread-safe), shared by every handle
#if REX_EN_HUMANOID
struct HumModels { hum::Model<double> md; hum::Model<float> mf; const char* err = nullptr; };
static const HumModels& hum_models() {
  static const HumModels* m = [] {
    HumModels* p = new HumModels();
    hum::build_model(p->md);
    if (!hum::check_topology(p->md)) p->err = "humanoid: compile-time dof tree differs from the model tables";
    else if (!hum::pr::check_pair_model(p->md)) p->err = "humanoid: a side body carries an orientation offset (humanoid_pair.hpp assumes none)";
    else hum::convert_model(p->md, p->mf);
    return p;
  }();
  return *m;
}
#endif

// everything of rex_create that can fail after the handle exists: on any error the caller destroys the handle, which frees
// whatever was allocated up to that point (every device pointer of a fresh rex_env is null)
static int create_body(rex_env* h, int env_kind, int variant, int64_t batch, int device_id, const rex_dims& dims, const rex_dims& full) {
  const int simds = simds_of(device_id);
  h->lanes = lanes_for(batch, simds);
  if (h->lanes < 8 || h->lanes > 64 || (h->lanes & (h->lanes - 1))) return set_err(REX_ERR_ARG, "REX_LANES must be 8, 16, 32 or 64 (got %d)", h->lanes);
  h->dims = dims;
  h->flags.endless = 0; h->flags.noisy = 0; h->flags.time_limit = 1; h->flags.max_steps = dims.max_episode_steps;
  h->flags.noise_std = 0.0f; h->flags.info = nullptr; h->flags.readonly = 0;
  h->full_dim = full.task_dim;
  h->dr.type = REX_DR_NONE; h->dr.dim = dims.task_dim;
  variant_map(env_kind, variant, full.task_dim, h->dr.map);
  const size_t B = (size_t)batch;
  DevState& d = h->dev;
  d.B = batch; d.env_offset = h->env_offset; d.seed = h->seed;
  HIP_TRY(hipMalloc(&d.qpos, sizeof(float) * dims.nq * B));
  HIP_TRY(hipMalloc(&d.qvel, sizeof(float) * dims.nv * B));
  HIP_TRY(hipMalloc(&d.xi, sizeof(float) * full.task_dim * B));
  HIP_TRY(hipMalloc(&d.t, sizeof(int) * B));
  HIP_TRY(hipMalloc(&d.episode, sizeof(unsigned) * B));
  HIP_TRY(hipMalloc(&d.done, B));
  HIP_TRY(hipMalloc(&d.counters, sizeof(unsigned long long) * 4));
  HIP_TRY(hipMalloc(&h->d_scratch, sizeof(float) * MAX_XI));
  HIP_TRY(hipMalloc(&h->d_chol, sizeof(float) * MAX_XI * MAX_XI));
  HIP_TRY(hipMemset(h->d_chol, 0, sizeof(float) * MAX_XI * MAX_XI));
  h->dr.chol = h->d_chol;
  HIP_TRY(hipMemset(d.qpos, 0, sizeof(float) * dims.nq * B));
  HIP_TRY(hipMemset(d.qvel, 0, sizeof(float) * dims.nv * B));
  HIP_TRY(hipMemset(d.t, 0, sizeof(int) * B));
  HIP_TRY(hipMemset(d.episode, 0, sizeof(unsigned) * B));
  HIP_TRY(hipMemset(d.done, 0, B));
  HIP_TRY(hipMemset(d.counters, 0, sizeof(unsigned long long) * 4));
  d.geom = nullptr; d.aux = nullptr;
  float noise_var = 0;
  switch (env_kind) {
    case REX_CARTPOLE: { const float t0[4] = {9.8f, 1.0f, 0.1f, 0.5f}; memcpy(h->nominal_xi, t0, sizeof t0); break; }   // random_cartpole.py:74-78
    case REX_HOPPER: host_derive<HopperSpec>(h, h->g_hopper); noise_var = HopperSpec::DEFAULT_NOISE_VAR;
                     if (variant) h->nominal_xi[0] *= 0.8f;                                                               // random_hopper_unmodeled.py:24-26
                     break;
    case REX_HALFCHEETAH: host_derive<HalfCheetahSpec>(h, h->g_cheetah); h->nominal_xi[7] = 0.4f;                        // random_half_cheetah.py:37
                          if (variant) for (int b = 0; b < 3; b++) h->nominal_xi[b] *= 0.8f;                             // random_half_cheetah_unmodeled.py:28-31
                          noise_var = HalfCheetahSpec::DEFAULT_NOISE_VAR; break;
    case REX_WALKER2D: {
      double wsize[4]; for (int k = 0; k < 4; k++) wsize[k] = Walker2dSpec::default_size[k];
      if (variant) wsize[0] *= 0.8;                                                                                      // random_walker2d_unmodeled.py:25-27
      host_derive<Walker2dSpec>(h, h->g_walker, wsize);
      if (variant) for (int b = 0; b < 3; b++) h->nominal_xi[b] *= 0.8f;                                                 // :33-36 (until the first set_task, Q6)
      for (int k = 0; k < 4; k++) h->nominal_xi[7 + k] = (float)wsize[k];                                                // random_walker2d.py:21
      h->nominal_xi[11] = 0.9f; h->nominal_xi[12] = 1.9f;                                                                // random_walker2d.py:37
      HIP_TRY(hipMalloc(&d.geom, sizeof(float) * kWalkerCompact * B));
      noise_var = Walker2dSpec::DEFAULT_NOISE_VAR; break; }
#if REX_EN_HUMANOID
    case REX_HUMANOID: {
      const HumModels& hm = hum_models();
      if (hm.err) return set_err(REX_ERR_ARG, "%s", hm.err);
      const hum::Model<double>& md = hm.md;
      { hum::Model<float> up = hm.mf;
#if defined(REX_TUNING)   // timing experiments only (changes the physics): a cap on the PGS sweeps
        if (knob("REX_HUM_ITERS")) up.iterations = atoi(knob("REX_HUM_ITERS"));
#endif
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_hum), &up, sizeof up)); }
      {   // 64-lane blocks need more than the default 64 KB of dynamic LDS
        const int lds = (int)(sizeof(float) * hum::DUAL_WORDS * 64);
        HIP_TRY(hipFuncSetAttribute((const void*)humanoid_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)humanoid_reset_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)humanoid_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      }
      for (int b = 0; b < 13; b++) h->nominal_xi[b] = (float)md.body_mass0[1 + b];          // random_humanoid.py:46
      for (int k = 0; k < 17; k++) h->nominal_xi[13 + k] = (float)md.dof_damping0[6 + k];   // :47
      if (variant) {   // random_humanoid_unmodeled.py:40-50: masses 1..4 and dampings 6..8 frozen at 0.8x
        for (int b = 0; b < 4; b++) h->nominal_xi[b] *= 0.8f;
        for (int k = 0; k < 3; k++) h->nominal_xi[13 + k] *= 0.8f;
      }
      HIP_TRY(hipMalloc(&d.aux, sizeof(float) * hum::NBODY * B));
      HIP_TRY(hipMemset(d.aux, 0, sizeof(float) * hum::NBODY * B));
      noise_var = 1e-3f;                                                                      // :39
      break; }
#endif
  }
  h->flags.noise_std = sqrtf(noise_var);
#if defined(REX_TUNING)   // timing diagnostics only (changes the physics): no floor contacts ever
  if (knob("REX_DIAG_NOCONTACT")) h->sp.con_margin = -1e9f;
#endif
  // solver schedule (every schedule reaches the same minimiser: tests/test_gpu_planar.py runs them all against the oracle)
  if (knob("REX_LS_MAX")) h->sp.ls_max = atoi(knob("REX_LS_MAX"));
  if (knob("REX_WARM")) h->sp.warm = atoi(knob("REX_WARM"));
  if (knob("REX_LS_FREE")) h->sp.ls_free = atoi(knob("REX_LS_FREE"));
  if (knob("REX_CORR")) h->sp.corr = atoi(knob("REX_CORR"));
  if (knob("REX_FAST")) h->sp.fast = atoi(knob("REX_FAST"));
  // launch shape by batch (lanes_for above has the measurements); rex_set_launch_shape overrides it per handle
  h->pair = batch <= 32ll * simds ? 1 : 0;
  h->pair_lanes = pair_lanes_for(env_kind, batch, simds);
  if (h->pair_lanes < 8 || h->pair_lanes > 64 || (h->pair_lanes & (h->pair_lanes - 1))) return set_err(REX_ERR_ARG, "REX_LANES must be 8, 16, 32 or 64 (got %d)", h->pair_lanes);
  h->rolled = (env_kind == REX_HOPPER && batch > 64ll * simds) ? 1 : 0;
  if (knob("REX_PAIR")) h->pair = atoi(knob("REX_PAIR")) ? 1 : 0;
  if (knob("REX_ROLLED")) h->rolled = (env_kind == REX_HOPPER && atoi(knob("REX_ROLLED"))) ? 1 : 0;
  if (knob("REX_HUM_PAIR")) h->hum_pair = atoi(knob("REX_HUM_PAIR")) ? 1 : 0;
  if (knob("REX_HUM_FUSED_RESET")) h->hum_fused_reset = atoi(knob("REX_HUM_FUSED_RESET")) ? 1 : 0;
  // walker2d: derive fused into the step kernel (inlined in the pair kernel, a call in the one-lane one) while the two small launches behind a
  // step are a visible share of it (32 768 envs: + 10 % env-steps/s, 65 536: + 9 %, 131 072: + 4.5 %, 2^20: - 0.5 %)
  h->fused_derive = batch < 524288 ? 1 : 0;
  if (knob("REX_FUSED_DERIVE")) h->fused_derive = atoi(knob("REX_FUSED_DERIVE")) ? 1 : 0;
  if (!h->sp.fast && !knob("REX_PAIR")) h->pair = 0;   // REX_FAST=0 is the strict-lane-independence mode: one lane per env unless REX_PAIR asks for the pair kernel
                                                        // (whose general path is the list solver: REX_FAST=0 REX_PAIR=1 runs it on every lane)
  if (h->pair) h->rolled = 0;     // the two-waves-per-SIMD kernel is a one-lane-per-env one
  // xi <- nominal task, state <- qpos0
  HIP_TRY(hipMemcpy(h->d_scratch, h->nominal_xi, sizeof(float) * full.task_dim, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, 0, d.xi, h->d_scratch, full.task_dim, (long long)B);
  HIP_TRY(hipGetLastError());
  if (env_kind == REX_WALKER2D) { int rc = launch_walker_derive(h, nullptr, 0, 0, 0); if (rc) return rc; }
#if REX_EN_HUMANOID
  if (env_kind == REX_HUMANOID) {
    float q0[MAX_XI] = {0}; q0[2] = 1.4f; q0[3] = 1.0f;                                       // humanoid.xml:30,32
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h->d_scratch, q0, sizeof(float) * dims.nq, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, 0, d.qpos, h->d_scratch, dims.nq, (long long)B);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(humanoid_forward_kernel, dim3(grid_for(h)), dim3(lanes_of(h)), hum_lds_bytes(h), 0, h->dev, (float*)nullptr);
    HIP_TRY(hipGetLastError());
  }
#endif
  if (env_kind == REX_HOPPER || env_kind == REX_WALKER2D) {
    float q0[MAX_XI] = {0}; q0[1] = 1.25f;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h->d_scratch, q0, sizeof(float) * dims.nq, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, 0, d.qpos, h->d_scratch, dims.nq, (long long)B);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipDeviceSynchronize());
  return REX_OK;
}

extern "C" int rex_destroy(rex_t* h);
extern "C" int rex_create(int env_kind, int variant, int64_t batch, int device_id, uint64_t seed, int64_t env_offset,
                          rex_t** out) {
  if (!out) return set_err(REX_ERR_ARG, "rex_create: null out");
  *out = nullptr;
  if (batch <= 0) return set_err(REX_ERR_ARG, "rex_create: batch must be > 0 (got %lld)", (long long)batch);
  rex_dims dims, full;
  if (fill_dims(env_kind, variant, &dims) || fill_dims(env_kind, 0, &full))
    return set_err(REX_ERR_ARG, "unknown env kind %d / variant %d", env_kind, variant);
#ifdef REX_ONLY_KIND
  if (env_kind != REX_ONLY_KIND) return set_err(REX_ERR_UNSUPPORTED, "this tuning build holds env kind %d only", (int)REX_ONLY_KIND);
#endif
  if (const char* k = stray_knob())
    return set_err(REX_ERR_STATE, "rex_create: %s is set but REX_ALLOW_TUNING=1 is not: tuning knobs are refused in production (unset it, or set REX_ALLOW_TUNING=1)", k);
#if !defined(REX_TUNING)
  if (getenv("REX_DIAG_NOCONTACT") || getenv("REX_HUM_ITERS"))
    return set_err(REX_ERR_UNSUPPORTED, "rex_create: REX_DIAG_NOCONTACT / REX_HUM_ITERS change the physics and exist in -DREX_TUNING builds only");
#endif
  HIP_TRY(hipSetDevice(device_id));
  rex_env* h = new (std::nothrow) rex_env();
  if (!h) return set_err(REX_ERR_ARG, "out of host memory");
  h->kind = env_kind; h->variant = variant; h->device = device_id; h->B = batch; h->env_offset = env_offset; h->seed = seed;
  const int rc = create_body(h, env_kind, variant, batch, device_id, dims, full);
  if (rc != REX_OK) {   // one cleanup for every failure path: the handle and whatever it had allocated (the message of the failure is kept)
    char keep[sizeof g_err]; memcpy(keep, g_err, sizeof keep);
    rex_destroy(h);
    memcpy(g_err, keep, sizeof keep);
    return rc;
  }
  *out = h;
  return REX_OK;
}

extern "C" int rex_destroy(rex_t* h) {
  if (!h) return REX_OK;
  hipSetDevice(h->device);
  hipDeviceSynchronize();
  hipFree(h->dev.qpos); hipFree(h->dev.qvel); hipFree(h->dev.xi); hipFree(h->dev.t); hipFree(h->dev.episode);
  hipFree(h->dev.done); hipFree(h->dev.counters); hipFree(h->d_scratch); hipFree(h->d_chol);
  if (h->dev.geom) hipFree(h->dev.geom);
  if (h->dev.aux) hipFree(h->dev.aux);
  if (h->rp_xi) hipFree(h->rp_xi);
  if (h->rp_rows) hipFree(h->rp_rows);
  if (h->d_map) hipFree(h->d_map);
  for (auto e : h->ev0) hipEventDestroy(e);
  for (auto e : h->ev1) hipEventDestroy(e);
  delete h;
  return REX_OK;
}

extern "C" int rex_set_dr(rex_t* h, int dr_type, const float* params, int n_params, const float* lower_bounds) {
  REX_ENTER(h, "rex_set_dr");
  const int d = h->dims.task_dim;
  DRParams& dr = h->dr;
  switch (dr_type) {
    case REX_DR_UNIFORM: case REX_DR_TRUNCNORM: case REX_DR_GAUSSIAN:
      if (!params || n_params != 2 * d) return set_err(REX_ERR_ARG, "set_dr: expected %d params, got %d", 2 * d, n_params);
      for (int i = 0; i < d; i++) { dr.a[i] = params[2 * i]; dr.b[i] = params[2 * i + 1]; }   // interleaved, random_env.py:102-121
      break;
    case REX_DR_FULLGAUSSIAN:
      if (!params || n_params != d + d * d + 2 * d) return set_err(REX_ERR_ARG, "set_dr(fullgaussian): expected %d params, got %d", 3 * d + d * d, n_params);
      for (int i = 0; i < d; i++) dr.a[i] = params[i];
      { static thread_local float hc[MAX_XI * MAX_XI]; memset(hc, 0, sizeof hc);
        for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) hc[i * MAX_XI + j] = params[d + i * d + j];
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(h->d_chol, hc, sizeof hc, hipMemcpyHostToDevice)); }
      for (int i = 0; i < d; i++) { dr.lo[i] = params[d + d * d + i]; dr.hi[i] = params[2 * d + d * d + i]; }
      break;
    case REX_DR_NONE: break;
    default: return set_err(REX_ERR_ARG, "Unknown dr_type:%d", dr_type);   // random_env.py:90
  }
  for (int i = 0; i < d; i++) dr.lower[i] = lower_bounds ? lower_bounds[i] : 0.0f;
  dr.type = dr_type; dr.dim = d;
  return REX_OK;
}
extern "C" int rex_set_dr_training(rex_t* h, int flag) { if (!h) return set_err(REX_ERR_ARG, "null handle"); h->dr_training = flag ? 1 : 0; return REX_OK; }
extern "C" int rex_set_flags(rex_t* h, int endless, int noisy, float noise_var) {
  if (!h) return set_err(REX_ERR_ARG, "null handle");
  h->flags.endless = endless ? 1 : 0; h->flags.noisy = noisy ? 1 : 0;
  if (noise_var >= 0) h->flags.noise_std = sqrtf(noise_var);
  return REX_OK;
}
extern "C" int rex_set_autoreset(rex_t* h, int autoreset, int time_limit) {
  if (!h) return set_err(REX_ERR_ARG, "null handle");
  h->autoreset = autoreset ? 1 : 0; h->flags.time_limit = time_limit ? 1 : 0;
  return REX_OK;
}
extern "C" int rex_seed(rex_t* h, uint64_t seed) { if (!h) return set_err(REX_ERR_ARG, "null handle"); h->seed = seed; h->dev.seed = seed; return REX_OK; }

static int do_reset(rex_t* h, const unsigned char* mask, int bit, int resample, int reset_state, float* obs, hipStream_t st) {
  const dim3 g(grid_for(h)), b(lanes_of(h));
  if (resample && h->dr.type == REX_DR_NONE) return set_err(REX_ERR_STATE,
      "sampling value of random env needs to be set before using sample_task() or set_random_task()");   // random_env.py:201
  switch (h->kind) {
#if REX_EN_CARTPOLE
    case REX_CARTPOLE: hipLaunchKernelGGL(cartpole_reset_kernel, g, b, 0, st, h->dev, h->dr, resample, reset_state, mask, bit, obs); break;
#endif
#if REX_EN_HOPPER
    case REX_HOPPER: hipLaunchKernelGGL(planar_reset_kernel<HopperSpec>, g, b, 0, st, h->dev, h->flags, h->dr, resample, reset_state, mask, bit, obs, 0); break;
#endif
#if REX_EN_HALFCHEETAH
    case REX_HALFCHEETAH: hipLaunchKernelGGL(planar_reset_kernel<HalfCheetahSpec>, g, b, 0, st, h->dev, h->flags, h->dr, resample, reset_state, mask, bit, obs, 0); break;
#endif
#if REX_EN_WALKER2D
    case REX_WALKER2D: hipLaunchKernelGGL(planar_reset_kernel<Walker2dSpec>, g, b, 0, st, h->dev, h->flags, h->dr, resample, reset_state, mask, bit, obs,
                                          (resample && mask == h->dev.done) ? DERIVE_PENDING_BIT : 0); break;
#endif
#if REX_EN_HUMANOID
    case REX_HUMANOID: hipLaunchKernelGGL(humanoid_reset_kernel, g, b, hum_lds_bytes(h), st, h->dev, h->flags, h->dr, resample, reset_state, mask, bit, obs); break;
#endif
  }
  HIP_TRY(hipGetLastError());
  if (h->kind == REX_WALKER2D && resample) return launch_walker_derive(h, mask, bit, st, 1);
  return REX_OK;
}

extern "C" int rex_reset(rex_t* h, const uint8_t* mask, float* obs_out, void* stream) {
  REX_ENTER(h, "rex_reset");
  // CartPole.reset() never resamples (random_cartpole.py:226-229); the MuJoCo envs do when dr_training
  int resample = (h->dr_training && h->kind != REX_CARTPOLE) ? 1 : 0;
  return do_reset(h, mask, 1, resample, 1, obs_out, (hipStream_t)stream);
}
extern "C" int rex_set_random_task(rex_t* h, const uint8_t* mask, void* stream) {
  REX_ENTER(h, "rex_set_random_task");
  return do_reset(h, mask, 1, 1, 0, nullptr, (hipStream_t)stream);
}

template <class S>
static void launch_planar_step(rex_env* h, const DevState& dev, const StepFlags& flags, const PlanarGeom<float, S>& geom, const float* action,
                               float* obs_out, float* reward_out, uint8_t* done_out, uint8_t* truncated_out, float* terminal_obs_out,
                               int fused, int resample, hipStream_t st) {
  if (h->pair) {   // 2 B lanes in blocks of pair_lanes: 32 envs per wave, fewer for a walker2d / half-cheetah batch of 8 192 .. 16 384 (pair_lanes_for)
    const unsigned L = (unsigned)h->pair_lanes, blocks = (unsigned)((2 * h->B + L - 1) / L);
    hipLaunchKernelGGL((planar_step_kernel<S, true>), dim3(blocks), dim3(L), 0, st, dev, flags, geom, h->sp, action, obs_out, reward_out,
                       done_out, truncated_out, terminal_obs_out, h->dr, fused, resample);
  } else {
    if constexpr (S::KIND == 1) {
      if (h->rolled) {   // two waves per SIMD (rex_create: more full waves than SIMDs)
        hipLaunchKernelGGL((planar_step_kernel<S, false, true>), dim3(grid_for(h)), dim3(lanes_of(h)), 0, st, dev, flags, geom, h->sp, action, obs_out,
                           reward_out, done_out, truncated_out, terminal_obs_out, h->dr, fused, resample);
        return;
      }
    }
    hipLaunchKernelGGL((planar_step_kernel<S, false>), dim3(grid_for(h)), dim3(lanes_of(h)), 0, st, dev, flags, geom, h->sp, action, obs_out,
                       reward_out, done_out, truncated_out, terminal_obs_out, h->dr, fused, resample);
  }
}

#if REX_EN_HUMANOID
static void launch_humanoid_step(rex_env* h, const DevState& dev, const StepFlags& flags, const float* action, float* obs_out, float* reward_out,
                                 uint8_t* done_out, uint8_t* truncated_out, float* terminal_obs_out, hipStream_t st, int fused = 0, int resample = 0) {
  if (h->hum_pair) {   // 2 B lanes in 64-lane blocks: 32 envs per wave, one LDS column per env
    const unsigned blocks = (unsigned)((2 * h->B + 63) / 64);
    hipLaunchKernelGGL(humanoid_pair_step_kernel, dim3(blocks), dim3(64), sizeof(float) * hum::pr::PAIR_WORDS * 32, st, dev, flags, action, obs_out,
                       reward_out, done_out, truncated_out, terminal_obs_out, h->dr, fused, resample);
  } else {
    hipLaunchKernelGGL(humanoid_step_kernel, dim3(grid_for(h)), dim3(lanes_of(h)), hum_lds_bytes(h), st, dev, flags, action, obs_out, reward_out,
                       done_out, truncated_out, terminal_obs_out);
  }
}
#endif

extern "C" int rex_step(rex_t* h, const void* action, float* obs_out, float* reward_out, uint8_t* done_out,
                        uint8_t* truncated_out, float* terminal_obs_out, void* stream) {
  REX_ENTER(h, "rex_step");
  if (!action || !obs_out || !reward_out || !done_out) return set_err(REX_ERR_ARG, "rex_step: null buffer");
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(grid_for(h)), b(lanes_of(h));
  const int resample_on_reset = (h->dr_training && h->kind != REX_CARTPOLE) ? 1 : 0;
  // planar envs reset finished lanes inside the step kernel (walker2d under DR re-derives the lane's geometry there as well)
  const bool walker_dr = h->kind == REX_WALKER2D && resample_on_reset && h->dr.type != REX_DR_NONE;
  const int fused = (h->autoreset && (h->kind == REX_HOPPER || h->kind == REX_HALFCHEETAH || (h->kind == REX_WALKER2D && (!walker_dr || h->fused_derive)) ||
                                      (h->kind == REX_HUMANOID && h->hum_pair && h->hum_fused_reset))) ? 1 : 0;
  int rs = resample_on_reset ? RS_RESAMPLE : 0;
  if (walker_dr && h->fused_derive) rs |= RS_DERIVE | (h->variant ? RS_REFRESH : 0);
  // every `timing`-th launch is bracketed by two events of the pool rex_enable_timing created (ring): the two event packets
  // cost ~8 us of stream time per launch, 9 % of a hopper step, so a throughput run samples (bench.py: every 8th launch)
  const bool timed = h->timing > 0 && (h->launches++ % (unsigned long long)h->timing) == 0;
  const size_t ev_slot = timed ? h->ev_n % h->ev0.size() : 0;   // (timing > 0 implies a complete pool)
  if (timed) HIP_TRY(hipEventRecord(h->ev0[ev_slot], st));
  switch (h->kind) {
#if REX_EN_CARTPOLE
    case REX_CARTPOLE:
      hipLaunchKernelGGL(cartpole_step_kernel, g, b, 0, st, h->dev, h->flags, (const int*)action, obs_out, reward_out, done_out, truncated_out, terminal_obs_out); break;
#endif
#if REX_EN_HOPPER
    case REX_HOPPER:
      launch_planar_step<HopperSpec>(h, h->dev, h->flags, h->g_hopper, (const float*)action, obs_out, reward_out, done_out, truncated_out, terminal_obs_out, fused, rs, st); break;
#endif
#if REX_EN_HALFCHEETAH
    case REX_HALFCHEETAH:
      launch_planar_step<HalfCheetahSpec>(h, h->dev, h->flags, h->g_cheetah, (const float*)action, obs_out, reward_out, done_out, truncated_out, terminal_obs_out, fused, rs, st); break;
#endif
#if REX_EN_WALKER2D
    case REX_WALKER2D:
      launch_planar_step<Walker2dSpec>(h, h->dev, h->flags, h->g_walker, (const float*)action, obs_out, reward_out, done_out, truncated_out, terminal_obs_out, fused, rs, st); break;
#endif
#if REX_EN_HUMANOID
    case REX_HUMANOID:
      launch_humanoid_step(h, h->dev, h->flags, (const float*)action, obs_out, reward_out, done_out, truncated_out, terminal_obs_out, st, fused, resample_on_reset); break;
#endif
  }
  if (timed) { HIP_TRY(hipEventRecord(h->ev1[ev_slot], st)); h->ev_n++; }
  HIP_TRY(hipGetLastError());
  h->step_count += h->B;
  if (h->autoreset && !fused) return do_reset(h, h->dev.done, 2, resample_on_reset, 1, obs_out, st);
  return REX_OK;
}

// Offline replay (random_hopper.py:128-152, random_half_cheetah.py:136-158, random_walker2d.py:161-185,
// random_humanoid.py:244-270: get_full_mjstate / set_sim_state + step): one env.step per lane from the CALLER's (qpos, qvel,
// xi, action) straight into the caller's outputs, and nothing of the handle changes -- its state, task, counters and RNG
// position stay where they were.  hopper / half-cheetah: ONE launch.  humanoid: the forward launch of set_state (data.xipos for
// mass_center(), jinja_mujoco_env.py:154) into replay scratch, then the step launch.  walker2d: the per-env geometry is a function of the xi
// lengths (its set_task rebuilds the model, random_walker2d.py:106-113), so the derive launch precedes the step launch, into
// replay scratch of the handle.  Unmodeled ids: `xi` is the reduced task; one scatter launch places it over the handle's
// frozen rows in a scratch copy of the full xi block first (random_hopper_unmodeled.py:71-76 ...).
__global__ void replay_xi_kernel(float* __restrict__ dst, const float* __restrict__ frozen, const float* __restrict__ task,
                                 const int* __restrict__ map, int full_dim, int task_dim, long long B) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  for (int k = 0; k < full_dim; k++) dst[(size_t)k * B + i] = frozen[(size_t)k * B + i];
  for (int k = 0; k < task_dim; k++) dst[(size_t)map[k] * B + i] = task[(size_t)k * B + i];
}

// rex_replay's scratch, complete or absent: the first replay of a handle allocates what its kind / id needs (the allocation
// synchronises the device; later calls only enqueue).  A partial failure frees what it got, so no later call can launch with a
// half-built set.  The scratch is per HANDLE: replays of one handle must be issued on one stream at a time (rex.h).
static int ensure_replay_scratch(rex_env* h) {
  const size_t B = (size_t)h->B;
  const bool need_xi = h->variant != 0, need_rows = h->kind == REX_WALKER2D || h->kind == REX_HUMANOID;
  const bool have_xi = h->rp_xi && h->d_map, have_rows = h->rp_rows != nullptr;
  if ((!need_xi || have_xi) && (!need_rows || have_rows)) return REX_OK;
  auto drop = [&] { if (h->rp_xi) hipFree(h->rp_xi); if (h->d_map) hipFree(h->d_map); if (h->rp_rows) hipFree(h->rp_rows);
                    h->rp_xi = nullptr; h->d_map = nullptr; h->rp_rows = nullptr; };
  hipError_t e = hipSuccess;
  if (need_xi && !have_xi) {
    if (e == hipSuccess && !h->rp_xi) e = hipMalloc(&h->rp_xi, sizeof(float) * h->full_dim * B);
    if (e == hipSuccess && !h->d_map) e = hipMalloc(&h->d_map, sizeof(int) * MAX_XI);
    if (e == hipSuccess) e = hipMemcpy(h->d_map, h->dr.map, sizeof(int) * MAX_XI, hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && need_rows && !have_rows) {
    const size_t rows = h->kind == REX_WALKER2D ? (size_t)kWalkerCompact : (size_t)hum::NBODY;
    e = hipMalloc(&h->rp_rows, sizeof(float) * rows * B);
  }
  if (e != hipSuccess) { drop(); return set_err(REX_ERR_HIP, "rex_replay: scratch allocation failed: %s", hipGetErrorString(e)); }
  return REX_OK;
}

extern "C" int rex_replay(rex_t* h, const float* qpos, const float* qvel, const float* xi, const float* action,
                          float* obs_out, float* reward_out, uint8_t* done_out, void* stream) {
  REX_ENTER(h, "rex_replay");
  if (!qpos || !qvel || !xi || !action || !obs_out || !reward_out || !done_out) return set_err(REX_ERR_ARG, "rex_replay: null argument");
  if (h->kind == REX_CARTPOLE) return set_err(REX_ERR_UNSUPPORTED, "rex_replay: RandomCartPoleEnv has no get_full_mjstate / set_sim_state (random_cartpole.py)");
  hipStream_t st = (hipStream_t)stream;
  const size_t B = (size_t)h->B;
  DevState dev = h->dev;                    // the handle's device view with the state rows replaced by the caller's buffers
  dev.qpos = const_cast<float*>(qpos); dev.qvel = const_cast<float*>(qvel); dev.xi = const_cast<float*>(xi);
  { int rc = ensure_replay_scratch(h); if (rc) return rc; }
  if (h->variant) {                          // reduced task -> scratch copy of the full xi block
    hipLaunchKernelGGL(replay_xi_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, h->rp_xi, h->dev.xi, xi, h->d_map,
                       h->full_dim, h->dims.task_dim, (long long)B);
    HIP_TRY(hipGetLastError());
    dev.xi = h->rp_xi;
  }
#if REX_EN_WALKER2D
  if (h->kind == REX_WALKER2D) {             // geometry of the caller's xi lengths (what set_task's build_model does)
    dev.geom = h->rp_rows;
    hipLaunchKernelGGL(walker_derive_kernel, dim3(grid_for(h)), dim3(lanes_of(h)), 0, st, dev, (const unsigned char*)nullptr, 0, h->variant ? 1 : 0, 0);
    HIP_TRY(hipGetLastError());
  }
#endif
  StepFlags flags = h->flags; flags.readonly = 1; flags.info = nullptr;
  const bool timed = h->timing > 0 && (h->launches++ % (unsigned long long)h->timing) == 0;   // same sampling as rex_step
  const size_t ev_slot = timed ? h->ev_n % h->ev0.size() : 0;
  if (timed) HIP_TRY(hipEventRecord(h->ev0[ev_slot], st));
  switch (h->kind) {
#if REX_EN_HOPPER
    case REX_HOPPER: launch_planar_step<HopperSpec>(h, dev, flags, h->g_hopper, action, obs_out, reward_out, done_out, nullptr, nullptr, 0, 0, st); break;
#endif
#if REX_EN_HALFCHEETAH
    case REX_HALFCHEETAH: launch_planar_step<HalfCheetahSpec>(h, dev, flags, h->g_cheetah, action, obs_out, reward_out, done_out, nullptr, nullptr, 0, 0, st); break;
#endif
#if REX_EN_WALKER2D
    case REX_WALKER2D: launch_planar_step<Walker2dSpec>(h, dev, flags, h->g_walker, action, obs_out, reward_out, done_out, nullptr, nullptr, 0, 0, st); break;
#endif
#if REX_EN_HUMANOID
    case REX_HUMANOID:   // set_state's sim.forward() (jinja_mujoco_env.py:154) leaves data.xipos for mass_center(): a forward launch into
                         // replay scratch, then the step launch
      dev.aux = h->rp_rows;
      hipLaunchKernelGGL(humanoid_forward_kernel, dim3(grid_for(h)), dim3(lanes_of(h)), hum_lds_bytes(h), st, dev, (float*)nullptr);
      HIP_TRY(hipGetLastError());
      launch_humanoid_step(h, dev, flags, action, obs_out, reward_out, done_out, nullptr, nullptr, st);
      break;
#endif
    default: break;
  }
  if (timed) { HIP_TRY(hipEventRecord(h->ev1[ev_slot], st)); h->ev_n++; }
  HIP_TRY(hipGetLastError());
  return REX_OK;
}

static int copy_rows(float* dst, const float* src, int rows, long long B, hipStream_t st) {
  HIP_TRY(hipMemcpyAsync(dst, src, sizeof(float) * rows * (size_t)B, hipMemcpyDeviceToDevice, st));
  return REX_OK;
}
extern "C" int rex_get_state(rex_t* h, float* qpos, float* qvel, void* stream) {
  REX_ENTER(h, "rex_get_state");
  if (!qpos || !qvel) return set_err(REX_ERR_ARG, "rex_get_state: null argument");
  int rc = copy_rows(qpos, h->dev.qpos, h->dims.nq, h->B, (hipStream_t)stream); if (rc) return rc;
  return copy_rows(qvel, h->dev.qvel, h->dims.nv, h->B, (hipStream_t)stream);
}
extern "C" int rex_set_state(rex_t* h, const float* qpos, const float* qvel, void* stream) {
  REX_ENTER(h, "rex_set_state");
  if (!qpos || !qvel) return set_err(REX_ERR_ARG, "rex_set_state: null argument");
  int rc = copy_rows(h->dev.qpos, qpos, h->dims.nq, h->B, (hipStream_t)stream); if (rc) return rc;
  rc = copy_rows(h->dev.qvel, qvel, h->dims.nv, h->B, (hipStream_t)stream); if (rc) return rc;
  HIP_TRY(hipMemsetAsync(h->dev.done, 0, (size_t)h->B, (hipStream_t)stream));   // steps_beyond_done = None
#if REX_EN_HUMANOID
  if (h->kind == REX_HUMANOID) {   // set_state runs sim.forward(): refreshes data.xipos (jinja_mujoco_env.py:154)
    hipLaunchKernelGGL(humanoid_forward_kernel, dim3(grid_for(h)), dim3(lanes_of(h)), hum_lds_bytes(h), (hipStream_t)stream, h->dev, (float*)nullptr);
    HIP_TRY(hipGetLastError());
  }
#endif
  return REX_OK;
}
extern "C" int rex_get_task(rex_t* h, float* xi, void* stream) {
  REX_ENTER(h, "rex_get_task");
  if (!xi) return set_err(REX_ERR_ARG, "rex_get_task: null argument");
  for (int k = 0; k < h->dims.task_dim; k++) {   // task row k = row map[k] of the full xi block
    int rc = copy_rows(xi + (size_t)k * h->B, h->dev.xi + (size_t)h->dr.map[k] * h->B, 1, h->B, (hipStream_t)stream); if (rc) return rc;
  }
  return REX_OK;
}
extern "C" int rex_set_task(rex_t* h, const float* xi, void* stream) {
  REX_ENTER(h, "rex_set_task");
  if (!xi) return set_err(REX_ERR_ARG, "rex_set_task: null argument");
  int rc = REX_OK;
  if (!h->variant) rc = copy_rows(h->dev.xi, xi, h->dims.task_dim, h->B, (hipStream_t)stream);   // identity map: one copy
  else for (int k = 0; k < h->dims.task_dim; k++) {
    rc = copy_rows(h->dev.xi + (size_t)h->dr.map[k] * h->B, xi + (size_t)k * h->B, 1, h->B, (hipStream_t)stream); if (rc) return rc;
  }
  if (rc) return rc;
  if (h->kind == REX_WALKER2D) return launch_walker_derive(h, nullptr, 0, (hipStream_t)stream, 1);
  return REX_OK;
}
extern "C" int rex_get_obs(rex_t* h, float* obs_out, void* stream) {
  REX_ENTER(h, "rex_get_obs");
  if (!obs_out) return set_err(REX_ERR_ARG, "rex_get_obs: null argument");
  const dim3 g(grid_for(h)), b(lanes_of(h)); hipStream_t st = (hipStream_t)stream;
  switch (h->kind) {
#if REX_EN_CARTPOLE
    case REX_CARTPOLE: hipLaunchKernelGGL(cartpole_obs_kernel, g, b, 0, st, h->dev, obs_out); break;
#endif
#if REX_EN_HOPPER
    case REX_HOPPER: hipLaunchKernelGGL(planar_obs_kernel<HopperSpec>, g, b, 0, st, h->dev, obs_out); break;
#endif
#if REX_EN_HALFCHEETAH
    case REX_HALFCHEETAH: hipLaunchKernelGGL(planar_obs_kernel<HalfCheetahSpec>, g, b, 0, st, h->dev, obs_out); break;
#endif
#if REX_EN_WALKER2D
    case REX_WALKER2D: hipLaunchKernelGGL(planar_obs_kernel<Walker2dSpec>, g, b, 0, st, h->dev, obs_out); break;
#endif
#if REX_EN_HUMANOID
    case REX_HUMANOID: hipLaunchKernelGGL(humanoid_forward_kernel, g, b, hum_lds_bytes(h), st, h->dev, obs_out); break;
#endif
  }
  HIP_TRY(hipGetLastError());
  return REX_OK;
}

extern "C" int64_t rex_step_count(const rex_t* h) { return h ? h->step_count : 0; }
extern "C" int rex_get_counters(rex_t* h, int64_t* out) {
  REX_ENTER(h, "rex_get_counters");
  if (!out) return set_err(REX_ERR_ARG, "rex_get_counters: null argument");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, h->dev.counters, sizeof(int64_t) * 4, hipMemcpyDeviceToHost));
  return REX_OK;
}
extern "C" int rex_get_launch_shape(const rex_t* h, int32_t* out) {
  if (!h || !out) return set_err(REX_ERR_ARG, "rex_get_launch_shape: null argument");
  const bool planar = h->kind == REX_HOPPER || h->kind == REX_HALFCHEETAH || h->kind == REX_WALKER2D;
  out[0] = (planar && h->pair) ? h->pair_lanes : h->lanes; out[1] = (planar && h->pair) ? 1 : 0; out[2] = (h->kind == REX_HOPPER && !h->pair && h->rolled) ? 1 : 0;
  out[3] = (h->kind == REX_HUMANOID && h->hum_pair) ? 1 : 0;
  return REX_OK;
}
extern "C" int rex_set_launch_shape(rex_t* h, const int32_t* shape) {
  if (!h || !shape) return set_err(REX_ERR_ARG, "rex_set_launch_shape: null argument");
  const bool planar = h->kind == REX_HOPPER || h->kind == REX_HALFCHEETAH || h->kind == REX_WALKER2D;
  int lanes = shape[0] < 0 ? ((planar && h->pair) ? h->pair_lanes : h->lanes) : shape[0];
  int pair = shape[1] < 0 ? h->pair : (shape[1] ? 1 : 0), rolled = shape[2] < 0 ? h->rolled : (shape[2] ? 1 : 0);
  int hum_pair = shape[3] < 0 ? h->hum_pair : (shape[3] ? 1 : 0);
  if (lanes < 8 || lanes > 64 || (lanes & (lanes - 1))) return set_err(REX_ERR_ARG, "rex_set_launch_shape: lanes must be 8, 16, 32 or 64 (got %d)", lanes);
  if (shape[1] > 0 && !planar) return set_err(REX_ERR_ARG, "rex_set_launch_shape: two lanes per env (pair) is a shape of the planar chains");
  if (shape[2] > 0 && h->kind != REX_HOPPER) return set_err(REX_ERR_ARG, "rex_set_launch_shape: the rolled kernel exists for the hopper only");
  if (shape[3] > 0 && h->kind != REX_HUMANOID) return set_err(REX_ERR_ARG, "rex_set_launch_shape: hum_pair is a shape of the humanoid");
  if (pair && rolled) return set_err(REX_ERR_ARG, "rex_set_launch_shape: the rolled kernel is a one-lane-per-env kernel (pair and rolled exclude each other)");
  if (shape[0] >= 0) { h->lanes = lanes; h->pair_lanes = lanes; }
  h->pair = pair; h->rolled = rolled; h->hum_pair = hum_pair;
  return REX_OK;
}
extern "C" int rex_enable_timing(rex_t* h, int enable) {
  REX_ENTER(h, "rex_enable_timing");
  if (enable && h->ev0.empty()) {   // the only place events are created: rex_step never allocates
    h->ev0.reserve(EV_POOL); h->ev1.reserve(EV_POOL);
    hipError_t err = hipSuccess;
    for (size_t k = 0; k < EV_POOL && err == hipSuccess; k++) {
      hipEvent_t a = nullptr, c = nullptr;
      err = hipEventCreate(&a);
      if (err == hipSuccess) { err = hipEventCreate(&c); if (err != hipSuccess) hipEventDestroy(a); }
      if (err == hipSuccess) { h->ev0.push_back(a); h->ev1.push_back(c); }
    }
    if (err != hipSuccess) {   // all or nothing: a partial pool would be indexed past its end by the ring
      for (auto e : h->ev0) hipEventDestroy(e);
      for (auto e : h->ev1) hipEventDestroy(e);
      h->ev0.clear(); h->ev1.clear(); h->timing = 0;
      return set_err(REX_ERR_HIP, "rex_enable_timing: hipEventCreate failed: %s", hipGetErrorString(err));
    }
  }
  h->timing = (enable > 0 && !h->ev0.empty()) ? enable : 0; h->ev_n = 0; h->launches = 0;
  return REX_OK;
}
extern "C" int rex_read_timing(rex_t* h, float* ms_out, int max_n) {
  if (!h || !ms_out) { set_err(REX_ERR_ARG, "rex_read_timing: null argument"); return REX_ERR_ARG; }
  if (hipSetDevice(h->device) != hipSuccess) return set_err(REX_ERR_HIP, "rex_read_timing: hipSetDevice(%d) failed", h->device);
  int n = 0;
  const size_t pool = h->ev0.size();
  const size_t first = h->ev_n > pool ? h->ev_n - pool : 0;   // the ring keeps the last `pool` launches
  for (size_t k = first; pool && k < h->ev_n && n < max_n; k++) {
    const size_t slot = k % pool;
    if (hipEventSynchronize(h->ev1[slot]) != hipSuccess) break;
    float ms = 0; if (hipEventElapsedTime(&ms, h->ev0[slot], h->ev1[slot]) != hipSuccess) break;
    ms_out[n++] = ms;
  }
  h->ev_n = 0;
  return n;
}

// ------------------------------------------------------------------------------------------
// episode bookkeeping, lane export, side-effect-free xi draws
// ------------------------------------------------------------------------------------------
extern "C" int rex_get_counters_state(rex_t* h, int32_t* t, uint32_t* episode, uint8_t* done, void* stream) {
  REX_ENTER(h, "rex_get_counters_state");
  if (!t || !episode || !done) return set_err(REX_ERR_ARG, "rex_get_counters_state: null argument");
  hipStream_t st = (hipStream_t)stream; const size_t B = (size_t)h->B;
  HIP_TRY(hipMemcpyAsync(t, h->dev.t, sizeof(int) * B, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(episode, h->dev.episode, sizeof(unsigned) * B, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(done, h->dev.done, B, hipMemcpyDeviceToDevice, st));
  return REX_OK;
}
extern "C" int rex_set_counters_state(rex_t* h, const int32_t* t, const uint32_t* episode, const uint8_t* done, void* stream) {
  REX_ENTER(h, "rex_set_counters_state");
  if (!t || !episode || !done) return set_err(REX_ERR_ARG, "rex_set_counters_state: null argument");
  hipStream_t st = (hipStream_t)stream; const size_t B = (size_t)h->B;
  HIP_TRY(hipMemcpyAsync(h->dev.t, t, sizeof(int) * B, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(h->dev.episode, episode, sizeof(unsigned) * B, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(h->dev.done, done, B, hipMemcpyDeviceToDevice, st));
  return REX_OK;
}
extern "C" int rex_get_aux(rex_t* h, float* aux, void* stream) {
  REX_ENTER(h, "rex_get_aux");
  if (!aux) return set_err(REX_ERR_ARG, "rex_get_aux: null argument");
  if (!h->dims.n_aux) return set_err(REX_ERR_UNSUPPORTED, "this env kind keeps no auxiliary sim data");
  return copy_rows(aux, h->dev.aux, h->dims.n_aux, h->B, (hipStream_t)stream);
}
extern "C" int rex_set_aux(rex_t* h, const float* aux, void* stream) {
  REX_ENTER(h, "rex_set_aux");
  if (!aux) return set_err(REX_ERR_ARG, "rex_set_aux: null argument");
  if (!h->dims.n_aux) return set_err(REX_ERR_UNSUPPORTED, "this env kind keeps no auxiliary sim data");
  return copy_rows(h->dev.aux, aux, h->dims.n_aux, h->B, (hipStream_t)stream);
}
extern "C" int rex_set_info_buffer(rex_t* h, float* info) {
  if (!h) return set_err(REX_ERR_ARG, "null handle");
  if (info && h->dims.n_info == 0) return set_err(REX_ERR_UNSUPPORTED, "this env kind has no per-term reward info");
  h->flags.info = info;
  return REX_OK;
}
extern "C" int rex_export_lane(rex_t* h, int64_t lane, float* qpos, float* qvel, float* xi) {
  REX_ENTER(h, "rex_export_lane");
  if (!qpos || !qvel || !xi) return set_err(REX_ERR_ARG, "rex_export_lane: null argument");
  if (lane < 0 || lane >= h->B) return set_err(REX_ERR_ARG, "rex_export_lane: lane %lld outside [0, %lld)", (long long)lane, h->B);
  HIP_TRY(hipDeviceSynchronize());
  const size_t B = (size_t)h->B;   // SoA rows: element `lane` of every row (a strided 2-D copy)
  HIP_TRY(hipMemcpy2D(qpos, sizeof(float), h->dev.qpos + lane, sizeof(float) * B, sizeof(float), h->dims.nq, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy2D(qvel, sizeof(float), h->dev.qvel + lane, sizeof(float) * B, sizeof(float), h->dims.nv, hipMemcpyDeviceToHost));
  for (int k = 0; k < h->dims.task_dim; k++)
    HIP_TRY(hipMemcpy(xi + k, h->dev.xi + (size_t)h->dr.map[k] * B + lane, sizeof(float), hipMemcpyDeviceToHost));
  return REX_OK;
}

// RandomEnv.sample_task (random_env.py:148-203) for every lane WITHOUT applying it (no episode bump, xi untouched):
// draw `draw_index` of a stream family of its own, into the caller's [task_dim][batch] buffer.
__global__ void __launch_bounds__(64) sample_task_kernel(DevState s, DRParams dr, unsigned long long draw_index, float* __restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.B) return;
  sample_task(dr, s.seed ^ SAMPLE_SEED_SALT, (unsigned long long)(s.env_offset + i), draw_index * 256ull, out, (size_t)s.B, i, s.counters);
}
extern "C" int rex_sample_task(rex_t* h, float* xi_out, uint64_t draw_index, void* stream) {
  REX_ENTER(h, "rex_sample_task");
  if (!xi_out) return set_err(REX_ERR_ARG, "rex_sample_task: null argument");
  if (h->dr.type == REX_DR_NONE) return set_err(REX_ERR_STATE,
      "sampling value of random env needs to be set before using sample_task() or set_random_task()");   // random_env.py:201
  DRParams dr = h->dr;
  for (int k = 0; k < dr.dim; k++) dr.map[k] = k;   // task order, not the kernels' full xi block
  hipLaunchKernelGGL(sample_task_kernel, dim3(grid_for(h)), dim3(lanes_of(h)), 0, (hipStream_t)stream, h->dev, dr, (unsigned long long)draw_index, xi_out);
  HIP_TRY(hipGetLastError());
  return REX_OK;
}
