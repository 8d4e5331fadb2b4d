// planar_spec.hpp -- compile-time descriptions of the three planar kinematic trees
// (hopper, walker2d, half-cheetah) and the run-time parameter block the kernels read.
//
// Every constant below is transcribed from the reference's MJCF templates
// (random_envs/jinja/assets/{hopper,walker2d,half_cheetah}.xml) and task files; the cited
// line is next to each value.  The trees are planar: slide-x, slide-z, hinge-y root followed by
// hinges about +/-y, so the kernels work with 2-D (x,z) vectors and one angle per body.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define REX_HD __host__ __device__ __forceinline__
#else
#define REX_HD inline
#endif

// true if the predicate holds on any active lane of the wave (device); the predicate itself on the host
#if defined(__HIP_DEVICE_COMPILE__) && !defined(REX_WAVE_ANY_OFF)
#define REX_WAVE_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0ull)
#else
#define REX_WAVE_ANY(x) (x)
#endif

namespace rex {

// sin and cos of one fp32 angle: three-term Cody-Waite reduction to [-pi/4, pi/4] by quadrants + the cephes minimax
// polynomials.  Worst error measured over |x| <= 400: 2.2 * 2^-24 * max(|f|, 0.25) (libm sinf: 1.1); about 25 instructions and
// no large-argument branch, against ~60 for the device library's sincosf (the angles here are joint angles).
REX_HD void sincos_poly(float x, float& s, float& c) {
  const float j = rintf(x * 0.636619772367581343f);   // x * 2 / pi
  float r = fmaf(-j, 1.5703125f, x);
  r = fmaf(-j, 4.837512969970703125e-4f, r);
  r = fmaf(-j, 7.54978995489188e-8f, r);
  const float r2 = r * r;
  const float ps = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
  const float sn = fmaf(ps * r2, r, r);
  const float pc = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
  const float cs = fmaf(pc * r2, r2, fmaf(-0.5f, r2, 1.0f));
  const int q = (int)j & 3;
  const float a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;
  s = (q & 2) ? -a : a;
  c = ((q + 1) & 2) ? -b : b;
}

// fast reciprocal / division: v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE sequence.
// The parity tolerance (1e-4 relative on qvel) is four orders above its error.
REX_HD float rcp_t(float a) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(a);
#else
  return 1.0f / a;
#endif
}
REX_HD double rcp_t(double a) { return 1.0 / a; }

// Optimisation barrier: tells the compiler the value may have changed.  Used at the top of the
// solver loops so that LLVM's loop-invariant code motion does not hoist every lever arm and
// Jacobian entry of every contact slot out of the loops (it does, speculatively, and the ~200
// hoisted values then spill to scratch memory: 590 MB of HBM writes per launch were measured).
template <class T> REX_HD void opaque(T& x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x));
#else
  (void)x;
#endif
}


// ---- two lanes per environment (the PAIR instantiations of the planar step; device only) ------------------------------
// Lanes 2i and 2i + 1 hold the SAME environment: everything is replicated bit for bit except the work that splits cleanly
// by capsule END (lane parity = end): the floor tests and, in the feet-only solver, the whole per-slot part of both passes.
// Partial results are exchanged with DPP quad permutes (lane ^ 1 inside a quad: no LDS, one VALU instruction, usually fused
// into the consuming add); sums are commutative, so both lanes of a pair hold identical bits afterwards.
#if defined(__HIP_DEVICE_COMPILE__)
REX_HD unsigned pair_parity() { return threadIdx.x & 1u; }
REX_HD float pair_xchg(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true)); }
REX_HD unsigned pair_xchg(unsigned x) { return (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true); }
#else
inline unsigned pair_parity() { return 0u; }
template <class T> inline T pair_xchg(T x) { return x; }   // PAIR is never instantiated on the host
#endif

template <int N>
struct IC { static constexpr int value = N; constexpr operator int() const { return N; } };

template <int B, int E, class F>
REX_HD void static_for(F&& f) {
  if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
}
// descending: E-1 ... B
template <int B, int E, class F>
REX_HD void static_rfor(F&& f) {
  if constexpr (B < E) { f(IC<E - 1>{}); static_rfor<B, E - 1>(f); }
}

constexpr double kDeg = 3.14159265358979323846 / 180.0;

// ---------------------------------------------------------------------------------------
// Per-tree compile-time spec.
//   body i (0 = torso/root) has hinge dof i+2; dofs 0,1 are root slide-x / slide-z.
//   parent[i]: parent body (-1 for the root);  sgn[i]: hinge axis is sgn*y.
//   geoms are capsules; geom_body[g] is the body they belong to.
// ---------------------------------------------------------------------------------------
struct HopperSpec {
  static constexpr int KIND = 1;
  static constexpr int NB = 4, NV = 6, NU = 3, NG = 4, NXI = 4, NOBS = 11, NSIZE = 4;
  static constexpr int parent[NB] = {-1, 0, 1, 2};
  static constexpr int sgn[NB] = {+1, -1, -1, -1};                 // hopper.xml:31,34,37,40
  static constexpr int geom_body[NG] = {0, 1, 2, 3};
  static constexpr bool RK4 = true;                                 // hopper.xml:17
  static constexpr int FRAME_SKIP = 4;                              // random_hopper.py:30
  static constexpr double TIMESTEP = 0.002;                         // hopper.xml:17
  static constexpr double GRAVITY = 9.81;                           // MuJoCo default
  static constexpr double Z_REF = 0.0;   // qpos[1] IS the root z (ref="1.25", hopper.xml:30)
  // self-collision pairs (contype=conaffinity=1, hopper.xml:5; parent-child pairs filtered)
  static constexpr int NSELF = 3;
  static constexpr int self_a[3] = {0, 0, 1};
  static constexpr int self_b[3] = {2, 3, 3};
  static constexpr float gear[NU] = {200.f, 200.f, 200.f};          // hopper.xml:48-50
  static constexpr bool limited[NB] = {false, true, true, true};    // hopper.xml:4,29-31
  static constexpr double range_lo[NB] = {0, -150 * kDeg, -150 * kDeg, -45 * kDeg}; // :34,37,40
  static constexpr double range_hi[NB] = {0, 0, 0, 45 * kDeg};
  static constexpr double default_size[4] = {.4, .45, .5, .39};  // random_hopper.py:18
  // reward / done (random_hopper.py:83-98)
  static constexpr float ALIVE = 1.0f, CTRL_COST = 1e-3f;
  static constexpr float DEFAULT_NOISE_VAR = 1e-4f;                 // random_hopper.py:28
  static constexpr float INIT_NOISE = 0.005f;                       // random_hopper.py:113-114
  static constexpr bool DR_BEFORE_STATE = false;                    // random_hopper.py:113-118
  static constexpr unsigned FAST_SLOTS = 0xC0u;                     // both ends of the foot capsule (slot = 2 * geom + end)
};

struct Walker2dSpec {
  static constexpr int KIND = 3;
  static constexpr int NB = 7, NV = 9, NU = 6, NG = 7, NXI = 13, NOBS = 17, NSIZE = 4;
  static constexpr int parent[NB] = {-1, 0, 1, 2, 0, 4, 5};
  static constexpr int sgn[NB] = {+1, -1, -1, -1, -1, -1, -1};      // walker2d.xml:29,32,35,38,45,48,51
  static constexpr int geom_body[NG] = {0, 1, 2, 3, 4, 5, 6};
  static constexpr bool RK4 = true;                                 // walker2d.xml:18
  static constexpr int FRAME_SKIP = 4;                              // random_walker2d.py:32
  static constexpr double TIMESTEP = 0.002;
  static constexpr double GRAVITY = 9.81;
  static constexpr double Z_REF = 0.0;                              // ref="1.25", walker2d.xml:28
  static constexpr int NSELF = 0;                                   // conaffinity=0 (walker2d.xml:5)
  static constexpr int self_a[1] = {0};
  static constexpr int self_b[1] = {0};
  static constexpr float gear[NU] = {100.f, 100.f, 100.f, 100.f, 100.f, 100.f}; // :60-65
  static constexpr bool limited[NB] = {false, true, true, true, true, true, true};
  static constexpr double range_lo[NB] = {0, -150 * kDeg, -150 * kDeg, -45 * kDeg, -150 * kDeg, -150 * kDeg, -45 * kDeg};
  static constexpr double range_hi[NB] = {0, 0, 0, 45 * kDeg, 0, 0, 45 * kDeg};
  static constexpr double default_size[4] = {.4, .45, .6, .2};   // random_walker2d.py:21
  static constexpr float ALIVE = 1.0f, CTRL_COST = 1e-3f;           // random_walker2d.py:116-131
  static constexpr float DEFAULT_NOISE_VAR = 1e-3f;                 // random_walker2d.py:30
  static constexpr float INIT_NOISE = 0.005f;                       // random_walker2d.py:148-151
  static constexpr bool DR_BEFORE_STATE = true;                     // random_walker2d.py:145-151
  static constexpr unsigned FAST_SLOTS = (3u << 6) | (3u << 12);    // the two foot capsules (geoms 3 and 6)
};

struct HalfCheetahSpec {
  static constexpr int KIND = 2;
  static constexpr int NB = 7, NV = 9, NU = 6, NG = 8, NXI = 8, NOBS = 17, NSIZE = 8;
  static constexpr int parent[NB] = {-1, 0, 1, 2, 0, 4, 5};
  static constexpr int sgn[NB] = {+1, +1, +1, +1, +1, +1, +1};      // axis="0 1 0" everywhere
  static constexpr int geom_body[NG] = {0, 0, 1, 2, 3, 4, 5, 6};    // torso, head, bthigh..ffoot
  static constexpr bool RK4 = false;                                // half_cheetah.xml:72 (Euler)
  static constexpr int FRAME_SKIP = 5;                              // random_half_cheetah.py:33
  static constexpr double TIMESTEP = 0.01;                          // half_cheetah.xml:72
  static constexpr double GRAVITY = 9.81;                           // half_cheetah.xml:72
  static constexpr double Z_REF = 0.7;   // body pos z=.7, no joint ref (half_cheetah.xml:86,88)
  static constexpr int NSELF = 0;                                   // conaffinity=0 (:57)
  static constexpr int self_a[1] = {0};
  static constexpr int self_b[1] = {0};
  static constexpr float gear[NU] = {120.f, 90.f, 60.f, 120.f, 60.f, 30.f}; // :121-126
  static constexpr bool limited[NB] = {false, true, true, true, true, true, true};
  static constexpr double range_lo[NB] = {0, -.52, -.785, -.4, -1., -1.2, -.5};   // :95-113
  static constexpr double range_hi[NB] = {0, 1.05, .785, .785, .7, .87, .5};
  static constexpr double default_size[8] = {1., .15, .145, .15, .094, .133, .106, .07}; // random_half_cheetah.py:19
  static constexpr float ALIVE = 0.0f, CTRL_COST = 0.1f;            // random_half_cheetah.py:101-110
  static constexpr float DEFAULT_NOISE_VAR = 1e-4f;                 // random_half_cheetah.py:30
  static constexpr float INIT_NOISE = 0.1f;                         // random_half_cheetah.py:124-125
  static constexpr bool DR_BEFORE_STATE = false;
  static constexpr unsigned FAST_SLOTS = (3u << 8) | (3u << 14);    // bfoot and ffoot (geoms 4 and 7)
};

// compile-time tree helpers --------------------------------------------------------------
template <class S>
constexpr bool is_anc_or_self(int a, int b) {  // a is b or an ancestor of b
  while (b >= 0) { if (a == b) return true; b = S::parent[b]; }
  return false;
}
// dof-level coupling: dofs 0,1 (root slides) couple with everything.
template <class S>
constexpr bool dof_coupled(int i, int j) {
  if (i < 2 || j < 2) return true;
  return is_anc_or_self<S>(i - 2, j - 2) || is_anc_or_self<S>(j - 2, i - 2);
}
// parent in the dof tree: 0 <- 1 <- 2 <- hinges
template <class S>
constexpr int dof_parent(int k) {
  if (k <= 2) return k - 1;
  return S::parent[k - 2] + 2;
}

// ---------------------------------------------------------------------------------------
// Run-time model constants ("compiled model").  T = float in the kernels; the host derives
// them in double.  Uniform across a batch for hopper / half-cheetah; per-env for walker2d
// (geometry depends on the xi lengths).
// ---------------------------------------------------------------------------------------
template <class T, class S>
struct PlanarGeom {
  T ja[S::NB][2];     // joint anchor of body i relative to its parent's anchor (parent local axes)
  T co[S::NB][2];     // COM of body i relative to its own anchor (body local axes)
  T iyy[S::NB];       // rotational inertia about y through the COM (NOMINAL)
  T e1[S::NG][2];     // capsule end points relative to the owning body's anchor (body local axes)
  T e2[S::NG][2];
  T radius[S::NG];
  T tran_invw[S::NB]; // body_invweight0[.,0] (translational), nominal
  T dof_invw[S::NB];  // dof_invweight0 of the hinge of body i (i>=1; [0] unused)
  T armature[S::NB], damping[S::NB], stiffness[S::NB]; // per hinge ([0] = root hinge: 0)
};

// solver / contact constants shared by a batch
template <class T>
struct SolParams {
  // contact: solref -> K,B ; solimp (dmin,dmax,width); margin
  T con_K, con_B, con_dmin, con_dmax, con_width, con_margin;
  // joint limits
  T lim_K, lim_B, lim_dmin, lim_dmax, lim_width;
  T meaninertia;   // scale of the convergence test
  int ls_max;      // cap on extra line-search evaluations per Newton iteration (tuning knob)
  int corr;        // single-row Sherman-Morrison correction after a non-exact full step (tuning knob)
  int ls_free;     // leading Newton iterations of a solve that take the full step without a line search (tuning knob)
  int warm;        // start the solver from the previous evaluation's qacc (tuning knob)
  int fast;        // allow the feet-only straight-line solver instantiation (tuning knob / A-B tests; results agree to rounding)
};

}  // namespace rex
