// planar_engine.hpp -- per-environment forward dynamics + soft-constraint solve + integrators
// for the planar kinematic trees, written for ONE ENVIRONMENT PER GPU LANE: every array below
// has compile-time size and is only ever indexed with compile-time constants (static_for), so
// hipcc keeps the whole state in VGPRs -- no scratch, no LDS, no cross-lane traffic.
//
// What it computes is what the reference reaches through `self.sim.step()`
// (random_envs/jinja/jinja_mujoco_env.py:170-173), i.e. MuJoCo's mj_step pipeline
//   kinematics -> M (+armature) -> collision -> constraint rows (joint limits, pyramidal floor
//   contacts) -> passive + bias + actuation -> qacc_smooth -> primal Newton solve -> RK4 / Euler
// specialised to 2-D: rotations are one angle per body, spatial inertia is (m, m, Iyy), and the
// four pyramid edges of a condim-3 floor contact collapse to  n+mu*t, n-mu*t, n, n  because the
// second tangent (y) has a zero Jacobian in a planar tree.
//
// The file is also compilable by a host C++17 compiler (REX_HD is empty there); tests use that
// to compare this exact code in fp32/fp64 against the independent 3-D oracle without a GPU.
#pragma once
#include <math.h>
#include <stdio.h>

#include "planar_spec.hpp"

namespace rex {

REX_HD void sincos_t(float a, float& s, float& c) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(REX_FAST_SINCOS)
  s = __sinf(a); c = __cosf(a);
#elif defined(REX_LIBM_SINCOS) && defined(__HIP_DEVICE_COMPILE__)
  sincosf(a, &s, &c);
#elif defined(REX_LIBM_SINCOS)
  s = sinf(a); c = cosf(a);
#else
  sincos_poly(a, s, c);
#endif
}
REX_HD void sincos_t(double a, double& s, double& c) { s = sin(a); c = cos(a); }
REX_HD float sqrt_t(float a) { return sqrtf(a); }
REX_HD double sqrt_t(double a) { return sqrt(a); }
REX_HD float abs_t(float a) { return fabsf(a); }
REX_HD double abs_t(double a) { return fabs(a); }
template <class T> REX_HD T min_t(T a, T b) { return a < b ? a : b; }
template <class T> REX_HD T max_t(T a, T b) { return a > b ? a : b; }

// per-lane dynamic parameters (the randomised part of the model)
template <class T, class S>
struct LaneParams {
  T mass[S::NB];   // xi masses (body_mass[1:]), random_hopper.py:79-80 etc.
  T mu[S::NG];     // sliding friction of each geom-floor pair
};

// kinematic quantities of one configuration, all positions relative to the root anchor
template <class T, class S>
struct Kin {
  T c[S::NB], s[S::NB];      // cos/sin of the absolute body angle phi_i (rotation about +y)
  T A[S::NB][2];             // joint anchors
  T rc[S::NB][2];            // COM relative to own anchor, world axes
  T zroot;                   // absolute z of the root anchor
};

// rotate local (u,w) by phi about +y:  x' = u c + w s ; z' = -u s + w c
template <class T>
REX_HD void rot(T c, T s, T u, T w, T& x, T& z) { x = u * c + w * s; z = -u * s + w * c; }

template <class T, class S, bool PAIR = false>
REX_HD void kinematics(const T (&q)[S::NV], const PlanarGeom<T, S>& G, Kin<T, S>& K) {
  T phi[S::NB];
  static_for<0, S::NB>([&](auto I) {
    constexpr int i = I;
    if constexpr (i == 0) phi[0] = q[2];
    else phi[i] = phi[S::parent[i]] + T(S::sgn[i]) * q[i + 2];
  });
  if constexpr (PAIR) {   // two lanes per env: each lane evaluates every other body's sin / cos, then they swap
    const bool odd = pair_parity() != 0u;
    static_for<0, (S::NB + 1) / 2>([&](auto HH) {
      constexpr int i0 = 2 * HH, i1 = (2 * HH + 1 < S::NB) ? 2 * HH + 1 : 2 * HH;
      T so, co; sincos_t(odd ? phi[i1] : phi[i0], so, co);
      const T sx = pair_xchg(so), cx = pair_xchg(co);
      K.s[i0] = odd ? sx : so; K.c[i0] = odd ? cx : co;
      if constexpr (i1 != i0) { K.s[i1] = odd ? so : sx; K.c[i1] = odd ? co : cx; }
    });
  }
  static_for<0, S::NB>([&](auto I) {
    constexpr int i = I;
    if constexpr (!PAIR) sincos_t(phi[i], K.s[i], K.c[i]);
    if constexpr (i == 0) { K.A[0][0] = T(0); K.A[0][1] = T(0); }
    else {
      constexpr int p = S::parent[i];
      T dx, dz; rot(K.c[p], K.s[p], G.ja[i][0], G.ja[i][1], dx, dz);
      K.A[i][0] = K.A[p][0] + dx; K.A[i][1] = K.A[p][1] + dz;
    }
    rot(K.c[i], K.s[i], G.co[i][0], G.co[i][1], K.rc[i][0], K.rc[i][1]);
  });
  K.zroot = q[1] + T(S::Z_REF);
}

// Joint-space inertia (lower triangle, only coupled entries are touched) and bias forces.
// M: composite-body recursion about each joint anchor; bias: planar Newton-Euler with the
// velocity-product accelerations (no angular term in 2-D) and gravity.
template <class T, class S>
REX_HD void mass_and_bias(const T (&v)[S::NV], const PlanarGeom<T, S>& G, const LaneParams<T, S>& P,
                          const Kin<T, S>& K, T (&M)[S::NV][S::NV], T (&bias)[S::NV]) {
  T mu[S::NB], h[S::NB][2], I[S::NB];   // composite mass, first moment about anchor, inertia about anchor
  T w[S::NB], Aacc[S::NB][2], Phi[S::NB][2], N[S::NB];
  static_for<0, S::NB>([&](auto II) {
    constexpr int i = II;
    if constexpr (i == 0) { w[0] = v[2]; Aacc[0][0] = T(0); Aacc[0][1] = T(0); }
    else {
      constexpr int p = S::parent[i];
      w[i] = w[p] + T(S::sgn[i]) * v[i + 2];
      T w2 = w[p] * w[p];
      Aacc[i][0] = Aacc[p][0] - w2 * (K.A[i][0] - K.A[p][0]);
      Aacc[i][1] = Aacc[p][1] - w2 * (K.A[i][1] - K.A[p][1]);
    }
    T m = P.mass[i];
    mu[i] = m; h[i][0] = m * K.rc[i][0]; h[i][1] = m * K.rc[i][1];
    I[i] = G.iyy[i] + m * (K.rc[i][0] * K.rc[i][0] + K.rc[i][1] * K.rc[i][1]);
    T wi2 = w[i] * w[i];
    T ax = Aacc[i][0] - wi2 * K.rc[i][0], az = Aacc[i][1] - wi2 * K.rc[i][1] + T(S::GRAVITY);
    Phi[i][0] = m * ax; Phi[i][1] = m * az;
    N[i] = K.rc[i][1] * Phi[i][0] - K.rc[i][0] * Phi[i][1];
  });
  static_rfor<1, S::NB>([&](auto CC) {   // children into parents
    constexpr int c = CC; constexpr int p = S::parent[c];
    T dx = K.A[c][0] - K.A[p][0], dz = K.A[c][1] - K.A[p][1];
    I[p] += I[c] + T(2) * (dx * h[c][0] + dz * h[c][1]) + mu[c] * (dx * dx + dz * dz);
    h[p][0] += h[c][0] + mu[c] * dx; h[p][1] += h[c][1] + mu[c] * dz;
    mu[p] += mu[c];
    N[p] += N[c] + dz * Phi[c][0] - dx * Phi[c][1];
    Phi[p][0] += Phi[c][0]; Phi[p][1] += Phi[c][1];
  });
  M[0][0] = mu[0]; M[1][1] = mu[0]; M[1][0] = T(0);
  bias[0] = Phi[0][0]; bias[1] = Phi[0][1];
  static_for<0, S::NB>([&](auto JJ) {
    constexpr int j = JJ;
    constexpr T sj = T(S::sgn[j]);
    M[j + 2][0] = sj * h[j][1];
    M[j + 2][1] = -sj * h[j][0];
    M[j + 2][j + 2] = I[j] + G.armature[j];
    bias[j + 2] = sj * N[j];
    static_for<0, j>([&](auto II) {
      constexpr int i = II;
      if constexpr (is_anc_or_self<S>(i, j)) {
        constexpr T si = T(S::sgn[i]);
        M[j + 2][i + 2] = si * sj * (I[j] + (K.A[j][0] - K.A[i][0]) * h[j][0] + (K.A[j][1] - K.A[i][1]) * h[j][1]);
      }
    });
  });
}

// y = M x using the lower triangle + tree sparsity
template <class T, class S>
REX_HD void sym_matvec(const T (&M)[S::NV][S::NV], const T (&x)[S::NV], T (&y)[S::NV]) {
  static_for<0, S::NV>([&](auto II) {
    constexpr int i = II;
    T acc = M[i][i] * x[i];
    static_for<0, S::NV>([&](auto JJ) {
      constexpr int j = JJ;
      if constexpr (j < i) { if constexpr (dof_coupled<S>(i, j)) acc += M[i][j] * x[j]; }
      else if constexpr (j > i) { if constexpr (dof_coupled<S>(i, j)) acc += M[j][i] * x[j]; }
    });
    y[i] = acc;
  });
}

// In-place sparse L^T D L factorisation (Featherstone): after the call H[k][k] = 1 / D_k (the solve multiplies) and
// H[k][i] (i ancestor dof of k) = L_ki.  Branch-induced zeros of the tree are never touched,
// and H = M + J^T D J keeps M's sparsity because every constraint row lives on one root path.
template <class T, class S>
REX_HD void ldl_factor(T (&H)[S::NV][S::NV]) {
  static_rfor<1, S::NV>([&](auto KK) {
    constexpr int k = KK;
    T inv = rcp_t(H[k][k]);
    H[k][k] = inv;
    static_rfor<0, k>([&](auto II) {   // ancestors of k, deepest first
      constexpr int i = II;
      if constexpr (dof_coupled<S>(k, i)) {   // i < k and coupled  <=>  i is an ancestor dof of k
        T a = H[k][i] * inv;
        static_for<0, i + 1>([&](auto JJ) {
          constexpr int j = JJ;       // j <= i, ancestor-or-self of i (hence of k)
          if constexpr (dof_coupled<S>(i, j)) H[i][j] -= a * H[k][j];
        });
        H[k][i] = a;
      }
    });
  });
  H[0][0] = rcp_t(H[0][0]);
}
template <class T, class S>
REX_HD void ldl_solve(const T (&H)[S::NV][S::NV], T (&b)[S::NV]) {
  static_rfor<1, S::NV>([&](auto KK) {
    constexpr int k = KK;
    static_for<0, k>([&](auto II) { constexpr int i = II; if constexpr (dof_coupled<S>(k, i)) b[i] -= H[k][i] * b[k]; });
  });
  static_for<0, S::NV>([&](auto KK) { constexpr int k = KK; b[k] = b[k] * H[k][k]; });
  static_for<1, S::NV>([&](auto KK) {
    constexpr int k = KK;
    static_for<0, k>([&](auto II) { constexpr int i = II; if constexpr (dof_coupled<S>(k, i)) b[k] -= H[k][i] * b[i]; });
  });
}

// [3P getimpedance] sigmoid impedance, power = 2, midpoint = 0.5 (every solimp in the four XMLs)
template <class T>
REX_HD T impedance(T dmin, T dmax, T width, T x_abs) {
  T x = x_abs * rcp_t(width);
  // both arms computed, then ONE select: written as a ternary over expressions the compiler emits a divergent branch
  // (s_and_saveexec / s_xor / s_or exec around three instructions, ~13 of them per evaluation)
  const T ya = T(2) * x * x, t1 = T(1) - x, yb = T(1) - T(2) * t1 * t1;
  T y = x < T(0.5) ? ya : yb;
  T imp = dmin + y * (dmax - dmin);
  imp = x >= T(1) ? dmax : imp;
  return (dmin == dmax) ? dmin : imp;
}

// Constraint data of one configuration.  J rows are never stored: they are rebuilt from the
// contact point and the joint anchors whenever needed.  Slot k = 2*geom + end.
template <class T, class S>
struct Constraints {
  static constexpr int NC = 2 * S::NG;
  T px[NC], pz[NC];      // contact point relative to the root anchor
  T dist[NC];            // signed distance of the capsule end to the floor
  T D[NC];               // 1/R of the pyramid edges
  T an[NC], at[NC];      // reference accelerations: edges are (an +/- at) and an (twice)
  unsigned con_mask;     // bit k: slot k within margin
  T lsig[S::NB], lD[S::NB], laref[S::NB];   // joint limits: row = lsig * e_dof
  unsigned lim_mask;
  bool any;              // any row at all in this lane
  unsigned self_possible;   // bit p: capsule-capsule self pair p may be in contact (bounding-circle cull)
};
// capsule-capsule self contacts (condim 1); only materialised on the rare path
template <class T, class S>
struct SelfRows {
  static constexpr int NS = S::NSELF > 0 ? 2 * S::NSELF : 1;   // up to two contacts per pair (parallel axes)
  T px[NS], pz[NS], nx[NS], nz[NS], D[NS], aref[NS];
  unsigned mask;
};

// (t, n) components of J_point * x for a point P on body B
template <class T, class S, int B>
REX_HD void jdot(const Kin<T, S>& K, T px, T pz, const T (&x)[S::NV], T& t, T& n) {
  t = x[0]; n = x[1];
  static_for<0, S::NB>([&](auto JJ) {
    constexpr int j = JJ;
    if constexpr (is_anc_or_self<S>(j, B)) {
      constexpr T sj = T(S::sgn[j]);
      T rx = px - K.A[j][0], rz = pz - K.A[j][1];
      t += sj * rz * x[j + 2]; n -= sj * rx * x[j + 2];
    }
  });
}
// the same for two vectors at once (shares the lever arms)
template <class T, class S, int B>
REX_HD void jdot2(const Kin<T, S>& K, T px, T pz, const T (&x)[S::NV], const T (&y)[S::NV], T& tx, T& nx, T& ty, T& ny) {
  tx = x[0]; nx = x[1]; ty = y[0]; ny = y[1];
  static_for<0, S::NB>([&](auto JJ) {
    constexpr int j = JJ;
    if constexpr (is_anc_or_self<S>(j, B)) {
      constexpr T sj = T(S::sgn[j]);
      T rx = sj * (px - K.A[j][0]), rz = sj * (pz - K.A[j][1]);
      tx += rz * x[j + 2]; nx -= rx * x[j + 2]; ty += rz * y[j + 2]; ny -= rx * y[j + 2];
    }
  });
}
// g += J_point^T (ft, fn)
template <class T, class S, int B>
REX_HD void jt_accum(const Kin<T, S>& K, T px, T pz, T ft, T fn, T (&g)[S::NV]) {
  g[0] += ft; g[1] += fn;
  static_for<0, S::NB>([&](auto JJ) {
    constexpr int j = JJ;
    if constexpr (is_anc_or_self<S>(j, B)) {
      constexpr T sj = T(S::sgn[j]);
      T rx = px - K.A[j][0], rz = pz - K.A[j][1];
      g[j + 2] += sj * (rz * ft - rx * fn);
    }
  });
}
// H += sum_ab u_a^T C u_b over the dofs of body B's root path, u_a = (Jt_a, Jn_a), C = [[ctt,cnt],[cnt,cnn]]
template <class T, class S, int B>
REX_HD void hess_accum(const Kin<T, S>& K, T px, T pz, T ctt, T cnt, T cnn, T (&H)[S::NV][S::NV]) {
  T ut[S::NV], un[S::NV], wt[S::NV], wn[S::NV];
  ut[0] = T(1); un[0] = T(0); ut[1] = T(0); un[1] = T(1);
  static_for<0, S::NB>([&](auto JJ) {
    constexpr int j = JJ;
    if constexpr (is_anc_or_self<S>(j, B)) {
      constexpr T sj = T(S::sgn[j]);
      ut[j + 2] = sj * (pz - K.A[j][1]); un[j + 2] = -sj * (px - K.A[j][0]);
    }
  });
  static_for<0, S::NV>([&](auto AA) {
    constexpr int a = AA;
    if constexpr (a < 2 || is_anc_or_self<S>(a - 2, B)) { wt[a] = ctt * ut[a] + cnt * un[a]; wn[a] = cnt * ut[a] + cnn * un[a]; }
  });
  static_for<0, S::NV>([&](auto AA) {
    constexpr int a = AA;
    if constexpr (a < 2 || is_anc_or_self<S>(a - 2, B))
      static_for<0, a + 1>([&](auto BB) {
        constexpr int b = BB;
        if constexpr (b < 2 || is_anc_or_self<S>(b - 2, B)) H[a][b] += wt[a] * ut[b] + wn[a] * un[b];
      });
  });
}

// The same four operations with the hinge columns of the point Jacobian given (jt[j], jn[j] = tangential / normal entry of
// dof j + 2, ancestors of body B only): the straight-line solver instantiation forms them once per evaluation instead of
// re-deriving the lever arms at every use.
template <class T, class S, int B>
REX_HD void point_jac(const Kin<T, S>& K, T px, T pz, T (&jt)[S::NB], T (&jn)[S::NB]) {
  static_for<0, S::NB>([&](auto JJ) {
    constexpr int j = JJ;
    if constexpr (is_anc_or_self<S>(j, B)) { constexpr T sj = T(S::sgn[j]); jt[j] = sj * (pz - K.A[j][1]); jn[j] = -sj * (px - K.A[j][0]); }
  });
}
template <class T, class S, int B>
REX_HD void jdot_pre(const T (&jt)[S::NB], const T (&jn)[S::NB], const T (&x)[S::NV], T& t, T& n) {
  t = x[0]; n = x[1];
  static_for<0, S::NB>([&](auto JJ) { constexpr int j = JJ; if constexpr (is_anc_or_self<S>(j, B)) { t += jt[j] * x[j + 2]; n += jn[j] * x[j + 2]; } });
}
template <class T, class S, int B>
REX_HD void jt_accum_pre(const T (&jt)[S::NB], const T (&jn)[S::NB], T ft, T fn, T (&g)[S::NV]) {
  g[0] += ft; g[1] += fn;
  static_for<0, S::NB>([&](auto JJ) { constexpr int j = JJ; if constexpr (is_anc_or_self<S>(j, B)) g[j + 2] += jt[j] * ft + jn[j] * fn; });
}
template <class T, class S, int B>
REX_HD void hess_accum_pre(const T (&jt)[S::NB], const T (&jn)[S::NB], T ctt, T cnt, T cnn, T (&H)[S::NV][S::NV]) {
  T ut[S::NV], un[S::NV], wt[S::NV], wn[S::NV];
  ut[0] = T(1); un[0] = T(0); ut[1] = T(0); un[1] = T(1);
  static_for<0, S::NB>([&](auto JJ) { constexpr int j = JJ; if constexpr (is_anc_or_self<S>(j, B)) { ut[j + 2] = jt[j]; un[j + 2] = jn[j]; } });
  static_for<0, S::NV>([&](auto AA) {
    constexpr int a = AA;
    if constexpr (a < 2 || is_anc_or_self<S>(a - 2, B)) { wt[a] = ctt * ut[a] + cnt * un[a]; wn[a] = cnt * ut[a] + cnn * un[a]; }
  });
  static_for<0, S::NV>([&](auto AA) {
    constexpr int a = AA;
    if constexpr (a < 2 || is_anc_or_self<S>(a - 2, B))
      static_for<0, a + 1>([&](auto BB) {
        constexpr int b = BB;
        if constexpr (b < 2 || is_anc_or_self<S>(b - 2, B)) H[a][b] += wt[a] * ut[b] + wn[a] * un[b];
      });
  });
}

// capsule centre / half-axis of geom g in world axes (relative to the root anchor)
template <class T, class S, int g>
REX_HD void capsule_pose(const Kin<T, S>& K, const PlanarGeom<T, S>& G, T (&p)[2], T (&a)[2], T& l) {
  constexpr int b = S::geom_body[g];
  T e1x, e1z, e2x, e2z;
  rot(K.c[b], K.s[b], G.e1[g][0], G.e1[g][1], e1x, e1z); rot(K.c[b], K.s[b], G.e2[g][0], G.e2[g][1], e2x, e2z);
  p[0] = K.A[b][0] + T(0.5) * (e1x + e2x); p[1] = K.A[b][1] + T(0.5) * (e1z + e2z);
  T hx = T(0.5) * (e1x - e2x), hz = T(0.5) * (e1z - e2z);
  l = sqrt_t(hx * hx + hz * hz);
  const T il = rcp_t(l); a[0] = hx * il; a[1] = hz * il;
}

// closest points of two 2-D segments ([3P] mjc_CapsuleCapsule restated in the plane), then
// circle-circle.  Up to two contacts (parallel axes), returned as scalars (no stack arrays).
template <class T>
struct Hit2 { T cx0, cz0, nx0, nz0, d0, cx1, cz1, nx1, nz1, d1; bool h0, h1; };
template <class T>
REX_HD void capsule_capsule_2d(const T (&p1)[2], const T (&a1)[2], T l1, T r1, const T (&p2)[2], const T (&a2)[2],
                               T l2, T r2, T margin, Hit2<T>& H) {
  H.h0 = H.h1 = false;
  H.cx0 = H.cz0 = H.nx0 = H.nz0 = H.d0 = H.cx1 = H.cz1 = H.nx1 = H.nz1 = H.d1 = T(0);
  auto sphere = [&](T c1x, T c1z, T c2x, T c2z) {   // fills slot 0, then slot 1
    T dx = c2x - c1x, dz = c2z - c1z;
    T len = sqrt_t(dx * dx + dz * dz), d = len - r1 - r2;
    if (d > margin) return;
    T ux = T(1), uz = T(0);
    if (len >= T(1e-15)) { const T il = rcp_t(len); ux = dx * il; uz = dz * il; }
    T px_ = c1x + ux * (r1 + d * T(0.5)), pz_ = c1z + uz * (r1 + d * T(0.5));
    // value selects, not `if (!H.h0) H.d0 = d; else H.d1 = d;`: LLVM sinks the two stores into one store through a selected
    // ADDRESS, which puts H into scratch (a store + a load with its own wait per field and evaluation)
    const bool first = !H.h0, second = H.h0 && !H.h1;
    H.d0 = first ? d : H.d0; H.nx0 = first ? ux : H.nx0; H.nz0 = first ? uz : H.nz0; H.cx0 = first ? px_ : H.cx0; H.cz0 = first ? pz_ : H.cz0;
    H.d1 = second ? d : H.d1; H.nx1 = second ? ux : H.nx1; H.nz1 = second ? uz : H.nz1; H.cx1 = second ? px_ : H.cx1; H.cz1 = second ? pz_ : H.cz1;
    H.h1 = H.h1 || second; H.h0 = true;
  };
  T difx = p1[0] - p2[0], difz = p1[1] - p2[1];
  T ma = a1[0] * a1[0] + a1[1] * a1[1], mb = -(a1[0] * a2[0] + a1[1] * a2[1]), mc = a2[0] * a2[0] + a2[1] * a2[1];
  T u = -(a1[0] * difx + a1[1] * difz), v = a2[0] * difx + a2[1] * difz, det = ma * mc - mb * mb;
  if (abs_t(det) >= T(1e-15)) {
    const T idet = rcp_t(det), imc = rcp_t(mc), ima = rcp_t(ma);
    T x1 = (mc * u - mb * v) * idet, x2 = (ma * v - mb * u) * idet;
    if (x1 > l1) { x1 = l1; x2 = (v - mb * l1) * imc; } else if (x1 < -l1) { x1 = -l1; x2 = (v + mb * l1) * imc; }
    if (x2 > l2) { x2 = l2; x1 = (u - mb * l2) * ima; } else if (x2 < -l2) { x2 = -l2; x1 = (u + mb * l2) * ima; }
    if (x1 > l1) x1 = l1; else if (x1 < -l1) x1 = -l1;
    sphere(p1[0] + a1[0] * x1, p1[1] + a1[1] * x1, p2[0] + a2[0] * x2, p2[1] + a2[1] * x2);
    return;
  }
  // parallel axes: end points of segment 1 against segment 2, then of 2 against 1; first two hits
  static_for<0, 2>([&](auto SS) {
    constexpr int sg = 2 * int(SS) - 1;
    T c1x = p1[0] + a1[0] * T(sg) * l1, c1z = p1[1] + a1[1] * T(sg) * l1;
    T x2 = (c1x - p2[0]) * a2[0] + (c1z - p2[1]) * a2[1];
    if (x2 >= -l2 && x2 <= l2) sphere(c1x, c1z, p2[0] + a2[0] * x2, p2[1] + a2[1] * x2);
  });
  static_for<0, 2>([&](auto SS) {
    constexpr int sg = 2 * int(SS) - 1;
    T c2x = p2[0] + a2[0] * T(sg) * l2, c2z = p2[1] + a2[1] * T(sg) * l2;
    T x1 = (c2x - p1[0]) * a1[0] + (c2z - p1[1]) * a1[1];
    if (x1 >= -l1 && x1 <= l1) sphere(p1[0] + a1[0] * x1, p1[1] + a1[1] * x1, c2x, c2z);
  });
}

// f(IC<k>) for every slot k of the compile-time set SLOTS (bit k), in increasing order
template <unsigned SLOTS, class F>
REX_HD void for_slots(F&& f) {
  static_for<0, 32>([&](auto KK) { constexpr int k = KK; if constexpr ((SLOTS >> k) & 1u) f(IC<k>{}); });
}
template <class S> constexpr unsigned all_slots() { return (2 * S::NG >= 32) ? ~0u : ((1u << (2 * S::NG)) - 1u); }

// collision + constraint rows + reference accelerations  ([3P] mj_collision, mj_makeConstraint,
// mj_diagApprox, mj_makeImpedance, mj_referenceConstraint), in two parts:
//   detect_constraints  joint-limit rows, the distance of every capsule end to the floor (which slots are inside the
//                       margin), the bounding-circle cull of the self pairs -- cheap, branch-free, decides which solver
//                       instantiation the wave enters;
//   slot_rows<SLOTS>    impedance, regularisation and reference acceleration of the floor slots in SLOTS; with BR the
//                       slots no lane of the wave touches are skipped by a wave-uniform branch, without BR the code is
//                       straight-line (the fast path: a handful of slots, one basic block).
// PAIR (two lanes per env): each lane tests ONE end of every capsule (end = lane parity) and keeps that end's data in the
// even slot 2g; con_mask is exchanged and holds both ends' bits in both lanes.
template <class T, class S, bool PAIR = false>
REX_HD void detect_constraints(const T (&q)[S::NV], const T (&v)[S::NV], const PlanarGeom<T, S>& G,
                               const SolParams<T>& sp, const Kin<T, S>& K, Constraints<T, S>& C) {
  unsigned lim_mask = 0, con_mask = 0;
  // joint limits (hinge bodies 1..NB-1)
  static_for<1, S::NB>([&](auto JJ) {
    constexpr int j = JJ;
    if constexpr (S::limited[j]) {
      T val = q[j + 2];
      T dlo = val - T(S::range_lo[j]), dhi = T(S::range_hi[j]) - val;
      bool lo = dlo < T(0), hi = dhi < T(0);
      T dist = lo ? dlo : dhi;
      if (lo || hi) lim_mask |= 1u << j;
      T sg = lo ? T(1) : T(-1);
      T imp = impedance(sp.lim_dmin, sp.lim_dmax, sp.lim_width, abs_t(dist));
      // R = max(MINVAL, (1-imp)/imp * dof_invweight0);  D = 1/R
      C.lD[j] = imp * rcp_t(max_t(T(1e-15), (T(1) - imp) * G.dof_invw[j]));
      C.lsig[j] = sg;
      C.laref[j] = -sp.lim_B * (sg * v[j + 2]) - sp.lim_K * imp * dist;
    }
  });
  // capsule ends against the floor plane z = 0
  if constexpr (PAIR) {
    const bool odd = pair_parity() != 0u;
    static_for<0, S::NG>([&](auto GG) {
      constexpr int g = GG; constexpr int b = S::geom_body[g]; constexpr int k = 2 * g;
      T lx = odd ? G.e2[g][0] : G.e1[g][0], lz = odd ? G.e2[g][1] : G.e1[g][1];
      T ox, oz; rot(K.c[b], K.s[b], lx, lz, ox, oz);
      T cx = K.A[b][0] + ox, cz = K.A[b][1] + oz;
      T dist = (cz + K.zroot) - G.radius[g];
      if (dist < sp.con_margin) con_mask |= 1u << k;
      C.px[k] = cx; C.pz[k] = T(0.5) * dist - K.zroot; C.dist[k] = dist;
    });
    con_mask <<= pair_parity();                  // own end's bits sit at 2g + parity
    con_mask |= pair_xchg(con_mask);
  } else
  static_for<0, S::NG>([&](auto GG) {
    constexpr int g = GG; constexpr int b = S::geom_body[g];
    static_for<0, 2>([&](auto EE) {
      constexpr int e = EE; constexpr int k = 2 * g + e;
      T lx = e == 0 ? G.e1[g][0] : G.e2[g][0], lz = e == 0 ? G.e1[g][1] : G.e2[g][1];
      T ox, oz; rot(K.c[b], K.s[b], lx, lz, ox, oz);
      T cx = K.A[b][0] + ox, cz = K.A[b][1] + oz;     // sphere centre rel. root anchor
      T dist = (cz + K.zroot) - G.radius[g];
      if (dist < sp.con_margin) con_mask |= 1u << k;
      C.px[k] = cx; C.pz[k] = T(0.5) * dist - K.zroot;   // midpoint between the surfaces
      C.dist[k] = dist;
    });
  });
  C.lim_mask = lim_mask; C.con_mask = con_mask;
  // bounding-circle cull of the capsule-capsule self pairs
  unsigned sp_any = 0u;
  if constexpr (S::NSELF > 0 && PAIR) {
    // two lanes per env: the capsule centre is the mean of its two ends, one in each lane (C.px[2g] is the own end's x
    // relative to the root anchor; the end's z is recovered from the distance to the floor)
    T ccx[S::NG], ccz[S::NG];
    static_for<0, S::NG>([&](auto GG) { constexpr int g = GG; constexpr int k = 2 * g;
      const T ez = C.dist[k] + G.radius[g];   // end-sphere centre z, absolute (the pair difference below cancels the offset)
      ccx[g] = T(0.5) * (C.px[k] + pair_xchg(C.px[k])); ccz[g] = T(0.5) * (ez + pair_xchg(ez)); });
    static_for<0, S::NSELF>([&](auto PP) {
      constexpr int p = PP; constexpr int ga = S::self_a[p], gb = S::self_b[p];
      T dx = ccx[gb] - ccx[ga], dz = ccz[gb] - ccz[ga];
      T la2 = (G.e1[ga][0] - G.e2[ga][0]) * (G.e1[ga][0] - G.e2[ga][0]) + (G.e1[ga][1] - G.e2[ga][1]) * (G.e1[ga][1] - G.e2[ga][1]);
      T lb2 = (G.e1[gb][0] - G.e2[gb][0]) * (G.e1[gb][0] - G.e2[gb][0]) + (G.e1[gb][1] - G.e2[gb][1]) * (G.e1[gb][1] - G.e2[gb][1]);
      T reach = T(0.5) * (sqrt_t(la2) + sqrt_t(lb2)) + G.radius[ga] + G.radius[gb] + sp.con_margin;
      sp_any |= (dx * dx + dz * dz <= reach * reach) ? (1u << p) : 0u;
    });
  } else if constexpr (S::NSELF > 0) {
    static_for<0, S::NSELF>([&](auto PP) {
      constexpr int p = PP; constexpr int ga = S::self_a[p], gb = S::self_b[p];
      constexpr int ba = S::geom_body[ga], bb = S::geom_body[gb];
      T ax, az, bx, bz;
      rot(K.c[ba], K.s[ba], T(0.5) * (G.e1[ga][0] + G.e2[ga][0]), T(0.5) * (G.e1[ga][1] + G.e2[ga][1]), ax, az);
      rot(K.c[bb], K.s[bb], T(0.5) * (G.e1[gb][0] + G.e2[gb][0]), T(0.5) * (G.e1[gb][1] + G.e2[gb][1]), bx, bz);
      T dx = (K.A[bb][0] + bx) - (K.A[ba][0] + ax), dz = (K.A[bb][1] + bz) - (K.A[ba][1] + az);
      T la2 = (G.e1[ga][0] - G.e2[ga][0]) * (G.e1[ga][0] - G.e2[ga][0]) + (G.e1[ga][1] - G.e2[ga][1]) * (G.e1[ga][1] - G.e2[ga][1]);
      T lb2 = (G.e1[gb][0] - G.e2[gb][0]) * (G.e1[gb][0] - G.e2[gb][0]) + (G.e1[gb][1] - G.e2[gb][1]) * (G.e1[gb][1] - G.e2[gb][1]);
      T reach = T(0.5) * (sqrt_t(la2) + sqrt_t(lb2)) + G.radius[ga] + G.radius[gb] + sp.con_margin;
      sp_any |= (dx * dx + dz * dz <= reach * reach) ? (1u << p) : 0u;
    });
  }
  C.self_possible = sp_any;
  C.any = (lim_mask | con_mask) != 0u;
}

template <class T, class S, unsigned SLOTS, bool BR, bool PAIR = false>
REX_HD void slot_rows(const T (&v)[S::NV], const PlanarGeom<T, S>& G, const LaneParams<T, S>& P, const SolParams<T>& sp,
                      const Kin<T, S>& K, Constraints<T, S>& C) {
  const unsigned par = PAIR ? pair_parity() : 0u;   // PAIR: SLOTS are the even slots 2g, the lane's own end is 2g + parity
  for_slots<SLOTS>([&](auto KK) {
    constexpr int k = KK; constexpr int g = k / 2; constexpr int b = S::geom_body[g];
    const bool act = (C.con_mask >> (k + par)) & 1u;
    bool go = true;
    if constexpr (BR) go = REX_WAVE_ANY(act);
    if (go) {
      const T dist = C.dist[k];
      T mu = P.mu[g], mu2 = mu * mu;
      T imp = impedance(sp.con_dmin, sp.con_dmax, sp.con_width, abs_t(dist - sp.con_margin));
      // R1 = (1-imp)/imp * tran*(1+mu^2) ; Rpy = 2 mu^2 R1 ; D = 1/Rpy
      C.D[k] = imp * rcp_t(max_t(T(1e-15), T(2) * mu2 * (T(1) - imp) * (G.tran_invw[b] * (T(1) + mu2))));
      T vt, vn; jdot<T, S, b>(K, C.px[k], C.pz[k], v, vt, vn);
      C.an[k] = -sp.con_B * vn - sp.con_K * imp * (dist - sp.con_margin);
      C.at[k] = -sp.con_B * mu * vt;
    } else { C.D[k] = T(0); C.an[k] = T(0); C.at[k] = T(0); }
  });
}

template <class T, class S>
REX_HD void make_self_rows(const T (&v)[S::NV], const PlanarGeom<T, S>& G, const SolParams<T>& sp, const Kin<T, S>& K,
                           unsigned possible, SelfRows<T, S>& R) {
  unsigned mask = 0;
  if constexpr (S::NSELF > 0) {
    static_for<0, S::NSELF>([&](auto PP) {
      constexpr int p = PP; constexpr int ga = S::self_a[p], gb = S::self_b[p];
      constexpr int ba = S::geom_body[ga], bb = S::geom_body[gb];
      static_for<0, 2>([&](auto KK) { constexpr int r = 2 * p + KK; R.px[r] = R.pz[r] = R.nx[r] = R.nz[r] = R.D[r] = R.aref[r] = T(0); });
      if (REX_WAVE_ANY((possible >> p) & 1u)) {   // pairs whose bounding circles are apart in every lane cost nothing
        T p1[2], a1[2], p2[2], a2[2], l1, l2;
        capsule_pose<T, S, ga>(K, G, p1, a1, l1); capsule_pose<T, S, gb>(K, G, p2, a2, l2);
        // Second cull: separating axes of the two SEGMENTS (each axis and its normal), inflated by r1 + r2 + margin.  The
        // bounding circles of two long thin capsules overlap for most of a folded leg's range without the capsules being
        // anywhere near each other, and a wave that gets past the cull pays the whole narrow phase -- the launch ends with
        // its slowest wave.  Any axis with a gap > R is a certificate that no point pair is within R (conservative).
        {
          const T R = G.radius[ga] + G.radius[gb] + sp.con_margin;
          const T dx = p2[0] - p1[0], dz = p2[1] - p1[1];
          const T cr = abs_t(a1[0] * a2[1] - a1[1] * a2[0]), dt = abs_t(a1[0] * a2[0] + a1[1] * a2[1]);
          const T s1 = abs_t(-a1[1] * dx + a1[0] * dz) - l2 * cr;          // normal of segment 1
          const T s2 = abs_t(-a2[1] * dx + a2[0] * dz) - l1 * cr;          // normal of segment 2
          const T t1 = abs_t(a1[0] * dx + a1[1] * dz) - l1 - l2 * dt;      // along segment 1
          const T t2 = abs_t(a2[0] * dx + a2[1] * dz) - l2 - l1 * dt;      // along segment 2
          const bool apart = s1 > R || s2 > R || t1 > R || t2 > R;
          if (!REX_WAVE_ANY(((possible >> p) & 1u) && !apart)) return;     // (return from this pair's lambda)
        }
        Hit2<T> H;
        capsule_capsule_2d(p1, a1, l1, G.radius[ga], p2, a2, l2, G.radius[gb], sp.con_margin, H);
        static_for<0, 2>([&](auto KK) {
          constexpr int k = KK; constexpr int r = 2 * p + k;
          const bool hit_k = k == 0 ? H.h0 : H.h1; const T dist_k = k == 0 ? H.d0 : H.d1;
          bool act = hit_k && dist_k < sp.con_margin;
          if (act) {
            mask |= 1u << r;
            R.px[r] = k == 0 ? H.cx0 : H.cx1; R.pz[r] = k == 0 ? H.cz0 : H.cz1; R.nx[r] = k == 0 ? H.nx0 : H.nx1; R.nz[r] = k == 0 ? H.nz0 : H.nz1;
            T imp = impedance(sp.con_dmin, sp.con_dmax, sp.con_width, abs_t(dist_k - sp.con_margin));
            R.D[r] = imp * rcp_t(max_t(T(1e-15), (T(1) - imp) * (G.tran_invw[ba] + G.tran_invw[bb])));
            T ta, na, tb, nb; jdot<T, S, ba>(K, R.px[r], R.pz[r], v, ta, na); jdot<T, S, bb>(K, R.px[r], R.pz[r], v, tb, nb);
            T vel = R.nx[r] * (tb - ta) + R.nz[r] * (nb - na);
            R.aref[r] = -sp.con_B * vel - sp.con_K * imp * (dist_k - sp.con_margin);
          }
        });
      }
    });
  }
  R.mask = mask;
}

// -DREX_MARKS: comment markers in the ISA (profiles/isa_regions.py counts the instructions between them)
#if defined(REX_MARKS) && defined(__HIP_DEVICE_COMPILE__)
#define REX_MARK(name) asm volatile("; REXMARK " name)
#else
#define REX_MARK(name) ((void)0)
#endif

#if (defined(REX_WAVETIME) || defined(REX_PHASES)) && !defined(REX_NOPHASES) && defined(__HIP_DEVICE_COMPILE__)
// diagnostic build only: cycles per phase of forward(), summed per wave (lane 0) with fire-and-forget atomics.  The stamp
// takes a value the phase produced as an input, so that value is complete before the clock is read.
extern __device__ unsigned long long g_evalphase[8192][16];
#define REX_PSTAMP(var, dep) unsigned long long var; { float dep_ = (float)(dep); asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) : "v"(dep_) : "memory"); }
#define REX_PACC(slot, t0, t1) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_evalphase[blockIdx.x & 8191][slot], (t1) - (t0)); } while (0)
#else
#define REX_PSTAMP(var, dep) ((void)0)
#define REX_PACC(slot, t0, t1) ((void)0)
#endif

struct SolveStats { int iters; bool capped; int mode; };   // mode: solver instantiation forward() entered (0 none, 1 general, 2 general + self rows, 3 feet-only straight-line)
#if defined(REX_STATS) && !defined(__HIP_DEVICE_COMPILE__)
struct GlobalStats { long solves, iters, pass1, pass2, ls_evals, nocon, slots_active; long toggles[4][4]; int trace[64], ntrace; };   // trace: mode * 100 + Newton iterations of the last solves
inline GlobalStats& gstats() { static GlobalStats g{}; return g; }
#define REX_COUNT(field, n) (gstats().field += (n))
#elif defined(REX_KSTATS) && defined(__HIP_DEVICE_COMPILE__)
// diagnostic build only: wave-level event counts (lane 0 of each wave adds)
extern __device__ unsigned long long g_kstats[8];
enum { KS_solves = 0, KS_iters = 1, KS_pass1 = 2, KS_pass2 = 3, KS_ls_evals = 4, KS_nocon = 5, KS_slots_active = 6 };
#define REX_COUNT(field, n) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_kstats[KS_##field], (unsigned long long)(n)); } while (0)
#elif defined(REX_WAVETIME) && defined(__HIP_DEVICE_COMPILE__)
// diagnostic build only: the same counts per WAVE (slot = workgroup index), next to the wave's cycle count
extern __device__ unsigned long long g_waveinfo[8192][8];
enum { KS_solves = 0, KS_iters = 1, KS_pass1 = 2, KS_pass2 = 3, KS_ls_evals = 4, KS_nocon = 5, KS_slots_active = 6, KS_selfpath = 7 };
#define REX_COUNT(field, n) do { if ((threadIdx.x & 63) == 0) g_waveinfo[blockIdx.x & 8191][KS_##field] += (unsigned long long)(n); } while (0)
#else
#define REX_COUNT(field, n) ((void)0)
#endif

// ---- Woodbury correction after a full Newton step: shared pieces ---------------------------------------------------------------------------
// x1 = x + sr minimises the quadratic model of the set A its Hessian H was built for.  If the set at x1 differs from A in a few GROUPS of rows
// -- a joint limit (one column e_j), or the pyramid edges of one floor slot (all combinations of its two basis rows j_t, j_n) -- the new set's
// Newton step from x1 needs no new factorisation:  H1 = H + U dC U^T,  grad_A1(x1) = U w  =>  x2 = x1 - V (I + dC G)^-1 w,  V = H^-1 U,  G = U^T V.
// One group is a 2 x 2 system (round 2).  TWO groups, one per lane of a pair (round 4): each lane solves for its own group's columns only and the
// coupling G_AB = U_A^T V_B is formed from the partner's V (DPP); block elimination of the partner's 2 x 2 block gives each lane its own z.  The
// groups are dealt by capsule END (slot parity = lane parity; a limit goes to the lane without a slot, two limits one each), so every combination
// of two groups except two slots of the same end is covered: limit + slot is 3/4 of what keeps the hopper's slowest waves iterating.
template <class T, class S>
struct CorrGroup { T Ut[S::NV], Un[S::NV]; T ctt, ctn, cnn, wt, wn; };   // columns (U_n = 0 for a limit), dC (symmetric 2 x 2), w
// What the toggles (limits t_lim, floor slots tm) allow: `can` one group, or two that the two lanes can share (SPLIT); and which of them
// virtual lane vp takes: slots by END (slot parity = lane parity), a single limit goes to the lane without a slot, two limits one each.
template <bool SPLIT>
struct CorrPlan {
  unsigned t_lim, even, odd; int nl; bool can, two;
  REX_HD CorrPlan(unsigned t_lim_, unsigned tm, int corr, bool base) : t_lim(t_lim_), even(SPLIT ? (tm & 0x55555555u) : tm), odd(SPLIT ? (tm & 0xAAAAAAAAu) : 0u) {
    nl = __builtin_popcount(t_lim);
    const int ne = __builtin_popcount(even), no = __builtin_popcount(odd), ng = nl + ne + no;
    two = SPLIT && corr >= 2 && base && ng == 2 && ne <= 1 && no <= 1;
    can = two || (base && corr != 0 && ng == 1);
  }
  REX_HD unsigned my_slots(unsigned vp) const { return !can ? 0u : (SPLIT ? (vp ? odd : even) : even); }
  REX_HD unsigned my_lim(unsigned vp) const {
    if (!can) return 0u;
    if constexpr (!SPLIT) return t_lim;
    const unsigned lane_for_one = (even != 0u && odd == 0u) ? 1u : 0u;          // a single limit: the lane without a slot (lane 0 if neither has one)
    const unsigned lo = t_lim & (0u - t_lim), hi = t_lim & (t_lim - 1u);        // two limits: the lower one to lane 0
    return nl == 1 ? (vp == lane_for_one ? t_lim : 0u) : (vp ? hi : lo);        // (nl == 0: lo = hi = 0)
  }
};
// z of MY group in the coupled system  [I + Cm Gmm, Cm X; Ct X^T, I + Ct Gtt] [zm; zt] = [wm; wt]  (X = Um^T Vt): eliminate the partner's block
template <class T>
REX_HD bool corr_pair_solve(T cmtt, T cmtn, T cmnn, T gmtt, T gmtn, T gmnn, T wmt, T wmn,
                            T c2tt, T c2tn, T c2nn, T g2tt, T g2tn, T g2nn, T w2t, T w2n,
                            T xtt, T xtn, T xnt, T xnn, T& zt, T& zn) {
  const T t11 = T(1) + c2tt * g2tt + c2tn * g2tn, t12 = c2tt * g2tn + c2tn * g2nn, t21 = c2tn * g2tt + c2nn * g2tn, t22 = T(1) + c2tn * g2tn + c2nn * g2nn;
  const T dT = t11 * t22 - t12 * t21;
  bool ok = dT > T(1e-3);
  const T iT = ok ? rcp_t(dT) : T(0);
  const T b11 = c2tt * xtt + c2tn * xtn, b12 = c2tt * xnt + c2tn * xnn, b21 = c2tn * xtt + c2nn * xtn, b22 = c2tn * xnt + c2nn * xnn;   // Ct X^T
  const T y11 = (t22 * b11 - t12 * b21) * iT, y12 = (t22 * b12 - t12 * b22) * iT, y21 = (t11 * b21 - t21 * b11) * iT, y22 = (t11 * b22 - t21 * b12) * iT;
  const T y1 = (t22 * w2t - t12 * w2n) * iT, y2 = (t11 * w2n - t21 * w2t) * iT;
  const T c11 = cmtt * xtt + cmtn * xnt, c12 = cmtt * xtn + cmtn * xnn, c21 = cmtn * xtt + cmnn * xnt, c22 = cmtn * xtn + cmnn * xnn;   // Cm X
  const T s11 = T(1) + cmtt * gmtt + cmtn * gmtn - (c11 * y11 + c12 * y21), s12 = cmtt * gmtn + cmtn * gmnn - (c11 * y12 + c12 * y22);
  const T s21 = cmtn * gmtt + cmnn * gmtn - (c21 * y11 + c22 * y21), s22 = T(1) + cmtn * gmtn + cmnn * gmnn - (c21 * y12 + c22 * y22);
  const T r1 = wmt - (c11 * y1 + c12 * y2), r2 = wmn - (c21 * y1 + c22 * y2);
  const T dS = s11 * s22 - s12 * s21;
  ok = ok && dS > T(1e-3);
  const T iS = ok ? rcp_t(dS) : T(0);
  zt = (s22 * r1 - s12 * r2) * iS; zn = (s11 * r2 - s21 * r1) * iS;
  return ok;
}
// dx = V (I + dC G)^-1 w for the group(s) `build(vp, group)` describes (all-zero group: none).  MODE 0: one group in this lane (the one-lane
// device kernels); 1: this lane's group + the pair partner's (DPP); 2: both virtual lanes in this lane (host builds: the same arithmetic
// as mode 1, so the CPU tests hold the two-group algebra to the oracle).  `two` (wave-uniform): some lane has two groups, form the coupling.
// `use`: this lane (pair) applies the step; dx = 0 where it does not or where a pivot was too small (returns false there).
template <class T, class S, int MODE, class Build>
REX_HD bool corr_step(const T (&H)[S::NV][S::NV], bool two, bool use, unsigned par, Build&& build, T (&dx)[S::NV]) {
  constexpr int NG_ = MODE == 2 ? 2 : 1;
  CorrGroup<T, S> g[NG_];
  T vt[NG_][S::NV], vn[NG_][S::NV], Gtt[NG_], Gtn[NG_], Gnn[NG_];
  static_for<0, NG_>([&](auto LL) { constexpr int l = LL;
    build(MODE == 2 ? unsigned(l) : par, g[l]);
    static_for<0, S::NV>([&](auto II) { vt[l][II] = g[l].Ut[II]; vn[l][II] = g[l].Un[II]; });
    ldl_solve<T, S>(H, vt[l]); ldl_solve<T, S>(H, vn[l]);
    Gtt[l] = Gtn[l] = Gnn[l] = T(0);
    static_for<0, S::NV>([&](auto II) { Gtt[l] += g[l].Ut[II] * vt[l][II]; Gtn[l] += g[l].Ut[II] * vn[l][II]; Gnn[l] += g[l].Un[II] * vn[l][II]; }); });
  bool good;
  if constexpr (MODE == 0) {
    const T a11 = T(1) + g[0].ctt * Gtt[0] + g[0].ctn * Gtn[0], a12 = g[0].ctt * Gtn[0] + g[0].ctn * Gnn[0];
    const T a21 = g[0].ctn * Gtt[0] + g[0].cnn * Gtn[0], a22 = T(1) + g[0].ctn * Gtn[0] + g[0].cnn * Gnn[0];
    const T det = a11 * a22 - a12 * a21;
    good = use && det > T(1e-3);
    const T idet = good ? rcp_t(det) : T(0);
    const T zt = (a22 * g[0].wt - a12 * g[0].wn) * idet, zn = (a11 * g[0].wn - a21 * g[0].wt) * idet;
    static_for<0, S::NV>([&](auto II) { dx[II] = zt * vt[0][II] + zn * vn[0][II]; });
  } else if constexpr (MODE == 1) {
    T zt, zn;
    if (two) {   // (wave-uniform) some pair couples two groups: the partner's scalars and V, the coupling X = Um^T Vt, block elimination
      const T c2tt = pair_xchg(g[0].ctt), c2tn = pair_xchg(g[0].ctn), c2nn = pair_xchg(g[0].cnn), w2t = pair_xchg(g[0].wt), w2n = pair_xchg(g[0].wn);
      const T g2tt = pair_xchg(Gtt[0]), g2tn = pair_xchg(Gtn[0]), g2nn = pair_xchg(Gnn[0]);
      T xtt = T(0), xtn = T(0), xnt = T(0), xnn = T(0);
      static_for<0, S::NV>([&](auto II) { const T pt = pair_xchg(vt[0][II]), pn = pair_xchg(vn[0][II]);
        xtt += g[0].Ut[II] * pt; xtn += g[0].Ut[II] * pn; xnt += g[0].Un[II] * pt; xnn += g[0].Un[II] * pn; });
      good = corr_pair_solve(g[0].ctt, g[0].ctn, g[0].cnn, Gtt[0], Gtn[0], Gnn[0], g[0].wt, g[0].wn, c2tt, c2tn, c2nn, g2tt, g2tn, g2nn, w2t, w2n,
                             xtt, xtn, xnt, xnn, zt, zn);
    } else {     // every pair has one group at most: each lane's own 2 x 2 (the lane without a group solves the identity)
      const T a11 = T(1) + g[0].ctt * Gtt[0] + g[0].ctn * Gtn[0], a12 = g[0].ctt * Gtn[0] + g[0].ctn * Gnn[0];
      const T a21 = g[0].ctn * Gtt[0] + g[0].cnn * Gtn[0], a22 = T(1) + g[0].ctn * Gtn[0] + g[0].cnn * Gnn[0];
      const T det = a11 * a22 - a12 * a21;
      good = det > T(1e-3);
      const T idet = good ? rcp_t(det) : T(0);
      zt = (a22 * g[0].wt - a12 * g[0].wn) * idet; zn = (a11 * g[0].wn - a21 * g[0].wt) * idet;
    }
    good = use && good && (pair_xchg(good ? 1u : 0u) != 0u);
    zt = good ? zt : T(0); zn = good ? zn : T(0);
    static_for<0, S::NV>([&](auto II) { constexpr int i = II; const T d = zt * vt[0][i] + zn * vn[0][i]; dx[i] = d + pair_xchg(d); });
  } else {
    T xtt = T(0), xtn = T(0), xnt = T(0), xnn = T(0);   // X = U_0^T V_1
    if (two) static_for<0, S::NV>([&](auto II) { xtt += g[0].Ut[II] * vt[1][II]; xtn += g[0].Ut[II] * vn[1][II]; xnt += g[0].Un[II] * vt[1][II]; xnn += g[0].Un[II] * vn[1][II]; });
    T z0t, z0n, z1t, z1n;
    const bool ok0 = corr_pair_solve(g[0].ctt, g[0].ctn, g[0].cnn, Gtt[0], Gtn[0], Gnn[0], g[0].wt, g[0].wn,
                                     g[1].ctt, g[1].ctn, g[1].cnn, Gtt[1], Gtn[1], Gnn[1], g[1].wt, g[1].wn, xtt, xtn, xnt, xnn, z0t, z0n);
    const bool ok1 = corr_pair_solve(g[1].ctt, g[1].ctn, g[1].cnn, Gtt[1], Gtn[1], Gnn[1], g[1].wt, g[1].wn,
                                     g[0].ctt, g[0].ctn, g[0].cnn, Gtt[0], Gtn[0], Gnn[0], g[0].wt, g[0].wn, xtt, xnt, xtn, xnn, z1t, z1n);   // (X^T)
    good = use && ok0 && ok1;
    static_for<0, S::NV>([&](auto II) { dx[II] = good ? (z0t * vt[0][II] + z0n * vn[0][II]) + (z1t * vt[1][II] + z1n * vn[1][II]) : T(0); });
  }
  return good;
}

// Primal Newton solve of   min_a 0.5 (a-a0)^T M (a-a0) + sum_rows 0.5 D min(0, J a - aref)^2
// ([3P] engine_solver, Newton, pyramidal cones): exact Hessian M + J^T D_active J with the tree
// sparsity of M, L^T D L factorisation, exact line search on the piecewise-quadratic 1-D cost.
// All lanes of a wave iterate together; a lane that has converged keeps alpha = 0.  Rows are
// re-derived from (contact point, anchors) on the fly and slots no lane of the wave touches are
// skipped with a wave-uniform branch: the solver's live state is M, H, g and 5 floats per slot.
// SLOTS: compile-time set of floor slots this instantiation looks at; BR: skip the slots no lane of the wave touches with
// a wave-uniform branch (general path) or run them all straight-line (fast path: few slots, no branches in an iteration).
template <bool BR> REX_HD bool any_lane(bool x) { if constexpr (BR) return REX_WAVE_ANY(x); else return true; }
// PAIR (two lanes per env, straight-line instantiation only): SLOTS are the even slots 2g of the feet; each lane runs the
// per-slot part of both passes for ITS end (slot 2g + parity, data kept at index 2g) and the partial gradient / Hessian /
// line-search sums and the active-edge bits are exchanged (pair_xchg); everything else is replicated.
template <class T, class S, bool SELF, unsigned SLOTS, bool BR, bool PAIR = false, int MAXIT = 24>
REX_HD SolveStats solve_newton(const T (&M)[S::NV][S::NV], const T (&qfrc_smooth)[S::NV], const T (&qacc_smooth)[S::NV],
                               const Kin<T, S>& K, const Constraints<T, S>& C, const SelfRows<T, S>& R,
                               const LaneParams<T, S>& P, T (&qacc)[S::NV], bool warm, bool have_a0, int ls_max, int ls_free = 0, int corr = 1) {
  // MuJoCo starts at qacc_smooth (warmstart is disabled in all the XMLs); the minimiser is unique, so
  // starting from the previous RK4 stage's solution only changes how fast the active set is found
  // (a lane without any row is only here because another lane of its wave has one: it must leave with qacc_smooth)
  // With a warm start qacc_smooth is NOT computed (forward() skips that factorisation): a lane without rows then takes one
  // Newton step like everybody else -- its Hessian is M itself, so the step lands on M^-1 qfrc_smooth exactly.
  // (have_a0: qacc_smooth was computed -- cold starts, and every solve when the `fast` knob is off: then a lane without rows
  // leaves with qacc_smooth itself and its arithmetic does not depend on what the other lanes of its wave are doing.)
  const bool has_rows = C.any || (SELF && R.mask != 0u);
  // (opaque: left alone, LLVM turns this select between two arrays into a select of POINTERS, which keeps both arrays in
  // scratch memory and costs a dependent ~500-cycle scratch round trip per solve)
  static_for<0, S::NV>([&](auto II) { T prev = qacc[II], cold = qacc_smooth[II]; opaque(prev); opaque(cold);
                                      qacc[II] = have_a0 ? ((warm && has_rows) ? prev : cold) : prev; });
  SolveStats st{0, false, 0};
  // stop when the force residual |M a - f - J^T f_c| is at rounding level relative to the forces
  // that balance in it (the piecewise-quadratic cost makes Newton exact once the active set is right)
  const T tol2 = sizeof(T) == 4 ? T(1e-9) : T(1e-24);
  const T stag = sizeof(T) == 4 ? T(1e-6) : T(1e-15);
  // previous iteration's active edges: if a Newton step lands in the region it was computed for,
  // every row kept its sign along the step (rows are linear in alpha), the cost was exactly
  // quadratic there and the step was its exact minimiser -> converged, independent of rounding
  unsigned p_lim = ~0u, p_e1 = ~0u, p_e2 = ~0u, p_e3 = ~0u, p_self = ~0u;
  bool lane_done = !has_rows && have_a0;
  constexpr int NC = 2 * S::NG;
  static_assert(!PAIR || (!BR && !SELF), "PAIR: straight-line feet-only instantiation");
  const unsigned par = PAIR ? pair_parity() : 0u;
  // straight-line instantiation: hinge columns of the point Jacobians once per solve; J qacc of pass 1 is reused by pass 2
  T Jt[BR ? 1 : NC][S::NB], Jn[BR ? 1 : NC][S::NB], lt[NC], ln[NC];
  if constexpr (!BR) for_slots<SLOTS>([&](auto KK) { constexpr int k = KK; point_jac<T, S, S::geom_body[k / 2]>(K, C.px[k], C.pz[k], Jt[k], Jn[k]); });
  T Ma[S::NV];   // M qacc: formed once, then carried along the accepted steps (Ma += alpha * M sr)
  sym_matvec<T, S>(M, qacc, Ma);
  bool ma_dirty = false;   // wave-uniform: a single-row correction moved qacc without updating Ma
#if defined(REX_DIAG_MAXIT)   // timing diagnostics only (wrong results): cap the Newton iterations of every solve
  const int maxit = REX_DIAG_MAXIT;
#else
  constexpr int maxit = MAXIT;
#endif
  for (int it = 0; it < maxit; ++it) {
    if (!REX_WAVE_ANY(!lane_done)) break;
    T cpx[NC], cpz[NC];   // per-iteration opaque copies of the contact points (see opaque())
    if constexpr (BR) for_slots<SLOTS>([&](auto KK) { constexpr int k = KK; cpx[k] = C.px[k]; cpz[k] = C.pz[k]; opaque(cpx[k]); opaque(cpz[k]); });
    if (ma_dirty) { sym_matvec<T, S>(M, qacc, Ma); ma_dirty = false; }
    REX_COUNT(pass1, 1);
    REX_MARK("pass1");
    REX_PSTAMP(s_0, qacc[0]);
    // ---- pass 1: gradient and active edges --------------------------------------------------
    T g[S::NV];
    T fref = T(0);
    static_for<0, S::NV>([&](auto II) { constexpr int i = II; g[i] = Ma[i] - qfrc_smooth[i]; fref += Ma[i] * Ma[i] + qfrc_smooth[i] * qfrc_smooth[i]; });
    unsigned lim_on = 0, e1 = 0, e2 = 0, e3 = 0, self_on = 0;
    T gs[PAIR ? S::NV : 1];   // PAIR: this lane's slot contributions to the gradient
    if constexpr (PAIR) static_for<0, S::NV>([&](auto II) { gs[II] = T(0); });
    static_for<1, S::NB>([&](auto JJ) {
      constexpr int j = JJ;
      if constexpr (S::limited[j]) {
        T jar = C.lsig[j] * qacc[j + 2] - C.laref[j];
        bool on = ((C.lim_mask >> j) & 1u) && jar < T(0);
        if (on) lim_on |= 1u << j;
        g[j + 2] += on ? C.lsig[j] * C.lD[j] * jar : T(0);
      }
    });
    for_slots<SLOTS>([&](auto KK) {
      {
        constexpr int k = KK; constexpr int gg = k / 2; constexpr int b = S::geom_body[gg];
        const bool act = (C.con_mask >> (k + par)) & 1u;
        if (any_lane<BR>(act)) {
          const T mu = P.mu[gg];
          T jt, jn;
          if constexpr (BR) jdot<T, S, b>(K, cpx[k], cpz[k], qacc, jt, jn);
          else { jdot_pre<T, S, b>(Jt[k], Jn[k], qacc, jt, jn); lt[k] = jt; ln[k] = jn; }
          T r1 = jn + mu * jt - (C.an[k] + C.at[k]), r2 = jn - mu * jt - (C.an[k] - C.at[k]), r3 = jn - C.an[k];
          bool s1 = act && r1 < T(0), s2 = act && r2 < T(0), s3 = act && r3 < T(0);
          if (s1) e1 |= 1u << (k + par); if (s2) e2 |= 1u << (k + par); if (s3) e3 |= 1u << (k + par);
          T f1 = s1 ? -C.D[k] * r1 : T(0), f2 = s2 ? -C.D[k] * r2 : T(0), f3 = s3 ? -C.D[k] * r3 : T(0);
          if constexpr (BR) jt_accum<T, S, b>(K, cpx[k], cpz[k], -(mu * (f1 - f2)), -(f1 + f2 + T(2) * f3), g);
          else if constexpr (PAIR) jt_accum_pre<T, S, b>(Jt[k], Jn[k], -(mu * (f1 - f2)), -(f1 + f2 + T(2) * f3), gs);
          else jt_accum_pre<T, S, b>(Jt[k], Jn[k], -(mu * (f1 - f2)), -(f1 + f2 + T(2) * f3), g);
        }
      }
    });
    if constexpr (PAIR) {   // both ends' contributions and active edges, identical in both lanes (commutative sums)
      static_for<0, S::NV>([&](auto II) { constexpr int i = II; g[i] += gs[i] + pair_xchg(gs[i]); });
      e1 |= pair_xchg(e1); e2 |= pair_xchg(e2); e3 |= pair_xchg(e3);
    }
    if constexpr (SELF && S::NSELF > 0) {
      static_for<0, 2 * S::NSELF>([&](auto PP) {
        constexpr int p = PP; constexpr int ba = S::geom_body[S::self_a[p / 2]], bb = S::geom_body[S::self_b[p / 2]];
        const bool act = (R.mask >> p) & 1u;
        if (REX_WAVE_ANY(act)) {
          T ta, na, tb, nb; jdot<T, S, ba>(K, R.px[p], R.pz[p], qacc, ta, na); jdot<T, S, bb>(K, R.px[p], R.pz[p], qacc, tb, nb);
          T jar = R.nx[p] * (tb - ta) + R.nz[p] * (nb - na) - R.aref[p];
          bool on = act && jar < T(0);
          if (on) self_on |= 1u << p;
          T f = on ? -R.D[p] * jar : T(0);
          jt_accum<T, S, bb>(K, R.px[p], R.pz[p], -R.nx[p] * f, -R.nz[p] * f, g);
          jt_accum<T, S, ba>(K, R.px[p], R.pz[p], R.nx[p] * f, R.nz[p] * f, g);
        }
      });
    }
    T gn = T(0);
    static_for<0, S::NV>([&](auto II) { gn += g[II] * g[II]; });
#if defined(REX_DEBUG_SOLVER) && !defined(__HIP_DEVICE_COMPILE__)
    if (it >= 12) printf("  it %d gn/fref %.3e con_mask %x e1 %x e2 %x e3 %x lim %x\n", it, double(gn / fref), C.con_mask, e1, e2, e3, lim_on);
#endif
    const bool same_set = lim_on == p_lim && e1 == p_e1 && e2 == p_e2 && e3 == p_e3 && self_on == p_self;
    p_lim = lim_on; p_e1 = e1; p_e2 = e2; p_e3 = e3; p_self = self_on;
    lane_done = lane_done || same_set || !(gn > tol2 * fref);   // NaN counts as done
    if (!REX_WAVE_ANY(!lane_done)) break;
    REX_PSTAMP(s_1, gn);
    REX_PACC(5, s_0, s_1);
    REX_COUNT(pass2, 1);
    REX_MARK("pass2_hess");
    // ---- pass 2: Hessian of the current active set, Newton direction ------------------------
    T H[S::NV][S::NV];
    static_for<0, S::NV>([&](auto II) { constexpr int i = II;
      static_for<0, i + 1>([&](auto JJ) { constexpr int j = JJ; if constexpr (dof_coupled<S>(i, j)) H[i][j] = M[i][j]; }); });
    static_for<1, S::NB>([&](auto JJ) { constexpr int j = JJ;
      if constexpr (S::limited[j]) H[j + 2][j + 2] += ((lim_on >> j) & 1u) ? C.lD[j] : T(0); });
    T Hs[PAIR ? S::NV : 1][PAIR ? S::NV : 1];   // PAIR: this lane's slot contributions to the Hessian
    if constexpr (PAIR) static_for<0, S::NV>([&](auto II) { constexpr int i = II;
      static_for<0, i + 1>([&](auto JJ) { constexpr int j = JJ; if constexpr (dof_coupled<S>(i, j)) Hs[i][j] = T(0); }); });
    for_slots<SLOTS>([&](auto KK) {
      {
        constexpr int k = KK; constexpr int gg = k / 2; constexpr int b = S::geom_body[gg];
        const unsigned bit = 1u << (k + par);
        if (any_lane<BR>(((e1 | e2 | e3) & bit) != 0u)) {
          const T mu = P.mu[gg];
          T s1 = (e1 & bit) ? T(1) : T(0), s2 = (e2 & bit) ? T(1) : T(0), s3 = (e3 & bit) ? T(1) : T(0);
          if constexpr (BR) hess_accum<T, S, b>(K, cpx[k], cpz[k], C.D[k] * mu * mu * (s1 + s2), C.D[k] * mu * (s1 - s2), C.D[k] * (s1 + s2 + T(2) * s3), H);
          else if constexpr (PAIR) hess_accum_pre<T, S, b>(Jt[k], Jn[k], C.D[k] * mu * mu * (s1 + s2), C.D[k] * mu * (s1 - s2), C.D[k] * (s1 + s2 + T(2) * s3), Hs);
          else hess_accum_pre<T, S, b>(Jt[k], Jn[k], C.D[k] * mu * mu * (s1 + s2), C.D[k] * mu * (s1 - s2), C.D[k] * (s1 + s2 + T(2) * s3), H);
        }
      }
    });
    if constexpr (PAIR) static_for<0, S::NV>([&](auto II) { constexpr int i = II;
      static_for<0, i + 1>([&](auto JJ) { constexpr int j = JJ; if constexpr (dof_coupled<S>(i, j)) H[i][j] += Hs[i][j] + pair_xchg(Hs[i][j]); }); });
    if constexpr (SELF && S::NSELF > 0) {
      static_for<0, 2 * S::NSELF>([&](auto PP) {
        constexpr int p = PP; constexpr int ba = S::geom_body[S::self_a[p / 2]], bb = S::geom_body[S::self_b[p / 2]];
        if (REX_WAVE_ANY(((self_on >> p) & 1u) != 0u)) {
          // row = n.(J_b - J_a): built explicitly, rank-1 update (hopper is a chain: H is dense)
          T row[S::NV];
          static_for<0, S::NV>([&](auto II) { row[II] = T(0); });
          jt_accum<T, S, bb>(K, R.px[p], R.pz[p], R.nx[p], R.nz[p], row); jt_accum<T, S, ba>(K, R.px[p], R.pz[p], -R.nx[p], -R.nz[p], row);
          T d = ((self_on >> p) & 1u) ? R.D[p] : T(0);
          static_for<0, S::NV>([&](auto AA) { constexpr int a = AA;
            static_for<0, a + 1>([&](auto BB) { constexpr int bq = BB; if constexpr (dof_coupled<S>(a, bq)) H[a][bq] += d * row[a] * row[bq]; }); });
        }
      });
    }
    REX_PSTAMP(s_h, H[S::NV - 1][0] + H[S::NV - 1][S::NV - 1] + H[2][2]);
    REX_PACC(7, s_1, s_h);
    REX_MARK("pass2_ldl");
    ldl_factor<T, S>(H);
    T sr[S::NV];
    static_for<0, S::NV>([&](auto II) { sr[II] = -g[II]; });
    ldl_solve<T, S>(H, sr);
    REX_PSTAMP(s_l, sr[0] + sr[S::NV - 1]);
    REX_PACC(8, s_h, s_l);
    REX_MARK("pass2_ls");
    // ---- exact line search on phi(alpha); phi'(0) = g.sr, Gauss curvature sr^T M sr ----------
    T Ms[S::NV];
    sym_matvec<T, S>(M, sr, Ms);
    T q1 = T(0), q2 = T(0), d0 = T(0);
    static_for<0, S::NV>([&](auto II) { q1 += sr[II] * (Ma[II] - qfrc_smooth[II]); q2 += sr[II] * Ms[II]; d0 += sr[II] * g[II]; });
    unsigned m_lim, m_e1, m_e2, m_e3, m_self;   // rows active at the last evaluated alpha
    // every row is linear in alpha: x(alpha) = r + alpha v with r = J qacc - aref, v = J sr.  The slot products J qacc, J sr
    // are formed once per Newton iteration; an evaluation of phi' is then a few multiply-adds per slot.
    T lvt[NC], lvn[NC];
    for_slots<SLOTS>([&](auto KK) {
      constexpr int k = KK; constexpr int b = S::geom_body[k / 2];
      if constexpr (BR) {
        lt[k] = ln[k] = lvt[k] = lvn[k] = T(0);
        if (REX_WAVE_ANY((C.con_mask >> k) & 1u)) jdot2<T, S, b>(K, cpx[k], cpz[k], qacc, sr, lt[k], ln[k], lvt[k], lvn[k]);
      } else jdot_pre<T, S, b>(Jt[k], Jn[k], sr, lvt[k], lvn[k]);   // J qacc: lt / ln of pass 1
    });
    auto deriv = [&](T a, T& d1, T& d2) {
      REX_COUNT(ls_evals, 1);
      REX_MARK("deriv");
      m_lim = m_e1 = m_e2 = m_e3 = m_self = 0u;
      d1 = q1 + a * q2; d2 = q2;
      static_for<1, S::NB>([&](auto JJ) { constexpr int j = JJ;
        if constexpr (S::limited[j]) {
          T lr = C.lsig[j] * qacc[j + 2] - C.laref[j], lv = C.lsig[j] * sr[j + 2];
          T x = lr + a * lv; bool on = ((C.lim_mask >> j) & 1u) && x < T(0);
          if (on) m_lim |= 1u << j;
          T dd = on ? C.lD[j] : T(0); d1 += dd * x * lv; d2 += dd * lv * lv; } });
      T d1s = T(0), d2s = T(0);   // PAIR: this lane's slot part of phi', phi''
      for_slots<SLOTS>([&](auto KK) {
        {
          constexpr int k = KK; constexpr int gg = k / 2;
          const bool act = (C.con_mask >> (k + par)) & 1u;
          if (any_lane<BR>(act)) {
            const T mu = P.mu[gg];
            const T jt = lt[k], jn = ln[k], vt = lvt[k], vn = lvn[k];
            T r0 = jn + mu * jt - (C.an[k] + C.at[k]), r1 = jn - mu * jt - (C.an[k] - C.at[k]), r2 = jn - C.an[k];
            T v0 = vn + mu * vt, v1 = vn - mu * vt, v2 = vn;
            T x0 = r0 + a * v0, x1 = r1 + a * v1, x2 = r2 + a * v2;
            const bool o0 = act && x0 < T(0), o1 = act && x1 < T(0), o2 = act && x2 < T(0);
            if (o0) m_e1 |= 1u << (k + par); if (o1) m_e2 |= 1u << (k + par); if (o2) m_e3 |= 1u << (k + par);
            T w0 = o0 ? C.D[k] : T(0), w1 = o1 ? C.D[k] : T(0), w2 = o2 ? T(2) * C.D[k] : T(0);
            if constexpr (PAIR) { d1s += w0 * x0 * v0 + w1 * x1 * v1 + w2 * x2 * v2; d2s += w0 * v0 * v0 + w1 * v1 * v1 + w2 * v2 * v2; }
            else { d1 += w0 * x0 * v0 + w1 * x1 * v1 + w2 * x2 * v2; d2 += w0 * v0 * v0 + w1 * v1 * v1 + w2 * v2 * v2; }
          }
        }
      });
      if constexpr (PAIR) {
        d1 += d1s + pair_xchg(d1s); d2 += d2s + pair_xchg(d2s);
        m_e1 |= pair_xchg(m_e1); m_e2 |= pair_xchg(m_e2); m_e3 |= pair_xchg(m_e3);
      }
      if constexpr (SELF && S::NSELF > 0) static_for<0, 2 * S::NSELF>([&](auto PP) {
        constexpr int p = PP; constexpr int ba = S::geom_body[S::self_a[p / 2]], bb = S::geom_body[S::self_b[p / 2]];
        const bool act = (R.mask >> p) & 1u;
        if (REX_WAVE_ANY(act)) {
          T ta, na, tb, nb, ua, ma_, ub, mb_;
          jdot2<T, S, ba>(K, R.px[p], R.pz[p], qacc, sr, ta, na, ua, ma_); jdot2<T, S, bb>(K, R.px[p], R.pz[p], qacc, sr, tb, nb, ub, mb_);
          T r = R.nx[p] * (tb - ta) + R.nz[p] * (nb - na) - R.aref[p], vv = R.nx[p] * (ub - ua) + R.nz[p] * (mb_ - ma_);
          T x = r + a * vv; const bool on = act && x < T(0);
          if (on) m_self |= 1u << p;
          T dd = on ? R.D[p] : T(0);
          d1 += dd * x * vv; d2 += dd * vv * vv;
        }
      });
    };
    // the Newton step itself (alpha = 1) is exact whenever the active set does not change along it
    const T d1ref = abs_t(d0) * T(sizeof(T) == 4 ? 1e-5 : 1e-13) + T(1e-30);
    T a = T(1), lo = T(0), hi = T(-1), d1, d2;
    deriv(a, d1, d2);
    bool ls_done = lane_done || abs_t(d1) <= d1ref;
    // The first `ls_free` iterations of a solve take the full Newton step without a line search (semismooth Newton: on
    // this piecewise-quadratic cost it usually finds the active set in as many iterations as with the exact search, and
    // every evaluation of phi' costs a pass over all rows); from then on the exact search, which guarantees descent,
    // takes over -- a solve that has not converged by then is a hard one (cycling active sets).
    const int ls_cap = it < ls_free ? 0 : ls_max;
    for (int ls = 0; ls < ls_cap; ++ls) {   // phi' is piecewise linear and increasing: safeguarded Newton
      if (!REX_WAVE_ANY(!ls_done)) break;
      if (d1 < T(0)) lo = a; else hi = a;
      T an_ = a - d1 * rcp_t(d2);
      if (hi >= T(0) && (an_ <= lo || an_ >= hi)) an_ = T(0.5) * (lo + hi);
      an_ = max_t(an_, lo);
      T prev = a;
      a = ls_done ? a : an_;
      deriv(a, d1, d2);
      ls_done = ls_done || abs_t(d1) <= d1ref || a == prev;
    }
    REX_PSTAMP(s_d, d1 + d2 + a);
    REX_PACC(9, s_l, s_d);
    REX_MARK("pass2_update");
    // full Newton step that stays in the region its Hessian was built for: exact minimiser, no
    // verification pass needed (same argument as `same_set` above)
    const bool exact_step = a == T(1) && m_lim == lim_on && m_e1 == e1 && m_e2 == e2 && m_e3 == e3 && m_self == self_on;
    a = lane_done ? T(0) : a;
    T amax = T(0), smax = T(0);
    static_for<0, S::NV>([&](auto II) { qacc[II] += a * sr[II]; Ma[II] += a * Ms[II]; amax = max_t(amax, abs_t(qacc[II])); smax = max_t(smax, abs_t(a * sr[II])); });
#if defined(REX_DEBUG_SOLVER) && !defined(__HIP_DEVICE_COMPILE__)
    if (it >= 12) printf("     alpha %.6g smax %.3e amax %.3e d1 %.3e d0 %.3e\n", double(a), double(smax), double(amax), double(d1), double(d0));
#endif
    lane_done = lane_done || exact_step || smax <= stag * (T(1) + amax);   // stagnation at rounding level
    REX_PSTAMP(s_u, qacc[0] + amax + smax);
    REX_PACC(10, s_d, s_u);
    // ---- one-group correction (straight-line instantiation) ---------------------------------------------------------
    // x1 = x + sr minimises the quadratic model of the set A its Hessian was built for.  The usual reason a lane needs
    // another iteration is that the set at x1 differs from A in ONE group of rows: one joint limit, or the (up to three)
    // pyramid edges of one floor slot (a contact making / breaking, stick <-> slip).  All rows of a slot are combinations
    // of its two basis rows U = [j_t j_n], so the new set's Newton step from x1 needs no new factorisation (Woodbury):
    //     H1 = H + U dC U^T,  grad_A1(x1) = U w   =>   x2 = x1 - V (I + dC G)^-1 w,   V = H^-1 U,  G = U^T V   (2 x 2),
    // dC / w = change of the slot's edge weights / the toggled edges' residual terms at x1.  Two independent solves with the
    // factorisation at hand + a check of the set at x2 (~250 instructions) instead of a full iteration (~600) -- and in a
    // 32-env wave a full iteration is paid by everybody whenever ONE env needs it.  x2 is accepted as converged only if the
    // set at x2 equals the set at x1 (then it is that set's exact minimiser); otherwise the regular iterations go on from x2.
    // PAIR: the lane that owns the toggled slot computes the step, the other one contributes zero; limits: the even lane.
    if constexpr (!BR && !SELF) {
#if defined(REX_STATS) && !defined(__HIP_DEVICE_COMPILE__)
      { const int nl = __builtin_popcount(lim_on ^ m_lim), ns = __builtin_popcount((e1 ^ m_e1) | (e2 ^ m_e2) | (e3 ^ m_e3));
        if (!lane_done && a == T(1)) gstats().toggles[nl < 3 ? nl : 3][ns < 3 ? ns : 3]++; }
#endif
      // groups are dealt to the two lanes of a pair; host builds deal them to two virtual lanes of the one lane (same arithmetic, so the CPU
      // tests cover it); the one-lane device kernels keep the single-group form (a second group there is two more solves and their registers)
#if defined(__HIP_DEVICE_COMPILE__)
      constexpr int CMODE = PAIR ? 1 : 0;
#else
      constexpr int CMODE = 2;
#endif
      constexpr bool SPLIT = CMODE != 0;
      // (a second ROUND from x1 against the set found at x2 -- host replay: two-iteration hopper solves 4 284 -> 376 -- was measured on the
      // device and dropped: a wave repeats the round when ANY lane asks for it and still pays the full iteration when another lane needs
      // that: + 1 % hopper, + 5 % walker2d, + 8 % half-cheetah.)  corr: 1 = one group (round 2), 2 = two groups.
      const CorrPlan<SPLIT> plan(lim_on ^ m_lim, (e1 ^ m_e1) | (e2 ^ m_e2) | (e3 ^ m_e3), corr, !lane_done && a == T(1));
      bool can = plan.can;
      if (REX_WAVE_ANY(can)) {
        const bool two = SPLIT && REX_WAVE_ANY(plan.two);
        T jt1[NC], jn1[NC];
        for_slots<SLOTS>([&](auto KK) { constexpr int k = KK; jt1[k] = lt[k] + lvt[k]; jn1[k] = ln[k] + lvn[k]; });   // J x1 (alpha = 1)
        auto build = [&](unsigned vp, CorrGroup<T, S>& g) {
          const unsigned my_lim = plan.my_lim(vp), my_slots = plan.my_slots(vp);
          static_for<0, S::NV>([&](auto II) { g.Ut[II] = T(0); g.Un[II] = T(0); });
          g.ctt = g.ctn = g.cnn = g.wt = g.wn = T(0);
          static_for<1, S::NB>([&](auto JJ) { constexpr int j = JJ;
            if constexpr (S::limited[j]) {
              const bool b = (my_lim >> j) & 1u;
              const T sg = ((m_lim >> j) & 1u) ? T(1) : T(-1);            // switched on / off
              const T xr = C.lsig[j] * qacc[j + 2] - C.laref[j];          // the row at x1
              g.Ut[j + 2] = b ? C.lsig[j] : T(0); g.ctt += b ? sg * C.lD[j] : T(0); g.wt += b ? sg * C.lD[j] * xr : T(0); } });
          for_slots<SLOTS>([&](auto KK) {
            constexpr int k = KK; constexpr int gg = k / 2; constexpr int b = S::geom_body[gg];
            const T mu = P.mu[gg]; const unsigned kk = k + par;
            const T x0 = jn1[k] + mu * jt1[k] - (C.an[k] + C.at[k]), x1 = jn1[k] - mu * jt1[k] - (C.an[k] - C.at[k]), x2 = jn1[k] - C.an[k];
            const T d1_ = T(int((m_e1 >> kk) & 1u) - int((e1 >> kk) & 1u)), d2_ = T(int((m_e2 >> kk) & 1u) - int((e2 >> kk) & 1u)),
                    d3_ = T(int((m_e3 >> kk) & 1u) - int((e3 >> kk) & 1u));   // +1 edge switched on, -1 off, 0 unchanged
            const T any = ((my_slots >> kk) & 1u) ? T(1) : T(0);
            const T Dk = any * C.D[k];
            g.ctt += Dk * mu * mu * (d1_ + d2_); g.ctn += Dk * mu * (d1_ - d2_); g.cnn += Dk * (d1_ + d2_ + T(2) * d3_);
            g.wt += Dk * mu * (d1_ * x0 - d2_ * x1); g.wn += Dk * (d1_ * x0 + d2_ * x1 + T(2) * d3_ * x2);
            jt_accum_pre<T, S, b>(Jt[k], Jn[k], any, T(0), g.Ut);
            jt_accum_pre<T, S, b>(Jt[k], Jn[k], T(0), any, g.Un);
          });
        };
        T dx[S::NV];
        can = corr_step<T, S, CMODE>(H, two, can, par, build, dx);
        static_for<0, S::NV>([&](auto II) { constexpr int i = II; qacc[i] -= dx[i]; });
        // the set at x2
        unsigned v_lim = 0u, v1 = 0u, v2 = 0u, v3 = 0u;
        static_for<1, S::NB>([&](auto JJ) { constexpr int j = JJ;
          if constexpr (S::limited[j]) { const T x = C.lsig[j] * qacc[j + 2] - C.laref[j]; if (((C.lim_mask >> j) & 1u) && x < T(0)) v_lim |= 1u << j; } });
        for_slots<SLOTS>([&](auto KK) {
          constexpr int k = KK; constexpr int gg = k / 2; constexpr int b = S::geom_body[gg];
          const T mu = P.mu[gg]; const bool act = (C.con_mask >> (k + par)) & 1u;
          T ut, un; jdot_pre<T, S, b>(Jt[k], Jn[k], dx, ut, un);
          const T jt2 = jt1[k] - ut, jn2 = jn1[k] - un;
          const T x0 = jn2 + mu * jt2 - (C.an[k] + C.at[k]), x1 = jn2 - mu * jt2 - (C.an[k] - C.at[k]), x2 = jn2 - C.an[k];
          if (act && x0 < T(0)) v1 |= 1u << (k + par); if (act && x1 < T(0)) v2 |= 1u << (k + par); if (act && x2 < T(0)) v3 |= 1u << (k + par);
        });
        if constexpr (PAIR) { v1 |= pair_xchg(v1); v2 |= pair_xchg(v2); v3 |= pair_xchg(v3); }
#if defined(REX_DIAG_CORR) && REX_DIAG_CORR >= 1
        const bool ok2 = true;
#else
        const bool ok2 = v_lim == m_lim && v1 == m_e1 && v2 == m_e2 && v3 == m_e3;
#endif
#if defined(REX_STATS) && !defined(__HIP_DEVICE_COMPILE__)
        if (can) { gstats().toggles[3][3]++; if (ok2) gstats().toggles[3][2]++; }   // corrections tried / accepted
#endif
        lane_done = lane_done || (can && ok2);
        // x2 was computed for the set at x1: that is what the next gradient pass compares with
        p_lim = can ? m_lim : p_lim; p_e1 = can ? m_e1 : p_e1; p_e2 = can ? m_e2 : p_e2; p_e3 = can ? m_e3 : p_e3;
        ma_dirty = true;
        REX_COUNT(nocon, 1);   // (diagnostic builds: slot "nocon" counts the correction trips of the wave)
      }
    }
    REX_PSTAMP(s_2, qacc[0] + amax);
    REX_PACC(6, s_1, s_2); REX_PACC(11, s_u, s_2);
    st.iters = it + 1;
    if (it == MAXIT - 1) st.capped = REX_WAVE_ANY(!lane_done);
  }
  return st;
}

// The general instantiation, ROLLED (template flag of forward() / substep()): the same primal Newton solve over an explicit list of constraint rows -- joint
// limits, the pyramid edges n + mu t, n - mu t, n (weight 2) of every floor slot inside the margin, capsule-capsule self rows -- kept in
// runtime-indexed local arrays (scratch on the device) and walked by runtime loops.  Far slower per solve than the unrolled general
// instantiation, and far smaller in registers: for a chain whose batches practically never leave the feet-only path (the hopper: 100 % of
// the wave-solves under a random policy) the kernel's register allocation is then the feet-only path's own -- 256 VGPRs, no AGPR --, so TWO
// waves share a SIMD and fill each other's issue slots.  That pays where a SIMD has waves queued: the one-lane-per-env hopper kernel past
// 65 536 envs per GPU (131 072 envs: 709 -> 777 M env-steps/s, 2^20: 866 -> 1 422 M).  While every wave has a SIMD to itself the feet-only path
// is 5-20 % slower on 256 registers than on 458, and a solve that does take this path costs several times the unrolled one (dependent scratch
// reads per row and pass), which at one wave per SIMD is what the launch then waits for: those batches keep the unrolled general
// instantiation (rex_hip.hip: rex_create picks per handle).
template <class T, class S>
struct RowList {
  static constexpr int MAXR = (S::NB - 1) + 3 * 2 * S::NG + (S::NSELF > 0 ? 2 * S::NSELF : 0);
  T J[MAXR][S::NV], D[MAXR], aref[MAXR];
  int n;
};
template <class T, class S>
REX_HD void build_rows(const Kin<T, S>& K, const Constraints<T, S>& C, const SelfRows<T, S>& R, const LaneParams<T, S>& P, bool self, RowList<T, S>& L) {
  int n = 0;
  auto push = [&](const T (&row)[S::NV], T D, T aref) { for (int k = 0; k < S::NV; k++) L.J[n][k] = row[k]; L.D[n] = D; L.aref[n] = aref; n++; };
  static_for<1, S::NB>([&](auto JJ) { constexpr int j = JJ;
    if constexpr (S::limited[j]) { if ((C.lim_mask >> j) & 1u) { T row[S::NV]; static_for<0, S::NV>([&](auto II) { row[II] = T(0); }); row[j + 2] = C.lsig[j]; push(row, C.lD[j], C.laref[j]); } } });
  static_for<0, 2 * S::NG>([&](auto KK) { constexpr int k = KK; constexpr int gg = k / 2; constexpr int b = S::geom_body[gg];
    if (REX_WAVE_ANY((C.con_mask >> k) & 1u)) {
      if ((C.con_mask >> k) & 1u) {
        T ut[S::NV], un[S::NV];
        static_for<0, S::NV>([&](auto II) { ut[II] = T(0); un[II] = T(0); });
        jt_accum<T, S, b>(K, C.px[k], C.pz[k], T(1), T(0), ut); jt_accum<T, S, b>(K, C.px[k], C.pz[k], T(0), T(1), un);
        const T mu = P.mu[gg];
        T r1[S::NV], r2[S::NV];
        static_for<0, S::NV>([&](auto II) { r1[II] = un[II] + mu * ut[II]; r2[II] = un[II] - mu * ut[II]; });
        push(r1, C.D[k], C.an[k] + C.at[k]); push(r2, C.D[k], C.an[k] - C.at[k]); push(un, T(2) * C.D[k], C.an[k]);
      }
    } });
  if constexpr (S::NSELF > 0) {
    if (self) static_for<0, 2 * S::NSELF>([&](auto PP) { constexpr int q = PP; constexpr int ba = S::geom_body[S::self_a[q / 2]], bb = S::geom_body[S::self_b[q / 2]];
      if ((R.mask >> q) & 1u) {
        T row[S::NV]; static_for<0, S::NV>([&](auto II) { row[II] = T(0); });
        jt_accum<T, S, bb>(K, R.px[q], R.pz[q], R.nx[q], R.nz[q], row); jt_accum<T, S, ba>(K, R.px[q], R.pz[q], -R.nx[q], -R.nz[q], row);
        push(row, R.D[q], R.aref[q]);
      } });
  }
  L.n = n;
}
template <class T, class S, int MAXIT = 24>
REX_HD SolveStats solve_newton_rolled(const T (&M)[S::NV][S::NV], const T (&qfrc_smooth)[S::NV], const T (&qacc_smooth)[S::NV], const RowList<T, S>& L,
                                      T (&qacc)[S::NV], bool warm, bool have_a0, int ls_max, int ls_free) {
  const int n = L.n;
  const bool has_rows = n > 0;
  static_for<0, S::NV>([&](auto II) { T prev = qacc[II], cold = qacc_smooth[II]; opaque(prev); opaque(cold);
                                      qacc[II] = have_a0 ? ((warm && has_rows) ? prev : cold) : prev; });
  SolveStats st{0, false, 0};
  const T tol2 = sizeof(T) == 4 ? T(1e-9) : T(1e-24);
  const T stag = sizeof(T) == 4 ? T(1e-6) : T(1e-15);
  unsigned long long p_on = ~0ull;
  bool lane_done = !has_rows && have_a0;
  static_assert(RowList<T, S>::MAXR <= 64, "active-row mask");
  for (int it = 0; it < MAXIT; ++it) {
    if (!REX_WAVE_ANY(!lane_done)) break;
    T Ma[S::NV], g[S::NV];
    sym_matvec<T, S>(M, qacc, Ma);
    T fref = T(0);
    static_for<0, S::NV>([&](auto II) { constexpr int i = II; g[i] = Ma[i] - qfrc_smooth[i]; fref += Ma[i] * Ma[i] + qfrc_smooth[i] * qfrc_smooth[i]; });
    unsigned long long on = 0ull;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int r = 0; r < n; r++) {
      T jar = -L.aref[r];
      static_for<0, S::NV>([&](auto II) { jar += L.J[r][II] * qacc[II]; });
      if (jar < T(0)) { on |= 1ull << r; const T w = L.D[r] * jar; static_for<0, S::NV>([&](auto II) { g[II] += L.J[r][II] * w; }); }
    }
    T gn = T(0);
    static_for<0, S::NV>([&](auto II) { gn += g[II] * g[II]; });
    const bool same_set = on == p_on;
    p_on = on;
    lane_done = lane_done || same_set || !(gn > tol2 * fref);
    if (!REX_WAVE_ANY(!lane_done)) break;
    T H[S::NV][S::NV];
    static_for<0, S::NV>([&](auto II) { constexpr int i = II;
      static_for<0, i + 1>([&](auto JJ) { constexpr int j = JJ; if constexpr (dof_coupled<S>(i, j)) H[i][j] = M[i][j]; }); });
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int r = 0; r < n; r++) {
      if ((on >> r) & 1ull) {
        T row[S::NV]; static_for<0, S::NV>([&](auto II) { row[II] = L.J[r][II]; });
        const T d = L.D[r];
        static_for<0, S::NV>([&](auto AA) { constexpr int a = AA;
          static_for<0, a + 1>([&](auto BB) { constexpr int b = BB; if constexpr (dof_coupled<S>(a, b)) H[a][b] += d * row[a] * row[b]; }); });
      }
    }
    ldl_factor<T, S>(H);
    T sr[S::NV];
    static_for<0, S::NV>([&](auto II) { sr[II] = -g[II]; });
    ldl_solve<T, S>(H, sr);
    T Ms[S::NV];
    sym_matvec<T, S>(M, sr, Ms);
    T q1 = T(0), q2 = T(0), d0 = T(0);
    static_for<0, S::NV>([&](auto II) { q1 += sr[II] * (Ma[II] - qfrc_smooth[II]); q2 += sr[II] * Ms[II]; d0 += sr[II] * g[II]; });
    unsigned long long m_on = 0ull;
    auto deriv = [&](T a, T& d1, T& d2) {
      m_on = 0ull; d1 = q1 + a * q2; d2 = q2;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
      for (int r = 0; r < n; r++) {
        T x = -L.aref[r], vv = T(0);
        static_for<0, S::NV>([&](auto II) { x += L.J[r][II] * qacc[II]; vv += L.J[r][II] * sr[II]; });
        x += a * vv;
        if (x < T(0)) { m_on |= 1ull << r; d1 += L.D[r] * x * vv; d2 += L.D[r] * vv * vv; }
      }
    };
    const T d1ref = abs_t(d0) * T(sizeof(T) == 4 ? 1e-5 : 1e-13) + T(1e-30);
    T a = T(1), lo = T(0), hi = T(-1), d1, d2;
    deriv(a, d1, d2);
    bool ls_done = lane_done || abs_t(d1) <= d1ref;
    const int ls_cap = it < ls_free ? 0 : ls_max;
    for (int ls = 0; ls < ls_cap; ++ls) {
      if (!REX_WAVE_ANY(!ls_done)) break;
      if (d1 < T(0)) lo = a; else hi = a;
      T an_ = a - d1 * rcp_t(d2);
      if (hi >= T(0) && (an_ <= lo || an_ >= hi)) an_ = T(0.5) * (lo + hi);
      an_ = max_t(an_, lo);
      T prev = a;
      a = ls_done ? a : an_;
      deriv(a, d1, d2);
      ls_done = ls_done || abs_t(d1) <= d1ref || a == prev;
    }
    const bool exact_step = a == T(1) && m_on == on;
    a = lane_done ? T(0) : a;
    T amax = T(0), smax = T(0);
    static_for<0, S::NV>([&](auto II) { qacc[II] += a * sr[II]; amax = max_t(amax, abs_t(qacc[II])); smax = max_t(smax, abs_t(a * sr[II])); });
    lane_done = lane_done || exact_step || smax <= stag * (T(1) + amax);
    st.iters = it + 1;
    if (it == MAXIT - 1) st.capped = REX_WAVE_ANY(!lane_done);
  }
  return st;
}

// ---- the general instantiation as a LIST solver (GEN = 2 of forward() / substep()) -----------------------------------------------------
// The unrolled general instantiation keeps 12 values per floor slot in registers for all 2 NG slots at once (96 / 168 / 192 registers for
// hopper / walker2d / half-cheetah) and spends a wave-uniform branch per slot and pass; being in the same kernel it taxes the feet-only path
// (its long-lived state gets AGPR homes: hopper 0.0885 -> 0.0788 ms with the general instantiations compiled out), and for walker2d and the
// half-cheetah it IS the cost centre (a wave with a thigh on the floor is what a launch waits for; 69 % of the half-cheetah's wave-solves).
// Here the same primal Newton solve walks a runtime LIST of the contact units some lane of the wave has inside the margin:
//   * a unit is one capsule END (slot 2 g + end).  With two lanes per env (PAIR) lane parity = end, exactly as in the feet-only path: each
//     lane runs the per-unit part of every pass for ITS end of every listed capsule and the partial gradients / Hessians / phi' sums and
//     edge bits are exchanged with DPP (pair_xchg); with one lane per env a lane walks both ends (units = slots).
//   * per-unit data (contact point, D, reference accelerations, friction, J qacc, J sr) sits in a per-lane column of LDS (SlotMem: field-major,
//     lane-interleaved, conflict-free; a plain array on the host) and is read where a pass needs it: nothing per unit is live in registers
//     between passes, so the solver's register footprint does not depend on NG.
//   * the hinge columns of a unit's point Jacobian are rebuilt from (contact point, joint anchors) with a SCALAR ancestor mask of the unit's body
//     (anc_pack: 8 bits per capsule): non-ancestors get weight 0 instead of a compile-time skip.
//   * the one-group Woodbury correction of the feet-only instantiation works here too (the toggled unit is a per-lane runtime index into the
//     column) -- the unrolled general instantiation never had room for it.
enum SlotField { SF_PX = 0, SF_PZ, SF_D, SF_AN, SF_AT, SF_MU, SF_LT, SF_LN, SF_LVT, SF_LVN, SF_COUNT };
enum SelfField { SR_PX = 0, SR_PZ, SR_NX, SR_NZ, SR_D, SR_AREF, SR_JQ, SR_JV, SR_COUNT };   // a capsule-capsule row: contact point, normal, D, aref, J qacc, J sr
template <class T, class S, bool PAIR>
struct SlotMem {
  static constexpr int NUNIT = PAIR ? S::NG : 2 * S::NG;
#if defined(__HIP_DEVICE_COMPILE__)
  static constexpr int STRIDE = 64;          // lanes of a workgroup of the step kernels that use it
#else
  static constexpr int STRIDE = 1;
#endif
  static constexpr int NSROW = S::NSELF > 0 ? 2 * S::NSELF : 0;            // capsule-capsule rows (hopper): SR_COUNT words each, behind the units
  static constexpr int SELF_BASE = SF_COUNT * NUNIT * STRIDE;
  static constexpr int WORDS = (SF_COUNT * NUNIT + SR_COUNT * NSROW) * STRIDE;   // per workgroup (device) / per env (host)
  T* p;                                      // this lane's word of (field 0, unit 0)
  REX_HD T* unit(unsigned u) const { return p + u * unsigned(STRIDE); }
  static constexpr int off(int f) { return f * NUNIT * STRIDE; }
  REX_HD T* srow(unsigned r) const { return p + SELF_BASE + r * unsigned(STRIDE); }
  static constexpr int soff(int f) { return f * NSROW * STRIDE; }
};
template <class S> constexpr unsigned long long anc_pack() {   // byte g: bit j set <=> body j is geom g's body or an ancestor of it
  unsigned long long r = 0;
  for (int g = 0; g < S::NG; g++) { unsigned m = 0; for (int j = 0; j < S::NB; j++) if (is_anc_or_self<S>(j, S::geom_body[g])) m |= 1u << j; r |= (unsigned long long)m << (8 * g); }
  return r;
}
static_assert(HopperSpec::NB <= 8 && Walker2dSpec::NB <= 8 && HalfCheetahSpec::NB <= 8 && HalfCheetahSpec::NG <= 8, "anc_pack: 8 bodies x 8 capsules");
template <class S> REX_HD unsigned anc_of_geom(unsigned g) { constexpr unsigned long long A = anc_pack<S>(); return (unsigned)(A >> (8u * g)) & 0xffu; }   // (constexpr local: evaluated by the compiler, not walked at run time)
// Root paths of the tree's leaves (hopper: one, {0,1,2,3}; walker2d / half-cheetah: two, {0,1,2,3} and {0,4,5,6}): a unit's body lies on one of
// them, so its Jacobian columns, dot products and Hessian block only run over that path's bodies -- picked by a scalar branch per unit.
template <class S> constexpr unsigned leaf_path(int which) {
  int seen = 0;
  for (int b = 0; b < S::NB; b++) {
    bool leaf = true;
    for (int c = 0; c < S::NB; c++) if (S::parent[c] == b) leaf = false;
    if (!leaf) continue;
    if (seen++ == which) { unsigned m = 0; for (int j = 0; j < S::NB; j++) if (is_anc_or_self<S>(j, b)) m |= 1u << j; return m; }
  }
  return 0u;
}
template <class S> constexpr int leaf_paths() { int n = 0; while (n < 4 && leaf_path<S>(n) != 0u) n++; return n; }
static_assert(leaf_paths<HopperSpec>() == 1 && leaf_paths<Walker2dSpec>() == 2 && leaf_paths<HalfCheetahSpec>() == 2, "list solver: one or two leaf paths");
// f(IC<path mask>) for the leaf path that holds every body of `anc` (wave-uniform choice)
template <class S, class F> REX_HD void on_path(unsigned anc, F&& f) {
  constexpr unsigned P0 = leaf_path<S>(0);
  if constexpr (leaf_paths<S>() == 1) { (void)anc; f(IC<int(P0)>{}); }
  else { constexpr unsigned P1 = leaf_path<S>(1); if ((anc & ~P0) != 0u) f(IC<int(P1)>{}); else f(IC<int(P0)>{}); }
}
// hinge columns of the point Jacobian of a point on the body with ancestor mask `anc`, over the bodies of path SUB (zero for non-ancestors)
template <class T, class S, unsigned SUB>
REX_HD void list_jac(const Kin<T, S>& K, T px, T pz, unsigned anc, T (&jt)[S::NB], T (&jn)[S::NB]) {
  static_for<0, S::NB>([&](auto JJ) { constexpr int j = JJ;
    if constexpr ((SUB >> j) & 1u) {
      const T w = ((anc >> j) & 1u) ? T(S::sgn[j]) : T(0);
      jt[j] = w * (pz - K.A[j][1]); jn[j] = -(w * (px - K.A[j][0])); } });
}
template <class T, class S, unsigned SUB>
REX_HD void list_dot(const T (&jt)[S::NB], const T (&jn)[S::NB], const T (&x)[S::NV], T& t, T& n) {
  t = x[0]; n = x[1];
  static_for<0, S::NB>([&](auto JJ) { constexpr int j = JJ; if constexpr ((SUB >> j) & 1u) { t += jt[j] * x[j + 2]; n += jn[j] * x[j + 2]; } });
}
template <class T, class S, unsigned SUB>
REX_HD void list_accum(const T (&jt)[S::NB], const T (&jn)[S::NB], T ft, T fn, T (&g)[S::NV]) {
  g[0] += ft; g[1] += fn;
  static_for<0, S::NB>([&](auto JJ) { constexpr int j = JJ; if constexpr ((SUB >> j) & 1u) g[j + 2] += jt[j] * ft + jn[j] * fn; });
}
template <class T, class S, unsigned SUB>
REX_HD void list_hess(const T (&jt)[S::NB], const T (&jn)[S::NB], T ctt, T cnt, T cnn, T (&H)[S::NV][S::NV]) {
  T wt[S::NB], wn[S::NB];
  static_for<0, S::NB>([&](auto JJ) { constexpr int j = JJ; if constexpr ((SUB >> j) & 1u) { wt[j] = ctt * jt[j] + cnt * jn[j]; wn[j] = cnt * jt[j] + cnn * jn[j]; } });
  H[0][0] += ctt; H[1][0] += cnt; H[1][1] += cnn;                       // (ut, un) of the root slides: (1, 0), (0, 1)
  static_for<0, S::NB>([&](auto AA) { constexpr int a = AA;
    if constexpr ((SUB >> a) & 1u) {
      H[a + 2][0] += wt[a]; H[a + 2][1] += wn[a];
      static_for<0, a + 1>([&](auto BB) { constexpr int b = BB;
        if constexpr (((SUB >> b) & 1u) && dof_coupled<S>(a + 2, b + 2)) H[a + 2][b + 2] += wt[a] * jt[b] + wn[a] * jn[b]; }); } });
}
// capsule-capsule rows in the list solver: row = n . (J_b - J_a) at the contact point; the root slides cancel and hinge j carries
// sgn_j ([j on b's root path] - [j on a's]) (n_x r_z - n_z r_x).  Ancestor masks of the two bodies of self pair q: bytes 2 q, 2 q + 1.
template <class S> constexpr unsigned long long self_pack() {
  unsigned long long r = 0;
  for (int q = 0; q < S::NSELF; q++) {
    unsigned ma = 0, mb = 0;
    for (int j = 0; j < S::NB; j++) { if (is_anc_or_self<S>(j, S::geom_body[S::self_a[q]])) ma |= 1u << j; if (is_anc_or_self<S>(j, S::geom_body[S::self_b[q]])) mb |= 1u << j; }
    r |= (unsigned long long)ma << (16 * q); r |= (unsigned long long)mb << (16 * q + 8);
  }
  return r;
}
template <class T, class S>
REX_HD void self_row(const Kin<T, S>& K, T px, T pz, T nx, T nz, unsigned r, T (&row)[S::NB]) {
  constexpr unsigned long long A = self_pack<S>();
  const unsigned ma = (unsigned)(A >> (16u * (r >> 1))) & 0xffu, mb = (unsigned)(A >> (16u * (r >> 1) + 8u)) & 0xffu;
  static_for<0, S::NB>([&](auto JJ) { constexpr int j = JJ;
    const T w = T(S::sgn[j]) * (T(int((mb >> j) & 1u)) - T(int((ma >> j) & 1u)));
    row[j] = w * (nx * (pz - K.A[j][1]) - nz * (px - K.A[j][0])); });
}
template <class T, class S, bool PAIR>
REX_HD void list_store_self(const SelfRows<T, S>& R, const SlotMem<T, S, PAIR>& L) {
  using LM = SlotMem<T, S, PAIR>;
  if constexpr (S::NSELF > 0) static_for<0, LM::NSROW>([&](auto RR) { constexpr int r = RR;
    T* q = L.srow(r);
    q[LM::soff(SR_PX)] = R.px[r]; q[LM::soff(SR_PZ)] = R.pz[r]; q[LM::soff(SR_NX)] = R.nx[r]; q[LM::soff(SR_NZ)] = R.nz[r];
    q[LM::soff(SR_D)] = R.D[r]; q[LM::soff(SR_AREF)] = R.aref[r]; });
}
// slot_rows' results (registers, compile-time indices) into the column; friction per unit
template <class T, class S, bool PAIR>
REX_HD void list_store(const Constraints<T, S>& C, const LaneParams<T, S>& P, const SlotMem<T, S, PAIR>& L) {
  using LM = SlotMem<T, S, PAIR>;
  static_for<0, LM::NUNIT>([&](auto UU) { constexpr int u = UU; constexpr int k = PAIR ? 2 * u : u; constexpr int g = PAIR ? u : u / 2;
    T* q = L.unit(u);
    q[LM::off(SF_PX)] = C.px[k]; q[LM::off(SF_PZ)] = C.pz[k]; q[LM::off(SF_D)] = C.D[k]; q[LM::off(SF_AN)] = C.an[k]; q[LM::off(SF_AT)] = C.at[k];
    q[LM::off(SF_MU)] = P.mu[g]; });
}
template <class T, class S, bool SELF, bool PAIR, int MAXIT = 24>
REX_HD SolveStats solve_newton_list(const T (&M)[S::NV][S::NV], const T (&qfrc_smooth)[S::NV], const T (&qacc_smooth)[S::NV],
                                    const Kin<T, S>& K, const Constraints<T, S>& C, unsigned self_mask, const SlotMem<T, S, PAIR>& L,
                                    T (&qacc)[S::NV], bool warm, bool have_a0, int ls_max, int ls_free, int corr) {
  // self_mask: this lane's capsule-capsule rows (their data is in the column: list_store_self); replicated in both lanes of a pair
  using LM = SlotMem<T, S, PAIR>;
  constexpr int NU_ = LM::NUNIT;
  struct { unsigned mask; } R{SELF ? self_mask : 0u};
  unsigned ums_all = 0u;   // the capsule-capsule rows some lane of the wave has (wave-uniform)
  if constexpr (SELF && S::NSELF > 0) {
    static_for<0, LM::NSROW>([&](auto RR) { constexpr int r = RR; ums_all |= REX_WAVE_ANY(((R.mask >> r) & 1u) != 0u) ? (1u << r) : 0u; });
#if defined(__HIP_DEVICE_COMPILE__)
    ums_all = __builtin_amdgcn_readfirstlane(ums_all);
#endif
  }
  const unsigned par = PAIR ? pair_parity() : 0u;
  auto bit_of = [&](unsigned u) { return PAIR ? 2u * u + par : u; };   // slot bit (in con_mask / the edge sets) of unit u in THIS lane
  auto anc_of = [&](unsigned u) { return anc_of_geom<S>(PAIR ? u : u >> 1); };
  const bool has_rows = C.any || (SELF && R.mask != 0u);
  static_for<0, S::NV>([&](auto II) { T prev = qacc[II], cold = qacc_smooth[II]; opaque(prev); opaque(cold);
                                      qacc[II] = have_a0 ? ((warm && has_rows) ? prev : cold) : prev; });
  SolveStats st{0, false, 0};
  const T tol2 = sizeof(T) == 4 ? T(1e-9) : T(1e-24);
  const T stag = sizeof(T) == 4 ? T(1e-6) : T(1e-15);
  unsigned p_lim = ~0u, p_e1 = ~0u, p_e2 = ~0u, p_e3 = ~0u, p_self = ~0u;
  bool lane_done = !has_rows && have_a0;
  // the list: units some lane of the wave has inside the margin (wave-uniform; on the device a scalar register)
  unsigned um_all = 0u;
  static_for<0, NU_>([&](auto UU) { constexpr int u = UU; constexpr int k = PAIR ? 2 * u : u;
    um_all |= REX_WAVE_ANY(((C.con_mask >> k) & (PAIR ? 3u : 1u)) != 0u) ? (1u << u) : 0u; });
#if defined(__HIP_DEVICE_COMPILE__)
  um_all = __builtin_amdgcn_readfirstlane(um_all);
#endif
  T Ma[S::NV];
  sym_matvec<T, S>(M, qacc, Ma);
  bool ma_dirty = false;
  for (int it = 0; it < MAXIT; ++it) {
    if (!REX_WAVE_ANY(!lane_done)) break;
    // A later iteration walks the units / capsule-capsule rows of the lanes that are STILL ITERATING only: a wave repeats a pass for one or two
    // of its lanes, and the pass costs per listed unit.  (A unit another lane owns adds exact zeros to this lane's sums: same bits either way.)
    // Measured (step kernel, 32 768 envs): half-cheetah 0.1031 -> 0.0956 ms, C3 0.1008 -> 0.0940, walker2d 0.1981 -> 0.1931; the hopper, whose waves
    // are on this solver in 3 % of their evaluations, only pays for the ballots (0.0795 -> 0.0807) and keeps the solve's list.
    unsigned um = um_all, ums = ums_all;
    if (S::KIND != 1 && it > 0) {
      um = 0u;
      static_for<0, NU_>([&](auto UU) { constexpr int u = UU; constexpr int k = PAIR ? 2 * u : u;
        um |= REX_WAVE_ANY(!lane_done && ((C.con_mask >> k) & (PAIR ? 3u : 1u)) != 0u) ? (1u << u) : 0u; });
      if constexpr (SELF && S::NSELF > 0) {
        ums = 0u;
        static_for<0, LM::NSROW>([&](auto RR) { constexpr int r = RR; ums |= REX_WAVE_ANY(!lane_done && ((R.mask >> r) & 1u) != 0u) ? (1u << r) : 0u; });
      }
#if defined(__HIP_DEVICE_COMPILE__)
      um = __builtin_amdgcn_readfirstlane(um); ums = __builtin_amdgcn_readfirstlane(ums);
#endif
    }
    if (ma_dirty) { sym_matvec<T, S>(M, qacc, Ma); ma_dirty = false; }
    REX_COUNT(pass1, 1);
    // ---- gradient, active edges and the Hessian's unit blocks in ONE walk over the list --------------------------------------------------------
    // (the Hessian block of a unit needs only that unit's own active edges; a solve whose last walk finds every lane converged built its
    // blocks for nothing, which is rare here: solves end on an exact step or an accepted correction)
    T g[S::NV];
    T fref = T(0);
    static_for<0, S::NV>([&](auto II) { constexpr int i = II; g[i] = Ma[i] - qfrc_smooth[i]; fref += Ma[i] * Ma[i] + qfrc_smooth[i] * qfrc_smooth[i]; });
    unsigned lim_on = 0, e1 = 0, e2 = 0, e3 = 0, self_on = 0;
    static_for<1, S::NB>([&](auto JJ) { constexpr int j = JJ;
      if constexpr (S::limited[j]) {
        T jar = C.lsig[j] * qacc[j + 2] - C.laref[j];
        bool on = ((C.lim_mask >> j) & 1u) && jar < T(0);
        if (on) lim_on |= 1u << j;
        g[j + 2] += on ? C.lsig[j] * C.lD[j] * jar : T(0); } });
    T Hs[S::NV][S::NV];   // this lane's unit blocks (PAIR: summed over the pair below)
    static_for<0, S::NV>([&](auto II) { constexpr int i = II;
      static_for<0, i + 1>([&](auto JJ) { constexpr int j = JJ; if constexpr (dof_coupled<S>(i, j)) Hs[i][j] = T(0); }); });
    T Hself[(SELF && S::NSELF > 0) ? S::NV : 1][(SELF && S::NSELF > 0) ? S::NV : 1];   // capsule-capsule rows' terms (replicated: NOT summed over the pair)
    if constexpr (SELF && S::NSELF > 0) static_for<2, S::NV>([&](auto II) { constexpr int i = II;
      static_for<2, i + 1>([&](auto JJ) { constexpr int j = JJ; if constexpr (dof_coupled<S>(i, j)) Hself[i][j] = T(0); }); });
    {
      T gs[S::NV];
      static_for<0, S::NV>([&](auto II) { gs[II] = T(0); });
      for (unsigned m = um; m; m &= m - 1u) {
        const unsigned u = (unsigned)__builtin_ctz(m);
        T* q = L.unit(u);
        const T px = q[LM::off(SF_PX)], pz = q[LM::off(SF_PZ)], D = q[LM::off(SF_D)], an = q[LM::off(SF_AN)], at = q[LM::off(SF_AT)], mu = q[LM::off(SF_MU)];
        const unsigned b = bit_of(u), anc = anc_of(u);
        const bool act = (C.con_mask >> b) & 1u;
        on_path<S>(anc, [&](auto PC) { constexpr unsigned SUB = unsigned(int(PC));
          T jt[S::NB], jn[S::NB];
          list_jac<T, S, SUB>(K, px, pz, anc, jt, jn);
          T t, n; list_dot<T, S, SUB>(jt, jn, qacc, t, n);
          q[LM::off(SF_LT)] = t; q[LM::off(SF_LN)] = n;                 // J qacc: read again by the line-search walk
          const T r1 = n + mu * t - (an + at), r2 = n - mu * t - (an - at), r3 = n - an;
          const bool s1 = act && r1 < T(0), s2 = act && r2 < T(0), s3 = act && r3 < T(0);
          e1 |= s1 ? (1u << b) : 0u; e2 |= s2 ? (1u << b) : 0u; e3 |= s3 ? (1u << b) : 0u;
          const T f1 = s1 ? -D * r1 : T(0), f2 = s2 ? -D * r2 : T(0), f3 = s3 ? -D * r3 : T(0);
          list_accum<T, S, SUB>(jt, jn, -(mu * (f1 - f2)), -(f1 + f2 + T(2) * f3), gs);
          const T c1 = s1 ? D : T(0), c2 = s2 ? D : T(0), c3 = s3 ? D : T(0);
          list_hess<T, S, SUB>(jt, jn, mu * mu * (c1 + c2), mu * (c1 - c2), c1 + c2 + T(2) * c3, Hs); });
      }
      if constexpr (PAIR) {
        static_for<0, S::NV>([&](auto II) { constexpr int i = II; g[i] += gs[i] + pair_xchg(gs[i]); });
        e1 |= pair_xchg(e1); e2 |= pair_xchg(e2); e3 |= pair_xchg(e3);
      } else static_for<0, S::NV>([&](auto II) { constexpr int i = II; g[i] += gs[i]; });
    }
    if constexpr (SELF && S::NSELF > 0) {   // capsule-capsule rows: gradient and (rank-1) Hessian terms, replicated in both lanes of a pair
      for (unsigned m = ums; m; m &= m - 1u) {
        const unsigned r = (unsigned)__builtin_ctz(m);
        T* q = L.srow(r);
        T row[S::NB];
        self_row<T, S>(K, q[LM::soff(SR_PX)], q[LM::soff(SR_PZ)], q[LM::soff(SR_NX)], q[LM::soff(SR_NZ)], r, row);
        const T D = q[LM::soff(SR_D)];
        T jq = T(0);
        static_for<0, S::NB>([&](auto JJ) { constexpr int j = JJ; jq += row[j] * qacc[j + 2]; });
        q[LM::soff(SR_JQ)] = jq;
        const T jar = jq - q[LM::soff(SR_AREF)];
        const bool on = ((R.mask >> r) & 1u) && jar < T(0);
        self_on |= on ? (1u << r) : 0u;
        const T w = on ? D * jar : T(0), d = on ? D : T(0);
        static_for<0, S::NB>([&](auto AA) { constexpr int a = AA;
          g[a + 2] += row[a] * w;
          static_for<0, a + 1>([&](auto BB) { constexpr int bq = BB;
            if constexpr (dof_coupled<S>(a + 2, bq + 2)) Hself[a + 2][bq + 2] += d * row[a] * row[bq]; }); });
      }
    }
    T gn = T(0);
    static_for<0, S::NV>([&](auto II) { gn += g[II] * g[II]; });
    const bool same_set = lim_on == p_lim && e1 == p_e1 && e2 == p_e2 && e3 == p_e3 && self_on == p_self;
    p_lim = lim_on; p_e1 = e1; p_e2 = e2; p_e3 = e3; p_self = self_on;
    lane_done = lane_done || same_set || !(gn > tol2 * fref);   // NaN counts as done
    if (!REX_WAVE_ANY(!lane_done)) break;
    REX_COUNT(pass2, 1);
    // ---- Hessian of the current active set, Newton direction ----------------------------------------------------------------------------------
    T H[S::NV][S::NV];
    static_for<0, S::NV>([&](auto II) { constexpr int i = II;
      static_for<0, i + 1>([&](auto JJ) { constexpr int j = JJ;
        if constexpr (dof_coupled<S>(i, j)) { if constexpr (PAIR) H[i][j] = M[i][j] + (Hs[i][j] + pair_xchg(Hs[i][j])); else H[i][j] = M[i][j] + Hs[i][j]; } }); });
    static_for<1, S::NB>([&](auto JJ) { constexpr int j = JJ;
      if constexpr (S::limited[j]) H[j + 2][j + 2] += ((lim_on >> j) & 1u) ? C.lD[j] : T(0); });
    if constexpr (SELF && S::NSELF > 0) static_for<2, S::NV>([&](auto II) { constexpr int i = II;
      static_for<2, i + 1>([&](auto JJ) { constexpr int j = JJ; if constexpr (dof_coupled<S>(i, j)) H[i][j] += Hself[i][j]; }); });
    ldl_factor<T, S>(H);
    T sr[S::NV];
    static_for<0, S::NV>([&](auto II) { sr[II] = -g[II]; });
    ldl_solve<T, S>(H, sr);
    if constexpr (SELF && S::NSELF > 0) {   // J sr of the capsule-capsule rows into the column
      for (unsigned m = ums; m; m &= m - 1u) {
        const unsigned r = (unsigned)__builtin_ctz(m);
        T* q = L.srow(r);
        T row[S::NB];
        self_row<T, S>(K, q[LM::soff(SR_PX)], q[LM::soff(SR_PZ)], q[LM::soff(SR_NX)], q[LM::soff(SR_NZ)], r, row);
        T jv = T(0);
        static_for<0, S::NB>([&](auto JJ) { constexpr int j = JJ; jv += row[j] * sr[j + 2]; });
        q[LM::soff(SR_JV)] = jv;
      }
    }
    // ---- line search on phi(alpha): J sr of every listed unit into the column and phi'(1) in the same walk ------------------------------------
    T Ms[S::NV];
    sym_matvec<T, S>(M, sr, Ms);
    T q1 = T(0), q2 = T(0), d0 = T(0);
    static_for<0, S::NV>([&](auto II) { q1 += sr[II] * (Ma[II] - qfrc_smooth[II]); q2 += sr[II] * Ms[II]; d0 += sr[II] * g[II]; });
    unsigned m_lim, m_e1, m_e2, m_e3, m_self;
    // phi'(a), phi''(a) without the units' part, and the limit / self rows active at a
    auto deriv_head = [&](T a, T& d1, T& d2) {
      REX_COUNT(ls_evals, 1);
      m_lim = m_e1 = m_e2 = m_e3 = m_self = 0u;
      d1 = q1 + a * q2; d2 = q2;
      static_for<1, S::NB>([&](auto JJ) { constexpr int j = JJ;
        if constexpr (S::limited[j]) {
          T lr = C.lsig[j] * qacc[j + 2] - C.laref[j], lv = C.lsig[j] * sr[j + 2];
          T x = lr + a * lv; bool on = ((C.lim_mask >> j) & 1u) && x < T(0);
          if (on) m_lim |= 1u << j;
          T dd = on ? C.lD[j] : T(0); d1 += dd * x * lv; d2 += dd * lv * lv; } });
      if constexpr (SELF && S::NSELF > 0) {
        for (unsigned m = ums; m; m &= m - 1u) {
          const unsigned r = (unsigned)__builtin_ctz(m);
          const T* q = L.srow(r);
          const T vv = q[LM::soff(SR_JV)], x = q[LM::soff(SR_JQ)] - q[LM::soff(SR_AREF)] + a * vv;
          const bool on = ((R.mask >> r) & 1u) && x < T(0);
          m_self |= on ? (1u << r) : 0u;
          const T dd = on ? q[LM::soff(SR_D)] : T(0);
          d1 += dd * x * vv; d2 += dd * vv * vv;
        }
      }
    };
    // one unit's part of phi', phi'' and of the sets at a, from (J qacc, J sr)
    auto deriv_unit = [&](T a, unsigned b, bool act, T D, T an, T at, T mu, T jt, T jn, T vt, T vn, T& d1s, T& d2s) {
      const T r0 = jn + mu * jt - (an + at), r1 = jn - mu * jt - (an - at), r2 = jn - an;
      const T v0 = vn + mu * vt, v1 = vn - mu * vt, v2 = vn;
      const T x0 = r0 + a * v0, x1 = r1 + a * v1, x2 = r2 + a * v2;
      const bool o0 = act && x0 < T(0), o1 = act && x1 < T(0), o2 = act && x2 < T(0);
      m_e1 |= o0 ? (1u << b) : 0u; m_e2 |= o1 ? (1u << b) : 0u; m_e3 |= o2 ? (1u << b) : 0u;
      const T w0 = o0 ? D : T(0), w1 = o1 ? D : T(0), w2 = o2 ? T(2) * D : T(0);
      d1s += w0 * x0 * v0 + w1 * x1 * v1 + w2 * x2 * v2; d2s += w0 * v0 * v0 + w1 * v1 * v1 + w2 * v2 * v2;
    };
    auto deriv_tail = [&](T& d1, T& d2, T d1s, T d2s) {
      if constexpr (PAIR) {
        d1 += d1s + pair_xchg(d1s); d2 += d2s + pair_xchg(d2s);
        m_e1 |= pair_xchg(m_e1); m_e2 |= pair_xchg(m_e2); m_e3 |= pair_xchg(m_e3);
      } else { d1 += d1s; d2 += d2s; }
    };
    const T d1ref = abs_t(d0) * T(sizeof(T) == 4 ? 1e-5 : 1e-13) + T(1e-30);
    T a = T(1), lo = T(0), hi = T(-1), d1, d2;
    {
      deriv_head(a, d1, d2);
      T d1s = T(0), d2s = T(0);
      for (unsigned m = um; m; m &= m - 1u) {
        const unsigned u = (unsigned)__builtin_ctz(m);
        T* q = L.unit(u);
        const T px = q[LM::off(SF_PX)], pz = q[LM::off(SF_PZ)], D = q[LM::off(SF_D)], an = q[LM::off(SF_AN)], at = q[LM::off(SF_AT)], mu = q[LM::off(SF_MU)];
        const T t = q[LM::off(SF_LT)], n = q[LM::off(SF_LN)];
        const unsigned b = bit_of(u), anc = anc_of(u);
        const bool act = (C.con_mask >> b) & 1u;
        on_path<S>(anc, [&](auto PC) { constexpr unsigned SUB = unsigned(int(PC));
          T jt[S::NB], jn[S::NB];
          list_jac<T, S, SUB>(K, px, pz, anc, jt, jn);
          T vt, vn; list_dot<T, S, SUB>(jt, jn, sr, vt, vn);
          q[LM::off(SF_LVT)] = vt; q[LM::off(SF_LVN)] = vn;
          deriv_unit(a, b, act, D, an, at, mu, t, n, vt, vn, d1s, d2s); });
      }
      deriv_tail(d1, d2, d1s, d2s);
    }
    auto deriv = [&](T aa, T& dd1, T& dd2) {   // phi' at another alpha: everything it needs is in the column
      deriv_head(aa, dd1, dd2);
      T d1s = T(0), d2s = T(0);
      for (unsigned m = um; m; m &= m - 1u) {
        const unsigned u = (unsigned)__builtin_ctz(m);
        const T* q = L.unit(u);
        const unsigned b = bit_of(u);
        deriv_unit(aa, b, (C.con_mask >> b) & 1u, q[LM::off(SF_D)], q[LM::off(SF_AN)], q[LM::off(SF_AT)], q[LM::off(SF_MU)],
                   q[LM::off(SF_LT)], q[LM::off(SF_LN)], q[LM::off(SF_LVT)], q[LM::off(SF_LVN)], d1s, d2s);
      }
      deriv_tail(dd1, dd2, d1s, d2s);
    };
    bool ls_done = lane_done || abs_t(d1) <= d1ref;
    const int ls_cap = it < ls_free ? 0 : ls_max;
    for (int ls = 0; ls < ls_cap; ++ls) {
      if (!REX_WAVE_ANY(!ls_done)) break;
      if (d1 < T(0)) lo = a; else hi = a;
      T an_ = a - d1 * rcp_t(d2);
      if (hi >= T(0) && (an_ <= lo || an_ >= hi)) an_ = T(0.5) * (lo + hi);
      an_ = max_t(an_, lo);
      T prev = a;
      a = ls_done ? a : an_;
      deriv(a, d1, d2);
      ls_done = ls_done || abs_t(d1) <= d1ref || a == prev;
    }
    const bool exact_step = a == T(1) && m_lim == lim_on && m_e1 == e1 && m_e2 == e2 && m_e3 == e3 && m_self == self_on;
    a = lane_done ? T(0) : a;
    T amax = T(0), smax = T(0);
    static_for<0, S::NV>([&](auto II) { qacc[II] += a * sr[II]; Ma[II] += a * Ms[II]; amax = max_t(amax, abs_t(qacc[II])); smax = max_t(smax, abs_t(a * sr[II])); });
    lane_done = lane_done || exact_step || smax <= stag * (T(1) + amax);
    // ---- one-group correction (see solve_newton): one joint limit or the edges of ONE unit toggled along a full step ----------------------
    {
#if defined(REX_STATS) && !defined(__HIP_DEVICE_COMPILE__)
      { const int nl = __builtin_popcount(lim_on ^ m_lim), ns = __builtin_popcount((e1 ^ m_e1) | (e2 ^ m_e2) | (e3 ^ m_e3));
        if (!lane_done && a == T(1)) gstats().toggles[nl < 3 ? nl : 3][ns < 3 ? ns : 3]++; }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
      constexpr int CMODE = PAIR ? 1 : 0;
#else
      constexpr int CMODE = 2;
#endif
      constexpr bool SPLIT = CMODE != 0;
      const CorrPlan<SPLIT> plan(lim_on ^ m_lim, (e1 ^ m_e1) | (e2 ^ m_e2) | (e3 ^ m_e3), corr, !lane_done && a == T(1) && m_self == self_on);
      bool can = plan.can;
      if (REX_WAVE_ANY(can)) {
        const bool two = SPLIT && REX_WAVE_ANY(plan.two);
        auto build = [&](unsigned vp, CorrGroup<T, S>& g) {
          const unsigned my_lim = plan.my_lim(vp), my_slots = plan.my_slots(vp);
          static_for<0, S::NV>([&](auto II) { g.Ut[II] = T(0); g.Un[II] = T(0); });
          g.ctt = g.ctn = g.cnn = g.wt = g.wn = T(0);
          static_for<1, S::NB>([&](auto JJ) { constexpr int j = JJ;
            if constexpr (S::limited[j]) {
              const bool b = (my_lim >> j) & 1u;
              const T sg = ((m_lim >> j) & 1u) ? T(1) : T(-1);
              const T xr = C.lsig[j] * qacc[j + 2] - C.laref[j];
              g.Ut[j + 2] = b ? C.lsig[j] : T(0); g.ctt += b ? sg * C.lD[j] : T(0); g.wt += b ? sg * C.lD[j] * xr : T(0); } });
          // the toggled unit of this (virtual) lane: a per-lane index into the column (lanes without one read unit 0 and contribute nothing)
          const bool mine = my_slots != 0u;
          const unsigned kb = (unsigned)__builtin_ctz(my_slots | 0x80000000u) & 31u;                // slot bit
          const unsigned u = mine ? (PAIR ? kb >> 1 : kb) : 0u;
          const T* q = L.unit(u);
          const T px = q[LM::off(SF_PX)], pz = q[LM::off(SF_PZ)], Dk = q[LM::off(SF_D)], an = q[LM::off(SF_AN)], at = q[LM::off(SF_AT)], mu = q[LM::off(SF_MU)];
          // (J qacc / J sr of unit 0 were never written if it is not on the list: select, do not multiply by zero -- 0 * garbage is NaN)
          const T jt1 = mine ? q[LM::off(SF_LT)] + q[LM::off(SF_LVT)] : T(0), jn1 = mine ? q[LM::off(SF_LN)] + q[LM::off(SF_LVN)] : T(0);   // J x1 (alpha = 1)
          T jt[S::NB], jn[S::NB];
          list_jac<T, S, (1u << S::NB) - 1u>(K, px, pz, anc_of(u), jt, jn);   // (per-lane unit: every body, per-lane weights)
          const T x0 = jn1 + mu * jt1 - (an + at), x1 = jn1 - mu * jt1 - (an - at), x2 = jn1 - an;
          const unsigned ks = mine ? kb : 0u;
          const T d1_ = T(int((m_e1 >> ks) & 1u) - int((e1 >> ks) & 1u)), d2_ = T(int((m_e2 >> ks) & 1u) - int((e2 >> ks) & 1u)),
                  d3_ = T(int((m_e3 >> ks) & 1u) - int((e3 >> ks) & 1u));
          const T on = mine ? T(1) : T(0), Dm = mine ? Dk : T(0);
          g.ctt += Dm * mu * mu * (d1_ + d2_); g.ctn += Dm * mu * (d1_ - d2_); g.cnn += Dm * (d1_ + d2_ + T(2) * d3_);
          g.wt += Dm * mu * (d1_ * x0 - d2_ * x1); g.wn += Dm * (d1_ * x0 + d2_ * x1 + T(2) * d3_ * x2);
          list_accum<T, S, (1u << S::NB) - 1u>(jt, jn, on, T(0), g.Ut); list_accum<T, S, (1u << S::NB) - 1u>(jt, jn, T(0), on, g.Un);
        };
        T dx[S::NV];
        can = corr_step<T, S, CMODE>(H, two, can, par, build, dx);
        static_for<0, S::NV>([&](auto II) { constexpr int i = II; qacc[i] -= dx[i]; });
        // the set at x2
        unsigned v_lim = 0u, v1 = 0u, v2 = 0u, v3 = 0u;
        static_for<1, S::NB>([&](auto JJ) { constexpr int j = JJ;
          if constexpr (S::limited[j]) { const T x = C.lsig[j] * qacc[j + 2] - C.laref[j]; if (((C.lim_mask >> j) & 1u) && x < T(0)) v_lim |= 1u << j; } });
        for (unsigned m = um; m; m &= m - 1u) {
          const unsigned u = (unsigned)__builtin_ctz(m);
          const T* q = L.unit(u);
          const T px = q[LM::off(SF_PX)], pz = q[LM::off(SF_PZ)], an = q[LM::off(SF_AN)], at = q[LM::off(SF_AT)], mu = q[LM::off(SF_MU)];
          const T jt1 = q[LM::off(SF_LT)] + q[LM::off(SF_LVT)], jn1 = q[LM::off(SF_LN)] + q[LM::off(SF_LVN)];
          const unsigned b = bit_of(u), anc = anc_of(u);
          const bool act = (C.con_mask >> b) & 1u;
          on_path<S>(anc, [&](auto PC) { constexpr unsigned SUB = unsigned(int(PC));
            T jt[S::NB], jn[S::NB];
            list_jac<T, S, SUB>(K, px, pz, anc, jt, jn);
            T ut, un; list_dot<T, S, SUB>(jt, jn, dx, ut, un);
            const T jt2 = jt1 - ut, jn2 = jn1 - un;
            const T x0 = jn2 + mu * jt2 - (an + at), x1 = jn2 - mu * jt2 - (an - at), x2 = jn2 - an;
            v1 |= (act && x0 < T(0)) ? (1u << b) : 0u; v2 |= (act && x1 < T(0)) ? (1u << b) : 0u; v3 |= (act && x2 < T(0)) ? (1u << b) : 0u; });
        }
        if constexpr (PAIR) { v1 |= pair_xchg(v1); v2 |= pair_xchg(v2); v3 |= pair_xchg(v3); }
        bool ok2 = v_lim == m_lim && v1 == m_e1 && v2 == m_e2 && v3 == m_e3;
        if constexpr (SELF && S::NSELF > 0) {   // a lane with self rows: they must stay as they were at x1 (their toggles are not part of the group)
          for (unsigned m = ums; m; m &= m - 1u) {
            const unsigned r = (unsigned)__builtin_ctz(m);
            const T* q = L.srow(r);
            T row[S::NB];
            self_row<T, S>(K, q[LM::soff(SR_PX)], q[LM::soff(SR_PZ)], q[LM::soff(SR_NX)], q[LM::soff(SR_NZ)], r, row);
            T jar = -q[LM::soff(SR_AREF)];
            static_for<0, S::NB>([&](auto JJ) { constexpr int j = JJ; jar += row[j] * qacc[j + 2]; });
            ok2 = ok2 && ((((R.mask >> r) & 1u) && jar < T(0)) == (((m_self >> r) & 1u) != 0u));
          }
        }
#if defined(REX_STATS) && !defined(__HIP_DEVICE_COMPILE__)
        if (can) { gstats().toggles[3][3]++; if (ok2) gstats().toggles[3][2]++; }   // corrections tried / accepted
#endif
        lane_done = lane_done || (can && ok2);
        p_lim = can ? m_lim : p_lim; p_e1 = can ? m_e1 : p_e1; p_e2 = can ? m_e2 : p_e2; p_e3 = can ? m_e3 : p_e3; p_self = can ? m_self : p_self;
        ma_dirty = true;
        REX_COUNT(nocon, 1);
      }
    }
    st.iters = it + 1;
    if (it == MAXIT - 1) st.capped = REX_WAVE_ANY(!lane_done);
  }
  return st;
}

#if defined(REX_KTIME) && defined(__HIP_DEVICE_COMPILE__)
// diagnostic build only: per-phase cycle stamps (s_memtime), summed per wave into g_ktime[]
extern __device__ unsigned long long g_ktime[24 + 72];
#define REX_STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#define REX_TACC(slot, t0, t1) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_ktime[slot], (t1) - (t0)); } while (0)
#else
#define REX_STAMP(var) ((void)0)
#define REX_TACC(slot, t0, t1) ((void)0)
#endif

// one forward-dynamics evaluation: qacc(q, v, ctrl)  ([3P] mj_forward)
// GEN: which code the general solver modes (1, 2) run -- 0 the unrolled per-slot instantiations, 1 the rolled ROW-list solver (the hopper's
// 256-register two-waves-per-SIMD kernel), 2 the LIST solver (per-unit data in `slot_mem`, this lane's column: LDS on the device)
template <class T, class S, bool PAIR = false, int GEN = 0>
REX_HD SolveStats forward(const T (&q)[S::NV], const T (&v)[S::NV], const T (&ctrl)[S::NU], const PlanarGeom<T, S>& G,
                          const LaneParams<T, S>& P, const SolParams<T>& sp, T (&qacc)[S::NV], T (&M)[S::NV][S::NV],
                          bool warm = false, T* slot_mem = nullptr) {
  constexpr bool ROLLED = GEN == 1;
  REX_STAMP(t_0);
  REX_MARK("kinematics");
  REX_PSTAMP(p_0, q[0]);
  Kin<T, S> K;
  kinematics<T, S, PAIR>(q, G, K);
  REX_PSTAMP(p_1, K.c[S::NB - 1] + K.A[S::NB - 1][0] + K.rc[S::NB - 1][1] + K.s[1]);
  REX_STAMP(t_1);
  T f[S::NV], a0[S::NV];
  {
    T bias[S::NV];
    mass_and_bias<T, S>(v, G, P, K, M, bias);
    f[0] = -bias[0]; f[1] = -bias[1]; f[2] = -bias[2];
    static_for<1, S::NB>([&](auto JJ) {
      constexpr int j = JJ;
      T c = min_t(max_t(ctrl[j - 1], T(-1)), T(1));   // ctrlrange -1..1 on every motor of the three XMLs
      f[j + 2] = -G.damping[j] * v[j + 2] - G.stiffness[j] * q[j + 2] - bias[j + 2] + T(S::gear[j - 1]) * c;
    });
  }
  REX_PSTAMP(p_2, f[S::NV - 1] + f[0] + M[S::NV - 1][0] + M[2][2]);
  REX_STAMP(t_2);
  REX_STAMP(t_3);
  REX_MARK("detect");
  Constraints<T, S> C;
  detect_constraints<T, S, PAIR>(q, v, G, sp, K, C);
  REX_PSTAMP(p_3, C.dist[2 * S::NG - 1] + C.lD[S::NB - 1] + T(C.self_possible));
  REX_STAMP(t_4);
  SolveStats st{0, false, 0};
#if defined(REX_KSTATS) && defined(__HIP_DEVICE_COMPILE__)
  { unsigned um = 0; for (int k = 0; k < 2 * S::NG; k++) if (REX_WAVE_ANY((C.con_mask >> k) & 1u)) um |= 1u << k;
    REX_COUNT(solves, 1); if (!REX_WAVE_ANY(C.any)) REX_COUNT(nocon, 1); REX_COUNT(slots_active, __popc(um)); }
#else
  REX_COUNT(solves, 1); if (!C.any) REX_COUNT(nocon, 1); REX_COUNT(slots_active, __builtin_popcount(C.con_mask));
#endif
  // Capsule-capsule self contacts (hopper): the bounding-circle cull is loose -- a sharply folded leg passes it for hundreds of
  // steps without touching -- and the kernel time at B = 32 768 is the SLOWEST wave's, so the dearer solver instantiation is
  // entered only when the narrow phase has actually produced a row somewhere in the wave.
  SelfRows<T, S> R; R.mask = 0u;
  bool self_rows = false;
  if constexpr (S::NSELF > 0) {
    if (REX_WAVE_ANY(C.self_possible != 0u)) {
#if defined(REX_WAVETIME) && defined(__HIP_DEVICE_COMPILE__)
      REX_COUNT(selfpath, 1);
#endif
      make_self_rows<T, S>(v, G, sp, K, C.self_possible, R);
      self_rows = REX_WAVE_ANY(R.mask != 0u);
    }
  }
  // Which solver instantiation the wave enters (wave-uniform):
  //   0  no row in any lane                              -> qacc = qacc_smooth
  //   3  only the feet touch (S::FAST_SLOTS), no self row -> straight-line instantiation over those slots: no per-slot branch,
  //      a quarter of the slot state live -- the common case of an upright walker, and the launch ends with its slowest wave
  //   1  any other floor slot                            -> general instantiation (wave-uniform skipping of untouched slots)
  //   2  a capsule-capsule self row (hopper)             -> general instantiation + self rows
  constexpr unsigned ALL = all_slots<S>(), FAST = S::FAST_SLOTS & ALL;
  const bool off_fast = REX_WAVE_ANY((C.con_mask & ~FAST) != 0u) || !sp.fast;
  int mode = self_rows ? 2 : (REX_WAVE_ANY(C.any) ? (off_fast ? 1 : 3) : 0);
#if defined(__HIP_DEVICE_COMPILE__)
  mode = __builtin_amdgcn_readfirstlane(mode);   // wave-uniform by construction: make it a scalar branch
#endif
  // qacc_smooth = M^-1 qfrc_smooth ([3P] mj_fwdAcceleration) is the solver's cold start and the answer when no lane of
  // the wave has a row; a warm-started solve does not read it (solve_newton), so 15 of the 16 evaluations of a
  // hopper / walker2d step skip this factorisation.
  const bool have_a0 = mode == 0 || !warm || !sp.fast;
  if (have_a0) {
    T L[S::NV][S::NV];
    static_for<0, S::NV>([&](auto II) { constexpr int i = II; a0[i] = f[i];
      static_for<0, i + 1>([&](auto JJ) { constexpr int j = JJ; if constexpr (dof_coupled<S>(i, j)) L[i][j] = M[i][j]; }); });
    ldl_factor<T, S>(L);
    ldl_solve<T, S>(L, a0);
  } else static_for<0, S::NV>([&](auto II) { a0[II] = T(0); });
  REX_MARK("dispatch");
  REX_PSTAMP(p_4, a0[0] + T(R.mask));
  if (mode == 3) {
    REX_MARK("fast_rows");
    if constexpr (PAIR) {   // two lanes per env: each lane its own end of the feet (even slot 2g holds the own end's data)
      constexpr unsigned FASTP = FAST & 0x55555555u;
      slot_rows<T, S, FASTP, false, true>(v, G, P, sp, K, C);
      st = solve_newton<T, S, false, FASTP, false, true>(M, f, a0, K, C, R, P, qacc, warm, have_a0, sp.ls_max, sp.ls_free, sp.corr);
    } else {
      slot_rows<T, S, FAST, false>(v, G, P, sp, K, C);
      st = solve_newton<T, S, false, FAST, false>(M, f, a0, K, C, R, P, qacc, warm, have_a0, sp.ls_max, sp.ls_free, sp.corr);
    }
  }
#if defined(REX_DIAG_NOGENERAL)   // timing diagnostics only (WRONG results for waves that leave the feet-only path): what the general instantiations cost the kernel by being in it
  else if (mode == 1 || mode == 2) static_for<0, S::NV>([&](auto II) { qacc[II] = a0[II]; });
#else
  else if (GEN == 2 && (mode == 1 || mode == 2)) {   // the LIST solver: one instantiation for both modes (a lane without self rows has R.mask == 0)
    if constexpr (GEN == 2) {
      constexpr unsigned UNITS = PAIR ? (ALL & 0x55555555u) : ALL;   // PAIR: the even slot 2 g holds the lane's own end (detect_constraints<PAIR>)
      slot_rows<T, S, UNITS, true, PAIR>(v, G, P, sp, K, C);
      SlotMem<T, S, PAIR> L;
#if defined(__HIP_DEVICE_COMPILE__)
      L.p = slot_mem;
#else
      T host_col[SlotMem<T, S, PAIR>::WORDS]; L.p = slot_mem ? slot_mem : host_col;
#endif
      list_store<T, S, PAIR>(C, P, L);
      if (mode == 2) list_store_self<T, S, PAIR>(R, L);   // (mode 1: R.mask == 0 in every lane, nothing of R is read)
      st = solve_newton_list<T, S, (S::NSELF > 0), PAIR>(M, f, a0, K, C, R.mask, L, qacc, warm, have_a0, sp.ls_max, sp.ls_free, sp.corr);
    }
  } else if (ROLLED && (mode == 1 || mode == 2)) {   // (one call site for both: the row list carries the self rows when there are any)
    if constexpr (ROLLED) {
      if constexpr (PAIR) detect_constraints<T, S, false>(q, v, G, sp, K, C);
      slot_rows<T, S, ALL, true>(v, G, P, sp, K, C);
      RowList<T, S> RL;
      build_rows<T, S>(K, C, R, P, mode == 2, RL);
      st = solve_newton_rolled<T, S>(M, f, a0, RL, qacc, warm, have_a0, sp.ls_max, sp.ls_free);
    }
  } else if (mode == 2) {
    if constexpr (PAIR) detect_constraints<T, S, false>(q, v, G, sp, K, C);   // the general instantiations run replicated in both lanes of a pair: every slot
    if constexpr (S::NSELF > 0 && GEN == 0) { slot_rows<T, S, ALL, true>(v, G, P, sp, K, C); st = solve_newton<T, S, true, ALL, true>(M, f, a0, K, C, R, P, qacc, warm, have_a0, sp.ls_max, sp.ls_free, sp.corr); }
  } else if (mode == 1) {
    if constexpr (GEN == 0) {
      if constexpr (PAIR) detect_constraints<T, S, false>(q, v, G, sp, K, C);
      slot_rows<T, S, ALL, true>(v, G, P, sp, K, C);
      st = solve_newton<T, S, false, ALL, true>(M, f, a0, K, C, R, P, qacc, warm, have_a0, sp.ls_max, sp.ls_free, sp.corr);
    }
  }
#endif
  else static_for<0, S::NV>([&](auto II) { qacc[II] = a0[II]; });
#if defined(REX_KSTATS) && defined(__HIP_DEVICE_COMPILE__)
  if (mode == 3 && (threadIdx.x & 63) == 0) atomicAdd(&g_kstats[7], 1ull);   // wave-solves on the fast path
#endif
#if defined(REX_WAVETIME) && defined(__HIP_DEVICE_COMPILE__)
  if constexpr (S::NSELF == 0) { if (mode == 1) REX_COUNT(selfpath, 1); }   // (slot "selfpath" of a chain without self pairs: general-path solves)
#endif
  REX_MARK("forward_end");
#if defined(REX_STATS) && !defined(__HIP_DEVICE_COMPILE__)
  gstats().trace[gstats().ntrace++ & 63] = st.iters + 100 * mode;
#endif
  REX_PSTAMP(p_5, qacc[0] + qacc[S::NV - 1]);
  REX_PACC(0, p_0, p_1); REX_PACC(1, p_1, p_2); REX_PACC(2, p_2, p_3); REX_PACC(3, p_3, p_4); REX_PACC(4, p_4, p_5);
  st.mode = mode;
  REX_STAMP(t_5);
  REX_TACC(0, t_0, t_1); REX_TACC(1, t_1, t_2); REX_TACC(2, t_2, t_3); REX_TACC(3, t_3, t_4); REX_TACC(4, t_4, t_5);
#if defined(REX_KTIME) && defined(__HIP_DEVICE_COMPILE__)
  if ((threadIdx.x & 63) == 0) atomicAdd(&g_ktime[7], 1ull);
#endif
  return st;
}

// one mj_step: RK4 ([3P] mj_RungeKutta, N=4) or semi-implicit Euler with implicit joint damping
// ([3P] mj_Euler).  Returns the OR of "solver hit its cap".
template <class T, class S, bool PAIR = false, int GEN = 0>
REX_HD bool substep(T (&q)[S::NV], T (&v)[S::NV], const T (&ctrl)[S::NU], const PlanarGeom<T, S>& G,
                    const LaneParams<T, S>& P, const SolParams<T>& sp, T (&acc)[S::NV], bool warm, T* slot_mem = nullptr) {
  // acc: in = qacc of the previous evaluation (solver warm start when `warm`), out = qacc of the last one
  const T h = T(S::TIMESTEP);
  T M[S::NV][S::NV];
  bool capped = false;
  if constexpr (S::RK4) {
    // stage loop kept rolled: one instance of forward() in the kernel, 4x smaller code and far
    // lower register pressure than four inlined copies
    T q0[S::NV], v0[S::NV], dq[S::NV], dv[S::NV];
    static_for<0, S::NV>([&](auto II) { q0[II] = q[II]; v0[II] = v[II]; dq[II] = T(0); dv[II] = T(0); });
#pragma unroll 1
    for (int stage = 0; stage < 4; ++stage) {
      capped |= forward<T, S, PAIR, GEN>(q, v, ctrl, G, P, sp, acc, M, sp.warm && (warm || stage > 0), slot_mem).capped;
      const T w = (stage == 0 || stage == 3) ? T(1.0 / 6) : T(1.0 / 3);   // B = [1/6 1/3 1/3 1/6]
      const T c = stage == 2 ? h : T(0.5) * h;                             // A = [.5; 0 .5; 0 0 1]
      static_for<0, S::NV>([&](auto II) { constexpr int i = II;
        dq[i] += w * v[i]; dv[i] += w * acc[i];
        T qn = q0[i] + c * v[i], vn = v0[i] + c * acc[i];
        q[i] = stage == 3 ? q0[i] + h * dq[i] : qn;
        v[i] = stage == 3 ? v0[i] + h * dv[i] : vn; });
    }
  } else {
    T rhs[S::NV];
    capped |= forward<T, S, PAIR, GEN>(q, v, ctrl, G, P, sp, acc, M, sp.warm && warm, slot_mem).capped;
    // (M + h*diag(damping)) a = qfrc_smooth + qfrc_constraint = M qacc
    sym_matvec<T, S>(M, acc, rhs);
    static_for<1, S::NB>([&](auto JJ) { constexpr int j = JJ; M[j + 2][j + 2] += h * G.damping[j]; });
    ldl_factor<T, S>(M);
    ldl_solve<T, S>(M, rhs);
    static_for<0, S::NV>([&](auto II) { constexpr int i = II; v[i] += h * rhs[i]; q[i] += h * v[i]; });
  }
  return capped;
}

}  // namespace rex
