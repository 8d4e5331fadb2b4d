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

#include "planar_spec.hpp"

namespace rex {

REX_HD void sincos_t(float a, float& s, float& c) {
#if defined(__HIP_DEVICE_COMPILE__)
  sincosf(a, &s, &c);
#else
  s = sinf(a); c = cosf(a);
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

template <class T, class S>
REX_HD void kinematics(const T (&q)[S::NV], const PlanarGeom<T, S>& G, Kin<T, S>& K) {
  T phi[S::NB];
  static_for<0, S::NB>([&](auto I) {
    constexpr int i = I;
    if constexpr (i == 0) phi[0] = q[2];
    else phi[i] = phi[S::parent[i]] + T(S::sgn[i]) * q[i + 2];
    sincos_t(phi[i], K.s[i], K.c[i]);
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

// In-place sparse L^T D L factorisation (Featherstone): after the call H[k][k] = D_k and
// H[k][i] (i ancestor dof of k) = L_ki.  Branch-induced zeros of the tree are never touched,
// and H = M + J^T D J keeps M's sparsity because every constraint row lives on one root path.
template <class T, class S>
REX_HD void ldl_factor(T (&H)[S::NV][S::NV]) {
  static_rfor<1, S::NV>([&](auto KK) {
    constexpr int k = KK;
    T inv = T(1) / H[k][k];
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
}
template <class T, class S>
REX_HD void ldl_solve(const T (&H)[S::NV][S::NV], T (&b)[S::NV]) {
  static_rfor<1, S::NV>([&](auto KK) {
    constexpr int k = KK;
    static_for<0, k>([&](auto II) { constexpr int i = II; if constexpr (dof_coupled<S>(k, i)) b[i] -= H[k][i] * b[k]; });
  });
  static_for<0, S::NV>([&](auto KK) { constexpr int k = KK; b[k] = b[k] / H[k][k]; });
  static_for<1, S::NV>([&](auto KK) {
    constexpr int k = KK;
    static_for<0, k>([&](auto II) { constexpr int i = II; if constexpr (dof_coupled<S>(k, i)) b[k] -= H[k][i] * b[i]; });
  });
}

// [3P getimpedance] sigmoid impedance, power = 2, midpoint = 0.5 (every solimp in the four XMLs)
template <class T>
REX_HD T impedance(T dmin, T dmax, T width, T x_abs) {
  T x = x_abs / width;
  T y = x < T(0.5) ? T(2) * x * x : T(1) - T(2) * (T(1) - x) * (T(1) - x);
  T imp = dmin + y * (dmax - dmin);
  imp = x >= T(1) ? dmax : imp;
  return (dmin == dmax) ? dmin : imp;
}

// One floor-contact slot (one capsule end).  J rows are never stored: they are rebuilt from the
// contact point and the joint anchors whenever needed.
template <class T>
struct ContactSlot {
  T px, pz;      // contact point relative to the root anchor
  T D;           // 1/R of the pyramid edges
  T an, at;      // reference accelerations: edges are (an +/- at) and an (twice)
  T mu;
  bool active;
};
// One capsule-capsule self-contact (condim 1)
template <class T>
struct SelfSlot {
  T px, pz, nx, nz, D, aref;
  bool active;
};
template <class T>
struct LimitSlot {
  T sigma, D, aref;   // row = sigma * e_dof
  bool active;
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

template <class T, class S>
struct Constraints {
  ContactSlot<T> con[2 * S::NG];
  LimitSlot<T> lim[S::NB];
  SelfSlot<T> self[S::NSELF > 0 ? 2 * S::NSELF : 1];   // up to two contacts per pair (parallel axes)
  bool any;
};

// closest points of two 2-D segments ([3P] mjc_CapsuleCapsule restated in the plane), then
// circle-circle.  Returns up to two contacts in out[0..1].
template <class T>
REX_HD void capsule_capsule_2d(const T (&p1)[2], const T (&a1)[2], T l1, T r1, const T (&p2)[2], const T (&a2)[2],
                               T l2, T r2, T margin, T (&cx)[2], T (&cz)[2], T (&nx)[2], T (&nz)[2], T (&dist)[2],
                               bool (&hit)[2]) {
  hit[0] = hit[1] = false;
  auto sphere = [&](T c1x, T c1z, T c2x, T c2z, int k) {
    T dx = c2x - c1x, dz = c2z - c1z;
    T len = sqrt_t(dx * dx + dz * dz), d = len - r1 - r2;
    if (d > margin) return;
    T ux = T(1), uz = T(0);
    if (len >= T(1e-15)) { ux = dx / len; uz = dz / len; }
    T px_ = c1x + ux * (r1 + d * T(0.5)), pz_ = c1z + uz * (r1 + d * T(0.5));
    if (k == 0) { hit[0] = true; dist[0] = d; nx[0] = ux; nz[0] = uz; cx[0] = px_; cz[0] = pz_; }
    else        { hit[1] = true; dist[1] = d; nx[1] = ux; nz[1] = uz; cx[1] = px_; cz[1] = pz_; }
  };
  T difx = p1[0] - p2[0], difz = p1[1] - p2[1];
  T ma = a1[0] * a1[0] + a1[1] * a1[1], mb = -(a1[0] * a2[0] + a1[1] * a2[1]), mc = a2[0] * a2[0] + a2[1] * a2[1];
  T u = -(a1[0] * difx + a1[1] * difz), v = a2[0] * difx + a2[1] * difz, det = ma * mc - mb * mb;
  if (abs_t(det) >= T(1e-15)) {
    T x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > l1) { x1 = l1; x2 = (v - mb * l1) / mc; } else if (x1 < -l1) { x1 = -l1; x2 = (v + mb * l1) / mc; }
    if (x2 > l2) { x2 = l2; x1 = (u - mb * l2) / ma; } else if (x2 < -l2) { x2 = -l2; x1 = (u + mb * l2) / ma; }
    if (x1 > l1) x1 = l1; else if (x1 < -l1) x1 = -l1;
    sphere(p1[0] + a1[0] * x1, p1[1] + a1[1] * x1, p2[0] + a2[0] * x2, p2[1] + a2[1] * x2, 0);
    return;
  }
  // parallel axes: end points of segment 1 against segment 2, then of 2 against 1; first two hits
  auto put = [&](T c1x, T c1z, T c2x, T c2z) { if (!hit[0]) sphere(c1x, c1z, c2x, c2z, 0); else if (!hit[1]) sphere(c1x, c1z, c2x, c2z, 1); };
  static_for<0, 2>([&](auto SS) {
    constexpr int sg = 2 * int(SS) - 1;
    T c1x = p1[0] + a1[0] * T(sg) * l1, c1z = p1[1] + a1[1] * T(sg) * l1;
    T x2 = (c1x - p2[0]) * a2[0] + (c1z - p2[1]) * a2[1];
    if (x2 >= -l2 && x2 <= l2) put(c1x, c1z, p2[0] + a2[0] * x2, p2[1] + a2[1] * x2);
  });
  static_for<0, 2>([&](auto SS) {
    constexpr int sg = 2 * int(SS) - 1;
    T c2x = p2[0] + a2[0] * T(sg) * l2, c2z = p2[1] + a2[1] * T(sg) * l2;
    T x1 = (c2x - p1[0]) * a1[0] + (c2z - p1[1]) * a1[1];
    if (x1 >= -l1 && x1 <= l1) put(p1[0] + a1[0] * x1, p1[1] + a1[1] * x1, c2x, c2z);
  });
}

// collision + constraint rows + reference accelerations  ([3P] mj_collision, mj_makeConstraint,
// mj_diagApprox, mj_makeImpedance, mj_referenceConstraint)
template <class T, class S>
REX_HD void make_constraints(const T (&q)[S::NV], const T (&v)[S::NV], const PlanarGeom<T, S>& G,
                             const LaneParams<T, S>& P, const SolParams<T>& sp, const Kin<T, S>& K,
                             Constraints<T, S>& C) {
  bool any = false;
  // joint limits (hinge bodies 1..NB-1)
  static_for<1, S::NB>([&](auto JJ) {
    constexpr int j = JJ;
    LimitSlot<T>& L = C.lim[j];
    L.active = false;
    if constexpr (S::limited[j]) {
      T val = q[j + 2];
      T dlo = val - T(S::range_lo[j]), dhi = T(S::range_hi[j]) - val;
      bool lo = dlo < T(0), hi = dhi < T(0);
      T dist = lo ? dlo : dhi;
      L.active = lo || hi;
      L.sigma = lo ? T(1) : T(-1);
      T imp = impedance(sp.lim_dmin, sp.lim_dmax, sp.lim_width, abs_t(dist));
      T R = max_t(T(1e-15), (T(1) - imp) * G.dof_invw[j] / imp);
      L.D = T(1) / R;
      L.aref = -sp.lim_B * (L.sigma * v[j + 2]) - sp.lim_K * imp * dist;
      any = any || L.active;
    }
  });
  // capsule ends against the floor plane z = 0
  static_for<0, S::NG>([&](auto GG) {
    constexpr int g = GG; constexpr int b = S::geom_body[g];
    static_for<0, 2>([&](auto EE) {
      constexpr int e = EE;
      ContactSlot<T>& c = C.con[2 * g + e];
      T lx = e == 0 ? G.e1[g][0] : G.e2[g][0], lz = e == 0 ? G.e1[g][1] : G.e2[g][1];
      T ox, oz; rot(K.c[b], K.s[b], lx, lz, ox, oz);
      T cx = K.A[b][0] + ox, cz = K.A[b][1] + oz;     // sphere centre rel. root anchor
      T dist = (cz + K.zroot) - G.radius[g];
      c.active = dist < sp.con_margin;
      c.px = cx; c.pz = T(0.5) * dist - K.zroot;      // midpoint between the surfaces
      c.mu = P.mu[g];
      T imp = impedance(sp.con_dmin, sp.con_dmax, sp.con_width, abs_t(dist - sp.con_margin));
      T mu2 = c.mu * c.mu;
      T R1 = max_t(T(1e-15), (T(1) - imp) * (G.tran_invw[b] * (T(1) + mu2)) / imp);
      c.D = T(1) / (T(2) * mu2 * R1);
      T vt, vn; jdot<T, S, b>(K, c.px, c.pz, v, vt, vn);
      c.an = -sp.con_B * vn - sp.con_K * imp * (dist - sp.con_margin);
      c.at = -sp.con_B * c.mu * vt;
      any = any || c.active;
    });
  });
  if constexpr (S::NSELF > 0) {
    static_for<0, S::NSELF>([&](auto PP) {
      constexpr int p = PP; constexpr int ga = S::self_a[p], gb = S::self_b[p];
      constexpr int ba = S::geom_body[ga], bb = S::geom_body[gb];
      T e1x, e1z, e2x, e2z, p1[2], a1[2], p2[2], a2[2];
      rot(K.c[ba], K.s[ba], G.e1[ga][0], G.e1[ga][1], e1x, e1z); rot(K.c[ba], K.s[ba], G.e2[ga][0], G.e2[ga][1], e2x, e2z);
      p1[0] = K.A[ba][0] + T(0.5) * (e1x + e2x); p1[1] = K.A[ba][1] + T(0.5) * (e1z + e2z);
      T l1 = T(0.5) * sqrt_t((e1x - e2x) * (e1x - e2x) + (e1z - e2z) * (e1z - e2z));
      a1[0] = T(0.5) * (e1x - e2x) / l1; a1[1] = T(0.5) * (e1z - e2z) / l1;
      rot(K.c[bb], K.s[bb], G.e1[gb][0], G.e1[gb][1], e1x, e1z); rot(K.c[bb], K.s[bb], G.e2[gb][0], G.e2[gb][1], e2x, e2z);
      p2[0] = K.A[bb][0] + T(0.5) * (e1x + e2x); p2[1] = K.A[bb][1] + T(0.5) * (e1z + e2z);
      T l2 = T(0.5) * sqrt_t((e1x - e2x) * (e1x - e2x) + (e1z - e2z) * (e1z - e2z));
      a2[0] = T(0.5) * (e1x - e2x) / l2; a2[1] = T(0.5) * (e1z - e2z) / l2;
      T cx[2], cz[2], nx[2], nz[2], dist[2]; bool hit[2];
      capsule_capsule_2d(p1, a1, l1, G.radius[ga], p2, a2, l2, G.radius[gb], sp.con_margin, cx, cz, nx, nz, dist, hit);
      static_for<0, 2>([&](auto KK) {
        constexpr int k = KK;
        SelfSlot<T>& s = C.self[2 * p + k];
        s.active = hit[k] && dist[k] < sp.con_margin;
        if (s.active) {
          s.px = cx[k]; s.pz = cz[k]; s.nx = nx[k]; s.nz = nz[k];
          T imp = impedance(sp.con_dmin, sp.con_dmax, sp.con_width, abs_t(dist[k] - sp.con_margin));
          T R = max_t(T(1e-15), (T(1) - imp) * (G.tran_invw[ba] + G.tran_invw[bb]) / imp);
          s.D = T(1) / R;
          T ta, na, tb, nb; jdot<T, S, ba>(K, s.px, s.pz, v, ta, na); jdot<T, S, bb>(K, s.px, s.pz, v, tb, nb);
          T vel = s.nx * (tb - ta) + s.nz * (nb - na);
          s.aref = -sp.con_B * vel - sp.con_K * imp * (dist[k] - sp.con_margin);
          any = true;
        } else { s.px = s.pz = s.nx = s.nz = s.D = s.aref = T(0); }
      });
    });
  }
  C.any = any;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define REX_WAVE_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0ull)
#else
#define REX_WAVE_ANY(x) (x)
#endif

struct SolveStats { int iters; bool capped; };

// Primal Newton solve of   min_a 0.5 (a-a0)^T M (a-a0) + sum_rows 0.5 D min(0, J a - aref)^2
// ([3P] engine_solver, Newton, pyramidal cones): exact Hessian M + J^T D_active J with the tree
// sparsity of M, L^T D L factorisation, exact line search on the piecewise-quadratic 1-D cost.
// All lanes of a wave iterate together; a lane that has converged keeps alpha = 0.
template <class T, class S, int MAXIT = 24>
REX_HD SolveStats solve_newton(const T (&M)[S::NV][S::NV], const T (&qfrc_smooth)[S::NV], const T (&qacc_smooth)[S::NV],
                               const Kin<T, S>& K, const Constraints<T, S>& C, const SolParams<T>& sp, T (&qacc)[S::NV]) {
  static_for<0, S::NV>([&](auto II) { qacc[II] = qacc_smooth[II]; });
  SolveStats st{0, false};
  // stop when the force residual |M a - f - J^T f_c| is at rounding level relative to the forces
  // that balance in it (the piecewise-quadratic cost makes Newton exact once the active set is
  // right, so the residual drops from O(1) to rounding in one step)
  const T tol2 = sizeof(T) == 4 ? T(1e-10) : T(1e-24);
  bool lane_done = !C.any;
  for (int it = 0; it < MAXIT; ++it) {
    if (!REX_WAVE_ANY(!lane_done)) break;
    T g[S::NV], Ma[S::NV], H[S::NV][S::NV];
    sym_matvec<T, S>(M, qacc, Ma);
    T fref = T(0);
    static_for<0, S::NV>([&](auto II) {
      constexpr int i = II; g[i] = Ma[i] - qfrc_smooth[i]; fref += Ma[i] * Ma[i] + qfrc_smooth[i] * qfrc_smooth[i];
      static_for<0, i + 1>([&](auto JJ) { constexpr int j = JJ; if constexpr (dof_coupled<S>(i, j)) H[i][j] = M[i][j]; });
    });
    static_for<1, S::NB>([&](auto JJ) {
      constexpr int j = JJ;
      if constexpr (S::limited[j]) {
        const LimitSlot<T>& L = C.lim[j];
        T jar = L.sigma * qacc[j + 2] - L.aref;
        bool on = L.active && jar < T(0);
        T f = on ? -L.D * jar : T(0);
        g[j + 2] -= L.sigma * f;
        H[j + 2][j + 2] += on ? L.D : T(0);
      }
    });
    static_for<0, S::NG>([&](auto GG) {
      constexpr int gg = GG; constexpr int b = S::geom_body[gg];
      static_for<0, 2>([&](auto EE) {
        const ContactSlot<T>& c = C.con[2 * gg + EE];
        if (REX_WAVE_ANY(c.active)) {
          T jt, jn; jdot<T, S, b>(K, c.px, c.pz, qacc, jt, jn);
          T r1 = jn + c.mu * jt - (c.an + c.at), r2 = jn - c.mu * jt - (c.an - c.at), r3 = jn - c.an;
          T s1 = (c.active && r1 < T(0)) ? T(1) : T(0), s2 = (c.active && r2 < T(0)) ? T(1) : T(0);
          T s3 = (c.active && r3 < T(0)) ? T(1) : T(0);
          T f1 = -c.D * r1 * s1, f2 = -c.D * r2 * s2, f3 = -c.D * r3 * s3;
          jt_accum<T, S, b>(K, c.px, c.pz, -(c.mu * (f1 - f2)), -(f1 + f2 + T(2) * f3), g);
          T mu2 = c.mu * c.mu;
          hess_accum<T, S, b>(K, c.px, c.pz, c.D * mu2 * (s1 + s2), c.D * c.mu * (s1 - s2), c.D * (s1 + s2 + T(2) * s3), H);
        }
      });
    });
    if constexpr (S::NSELF > 0) {
      static_for<0, 2 * S::NSELF>([&](auto PP) {
        constexpr int p = PP; constexpr int ba = S::geom_body[S::self_a[p / 2]], bb = S::geom_body[S::self_b[p / 2]];
        const SelfSlot<T>& s = C.self[p];
        if (REX_WAVE_ANY(s.active)) {
          T ta, na, tb, nb; jdot<T, S, ba>(K, s.px, s.pz, qacc, ta, na); jdot<T, S, bb>(K, s.px, s.pz, qacc, tb, nb);
          T jar = s.nx * (tb - ta) + s.nz * (nb - na) - s.aref;
          bool on = s.active && jar < T(0);
          T f = on ? -s.D * jar : T(0);
          jt_accum<T, S, bb>(K, s.px, s.pz, -s.nx * f, -s.nz * f, g);
          jt_accum<T, S, ba>(K, s.px, s.pz, s.nx * f, s.nz * f, g);
          // row = n.(J_b - J_a): build it explicitly, rank-1 update on the (dense) root path union
          T row[S::NV];
          static_for<0, S::NV>([&](auto II) { row[II] = T(0); });
          jt_accum<T, S, bb>(K, s.px, s.pz, s.nx, s.nz, row); jt_accum<T, S, ba>(K, s.px, s.pz, -s.nx, -s.nz, row);
          T d = on ? s.D : T(0);
          static_for<0, S::NV>([&](auto AA) { constexpr int a = AA;
            static_for<0, a + 1>([&](auto BB) { constexpr int bq = BB; if constexpr (dof_coupled<S>(a, bq)) H[a][bq] += d * row[a] * row[bq]; }); });
        }
      });
    }
    T gn = T(0);
    static_for<0, S::NV>([&](auto II) { gn += g[II] * g[II]; });
    lane_done = lane_done || !(gn > tol2 * fref);   // NaN counts as done
    if (!REX_WAVE_ANY(!lane_done)) break;
    // search = -H^-1 g
    ldl_factor<T, S>(H);
    T sr[S::NV];
    static_for<0, S::NV>([&](auto II) { sr[II] = -g[II]; });
    ldl_solve<T, S>(H, sr);
    // line search: phi(alpha) = quad + sum rows
    T Ms[S::NV];
    sym_matvec<T, S>(M, sr, Ms);
    T q1 = T(0), q2 = T(0);   // phi'(0) gauss part, half curvature
    static_for<0, S::NV>([&](auto II) { q1 += sr[II] * (Ma[II] - qfrc_smooth[II]); q2 += sr[II] * Ms[II]; });
    // per-row (jar, jv) pairs
    T lr[S::NB], lv[S::NB];
    static_for<1, S::NB>([&](auto JJ) { constexpr int j = JJ;
      if constexpr (S::limited[j]) { lr[j] = C.lim[j].sigma * qacc[j + 2] - C.lim[j].aref; lv[j] = C.lim[j].sigma * sr[j + 2]; } });
    T cr[2 * S::NG][3], cv[2 * S::NG][3];
    static_for<0, S::NG>([&](auto GG) {
      constexpr int gg = GG; constexpr int b = S::geom_body[gg];
      static_for<0, 2>([&](auto EE) {
        constexpr int k = 2 * gg + EE;
        const ContactSlot<T>& c = C.con[k];
        T jt, jn, vt, vn; jdot<T, S, b>(K, c.px, c.pz, qacc, jt, jn); jdot<T, S, b>(K, c.px, c.pz, sr, vt, vn);
        cr[k][0] = jn + c.mu * jt - (c.an + c.at); cr[k][1] = jn - c.mu * jt - (c.an - c.at); cr[k][2] = jn - c.an;
        cv[k][0] = vn + c.mu * vt; cv[k][1] = vn - c.mu * vt; cv[k][2] = vn;
      });
    });
    T sfr[S::NSELF > 0 ? 2 * S::NSELF : 1], sfv[S::NSELF > 0 ? 2 * S::NSELF : 1];
    if constexpr (S::NSELF > 0) {
      static_for<0, 2 * S::NSELF>([&](auto PP) {
        constexpr int p = PP; constexpr int ba = S::geom_body[S::self_a[p / 2]], bb = S::geom_body[S::self_b[p / 2]];
        const SelfSlot<T>& s = C.self[p];
        T ta, na, tb, nb; jdot<T, S, ba>(K, s.px, s.pz, qacc, ta, na); jdot<T, S, bb>(K, s.px, s.pz, qacc, tb, nb);
        sfr[p] = s.nx * (tb - ta) + s.nz * (nb - na) - s.aref;
        jdot<T, S, ba>(K, s.px, s.pz, sr, ta, na); jdot<T, S, bb>(K, s.px, s.pz, sr, tb, nb);
        sfv[p] = s.nx * (tb - ta) + s.nz * (nb - na);
      });
    }
    auto deriv = [&](T a, T& d1, T& d2) {
      d1 = q1 + a * q2; d2 = q2;
      static_for<1, S::NB>([&](auto JJ) { constexpr int j = JJ;
        if constexpr (S::limited[j]) { T x = lr[j] + a * lv[j]; bool on = C.lim[j].active && x < T(0);
          T dd = on ? C.lim[j].D : T(0); d1 += dd * x * lv[j]; d2 += dd * lv[j] * lv[j]; } });
      static_for<0, 2 * S::NG>([&](auto KK) { constexpr int k = KK; const ContactSlot<T>& c = C.con[k];
        static_for<0, 3>([&](auto RR) { constexpr int r = RR;
          T x = cr[k][r] + a * cv[k][r]; bool on = c.active && x < T(0);
          T dd = on ? (r == 2 ? T(2) * c.D : c.D) : T(0); d1 += dd * x * cv[k][r]; d2 += dd * cv[k][r] * cv[k][r]; }); });
      if constexpr (S::NSELF > 0) static_for<0, 2 * S::NSELF>([&](auto PP) { constexpr int p = PP;
        T x = sfr[p] + a * sfv[p]; bool on = C.self[p].active && x < T(0);
        T dd = on ? C.self[p].D : T(0); d1 += dd * x * sfv[p]; d2 += dd * sfv[p] * sfv[p]; });
    };
    // phi' is piecewise linear and increasing: safeguarded Newton from alpha = 0
    T a = T(0), lo = T(0), hi = T(-1), d1, d2;
    deriv(a, d1, d2);
    const T d1ref = abs_t(d1) * T(sizeof(T) == 4 ? 1e-5 : 1e-13) + T(1e-30);
    bool ls_done = lane_done;
    for (int ls = 0; ls < 12; ++ls) {
      if (!REX_WAVE_ANY(!ls_done)) break;
      if (d1 < T(0)) lo = a; else hi = a;
      T an = a - d1 / d2;
      if (hi >= T(0) && (an <= lo || an >= hi)) an = T(0.5) * (lo + hi);
      an = max_t(an, lo);
      T prev = a;
      a = ls_done ? a : an;
      deriv(a, d1, d2);
      ls_done = ls_done || abs_t(d1) <= d1ref || a == prev;
    }
    a = lane_done ? T(0) : a;
    static_for<0, S::NV>([&](auto II) { qacc[II] += a * sr[II]; });
    st.iters = it + 1;
    if (it == MAXIT - 1) st.capped = REX_WAVE_ANY(!lane_done);
  }
  return st;
}

// one forward-dynamics evaluation: qacc(q, v, ctrl)  ([3P] mj_forward)
template <class T, class S>
REX_HD SolveStats forward(const T (&q)[S::NV], const T (&v)[S::NV], const T (&ctrl)[S::NU], const PlanarGeom<T, S>& G,
                          const LaneParams<T, S>& P, const SolParams<T>& sp, T (&qacc)[S::NV], T (&M)[S::NV][S::NV]) {
  Kin<T, S> K;
  kinematics<T, S>(q, G, K);
  T bias[S::NV], f[S::NV], a0[S::NV];
  mass_and_bias<T, S>(v, G, P, K, M, bias);
  f[0] = -bias[0]; f[1] = -bias[1]; f[2] = -bias[2];
  static_for<1, S::NB>([&](auto JJ) {
    constexpr int j = JJ;
    T c = min_t(max_t(ctrl[j - 1], T(-1)), T(1));   // ctrlrange -1..1 on every motor of the three XMLs
    f[j + 2] = -G.damping[j] * v[j + 2] - G.stiffness[j] * q[j + 2] - bias[j + 2] + T(S::gear[j - 1]) * c;
  });
  T L[S::NV][S::NV];
  static_for<0, S::NV>([&](auto II) { constexpr int i = II; a0[i] = f[i];
    static_for<0, i + 1>([&](auto JJ) { constexpr int j = JJ; if constexpr (dof_coupled<S>(i, j)) L[i][j] = M[i][j]; }); });
  ldl_factor<T, S>(L);
  ldl_solve<T, S>(L, a0);
  Constraints<T, S> C;
  make_constraints<T, S>(q, v, G, P, sp, K, C);
  SolveStats st{0, false};
  if (REX_WAVE_ANY(C.any)) st = solve_newton<T, S>(M, f, a0, K, C, sp, qacc);
  else static_for<0, S::NV>([&](auto II) { qacc[II] = a0[II]; });
  return st;
}

// one mj_step: RK4 ([3P] mj_RungeKutta, N=4) or semi-implicit Euler with implicit joint damping
// ([3P] mj_Euler).  Returns the OR of "solver hit its cap".
template <class T, class S>
REX_HD bool substep(T (&q)[S::NV], T (&v)[S::NV], const T (&ctrl)[S::NU], const PlanarGeom<T, S>& G,
                    const LaneParams<T, S>& P, const SolParams<T>& sp) {
  const T h = T(S::TIMESTEP);
  T M[S::NV][S::NV];
  bool capped = false;
  if constexpr (S::RK4) {
    // stage loop kept rolled: one instance of forward() in the kernel, 4x smaller code and far
    // lower register pressure than four inlined copies
    T q0[S::NV], v0[S::NV], dq[S::NV], dv[S::NV], acc[S::NV];
    static_for<0, S::NV>([&](auto II) { q0[II] = q[II]; v0[II] = v[II]; dq[II] = T(0); dv[II] = T(0); });
#pragma unroll 1
    for (int stage = 0; stage < 4; ++stage) {
      capped |= forward<T, S>(q, v, ctrl, G, P, sp, acc, M).capped;
      const T w = (stage == 0 || stage == 3) ? T(1.0 / 6) : T(1.0 / 3);   // B = [1/6 1/3 1/3 1/6]
      const T c = stage == 2 ? h : T(0.5) * h;                             // A = [.5; 0 .5; 0 0 1]
      static_for<0, S::NV>([&](auto II) { constexpr int i = II;
        dq[i] += w * v[i]; dv[i] += w * acc[i];
        T qn = q0[i] + c * v[i], vn = v0[i] + c * acc[i];
        q[i] = stage == 3 ? q0[i] + h * dq[i] : qn;
        v[i] = stage == 3 ? v0[i] + h * dv[i] : vn; });
    }
  } else {
    T acc[S::NV], rhs[S::NV];
    capped |= forward<T, S>(q, v, ctrl, G, P, sp, acc, M).capped;
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
