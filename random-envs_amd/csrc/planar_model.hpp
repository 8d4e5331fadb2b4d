// planar_model.hpp -- the product's own "model compiler": turns the reference's Jinja `size`
// list into the constants the kernels need (PlanarGeom, SolParams).  It replaces
// build_model() = Jinja render + mujoco_py.load_model_from_xml + MjSim
// (random_envs/jinja/jinja_mujoco_env.py:92-97, template_renderer.py:16-19): no XML text, no
// model object -- geometry formulas of the templates evaluated directly.  Host + device code:
// hopper / half-cheetah constants are derived once on the host (geometry never changes there),
// walker2d constants are derived per environment on the device at reset time because
// RandomWalker2dEnv.set_task rebuilds the model from the xi lengths (random_walker2d.py:106-113).
//
// MuJoCo-2.1.0 compile rules restated here (all [3P], see oracle/mjo_core.c for provenance):
//   capsule mass = density * pi r^2 (L + r)  (the 2.1.0 volume)
//   inertiafromgeom: body COM / inertia from its geoms; settotalmass rescales mass and inertia
//   dof_invweight0 = diag(M^-1) at qpos0; body_invweight0[.,0] = mean diag of Jcom M^-1 Jcom^T
#pragma once
#include "planar_engine.hpp"

namespace rex {

constexpr double kPi = 3.14159265358979323846;

// geometry of one tree at qpos0 in world (x,z) coordinates
template <class T, class S>
struct RawGeom {
  T anchor[S::NB][2];            // joint anchor of each body (root: body origin)
  T g1[S::NG][2], g2[S::NG][2];  // capsule end points
  T radius[S::NG];
  T armature[S::NB], damping[S::NB], stiffness[S::NB];
  T density, settotalmass;       // settotalmass <= 0: off
};

// hopper.xml:27-41 (coordinate="global")
template <class T>
REX_HD void raw_geometry(const HopperSpec&, const T* s, RawGeom<T, HopperSpec>& r) {
  T zt = s[0] / 2 + s[1] + s[2] + T(0.1);
  T z1 = s[1] + s[2] + T(0.1), z2 = s[2] + T(0.1), z3 = T(0.1);
  r.anchor[0][0] = 0; r.anchor[0][1] = zt;     // :27,31
  r.anchor[1][0] = 0; r.anchor[1][1] = z1;     // :34
  r.anchor[2][0] = 0; r.anchor[2][1] = z2;     // :37
  r.anchor[3][0] = 0; r.anchor[3][1] = z3;     // :40
  r.g1[0][0] = 0; r.g1[0][1] = s[0] + z1; r.g2[0][0] = 0; r.g2[0][1] = z1; r.radius[0] = T(0.05);   // :32
  r.g1[1][0] = 0; r.g1[1][1] = z1; r.g2[1][0] = 0; r.g2[1][1] = z2; r.radius[1] = T(0.05);          // :35
  r.g1[2][0] = 0; r.g1[2][1] = z2; r.g2[2][0] = 0; r.g2[2][1] = z3; r.radius[2] = T(0.04);          // :38
  r.g1[3][0] = -s[3] / 3; r.g1[3][1] = z3; r.g2[3][0] = s[3] * 2 / 3; r.g2[3][1] = z3; r.radius[3] = T(0.06); // :41
  r.armature[0] = 0; r.damping[0] = 0; r.stiffness[0] = 0;                                           // :29-31
  for (int i = 1; i < 4; i++) { r.armature[i] = 1; r.damping[i] = 1; r.stiffness[i] = 0; }           // :4
  r.density = 1000; r.settotalmass = 0;
}

// walker2d.xml:25-56
template <class T>
REX_HD void raw_geometry(const Walker2dSpec&, const T* s, RawGeom<T, Walker2dSpec>& r) {
  T z1 = s[1] + s[2], z2 = s[2], z3 = T(0.1);
  r.anchor[0][0] = 0; r.anchor[0][1] = T(1.25);                                                     // :25,29
  r.g1[0][0] = 0; r.g1[0][1] = s[0] + z1; r.g2[0][0] = 0; r.g2[0][1] = z1; r.radius[0] = T(0.05);   // :30
  for (int side = 0; side < 2; side++) {
    int b = 1 + 3 * side;
    r.anchor[b][0] = 0; r.anchor[b][1] = z1; r.anchor[b + 1][0] = 0; r.anchor[b + 1][1] = z2;       // :32,35
    r.anchor[b + 2][0] = 0; r.anchor[b + 2][1] = z3;                                                // :38
    r.g1[b][0] = 0; r.g1[b][1] = z1; r.g2[b][0] = 0; r.g2[b][1] = z2; r.radius[b] = T(0.05);        // :33
    r.g1[b + 1][0] = 0; r.g1[b + 1][1] = z2; r.g2[b + 1][0] = 0; r.g2[b + 1][1] = z3; r.radius[b + 1] = T(0.04); // :36
    r.g1[b + 2][0] = 0; r.g1[b + 2][1] = z3; r.g2[b + 2][0] = s[3]; r.g2[b + 2][1] = z3; r.radius[b + 2] = T(0.06); // :39
  }
  r.armature[0] = 0; r.damping[0] = 0; r.stiffness[0] = 0;
  for (int i = 1; i < 7; i++) { r.armature[i] = T(0.01); r.damping[i] = T(0.1); r.stiffness[i] = 0; } // :4
  r.density = 1000; r.settotalmass = 0;                                                              // :5
}

// half_cheetah.xml:35-51,86-118 (coordinate="local": accumulate body offsets)
template <class T>
REX_HD void raw_geometry(const HalfCheetahSpec&, const T* s, RawGeom<T, HalfCheetahSpec>& r) {
  const T tl = s[0], head = s[1], ha = T(0.87);
  const T ang[6] = {T(-3.8), T(-2.03), T(-0.27), T(0.52), T(-0.6), T(-0.6)};
  const T len[6] = {s[2], s[3], s[4], s[5], s[6], s[7]};
  const T rad = T(0.046);
  auto cap = [&](int g, T bx, T bz, T px, T pz, T a, T half) {   // geom pos + axisangle about y
    T dx = half * sin(a), dz = half * cos(a);
    r.g1[g][0] = bx + px + dx; r.g1[g][1] = bz + pz + dz; r.g2[g][0] = bx + px - dx; r.g2[g][1] = bz + pz - dz; r.radius[g] = rad;
  };
  T bx = 0, bz = T(0.7);
  r.anchor[0][0] = bx; r.anchor[0][1] = bz;                                                         // :86-90
  r.g1[0][0] = bx - tl / 2; r.g1[0][1] = bz; r.g2[0][0] = bx + tl / 2; r.g2[0][1] = bz; r.radius[0] = rad; // :91
  cap(1, bx, bz, tl / 2 + head * cos(ha), head * cos(ha), ha, head);                                // :92
  // back leg :93-104
  T x = bx - tl / 2, z = bz;
  r.anchor[1][0] = x; r.anchor[1][1] = z;
  cap(2, x, z, len[0] * sin(ang[0]), len[0] * cos(ang[0]), ang[0], len[0]);
  x += 2 * len[0] * sin(ang[0]); z += 2 * len[0] * cos(ang[0]);
  r.anchor[2][0] = x; r.anchor[2][1] = z;
  cap(3, x, z, len[1] * sin(ang[1]), len[1] * cos(ang[1]), ang[1], len[1]);
  x += 2 * len[1] * sin(ang[1]); z += 2 * len[1] * cos(ang[1]);
  r.anchor[3][0] = x; r.anchor[3][1] = z;
  cap(4, x, z, sin(-ang[2]) * len[2], -len[2], ang[2], len[2]);
  // front leg :105-117
  x = bx + tl / 2; z = bz;
  r.anchor[4][0] = x; r.anchor[4][1] = z;
  cap(5, x, z, len[3] * sin(-ang[3]), -len[3] * cos(ang[3]), ang[3], len[3]);
  x += 2 * len[3] * sin(-ang[3]); z += -2 * len[3] * cos(ang[3]);
  r.anchor[5][0] = x; r.anchor[5][1] = z;
  cap(6, x, z, len[4] * sin(-ang[4]), -len[4] * cos(ang[4]), ang[4], len[4]);
  x += 2 * len[4] * sin(-ang[4]); z += -2 * len[4] * cos(ang[4]);
  r.anchor[6][0] = x; r.anchor[6][1] = z;
  cap(7, x, z, sin(-ang[5]) * len[5] * 9 / 8, -len[5], ang[5], len[5]);
  const T damp[7] = {0, 6, T(4.5), 3, T(4.5), 3, T(1.5)}, stiff[7] = {0, 240, 180, 120, 180, 120, 60}; // :95-113
  for (int i = 0; i < 7; i++) { r.armature[i] = i ? T(0.1) : T(0); r.damping[i] = damp[i]; r.stiffness[i] = stiff[i]; } // :56
  r.density = 1000; r.settotalmass = 14;                                                             // :54
}

// solref / solimp of the contact pairs and joint limits
template <class T> REX_HD void sol_params(const HopperSpec&, SolParams<T>& sp, T* dmins) {
  dmins[0] = T(0.8); dmins[1] = T(0.8); dmins[2] = T(0.01); dmins[3] = T(0.001);   // geom solimp, margin: hopper.xml:5
  dmins[4] = T(0.9); dmins[5] = T(0.95); dmins[6] = T(0.001);                      // joint solimplimit: MuJoCo default
  (void)sp;
}
template <class T> REX_HD void sol_params(const Walker2dSpec&, SolParams<T>& sp, T* dmins) {
  dmins[0] = T(0.9); dmins[1] = T(0.95); dmins[2] = T(0.001); dmins[3] = T(0);     // defaults (walker2d.xml:5 sets none)
  dmins[4] = T(0.9); dmins[5] = T(0.95); dmins[6] = T(0.001);
  (void)sp;
}
template <class T> REX_HD void sol_params(const HalfCheetahSpec&, SolParams<T>& sp, T* dmins) {
  dmins[0] = T(0.0); dmins[1] = T(0.8); dmins[2] = T(0.01); dmins[3] = T(0);       // half_cheetah.xml:57,64
  dmins[4] = T(0.0); dmins[5] = T(0.8); dmins[6] = T(0.03);                        // solimplimit :56
  (void)sp;
}

// nominal geom-floor friction of geom g before xi is applied
REX_HD float nominal_mu(const HopperSpec&, int g) { return g == 3 ? 2.0f : 1.0f; }      // max(geom, floor=1): hopper.xml:26,32-41
REX_HD float nominal_mu(const Walker2dSpec&, int g) { return g == 6 ? 1.9f : 0.9f; }    // walker2d.xml:30-52,70-71 (floor .7)
REX_HD float nominal_mu(const HalfCheetahSpec&, int) { return 0.4f; }                    // half_cheetah.xml:57,64

// Full derivation: RawGeom -> PlanarGeom (+ nominal masses, SolParams).
template <class T, class S>
REX_HD void derive_model(const T* size, PlanarGeom<T, S>& G, T (&nominal_mass)[S::NB], SolParams<T>& sp) {
  RawGeom<T, S> r;
  raw_geometry(S{}, size, r);
  // inertiafromgeom
  T mass[S::NB], cx[S::NB], cz[S::NB], iyy[S::NB], total = 0;
  for (int b = 0; b < S::NB; b++) { mass[b] = 0; cx[b] = 0; cz[b] = 0; iyy[b] = 0; }
  T gm[S::NG], gi[S::NG], gx[S::NG], gz[S::NG];
  for (int g = 0; g < S::NG; g++) {
    int b = S::geom_body[g];
    T dx = r.g1[g][0] - r.g2[g][0], dz = r.g1[g][1] - r.g2[g][1];
    T L = sqrt_t(dx * dx + dz * dz), rad = r.radius[g];
    T m = r.density * T(kPi) * rad * rad * (L + rad);            // 2.1.0 capsule volume
    T ms = m * rad / (L + rad), mc = m - ms;                     // caps / cylinder split of that volume
    gi[g] = mc * (3 * rad * rad + L * L) / 12 + 2 * ms * rad * rad / 5 + ms * L * (3 * rad + 2 * L) / 8;
    gm[g] = m; gx[g] = T(0.5) * (r.g1[g][0] + r.g2[g][0]); gz[g] = T(0.5) * (r.g1[g][1] + r.g2[g][1]);
    mass[b] += m; cx[b] += m * gx[g]; cz[b] += m * gz[g];
  }
  for (int b = 0; b < S::NB; b++) { cx[b] /= mass[b]; cz[b] /= mass[b]; total += mass[b]; }
  for (int g = 0; g < S::NG; g++) {
    int b = S::geom_body[g];
    T dx = gx[g] - cx[b], dz = gz[g] - cz[b];
    iyy[b] += gi[g] + gm[g] * (dx * dx + dz * dz);
  }
  T scl = r.settotalmass > 0 ? r.settotalmass / total : T(1);
  for (int b = 0; b < S::NB; b++) {
    nominal_mass[b] = mass[b] * scl; G.iyy[b] = iyy[b] * scl;
    int p = S::parent[b];
    G.ja[b][0] = p < 0 ? T(0) : r.anchor[b][0] - r.anchor[p][0];
    G.ja[b][1] = p < 0 ? T(0) : r.anchor[b][1] - r.anchor[p][1];
    G.co[b][0] = cx[b] - r.anchor[b][0]; G.co[b][1] = cz[b] - r.anchor[b][1];
    G.armature[b] = r.armature[b]; G.damping[b] = r.damping[b]; G.stiffness[b] = r.stiffness[b];
  }
  for (int g = 0; g < S::NG; g++) {
    int b = S::geom_body[g];
    G.e1[g][0] = r.g1[g][0] - r.anchor[b][0]; G.e1[g][1] = r.g1[g][1] - r.anchor[b][1];
    G.e2[g][0] = r.g2[g][0] - r.anchor[b][0]; G.e2[g][1] = r.g2[g][1] - r.anchor[b][1];
    G.radius[g] = r.radius[g];
  }
  // mj_setConst at qpos0 (all joint angles 0 -> every body angle 0)
  T q0[S::NV], v0[S::NV];
  static_for<0, S::NV>([&](auto II) { q0[II] = 0; v0[II] = 0; });
  LaneParams<T, S> P;
  for (int b = 0; b < S::NB; b++) P.mass[b] = nominal_mass[b];
  Kin<T, S> K; kinematics<T, S>(q0, G, K);
  T M[S::NV][S::NV], bias[S::NV], F[S::NV][S::NV];
  mass_and_bias<T, S>(v0, G, P, K, M, bias);
  T tr = 0;
  static_for<0, S::NV>([&](auto II) { constexpr int i = II; tr += M[i][i];
    static_for<0, i + 1>([&](auto JJ) { constexpr int j = JJ; if constexpr (dof_coupled<S>(i, j)) F[i][j] = M[i][j]; }); });
  sp.meaninertia = tr / T(S::NV);
  ldl_factor<T, S>(F);
  static_for<1, S::NB>([&](auto JJ) {
    constexpr int j = JJ;
    T e[S::NV]; static_for<0, S::NV>([&](auto II) { e[II] = T(0); }); e[j + 2] = T(1);
    ldl_solve<T, S>(F, e); G.dof_invw[j] = e[j + 2];
  });
  G.dof_invw[0] = T(0);
  static_for<0, S::NB>([&](auto BB) {
    constexpr int b = BB;
    T px = K.A[b][0] + K.rc[b][0], pz = K.A[b][1] + K.rc[b][1];
    T rowx[S::NV], rowz[S::NV], sx[S::NV], sz[S::NV];
    static_for<0, S::NV>([&](auto II) { rowx[II] = T(0); rowz[II] = T(0); });
    jt_accum<T, S, b>(K, px, pz, T(1), T(0), rowx); jt_accum<T, S, b>(K, px, pz, T(0), T(1), rowz);
    static_for<0, S::NV>([&](auto II) { sx[II] = rowx[II]; sz[II] = rowz[II]; });
    ldl_solve<T, S>(F, sx); ldl_solve<T, S>(F, sz);
    T ax = 0, az = 0;
    static_for<0, S::NV>([&](auto II) { ax += rowx[II] * sx[II]; az += rowz[II] * sz[II]; });
    G.tran_invw[b] = (ax + az) / T(3);   // the y translation of a planar tree has zero Jacobian
  });
  // solref (.02, 1) on every contact and limit of the three XMLs; refsafe: timeconst >= 2*timestep
  T d[7]; sol_params(S{}, sp, d);
  auto clampimp = [](T x) { return x < T(0.0001) ? T(0.0001) : (x > T(0.9999) ? T(0.9999) : x); };
  T tc = max_t(T(0.02), T(2) * T(S::TIMESTEP));
  sp.con_dmin = clampimp(d[0]); sp.con_dmax = clampimp(d[1]); sp.con_width = d[2]; sp.con_margin = d[3];
  sp.lim_dmin = clampimp(d[4]); sp.lim_dmax = clampimp(d[5]); sp.lim_width = d[6];
  sp.con_K = T(1) / (sp.con_dmax * sp.con_dmax * tc * tc); sp.con_B = T(2) / (sp.con_dmax * tc);
  sp.lim_K = T(1) / (sp.lim_dmax * sp.lim_dmax * tc * tc); sp.lim_B = T(2) / (sp.lim_dmax * tc);
  // measured on MI355X at B = 32768 (kernel ms, ls_max/warm): hopper 16/0 .247, 3/1 .205; walker2d .476 -> .386;
  // half-cheetah (one evaluation per mj_step: the previous qacc is a poor guess) 16/0 .152, 3/0 .143, 3/1 .165
  // Re-measured once the kernel time was understood to be the SLOWEST wave's (kernel ms at ls_max 0/1/2/3): hopper
  // .138/.139/.147/.151, half-cheetah .117/.129/.134/.135, walker2d .307/.323/.345/.353 -- but without any line search (0) a few
  // walker2d waves hit the iteration cap, and 1 still did (4 waves in 1 500 steps); every env keeps at least one safeguarded step.
  // line search: the first 4 Newton iterations of a solve take the full step (no phi' evaluations beyond the one at alpha = 1
  // that detects an exact step); a solve still running after that gets the safeguarded exact search, up to 3 evaluations
  // per iteration (measured at B = 32768: hopper 0.0945 -> 0.0876 ms, walker2d 0.300 -> 0.265, half-cheetah 0.127 -> 0.118;
  // no solve hit the iteration cap, results unchanged to rounding: tests/test_planar_engine_host.py)
  sp.ls_max = 3; sp.ls_free = 4; sp.warm = S::RK4 ? 1 : 0; sp.fast = 1; sp.corr = 2;
}

// ---------------------------------------------------------------------------------------------------
// Walker2d per-env geometry, compact form.  Of the 105 floats of PlanarGeom only 25 distinct values depend
// on the xi lengths (both legs are built from the same formulas, x offsets are zero except the feet, radii /
// armature / damping / stiffness are constants): the per-env SoA block stores those 25, the step kernel expands
// them over a uniform nominal PlanarGeom.  kWalkerSlot[f] = compact slot of flat field f, or -1 (uniform).
// tests/test_planar_engine_host.py checks the table against derive_model() for random lengths.
// ---------------------------------------------------------------------------------------------------
constexpr int kWalkerCompact = 25;
struct WalkerMap {
  int slot[105];
  constexpr WalkerMap() : slot() {
    for (int f = 0; f < 105; f++) slot[f] = -1;
    constexpr int JA = 0, CO = 14, IYY = 28, E1 = 35, E2 = 49, TR = 70, DW = 77;   // field bases (floats)
    for (int side = 0; side < 2; side++) {
      const int b = 1 + 3 * side;
      slot[JA + 2 * b + 1] = 0; slot[JA + 2 * (b + 1) + 1] = 1; slot[JA + 2 * (b + 2) + 1] = 2;       // anchors (z)
      slot[CO + 2 * b + 1] = 4; slot[CO + 2 * (b + 1) + 1] = 5; slot[CO + 2 * (b + 2) + 0] = 6;       // COM offsets
      slot[IYY + b] = 8; slot[IYY + b + 1] = 9; slot[IYY + b + 2] = 10;
      slot[E2 + 2 * b + 1] = 13;                                                                      // thigh lower end z
      slot[E2 + 2 * (b + 1) + 1] = 14;                                                                // leg lower end z
      slot[E2 + 2 * (b + 2) + 0] = 15;                                                                // foot far end x
      slot[TR + b] = 17; slot[TR + b + 1] = 18; slot[TR + b + 2] = 19;
      slot[DW + b] = 20; slot[DW + b + 1] = 21; slot[DW + b + 2] = 22;
    }
    slot[CO + 1] = 3;            // torso COM z
    slot[IYY + 0] = 7;
    slot[E1 + 1] = 11; slot[E2 + 1] = 12;   // torso capsule ends z
    slot[TR + 0] = 16;
    // slots 23, 24 spare
  }
};
constexpr WalkerMap kWalkerMap{};

template <class T>
REX_HD void walker_compact_from_geom(const PlanarGeom<T, Walker2dSpec>& G, T* c) {
  const T* flat = reinterpret_cast<const T*>(&G);
  for (int k = 0; k < kWalkerCompact; k++) c[k] = T(0);
  for (int f = 0; f < 105; f++) if (kWalkerMap.slot[f] >= 0) c[kWalkerMap.slot[f]] = flat[f];
}
template <class T, class Load>
REX_HD void walker_expand(const PlanarGeom<T, Walker2dSpec>& uniform, Load&& load, PlanarGeom<T, Walker2dSpec>& G) {
  G = uniform;
  T c[kWalkerCompact];
  static_for<0, 23>([&](auto KK) { constexpr int k = KK; c[k] = load(k); });
  T* flat = reinterpret_cast<T*>(&G);
  static_for<0, 105>([&](auto FF) { constexpr int f = FF; if constexpr (kWalkerMap.slot[f] >= 0) flat[f] = c[kWalkerMap.slot[f]]; });
}

// xi -> per-lane dynamic parameters (get_task/set_task scatter maps)
template <class T>
REX_HD void lane_params(const HopperSpec&, const T* xi, LaneParams<T, HopperSpec>& P) {
  for (int i = 0; i < 4; i++) P.mass[i] = xi[i];                         // random_hopper.py:79-80
  for (int g = 0; g < 4; g++) P.mu[g] = T(nominal_mu(HopperSpec{}, g));
}
template <class T>
REX_HD void lane_params(const Walker2dSpec&, const T* xi, LaneParams<T, Walker2dSpec>& P) {
  for (int i = 0; i < 7; i++) P.mass[i] = xi[i];                         // random_walker2d.py:111
  for (int g = 0; g < 7; g++) P.mu[g] = T(0.9f);
  P.mu[3] = xi[11]; P.mu[6] = xi[12];                                    // random_walker2d.py:112-113
}
template <class T>
REX_HD void lane_params(const HalfCheetahSpec&, const T* xi, LaneParams<T, HalfCheetahSpec>& P) {
  for (int i = 0; i < 7; i++) P.mass[i] = xi[i];                         // random_half_cheetah.py:97
  for (int g = 0; g < 8; g++) P.mu[g] = T(0.4f);
  P.mu[4] = xi[7]; P.mu[7] = xi[7];                                      // random_half_cheetah.py:98 (bfoot, ffoot pairs)
}

}  // namespace rex
