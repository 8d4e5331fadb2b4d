// humanoid_pair_host.cpp -- TEST HARNESS ONLY (never loaded by the product): the two-lanes-per-env humanoid engine
// (random-envs_amd/csrc/humanoid_pair.hpp) on the host.  The two lanes of a pair run as two threads in lock step: an exchange
// is a rendezvous (write own slot, barrier, read the partner's), the env's LDS column is an array both threads see.
#include <atomic>
#include <cstring>
#include <thread>
#include "../../random-envs_amd/csrc/humanoid_model.hpp"
#include "../../random-envs_amd/csrc/humanoid_pair.hpp"

using namespace rex::hum;

struct Barrier2 {   // sense-reversing spin barrier for two threads
  std::atomic<int> count{0}; std::atomic<int> sense{0};
  void wait(int& local) {
    local ^= 1;
    if (count.fetch_add(1, std::memory_order_acq_rel) == 1) { count.store(0, std::memory_order_relaxed); sense.store(local, std::memory_order_release); }
    else while (sense.load(std::memory_order_acquire) != local) { }
  }
};
template <class T> struct Shared { Barrier2 bar; alignas(16) T col[pr::PAIR_WORDS]; unsigned char slot[2][16]; };
template <class T> struct HostPair {
  Shared<T>* sh; int s; mutable int sense = 0;
  int side() const { return s; }
  template <class U> U xchg(U x) const { memcpy(sh->slot[s], &x, sizeof(U)); sh->bar.wait(sense); U r; memcpy(&r, sh->slot[s ^ 1], sizeof(U)); sh->bar.wait(sense); return r; }
  bool any(bool b) const { const unsigned o = xchg<unsigned>(b ? 1u : 0u); return b || o != 0u; }   // (the device asks the whole wave: any superset of the pair is correct)
  T* col() const { return sh->col; }
  void sync() const { sh->bar.wait(sense); }
};

template <class T> struct Ctx { Model<T> m; };
template <class T> static Ctx<T>* ctx() {
  static Ctx<T>* c = nullptr;
  if (!c) { c = new Ctx<T>(); Model<double> md; build_model(md); convert_model(md, c->m); }
  return c;
}

template <class T>
static void lane_main(int s, Shared<T>* sh, int n, const double* qpos, const double* qvel, const double* act, const double* xi, const double* xprev,
                      double* qpos_out, double* qvel_out, double* obs, double* reward, unsigned char* done, double* xout, int* overflow, int* nrows) {
  const Model<T>& m = ctx<T>()->m;
  HostPair<T> p{sh, s};
  const bool left = s != 0;
  pr::PKin<T>* K = new pr::PKin<T>(); pr::PScratch<T>* sc = new pr::PScratch<T>();
  for (int i = 0; i < n; i++) {
    auto gd = [&](int ld) { return left ? pr::gdL(ld) : pr::gdR(ld); };
    auto gb = [&](int lb) { return left ? pr::gbL(lb) : pr::gbR(lb); };
    pr::PLane<T> L;
    for (int lb = 0; lb < pr::LB; lb++) L.mass[lb] = T(xi[(size_t)(gb(lb) - 1) * n + i]);
    for (int ld = 0; ld < pr::LD; ld++) L.damping[ld] = ld < 6 ? T(0) : T(xi[(size_t)(13 + gd(ld) - 6) * n + i]);
    T ql[pr::LQ], vl[pr::LD], cl[pr::LU], xp[pr::LB];
    for (int k = 0; k < 7; k++) ql[k] = T(qpos[(size_t)k * n + i]);
    for (int ld = 6; ld < pr::LD; ld++) ql[ld + 1] = T(qpos[(size_t)(gd(ld) + 1) * n + i]);
    for (int ld = 0; ld < pr::LD; ld++) vl[ld] = T(qvel[(size_t)gd(ld) * n + i]);
    T asq = 0;
    for (int u = 0; u < NU; u++) { const T a = T(act[(size_t)u * n + i]); asq += a * a;
      for (int ld = 6; ld < pr::LD; ld++) if (kActDof[u] == gd(ld)) cl[ld - 6] = a; }
    if (xprev && xprev[i] == xprev[i]) for (int lb = 0; lb < pr::LB; lb++) xp[lb] = T(xprev[(size_t)gb(lb) * n + i]);
    else {   // set_state's sim.forward(): the one-lane engine (both threads compute the same)
      Lane<T> L1; L1.mass[0] = 0; for (int b = 1; b < NBODY; b++) L1.mass[b] = T(xi[(size_t)(b - 1) * n + i]);
      for (int d = 0; d < NV; d++) L1.damping[d] = d < 6 ? T(0) : T(xi[(size_t)(13 + d - 6) * n + i]);
      T q[NQ], v[NV], x14[NBODY]; Kin<T>* k1 = new Kin<T>(); Scratch<T>* s1 = new Scratch<T>();
      for (int k = 0; k < NQ; k++) q[k] = T(qpos[(size_t)k * n + i]);
      for (int k = 0; k < NV; k++) v[k] = T(qvel[(size_t)k * n + i]);
      env_reset_obs(m, L1, q, v, x14, *k1, *s1, [](int, T) {});
      for (int lb = 0; lb < pr::LB; lb++) xp[lb] = x14[gb(lb)];
      delete k1; delete s1;
    }
    pr::PObs<T> park; T r; bool d;
    pr::env_step(p, m, L, ql, vl, cl, asq, xp, *K, *sc, park, r, d);
    pr::emit_obs(p, ql, vl, park, [&](auto RR, auto RL, T val) { constexpr int rr = RR, rl = RL; obs[(size_t)(left ? rl : rr) * n + i] = double(val); });
    if (!left) { for (int k = 0; k < 10; k++) qpos_out[(size_t)k * n + i] = double(ql[k]); for (int k = 0; k < 9; k++) qvel_out[(size_t)k * n + i] = double(vl[k]);
                 reward[i] = double(r); done[i] = d; if (overflow) overflow[i] = K->overflow; if (nrows) nrows[i] = K->nefc; if (xout) xout[i] = 0; }
    for (int ld = 9; ld < pr::LD; ld++) { qpos_out[(size_t)(gd(ld) + 1) * n + i] = double(ql[ld + 1]); qvel_out[(size_t)gd(ld) * n + i] = double(vl[ld]); }
    if (xout) for (int lb = left ? 3 : 0; lb < pr::LB; lb++) xout[(size_t)gb(lb) * n + i] = double(xp[lb]);
    p.sync();
  }
  delete K; delete sc;
}

template <class T>
static void run(int n, const double* qpos, const double* qvel, const double* act, const double* xi, const double* xprev,
                double* qpos_out, double* qvel_out, double* obs, double* reward, unsigned char* done, double* xout, int* overflow, int* nrows) {
  (void)ctx<T>();   // build the model before the lanes start
  Shared<T>* sh = new Shared<T>();
  std::thread t1(lane_main<T>, 1, sh, n, qpos, qvel, act, xi, xprev, qpos_out, qvel_out, obs, reward, done, xout, overflow, nrows);
  lane_main<T>(0, sh, n, qpos, qvel, act, xi, xprev, qpos_out, qvel_out, obs, reward, done, xout, overflow, nrows);
  t1.join();
  delete sh;
}

extern "C" {
int hp_step(int f32, int n, const double* qpos, const double* qvel, const double* act, const double* xi, const double* xprev,
            double* qpos_out, double* qvel_out, double* obs, double* reward, unsigned char* done, double* xout, int* overflow, int* nrows) {
  if (f32) run<float>(n, qpos, qvel, act, xi, xprev, qpos_out, qvel_out, obs, reward, done, xout, overflow, nrows);
  else run<double>(n, qpos, qvel, act, xi, xprev, qpos_out, qvel_out, obs, reward, done, xout, overflow, nrows);
  return 0;
}
int hp_check_model() { Model<double> md; build_model(md); return pr::check_pair_model(md) ? 1 : 0; }
}
