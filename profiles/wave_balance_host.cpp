// Diagnostic (host, fp32 build of the shipped planar engine with -DREX_STATS hooks): Newton iterations per lane, per wave and per LAUNCH in steady state,
// and what dealing the envs to the waves by a predictor (or by an oracle) would do to the slowest wave.
//   g++ -O2 -std=c++17 -o /tmp/wave_balance profiles/wave_balance_host.cpp && /tmp/wave_balance <kind 1 hopper / 2 half-cheetah / 3 walker2d> <envs> <measured steps>
#define REX_STATS 1
#ifndef GENX
#define GENX 2
#endif
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <algorithm>
#include <array>
#include "../random-envs_amd/csrc/planar_model.hpp"
using namespace rex;
template <class S> void run(int N, int settle, int meas) {
  using T = float;
  PlanarGeom<T, S> G; SolParams<T> sp; T nominal[S::NB]; T sz[8];
  for (int k = 0; k < S::NSIZE; k++) sz[k] = T(S::default_size[k]);
  derive_model<T, S>(sz, G, nominal, sp); if (getenv("CORR")) sp.corr = atoi(getenv("CORR")); if (getenv("WARM")) sp.warm = atoi(getenv("WARM")); if (getenv("FAST")) sp.fast = atoi(getenv("FAST")); if (getenv("LSFREE")) sp.ls_free = atoi(getenv("LSFREE"));
  std::mt19937 rng(1); std::uniform_real_distribution<float> U(-1.f, 1.f);
  struct Env { T q[S::NV], v[S::NV], acc[S::NV], xi[S::NXI]; int t; };
  std::vector<Env> E(N);
  auto reset = [&](Env& e) { for (int k = 0; k < S::NV; k++) { e.q[k] = 0.005f * U(rng); e.v[k] = 0.005f * U(rng); e.acc[k] = 0; } if (S::KIND != 2) e.q[1] += 1.25f; e.t = 0;
    for (int k = 0; k < S::NXI; k++) e.xi[k] = 0; for (int b = 0; b < S::NB; b++) e.xi[b] = nominal[b] * (1.f + 0.1f * U(rng));
    if (S::KIND == 2) e.xi[7] = 0.4f; if (S::KIND == 3) { for (int k = 0; k < 4; k++) e.xi[7 + k] = sz[k]; e.xi[11] = 0.9f; e.xi[12] = 1.9f; } };
  for (auto& e : E) reset(e);
  long hist[4][10] = {}; long heavy[4][4] = {}; long nheavy = 0; const long HEAVY = S::FRAME_SKIP * (S::RK4 ? 4 : 1) + 4; int nprint = 0; std::vector<long> perlane; std::vector<std::array<int,20>> tr;
  for (int step = 0; step < settle + meas; step++) {
    if (step == settle) gstats() = GlobalStats{};
    for (auto& e : E) {
      T c[S::NU]; for (int k = 0; k < S::NU; k++) c[k] = U(rng);
      LaneParams<T, S> P; lane_params(S{}, e.xi, P);
      long p2before = gstats().pass2; gstats().ntrace = 0; long tg0[4][4]; for (int a_ = 0; a_ < 4; a_++) for (int b_ = 0; b_ < 4; b_++) tg0[a_][b_] = gstats().toggles[a_][b_];
      T x0 = e.q[0];
      for (int s = 0; s < S::FRAME_SKIP; s++) substep<T, S, false, GENX>(e.q, e.v, c, G, P, sp, e.acc, s > 0);
      if (step >= settle) { perlane.push_back(gstats().pass2 - p2before); std::array<int,20> a{}; for (int k = 0; k < 20 && k < gstats().ntrace; k++) { a[k] = gstats().trace[k] % 100; int md = gstats().trace[k] / 100; hist[md][std::min(a[k], 9)]++; } tr.push_back(a); if (gstats().pass2 - p2before >= HEAVY) { for (int a_ = 0; a_ < 4; a_++) for (int b_ = 0; b_ < 4; b_++) heavy[a_][b_] += gstats().toggles[a_][b_] - tg0[a_][b_]; nheavy++; }
        if (gstats().pass2 - p2before >= HEAVY && nprint < 8) { nprint++; printf("lane-step with %ld iterations: per solve", gstats().pass2 - p2before); for (int k = 0; k < gstats().ntrace; k++) printf(" %d", gstats().trace[k]); printf("\n"); } }
      e.t++;
      bool dn = false;
      if (S::KIND == 1) dn = !(e.q[1] > 0.7f && fabsf(e.q[2]) < 0.2f);
      if (S::KIND == 3) dn = !(e.q[1] > 0.8f && e.q[1] < 2.0f && e.q[2] > -1.0f && e.q[2] < 1.0f);
      if (dn || e.t >= 500) reset(e);
    }
  }
  auto& g = gstats();
  printf("solves %ld iters %ld pass1 %ld pass2 %ld ls_evals %ld  per solve: pass1 %.3f pass2 %.3f ls %.3f\n", g.solves, g.iters, g.pass1, g.pass2, g.ls_evals,
         double(g.pass1) / g.solves, double(g.pass2) / g.solves, double(g.ls_evals) / g.solves);
  for (int md = 0; md < 4; md++) { printf("mode %d iterations histogram:", md); for (int k = 0; k < 10; k++) printf(" %ld", hist[md][k]); printf("\n"); }
  printf("toggles [nl][ns] after a full non-exact step (3 = 3+):\n");
  for (int a = 0; a < 4; a++) { for (int b = 0; b < 4; b++) printf(" %8ld", g.toggles[a][b]); printf("\n"); }
  printf("toggles of the %ld lane-steps with >= %ld iterations:\n", nheavy, HEAVY);
  for (int a = 0; a < 4; a++) { for (int b = 0; b < 4; b++) printf(" %8ld", heavy[a][b]); printf("\n"); }
  std::sort(perlane.begin(), perlane.end());
  size_t n = perlane.size();
  printf("pass2 per lane-step: mean %.2f p50 %ld p90 %ld p99 %ld p99.9 %ld max %ld\n", double(g.pass2) / n, perlane[n / 2], perlane[n * 9 / 10], perlane[n * 99 / 100], perlane[n * 999 / 1000], perlane[n - 1]);
  // wave view: 32 consecutive envs of the same step = one wave; its cost = sum over solves of the max iterations of its lanes
  { int ns = S::FRAME_SKIP * (S::RK4 ? 4 : 1); std::vector<long> wv; double lanesum = 0;
    for (size_t w = 0; w + 32 <= tr.size(); w += 32) { long c = 0; for (int k = 0; k < ns; k++) { int m = 0; for (int l = 0; l < 32; l++) { m = std::max(m, tr[w + l][k]); lanesum += tr[w + l][k]; } c += m; } wv.push_back(c); }
    std::sort(wv.begin(), wv.end()); size_t m = wv.size(); double mean = 0; for (long c : wv) mean += c; mean /= m;
    printf("iterations per WAVE-step (32 envs): mean %.2f p50 %ld p90 %ld p99 %ld max %ld   (lane mean %.2f); max over 1024 waves ~ p99.9\n", mean, wv[m / 2], wv[m * 9 / 10], wv[m * 99 / 100], wv[m - 1], lanesum / (m * 32));
    // kernel view: a launch waits for its slowest wave.  Natural grouping (env i in wave i / 32) against envs dealt to the waves by a predictor
    // (the env's extra iterations in the PREVIOUS step, sorted descending, round-robin over the waves) and by the oracle (this step's own count).
    { const size_t NW = N / 32; const int steps = (int)(tr.size() / N);
      auto cost = [&](const std::vector<int>& order, int st) { long mx = 0; double mean = 0;
        for (size_t w = 0; w < NW; w++) { long c = 0; for (int k = 0; k < ns; k++) { int m = 0; for (int l = 0; l < 32; l++) m = std::max(m, tr[(size_t)st * N + order[w * 32 + l]][k]); c += m; } mx = std::max(mx, c); mean += c; }
        return std::make_pair(mx, mean / NW); };
      auto extra = [&](int st, int e) { int x = 0; for (int k = 0; k < ns; k++) x += std::max(0, tr[(size_t)st * N + e][k] - 1); return x; };
      auto deal = [&](std::vector<int> idx) { std::vector<int> order(N); for (size_t j = 0; j < (size_t)N; j++) order[(j % NW) * 32 + j / NW] = idx[j]; return order; };
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0, m0 = 0, m1 = 0, m2 = 0; int cnt = 0; std::vector<int> nat(N); for (int i = 0; i < N; i++) nat[i] = i;
      std::vector<int> acc(N, 0);
      for (int st = 1; st < steps; st++) {
        std::vector<int> idx(N); for (int i = 0; i < N; i++) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return extra(st - 1, a) > extra(st - 1, b); });
        auto c0 = cost(nat, st), c1 = cost(deal(idx), st);
        std::vector<int> id2(N); for (int i = 0; i < N; i++) id2[i] = i;
        std::stable_sort(id2.begin(), id2.end(), [&](int a, int b) { return extra(st, a) > extra(st, b); });
        auto c2 = cost(deal(id2), st);
        for (int i = 0; i < N; i++) acc[i] = (acc[i] * 3) / 4 + 4 * extra(st - 1, i);    // decaying sum of the earlier steps
        std::vector<int> id3(N); for (int i = 0; i < N; i++) id3[i] = i;
        std::stable_sort(id3.begin(), id3.end(), [&](int a, int b) { return acc[a] > acc[b]; });
        auto c3 = cost(deal(id3), st);
        a0 += c0.first; a1 += c1.first; a2 += c2.first; a3 += c3.first; m0 += c0.second; m1 += c1.second; m2 += c2.second; cnt++; }
      printf("slowest of %zu waves per step, iterations: natural %.1f (mean wave %.2f) | dealt by the previous step's extra iterations %.1f (mean %.2f) | by a decaying sum %.1f | oracle %.1f (mean %.2f)\n",
             NW, a0 / cnt, m0 / cnt, a1 / cnt, m1 / cnt, a3 / cnt, a2 / cnt, m2 / cnt);
      // persistence: P(extra > 0 at t | extra > 0 at t - 1) against the base rate
      long both = 0, prev = 0, cur = 0, tot = 0; for (int st = 1; st < steps; st++) for (int e = 0; e < N; e++) { bool p = extra(st - 1, e) > 0, c = extra(st, e) > 0; both += p && c; prev += p; cur += c; tot++; }
      printf("env-steps with extra iterations: %.1f %%; after a step with extra iterations: %.1f %%\n", 100.0 * cur / tot, 100.0 * both / std::max(prev, 1L)); }
    for (int W : {16, 8}) { std::vector<long> w2; for (size_t w = 0; w + W <= tr.size(); w += W) { long c = 0; for (int k = 0; k < ns; k++) { int mm = 0; for (int l = 0; l < W; l++) mm = std::max(mm, tr[w + l][k]); c += mm; } w2.push_back(c); }
      std::sort(w2.begin(), w2.end()); double me = 0; for (long c : w2) me += c; me /= w2.size(); printf("  %d envs per wave: mean %.2f p99 %ld p99.9 %ld max %ld\n", W, me, w2[w2.size() * 99 / 100], w2[w2.size() * 999 / 1000], w2.back()); } }
}
int main(int argc, char** argv) {
  int kind = argc > 1 ? atoi(argv[1]) : 1, N = argc > 2 ? atoi(argv[2]) : 2048;
  int meas = argc > 3 ? atoi(argv[3]) : 20;
  if (kind == 1) run<HopperSpec>(N, 300, meas); else if (kind == 3) run<Walker2dSpec>(N, 300, meas); else run<HalfCheetahSpec>(N, 300, meas);
}
