// Diagnostic build of mlp_bwd_layer_kernel (mmego_amd/csrc/mlp_train.hip) with in-kernel stamps: shader cycles of its prologue (weights,
// BatchNorm states, gather of the 256 partial records), its tile loop (two rounds of 128 rows) and its epilogue, at 65 536 rows.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -DMMEGO_STAMP scripts/mlp_bwd_probe.hip -o scripts/exp/mlp_bwd_probe
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../mmego_amd/csrc/mlp_train.hip"

static float* dev_random(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  srand(seed);
  for (size_t i = 0; i < n; ++i) h[i] = scale * (2.0f * rand() / (float)RAND_MAX - 1.0f);
  float* d;
  hipMalloc(&d, n * sizeof(float));
  hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return d;
}

int main(int argc, char** argv) {
  const long rows = 65536;
  const int cfg[3][2] = {{64, 48}, {24, 16}, {8, 6}};       // (Cout, Cin)
  for (int ci = 0; ci < 3; ++ci) {
    const int Cout = cfg[ci][0], Cin = cfg[ci][1];
    float *dY = dev_random(rows * Cout, 1.f, 1), *Z = dev_random(rows * Cout, 1.f, 2), *X = dev_random(rows * Cin, 1.f, 3);
    float *W = dev_random(Cout * Cin, 0.2f, 4), *state = dev_random(4 * 64, 0.5f, 5), *in_state = dev_random(4 * 64, 0.5f, 6);
    float *dX, *dWp, *dg, *db;
    double *gp, *gprev;
    hipMalloc(&dX, rows * Cin * 4); hipMalloc(&dWp, 256 * 4096 * 4); hipMalloc(&dg, 256); hipMalloc(&db, 256);
    hipMalloc(&gp, 256 * 128 * 8); hipMemset(gp, 0, 256 * 128 * 8); hipMalloc(&gprev, 256 * 128 * 8);
    hipStream_t st = 0;
    auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 1.0) {
      for (int i = 0; i < 100; ++i) {
        int rc = mmego_mlp_bwd_layer(st, dY, Cout, Z, Cout, rows, Cout, state, gp, dg, db, X, Cin, Cin, in_state, W, dX, Cin, gprev, dWp);
        if (rc) { printf("rc %d\n", rc); return 1; }
      }
      hipStreamSynchronize(st);
      n += 100;
    }
    double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const int nwg = 256;
    std::vector<unsigned long long> sv((size_t)nwg * MMEGO_STAMP_SLOTS * 2);
    hipMemcpyFromSymbol(sv.data(), HIP_SYMBOL(mmego_stamp_buf), sv.size() * 8);
    std::vector<double> clk, pro, loop, epi;
    unsigned long long r0 = ~0ull, r1 = 0;
    for (int b = 0; b < nwg; ++b) {
      const unsigned long long* w = &sv[(size_t)b * MMEGO_STAMP_SLOTS * 2];
      double dt = (double)(w[6] - w[0]), dr = (double)(w[7] - w[1]);
      if (dr <= 0) continue;
      clk.push_back(dt / dr * 0.1); pro.push_back((double)(w[2] - w[0])); loop.push_back((double)(w[4] - w[2])); epi.push_back((double)(w[6] - w[4]));
      r0 = std::min(r0, w[1]); r1 = std::max(r1, w[7]);
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    double ck = med(clk);
    printf("Cout %2d Cin %2d: %.1f us per launch back to back | clock %.2f GHz | cycles prologue %.0f  tile loop %.0f  epilogue %.0f | us %.2f / %.2f / %.2f | first start -> last end %.2f us\n",
           Cout, Cin, el / n * 1e6, ck, med(pro), med(loop), med(epi), med(pro) / ck / 1e3, med(loop) / ck / 1e3, med(epi) / ck / 1e3, (r1 - r0) * 0.01);
  }
  return 0;
}
