// Diagnostic build (MI355X_MICROARCH.md "DVFS give-back" item 6): which clock does the chip hold inside the two
// MFMA-dense kernels of the path, and how many shader cycles does their main loop take?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -DMMEGO_STAMP scripts/clock_probe.hip -o /tmp/clock_probe
// The kernels are the product sources compiled with stamps enabled; the product library never contains a stamp.
// Stamp values go only to mmego_stamp_buf.  Do not quote this build's run time, only clocks and cycle shares.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../mmego_amd/csrc/gemm_tile.hip"
#include "../mmego_amd/csrc/lstm_step.hip"

static float* dev_random(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  srand(seed);
  for (size_t i = 0; i < n; ++i) h[i] = scale * (2.0f * rand() / (float)RAND_MAX - 1.0f);
  float* d;
  hipMalloc(&d, n * sizeof(float));
  hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return d;
}

static void report(const char* name, int nwg, double ideal_loop_cycles) {
  std::vector<unsigned long long> s((size_t)nwg * MMEGO_STAMP_SLOTS * 2);
  hipMemcpyFromSymbol(s.data(), HIP_SYMBOL(mmego_stamp_buf), s.size() * 8);
  std::vector<double> clk, pro, loop, epi;
  for (int b = 0; b < nwg; ++b) {
    const unsigned long long* w = &s[(size_t)b * MMEGO_STAMP_SLOTS * 2];
    double dt = (double)(w[6] - w[0]), dr = (double)(w[7] - w[1]);
    if (dr <= 0) continue;
    clk.push_back(dt / dr * 0.1);  // GHz (s_memrealtime ticks at 100 MHz)
    pro.push_back((double)(w[2] - w[0]));
    loop.push_back((double)(w[4] - w[2]));
    epi.push_back((double)(w[6] - w[4]));
  }
  if (getenv("PROBE_TIMELINE")) {   // per-workgroup placement and start/end times (us since the first start)
    std::vector<unsigned int> hw((size_t)nwg * 2);
    hipMemcpyFromSymbol(hw.data(), HIP_SYMBOL(mmego_stamp_hw), hw.size() * 4);
    unsigned long long r0 = ~0ull;
    for (int b = 0; b < nwg; ++b) r0 = std::min(r0, s[(size_t)b * MMEGO_STAMP_SLOTS * 2 + 1]);
    char fn[256];
    snprintf(fn, sizeof fn, "%s/timeline_%s.csv", getenv("PROBE_TIMELINE"), name);
    for (char* q = fn + strlen(getenv("PROBE_TIMELINE")); *q; ++q) if (*q == ' ' || *q == '<' || *q == '>' || *q == ',' || *q == '=') *q = '_';
    FILE* f = fopen(fn, "w");
    if (f) {
      fprintf(f, "block,xcc,se,cu,start_us,end_us\n");
      for (int b = 0; b < nwg; ++b) {
        const unsigned long long* w = &s[(size_t)b * MMEGO_STAMP_SLOTS * 2];
        unsigned h = hw[2 * b], x = hw[2 * b + 1];
        fprintf(f, "%d,%u,%u,%u,%.2f,%.2f\n", b, x & 15, (h >> 13) & 7, (h >> 8) & 15, (w[1] - r0) * 0.01, (w[7] - r0) * 0.01);
      }
      fclose(f);
    }
  }
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  double c = med(clk), l = med(loop);
  printf("%-34s clock %.3f GHz | cycles: prologue %.0f  loop %.0f (ideal MFMA %.0f, %.2f)  epilogue %.0f | loop %.2f us\n", name, c,
         med(pro), l, ideal_loop_cycles, ideal_loop_cycles / l, med(epi), l / c / 1e3);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 2.5;
  hipStream_t st = 0;
  if (!getenv("PROBE_SKIP_GEMM")) {  // LSTM input projection shape of IMU_Net: 10240 x 2048 x 1024
    const int M = 10240, N = 2048, K = getenv("PROBE_K") ? atoi(getenv("PROBE_K")) : 1024;
    float *A = dev_random((size_t)M * K, 1.0f, 1), *W = dev_random((size_t)N * K, 0.05f, 2), *C, *bias = dev_random(N, 0.1f, 3);
    hipMalloc(&C, (size_t)M * N * 4);
    TileP tp = {A, W, C, bias, M, N, K, K, K, N, 0, 0, 1, K, nullptr, 1, 0, 0, 0};
    auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
      for (int i = 0; i < 50; ++i) mmego_detail::gemm_tile_launch(st, tp, true, true);
      hipStreamSynchronize(st);
      n += 50;
    }
    double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("gemm_tile 10240x2048x%d: %ld launches, %.1f us each (stamped build)\n", K, n, el / n * 1e6);
    // 2 workgroups share a CU: a wave owns its SIMD's matrix pipe half of the time at best -> ideal = 2 x MFMA cycles
    report("gemm_tile", 1536, 2.0 * (K / 64) * 8 * 16 * 64);
    hipFree(A); hipFree(W); hipFree(C); hipFree(bias);
  }
  {  // recurrent step of rnn_fast: Bn=512, H=512, both directions; lstm_step_dma_kernel<32> (both directions in one launch)
    const int Bn = 512, H = 512, T = 20;
    mmego_step_dbg = getenv("PROBE_STEP_DBG") ? atoi(getenv("PROBE_STEP_DBG")) : 0;
    float* out = dev_random((size_t)Bn * T * 2 * H, 0.5f, 4);
    float* xp = dev_random((size_t)Bn * T * 8 * H, 0.5f, 5);
    float *w0 = dev_random((size_t)4 * H * H, 0.04f, 6), *w1 = dev_random((size_t)4 * H * H, 0.04f, 7);
    float *b0 = dev_random(4 * H, 0.04f, 8), *b1 = dev_random(4 * H, 0.04f, 9);
    float* c = dev_random((size_t)2 * Bn * H, 0.5f, 10);
    const long xs = (long)T * 8 * H, os = getenv("PROBE_DENSE_H") ? (long)2 * H : (long)T * 2 * H;   // dense: h rows 4 KB apart
    const int ndir = getenv("PROBE_NDIR") ? atoi(getenv("PROBE_NDIR")) : 2;      // 1: single-direction launches (the two-chain form's)
    auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
      for (int i = 0; i < 200; ++i) {
        int s = 1 + (i % (T - 2)), t1 = T - 1 - s;
        mmego_lstm_step(st, ndir, Bn, H, 0, out + (long)(s - 1) * 2 * H, out + (long)(t1 + 1) * 2 * H + H, os, w0, w1, b0, b1,
                        xp + (long)s * 8 * H, xp + (long)t1 * 8 * H + 4 * H, xs, out + (long)s * 2 * H, out + (long)t1 * 2 * H + H, os,
                        c, c + (long)Bn * H, nullptr, nullptr, nullptr, nullptr);
      }
      hipStreamSynchronize(st);
      n += 200;
    }
    double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("lstm_step Bn=512: %ld launches, %.1f us each (stamped build)\n", n, el / n * 1e6);
    {
      const int ht = 32;
      // ideal = MFMA issue cycles of the work that shares one SIMD's matrix pipe (two workgroups per CU at HT = 16)
      report(ht == 16 ? "lstm_step_dma_kernel<16>" : "lstm_step_dma_kernel<32>", ndir * (H / ht) * (Bn / 64),
             ndir == 2 ? 8.0 * 4 * 32 * 32 : 8.0 * 4 * 32 * 16);
    }
    hipFree(out); hipFree(xp); hipFree(w0); hipFree(w1); hipFree(b0); hipFree(b1); hipFree(c);
  }
  return 0;
}
