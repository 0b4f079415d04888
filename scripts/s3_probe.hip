// Diagnostic build of the split3 recurrent step (mmego_amd/csrc/split3.hip) with in-kernel stamps: which clock does the chip hold in
// its bf16 MFMA loop, and how many shader cycles do prologue / product loop / reduction + cell update take?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -DMMEGO_STAMP scripts/s3_probe.hip -o scripts/exp/s3_probe
// The product library never contains a stamp.  Do not quote this build's run time, only clocks and cycle shares.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../mmego_amd/csrc/split3.hip"

static void* dev_random_bits(size_t nbytes, unsigned seed, bool bf16_small) {
  std::vector<unsigned short> h(nbytes / 2);
  srand(seed);
  for (size_t i = 0; i < h.size(); ++i) {
    // bf16 with a random mantissa, exponents around 2^-3 .. 2^-12 (pieces of values ~0.1), random sign
    unsigned e = 127 - 3 - (rand() % (bf16_small ? 10 : 3));
    h[i] = (unsigned short)(((rand() & 1) << 15) | (e << 7) | (rand() & 127));
  }
  void* d;
  hipMalloc(&d, nbytes);
  hipMemcpy(d, h.data(), nbytes, hipMemcpyHostToDevice);
  return d;
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 2.0;
  const int Bn = 512, H = 512, T = 20, nrb = Bn / 32, S2 = 2 * H / 16;
  hipStream_t st = 0;
  unsigned short* O = (unsigned short*)dev_random_bits((size_t)T * Bn * 2 * H * 6, 1, true);
  unsigned short* w0 = (unsigned short*)dev_random_bits((size_t)4 * H * H * 6, 2, true);
  unsigned short* w1 = (unsigned short*)dev_random_bits((size_t)4 * H * H * 6, 3, true);
  float *xpf, *c;
  hipMalloc(&xpf, (size_t)T * Bn * 8 * H * 4);
  hipMemset(xpf, 0, (size_t)T * Bn * 8 * H * 4);
  hipMalloc(&c, (size_t)2 * Bn * H * 4);
  hipMemset(c, 0, (size_t)2 * Bn * H * 4);
  auto win = [&](int t, int d) { return O + ((size_t)(t * nrb) * S2 + d * (H / 16)) * 3 * 512; };
  auto t0 = std::chrono::steady_clock::now();
  long n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int i = 0; i < 200; ++i) {
      int s = 1 + (i % (T - 2)), t1 = T - 1 - s;
      int rc = mmego_split3_step(st, 2, Bn, H, 0, win(s - 1, 0), win(t1 + 1, 1), S2 * 3, w0, w1, xpf, (long)s * nrb, (long)t1 * nrb, nullptr, nullptr, 0,
                                 win(s, 0), win(t1, 1), S2 * 3, c, c + (size_t)Bn * H, 6, 0);
      if (rc) { printf("rc %d\n", rc); return 1; }
    }
    hipStreamSynchronize(st);
    n += 200;
  }
  double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("split3 step Bn=512 H=512 both directions: %ld launches, %.1f us each (stamped build, eager back to back)\n", n, el / n * 1e6);
  const int nwg = 2 * 16 * 8;
  std::vector<unsigned long long> sv((size_t)nwg * MMEGO_STAMP_SLOTS * 2);
  hipMemcpyFromSymbol(sv.data(), HIP_SYMBOL(mmego_stamp_buf), sv.size() * 8);
  std::vector<double> clk, pro, loop, epi, startspread;
  unsigned long long r0 = ~0ull, r1 = 0;
  for (int b = 0; b < nwg; ++b) {
    const unsigned long long* w = &sv[(size_t)b * MMEGO_STAMP_SLOTS * 2];
    double dt = (double)(w[6] - w[0]), dr = (double)(w[7] - w[1]);
    if (dr <= 0) continue;
    clk.push_back(dt / dr * 0.1);
    pro.push_back((double)(w[2] - w[0]));
    loop.push_back((double)(w[4] - w[2]));
    epi.push_back((double)(w[6] - w[4]));
    r0 = std::min(r0, w[1]);
    r1 = std::max(r1, w[7]);
  }
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  double ck = med(clk);
  printf("clock %.3f GHz | cycles: prologue %.0f  product loop %.0f (MFMA issue 12288: %.2f)  reduction + cell + stores %.0f | us: %.2f / %.2f / %.2f | first start -> last end %.2f us\n",
         ck, med(pro), med(loop), 12288.0 / med(loop), med(epi), med(pro) / ck / 1e3, med(loop) / ck / 1e3, med(epi) / ck / 1e3, (r1 - r0) * 0.01);
  return 0;
}
