// Timing harness for the few-channel transposed convolution (deconv_rows_kernel in conv.hip) outside
// Python: the production entry point on the two layer shapes that take it.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off deconv_rows.hip -o deconv_rows
#include "../../disentangled-vae_amd/csrc/conv.hip"
#include <stdarg.h>
#include <stdio.h>
#include <vector>

void dvae_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
int dvae_mfma_precision() { return DVAE_PRECISION_F32; }

static void run(int64_t N, int64_t H, int64_t Cin, int64_t Cout, bool xf) {
  float *x, *wp, *y, *stats, *sc, *sh;
  const int64_t nx = N * H * H * Cin, ny = N * 4 * H * H * Cout;
  hipMalloc(&x, nx * 4);
  hipMalloc(&y, ny * 4);
  hipMalloc(&wp, 16 * Cin * Cout * 4);
  hipMalloc(&stats, 4 * 1024 * 2 * Cout * 4);
  hipMalloc(&sc, Cin * 4);
  hipMalloc(&sh, Cin * 4);
  std::vector<float> h(nx);
  for (int64_t i = 0; i < nx; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice);
  hipMemcpy(wp, h.data(), 16 * Cin * Cout * 4, hipMemcpyHostToDevice);
  hipMemcpy(sc, h.data(), Cin * 4, hipMemcpyHostToDevice);
  hipMemcpy(sh, h.data() + 100, Cin * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto call = [&]() {
    return dvae_deconv4s2_fwd_f32(x, xf ? sc : nullptr, xf ? sh : nullptr, xf ? DVAE_ACT_LRELU : DVAE_ACT_NONE, wp, y,
                                  stats, N, H, H, Cin, Cout, nullptr);
  };
  if (call() != 0) return;
  hipDeviceSynchronize();
  const int reps = 10;
  hipEventRecord(e0, nullptr);
  for (int i = 0; i < reps; ++i) call();
  hipEventRecord(e1, nullptr);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, fl = 2.0 * N * H * H * 16 * Cin * Cout;
  printf("deconv %ld->%ld H%ld N%ld %s: %8.1f us  %6.1f TFLOP/s (%.3f of peak)\n", (long)Cin, (long)Cout, (long)H, (long)N,
         xf ? "bn+lrelu" : "plain   ", us, fl / us / 1e6, fl / us / 1e6 / 157.3);
  hipFree(x); hipFree(y); hipFree(wp); hipFree(stats); hipFree(sc); hipFree(sh);
}

int main() {
  run(5120, 16, 32, 16, false);
  run(5120, 16, 32, 16, true);
  run(5120, 16, 64, 32, false);
  run(5120, 16, 64, 32, true);
  return 0;
}
