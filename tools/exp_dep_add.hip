// Latency of a dependent chain of v_add_f64 (what bounds the strict Java-order sums, fa_ordered_chain): one wave, N dependent additions,
// wall clock per addition; with 1, 7 and 64 active lanes, and two interleaved chains per lane for the issue rate.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/exp_dep_add tools/exp_dep_add.hip && /tmp/exp_dep_add
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int CHAINS> __global__ void __launch_bounds__(64) dep_add(const double* in, double* out, int n, int lanes) {
  if ((int)threadIdx.x >= lanes) return;
  double v[8];
  for (int i = 0; i < 8; i++) v[i] = in[threadIdx.x * 8 + i];
  double s0 = 0.0, s1 = 0.0;
  for (int i = 0; i < n; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      s0 += v[u];
      if (CHAINS == 2) s1 += v[7 - u];
      asm volatile("" : "+v"(s0), "+v"(s1));
    }
  }
  out[threadIdx.x] = s0 + s1;
}

int main() {
  double *in, *out;
  CHECK(hipMalloc(&in, 64 * 8 * 8));
  CHECK(hipMalloc(&out, 64 * 8));
  std::vector<double> h(64 * 8, 1e-3);
  CHECK(hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  const int n = 1 << 24;
  for (int chains = 1; chains <= 2; chains++)
    for (int lanes : {1, 7, 64}) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(a));
        if (chains == 1) dep_add<1><<<1, 64>>>(in, out, n, lanes); else dep_add<2><<<1, 64>>>(in, out, n, lanes);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
      }
      printf("chains per lane %d, active lanes %2d: %.3f ns per step (%d steps)\n", chains, lanes, best * 1e6 / n, n);
    }
  return 0;
}
