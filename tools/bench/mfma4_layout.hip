// tools/bench/mfma4_layout.hip — determines the operand / result lane mapping of v_mfma_f64_4x4x4_4b_f64 and the cbsz/abid
// broadcast empirically with unit operands (development measurement; the mapping is used by pgo_front_kernels.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CBSZ, int ABID>
__global__ void k(double* out) {
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
      out[(size_t)(la * 64 + lb) * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
    }
}
static void report(const char* name, const std::vector<double>& h) {
  printf("== %s ==\n", name);
  // for each A lane: which (B lane -> D lanes)
  for (int la = 0; la < 64; la += 1) {
    int cnt = 0;
    for (int lb = 0; lb < 64; ++lb) for (int l = 0; l < 64; ++l) if (h[(size_t)(la * 64 + lb) * 64 + l] != 0.0) ++cnt;
    if (la < 20 || cnt == 0) {
      printf("A lane %2d: ", la);
      int shown = 0;
      for (int lb = 0; lb < 64 && shown < 8; ++lb) for (int l = 0; l < 64; ++l) if (h[(size_t)(la * 64 + lb) * 64 + l] != 0.0) { printf("(B%d->D%d) ", lb, l); ++shown; }
      printf(" total %d\n", cnt);
    }
  }
}
int main() {
  double* d; hipMalloc(&d, sizeof(double) * 64 * 64 * 64);
  std::vector<double> h(64 * 64 * 64);
  k<0, 0><<<1, 64>>>(d); hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost); report("plain", h);
  k<2, 1><<<1, 64>>>(d); hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost); report("cbsz 2 abid 1", h);
  return 0;
}
