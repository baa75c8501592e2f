// ceres/autodiff_cost_function.h — AutoDiffCostFunction<Functor, kNumResiduals, N0..N9>
// (used as <PoseGraph3dErrorTerm, 6, 3, 4, 3, 4> at PLUS/include/PoseGraph3dError.h:59).
// Evaluate() runs the functor on doubles (residuals only) or on Jets seeded over all parameter blocks.
#ifndef PGO_CERES_AUTODIFF_COST_FUNCTION_H_
#define PGO_CERES_AUTODIFF_COST_FUNCTION_H_
#include "ceres/jet.h"
#include "ceres/sized_cost_function.h"
#include "ceres/types.h"

namespace ceres {
namespace internal {
// calls functor(x0, ..., x_{n-1}, residuals) for n = 1..10 parameter blocks
template <int N> struct VariadicCall;
#define PGO_VC(n, ...) \
  template <> struct VariadicCall<n> { template <typename F, typename T> static bool Call(const F& f, T const* const* x, T* r) { return f(__VA_ARGS__, r); } };
PGO_VC(1, x[0]) PGO_VC(2, x[0], x[1]) PGO_VC(3, x[0], x[1], x[2]) PGO_VC(4, x[0], x[1], x[2], x[3])
PGO_VC(5, x[0], x[1], x[2], x[3], x[4]) PGO_VC(6, x[0], x[1], x[2], x[3], x[4], x[5])
PGO_VC(7, x[0], x[1], x[2], x[3], x[4], x[5], x[6]) PGO_VC(8, x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7])
PGO_VC(9, x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], x[8])
PGO_VC(10, x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], x[8], x[9])
#undef PGO_VC
}  // namespace internal

template <typename CostFunctor, int kNumResiduals, int N0, int N1 = 0, int N2 = 0, int N3 = 0, int N4 = 0, int N5 = 0,
          int N6 = 0, int N7 = 0, int N8 = 0, int N9 = 0>
class AutoDiffCostFunction : public SizedCostFunction<kNumResiduals, N0, N1, N2, N3, N4, N5, N6, N7, N8, N9> {
 public:
  enum { kNumBlocks = (N0 > 0) + (N1 > 0) + (N2 > 0) + (N3 > 0) + (N4 > 0) + (N5 > 0) + (N6 > 0) + (N7 > 0) + (N8 > 0) + (N9 > 0),
         kNumParameters = N0 + N1 + N2 + N3 + N4 + N5 + N6 + N7 + N8 + N9 };
  // Takes ownership of functor (Ceres default).
  explicit AutoDiffCostFunction(CostFunctor* functor) : functor_(functor), ownership_(TAKE_OWNERSHIP) {}
  AutoDiffCostFunction(CostFunctor* functor, Ownership ownership) : functor_(functor), ownership_(ownership) {}
  virtual ~AutoDiffCostFunction() { if (ownership_ == TAKE_OWNERSHIP) delete functor_; }

  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
    if (!jacobians) return internal::VariadicCall<kNumBlocks>::Call(*functor_, parameters, residuals);
    typedef Jet<double, kNumParameters> JetT;
    const int sizes[10] = {N0, N1, N2, N3, N4, N5, N6, N7, N8, N9};
    JetT x[kNumParameters];
    JetT const* blocks[kNumBlocks];
    int off = 0;
    for (int b = 0; b < kNumBlocks; ++b) {
      blocks[b] = x + off;
      for (int i = 0; i < sizes[b]; ++i) x[off + i] = JetT(parameters[b][i], off + i);
      off += sizes[b];
    }
    JetT out[kNumResiduals];
    if (!internal::VariadicCall<kNumBlocks>::Call(*functor_, blocks, out)) return false;
    off = 0;
    for (int b = 0; b < kNumBlocks; ++b) {
      if (jacobians[b])
        for (int r = 0; r < kNumResiduals; ++r)
          for (int i = 0; i < sizes[b]; ++i) jacobians[b][r * sizes[b] + i] = out[r].v[off + i];
      off += sizes[b];
    }
    for (int r = 0; r < kNumResiduals; ++r) residuals[r] = out[r].a;
    return true;
  }

 private:
  CostFunctor* functor_;
  Ownership ownership_;
};
}  // namespace ceres
#endif
