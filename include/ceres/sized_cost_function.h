// ceres/sized_cost_function.h
#ifndef PGO_CERES_SIZED_COST_FUNCTION_H_
#define PGO_CERES_SIZED_COST_FUNCTION_H_
#include "ceres/cost_function.h"
namespace ceres {
template <int kNumResiduals, int N0, int N1 = 0, int N2 = 0, int N3 = 0, int N4 = 0, int N5 = 0, int N6 = 0, int N7 = 0,
          int N8 = 0, int N9 = 0>
class SizedCostFunction : public CostFunction {
 public:
  SizedCostFunction() {
    set_num_residuals(kNumResiduals);
    const int sizes[10] = {N0, N1, N2, N3, N4, N5, N6, N7, N8, N9};
    for (int i = 0; i < 10 && sizes[i] > 0; ++i) mutable_parameter_block_sizes()->push_back(sizes[i]);
  }
  virtual ~SizedCostFunction() {}
};
}  // namespace ceres
#endif
