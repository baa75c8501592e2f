// ceres/cost_function.h — ceres::CostFunction interface (vtable imported by the reference driver).
#ifndef PGO_CERES_COST_FUNCTION_H_
#define PGO_CERES_COST_FUNCTION_H_
#include <vector>
namespace ceres {
typedef int int32;
class CostFunction {
 public:
  CostFunction() : num_residuals_(0) {}
  virtual ~CostFunction() {}
  // parameters[i] has parameter_block_sizes()[i] doubles; residuals has num_residuals(); jacobians may be
  // NULL, and jacobians[i] (row-major num_residuals x block size) may be NULL.
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  const std::vector<int32>& parameter_block_sizes() const { return parameter_block_sizes_; }
  int num_residuals() const { return num_residuals_; }
 protected:
  std::vector<int32>* mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }
 private:
  CostFunction(const CostFunction&);
  void operator=(const CostFunction&);
  std::vector<int32> parameter_block_sizes_;
  int num_residuals_;
};
}  // namespace ceres
#endif
