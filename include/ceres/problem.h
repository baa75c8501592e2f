// ceres/problem.h — ceres::Problem facade (finial.cpp:58,491-528) over the C ABI of include/pgo.h.
//
// Semantics kept from Ceres 1.13: parameter identity is the pointer value; blocks are
// created implicitly by AddResidualBlock with the sizes the cost function declares; re-adding a block or
// re-setting the same parameterization is a no-op; Problem owns (and deletes exactly once) every cost
// function, loss function and local parameterization handed to it; the user keeps parameter memory;
// API misuse aborts (Ceres CHECK-fails), solve failures are reported through Solver::Summary.
#ifndef PGO_CERES_PROBLEM_H_
#define PGO_CERES_PROBLEM_H_

#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#include "ceres/cost_function.h"
#include "ceres/local_parameterization.h"
#include "ceres/loss_function.h"
#include "ceres/types.h"

namespace ceres {

namespace internal {
struct ResidualBlock {
  CostFunction* cost;
  LossFunction* loss;
  std::vector<double*> blocks;
};
inline void Fatal(const char* what) {
  std::fprintf(stderr, "ceres (pgo facade) check failed: %s\n", what);
  std::abort();
}
}  // namespace internal

typedef internal::ResidualBlock* ResidualBlockId;

class Problem {
 public:
  struct Options {
    Options()
        : cost_function_ownership(TAKE_OWNERSHIP), loss_function_ownership(TAKE_OWNERSHIP),
          local_parameterization_ownership(TAKE_OWNERSHIP), enable_fast_removal(false), disable_all_safety_checks(false) {}
    Ownership cost_function_ownership, loss_function_ownership, local_parameterization_ownership;
    bool enable_fast_removal, disable_all_safety_checks;
  };

  Problem() {}
  explicit Problem(const Options& options) : options_(options) {}
  ~Problem() {
    std::set<CostFunction*> costs;
    std::set<LossFunction*> losses;
    std::set<LocalParameterization*> lps;
    for (size_t i = 0; i < residual_blocks_.size(); ++i) {
      costs.insert(residual_blocks_[i]->cost);
      if (residual_blocks_[i]->loss) losses.insert(residual_blocks_[i]->loss);
      delete residual_blocks_[i];
    }
    for (std::map<double*, LocalParameterization*>::iterator it = parameterizations_.begin(); it != parameterizations_.end(); ++it)
      if (it->second) lps.insert(it->second);
    if (options_.cost_function_ownership == TAKE_OWNERSHIP)
      for (std::set<CostFunction*>::iterator it = costs.begin(); it != costs.end(); ++it) delete *it;
    if (options_.loss_function_ownership == TAKE_OWNERSHIP)
      for (std::set<LossFunction*>::iterator it = losses.begin(); it != losses.end(); ++it) delete *it;
    if (options_.local_parameterization_ownership == TAKE_OWNERSHIP)
      for (std::set<LocalParameterization*>::iterator it = lps.begin(); it != lps.end(); ++it) delete *it;
  }

  ResidualBlockId AddResidualBlock(CostFunction* cost_function, LossFunction* loss_function,
                                   const std::vector<double*>& parameter_blocks) {
    if (!cost_function) internal::Fatal("AddResidualBlock: cost_function is NULL");
    const std::vector<int32>& sizes = cost_function->parameter_block_sizes();
    if (sizes.size() != parameter_blocks.size()) internal::Fatal("AddResidualBlock: wrong number of parameter blocks");
    for (size_t i = 0; i < sizes.size(); ++i) AddParameterBlock(parameter_blocks[i], sizes[i]);
    internal::ResidualBlock* rb = new internal::ResidualBlock;
    rb->cost = cost_function;
    rb->loss = loss_function;
    rb->blocks = parameter_blocks;
    residual_blocks_.push_back(rb);
    return rb;
  }
  // the x0..x9 overloads of Ceres 1.13 (finial.cpp:513-517 uses the four-block one)
  template <typename... Ts>
  ResidualBlockId AddResidualBlock(CostFunction* cost_function, LossFunction* loss_function, double* x0, Ts*... xs) {
    std::vector<double*> blocks;
    blocks.push_back(x0);
    double* rest[] = {xs..., static_cast<double*>(0)};
    for (size_t i = 0; i < sizeof...(xs); ++i) blocks.push_back(rest[i]);
    return AddResidualBlock(cost_function, loss_function, blocks);
  }

  void AddParameterBlock(double* values, int size) {
    if (!values) internal::Fatal("AddParameterBlock: NULL parameter block");
    std::map<double*, int>::iterator it = block_sizes_.find(values);
    if (it == block_sizes_.end()) { block_sizes_[values] = size; block_order_.push_back(values); }
    else if (it->second != size) internal::Fatal("AddParameterBlock: block re-added with a different size");
  }
  void AddParameterBlock(double* values, int size, LocalParameterization* lp) {
    AddParameterBlock(values, size);
    if (lp) SetParameterization(values, lp);
  }

  void SetParameterization(double* values, LocalParameterization* lp) {
    if (!block_sizes_.count(values)) internal::Fatal("SetParameterization: unknown parameter block");
    std::map<double*, LocalParameterization*>::iterator it = parameterizations_.find(values);
    if (it != parameterizations_.end() && it->second && lp && it->second != lp)
      internal::Fatal("SetParameterization: block already has a different parameterization");
    if (lp && lp->GlobalSize() != block_sizes_[values]) internal::Fatal("SetParameterization: GlobalSize mismatch");
    parameterizations_[values] = lp;
  }
  const LocalParameterization* GetParameterization(double* values) const {
    std::map<double*, LocalParameterization*>::const_iterator it = parameterizations_.find(values);
    return it == parameterizations_.end() ? 0 : it->second;
  }

  void SetParameterBlockConstant(double* values) {
    if (!block_sizes_.count(values)) internal::Fatal("SetParameterBlockConstant: unknown parameter block");
    constant_.insert(values);
  }
  void SetParameterBlockVariable(double* values) {
    if (!block_sizes_.count(values)) internal::Fatal("SetParameterBlockVariable: unknown parameter block");
    constant_.erase(values);
  }
  bool IsParameterBlockConstant(double* values) const { return constant_.count(values) != 0; }

  int NumParameterBlocks() const { return (int)block_sizes_.size(); }
  int NumResidualBlocks() const { return (int)residual_blocks_.size(); }
  int NumParameters() const { int n = 0; for (std::map<double*, int>::const_iterator it = block_sizes_.begin(); it != block_sizes_.end(); ++it) n += it->second; return n; }
  int NumResiduals() const { int n = 0; for (size_t i = 0; i < residual_blocks_.size(); ++i) n += residual_blocks_[i]->cost->num_residuals(); return n; }
  int ParameterBlockSize(const double* values) const {
    std::map<double*, int>::const_iterator it = block_sizes_.find(const_cast<double*>(values));
    return it == block_sizes_.end() ? 0 : it->second;
  }
  bool HasParameterBlock(const double* values) const { return block_sizes_.count(const_cast<double*>(values)) != 0; }

  // facade internals (used by ceres::Solve)
  const std::vector<internal::ResidualBlock*>& residual_blocks() const { return residual_blocks_; }
  const std::set<double*>& constant_blocks() const { return constant_; }

 private:
  Problem(const Problem&);
  void operator=(const Problem&);
  Options options_;
  std::map<double*, int> block_sizes_;
  std::vector<double*> block_order_;
  std::map<double*, LocalParameterization*> parameterizations_;
  std::set<double*> constant_;
  std::vector<internal::ResidualBlock*> residual_blocks_;
};

}  // namespace ceres
#endif
