// ceres/loss_function.h — LossFunction interface + HuberLoss (new HuberLoss(1.0), finial.cpp:495).
// rho[0] = rho(s), rho[1] = rho'(s), rho[2] = rho''(s) for s = |r|^2  [Ceres 1.13 semantics, SURVEY A.4].
#ifndef PGO_CERES_LOSS_FUNCTION_H_
#define PGO_CERES_LOSS_FUNCTION_H_
#include <algorithm>
#include <cmath>
#include <limits>
namespace ceres {
class LossFunction {
 public:
  virtual ~LossFunction() {}
  virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
class TrivialLoss : public LossFunction {
 public:
  virtual void Evaluate(double s, double rho[3]) const { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
};
class HuberLoss : public LossFunction {
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  virtual void Evaluate(double s, double rho[3]) const {
    if (s > b_) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a_ * r - b_;
      rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
  }
  double a() const { return a_; }
 private:
  const double a_, b_;
};
}  // namespace ceres
#endif
