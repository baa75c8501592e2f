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
class SoftLOneLoss : public LossFunction {
 public:
  explicit SoftLOneLoss(double a) : a_(a), b_(a * a), c_(1.0 / (a * a)) {}
  virtual void Evaluate(double s, double rho[3]) const {
    const double sum = 1.0 + s * c_, tmp = std::sqrt(sum);
    rho[0] = 2.0 * b_ * (tmp - 1.0);
    rho[1] = std::max(std::numeric_limits<double>::min(), 1.0 / tmp);
    rho[2] = -(c_ * rho[1]) / (2.0 * sum);
  }
  double a() const { return a_; }
 private:
  const double a_, b_, c_;
};
class CauchyLoss : public LossFunction {
 public:
  explicit CauchyLoss(double a) : a_(a), b_(a * a), c_(1.0 / (a * a)) {}
  virtual void Evaluate(double s, double rho[3]) const {
    const double sum = 1.0 + s * c_, inv = 1.0 / sum;
    rho[0] = b_ * std::log(sum);
    rho[1] = std::max(std::numeric_limits<double>::min(), inv);
    rho[2] = -c_ * (inv * inv);
  }
  double a() const { return a_; }
 private:
  const double a_, b_, c_;
};
class ArctanLoss : public LossFunction {
 public:
  explicit ArctanLoss(double a) : a_(a), b_(1.0 / (a * a)) {}
  virtual void Evaluate(double s, double rho[3]) const {
    const double sum = 1.0 + s * s * b_, inv = 1.0 / sum;
    rho[0] = a_ * std::atan2(s, a_);
    rho[1] = std::max(std::numeric_limits<double>::min(), inv);
    rho[2] = -2.0 * s * b_ * (inv * inv);
  }
  double a() const { return a_; }
 private:
  const double a_, b_;
};
// Not a Ceres class: the switchable-constraint edge of Suenderhauf & Protzel with its switch variable eliminated in closed
// form, rho(s) = Phi s / (Phi + s) (PGO_LOSS_SWITCHABLE, include/pgo.h).  Offered through the same LossFunction interface.
class SwitchableConstraintLoss : public LossFunction {
 public:
  explicit SwitchableConstraintLoss(double phi) : a_(phi) {}
  virtual void Evaluate(double s, double rho[3]) const {
    const double q = a_ / (a_ + s);
    rho[0] = s * q;
    rho[1] = std::max(std::numeric_limits<double>::min(), q * q);
    rho[2] = -2.0 * q * q / (a_ + s);
  }
  double a() const { return a_; }
 private:
  const double a_;
};
}  // namespace ceres
#endif
