// ceres/local_parameterization.h — LocalParameterization interface + EigenQuaternionParameterization
// (finial.cpp:496-497).  Behaviour per SURVEY.md §8a row a7; the in-tree statement the reference ships
// is src/other_projects/bundle_adjustment/ceres_extensions.h:25-50.
#ifndef PGO_CERES_LOCAL_PARAMETERIZATION_H_
#define PGO_CERES_LOCAL_PARAMETERIZATION_H_
#include <cmath>
namespace ceres {
class LocalParameterization {
 public:
  virtual ~LocalParameterization() {}
  virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const = 0;
  virtual bool ComputeJacobian(const double* x, double* jacobian) const = 0;  // GlobalSize x LocalSize, row-major
  virtual int GlobalSize() const = 0;
  virtual int LocalSize() const = 0;
};

// Quaternion stored x,y,z,w (Eigen coeffs).  Plus(q, d) = [sin|d| d/|d| ; cos|d|] (x) q.
class EigenQuaternionParameterization : public LocalParameterization {
 public:
  virtual ~EigenQuaternionParameterization() {}
  virtual bool Plus(const double* x, const double* d, double* out) const {
    const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (n > 0.0) {
      const double s = std::sin(n) / n, tx = s * d[0], ty = s * d[1], tz = s * d[2], tw = std::cos(n);
      const double qx = x[0], qy = x[1], qz = x[2], qw = x[3];
      out[0] = tw * qx + tx * qw + ty * qz - tz * qy;
      out[1] = tw * qy + ty * qw + tz * qx - tx * qz;
      out[2] = tw * qz + tz * qw + tx * qy - ty * qx;
      out[3] = tw * qw - tx * qx - ty * qy - tz * qz;
    } else {
      for (int i = 0; i < 4; ++i) out[i] = x[i];
    }
    return true;
  }
  virtual bool ComputeJacobian(const double* x, double* J) const {
    J[0] = x[3];  J[1] = x[2];   J[2] = -x[1];
    J[3] = -x[2]; J[4] = x[3];   J[5] = x[0];
    J[6] = x[1];  J[7] = -x[0];  J[8] = x[3];
    J[9] = -x[0]; J[10] = -x[1]; J[11] = -x[2];
    return true;
  }
  virtual int GlobalSize() const { return 4; }
  virtual int LocalSize() const { return 3; }
};
}  // namespace ceres
#endif
