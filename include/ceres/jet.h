// ceres/jet.h — forward-mode dual numbers for AutoDiffCostFunction (role of ceres::Jet<T,N>).
// Host-side only: the facade uses Jets to *recognise* SE(3) between factors and to evaluate user
// functors on the CPU when asked to (CostFunction::Evaluate); the solve itself runs analytic
// Jacobians on the GPU.  Written from the definition of dual numbers, not from Ceres' jet.h.
#ifndef PGO_CERES_JET_H_
#define PGO_CERES_JET_H_

#include <cmath>
#include <limits>
#include <ostream>

namespace ceres {

template <typename T, int N>
struct Jet {
  enum { DIMENSION = N };
  typedef T Scalar;
  T a;
  T v[N];

  Jet() : a() { for (int i = 0; i < N; ++i) v[i] = T(); }
  Jet(const T& value) : a(value) { for (int i = 0; i < N; ++i) v[i] = T(); }  // NOLINT(runtime/explicit)
  Jet(const T& value, int k) : a(value) { for (int i = 0; i < N; ++i) v[i] = T(); v[k] = T(1.0); }

  Jet& operator+=(const Jet& y) { a += y.a; for (int i = 0; i < N; ++i) v[i] += y.v[i]; return *this; }
  Jet& operator-=(const Jet& y) { a -= y.a; for (int i = 0; i < N; ++i) v[i] -= y.v[i]; return *this; }
  Jet& operator*=(const Jet& y) { *this = *this * y; return *this; }
  Jet& operator/=(const Jet& y) { *this = *this / y; return *this; }
  Jet& operator+=(const T& s) { a += s; return *this; }
  Jet& operator-=(const T& s) { a -= s; return *this; }
  Jet& operator*=(const T& s) { a *= s; for (int i = 0; i < N; ++i) v[i] *= s; return *this; }
  Jet& operator/=(const T& s) { a /= s; for (int i = 0; i < N; ++i) v[i] /= s; return *this; }
};

#define PGO_JET template <typename T, int N> inline
PGO_JET Jet<T, N> operator+(const Jet<T, N>& f) { return f; }
PGO_JET Jet<T, N> operator-(const Jet<T, N>& f) { Jet<T, N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
PGO_JET Jet<T, N> operator+(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
PGO_JET Jet<T, N> operator+(const Jet<T, N>& f, T s) { Jet<T, N> h = f; h.a += s; return h; }
PGO_JET Jet<T, N> operator+(T s, const Jet<T, N>& f) { Jet<T, N> h = f; h.a += s; return h; }
PGO_JET Jet<T, N> operator-(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
PGO_JET Jet<T, N> operator-(const Jet<T, N>& f, T s) { Jet<T, N> h = f; h.a -= s; return h; }
PGO_JET Jet<T, N> operator-(T s, const Jet<T, N>& f) { Jet<T, N> h = -f; h.a += s; return h; }
PGO_JET Jet<T, N> operator*(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
PGO_JET Jet<T, N> operator*(const Jet<T, N>& f, T s) { Jet<T, N> h; h.a = f.a * s; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h; }
PGO_JET Jet<T, N> operator*(T s, const Jet<T, N>& f) { return f * s; }
PGO_JET Jet<T, N> operator/(const Jet<T, N>& f, const Jet<T, N>& g) {
  Jet<T, N> h; const T inv = T(1.0) / g.a; h.a = f.a * inv;
  for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - h.a * g.v[i]) * inv;
  return h;
}
PGO_JET Jet<T, N> operator/(const Jet<T, N>& f, T s) { return f * (T(1.0) / s); }
PGO_JET Jet<T, N> operator/(T s, const Jet<T, N>& g) { Jet<T, N> h; const T inv = T(1.0) / g.a; h.a = s * inv; for (int i = 0; i < N; ++i) h.v[i] = -h.a * g.v[i] * inv; return h; }

#define PGO_JET_CMP(op) \
  PGO_JET bool operator op(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a op g.a; } \
  PGO_JET bool operator op(const Jet<T, N>& f, const T& s) { return f.a op s; }            \
  PGO_JET bool operator op(const T& s, const Jet<T, N>& f) { return s op f.a; }
PGO_JET_CMP(<) PGO_JET_CMP(<=) PGO_JET_CMP(>) PGO_JET_CMP(>=) PGO_JET_CMP(==) PGO_JET_CMP(!=)
#undef PGO_JET_CMP

// elementary functions: h = f(g.a), h' = f'(g.a) g'
#define PGO_JET_FN(name, value, deriv)                                                       \
  PGO_JET Jet<T, N> name(const Jet<T, N>& f) {                                               \
    Jet<T, N> h; h.a = (value); const T d = (deriv); for (int i = 0; i < N; ++i) h.v[i] = d * f.v[i]; return h; }
PGO_JET_FN(sqrt, std::sqrt(f.a), T(0.5) / std::sqrt(f.a))
PGO_JET_FN(exp, std::exp(f.a), std::exp(f.a))
PGO_JET_FN(log, std::log(f.a), T(1.0) / f.a)
PGO_JET_FN(sin, std::sin(f.a), std::cos(f.a))
PGO_JET_FN(cos, std::cos(f.a), -std::sin(f.a))
PGO_JET_FN(tan, std::tan(f.a), T(1.0) + std::tan(f.a) * std::tan(f.a))
PGO_JET_FN(asin, std::asin(f.a), T(1.0) / std::sqrt(T(1.0) - f.a * f.a))
PGO_JET_FN(acos, std::acos(f.a), -T(1.0) / std::sqrt(T(1.0) - f.a * f.a))
PGO_JET_FN(atan, std::atan(f.a), T(1.0) / (T(1.0) + f.a * f.a))
PGO_JET_FN(abs, std::fabs(f.a), (f.a < T(0.0) ? T(-1.0) : T(1.0)))
#undef PGO_JET_FN
PGO_JET Jet<T, N> atan2(const Jet<T, N>& g, const Jet<T, N>& f) {
  Jet<T, N> h; h.a = std::atan2(g.a, f.a); const T d = T(1.0) / (f.a * f.a + g.a * g.a);
  for (int i = 0; i < N; ++i) h.v[i] = d * (f.a * g.v[i] - g.a * f.v[i]);
  return h;
}
PGO_JET Jet<T, N> pow(const Jet<T, N>& f, double p) {
  Jet<T, N> h; h.a = std::pow(f.a, p); const T d = p * std::pow(f.a, p - 1.0);
  for (int i = 0; i < N; ++i) h.v[i] = d * f.v[i];
  return h;
}
PGO_JET bool isfinite(const Jet<T, N>& f) { if (!std::isfinite(f.a)) return false; for (int i = 0; i < N; ++i) if (!std::isfinite(f.v[i])) return false; return true; }
PGO_JET bool IsFinite(const Jet<T, N>& f) { return isfinite(f); }
PGO_JET std::ostream& operator<<(std::ostream& s, const Jet<T, N>& z) { return s << "[" << z.a << " ; ...]"; }
#undef PGO_JET

}  // namespace ceres

// Eigen interoperability (the reference functor instantiates Eigen::Quaternion<Jet<double,14>>,
// PLUS/include/PoseGraph3dError.h:24-51).  Compiled only where Eigen is installed.
#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
namespace Eigen {
template <typename T, int N>
struct NumTraits<ceres::Jet<T, N> > {
  typedef ceres::Jet<T, N> Real;
  typedef ceres::Jet<T, N> NonInteger;
  typedef ceres::Jet<T, N> Nested;
  typedef ceres::Jet<T, N> Literal;
  static typename ceres::Jet<T, N> dummy_precision() { return ceres::Jet<T, N>(1e-12); }
  static inline Real epsilon() { return Real(std::numeric_limits<T>::epsilon()); }
  static inline int digits10() { return NumTraits<T>::digits10(); }
  enum { IsComplex = 0, IsInteger = 0, IsSigned, ReadCost = 1, AddCost = 1, MulCost = 3, HasFloatingPoint = 1, RequireInitialization = 1 };
  static inline Real highest() { return Real(std::numeric_limits<T>::max()); }
  static inline Real lowest() { return Real(-std::numeric_limits<T>::max()); }
};
}  // namespace Eigen
#endif
#endif

#endif  // PGO_CERES_JET_H_
