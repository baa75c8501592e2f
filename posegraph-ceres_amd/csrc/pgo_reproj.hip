// pgo_reproj.hip — batched MotionEstimate solves on the GPU.
//
// Role of REF/src/MotionEstimate.cc:71-129 + REF/include/MotionEstimate.h:34-91: for a candidate frame pair, n matched
// 3-D points P_i (last frame) and pixels (u_i, v_i) (current frame), minimise  sum_i rho(|K (q * P_i + t) - uv_i|^2)
// over the translation t (the reference keeps the essential-matrix rotation q constant, :108) — or over (q, t) — with
// Ceres' Levenberg-Marquardt, HuberLoss(1.0), EigenQuaternionParameterization, max 1000 iterations, exact steps.
// The problems of different pairs are independent ("embarrassingly parallel across candidate pairs"): one WAVE per
// problem, the whole trust-region loop inside the kernel — lanes stride over the points (residual, analytic 2x6 local
// Jacobian, Huber weight), the 21 + 6 + 1 sums of the normal equations are folded on the DPP crossbar, the 6x6 damped
// system is solved redundantly by every lane.  No host round trip per iteration, no atomics, fixed summation order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/pgo.h"
#include "pgo_math.h"

int pgo_candidates_set_error(int code, const char* msg);   // pgo_problem.cpp (error channel of the library)

namespace {

using namespace pgo;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shifted_r(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_r(double v) {   // total in every lane (row_shr 1/2/4/8, row_bcast 15/31, lane 63)
  v += dpp_shifted_r<0x111, 0xf>(v);
  v += dpp_shifted_r<0x112, 0xf>(v);
  v += dpp_shifted_r<0x114, 0xf>(v);
  v += dpp_shifted_r<0x118, 0xf>(v);
  v += dpp_shifted_r<0x142, 0xa>(v);
  v += dpp_shifted_r<0x143, 0xc>(v);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

struct Intr { double fx, fy, cx, cy; };

// residual of one observation and its 2x6 local Jacobian [dtheta | dt]; q is NOT assumed unit: the rotation is Eigen's
// v + 2w(u x v) + 2u x (u x v) (MotionEstimate.h:44), differentiated exactly and chained with the Plus Jacobian.
__device__ __forceinline__ void reproj_point(const Intr& K, const double* obs, const Q4& q, const V3& t, const V3& P, bool want_jac,
                                             double* r, double* J) {
  const V3 u{q.x, q.y, q.z};
  V3 uv = cross(u, P);
  uv = V3{uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};              // 2 (u x P)
  const V3 c = cross(u, uv);                                    // 2 u x (u x P)
  const double x = P.x + q.w * uv.x + c.x + t.x, y = P.y + q.w * uv.y + c.y + t.y, z = P.z + q.w * uv.z + c.z + t.z;
  r[0] = (K.fx * x) / z + K.cx - obs[0];
  r[1] = (K.fy * y) / z + K.cy - obs[1];
  if (!want_jac) return;
  const double iz = 1.0 / z;
  const double a = K.fx * iz, b = -(K.fx * x) * iz * iz, cc = K.fy * iz, d = -(K.fy * y) * iz * iz;
  // D = d(rotated point)/d(qx,qy,qz,qw): 3x4
  const double ud = u.x * P.x + u.y * P.y + u.z * P.z;
  double D[12];
  const double uu[3] = {u.x, u.y, u.z}, vv[3] = {P.x, P.y, P.z};
  // -2w [v]x :  [v]x = [0 -vz vy; vz 0 -vx; -vy vx 0]
  const double vx[9] = {0.0, -P.z, P.y, P.z, 0.0, -P.x, -P.y, P.x, 0.0};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      D[4 * i + j] = -2.0 * q.w * vx[3 * i + j] + 2.0 * ((i == j ? ud : 0.0) + uu[i] * vv[j] - 2.0 * vv[i] * uu[j]);
  D[3] = uv.x; D[7] = uv.y; D[11] = uv.z;                       // d/dw = 2 (u x v)
  // Plus Jacobian (ceres_extensions.h:44-50), 4x3
  const double PJ[12] = {q.w, q.z, -q.y, -q.z, q.w, q.x, q.y, -q.x, q.w, -q.x, -q.y, -q.z};
  double Jt[9];   // d(rotated point)/d(dtheta), 3x3
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      Jt[3 * i + j] = D[4 * i] * PJ[j] + D[4 * i + 1] * PJ[3 + j] + D[4 * i + 2] * PJ[6 + j] + D[4 * i + 3] * PJ[9 + j];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    J[j] = a * Jt[j] + b * Jt[6 + j];
    J[6 + j] = cc * Jt[3 + j] + d * Jt[6 + j];
  }
  J[3] = a; J[4] = 0.0; J[5] = b;
  J[9] = 0.0; J[10] = cc; J[11] = d;
}

// Cholesky solve of the SPD 6x6 system A x = b (A row-major, destroyed); false on a non-positive pivot
__device__ __forceinline__ bool solve6(double* A, const double* b, double* x) {
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double dd = A[7 * j];
#pragma unroll
    for (int k = 0; k < j; ++k) dd -= A[6 * j + k] * A[6 * j + k];
    if (!(dd > 0.0)) return false;
    dd = sqrt(dd);
    A[7 * j] = dd;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double s = A[6 * i + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= A[6 * i + k] * A[6 * j + k];
      A[6 * i + j] = s / dd;
    }
  }
  double yv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[6 * i + k] * yv[k]; yv[i] = s / A[7 * i]; }
#pragma unroll
  for (int i = 5; i >= 0; --i) { double s = yv[i]; for (int k = i + 1; k < 6; ++k) s -= A[6 * k + i] * x[k]; x[i] = s / A[7 * i]; }
  return true;
}

struct State { Q4 q; V3 t; };

__device__ __forceinline__ State plus_state(const State& s, const double* dl, bool q_const, bool t_const) {
  State o = s;
  if (!q_const) o.q = quat_plus(s.q, V3{dl[0], dl[1], dl[2]});
  if (!t_const) o.t = V3{s.t.x + dl[3], s.t.y + dl[4], s.t.z + dl[5]};
  return o;
}
__device__ __forceinline__ double diff_sq(const State& a, const State& b, bool q_const, bool t_const) {
  double s = 0.0;
  if (!q_const) { const double dx = a.q.x - b.q.x, dy = a.q.y - b.q.y, dz = a.q.z - b.q.z, dw = a.q.w - b.q.w; s += dx * dx + dy * dy + dz * dz + dw * dw; }
  if (!t_const) { const double dx = a.t.x - b.t.x, dy = a.t.y - b.t.y, dz = a.t.z - b.t.z; s += dx * dx + dy * dy + dz * dz; }
  return s;
}
__device__ __forceinline__ double diff_max(const State& a, const State& b, bool q_const, bool t_const) {
  double m = 0.0;
  if (!q_const) m = fmax(fmax(fabs(a.q.x - b.q.x), fabs(a.q.y - b.q.y)), fmax(fabs(a.q.z - b.q.z), fabs(a.q.w - b.q.w)));
  if (!t_const) m = fmax(m, fmax(fabs(a.t.x - b.t.x), fmax(fabs(a.t.y - b.t.y), fabs(a.t.z - b.t.z))));
  return m;
}

// one wave = one problem
__global__ __launch_bounds__(64, 2) void k_reproj_solve(int n_problems, const long long* __restrict__ ptr, const double* __restrict__ points,
                                                     const double* __restrict__ obs, Intr K, double* __restrict__ qs, double* __restrict__ ts,
                                                     pgo_reproj_options o, pgo_reproj_summary* __restrict__ out) {
  __shared__ double red[28 * 65];
  __shared__ double tot[28];
  // lane-uniform LM state lives in LDS (one wave per workgroup), not in vector registers: with it in VGPRs the kernel
  // needed 256 VGPRs + AGPRs and ran one wave per SIMD
  __shared__ double H[36], g[6], gs[6], scale[6], diag[6];
  const int pb = blockIdx.x, lane = threadIdx.x;
  if (pb >= n_problems) return;
  const long long p0 = ptr[pb], p1 = ptr[pb + 1];
  const bool q_const = o.q_constant != 0, t_const = o.t_constant != 0;
  State x{Q4{qs[4 * pb], qs[4 * pb + 1], qs[4 * pb + 2], qs[4 * pb + 3]}, V3{ts[3 * pb], ts[3 * pb + 1], ts[3 * pb + 2]}};
  if (lane < 6) { scale[lane] = 1.0; diag[lane] = 0.0; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double x_cost = 0.0, gmax = 0.0;
  bool scaled_once = false;
  auto is_const = [&](int i) { return i < 3 ? q_const : t_const; };

  auto cost_at = [&](const State& s) {
    double c = 0.0;
    for (long long i = p0 + lane; i < p1; i += 64) {
      double r[2], Jd[12];
      reproj_point(K, obs + 2 * i, s.q, s.t, V3{points[3 * i], points[3 * i + 1], points[3 * i + 2]}, false, r, Jd);
      double rho0, rho1;
      loss_eval(o.loss_kind, o.loss_a, r[0] * r[0] + r[1] * r[1], &rho0, &rho1);
      c += 0.5 * rho0;
    }
    return wave_sum_r(c);
  };
  auto linearize = [&]() {
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0.0;
    for (long long i = p0 + lane; i < p1; i += 64) {
      double r[2], J[12];
      reproj_point(K, obs + 2 * i, x.q, x.t, V3{points[3 * i], points[3 * i + 1], points[3 * i + 2]}, true, r, J);
      double rho0, rho1;
      loss_eval(o.loss_kind, o.loss_a, r[0] * r[0] + r[1] * r[1], &rho0, &rho1);
      acc[27] += 0.5 * rho0;
      int k = 0;
#pragma unroll
      for (int u = 0; u < 6; ++u) {
#pragma unroll
        for (int v = u; v < 6; ++v) { acc[k] += rho1 * (J[u] * J[v] + J[6 + u] * J[6 + v]); ++k; }
        acc[21 + u] += rho1 * (J[u] * r[0] + J[6 + u] * r[1]);
      }
    }
    // fold the 28 per-lane sums through LDS: lane k < 28 adds column k in lane order, the totals are read back by all
    // (28 DPP wave reductions cost as much as the point loop itself; ~120 LDS operations instead of ~1700 DPP moves)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 28; ++k) red[k * 65 + lane] = acc[k];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 28) {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int l = 0; l < 64; l += 4) {
        s0 += red[lane * 65 + l]; s1 += red[lane * 65 + l + 1]; s2 += red[lane * 65 + l + 2]; s3 += red[lane * 65 + l + 3];
      }
      tot[lane] = (s0 + s1) + (s2 + s3);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // lane 0 builds the (scaled) system in LDS
    double neg[6];
    if (lane == 0) {
      int k = 0;
      for (int u = 0; u < 6; ++u)
        for (int v = u; v < 6; ++v) { const double hv = (is_const(u) || is_const(v)) ? 0.0 : tot[k]; H[6 * u + v] = hv; H[6 * v + u] = hv; ++k; }
      for (int u = 0; u < 6; ++u) { g[u] = is_const(u) ? 0.0 : tot[21 + u]; if (is_const(u)) H[7 * u] = 1.0; }
      if (o.jacobi_scaling) {
        if (!scaled_once) for (int u = 0; u < 6; ++u) scale[u] = 1.0 / (1.0 + sqrt(is_const(u) ? 0.0 : H[7 * u]));
        for (int u = 0; u < 6; ++u) for (int v = 0; v < 6; ++v) H[6 * u + v] *= scale[u] * scale[v];
        for (int u = 0; u < 6; ++u) if (is_const(u)) H[7 * u] = 1.0;
      }
      for (int u = 0; u < 6; ++u) gs[u] = g[u] * scale[u];
    }
    scaled_once = true;
    x_cost = tot[27];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 6; ++u) neg[u] = -g[u];
    gmax = diff_max(x, plus_state(x, neg, q_const, t_const), q_const, t_const);
  };
  auto x_norm_of = [&](const State& s) {
    double v = 0.0;
    if (!q_const) v += s.q.x * s.q.x + s.q.y * s.q.y + s.q.z * s.q.z + s.q.w * s.q.w;
    if (!t_const) v += s.t.x * s.t.x + s.t.y * s.t.y + s.t.z * s.t.z;
    return sqrt(v);
  };

  double radius = o.initial_trust_region_radius, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  double x_norm = x_norm_of(x);
  linearize();
  const double initial_cost = x_cost;
  int iteration = 0, n_records = 0, n_ok = 0, n_bad = 0, invalid = 0, term = 1, reason = 5;
  bool last_ok = true;
  double last_gmax = gmax;
  for (;;) {
    if (last_ok) ++n_ok; else ++n_bad;
    ++n_records;   // FinalizeIterationAndCheckIfMinimizerCanContinue records the iteration; one ended by a tolerance test below is not recorded
    if (iteration >= o.max_num_iterations) { term = 1; reason = 5; break; }
    if (last_ok && last_gmax <= o.gradient_tolerance) { term = 0; reason = 3; break; }
    if (radius <= o.min_trust_region_radius) { term = 0; reason = 4; break; }
    ++iteration;
    if (!reuse_diagonal) {
      if (lane < 6) diag[lane] = fmin(fmax(H[7 * lane], o.min_lm_diagonal), o.max_lm_diagonal);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    double A[36], step[6];
#pragma unroll
    for (int k = 0; k < 36; ++k) A[k] = H[k];
#pragma unroll
    for (int u = 0; u < 6; ++u) A[7 * u] += diag[u] / radius;
    bool lin_ok = solve6(A, gs, step);
#pragma unroll
    for (int u = 0; u < 6; ++u) { if (!isfinite(step[u])) lin_ok = false; step[u] = -step[u]; }
    reuse_diagonal = true;
    double model_cost_change = 0.0;
    bool step_valid = false;
    if (lin_ok) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        if (is_const(u)) continue;
        double hs = 0.0;
#pragma unroll
        for (int v = 0; v < 6; ++v) hs += H[6 * u + v] * step[v];
        a += step[u] * gs[u];
        b += step[u] * hs;
      }
      model_cost_change = -a - 0.5 * b;
      step_valid = model_cost_change > 0.0;
    }
    if (!step_valid) {
      ++invalid;
      if (invalid >= o.max_num_consecutive_invalid_steps) { term = 2; reason = 6; break; }
      radius *= 0.5;
      last_ok = false;
      continue;
    }
    invalid = 0;
    double delta[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) delta[u] = is_const(u) ? 0.0 : step[u] * scale[u];
    const State cand = plus_state(x, delta, q_const, t_const);
    const double cand_cost = cost_at(cand);
    const double step_norm = sqrt(diff_sq(x, cand, q_const, t_const));
    if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) { term = 0; reason = 2; break; }
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= o.function_tolerance * x_cost) { term = 0; reason = 1; break; }
    const double rel_dec = cost_change / model_cost_change;
    if (rel_dec > o.min_relative_decrease) {
      x = cand;
      x_norm = x_norm_of(x);
      linearize();
      last_ok = true;
      last_gmax = gmax;
      const double w = 2.0 * rel_dec - 1.0;
      radius = radius / fmax(1.0 / 3.0, 1.0 - w * w * w);
      radius = fmin(o.max_trust_region_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = false;
    } else {
      last_ok = false;
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
    }
  }
  if (lane == 0) {
    qs[4 * pb] = x.q.x; qs[4 * pb + 1] = x.q.y; qs[4 * pb + 2] = x.q.z; qs[4 * pb + 3] = x.q.w;
    ts[3 * pb] = x.t.x; ts[3 * pb + 1] = x.t.y; ts[3 * pb + 2] = x.t.z;
    if (out) {
      pgo_reproj_summary s;
      s.termination_type = term; s.reason = reason; s.num_iterations = n_records;
      s.num_successful_steps = n_ok; s.num_unsuccessful_steps = n_bad; s.num_points = (int)(p1 - p0);
      s.initial_cost = initial_cost; s.final_cost = x_cost;
      out[pb] = s;
    }
  }
}

struct Buf {
  void* p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
};

}  // namespace

#define RP_TRY(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return pgo_candidates_set_error(PGO_ERR_HIP, hipGetErrorString(e_));    \
  } while (0)

extern "C" void pgo_reproj_options_init(pgo_reproj_options* o) {
  if (!o) return;
  o->max_num_iterations = 1000;        // MotionEstimate.cc:121
  o->q_constant = 1;                   // MotionEstimate.cc:108
  o->t_constant = 0;
  o->loss_kind = PGO_LOSS_HUBER;       // MotionEstimate.cc:73
  o->loss_a = 1.0;
  o->jacobi_scaling = 1;
  o->max_num_consecutive_invalid_steps = 5;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
}

extern "C" int pgo_reproj_solve_batch(int n_problems, const long long* point_ptr, const double* points, const double* observations,
                                      const double intrinsics[4], double* q, double* t, const pgo_reproj_options* options,
                                      pgo_reproj_summary* summaries, double* kernel_ms) {
  if (n_problems < 0 || !point_ptr || !intrinsics || !q || !t || !options)
    return pgo_candidates_set_error(PGO_ERR_INVALID_ARGUMENT, "null argument to pgo_reproj_solve_batch");
  if (options->loss_kind < PGO_LOSS_TRIVIAL || options->loss_kind > PGO_LOSS_SWITCHABLE || (options->loss_kind != PGO_LOSS_TRIVIAL && !(options->loss_a > 0.0)))
    return pgo_candidates_set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_reproj_solve_batch: bad loss");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    return pgo_candidates_set_error(PGO_ERR_NO_DEVICE, "no HIP device: the batched reprojection solve has no CPU fallback");
  }
  if (n_problems == 0) return PGO_OK;
  const long long total = point_ptr[n_problems];
  if (point_ptr[0] != 0 || total < 0 || (total > 0 && (!points || !observations)))
    return pgo_candidates_set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_reproj_solve_batch: bad point_ptr / points");
  Buf d_ptr, d_pts, d_obs, d_q, d_t, d_sum;
  RP_TRY(d_ptr.alloc(sizeof(long long) * ((size_t)n_problems + 1)));
  RP_TRY(d_pts.alloc(sizeof(double) * 3 * (size_t)total));
  RP_TRY(d_obs.alloc(sizeof(double) * 2 * (size_t)total));
  RP_TRY(d_q.alloc(sizeof(double) * 4 * (size_t)n_problems));
  RP_TRY(d_t.alloc(sizeof(double) * 3 * (size_t)n_problems));
  RP_TRY(d_sum.alloc(sizeof(pgo_reproj_summary) * (size_t)n_problems));
  RP_TRY(hipMemcpy(d_ptr.p, point_ptr, sizeof(long long) * ((size_t)n_problems + 1), hipMemcpyHostToDevice));
  if (total > 0) {
    RP_TRY(hipMemcpy(d_pts.p, points, sizeof(double) * 3 * (size_t)total, hipMemcpyHostToDevice));
    RP_TRY(hipMemcpy(d_obs.p, observations, sizeof(double) * 2 * (size_t)total, hipMemcpyHostToDevice));
  }
  RP_TRY(hipMemcpy(d_q.p, q, sizeof(double) * 4 * (size_t)n_problems, hipMemcpyHostToDevice));
  RP_TRY(hipMemcpy(d_t.p, t, sizeof(double) * 3 * (size_t)n_problems, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  RP_TRY(hipEventCreate(&e0));
  RP_TRY(hipEventCreate(&e1));
  const Intr K{intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
  RP_TRY(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_reproj_solve, dim3(n_problems), dim3(64), 0, 0, n_problems, (const long long*)d_ptr.p, (const double*)d_pts.p,
                     (const double*)d_obs.p, K, (double*)d_q.p, (double*)d_t.p, *options, (pgo_reproj_summary*)d_sum.p);
  RP_TRY(hipEventRecord(e1, 0));
  RP_TRY(hipEventSynchronize(e1));
  RP_TRY(hipGetLastError());
  float ms = 0.f;
  RP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  RP_TRY(hipMemcpy(q, d_q.p, sizeof(double) * 4 * (size_t)n_problems, hipMemcpyDeviceToHost));
  RP_TRY(hipMemcpy(t, d_t.p, sizeof(double) * 3 * (size_t)n_problems, hipMemcpyDeviceToHost));
  if (summaries) RP_TRY(hipMemcpy(summaries, d_sum.p, sizeof(pgo_reproj_summary) * (size_t)n_problems, hipMemcpyDeviceToHost));
  if (kernel_ms) *kernel_ms = ms;
  return PGO_OK;
}
