// tools/generate_graph.cpp — the synthetic pose-graph generators of SURVEY.md §8d in C++ (Manhattan-SE3 of BASELINE config 2 /
// 4, sphere layers of config 5), writing g2o text (VERTEX_SE3:QUAT = dead-reckoning start, EDGE_SE3:QUAT with the upper
// triangle of the information matrix) that tools/pose_graph_solve, tools/ceres_baseline and datasets.read_g2o read.
// Same model as posegraph-ceres_amd/datasets.py (which stays the generator of the committed fixtures: numpy's PCG64 stream is
// not reproduced here — this one draws from std::mt19937_64, so the graphs are statistically, not bitwise, the same):
//   manhattan: 1 m steps along the heading, yaw +-90 deg w.p. 0.30, pitch +-90 deg w.p. 0.05, N(0, 0.01 rad) attitude jitter;
//              loop edges between poses closer than `radius` with id gap > 20, drawn until the edge count is met;
//   sphere   : `layers` spheres of rings x per_ring poses (radius 50 m, centres 110 m apart), chain + meridian edges +
//              random chords within `radius`;
//   both     : measurement = true relative pose (+) noise (sigma_t = 0.05 m, sigma_r = 0.01 rad as a half-angle vector),
//              information = diag(1/sigma^2), start = odometry dead reckoning.
// usage: generate_graph manhattan <poses> <edges> <seed> <out.g2o> [radius=3]
//        generate_graph sphere <layers> <rings> <per_ring> <edges> <seed> <out.g2o> [radius=8]
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

struct Q { double x, y, z, w; };
struct V { double x, y, z; };
struct Pose { V p; Q q; };

Q mul(const Q& a, const Q& b) {
  return Q{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
           a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
Q conj(const Q& q) { return Q{-q.x, -q.y, -q.z, q.w}; }
Q normalized(const Q& q) { const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w); return Q{q.x / n, q.y / n, q.z / n, q.w / n}; }
V rot(const Q& q, const V& v) {   // R(q) v
  const Q t = mul(mul(q, Q{v.x, v.y, v.z, 0.0}), conj(q));
  return V{t.x, t.y, t.z};
}
Q exp_half(const V& h) {          // [sin|h| h/|h| ; cos|h|]: h is the HALF rotation vector (EigenQuaternionParameterization)
  const double n = std::sqrt(h.x * h.x + h.y * h.y + h.z * h.z);
  if (n == 0.0) return Q{0, 0, 0, 1};
  const double s = std::sin(n) / n;
  return Q{s * h.x, s * h.y, s * h.z, std::cos(n)};
}

struct Edge { int a, b; V t; Q r; };   // a = id_begin, b = id_end, (t, r) = pose of b in the frame of a

// all pairs (i > j) closer than `radius` with id gap > min_gap, in random order, no duplicates: uniform grid of cell = radius
std::vector<std::pair<int, int>> loop_pairs(const std::vector<Pose>& truth, std::mt19937_64& rng, double radius, int min_gap) {
  const int n = (int)truth.size();
  auto key = [&](double x, double y, double z) {
    const long long cx = (long long)std::floor(x / radius), cy = (long long)std::floor(y / radius), cz = (long long)std::floor(z / radius);
    return (uint64_t)((cx + (1LL << 20)) | ((cy + (1LL << 20)) << 21) | ((cz + (1LL << 20)) << 42));
  };
  std::unordered_map<uint64_t, std::vector<int>> grid;
  for (int i = 0; i < n; ++i) grid[key(truth[i].p.x, truth[i].p.y, truth[i].p.z)].push_back(i);
  std::vector<std::pair<int, int>> all;
  for (int i = 0; i < n; ++i) {
    const V& p = truth[i].p;
    for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) for (int dz = -1; dz <= 1; ++dz) {
      auto it = grid.find(key(p.x + dx * radius, p.y + dy * radius, p.z + dz * radius));
      if (it == grid.end()) continue;
      for (int j : it->second) {
        if (i - j <= min_gap) continue;
        const double ex = p.x - truth[j].p.x, ey = p.y - truth[j].p.y, ez = p.z - truth[j].p.z;
        if (ex * ex + ey * ey + ez * ez <= radius * radius) all.emplace_back(i, j);
      }
    }
  }
  std::shuffle(all.begin(), all.end(), rng);
  return all;
}
// as datasets._loop_pairs: the radius grows by 1.5 until at least `want` pairs (beyond `have` pairs already taken) are admissible
std::vector<std::pair<int, int>> enough_pairs(const std::vector<Pose>& truth, std::mt19937_64& rng, double radius, int min_gap, long long want) {
  const long long n = (long long)truth.size(), possible = std::max(0LL, n - min_gap - 1) * std::max(0LL, n - min_gap) / 2;
  if (want > possible) { std::fprintf(stderr, "cannot place %lld loop edges: %lld poses admit only %lld pairs more than %d ids apart\n", want, n, possible, min_gap); std::exit(1); }
  for (double r = radius;; r *= 1.5) {
    auto all = loop_pairs(truth, rng, r, min_gap);
    if ((long long)all.size() >= want) {
      if (r != radius) std::fprintf(stderr, "loop radius grown to %.3f m for %lld pairs\n", r, want);
      return all;
    }
  }
}

void write_g2o(const char* path, const std::vector<Pose>& truth, const std::vector<std::pair<int, int>>& pairs, std::mt19937_64& rng,
               double sigma_t, double sigma_r) {
  std::normal_distribution<double> N01(0.0, 1.0);
  std::vector<Edge> edges;
  edges.reserve(pairs.size());
  for (const auto& ab : pairs) {
    const Pose &A = truth[ab.first], &B = truth[ab.second];
    const Q qa_inv = conj(A.q);
    V t = rot(qa_inv, V{B.p.x - A.p.x, B.p.y - A.p.y, B.p.z - A.p.z});
    Q r = mul(qa_inv, B.q);
    t = V{t.x + sigma_t * N01(rng), t.y + sigma_t * N01(rng), t.z + sigma_t * N01(rng)};
    r = normalized(mul(exp_half(V{sigma_r * N01(rng), sigma_r * N01(rng), sigma_r * N01(rng)}), r));
    edges.push_back(Edge{ab.first, ab.second, t, r});
  }
  // dead reckoning along the chain: the odometry edges are the first n - 1 pairs (i, i - 1): pose(i-1) = pose(i) (+) meas  =>  invert
  const int n = (int)truth.size();
  std::vector<Pose> init(n);
  init[0] = truth[0];
  for (int i = 1; i < n; ++i) {
    const Edge& e = edges[i - 1];        // a = i, b = i - 1: T_prev = T_i * M  =>  T_i = T_prev * M^-1
    const Q mr_inv = conj(e.r);
    const V mt_inv = rot(mr_inv, V{-e.t.x, -e.t.y, -e.t.z});
    const V d = rot(init[i - 1].q, mt_inv);
    init[i].p = V{init[i - 1].p.x + d.x, init[i - 1].p.y + d.y, init[i - 1].p.z + d.z};
    init[i].q = normalized(mul(init[i - 1].q, mr_inv));
  }
  FILE* f = std::fopen(path, "w");
  if (!f) { std::perror(path); std::exit(1); }
  for (int i = 0; i < n; ++i)
    std::fprintf(f, "VERTEX_SE3:QUAT %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", i, init[i].p.x, init[i].p.y, init[i].p.z, init[i].q.x, init[i].q.y, init[i].q.z, init[i].q.w);
  const double wt = 1.0 / (sigma_t * sigma_t), wr = 1.0 / (sigma_r * sigma_r);
  for (const Edge& e : edges) {
    std::fprintf(f, "EDGE_SE3:QUAT %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g", e.a, e.b, e.t.x, e.t.y, e.t.z, e.r.x, e.r.y, e.r.z, e.r.w);
    for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) std::fprintf(f, " %.17g", i == j ? (i < 3 ? wt : wr) : 0.0);
    std::fprintf(f, "\n");
  }
  std::fclose(f);
  std::printf("%s: %d poses, %zu edges\n", path, n, edges.size());
}

}  // namespace

int main(int argc, char** argv) {
  if (argc >= 6 && !std::strcmp(argv[1], "manhattan")) {
    const int n = std::atoi(argv[2]);
    const long long E = std::atoll(argv[3]);
    std::mt19937_64 rng((uint64_t)std::atoll(argv[4]));
    const double radius = argc > 6 ? std::atof(argv[6]) : 3.0;
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::normal_distribution<double> J(0.0, 0.01);
    std::vector<Pose> truth(n);
    truth[0] = Pose{V{0, 0, 0}, Q{0, 0, 0, 1}};
    const double h = std::sqrt(0.5);
    for (int i = 1; i < n; ++i) {
      const double turn = U(rng), sign = U(rng) < 0.5 ? -1.0 : 1.0;
      Q dq = exp_half(V{0.5 * J(rng), 0.5 * J(rng), 0.5 * J(rng)});
      if (turn < 0.30) dq = mul(Q{0, 0, sign * h, h}, dq);
      else if (turn < 0.35) dq = mul(Q{0, sign * h, 0, h}, dq);
      const Q q = normalized(mul(truth[i - 1].q, dq));
      const V d = rot(q, V{1, 0, 0});
      truth[i] = Pose{V{truth[i - 1].p.x + d.x, truth[i - 1].p.y + d.y, truth[i - 1].p.z + d.z}, q};
    }
    std::vector<std::pair<int, int>> pairs;
    for (int i = 1; i < n; ++i) pairs.emplace_back(i, i - 1);
    if (E < n - 1) { std::fprintf(stderr, "edges < poses - 1\n"); return 2; }
    auto loops = enough_pairs(truth, rng, radius, 20, E - (n - 1));
    loops.resize((size_t)(E - (n - 1)));
    std::sort(loops.begin(), loops.end());
    pairs.insert(pairs.end(), loops.begin(), loops.end());
    write_g2o(argv[5], truth, pairs, rng, 0.05, 0.01);
    return 0;
  }
  if (argc >= 8 && !std::strcmp(argv[1], "sphere")) {
    const int layers = std::atoi(argv[2]), rings = std::atoi(argv[3]), per_ring = std::atoi(argv[4]);
    const long long E = std::atoll(argv[5]);
    std::mt19937_64 rng((uint64_t)std::atoll(argv[6]));
    const double chord = argc > 8 ? std::atof(argv[8]) : 8.0, radius = 50.0, pi = 3.14159265358979323846;
    std::vector<Pose> truth;
    for (int s = 0; s < layers; ++s)
      for (int r = 0; r < rings; ++r) {
        const double phi = pi * (r + 0.5) / rings;
        for (int k = 0; k < per_ring; ++k) {
          const double th = 2 * pi * k / per_ring, yaw = th + pi / 2;
          truth.push_back(Pose{V{2.2 * radius * s + radius * std::sin(phi) * std::cos(th), radius * std::sin(phi) * std::sin(th), radius * std::cos(phi)},
                               Q{0, 0, std::sin(yaw / 2), std::cos(yaw / 2)}});
        }
      }
    const int n = (int)truth.size();
    std::vector<std::pair<int, int>> pairs;
    for (int i = 1; i < n; ++i) pairs.emplace_back(i, i - 1);
    for (int i = per_ring; i < n; ++i)
      if (i / (rings * per_ring) == (i - per_ring) / (rings * per_ring)) pairs.emplace_back(i, i - per_ring);   // meridian neighbour
    if (E > (long long)pairs.size()) {
      std::unordered_set<long long> have;
      for (const auto& ab : pairs) have.insert((long long)ab.first * n + ab.second);
      const auto loops = enough_pairs(truth, rng, chord, per_ring + 1, (E - (long long)pairs.size()) + (long long)pairs.size());   // duplicates of the chain / meridian edges may be among them
      std::vector<std::pair<int, int>> extra;
      for (const auto& ab : loops) {
        if ((long long)(pairs.size() + extra.size()) >= E) break;
        if (have.insert((long long)ab.first * n + ab.second).second) extra.push_back(ab);
      }
      if ((long long)(pairs.size() + extra.size()) < E) { std::fprintf(stderr, "could not place %lld edges\n", E); return 1; }
      std::sort(extra.begin(), extra.end());
      pairs.insert(pairs.end(), extra.begin(), extra.end());
    }
    write_g2o(argv[7], truth, pairs, rng, 0.05, 0.01);
    return 0;
  }
  std::fprintf(stderr, "usage: %s manhattan <poses> <edges> <seed> <out.g2o> [radius]\n       %s sphere <layers> <rings> <per_ring> <edges> <seed> <out.g2o> [radius]\n", argv[0], argv[0]);
  return 2;
}
