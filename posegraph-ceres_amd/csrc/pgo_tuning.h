// pgo_tuning.h — development / test knobs of the library: ONE process-wide table behind pgo_tuning_set (include/pgo.h).  r06: these were
// sixteen of the library's 31 environment variables; an environment variable is read by whatever process inherits it, a knob is set by the
// code that wants it (the tests name every one: tests/test_capi_cpu.py, and the A/B tests that use them).  Header only, no HIP: the
// host-only tools (tools/front_check_cli.cpp) compile pgo_front.cpp with it.
#pragma once
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace pgo {

// name, meaning (default)
struct TuningKnob { const char* name; const char* what; };
inline const TuningKnob* tuning_knobs(int* n) {
  static const TuningKnob knobs[] = {
      {"shard_boundary", "0: sharded sessions all-gather whole segments per CG iteration instead of the ranks' boundary rows (1; A/B)"},
      {"debug", "DeviceGraph::debug: timing ablations of the CG kernels, tools/ablate.py (0; results are wrong with a bit set)"},
      {"graph", "1: replay captured hipGraphs of the launch batches instead of enqueueing eagerly (0)"},
      {"front_zfrac", "multifrontal plan, relaxed amalgamation: admissible fraction of explicit zeros (0.25)"},
      {"front_small", "multifrontal plan: fronts of at most this many block columns always merge (8)"},
      {"front_maxcols", "multifrontal plan: cap on the pivot columns of a merged front (none)"},
      {"front_mixed", "multifrontal plan: subtrees of at most this many scalars take the small-front kernels (0 = off)"},
      {"front_tile32_below", "multifrontal plan: 64-tile count below which a Schur update uses 32 x 32 tiles (192)"},
      {"factor_fused", "1 / 0: always / never the single-launch form of the factorisations (default: up to 3 GFLOP)"},
      {"sym_lin_rows", "1: the symmetric form is written by the general linearisation body on every graph (0: the lean writer where it fits)"},
      {"sym_repack", "1: symmetric-form sessions keep the incidence-slot linearisation and copy its blocks per LM iteration (0)"},
      {"sym_rows", "rows per tile of the symmetric form (default: by the size of the rank's share)"},
      {"pipeline_pcg", "1: PCG sessions may use the allotted-sequence driver where the universal stream is off (0)"},
      {"hybrid_budget", "CG iterations a PCG try of the hybrid exact solver may take (default: from the plan's estimate)"},
      {"no_diag_info", "1: diagonal information is read as block-diagonal information, info_mode 2 (0)"},
      {"resident_abort_test", "1: the resident stream's abort word is set before the first launch (0; tests/test_gpu_resident.py)"},
  };
  *n = (int)(sizeof knobs / sizeof knobs[0]);
  return knobs;
}
struct TuningTable { std::mutex mu; std::map<std::string, double> v; };
inline TuningTable& tuning_table() { static TuningTable t; return t; }
inline bool tuning_known(const char* name) {
  int n = 0;
  const TuningKnob* k = tuning_knobs(&n);
  for (int i = 0; i < n; ++i) if (name && !std::strcmp(k[i].name, name)) return true;
  return false;
}
// the knob's value, or dflt while it is not set
inline double tuning(const char* name, double dflt) {
  TuningTable& t = tuning_table();
  std::lock_guard<std::mutex> lk(t.mu);
  auto it = t.v.find(name);
  return it == t.v.end() ? dflt : it->second;
}
inline bool tuning_is_set(const char* name) {
  TuningTable& t = tuning_table();
  std::lock_guard<std::mutex> lk(t.mu);
  return t.v.count(name) != 0;
}
// NaN: back to the default.  false: no such knob.
inline bool tuning_set(const char* name, double value) {
  if (!tuning_known(name)) return false;
  TuningTable& t = tuning_table();
  std::lock_guard<std::mutex> lk(t.mu);
  if (std::isnan(value)) t.v.erase(name); else t.v[name] = value;
  return true;
}

}  // namespace pgo
