// pgo_lean_kernels.hip — k_linearize_lean: the linearisation of a symmetric-form session (pgo_sym.h) with the per-incidence algebra
// of pgo_lin_lean.h.  Same work split, same outputs as k_linearize_symout (pgo_kernels.hip): one lane per incidence slot, the
// off-diagonal block stored at the slot's place g.sym_dst[t] of the form (none for the mirrored incidence of an interior edge), the
// row's diagonal block and gradient summed through LDS.  What differs, for gfx950:
//   * about 0.75 of the instructions (0.68 of the FP64 ones) — the arithmetic is 4 us of the launch now;
//   * every load whose address does not depend on another load is issued at the top, in front of the side test (indices, place in
//     the form, measurement, information): the dependent chain is indices -> poses / scales -> arithmetic, one step shorter;
//     addresses are a scalar base + ONE 32-bit offset register per lane (64-bit addresses per plane cost 30 registers);
//   * a value leaves the registers as soon as it is final: the block's quadrants to HBM when complete, the 27 row values to LDS
//     one by one — after the two lanes of an even/odd pair have been added on the DPP crossbar: the first slot of a row is its
//     diagonal slot, whose 27 values are zero, so a pair never mixes two rows.  Half the LDS (27 KiB per 256 slots instead of 54),
//     half the LDS reads of the row sums, and LDS no longer caps the work-groups per CU at two.
// Algorithmic bytes per launch as for k_linearize: 640 E + 392 N.
// Information with a position / rotation coupling (INFO 1) stays with k_linearize_symout.
#include <cstdlib>

#include "pgo_kernels.h"
#include "pgo_lin_lean.h"
#include "pgo_wave.h"

namespace pgo {

namespace {

#include "pgo_lean_body.h"

template <int INFO, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_linearize_lean(DeviceGraph g, int gate) {
  extern __shared__ double lds[];      // LEAN_NV doubles per lane PAIR
  if (gate == 1 && !g.cg->done) return;
  if (gate == 2 && (g.lm->halt || !g.lm->accepted)) return;
  lean_linearize_body<INFO, false>(g, lds);
}

}  // namespace

template <int WAVES>
static void launch_lean_w(const DeviceGraph& g, hipStream_t s, int gate) {
  const size_t lds = (size_t)LEAN_NV * (g.block / 2) * sizeof(double);
  const dim3 grid(g.n_wg), block(g.block);
  if (g.info_mode == 3) hipLaunchKernelGGL((k_linearize_lean<3, WAVES>), grid, block, lds, s, g, gate);
  else if (g.info_mode == 2) hipLaunchKernelGGL((k_linearize_lean<2, WAVES>), grid, block, lds, s, g, gate);
  else hipLaunchKernelGGL((k_linearize_lean<0, WAVES>), grid, block, lds, s, g, gate);
}
bool linearize_lean_fits(const DeviceGraph& g) {
  // no position / rotation coupling in the information; 32-bit byte offsets: 21 information planes, the form (at most one 288-byte
  // place per incidence slot) and the pose array each below 4 GiB
  return g.info_mode != 1 && (long long)g.n_slots < 14000000LL && (long long)g.N < 60000000LL;
}
void launch_linearize_lean(const DeviceGraph& g, hipStream_t s, int gate) {
  if (!linearize_lean_fits(g)) { launch_linearize_symout(g, s, gate); return; }
  // three waves per SIMD: 168 registers without a spill (INFO 0, 3); block-diagonal information needs 12 bytes of scratch there: two
  const int waves = g.info_mode == 2 ? 2 : 3;
  if (waves == 3) launch_lean_w<3>(g, s, gate);
  else launch_lean_w<2>(g, s, gate);
}

}  // namespace pgo
