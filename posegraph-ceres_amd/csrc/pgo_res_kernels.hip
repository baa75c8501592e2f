// pgo_res_kernels.hip — the RESIDENT universal stream (pgo_uni_resident.h: k_res_lh, k_res_cg, launch_uni_r) as a translation unit of its
// own: the same text as pgo_kernels.hip (the kernels share its device functions), of which only the resident stream's launcher is
// instantiated here.  Why two units: pgo_kernels.hip, "launchers".
#define PGO_TU_RESIDENT 1
#include "pgo_kernels.hip"
