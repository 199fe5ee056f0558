// conv_x3_lean.hip -- the bf16x3 implicit-GEMM kernels of conv_x3_kernels.h instantiated with the LEAN epilogues of conv_igemm.h (round 6): EPI = 1, a
// training step's forward (raw result + statistics partials), EPI = 2, its data gradients (LeanDgradEpilogue: addend with optional ReLU bits, fused
// BatchNorm-backward sums, parity classes of a stride-2 gradient; look-ahead under the last chunk) -- both bit-identical to the shared epilogue's results,
// on the tiles the automatic rule of conv_x3.hip picks.  A translation unit of its own so that the two sets of
// instantiations compile in parallel; conv_x3.hip's dispatch_x3 decides (lean_epilogue_choice) and calls in here.
#include "conv_x3_kernels.h"

namespace {

template <int EPI>
int dispatch_lean(const ConvP& p, int halo, int cfg, hipStream_t st) {
    if (halo == 1) return launch_x3h<128, 128, 2, 2, 3, 208, 2, EPI>(p, st);
    if (halo == 3) return launch_x3h<128, 64, 2, 2, 2, 272, 1, EPI>(p, st);
    switch (cfg) {
        case 3: return launch_x3<64, 64, 2, 2, 3, 0, false, EPI>(p, st);
        case 5: return launch_x3<128, 128, 2, 2, 3, 0, false, EPI>(p, st);
        case 7: return launch_x3<128, 64, 2, 2, 3, 0, false, EPI>(p, st);
        case 9: return launch_x3<128, 128, 4, 2, 3, 0, true, EPI>(p, st);
        case 11: return launch_x3<128, 64, 2, 2, 2, 0, true, EPI>(p, st);
        case 12: return launch_x3<256, 128, 4, 2, 2, 0, true, EPI>(p, st);
        default: break;
    }
    straps_set_error("conv_x3_lean: tile configuration %d has no lean instantiation", cfg);
    return STRAPS_EUNSUPPORTED;
}

}  // namespace

int straps_internal_dispatch_x3_lean(const void* pv, int halo, int cfg, int epi, hipStream_t st) {
    const ConvP& p = *static_cast<const ConvP*>(pv);
    return epi == 1 ? dispatch_lean<1>(p, halo, cfg, st) : dispatch_lean<2>(p, halo, cfg, st);
}
