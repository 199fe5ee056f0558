// conv_x3_lean.hip -- the bf16x3 implicit-GEMM kernels of conv_x3_kernels.h instantiated with the LEAN forward epilogue of conv_igemm.h (round 6): EPI = 1,
// a training step's forward (raw result + statistics partials, bit-identical to the shared epilogue's) -- on the tiles the automatic rule of conv_x3.hip
// picks.  (EPI = 2, the lean data-gradient form, is used by conv_x3f.hip's 1x1 kernels only: see conv_x3.hip, dispatch_x3.)  A translation unit of its own so that the two sets of
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
    if (epi != 1) {
        straps_set_error("conv_x3_lean: only the forward form (EPI = 1) is instantiated for the plane kernels");
        return STRAPS_EUNSUPPORTED;
    }
    return dispatch_lean<1>(p, halo, cfg, st);
}
