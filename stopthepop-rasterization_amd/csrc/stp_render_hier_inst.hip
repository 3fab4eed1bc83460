// stp_render_hier_inst.hip -- one slice of the hierarchical kernel's instantiation ladder.
// Compiled several times by the Makefile with -DSTP_INST_MID={8,12,20} -DSTP_INST_MODE={0,1,2}
// (0 forward, 1 resorting backward, 2 recording forward, 3 depth-visualisation forward), so the
// template instantiations (HEAD x CULL per slice) build in parallel.  Queue-size ladders follow the
// reference: forward HEAD in {4,8,16} (forward.cu:465-472), backward HEAD in {4,8,12,16}
// (backward.cu:745-752), MID in {8,12,20}.  -DSTP_FASTBUILD keeps only HEAD 4 (with MID 8), the
// counterpart of the reference's STOPTHEPOP_FASTBUILD (rasterizer.h:15-17,50-58).
#include "stp_render_hier.inc"

#ifndef STP_INST_MID
#error "STP_INST_MID must be defined"
#endif
#ifndef STP_INST_MODE
#error "STP_INST_MODE must be defined"
#endif

#define STP_CAT2(a, b) a##b
#define STP_CAT(a, b) STP_CAT2(a, b)
#if STP_INST_MODE == 1
#define STP_FN STP_CAT(launch_hier_bwd_mid, STP_INST_MID)
#elif STP_INST_MODE == 2
#define STP_FN STP_CAT(launch_hier_rec_mid, STP_INST_MID)
#elif STP_INST_MODE == 3
#define STP_FN STP_CAT(launch_hier_dbg_mid, STP_INST_MID)
#else
#define STP_FN STP_CAT(launch_hier_fwd_mid, STP_INST_MID)
#endif

namespace stp {

// returns hipErrorInvalidValue with *handled = false when this slice has no kernel for `head`
hipError_t STP_FN(const FrameParams& f, const RenderArgs& a, hipStream_t st, bool* handled)
{
    constexpr int MID = STP_INST_MID;
    constexpr int MODE = STP_INST_MODE;
    constexpr bool BWD = MODE == 1;
    const int head = f.s.queue_per_pixel;
    const bool cull = f.s.hierarchical_4x4_culling != 0;
    *handled = true;
#define STP_GO(H) return cull ? launch_hier_one<H, MID, true, MODE>(f, a, st) : launch_hier_one<H, MID, false, MODE>(f, a, st)
    // the default queue sizes' forward passes also exist without the reciprocal's domain check, for frames whose Sigma^-1
    // entries are all tame (every frame of a sane scene: FrameParams::wild_cov, from preprocess_kernel's status word)
    if constexpr (!BWD && MID == 8) {
        if (head == 4 && !f.wild_cov) return cull ? launch_hier_one<4, MID, true, MODE, true>(f, a, st) : launch_hier_one<4, MID, false, MODE, true>(f, a, st);
    }
    if (head == 4) STP_GO(4);
#ifndef STP_FASTBUILD
    if (head == 8) STP_GO(8);
    if (BWD && head == 12) STP_GO(12);
    if (head == 16) STP_GO(16);
#endif
#undef STP_GO
    *handled = false;
    return hipErrorInvalidValue;
}

} // namespace stp
