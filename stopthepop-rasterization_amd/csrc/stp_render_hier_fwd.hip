// stp_render_hier_fwd.hip -- forward instantiations + dispatch of the hierarchical kernel.
// Queue-size ladder as in reference forward.cu:445-494 (HEAD in {4,8,16}, MID in {8,12,20}).
// -DSTP_FASTBUILD compiles the default queue sizes only (the reference's STOPTHEPOP_FASTBUILD,
// rasterizer.h:15-17,50-58).
#include "stp_render_hier.inc"

namespace stp {

hipError_t launch_hier_fwd(const FrameParams& f, const RenderArgs& a, hipStream_t st, std::string* err)
{
    const int head = f.s.queue_per_pixel, mid = f.s.queue_tile_2x2;
    const bool cull = f.s.hierarchical_4x4_culling != 0;
#define STP_GO(H, M) return cull ? launch_hier_one<H, M, true, false>(f, a, st) : launch_hier_one<H, M, false, false>(f, a, st)
#ifdef STP_FASTBUILD
    if (head == 4 && mid == 8) STP_GO(4, 8);
#else
    if (mid == 8) {
        if (head == 4) STP_GO(4, 8);
        if (head == 8) STP_GO(8, 8);
        if (head == 16) STP_GO(16, 8);
    } else if (mid == 12) {
        if (head == 4) STP_GO(4, 12);
        if (head == 8) STP_GO(8, 12);
        if (head == 16) STP_GO(16, 12);
    } else if (mid == 20) {
        if (head == 4) STP_GO(4, 20);
        if (head == 8) STP_GO(8, 20);
        if (head == 16) STP_GO(16, 20);
    }
#endif
#undef STP_GO
    if (err) *err = (mid == 8 || mid == 12 || mid == 20) ? "Not supported head queue size" : "Not supported mid queue size";
    return hipErrorInvalidValue;
}

} // namespace stp
