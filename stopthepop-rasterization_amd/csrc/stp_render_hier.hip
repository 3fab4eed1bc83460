// stp_render_hier.hip -- dispatch of the hierarchical kernel over the queue-size ladder
// (reference forward.cu:445-494, backward.cu:739-767); the kernels themselves are instantiated in
// slices by stp_render_hier_inst.hip.
#include "stp_internal.h"
#include "stp_blend.h"

namespace stp {



#define STP_DECL(name) hipError_t name(const FrameParams& f, const RenderArgs& a, hipStream_t st, bool* handled)
STP_DECL(launch_hier_fwd_mid8);
STP_DECL(launch_hier_bwd_mid8);
STP_DECL(launch_hier_rec_mid8);
STP_DECL(launch_hier_dbg_mid8);
#ifndef STP_FASTBUILD
STP_DECL(launch_hier_dbg_mid12);
STP_DECL(launch_hier_dbg_mid20);
STP_DECL(launch_hier_fwd_mid12);
STP_DECL(launch_hier_fwd_mid20);
STP_DECL(launch_hier_bwd_mid12);
STP_DECL(launch_hier_bwd_mid20);
STP_DECL(launch_hier_rec_mid12);
STP_DECL(launch_hier_rec_mid20);
#endif
#undef STP_DECL

// mode: 0 forward, 1 resorting backward, 2 recording forward, 3 forward of the debug depth visualisation
static hipError_t dispatch(int mode, const FrameParams& f, const RenderArgs& a, hipStream_t st, std::string* err)
{
    const bool backward = mode == 1;
    const int head = f.s.queue_per_pixel, mid = f.s.queue_tile_2x2;
    bool handled = false, mid_ok = false;
    hipError_t e = hipErrorInvalidValue;
#define STP_PICK(M) (mode == 1 ? launch_hier_bwd_mid##M(f, a, st, &handled) : mode == 2 ? launch_hier_rec_mid##M(f, a, st, &handled) : \
                     mode == 3 ? launch_hier_dbg_mid##M(f, a, st, &handled) : launch_hier_fwd_mid##M(f, a, st, &handled))
    if (mid == 8) { mid_ok = true; e = STP_PICK(8); }
#ifndef STP_FASTBUILD
    else if (mid == 12) { mid_ok = true; e = STP_PICK(12); }
    else if (mid == 20) { mid_ok = true; e = STP_PICK(20); }
#endif
#undef STP_PICK
    if (handled) return e;
    if (err) {
        if (!mid_ok) *err = "Not supported mid queue size" + (backward ? " " + std::to_string(mid) : std::string());
        else *err = "Not supported head queue size" + (backward ? " " + std::to_string(head) : std::string());
    }
    return hipErrorInvalidValue;
}

hipError_t launch_hier_fwd(const FrameParams& f, const RenderArgs& a, hipStream_t st, std::string* err) { return dispatch(0, f, a, st, err); }
hipError_t launch_hier_bwd(const FrameParams& f, const RenderArgs& a, hipStream_t st, std::string* err) { return dispatch(1, f, a, st, err); }
hipError_t launch_hier_rec(const FrameParams& f, const RenderArgs& a, hipStream_t st, std::string* err) { return dispatch(2, f, a, st, err); }
hipError_t launch_hier_dbg(const FrameParams& f, const RenderArgs& a, hipStream_t st, std::string* err) { return dispatch(3, f, a, st, err); }

} // namespace stp
