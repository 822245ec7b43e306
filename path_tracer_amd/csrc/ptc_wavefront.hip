// Wavefront integrator (placeholder until the queue kernels land).
#include "ptc_internal.h"

namespace ptc
{
int launch_render_wavefront(ptc_context *, const ptc_scene *, ptc_film *, const DFrame &)
{
    set_error("wavefront integrator not built yet");
    return PTC_E_INVALID;
}
} // namespace ptc
