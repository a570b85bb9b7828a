// deform.hip -- placeholder until the fused HexPlane + MLP kernels land (next commit).
#include "common.h"
using namespace fdgs;
extern "C" int fdgs_deform_fwd(void*, const fdgs_deform_params*, const fdgs_deform_out*) {
    return fail(FDGS_E_INVALID, "%s", "fdgs_deform_fwd: not built yet");
}
extern "C" int fdgs_deform_bwd_scratch_bytes(const fdgs_deform_params*, size_t*) {
    return fail(FDGS_E_INVALID, "%s", "fdgs_deform_bwd_scratch_bytes: not built yet");
}
extern "C" int fdgs_deform_bwd(void*, const fdgs_deform_params*, const fdgs_deform_grads*) {
    return fail(FDGS_E_INVALID, "%s", "fdgs_deform_bwd: not built yet");
}
