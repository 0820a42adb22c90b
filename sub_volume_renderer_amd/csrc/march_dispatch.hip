// Dispatch of the march kernel on the number of LODs (one object file per count).
#include "svr_internal.h"

#define SVR_DECL(k) hipError_t svr_launch_march_nl##k(const MarchParams& p, int kind, hipStream_t stream);
SVR_DECL(1) SVR_DECL(2) SVR_DECL(3) SVR_DECL(4) SVR_DECL(5) SVR_DECL(6) SVR_DECL(7) SVR_DECL(8)

hipError_t svr_launch_march(const MarchParams& p, int kind, hipStream_t stream) {
    switch (p.num_lods) {
        case 1: return svr_launch_march_nl1(p, kind, stream);
        case 2: return svr_launch_march_nl2(p, kind, stream);
        case 3: return svr_launch_march_nl3(p, kind, stream);
        case 4: return svr_launch_march_nl4(p, kind, stream);
        case 5: return svr_launch_march_nl5(p, kind, stream);
        case 6: return svr_launch_march_nl6(p, kind, stream);
        case 7: return svr_launch_march_nl7(p, kind, stream);
        case 8: return svr_launch_march_nl8(p, kind, stream);
        default: return hipErrorInvalidValue;
    }
}
