// Version / error-string entry points of the C ABI.
#include "pn2_common.h"

// Share of the chip the persistent launches that FOLLOW may take (see pn2_set_cu_share in pn2.h): one definition for the library,
// read through pn2_num_cus() of pn2_common.h.
int pn2_cu_share_num = 1, pn2_cu_share_den = 1;

extern "C" {

int pn2_version(void) { return PN2_ABI_VERSION; }

int pn2_set_cu_share(int num, int den) {
    if (num < 1 || den < 1 || num > den) return PN2_EINVAL;
    pn2_cu_share_num = num;
    pn2_cu_share_den = den;
    return PN2_OK;
}

const char *pn2_error_string(int code) {
    switch (code) {
        case PN2_OK: return "ok";
        case PN2_EINVAL: return "invalid argument";
        case PN2_ELAUNCH: return "HIP launch failed";
        case PN2_EUNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}

}  // extern "C"
