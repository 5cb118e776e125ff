// Version / error-string entry points of the C ABI.
#include "pn2_common.h"

extern "C" {

int pn2_version(void) { return PN2_ABI_VERSION; }

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
