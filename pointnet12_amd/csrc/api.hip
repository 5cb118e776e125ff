// Version / error-string entry points of the C ABI.
#include "pn2_common.h"

#include <cxxabi.h>
#include <stdlib.h>
#include <string.h>

int pn2_option_table[PN2_OPT_COUNT] = {
#define PN2_X(name, dflt) dflt,
    PN2_OPTION_LIST(PN2_X)
#undef PN2_X
};

thread_local const void *pn2_last_kernel_fn = nullptr;

namespace {
const char *const kOptionNames[PN2_OPT_COUNT] = {
#define PN2_X(name, dflt) "PN2_" #name,
    PN2_OPTION_LIST(PN2_X)
#undef PN2_X
};
int option_index(const char *name) {
    if (name == nullptr) return -1;
    for (int i = 0; i < PN2_OPT_COUNT; ++i)
        if (strcmp(name, kOptionNames[i]) == 0 || strcmp(name, kOptionNames[i] + 4) == 0) return i;     // with or without "PN2_"
    return -1;
}
}  // namespace

extern "C" {

int pn2_set_option(const char *name, int value) {
    const int i = option_index(name);
    if (i < 0) return PN2_EINVAL;
    __atomic_store_n(&pn2_option_table[i], value, __ATOMIC_RELAXED);
    return PN2_OK;
}

int pn2_get_option(const char *name, int *value) {
    const int i = option_index(name);
    if (i < 0 || value == nullptr) return PN2_EINVAL;
    *value = pn2_opt(i);
    return PN2_OK;
}

const char *pn2_option_name(int index) { return index >= 0 && index < PN2_OPT_COUNT ? kOptionNames[index] : nullptr; }

int pn2_version(void) { return PN2_ABI_VERSION; }

const char *pn2_last_kernel(void) {
    // the runtime knows the device-side (mangled) name behind a kernel's host stub; demangled it is what rocprofv3 prints
    static thread_local char buf[1024];
    const void *fn = pn2_last_kernel_fn;
    if (fn == nullptr) return nullptr;
    const char *mangled = hipKernelNameRefByPtr(fn, nullptr);
    if (mangled == nullptr) return nullptr;
    int status = 0;
    size_t len = 0;
    char *dem = abi::__cxa_demangle(mangled, nullptr, &len, &status);
    const char *src = (status == 0 && dem) ? dem : mangled;
    size_t n = strlen(src);
    if (n >= sizeof(buf)) n = sizeof(buf) - 1;
    memcpy(buf, src, n);
    buf[n] = 0;
    free(dem);
    return buf;
}
void pn2_clear_last_kernel(void) { pn2_last_kernel_fn = nullptr; }

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
