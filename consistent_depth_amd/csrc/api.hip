// ABI bookkeeping entry points.
#include "cd_common.h"

#define CD_STR2(x) #x
#define CD_STR(x) CD_STR2(x)

extern "C" int cd_abi_version(void) { return CD_ABI_VERSION; }

extern "C" const char* cd_build_info(void) {
    return "consistent_depth_amd gfx950 hip " CD_STR(HIP_VERSION_MAJOR) "." CD_STR(HIP_VERSION_MINOR) " built " __DATE__;
}
