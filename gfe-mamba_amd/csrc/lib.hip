// Library identity + tiny device utilities shared by the C-ABI (include/gfe_hip.h).
#include "common.h"

extern "C" {
int gfe_abi_version(void) { return GFE_ABI_VERSION; }
const char* gfe_build_arch(void) { return "gfx950"; }
}
