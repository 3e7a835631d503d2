// Library identity (include/atvsnet_hip.h).
#include "common.h"

extern "C" int atvs_abi_version(void) { return ATVS_ABI_VERSION; }
extern "C" const char* atvs_target_arch(void) { return "gfx950"; }
