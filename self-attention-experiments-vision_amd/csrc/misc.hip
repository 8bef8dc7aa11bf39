#include "common.h"
#include "savit.h"
extern "C" int savit_abi_version(void) { return SAVIT_ABI_VERSION; }
