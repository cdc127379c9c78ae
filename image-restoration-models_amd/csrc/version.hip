// ABI version of libirm_hip.so (see include/irm_hip.h).
#include "irm_common.h"
extern "C" int irm_version(void) { return 1; }
