// Library identity for the loader-side ABI check.
#include "../../include/mgx.h"
extern "C" int mgx_version(void) { return 1; }
