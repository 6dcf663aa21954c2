// version.hip -- library identification
#include "common.h"
extern "C" const char *ir2rgb_version(void) { return "ir2rgb_hip 0.1.0 gfx950"; }
