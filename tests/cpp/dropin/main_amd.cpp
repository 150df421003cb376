// The reference's unmodified CLI (main.cpp) on libvelox_amd.so: this translation unit only puts the
// drop-in VeloxCore.h in front of it.  Built by oracle/Makefile into oracle/_ref/ (it compiles
// reference sources, so the binary is never committed).
#include "VeloxCore.h"
#include "main.cpp"
