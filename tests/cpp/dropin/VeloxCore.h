// Drop-in header: what a maintainer's VeloxCore.h becomes when the codec moves to the MI355X library
// (INTEGRATION.md §1).  VeloxFormat.h (and through it VeloxArch.h) still come from the reference; only
// the two codec classes change hands.  Defining the reference's include guard makes its own
// VeloxCore.h a no-op wherever this one was seen first.
#ifndef VELOX_CORE_H
#define VELOX_CORE_H
#include "VeloxFormat.h"
#include "VeloxCodecAMD.h"
#endif
