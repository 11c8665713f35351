// Source compatibility: a translation unit that includes the reference's "splat-c-types.h" gets the
// same C bridge types from spz_amd (see spz_amd_c_types.h).
#pragma once
#include "../spz_amd_c_types.h"
