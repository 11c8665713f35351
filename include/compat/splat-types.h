// Compatibility header for `#include "splat-types.h"` (reference: src/cc/splat-types.h).
#pragma once
#include "../spz_amd_host.hpp"
