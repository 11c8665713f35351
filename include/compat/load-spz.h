// Compatibility header: lets sources written against lanxinger/spz's `#include "load-spz.h"`
// (reference: src/cc/load-spz.h) compile unchanged against the MI355X drop-in layer.
#pragma once
#include "../spz_amd_host.hpp"
