/* oracle/shim — see cuda_runtime.h (test infrastructure only). */
#pragma once
#include "cuda_runtime.h"
