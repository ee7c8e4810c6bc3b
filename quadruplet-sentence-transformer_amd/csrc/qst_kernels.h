// The kernel-level C-ABI lives in include/qst_kernels.h (one copy, the public one).
#pragma once
#include "../../include/qst_kernels.h"
