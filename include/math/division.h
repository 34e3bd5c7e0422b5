// division.h -- kept so that `#include "math/division.h"` from code written against
// simpleMath keeps working; the Op policies live together in math/ops.h.
#pragma once
#include "ops.h"
