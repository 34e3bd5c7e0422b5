// sm.h -- umbrella header: `#include <sm.h>` and link with -lsmhip.
#pragma once
#include "SMArray.h"
#include "UserFunctions.h"
