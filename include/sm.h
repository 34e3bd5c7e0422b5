// sm.h -- the one header user code includes (drop-in for the reference's include/sm.h): sm::SMArray<T>, the free
// functions, and through them the C ABI of libsmhip.so (include/smhip.h).  Link with -lsmhip.
#pragma once

#include "UserFunctions.h"  // brings SMArray.h, the Op policies and the loop entry points with it
#include "Sharded.h"        // sm::set_devices, sm::Sharded<T>: the same operators over the GPUs of one node
