// macros.h -- small portability macros kept for source compatibility with code
// written against simpleMath (reference include/macros.h).  Nothing in the
// MI355X path depends on them: loops live in libsmhip's kernels, not here.
#pragma once

#if defined(__GNUC__) || defined(__clang__)
#  define likely(x)   __builtin_expect(!!(x), 1)
#  define unlikely(x) __builtin_expect(!!(x), 0)
#  define ALWAYS_INLINE inline __attribute__((always_inline))
#else
#  define likely(x)   (x)
#  define unlikely(x) (x)
#  define ALWAYS_INLINE inline
#endif

// The reference's OpenMP chunk (macros.h:16); kept as a name only.
#ifndef CHUNK_SIZE
#  define CHUNK_SIZE 1024
#endif
