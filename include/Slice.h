// Slice.h -- index / range selector for SMArray::operator() (drop-in for the
// reference's include/Slice.h: same global `Slice` type, same SLICE* macros).
//
//   arr(0, SLICE_ALL)        row 0 of a 2-D array, as a view
//   arr(SLICE(2, 5), 1)      rows 2..4 of column 1
//
// `end == size_t(-1)` means "to the end of the axis".  `step` is an enum with the single name SINGLE_STEP (= 0, a unit
// step), as in the reference; a larger value stored in it is honoured as a stride multiplier, as the reference's view code does.
#pragma once

#include <cstddef>

struct Slice {
    enum SliceStep { SINGLE_STEP = 0 };
    enum SliceType { INDEX = 0, SLICE };

    std::size_t start;
    std::size_t end;
    SliceStep step = SINGLE_STEP;
    SliceType sliceType = SLICE;

    constexpr Slice(std::size_t first, std::size_t last = static_cast<std::size_t>(-1)) : start(first), end(last) {}

    // An integer subscript: selects one position and drops the axis.
    static constexpr Slice index(std::size_t at) {
        Slice s(at);
        s.sliceType = INDEX;
        return s;
    }
    constexpr bool open_ended() const { return end == static_cast<std::size_t>(-1); }
};

#define SLICE(start, end) Slice(start, end)
#define SLICE_START(start) Slice(start)
#define SLICE_END(end) Slice(0, end)
#define SLICE_ALL Slice(0)
