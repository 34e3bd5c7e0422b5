// SMUtils.h -- shape layer of the drop-in host side: NumPy-style broadcasting.
//
// Same names and results as the reference's include/SMUtils.h:5-99
// (sm::BroadCastResult, sm::broadcast, sm::calculateTotalSize, sm::processIndex);
// the arithmetic is libsmhip's smhip_broadcast, so every binding (C++, ctypes)
// resolves shapes identically.  The one error the reference can raise --
// std::runtime_error("Cannot broadcast shapes: incompatible dimensions"),
// SMUtils.h:76-78 -- is raised here with the same type and text.
#pragma once

#include <concepts>
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "Slice.h"
#include "macros.h"
#include "smhip.h"

namespace sm {

struct BroadCastResult {
    std::vector<std::size_t> resultShape;
    std::vector<std::size_t> newShape1;
    std::vector<std::size_t> newStrides1;
    std::vector<std::size_t> newShape2;
    std::vector<std::size_t> newStrides2;
    std::size_t totalSize;
};

template <std::integral I>
constexpr Slice processIndex(I index) noexcept {
    return Slice::index(static_cast<std::size_t>(index));
}
constexpr Slice processIndex(Slice s) noexcept { return s; }

inline std::size_t calculateTotalSize(const std::vector<std::size_t> &shape) {
    std::size_t n = 1;
    for (std::size_t d : shape) n *= d;
    return n;
}

inline BroadCastResult broadcast(const std::vector<std::size_t> &shape1, const std::vector<std::size_t> &strides1,
                                 const std::vector<std::size_t> &shape2, const std::vector<std::size_t> &strides2) {
    const std::size_t rank = shape1.size() > shape2.size() ? shape1.size() : shape2.size();
    std::vector<std::int64_t> sh1(shape1.begin(), shape1.end()), st1(strides1.begin(), strides1.end());
    std::vector<std::int64_t> sh2(shape2.begin(), shape2.end()), st2(strides2.begin(), strides2.end());
    std::vector<std::int64_t> out(rank), n1(rank), n2(rank);
    std::int64_t total = 0;
    const int rc = smhip_broadcast(static_cast<int>(sh1.size()), sh1.data(), st1.data(), static_cast<int>(sh2.size()),
                                   sh2.data(), st2.data(), out.data(), n1.data(), n2.data(), &total);
    if (rc < 0) throw std::runtime_error(rc == SMHIP_ERR_BROADCAST ? "Cannot broadcast shapes: incompatible dimensions"
                                                                   : smhip_last_error());
    BroadCastResult r;
    r.resultShape.assign(out.begin(), out.end());
    r.newStrides1.assign(n1.begin(), n1.end());
    r.newStrides2.assign(n2.begin(), n2.end());
    // operand shapes right-aligned and padded with 1s, as the reference reports them
    r.newShape1.assign(rank, 1);
    r.newShape2.assign(rank, 1);
    for (std::size_t i = 0; i < shape1.size(); ++i) r.newShape1[rank - shape1.size() + i] = shape1[i];
    for (std::size_t i = 0; i < shape2.size(); ++i) r.newShape2[rank - shape2.size() + i] = shape2[i];
    r.totalSize = static_cast<std::size_t>(total);
    return r;
}

}  // namespace sm
