// Sharded.h -- sm::SMArray<T> across the GPUs of one node.
//
// The reference spreads element_wise_op / array_scalar_op over the cores of one socket with `#pragma omp parallel for`
// (include/math/calculate.h:47, :152).  Here the same independence is used one level up: sm::set_devices(n) makes
// GPUs 0..n-1 a group, and a sm::Sharded<T> is an array whose OUTERMOST dimension is cut into n near-equal blocks, block
// g resident on GPU g as an ordinary SMArray<T>.  + - * / (array or scalar), sm::pow and user Ops run block by block
// with no traffic between GPUs; an operand that is broadcast along the outermost dimension (a bias row, a per-column
// scale) is held whole on every GPU (Sharded<T>::replicate).  Only the whole-array reductions -- sum(), %, and the
// fused apply_sum<Op>() of BASELINE config 5 -- exchange anything: one RCCL all-reduce of an 8-byte value over xGMI.
//
//     sm::set_devices(8);
//     auto a = sm::Sharded<float>::ones(1u << 31), b = sm::Sharded<float>::ones(1u << 31);
//     auto c = a + b;                       // 8 independent 2^28-element adds
//     double s = sm::sum(c);                // 8 partial sums + ONE ncclAllReduce(1 x fp64)
//
// Everything goes through libsmhip's sharded C entry points (smhip.h: smhip_sharded_*), which take one pointer per
// device; one host thread drives all devices, each on its own stream.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <utility>
#include <vector>

#include "SMArray.h"

namespace sm {

// GPUs 0..n-1 become the device group (n = 0 dissolves it).  Loads RCCL and builds the communicators.
inline void set_devices(int n) { hip::check(smhip_set_devices(n)); }
inline int devices() {
    int n = 0;
    hip::check(smhip_get_devices(&n));
    return n;
}
// Waits for every GPU of the group.
inline void synchronize_devices() { hip::check(smhip_sharded_synchronize()); }

template <ArithmeticOrComplex T>
class Sharded {
    static_assert(hip::dtype_of<T>::id >= 0, "sm::Sharded: element type has no gfx950 kernels");

public:
    // ---- creation -------------------------------------------------------------------------
    template <typename... Dims>
    static Sharded empty(Dims... dims) { return make({static_cast<std::size_t>(dims)...}, nullptr); }
    template <typename... Dims>
    static Sharded ones(Dims... dims) { const T v{1}; return make({static_cast<std::size_t>(dims)...}, &v); }
    template <typename... Dims>
    static Sharded zeros(Dims... dims) { const T v{0}; return make({static_cast<std::size_t>(dims)...}, &v); }
    static Sharded full(std::vector<std::size_t> shape, T value) { return make(std::move(shape), &value); }

    // Block g of `whole`'s outermost dimension goes to GPU g.  A device-resident array travels device to device
    // (smhip_copy_peer: hipMemcpyPeerAsync over xGMI, all destinations in flight at once, stream-ordered on both sides); an
    // array that exists only in host memory is uploaded block by block straight from its host mirror.
    static Sharded scatter(const SMArray<T> &whole) {
        Sharded r = layout(whole.shape(), false);
        const std::size_t inner = r.inner_size();
        r.fill_parts(whole, [&](int g) { return std::pair<std::size_t, std::size_t>(r.starts_[g] * inner, r.rows_[g] * inner); });
        return r;
    }
    // A full copy of `whole` on every GPU: how an operand that is broadcast along the outermost dimension takes part.
    static Sharded replicate(const SMArray<T> &whole) {
        Sharded r = layout(whole.shape(), true);
        r.fill_parts(whole, [&](int) { return std::pair<std::size_t, std::size_t>(0, whole.totalSize); });
        return r;
    }
    // The whole array again, on the calling thread's current GPU: one peer copy per block, device to device.
    SMArray<T> gather() const {
        if (replicated_) {
            int here = 0;
            hip::check(smhip_get_device(&here));
            for (int g = 0; g < group(); ++g)
                if (parts_[g].device() == here) return parts_[g].contiguous();
            SMArray<T> out = SMArray<T>::device_empty(std::vector<std::size_t>(shape_));
            hip::check(smhip_copy_peer(out.device_data_mut(), here, parts_[0].device_data(), parts_[0].device(), out.totalSize * sizeof(T)));
            return out;
        }
        SMArray<T> out = SMArray<T>::device_empty(std::vector<std::size_t>(shape_));
        const std::size_t inner = inner_size();
        T *dst = out.device_data_mut();
        for (int g = 0; g < group(); ++g) {
            const std::size_t count = rows_[g] * inner;
            if (count == 0) continue;
            hip::check(smhip_copy_peer(dst + starts_[g] * inner, out.device(), parts_[g].device_data(), parts_[g].device(), count * sizeof(T)));
        }
        return out;
    }

    // ---- elementwise: no collective ---------------------------------------------------------
    Sharded operator+(const Sharded &rhs) const { return apply<AddOp<T>>(rhs); }
    Sharded operator-(const Sharded &rhs) const { return apply<SubtractOp<T>>(rhs); }
    Sharded operator*(const Sharded &rhs) const { return apply<MultiplyOp<T>>(rhs); }
    Sharded operator/(const Sharded &rhs) const { return apply<DivideOp<T>>(rhs); }
    Sharded operator+(T v) const { return apply_scalar<AddOp<T>>(v); }
    Sharded operator-(T v) const { return apply_scalar<SubtractOp<T>>(v); }
    Sharded operator*(T v) const { return apply_scalar<MultiplyOp<T>>(v); }
    Sharded operator/(T v) const { return apply_scalar<DivideOp<T>>(v); }

    // `*this Op rhs` with NumPy broadcasting, evaluated block by block (smhip_sharded_elementwise).
    template <typename Op>
    Sharded apply(const Sharded &rhs) const {
        static_assert(hip::on_device_v<T, Op>, "sm::Sharded: this Op has no device form (see SM_DEVICE_OP / SM_DEFINE_OP)");
        Plan p = plan(rhs);
        Sharded out = layout(p.shape, false);
        out.allocate();
        std::vector<void *> po(group());
        for (int g = 0; g < group(); ++g) po[g] = out.parts_[g].device_data_mut();
        hip::check(smhip_sharded_elementwise(hip::device_op<Op>::id(), hip::dtype_of<T>::id, p.a.data(), p.sa.data(), p.b.data(),
                                             p.sb.data(), p.shape64.data(), static_cast<int>(p.shape64.size()), po.data()));
        return out;
    }
    template <typename Op>
    Sharded apply_scalar(T value) const {
        static_assert(hip::on_device_v<T, Op>, "sm::Sharded: this Op has no device form (see SM_DEVICE_OP / SM_DEFINE_OP)");
        Sharded out = layout(shape_, replicated_);
        out.allocate();
        std::vector<const void *> pa(group());
        std::vector<void *> po(group());
        std::vector<std::size_t> n(group());
        for (int g = 0; g < group(); ++g) {
            pa[g] = parts_[g].device_data();
            po[g] = out.parts_[g].device_data_mut();
            n[g] = parts_[g].totalSize;
        }
        hip::check(smhip_sharded_array_scalar(hip::device_op<Op>::id(), hip::dtype_of<T>::id, pa.data(), &value, n.data(), po.data()));
        return out;
    }

    // ---- whole-array reductions: per-GPU partial + ONE all-reduce of 8 bytes ---------------------
    double sum() const {
        if (replicated_) return parts_[0].sum();
        Tables t = tables();
        double s = 0;
        hip::check(smhip_sharded_sum(hip::dtype_of<T>::id, t.p.data(), t.n.data(), &s));
        return s;
    }
    // Dot product over all elements (the reference's operator%, SMArray.h:213-215).
    T operator%(const Sharded &rhs) const {
        if (shape_ != rhs.shape_ || replicated_ != rhs.replicated_) throw std::runtime_error("dot product: sharded operands must have the same shape and layout");
        if (replicated_) return parts_[0] % rhs.parts_[0];
        Tables ta = tables(), tb = rhs.tables();
        T r{};
        hip::check(smhip_sharded_dot(hip::dtype_of<T>::id, ta.p.data(), tb.p.data(), ta.n.data(), &r));
        return r;
    }
    // BASELINE config 5: `*this Op rhs` and the sum of all its elements in ONE pass over HBM per GPU (the sum adds no
    // traffic) -- same-shape operands only.
    template <typename Op>
    Sharded apply_sum(const Sharded &rhs, double *total) const {
        static_assert(hip::on_device_v<T, Op>, "sm::Sharded: this Op has no device form");
        if (shape_ != rhs.shape_ || replicated_ || rhs.replicated_) throw std::runtime_error("apply_sum: operands must be sharded arrays of the same shape");
        if (hip::device_op<Op>::id() > SMHIP_OP_LEFT) {  // a user Op has no fused reduction kernel: two passes
            Sharded out = apply<Op>(rhs);
            *total = out.sum();
            return out;
        }
        Sharded out = layout(shape_, false);
        out.allocate();
        Tables ta = tables(), tb = rhs.tables();
        std::vector<void *> po(group());
        for (int g = 0; g < group(); ++g) po[g] = out.parts_[g].device_data_mut();
        hip::check(smhip_sharded_contiguous_sum(hip::device_op<Op>::id(), hip::dtype_of<T>::id, ta.p.data(), tb.p.data(), po.data(),
                                                ta.n.data(), total));
        return out;
    }

    // ---- inspection ---------------------------------------------------------------------------
    const std::vector<std::size_t> &shape() const { return shape_; }  // the GLOBAL shape
    bool replicated() const { return replicated_; }
    int group() const { return static_cast<int>(rows_.size()); }
    SMArray<T> &part(int g) { return parts_[g]; }  // GPU g's block (rows start(g) .. start(g) + rows(g))
    const SMArray<T> &part(int g) const { return parts_[g]; }
    std::size_t start(int g) const { return starts_[g]; }
    std::size_t rows(int g) const { return rows_[g]; }

    Sharded(Sharded &&) noexcept = default;
    Sharded &operator=(Sharded &&) noexcept = default;
    Sharded(const Sharded &) = delete;
    Sharded &operator=(const Sharded &) = delete;

private:
    std::vector<std::size_t> shape_;
    std::vector<std::size_t> starts_, rows_;  // along dim 0, per GPU
    std::vector<SMArray<T>> parts_;
    bool replicated_ = false;

    Sharded() = default;

    std::size_t inner_size() const {
        std::size_t n = 1;
        for (std::size_t i = 1; i < shape_.size(); ++i) n *= shape_[i];
        return n;
    }
    std::vector<std::size_t> part_shape(int g) const {
        std::vector<std::size_t> s = shape_;
        if (!replicated_ && !s.empty()) s[0] = rows_[g];
        return s;
    }

    // Shape bookkeeping only: which rows each GPU of the current group owns.
    static Sharded layout(const std::vector<std::size_t> &shape, bool replicated) {
        const int n = devices();
        if (n < 1) throw std::runtime_error("sm::Sharded: no device group -- call sm::set_devices(n) first");
        if (shape.empty()) throw std::runtime_error("sm::Sharded: a 0-d array cannot be sharded");
        Sharded r;
        r.shape_ = shape;
        r.replicated_ = replicated;
        r.starts_.resize(n);
        r.rows_.resize(n);
        for (int g = 0; g < n; ++g) {
            std::int64_t st = 0, ct = 0;
            hip::check(smhip_split_range(static_cast<std::int64_t>(shape[0]), n, g, &st, &ct));
            r.starts_[g] = replicated ? 0 : static_cast<std::size_t>(st);
            r.rows_[g] = replicated ? shape[0] : static_cast<std::size_t>(ct);
        }
        r.parts_.reserve(n);
        return r;
    }
    void allocate() {
        for (int g = 0; g < group(); ++g) {
            hip::DeviceGuard on(g);
            parts_.push_back(SMArray<T>::device_empty(part_shape(g)));
        }
    }
    // parts_[g] = elements [first, first + count) of `whole` (dense, row-major), as range(g) says, resident on GPU g
    template <typename Range>
    void fill_parts(const SMArray<T> &whole, Range &&range) {
        allocate();
        const auto &st = whole.data.storage();
        if (st && st->dev_valid) {  // device-resident: peer copies, no host staging
            std::unique_ptr<SMArray<T>> holder;
            const SMArray<T> *src = &whole;
            if (!whole.is_dense()) {
                holder.reset(new SMArray<T>(whole.contiguous()));
                src = holder.get();
            }
            const T *from = src->device_data();
            for (int g = 0; g < group(); ++g) {
                const auto [first, count] = range(g);
                if (count == 0) continue;
                hip::check(smhip_copy_peer(parts_[g].device_data_mut(), g, from + first, src->device(), count * sizeof(T)));
            }
            return;
        }
        // host-born: each GPU's block goes up from the host mirror directly
        std::unique_ptr<SMArray<T>> holder;
        const SMArray<T> *src = &whole;
        if (!whole.is_dense()) {  // a strided host view: made dense on the host side first (set-up, not hot path)
            T *flat = new T[whole.totalSize ? whole.totalSize : 1];
            whole.copy_dense_out(flat);
            holder.reset(new SMArray<T>(flat, std::vector<std::size_t>(whole.shape())));
            src = holder.get();
        }
        const T *from = src->cdata();
        for (int g = 0; g < group(); ++g) {
            const auto [first, count] = range(g);
            if (count == 0) continue;
            hip::DeviceGuard on(g);
            hip::check(smhip_upload(parts_[g].device_data_mut(), from + first, count * sizeof(T)));
        }
    }
    static Sharded make(std::vector<std::size_t> shape, const T *fill) {
        Sharded r = layout(shape, false);
        for (int g = 0; g < r.group(); ++g) {
            hip::DeviceGuard on(g);
            r.parts_.push_back(fill ? SMArray<T>::device_full(r.part_shape(g), *fill) : SMArray<T>::device_empty(r.part_shape(g)));
        }
        return r;
    }

    struct Tables {
        std::vector<const void *> p;
        std::vector<std::size_t> n;
    };
    Tables tables() const {
        Tables t;
        for (int g = 0; g < group(); ++g) {
            t.p.push_back(parts_[g].device_data());
            t.n.push_back(parts_[g].totalSize);
        }
        return t;
    }

    // The broadcast problem `*this op rhs` in the form smhip_sharded_elementwise takes: global result shape, the two
    // operands' broadcast strides, and per GPU the pointer to the first element of its block of each operand.
    struct Plan {
        std::vector<std::size_t> shape;
        std::vector<std::int64_t> shape64, sa, sb;
        std::vector<const void *> a, b;
    };
    Plan plan(const Sharded &rhs) const {
        if (group() != rhs.group()) throw std::runtime_error("sm::Sharded: operands belong to device groups of different sizes");
        const auto br = sm::broadcast(shape_, detail::dense_strides(shape_), rhs.shape_, detail::dense_strides(rhs.shape_));
        if (br.resultShape.size() > MAX_NDIM) throw std::runtime_error("rank exceeds MAX_NDIM");
        Plan p;
        p.shape = br.resultShape;
        p.shape64 = hip::to_i64(br.resultShape);
        p.sa = hip::to_i64(br.newStrides1);
        p.sb = hip::to_i64(br.newStrides2);
        auto side = [&](const Sharded &x, const std::vector<std::int64_t> &strides, std::vector<const void *> &ptrs) {
            const bool along0 = x.shape_.size() == br.resultShape.size() && x.shape_[0] == br.resultShape[0];
            if (!x.replicated_ && !along0 && br.resultShape[0] > 1)
                throw std::runtime_error("sm::Sharded: an operand that is broadcast along the outermost dimension must be held whole on "
                                         "every GPU -- build it with Sharded<T>::replicate(array)");
            for (int g = 0; g < group(); ++g) {
                const T *base = x.parts_[g].device_data();
                if (x.replicated_ && strides[0] != 0) {  // a replicated operand that does vary along dim 0: GPU g reads its rows of the copy
                    std::int64_t st = 0, ct = 0;
                    hip::check(smhip_split_range(p.shape64[0], group(), g, &st, &ct));
                    base += st * strides[0];
                }
                ptrs.push_back(base);
            }
        };
        side(*this, p.sa, p.a);
        side(rhs, p.sb, p.b);
        return p;
    }
};

template <typename T>
double sum(const Sharded<T> &arr) {
    return arr.sum();
}
template <typename T>
Sharded<T> pow(const Sharded<T> &arr, T val) {
    return arr.template apply_scalar<PowOp<T>>(val);
}

}  // namespace sm
