// SMArray.h -- sm::SMArray<T>, the drop-in array class, MI355X-resident.
//
// Public surface of the reference's class (include/SMArray.h:30-438): public
// `data` and `totalSize`; construction from (nested) initializer lists or from
// an owned `new T[]` buffer plus a shape; move-only; operator()(ints...) for
// element access and operator()(ints | Slice ...) for views; transpose(),
// repeat(); % (dot), + - * / with an array (NumPy broadcasting) or a scalar;
// toString(), shape(), strides().
//
// What is different underneath: an array's elements live in HBM.
//   * Every array (and all views of it) shares one Storage: a pooled device
//     buffer and, only when host code asks for it, a host mirror.  Two validity
//     bits track which side is current.  `data`, operator()(i...) and toString()
//     bring the host mirror up to date; arithmetic brings the device side up to
//     date.  Writable host access (`data`, the T& form of operator()) marks the
//     device side stale, so the reference's idiom `arr.data[i] = v; r = arr * 2;`
//     keeps working.  A chain of operators never leaves the GPU.
//   * + - * / % and sm::pow run libsmhip's gfx950 kernels through the C ABI
//     (smhip.h) at the points where the reference calls element_wise_op /
//     array_scalar_op / dot_product (SMArray.h:213-305).  Results are born on the
//     device from a pooled allocator instead of `new T[n]` (SMArray.h:219).
//   * There is no CPU arithmetic path: without libsmhip + a GPU the operators
//     throw std::runtime_error.
//
// Deliberate deviations from reference quirks:
//   1-D operands walk their strides (the reference reads them as dense);
//   views know their real ndim/totalSize, so scalar ops, % and toString work on
//   views; repeat() has NumPy semantics; rank > MAX_NDIM throws.
#pragma once

#include <cassert>
#include <complex>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "SMUtils.h"
#include "Slice.h"
#include "math/calculate.h"
#include "math/ops.h"
#include "math/product.h"

namespace sm {

template <typename T>
concept ArithmeticOrComplex =
    std::is_arithmetic_v<T> ||
    (requires { typename T::value_type; } && std::is_arithmetic_v<typename T::value_type> &&
     std::same_as<T, std::complex<typename T::value_type>>);

template <ArithmeticOrComplex T>
class SMArray;

namespace detail {

// ---- deferred operator chains ------------------------------------------------------------------------------------------
// The reference evaluates `(A * row + B) * 0.5f` as three operator calls with a temporary each (SMArray.h:217-305); here
// a temporary that feeds the next operator OF THE SAME FULL-EXPRESSION is never written to HBM: + - * / do not launch
// at once, they record a one-stage *chain* in the result's Storage, and an operator whose operand is that unevaluated
// temporary (an rvalue) extends the chain instead of reading it.  The chain is launched -- as ONE kernel where its
// operands allow, smhip_chain -- at the latest when the full-expression ends: every operator argument binds to a small
// wrapper (Operand / ScalarArg below) whose destructor, run when the statement's temporaries die, evaluates whatever is
// still pending on the thread.  So a NAMED value is always computed by the end of the statement that names it; nothing
// stays lazy across statements.  Before that point a chain is evaluated as soon as anything looks at its result (`data`,
// operator(), toString(), device_data(), another library call), and before any array is WRITTEN (host or device), so
// an operand cannot change under a recorded chain.  At most one chain is pending per thread.
struct PendingBase {
    virtual void run() = 0;  // evaluate into the Storage that owns this object (and destroy it)
    virtual ~PendingBase() = default;
};
inline thread_local PendingBase *tls_pending = nullptr;
inline void flush_pending() {
    if (PendingBase *p = tls_pending) {
        tls_pending = nullptr;
        p->run();
    }
}
// What the fusion did on this thread (tests and benchmarks read it through sm::fusion_stats()).
struct FusionStats {
    unsigned long long chains = 0;        // smhip_chain calls: expressions of two or more operators evaluated as one call
    unsigned long long fused_stages = 0;  // operators inside them
    unsigned long long single_ops = 0;    // deferred operators that ended up alone and ran as the plain operator
    unsigned long long direct_assignments = 0;  // `x = <expression>` evaluated straight into x (no temporary, no copy)
    unsigned long long summed_chains = 0;       // `(<expression>).sum()` taken in the chain's own pass, the value never written
};
inline thread_local FusionStats tls_fusion_stats;
// The end of a full-expression, seen from the destructor of one of its temporaries.
inline void end_of_expression(int exceptions_at_birth) {
    if (!tls_pending) return;
    if (std::uncaught_exceptions() > exceptions_at_birth) {  // unwinding: the result is being destroyed as well, nothing to compute
        tls_pending = nullptr;
        return;
    }
    flush_pending();
}

// One allocation, shared by an owning array and every view of it.
template <typename T>
struct Storage : std::enable_shared_from_this<Storage<T>> {
    T *host = nullptr;
    void *dev = nullptr;
    std::size_t count = 0;
    bool host_valid = false;  // host mirror holds the current values
    bool dev_valid = false;   // device buffer holds the current values

    int device = 0;           // the GPU the device buffer lives on (the creating thread's current device)
    unsigned inline_uses = 0; // times the host-only elements rode in a kernel's argument block instead of being uploaded
    unsigned fetches = 0;     // single elements read back one at a time (fetch()); after a few the whole array is mirrored
    std::unique_ptr<PendingBase> pending;  // the operator chain that will produce these elements, while it has not run

    // The values are about to be looked at: run the recorded chain first.
    void ensure() {
        if (pending) {
            if (tls_pending == pending.get()) tls_pending = nullptr;
            pending->run();  // resets `pending`
        }
    }
    // An array is about to be written: a chain recorded on this thread may still want its old values.
    void before_write() {
        ensure();
        if (tls_pending) flush_pending();
    }

    explicit Storage(std::size_t n) : count(n) { smhip_get_device(&device); }
    Storage(T *adopted, std::size_t n) : host(adopted), count(n), host_valid(true) { smhip_get_device(&device); }  // takes ownership of new T[]
    Storage(const Storage &) = delete;
    Storage &operator=(const Storage &) = delete;
    ~Storage() {
        if (pending && tls_pending == pending.get()) tls_pending = nullptr;  // a result nobody looked at: never computed
        if (dev) smhip_free(dev);
        delete[] host;
    }

    // host side current, device side left valid (read-only intent)
    const T *host_ro() {
        ensure();
        if (!host) host = new T[count ? count : 1];
        if (!host_valid) {
            if (dev_valid) {
                hip::DeviceGuard on(device);
                hip::check(smhip_download(host, dev, count * sizeof(T)));
            }
            host_valid = true;  // nothing valid anywhere: uninitialised, like sm::empty
        }
        return host;
    }
    // One element for reading, without mirroring the whole array: element (i) of a device-born result costs one
    // sizeof(T) copy, not a full download.
    // A loop over every element -- the reference's test idiom, `EXPECT_EQ(r(i, j, k, c), ...)` over 224 * 224 * 3 reads --
    // would be that many blocking 4-byte copies: after kFetchesBeforeMirror of them the array is mirrored once (the device
    // copy stays valid) and further reads are host reads.
    static constexpr unsigned kFetchesBeforeMirror = 8;
    T fetch(std::size_t index) {
        ensure();
        if (host_valid || !dev_valid || ++fetches > kFetchesBeforeMirror) return host_ro()[index];
        T v;
        hip::DeviceGuard on(device);
        hip::check(smhip_download(&v, static_cast<const T *>(dev) + index, sizeof(T)));
        return v;
    }
    // host side current and possibly about to be written: device side goes stale
    T *host_rw() {
        before_write();
        host_ro();
        dev_valid = false;
        return host;
    }
    // device side current, host mirror stays valid
    T *dev_ro() {
        ensure();
        if (!dev || !dev_valid) {
            hip::DeviceGuard on(device);
            if (!dev) hip::check(smhip_alloc(&dev, (count ? count : 1) * sizeof(T)));
            if (!dev_valid) {
                if (host_valid) hip::check(smhip_upload(dev, host, count * sizeof(T)));
                dev_valid = true;
            }
        }
        return static_cast<T *>(dev);
    }
    // device side current and about to be partly overwritten: host mirror goes stale
    T *dev_rw() {
        before_write();
        T *p = dev_ro();
        host_valid = false;
        return p;
    }
    // device side about to be overwritten entirely
    T *dev_wo() {
        before_write();
        if (!dev) {
            hip::DeviceGuard on(device);
            hip::check(smhip_alloc(&dev, (count ? count : 1) * sizeof(T)));
        }
        dev_valid = true;
        host_valid = false;
        return static_cast<T *>(dev);
    }
};

// What `arr.data` is.  Converts to T* / indexes like T*; touching it through a
// non-const route syncs the host mirror and marks the device copy stale.
template <typename T>
class HostPtr {
public:
    HostPtr() = default;
    HostPtr(std::shared_ptr<Storage<T>> st, std::size_t offset) : st_(std::move(st)), offset_(offset) {}

    operator T *() const { return st_ ? st_->host_rw() + offset_ : nullptr; }
    T &operator[](std::size_t i) const { return (st_->host_rw() + offset_)[i]; }
    T &operator*() const { return *(st_->host_rw() + offset_); }
    T *operator+(std::ptrdiff_t d) const { return static_cast<T *>(*this) + d; }
    explicit operator bool() const { return static_cast<bool>(st_); }
    bool operator==(std::nullptr_t) const { return !st_; }
    bool operator!=(std::nullptr_t) const { return static_cast<bool>(st_); }
    HostPtr &operator=(std::nullptr_t) {  // `arr.data = nullptr;` gives the buffer up
        st_.reset();
        offset_ = 0;
        return *this;
    }

    const T *read() const { return st_->host_ro() + offset_; }  // no invalidation
    T fetch(std::size_t i) const { return st_->fetch(offset_ + i); }  // one element, no mirror needed
    const std::shared_ptr<Storage<T>> &storage() const { return st_; }
    std::size_t offset() const { return offset_; }

private:
    std::shared_ptr<Storage<T>> st_;
    std::size_t offset_ = 0;
};

// What the array argument of + - * / binds to (implicitly, from an SMArray lvalue or rvalue).  It remembers whether the
// argument was a temporary -- only a temporary's unevaluated chain may be continued -- and its destructor marks the end
// of the full-expression (see "deferred operator chains" above).
template <typename T>
class Operand {
public:
    Operand(const SMArray<T> &a) : arr_(&a), rvalue_(false), exceptions_(std::uncaught_exceptions()) {}  // NOLINT: implicit on purpose
    Operand(SMArray<T> &&a) : arr_(&a), rvalue_(true), exceptions_(std::uncaught_exceptions()) {}       // NOLINT
    Operand(const Operand &) = delete;
    Operand &operator=(const Operand &) = delete;
    ~Operand() noexcept(false) { end_of_expression(exceptions_); }
    const SMArray<T> &array() const { return *arr_; }
    bool temporary() const { return rvalue_; }
private:
    const SMArray<T> *arr_;
    bool rvalue_;
    int exceptions_;
};
// The scalar argument of + - * / (implicitly from a T): same end-of-expression duty.
template <typename T>
class ScalarArg {
public:
    ScalarArg(T v) : value(v), exceptions_(std::uncaught_exceptions()) {}  // NOLINT: implicit on purpose
    ScalarArg(const ScalarArg &) = delete;
    ScalarArg &operator=(const ScalarArg &) = delete;
    ~ScalarArg() noexcept(false) { end_of_expression(exceptions_); }
    T value;
private:
    int exceptions_;
};

template <typename T>
struct Chain;

inline std::vector<std::size_t> dense_strides(const std::vector<std::size_t> &shape) {
    std::vector<std::size_t> st(shape.size());
    std::size_t acc = 1;
    for (std::size_t i = shape.size(); i-- > 0;) {
        st[i] = acc;
        acc *= shape[i];
    }
    return st;
}

}  // namespace detail

template <ArithmeticOrComplex T>
class SMArray {
public:
    detail::HostPtr<T> data;
    std::size_t totalSize = 0;

    SMArray(const std::initializer_list<T> &list) {
        _shape = {list.size()};
        finish_owning(new T[list.size() ? list.size() : 1]);
        std::copy(list.begin(), list.end(), static_cast<T *>(data));
    }

    SMArray(const std::initializer_list<SMArray> &list) {
        const SMArray &first = *list.begin();
        _shape.reserve(first._shape.size() + 1);
        _shape.push_back(list.size());
        _shape.insert(_shape.end(), first._shape.begin(), first._shape.end());
        finish_owning(new T[calculateTotalSize(_shape)]);
        T *dst = static_cast<T *>(data);
        for (const SMArray &child : list) {
            assert(child._shape == first._shape && "ragged nested initializer list");
            child.copy_dense_to(dst);
            dst += child.totalSize;
        }
    }

    // Takes ownership of `buffer` (allocated with new T[]), exactly like the reference.
    SMArray(T *buffer, std::vector<std::size_t> &&shape) {
        _shape = std::move(shape);
        finish_owning(buffer);
    }

    SMArray(SMArray &&other) noexcept = default;
    SMArray(const SMArray &) = delete;
    SMArray &operator=(const SMArray &) = delete;

    // Element-wise copy into an existing array of the same shape (reference SMArray.h:89-97).
    // The reference runs it as a host loop; here it is a strided device copy (smhip_copy_strided), so
    // `a(SLICE(1, 3), SLICE_ALL) = b * c;` never leaves HBM.  A source that shares storage with the
    // destination (`a = a.transpose()`) is made dense first.
    SMArray &operator=(const SMArray &&other) {
        if (_shape != other._shape) throw std::runtime_error("Shape mismatch in assignment");
        if (hip::transport_of<T>::id >= 0 && _shape.size() + (hip::transport_of<T>::lanes > 1 ? 1 : 0) <= MAX_NDIM) {
            if (totalSize == 0) return *this;
            if constexpr (hip::dtype_of<T>::id >= 0) {
                // `x = (a + b) * c;` -- the right-hand side is this thread's unevaluated temporary: evaluate it INTO x instead of
                // into a temporary that is then copied (12 instead of 20 bytes per element for `x = a + b`).  x must be a dense
                // block, and every operand that shares x's storage must be exactly x (same elements, same order: then each
                // element is read before it is written, by the same lane); anything else takes the copy below.
                if (detail::Chain<T> *c = other.continuable(); c && is_dense() && c->may_write_into(data.storage().get(), data.offset(), _shape)) {
                    detail::Storage<T> &mine = *data.storage();
                    if (mine.device == c->device || !mine.dev) {
                        mine.device = c->device;
                        mine.pending = std::move(other.data.storage()->pending);
                        c->retarget(&mine, _shape, data.offset(), data.offset() != 0 || totalSize != mine.count);
                        ++detail::tls_fusion_stats.direct_assignments;
                        mine.ensure();  // runs the chain now: an assignment is evaluated where it stands
                        return *this;
                    }
                }
            }
            hip::DeviceGuard on(common_device(*this, other));
            std::unique_ptr<SMArray> holder;
            const SMArray *src = &other;
            if (other.data.storage() == data.storage()) {
                holder.reset(new SMArray(other.contiguous()));
                src = holder.get();
            }
            auto sh = hip::to_i64(_shape), ss = hip::to_i64(src->_strides), sd = hip::to_i64(_strides);
            if (sh.empty()) { sh = {1}; ss = {1}; sd = {1}; }  // a 0-d array is one element
            lane_axis(sh, ss, &sd);
            const T *from = src->device_data();
            hip::check(smhip_copy_strided(hip::transport_of<T>::id, from, ss.data(), data.storage()->dev_rw() + data.offset(), sd.data(),
                                          sh.data(), static_cast<int>(sh.size())));
        } else {
            std::vector<T> tmp(other.totalSize ? other.totalSize : 1);
            other.copy_dense_to(tmp.data());
            T *base = static_cast<T *>(data);
            for_each_offset([&](std::size_t linear, std::size_t off) { base[off] = tmp[linear]; });
        }
        return *this;
    }

    // arr(i, j, k) -> value;  arr(i, SLICE_ALL) -> view
    template <typename... Args>
        requires((std::is_integral_v<std::remove_cvref_t<Args>> || std::is_same_v<std::remove_cvref_t<Args>, Slice>) && ...)
    auto operator()(Args &&...args) const {
        if constexpr ((std::is_integral_v<std::remove_cvref_t<Args>> && ...)) {
            return data.fetch(offset_of({static_cast<std::size_t>(args)...}, false));
        } else {
            return make_view({processIndex(std::forward<Args>(args))...});
        }
    }

    template <typename... Args>
        requires((std::is_integral_v<std::remove_cvref_t<Args>> && ...))
    T &operator()(Args &&...args) {
        return data[offset_of({static_cast<std::size_t>(args)...}, true)];
    }

    SMArray transpose() const {
        SMArray v;
        v.data = data;
        v._shape.assign(_shape.rbegin(), _shape.rend());
        v._strides.assign(_strides.rbegin(), _strides.rend());
        v.ndim = ndim;
        v.totalSize = totalSize;
        v.isView = true;
        return v;
    }

    // NumPy semantics: every element repeated `numberOfRepeats` times, flattened.  On the
    // device: a (N, r) view with strides (s, 0) gathered densely (the reference's host loop at
    // SMArray.h:138-159 writes newData[i + j], which is not a repeat -- SURVEY 8a quirk 7).
    SMArray repeat(int numberOfRepeats) const {
        assert(numberOfRepeats >= 1);
        const std::size_t r = static_cast<std::size_t>(numberOfRepeats);
        if constexpr (hip::transport_of<T>::id >= 0) {
            if (!is_dense()) return contiguous().repeat(numberOfRepeats);
            return gather({totalSize, r}, {1, 0}, {totalSize * r});  // a dense array seen as (N, r) with strides (1, 0)
        } else {  // element types without kernels (std::complex): plain host data movement
            std::vector<T> flat(totalSize ? totalSize : 1);
            copy_dense_to(flat.data());
            T *out = new T[totalSize * r ? totalSize * r : 1];
            for (std::size_t i = 0, o = 0; i < totalSize; ++i)
                for (std::size_t k = 0; k < r; ++k) out[o++] = flat[i];
            return SMArray(out, {totalSize * r});
        }
    }

    // Repeat along one axis: shape[axis] *= numberOfRepeats (a stride-0 axis inserted after `axis`).
    SMArray repeat(int numberOfRepeats, int axis) const {
        assert(axis >= 0 && axis < static_cast<int>(ndim) && numberOfRepeats >= 1);
        if (ndim == 1) return repeat(numberOfRepeats);
        const std::size_t r = static_cast<std::size_t>(numberOfRepeats);
        std::vector<std::size_t> newShape = _shape;
        newShape[axis] *= r;
        std::size_t inner = 1, outer = 1;
        for (std::size_t i = axis + 1; i < ndim; ++i) inner *= _shape[i];
        for (int i = 0; i < axis; ++i) outer *= _shape[i];
        if constexpr (hip::transport_of<T>::id >= 0) {
            if (!is_dense()) return contiguous().repeat(numberOfRepeats, axis);
            // dense: (outer, d, inner) seen as (outer, d, r, inner) with strides (d*inner, inner, 0, 1) -- rank 4 whatever ndim is
            return gather({outer, _shape[axis], r, inner}, {_shape[axis] * inner, inner, 0, 1}, std::move(newShape));
        } else {
            std::vector<T> flat(totalSize ? totalSize : 1);
            copy_dense_to(flat.data());
            T *out = new T[calculateTotalSize(newShape) ? calculateTotalSize(newShape) : 1];
            T *dst = out;
            for (std::size_t o = 0; o < outer; ++o)
                for (std::size_t j = 0; j < _shape[axis]; ++j)
                    for (std::size_t k = 0; k < r; ++k) {
                        std::memcpy(dst, flat.data() + (o * _shape[axis] + j) * inner, inner * sizeof(T));
                        dst += inner;
                    }
            return SMArray(out, std::move(newShape));
        }
    }

    // Dot product over all elements (reference SMArray.h:213-215 -> dot_product<T>).
    T operator%(const SMArray &arr) const {
        if (totalSize != arr.totalSize) throw std::runtime_error("dot product: element counts differ");
        if constexpr (hip::dot_dtype_of<T>::id >= 0) {  // the four kernel types and the generic template's other integer types
            hip::DeviceGuard on(common_device(*this, arr));
            std::unique_ptr<SMArray> lhs_holder, rhs_holder;
            const T *pa = dense_device(lhs_holder), *pb = arr.dense_device(rhs_holder);
            return hip::dot_device<T>(pa, pb, totalSize);
        } else if constexpr (std::is_same_v<T, std::complex<double>>) {
            // complex arrays are as resident as any other (16-byte elements in the same Storage): device pointers in, no host staging
            hip::DeviceGuard on(common_device(*this, arr));
            std::unique_ptr<SMArray> lhs_holder, rhs_holder;
            const T *pa = dense_device(lhs_holder), *pb = arr.dense_device(rhs_holder);
            return hip::dot_device_c64(pa, pb, totalSize);
        } else if constexpr (std::is_same_v<T, std::complex<float>>) {  // the generic template's instantiation, on resident arrays
            hip::DeviceGuard on(common_device(*this, arr));
            std::unique_ptr<SMArray> lhs_holder, rhs_holder;
            const T *pa = dense_device(lhs_holder), *pb = arr.dense_device(rhs_holder);
            return hip::dot_device_c32(pa, pb, totalSize);
        } else {
            static_assert(dependent_false<T>::value, "operator%: this element type has no gfx950 dot-product kernel (float, double, every "
                                                     "8- to 64-bit integer type, std::complex<float> and std::complex<double> do)");
        }
    }

    // + - * / with an array (NumPy broadcasting) or a scalar: reference SMArray.h:217-305.  The argument types convert
    // implicitly from SMArray / T (detail::Operand, detail::ScalarArg); the && forms are chosen when the left operand
    // is a temporary, whose not-yet-evaluated chain the operator then continues (see "deferred operator chains" above):
    // `(A * row + B) * 0.5f` is one kernel and 12 bytes per element, not three and 28.
    SMArray operator+(const detail::Operand<T> &rhs) const & { return deferred<AddOp<T>>(*this, false, &rhs, T{}); }
    SMArray operator-(const detail::Operand<T> &rhs) const & { return deferred<SubtractOp<T>>(*this, false, &rhs, T{}); }
    SMArray operator*(const detail::Operand<T> &rhs) const & { return deferred<MultiplyOp<T>>(*this, false, &rhs, T{}); }
    SMArray operator/(const detail::Operand<T> &rhs) const & { return deferred<DivideOp<T>>(*this, false, &rhs, T{}); }
    SMArray operator+(const detail::Operand<T> &rhs) && { return deferred<AddOp<T>>(*this, true, &rhs, T{}); }
    SMArray operator-(const detail::Operand<T> &rhs) && { return deferred<SubtractOp<T>>(*this, true, &rhs, T{}); }
    SMArray operator*(const detail::Operand<T> &rhs) && { return deferred<MultiplyOp<T>>(*this, true, &rhs, T{}); }
    SMArray operator/(const detail::Operand<T> &rhs) && { return deferred<DivideOp<T>>(*this, true, &rhs, T{}); }
    SMArray operator+(const detail::ScalarArg<T> &val) const & { return deferred<AddOp<T>>(*this, false, nullptr, val.value); }
    SMArray operator-(const detail::ScalarArg<T> &val) const & { return deferred<SubtractOp<T>>(*this, false, nullptr, val.value); }
    SMArray operator*(const detail::ScalarArg<T> &val) const & { return deferred<MultiplyOp<T>>(*this, false, nullptr, val.value); }
    SMArray operator/(const detail::ScalarArg<T> &val) const & { return deferred<DivideOp<T>>(*this, false, nullptr, val.value); }
    SMArray operator+(const detail::ScalarArg<T> &val) && { return deferred<AddOp<T>>(*this, true, nullptr, val.value); }
    SMArray operator-(const detail::ScalarArg<T> &val) && { return deferred<SubtractOp<T>>(*this, true, nullptr, val.value); }
    SMArray operator*(const detail::ScalarArg<T> &val) && { return deferred<MultiplyOp<T>>(*this, true, nullptr, val.value); }
    SMArray operator/(const detail::ScalarArg<T> &val) && { return deferred<DivideOp<T>>(*this, true, nullptr, val.value); }

    // Broadcasted `*this Op rhs` for any Op policy: what a user-added operator calls
    // (the README's "add the operator in SMArray.h" step, without editing the class).
    template <typename Op>
    SMArray apply(const SMArray &rhs) const {
        // equal shapes need no resolving (and none of sm::broadcast's vectors): the common case, and the one that
        // matters for tiny arrays, where the operator's host work is a visible part of its few microseconds
        auto br = _shape == rhs._shape ? BroadCastResult{_shape, {}, _strides, {}, rhs._strides, totalSize}
                                       : sm::broadcast(_shape, _strides, rhs._shape, rhs._strides);
        if (br.resultShape.size() > MAX_NDIM) throw std::runtime_error("rank exceeds MAX_NDIM");
        if constexpr (hip::on_device_v<T, Op>) {
            hip::DeviceGuard on(common_device(*this, rhs));  // the kernel runs where the operands live; the result is born there
            SMArray out = device_empty(std::move(br.resultShape));
            apply_into<Op>(rhs, br, out);
            return out;
        } else {
            T *result = new T[br.totalSize ? br.totalSize : 1];
            try {
                element_wise_op<T, Op>(data.read(), br.newStrides1, rhs.data.read(), br.newStrides2, br.totalSize, result,
                                       br.resultShape);
            } catch (...) {
                delete[] result;
                throw;
            }
            return SMArray(result, std::move(br.resultShape));
        }
    }

    // `*this Op scalar` element-wise; shape preserved (reference SMArray.h:226-237 and siblings).
    template <typename Op>
    SMArray apply_scalar(T value) const {
        if constexpr (hip::on_device_v<T, Op>) {
            hip::DeviceGuard on(device());
            SMArray out = device_empty(std::vector<std::size_t>(_shape));
            apply_scalar_into<Op>(value, out);
            return out;
        } else {
            std::vector<T> flat(totalSize ? totalSize : 1);
            copy_dense_to(flat.data());
            T *result = new T[totalSize ? totalSize : 1];
            try {
                array_scalar_op<T, Op>(flat.data(), value, totalSize, result);
            } catch (...) {
                delete[] result;
                throw;
            }
            return SMArray(result, std::vector<std::size_t>(_shape));
        }
    }

private:
    // The device-side body of apply<Op>: `*this Op rhs` into `out` (dense, of br's result shape, on the operands' GPU).
    template <typename Op>
    void apply_into(const SMArray &rhs, const BroadCastResult &br, SMArray &out) const {
        {
            if (!out._shape.empty() && out.totalSize <= SMHIP_INLINE_MAX_OUTPUTS && (host_only_small() || rhs.host_only_small()) &&
                hip::device_op<Op>::id() <= SMHIP_OP_POW) {
                // a tiny operand that exists only on the host rides in the kernel's argument block: one packet instead
                // of an upload packet plus a kernel packet (smhip_elementwise_inline)
                const std::size_t ia = host_only_small(), ib = rhs.host_only_small();
                const auto sa = hip::to_i64(br.newStrides1), sb = hip::to_i64(br.newStrides2), sh = hip::to_i64(out._shape);
                hip::check(smhip_elementwise_inline(hip::device_op<Op>::id(), hip::dtype_of<T>::id,
                                                    ia ? static_cast<const void *>(data.read()) : device_data(), ia, sa.data(),
                                                    ib ? static_cast<const void *>(rhs.data.read()) : rhs.device_data(), ib, sb.data(),
                                                    sh.data(), static_cast<int>(sh.size()), out.device_data_mut()));
                if (ia) ++data.storage()->inline_uses;
                if (ib) ++rhs.data.storage()->inline_uses;
                return;
            }
            hip::element_wise_op_device<T, Op>(device_data(), br.newStrides1, rhs.device_data(), br.newStrides2,
                                               out.device_data_mut(), out._shape);
        }
    }

    // The device-side body of apply_scalar<Op>, into `out` (dense, of this array's shape).
    template <typename Op>
    void apply_scalar_into(T value, SMArray &out) const {
        {
            if (!_shape.empty() && totalSize <= SMHIP_INLINE_MAX_OUTPUTS && host_only_small() && hip::device_op<Op>::id() <= SMHIP_OP_POW) {
                // host-built tiny array op scalar: both ride in the launch packet
                const auto sa = hip::to_i64(_strides), sh = hip::to_i64(_shape);
                const std::vector<std::int64_t> zeros(sh.size(), 0);
                hip::check(smhip_elementwise_inline(hip::device_op<Op>::id(), hip::dtype_of<T>::id, data.read(), host_only_small(), sa.data(),
                                                    &value, sizeof(T), zeros.data(), sh.data(), static_cast<int>(sh.size()),
                                                    out.device_data_mut()));
                ++data.storage()->inline_uses;
                return;
            }
            if (is_dense()) {
                hip::array_scalar_op_device<T, Op>(device_data(), value, totalSize, out.device_data_mut());
            } else {  // a view: honour its strides (the reference reads views as flat here)
                hip::DeviceBuffer s(sizeof(T));
                hip::check(smhip_upload(s.get(), &value, sizeof(T)));
                hip::element_wise_op_device<T, Op>(device_data(), _strides, s.template as<T>(),
                                                   std::vector<std::size_t>(_shape.size(), 0), out.device_data_mut(), _shape);
            }
        }
    }

public:
    [[nodiscard]] std::string toString() const {
        std::ostringstream oss;
        const T *base = data.read();
        std::function<void(std::size_t, std::size_t)> emit = [&](std::size_t offset, std::size_t dim) {
            oss << "[";
            for (std::size_t i = 0; i < _shape[dim]; ++i) {
                if (dim + 1 == _shape.size()) {
                    if (i) oss << ", ";
                    oss << base[offset + i * _strides[dim]];
                } else {
                    if (i) oss << ",\n";
                    emit(offset + i * _strides[dim], dim + 1);
                }
            }
            oss << "]";
        };
        if (_shape.empty()) return "[]";
        emit(0, 0);
        return oss.str();
    }

    [[nodiscard]] const std::vector<std::size_t> &shape() const { return _shape; }
    [[nodiscard]] const std::vector<std::size_t> &strides() const { return _strides; }

    ~SMArray() = default;  // Storage is reference-counted: the last owner/view releases it

    // ---- MI355X extensions (not in the reference) -------------------------------
    // Device pointer to this array's first element, values current.  Valid until the
    // array (and its views) die.  Work is stream-ordered on libsmhip's stream.
    // The host mirror for READING: brought up to date if needed, and -- unlike `data`, whose every use must assume a write
    // (`arr.data[i] = v` is the reference's idiom) -- the device copy stays valid, so a later operator uploads nothing.
    const T *cdata() const { return data.read(); }
    const T *device_data() const { return data.storage()->dev_ro() + data.offset(); }
    T *device_data_mut() { return data.storage()->dev_wo() + data.offset(); }
    bool is_dense() const { return is_contiguous(_shape, _strides); }
    bool is_view() const { return isView; }
    void copy_dense_out(T *dst) const { copy_dense_to(dst); }  // this array's elements, row-major, into host memory
    int device() const { return data.storage()->device; }  // the GPU this array's elements live on
    // The GPU two operands of one operator share.  Kernels run on the operands' device (the operators take a DeviceGuard on
    // it, whatever GPU the calling thread is on); operands resident on DIFFERENT GPUs cannot meet in one kernel -- peer
    // access is not assumed -- and that is an error, not a silent wrong-device launch.
    // An operand that has no device buffer yet (host-built, or a result still pending) belongs to no GPU in particular: it
    // adopts its partner's and is uploaded / born there.
    static int common_device(const SMArray &x, const SMArray &y) {
        auto &sx = *x.data.storage(), &sy = *y.data.storage();
        if (sx.device != sy.device) {
            if (!sx.dev && !sx.pending) sx.device = sy.device;
            else if (!sy.dev && !sy.pending) sy.device = sx.device;
        }
        const int dx = sx.device, dy = sy.device;
        if (dx != dy)
            throw std::runtime_error("simpleMath/MI355X: the operands live on different GPUs (" + std::to_string(dx) + " and " + std::to_string(dy) +
                                     "); bring them to one device first (sm::Sharded<T>::gather / scatter, or build both on the same GPU)");
        return dx;
    }

    // (a Op1 b) Op2 c (c == nullptr: Op2 scalar) as one recorded chain, evaluated before returning: what sm::fused is for
    // operands that broadcast against each other.
    template <typename Op1, typename Op2>
    static SMArray chain_of(const SMArray &a, const SMArray &b, const SMArray *c, T scalar) {
        detail::flush_pending();
        SMArray t = deferred_arrays<Op1>(a, false, &b, false, T{});
        SMArray r = deferred_arrays<Op2>(t, true, c, false, scalar);
        detail::flush_pending();
        return r;
    }

    // A new dense array whose elements exist only in HBM so far.
    static SMArray device_empty(std::vector<std::size_t> &&shape) {
        SMArray a;
        a._shape = std::move(shape);
        a._strides = detail::dense_strides(a._shape);
        a.ndim = a._shape.size();
        a.totalSize = calculateTotalSize(a._shape);
        a.data = detail::HostPtr<T>(std::make_shared<detail::Storage<T>>(a.totalSize), 0);
        return a;
    }
    static SMArray device_full(std::vector<std::size_t> &&shape, T value) {
        SMArray a = device_empty(std::move(shape));
        if constexpr (hip::dtype_of<T>::id >= 0) {
            hip::check(smhip_fill(hip::dtype_of<T>::id, a.device_data_mut(), &value, a.totalSize));
        } else {
            T *h = static_cast<T *>(a.data);
            for (std::size_t i = 0; i < a.totalSize; ++i) h[i] = value;
        }
        return a;
    }
    // Dense copy of a (possibly strided) array, made on the device.
    SMArray contiguous() const {
        if (hip::transport_of<T>::id >= 0 && _shape.size() + (hip::transport_of<T>::lanes > 1 ? 1 : 0) <= MAX_NDIM) {
            return gather(_shape, _strides, std::vector<std::size_t>(_shape));
        } else {
            T *out = new T[totalSize ? totalSize : 1];
            copy_dense_to(out);
            return SMArray(out, std::vector<std::size_t>(_shape));
        }
    }
    // Sum of all elements, accumulated in fp64 on the device.
    // ... of a TEMPORARY: sm::pow(a - b, 2.0f).sum().  The unevaluated result of this expression's chain, which nobody can look
    // at afterwards, is not computed at all: its sum is taken in the chain's own pass (smhip_chain_sum).  (Only the rvalue form:
    // a named array -- a by-value parameter bound to such a temporary -- may be read again after its sum.)
    double sum() && {
        static_assert(hip::dtype_of<T>::id >= 0, "sum(): element type has no kernels");
        if (detail::Chain<T> *c = pending_temporary()) return c->run_sum();
        return static_cast<const SMArray &>(*this).sum();
    }
    double sum() const & {
        static_assert(hip::dtype_of<T>::id >= 0, "sum(): element type has no kernels");
        hip::DeviceGuard on(device());
        std::unique_ptr<SMArray> holder;
        const T *p = dense_device(holder);
        double s = 0;
        hip::check(smhip_sum(hip::dtype_of<T>::id, p, totalSize, &s));
        return s;
    }

private:
    std::vector<std::size_t> _shape;
    std::vector<std::size_t> _strides;
    std::size_t ndim = 0;
    bool isView = false;

    SMArray() = default;  // internal: views and device-born results

    friend struct detail::Chain<T>;

    // Another handle on the same elements (same storage, offset, shape, strides): what a recorded chain keeps of an operand.
    SMArray alias() const {
        SMArray v;
        v.data = data;
        v._shape = _shape;
        v._strides = _strides;
        v.ndim = ndim;
        v.totalSize = totalSize;
        v.isView = true;
        return v;
    }

    // Is this array the unevaluated result of the chain pending on this thread, with nobody else holding its storage?  Only
    // then may an operator that consumes it (as a temporary) continue the chain instead of reading the result.
    detail::Chain<T> *continuable() const {
        detail::Chain<T> *c = pending_temporary();
        return (c && !c->full()) ? c : nullptr;
    }
    // ... the chain itself, room for another stage or not (a reduction of its value takes no stage)
    detail::Chain<T> *pending_temporary() const {
        const auto &st = data.storage();
        if (!st || !st->pending || detail::tls_pending != st->pending.get() || isView || data.offset() != 0 || st.use_count() != 1) return nullptr;
        return static_cast<detail::Chain<T> *>(st->pending.get());
    }

public:
    // sm::pow(<temporary>, s) (UserFunctions.h): when the temporary is the unevaluated result of this full-expression's chain --
    // sm::pow(a - b, 2.0f) -- the power is one more stage of that chain (s = 2 is fused into its kernel, any other exponent cuts
    // the chain and runs pow's own evaluation on the value so far: the same bits as the operator by itself).  The chain is
    // launched by the operators' argument temporaries at the end of the full-expression, like every chain.
    static SMArray pow_of(const SMArray &x, bool x_temporary, T value) {
        if constexpr (hip::on_device_v<T, PowOp<T>> && hip::is_builtin_op<PowOp<T>>::value) {
            if (x_temporary) {
                if (detail::Chain<T> *cx = x.continuable()) {
                    hip::DeviceGuard on(x.device());
                    SMArray out = device_empty(std::vector<std::size_t>(x._shape));
                    detail::Storage<T> &ost = *out.data.storage();
                    ost.pending = std::move(x.data.storage()->pending);
                    cx->retarget(&ost, out._shape);
                    cx->push(SMHIP_OP_POW, false, value);
                    return out;
                }
            }
        }
        return x.template apply_scalar<PowOp<T>>(value);
    }

private:
    // x Op y (y == nullptr: x Op scalar), recorded rather than launched when it can be part of a one-pass chain: built-in
    // + - * / on an element type with kernels.  See "deferred operator chains" at the top of this file.
    template <typename Op>
    static SMArray deferred(const SMArray &x, bool x_temporary, const detail::Operand<T> *rhs, T scalar) {
        return deferred_arrays<Op>(x, x_temporary, rhs ? &rhs->array() : nullptr, rhs && rhs->temporary(), scalar);
    }
    template <typename Op>
    static SMArray deferred_arrays(const SMArray &x, bool x_temporary, const SMArray *y, bool y_temporary, T scalar) {
        constexpr bool chainable = hip::on_device_v<T, Op> && hip::is_builtin_op<Op>::value && !std::is_same_v<Op, PowOp<T>>;
        if constexpr (!chainable) {
            return y ? x.template apply<Op>(*y) : x.template apply_scalar<Op>(scalar);
        } else {
            std::vector<std::size_t> shape = (!y || x._shape == y->_shape) ? x._shape : sm::broadcast(x._shape, x._strides, y->_shape, y->_strides).resultShape;
            if (shape.size() > MAX_NDIM) throw std::runtime_error("rank exceeds MAX_NDIM");
            const std::size_t n = calculateTotalSize(shape);
            // nothing to gain from waiting: empty and 0-d results, and tiny host-built operands (they ride in the launch
            // packet of the plain operator, apply_into)
            detail::Chain<T> *cx = x_temporary ? x.continuable() : nullptr;
            detail::Chain<T> *cy = (!cx && y && y_temporary) ? y->continuable() : nullptr;
            if (shape.empty() || n == 0 || (!cx && !cy && n <= SMHIP_INLINE_MAX_OUTPUTS && (x.host_only_small() || (y && y->host_only_small()))))
                return y ? x.template apply<Op>(*y) : x.template apply_scalar<Op>(scalar);
            const int device = y ? common_device(x, *y) : x.device();
            const int op = hip::device_op<Op>::id();
            hip::DeviceGuard on(device);
            SMArray out = device_empty(std::move(shape));  // no device buffer yet: Storage allocates on first use
            detail::Storage<T> &ost = *out.data.storage();
            if (cx) {  // (pending temporary) Op y: the chain goes on
                ost.pending = std::move(x.data.storage()->pending);
                cx->retarget(&ost, out._shape);
                if (y) cx->push(op, false, *y); else cx->push(op, false, scalar);
            } else if (cy) {  // x Op (pending temporary): the same, with the stage's operands exchanged
                ost.pending = std::move(y->data.storage()->pending);
                cy->retarget(&ost, out._shape);
                cy->push(op, true, x);
            } else {
                detail::flush_pending();  // one chain per thread
                auto c = std::make_unique<detail::Chain<T>>(&ost, out._shape, device, x);
                if (y) c->push(op, false, *y); else c->push(op, false, scalar);
                detail::tls_pending = c.get();
                ost.pending = std::move(c);
            }
            return out;
        }
    }

    void finish_owning(T *buffer) {
        ndim = _shape.size();
        totalSize = calculateTotalSize(_shape);
        _strides = detail::dense_strides(_shape);
        data = detail::HostPtr<T>(std::make_shared<detail::Storage<T>>(buffer, totalSize), 0);
    }

    std::size_t offset_of(std::initializer_list<std::size_t> indices, [[maybe_unused]] bool exact) const {
        assert((exact ? indices.size() == _shape.size() : indices.size() <= _shape.size()) && "wrong number of indices");
        std::size_t off = 0, k = 0;
        for (std::size_t idx : indices) {
            assert(idx < _shape[k] && "Index out of bounds");
            off += idx * _strides[k++];
        }
        return off;
    }

    // f(linear, element offset) over all elements in row-major order
    template <typename F>
    void for_each_offset(F &&f) const {
        std::vector<std::size_t> idx(_shape.size(), 0);
        for (std::size_t linear = 0; linear < totalSize; ++linear) {
            std::size_t off = 0;
            for (std::size_t k = 0; k < idx.size(); ++k) off += idx[k] * _strides[k];
            f(linear, off);
            for (std::size_t k = idx.size(); k-- > 0;) {
                if (++idx[k] < _shape[k]) break;
                idx[k] = 0;
            }
        }
    }

    void copy_dense_to(T *dst) const {
        const T *base = data.read();
        if (is_dense()) std::copy(base, base + totalSize, dst);
        else for_each_offset([&](std::size_t linear, std::size_t off) { dst[linear] = base[off]; });
    }

    // Dense device copy of this array's storage seen through (vshape, vstrides), labelled `outShape`
    // (same element count): SMHIP_OP_LEFT through the broadcast kernels.
    SMArray gather(const std::vector<std::size_t> &vshape, const std::vector<std::size_t> &vstrides,
                   std::vector<std::size_t> &&outShape) const {
        hip::DeviceGuard on(device());
        SMArray out = device_empty(std::move(outShape));
        auto sh = hip::to_i64(vshape), st = hip::to_i64(vstrides);
        lane_axis(sh, st, nullptr);
        const std::vector<std::int64_t> zeros(sh.size(), 0);
        const T *src = device_data();
        hip::check(smhip_elementwise(SMHIP_OP_LEFT, hip::transport_of<T>::id, src, st.data(), src, zeros.data(), sh.data(),
                                     static_cast<int>(sh.size()), out.device_data_mut()));
        return out;
    }

    // An element type that moves as `lanes` elements of a kernel type (std::complex: {re, im}) gets an innermost axis of
    // extent `lanes`, stride 1, and its element strides counted in lanes (math/ops.h: transport_of).
    static void lane_axis(std::vector<std::int64_t> &shape, std::vector<std::int64_t> &strides, std::vector<std::int64_t> *strides2) {
        constexpr int lanes = hip::transport_of<T>::lanes;
        if constexpr (lanes > 1) {
            for (auto &x : strides) x *= lanes;
            if (strides2) for (auto &x : *strides2) x *= lanes;
            shape.push_back(lanes);
            strides.push_back(1);
            if (strides2) strides2->push_back(1);
        } else {
            (void)shape; (void)strides; (void)strides2;
        }
    }

    // Bytes from this array's first element to the end of its storage when the elements exist ONLY in host memory and
    // that is at most SMHIP_INLINE_MAX_BYTES (such an operand can ride in a kernel's argument block); 0 otherwise.
    std::size_t host_only_small() const {
        const auto &st = data.storage();
        if (!st || st->dev_valid || !st->host_valid || st->count <= data.offset()) return 0;
        // Once only: an array that is used AGAIN is worth its upload -- reading the argument block costs the kernel a
        // trip over the fabric (a launch with an inline operand takes ~4.0 us against ~2.8 us with resident ones), so
        // the second use uploads it and every later one runs resident.  simple_check builds its array anew each time
        // and never gets that far; BM_SMArrayPow_1D / _2D reuse theirs and do.
        if (st->inline_uses != 0) return 0;
        const std::size_t bytes = (st->count - data.offset()) * sizeof(T);
        return bytes <= SMHIP_INLINE_MAX_BYTES ? bytes : 0;
    }

    // Device pointer to a dense version of this array (itself when already dense).
    const T *dense_device(std::unique_ptr<SMArray> &holder) const {
        if (is_dense()) return device_data();
        holder.reset(new SMArray(contiguous()));
        return holder->device_data();
    }

    // arr(i, SLICE(a, b), ...): INDEX entries drop their axis, Slice entries keep a sub-range;
    // missing trailing entries mean SLICE_ALL.  Shares storage (reference SMArray.h:397-437).
    SMArray make_view(std::initializer_list<Slice> slices) const {
        SMArray v;
        std::size_t off = data.offset();
        std::size_t axis = 0;
        auto it = slices.begin();
        for (; axis < _shape.size(); ++axis) {
            const Slice s = it != slices.end() ? *it++ : Slice(0);
            assert(s.start <= _shape[axis] && "slice start out of bounds");
            off += s.start * _strides[axis];
            if (s.sliceType == Slice::INDEX) continue;
            const std::size_t stop = s.open_ended() ? _shape[axis] : s.end;
            assert(stop <= _shape[axis] && stop >= s.start && "slice end out of bounds");
            // SliceStep only names SINGLE_STEP, but the reference's view code honours any positive value stored in
            // `step` (SMArray.h:416-424: extent = ceil(range / step), stride * step); so does this one
            const std::size_t step = static_cast<int>(s.step) > 1 ? static_cast<std::size_t>(s.step) : 1;
            v._shape.push_back((stop - s.start + step - 1) / step);
            v._strides.push_back(_strides[axis] * step);
        }
        v.data = detail::HostPtr<T>(data.storage(), off);
        v.ndim = v._shape.size();
        v.totalSize = calculateTotalSize(v._shape);
        v.isView = true;
        return v;
    }
};

namespace detail {

// A recorded operator chain: r = leaf[0]; r = r op[k] leaf[k + 1] (swapped[k]: leaf[k + 1] op[k] r).  Owned by the Storage
// of its (not yet computed) result; run() evaluates it there -- through smhip_chain (csrc/chain.hip: one pass where the
// operands allow), or, for a chain that stayed one operator long, through the plain operator's entry point.  A leaf keeps
// the operand's storage alive and a copy of its offset / shape / strides in place (no allocation per operand: recording an
// operator costs one allocation, the chain itself).
template <typename T>
struct Chain final : PendingBase {
    static constexpr int kMaxLeaves = 8;
    struct Leaf {
        std::shared_ptr<Storage<T>> st;
        std::size_t offset = 0, total = 0;
        int ndim = 0;
        std::size_t shape[MAX_NDIM] = {}, strides[MAX_NDIM] = {};
        bool is_scalar = false;
        T value{};
        bool dense() const {
            std::size_t expect = 1;
            for (int i = ndim; i-- > 0;) {
                if (strides[i] != expect) return false;
                expect *= shape[i];
            }
            return true;
        }
    };
    Storage<T> *out;
    std::size_t out_offset = 0;  // the result's first element in `out` (assignment into a dense block of an existing array)
    bool out_partial = false;    // ... which is then only part of that storage: its other elements stay what they are
    int out_ndim = 0;
    std::size_t out_shape[MAX_NDIM] = {};
    int device;
    int n_leaves = 0;
    Leaf leaves[kMaxLeaves];
    int ops[kMaxLeaves] = {}, swapped[kMaxLeaves] = {};

    Chain(Storage<T> *o, const std::vector<std::size_t> &sh, int dev, const SMArray<T> &head) : out(o), device(dev) {
        retarget(o, sh);
        set(leaves[n_leaves++], head);
    }
    bool full() const { return n_leaves == kMaxLeaves; }
    void retarget(Storage<T> *o, const std::vector<std::size_t> &sh, std::size_t offset = 0, bool partial = false) {
        out = o;
        out_offset = offset;
        out_partial = partial;
        out_ndim = static_cast<int>(sh.size());
        for (int i = 0; i < out_ndim; ++i) out_shape[i] = sh[i];
    }
    // May the chain's result be written straight into the dense block (st, offset, shape)?  Yes if no operand lives in that
    // storage, or every operand that does IS that block (same offset, same shape, dense): an in-place elementwise update.
    bool may_write_into(const Storage<T> *st, std::size_t offset, const std::vector<std::size_t> &sh) const {
        if (static_cast<int>(sh.size()) != out_ndim) return false;
        for (int i = 0; i < out_ndim; ++i)
            if (sh[i] != out_shape[i]) return false;
        for (int k = 0; k < n_leaves; ++k) {
            const Leaf &lf = leaves[k];
            if (lf.is_scalar || lf.st.get() != st) continue;
            if (lf.offset != offset || lf.ndim != out_ndim || !lf.dense()) return false;
            for (int i = 0; i < out_ndim; ++i)
                if (lf.shape[i] != out_shape[i]) return false;
        }
        return true;
    }
    void push(int op, bool swap, const SMArray<T> &operand) {
        ops[n_leaves - 1] = op;
        swapped[n_leaves - 1] = swap;
        set(leaves[n_leaves++], operand);
    }
    void push(int op, bool swap, T scalar) {
        ops[n_leaves - 1] = op;
        swapped[n_leaves - 1] = swap;
        Leaf &lf = leaves[n_leaves++];
        lf.is_scalar = true;
        lf.value = scalar;
    }

    void run() override {
        std::unique_ptr<PendingBase> self = std::move(out->pending);  // the result is no longer pending; *this lives to the end of run()
        if (tls_pending == this) tls_pending = nullptr;
        hip::DeviceGuard on(device);
        const int n = n_leaves, nd = out_ndim;
        std::int64_t strides[kMaxLeaves * MAX_NDIM] = {}, sh[MAX_NDIM];
        for (int i = 0; i < nd; ++i) sh[i] = static_cast<std::int64_t>(out_shape[i]);
        const void *ptrs[kMaxLeaves] = {};
        T scalars[kMaxLeaves] = {};
        for (int k = 0; k < n; ++k) {
            const Leaf &lf = leaves[k];
            if (lf.is_scalar) { scalars[k] = lf.value; continue; }
            const int shift = nd - lf.ndim;  // right-aligned, as sm::broadcast aligns them (SMUtils.h:51-72)
            for (int i = 0; i < lf.ndim; ++i)
                strides[k * nd + shift + i] = (lf.shape[i] == 1 && out_shape[shift + i] != 1) ? 0 : static_cast<std::int64_t>(lf.strides[i]);
            ptrs[k] = lf.st->dev_ro() + lf.offset;
        }
        T *dst = (out_partial ? out->dev_rw() : out->dev_wo()) + out_offset;
        if (n == 2) {  // one operator after all: the plain operator's entry points, exactly as before chains existed
            ++tls_fusion_stats.single_ops;
            const Leaf &x = leaves[0], &y = leaves[1];
            if (!y.is_scalar) {
                hip::check(smhip_elementwise(ops[0], hip::dtype_of<T>::id, ptrs[0], strides, ptrs[1], strides + nd, sh, nd, dst));
            } else if (x.dense()) {
                hip::check(smhip_array_scalar(ops[0], hip::dtype_of<T>::id, ptrs[0], &y.value, x.total, dst));
            } else {  // a view against a scalar: honour its strides (the reference reads views as flat here)
                hip::DeviceBuffer s(sizeof(T));
                hip::check(smhip_upload(s.get(), &y.value, sizeof(T)));
                hip::check(smhip_elementwise(ops[0], hip::dtype_of<T>::id, ptrs[0], strides, s.get(), strides + nd, sh, nd, dst));  // strides + nd: the scalar's row, all zero
            }
            return;
        }
        ++tls_fusion_stats.chains;
        tls_fusion_stats.fused_stages += static_cast<unsigned long long>(n - 1);
        hip::check(smhip_chain(hip::dtype_of<T>::id, n, ptrs, strides, scalars, ops, swapped, sh, nd, dst));
    }

    // The SUM of the chain's value instead of the value (SMArray::sum() of the expression's unevaluated temporary): the same
    // operands and stages through smhip_chain_sum; the result's storage is left without elements -- it is a temporary nobody
    // else refers to (pending_temporary()), and goes away with the expression.
    double run_sum() {
        std::unique_ptr<PendingBase> self = std::move(out->pending);
        if (tls_pending == this) tls_pending = nullptr;
        hip::DeviceGuard on(device);
        const int n = n_leaves, nd = out_ndim;
        std::int64_t strides[kMaxLeaves * MAX_NDIM] = {}, sh[MAX_NDIM];
        for (int i = 0; i < nd; ++i) sh[i] = static_cast<std::int64_t>(out_shape[i]);
        const void *ptrs[kMaxLeaves] = {};
        T scalars[kMaxLeaves] = {};
        for (int k = 0; k < n; ++k) {
            const Leaf &lf = leaves[k];
            if (lf.is_scalar) { scalars[k] = lf.value; continue; }
            const int shift = nd - lf.ndim;
            for (int i = 0; i < lf.ndim; ++i)
                strides[k * nd + shift + i] = (lf.shape[i] == 1 && out_shape[shift + i] != 1) ? 0 : static_cast<std::int64_t>(lf.strides[i]);
            ptrs[k] = lf.st->dev_ro() + lf.offset;
        }
        ++tls_fusion_stats.chains;
        ++tls_fusion_stats.summed_chains;
        tls_fusion_stats.fused_stages += static_cast<unsigned long long>(n - 1);
        double total = 0;
        hip::check(smhip_chain_sum(hip::dtype_of<T>::id, n, ptrs, strides, scalars, ops, swapped, sh, nd, &total));
        return total;
    }

private:
    static void set(Leaf &lf, const SMArray<T> &a) {
        lf.st = a.data.storage();
        lf.offset = a.data.offset();
        lf.total = a.totalSize;
        lf.ndim = static_cast<int>(a._shape.size());
        for (int i = 0; i < lf.ndim; ++i) { lf.shape[i] = a._shape[i]; lf.strides[i] = a._strides[i]; }
    }
};

}  // namespace detail

// What the operator fusion did on the calling thread so far (chains evaluated as one call, operators inside them, deferred
// operators that stayed alone).
inline detail::FusionStats fusion_stats() { return detail::tls_fusion_stats; }

}  // namespace sm
