// call_vec.h -- the vector type the addSeed call lists live in.  Lists reach tens of millions of 16-byte records
// per record (every pass-streak of the anchored scan is a call), so growing them must not value-initialise
// (zero-fill) half a gigabyte that is overwritten right away: resize() default-initialises instead.
#pragma once
#include <memory>
#include <vector>

#include "ribbit_hip.h"

namespace rb {

template <typename T>
struct DefaultInitAllocator : std::allocator<T> {
    template <typename U> struct rebind { using other = DefaultInitAllocator<U>; };
    using std::allocator<T>::allocator;
    template <typename U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... Args> void construct(U *p, Args &&...args) { ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...); }
};

using CallVec = std::vector<RibbitCall, DefaultInitAllocator<RibbitCall>>;

}  // namespace rb
