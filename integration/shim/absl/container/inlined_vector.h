// Syntax-check shim (integration/README.md): declaration-level stand-in, never linked.
#pragma once
#include <algorithm>  // (the real header pulls it in; the event engines rely on that for std::remove)
#include <vector>
namespace absl { template <typename T, size_t N, typename A = std::allocator<T>> class InlinedVector : public std::vector<T, A> { public: using std::vector<T, A>::vector; }; }
