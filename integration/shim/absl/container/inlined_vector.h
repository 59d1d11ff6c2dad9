#pragma once
#include <vector>
namespace absl { template <typename T, size_t N, typename A = std::allocator<T>> class InlinedVector : public std::vector<T, A> { public: using std::vector<T, A>::vector; }; }
