#pragma once
#include <memory>
namespace absl { using std::make_unique; template <typename T> std::unique_ptr<T> WrapUnique(T* p) { return std::unique_ptr<T>(p); } }
