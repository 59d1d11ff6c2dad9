// Syntax-check shim (integration/README): just enough of absl for the reference's iomgr headers.
#pragma once
#include <string>
#include <string_view>
namespace absl { using string_view = std::string_view; }
