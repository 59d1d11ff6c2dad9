// Syntax-check shim.
#pragma once
#include "absl/time/time.h"
