// Error reporting of the C ABI: int result codes + a thread-local message (twk_last_error()).
// Stands where the reference throws std::runtime_error with file:line (inc/CheckMacros.h:38-80);
// nothing is ever thrown across the C boundary.
#pragma once
#include "../../include/tweeker_hip.h"
#include <string>

int twkSetError(int code, const std::string& message);
