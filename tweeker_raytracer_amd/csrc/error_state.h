// Error reporting of the C ABI: int result codes + a thread-local message (twk_last_error()).
// Stands where the reference throws std::runtime_error with file:line (inc/CheckMacros.h:38-80);
// nothing is ever thrown across the C boundary.
#pragma once
#include "../../include/tweeker_hip.h"
#include <string>

int twkSetError(int code, const std::string& message);

// Every extern "C" entry point is a function-try-block that ends in TWK_CATCH: std::bad_alloc / length_error from the
// std::vector copies of caller data (or anything else a callee throws) becomes a result code, never std::terminate in
// the caller's process.
#include <exception>
#include <new>
#define TWK_CATCH(name)                                                                                                        \
  catch (const std::bad_alloc&) { return twkSetError(TWK_ERROR_OUT_OF_MEMORY, name ": out of host memory"); }                  \
  catch (const std::exception& e_) { return twkSetError(TWK_ERROR_INVALID_VALUE, std::string(name ": ") + e_.what()); }        \
  catch (...) { return twkSetError(TWK_ERROR_INVALID_VALUE, name ": unknown C++ exception"); }
