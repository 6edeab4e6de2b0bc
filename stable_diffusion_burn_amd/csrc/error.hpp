// error.hpp -- the exception every layer of libsdmi throws; the C ABI turns it into an sdmi_status + message.
// Kept free of HIP so that the host-only translation units (tokenizer, PNG writer, .mpk reader) build with a plain
// C++ compiler for the sanitizer test target (tests/san/).
#pragma once
#include <stdexcept>
#include <string>

#include "../../include/sdmi.h"

namespace sdmi {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string& m) : std::runtime_error(m), status(st) {}
};

}  // namespace sdmi
