// mnv_error.h -- thread-local last-error text behind mnv_last_error(); host-only header.
#pragma once

#include <string>

namespace mnv {
int set_error(int code, const std::string &msg);
}  // namespace mnv
