// rcm_internal.hpp -- shared between the host-only and the HIP translation units.
#pragma once
#include <string>

namespace rcm {
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);  // records msg, returns code
}  // namespace rcm
