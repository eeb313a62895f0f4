#include "data_format.hpp"

#include <cctype>
#include <cstdlib>

namespace viewer {

// Same accepted inputs as the reference parser: the leading alphabetic run names the
// format ("SH" -> SH, anything else -> RGBA); the rest is read with atoi as basis_dim;
// a string with no non-alphabetic character yields RGBA with basis_dim = -1.
void DataFormat::parse(const std::string &str) {
    size_t split = std::string::npos;
    for (size_t i = 0; i < str.size(); ++i) {
        if (!std::isalpha(static_cast<unsigned char>(str[i]))) {
            split = i;
            break;
        }
    }
    if (split == std::string::npos) {
        basis_dim = -1;
        format = RGBA;
        return;
    }
    basis_dim = std::atoi(str.c_str() + split);
    format = (str.compare(0, split, "SH") == 0) ? SH : RGBA;
}

std::string DataFormat::to_string() const {
    std::string out = format == SH ? "SH" : format == RGBA ? "RGBA" : "UNKNOWN";
    if (basis_dim != -1) out += std::to_string(basis_dim);
    return out;
}

}  // namespace viewer
