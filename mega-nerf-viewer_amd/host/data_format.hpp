// data_format.hpp -- viewer::DataFormat (reference include/data_format.hpp:7-22).
#pragma once

#include <string>

namespace viewer {

struct DataFormat {
    enum {
        RGBA,  // rows hold r, g, b, sigma
        SH,    // rows hold 3 * basis_dim SH coefficients + sigma
        _COUNT,
    } format = RGBA;

    // SH basis functions per colour channel (-1: none given)
    int basis_dim = -1;

    // Parse a string like "SH9" (reference src/data_format.cpp:5-24)
    void parse(const std::string &str);
    // Back to a string (reference src/data_format.cpp:26-41)
    std::string to_string() const;
};

}  // namespace viewer
