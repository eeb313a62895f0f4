// npz.hpp -- minimal .npy/.npz reader and writer for svox / PlenOctree tree files.
//
// Role of the reference's vendored cnpy (3rdparty/cnpy/cnpy.cpp:303-369 npz_load),
// written fresh: it accepts what that loader accepts -- stored and deflate members,
// ZIP64 size fields, little-endian numeric dtypes and numpy '<U' strings -- and is
// driven from the ZIP central directory rather than by scanning local headers.
#pragma once

#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace viewer::npz {

struct Array {
    std::vector<size_t> shape;
    char kind = 'f';        // numpy dtype kind: f, i, u, b, U, S ...
    size_t word_size = 0;   // bytes per element ('<U3' -> 12)
    bool fortran_order = false;
    std::vector<uint8_t> bytes;

    size_t num_vals() const {
        size_t n = 1;
        for (size_t s : shape) n *= s;
        return n;
    }
    template <typename T>
    const T *data() const { return reinterpret_cast<const T *>(bytes.data()); }
    template <typename T>
    T *data() { return reinterpret_cast<T *>(bytes.data()); }
};

using Archive = std::map<std::string, Array>;

// Parse one in-memory .npy image.
Array parse_npy(const uint8_t *buf, size_t len);
// Load every member of an .npz (names have the ".npy" suffix stripped).
Archive load(const std::string &path);

// Writer: stored (uncompressed) members, numpy-readable.
struct Writer {
    explicit Writer(const std::string &path);
    ~Writer();
    void add(const std::string &name, const std::string &descr, const std::vector<size_t> &shape,
             const void *data, size_t nbytes);
    void add_unicode(const std::string &name, const std::string &ascii);  // 0-d '<U{n}'
    void close();

private:
    struct Entry {
        std::string name;
        uint32_t crc;
        uint64_t size, offset;
    };
    std::FILE *fp_ = nullptr;
    std::vector<Entry> entries_;
    uint64_t pos_ = 0;
    void put(const void *p, size_t n);
};

}  // namespace viewer::npz
