// n3tree.hpp -- viewer::N3Tree, the drop-in tree loader of the reference
// (reference include/n3tree/n3tree.hpp:17-69, src/n3tree/n3tree.cpp:16-246).
//
// Same public surface (open, move_to_device, N, data_dim, data_format, scale, offset,
// data, child, parent, sample_counts, capacity, pack_index, unpack_index); the
// torch::Tensor members of the reference become plain host vectors plus raw HIP
// device buffers -- there is no libtorch in the product.  gen_wireframe (GL overlay
// only) is not carried.
#pragma once

#include <array>
#include <cstdint>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/mnv.h"
#include "data_format.hpp"

namespace viewer {

struct N3Tree {
    N3Tree();
    explicit N3Tree(const std::string &path);
    ~N3Tree();
    N3Tree(const N3Tree &) = delete;
    N3Tree &operator=(const N3Tree &) = delete;

    // Open an svox / PlenOctree .npz.  A missing file prints a message and leaves N == 0
    // ("draw nothing"), malformed content throws std::runtime_error -- as n3tree.cpp:16-26,
    // :114,:121,:180,:196,:200.
    void open(const std::string &path);

    // Allocate [max_capacity, ...] device arrays, copy the first `capacity` rows and build
    // the packed accel used by the tuned march kernel (n3tree.cpp:207-246).
    // Throws std::runtime_error on HIP failure.
    void move_to_device(long max_capacity, bool need_parent, bool need_sample_counts, void *hip_stream = nullptr);
    void free_device();
    // Rebuild the packed accel from the current device arrays (after pruning renumbered the chunks).
    void rebuild_accel(void *hip_stream = nullptr);
    // Patch the accel after chunks [old_capacity, capacity) were appended and / or the rows of `changed_nodes`
    // (device (chunk, child) pairs) were rewritten (mnv_accel_refresh).
    void refresh_accel(int old_capacity, const int32_t *changed_nodes, int n_changed, void *hip_stream = nullptr);
    // Copy the first `capacity` chunks of the device arrays back into the host vectors (after refinement).
    void copy_from_device(void *hip_stream = nullptr);

    // Spatial branching factor. Only 2 is supported on the device.
    int N = 0;
    // Halfs per voxel row
    int data_dim = 0;
    DataFormat data_format;
    // Scaling / translation of world coordinates into the unit cube
    std::array<float, 3> scale{{1.f, 1.f, 1.f}};
    std::array<float, 3> offset{{0.f, 0.f, 0.f}};

    int64_t pack_index(int nd, int i, int j, int k);
    std::tuple<int, int, int, int> unpack_index(int64_t packed);

    // Host arrays, reference layout
    std::vector<uint16_t> data;          // [capacity][N^3][data_dim] binary16
    std::vector<int32_t> child;          // [capacity][N^3] relative offsets, 0 = leaf
    std::vector<int32_t> parent;         // [capacity]
    std::vector<int16_t> sample_counts;  // [capacity][N^3], initialised to 8

    // Number of chunks in the tree
    int capacity = 0;

    struct Device {
        uint16_t *data = nullptr;
        int32_t *child = nullptr;
        int32_t *parent = nullptr;
        int16_t *sample_counts = nullptr;
        long max_capacity = 0;
        mnv_accel *accel = nullptr;
    } device;

    mnv_tree_view host_view() const;
    mnv_tree_view device_view() const;  // pointers are null before move_to_device
    bool on_device() const { return device.data != nullptr; }

    // Write the tree in the svox .npz layout the loader reads (plain `data` form).
    void save_npz(const std::string &path) const;

    // Adopt arrays (copies); validates sizes like load_npz.
    void assign(const mnv_tree_view &host_view);
    // Same, taking ownership of the vectors (no copy); `meta` supplies N, data_dim, format, scale, offset.
    void adopt(const mnv_tree_view &meta, std::vector<uint16_t> &&data, std::vector<int32_t> &&child,
               std::vector<int32_t> &&parent);

private:
    int N2_ = 0, N3_ = 0;
    void check_sizes();
};

}  // namespace viewer
