#include "n3tree.hpp"

#include <algorithm>

#include <hip/hip_runtime_api.h>

#include <cassert>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>

#include "npz.hpp"

namespace viewer {

N3Tree::N3Tree() {}
N3Tree::N3Tree(const std::string &path) { open(path); }
N3Tree::~N3Tree() { free_device(); }

namespace {
const npz::Array &need(const npz::Archive &a, const char *key) {
    auto it = a.find(key);
    if (it == a.end()) throw std::runtime_error(std::string("npz lacks required array '") + key + "'");
    return it->second;
}
}  // namespace

void N3Tree::open(const std::string &path) {
    if (!(path.size() > 3 && path.compare(path.size() - 4, 4, ".npz") == 0))
        throw std::runtime_error("N3Tree::open expects a .npz path: " + path);
    if (!std::ifstream(path)) {
        std::printf("Can't load because file does not exist: %s\n", path.c_str());
        return;  // N stays 0: the renderer draws nothing (renderer_kernel.cu:266)
    }
    const npz::Archive z = npz::load(path);

    // keys and dtypes as read by n3tree.cpp:28-107
    const npz::Array &dd = need(z, "data_dim");
    if (dd.word_size != 8) throw std::runtime_error("data_dim must be int64");
    data_dim = (int)*dd.data<int64_t>();

    const npz::Array &df = need(z, "data_format");
    std::string fmt;
    if (df.kind == 'U') {
        for (size_t i = 0; i + 3 < df.bytes.size(); i += 4) {  // UTF-32LE -> ASCII (n3tree.cpp:33-37)
            if (df.bytes[i]) fmt.push_back((char)df.bytes[i]);
        }
    } else {
        for (uint8_t c : df.bytes) if (c) fmt.push_back((char)c);
    }
    data_format.parse(fmt);
    std::cout << "Data format " << data_format.to_string() << std::endl;

    if (z.count("invradius3")) {
        const npz::Array &s = need(z, "invradius3");
        if (s.word_size != 4 || s.num_vals() < 3) throw std::runtime_error("invradius3 must be float32[3]");
        for (int i = 0; i < 3; ++i) scale[i] = s.data<float>()[i];
    } else {
        const npz::Array &s = need(z, "invradius");
        if (s.word_size != 8) throw std::runtime_error("invradius must be float64");
        scale[0] = scale[1] = scale[2] = (float)*s.data<double>();
    }
    const npz::Array &off = need(z, "offset");
    if (off.word_size != 4 || off.num_vals() < 3) throw std::runtime_error("offset must be float32[3]");
    for (int i = 0; i < 3; ++i) offset[i] = off.data<float>()[i];
    std::cout << "Scale: " << scale[0] << " " << scale[1] << " " << scale[2] << std::endl;
    std::cout << "Offset: " << offset[0] << " " << offset[1] << " " << offset[2] << std::endl;

    const npz::Array &ch = need(z, "child");
    if (ch.shape.size() < 2 || ch.word_size != 4) throw std::runtime_error("child must be int32 [cap,N,N,N]");
    N = (int)ch.shape[1];
    if (N != 2) std::cout << "WARNING: N != 2 probably doesn't work." << std::endl;
    N2_ = N * N;
    N3_ = N * N * N;
    const size_t cap = ch.shape[0];
    if (ch.num_vals() != cap * (size_t)N3_) throw std::runtime_error("child has unexpected shape");
    child.assign(ch.data<int32_t>(), ch.data<int32_t>() + cap * N3_);
    // the device kernels follow these links without bounds checks (as the reference's do): refuse a file whose child
    // offsets leave the tree instead of faulting on the GPU later
    for (size_t v = 0; v < child.size(); ++v) {
        if (child[v] == 0) continue;
        const int64_t target = (int64_t)(v / N3_) + child[v];
        if (target <= 0 || target >= (int64_t)cap) throw std::runtime_error("child offset points outside the tree");
    }

    const npz::Array &pd = need(z, "parent_depth");
    if (pd.word_size != 4 || pd.shape.size() != 2 || pd.shape[1] != 2) throw std::runtime_error("parent_depth must be int32 [cap,2]");
    parent.resize(pd.shape[0]);
    for (size_t i = 0; i < pd.shape[0]; ++i) parent[i] = pd.data<int32_t>()[i * 2];

    if (z.count("quant_colors")) {
        // VQ-compressed PlenOctree.  The reference's decode loop (n3tree.cpp:109-175) drops the
        // "+ basis" term and indexes the codebook by absolute basis number; this implements the
        // evident intent instead: rows are [channel][basis], bases 0..n_retain-1 come from
        // data_retained [n_retain][cap][N^3][3], the rest from quant_colors [n_q][65536][3]
        // through quant_map [n_q][cap][N^3].  Reference parity for this branch is UNPINNED.
        std::cout << "Decoding quantized colors" << std::endl;
        const npz::Array &qc = need(z, "quant_colors");
        if (qc.word_size != 2) throw std::runtime_error("codebook must be stored in half precision");
        const npz::Array &qm = need(z, "quant_map");
        if (qm.word_size != 2 || qm.shape.size() < 2) throw std::runtime_error("quant_map must be uint16 [n_basis,cap,N,N,N]");
        const size_t n_q = qm.shape[0];
        if (qc.shape.empty() || qc.shape[0] != n_q) throw std::runtime_error("codebook and map basis numbers does not match");
        const size_t n_retain = z.count("data_retained") ? need(z, "data_retained").shape[0] : 0;
        const size_t n_basis = n_q + n_retain;
        const size_t qcap = qm.shape[1];
        if ((size_t)data_dim < 3 * n_basis + 1) throw std::runtime_error("data_dim too small for the quantized bases");
        if (qm.num_vals() != n_q * qcap * N3_ || qc.num_vals() != n_q * 65536 * 3) throw std::runtime_error("quantized arrays have unexpected shapes");
        data.assign(qcap * N3_ * (size_t)data_dim, 0);
        const uint16_t *map = qm.data<uint16_t>();
        const uint16_t *book = qc.data<uint16_t>();
        for (size_t b = 0; b < n_q; ++b) {
            for (size_t v = 0; v < qcap * N3_; ++v) {
                const uint16_t id = map[b * qcap * N3_ + v];
                const uint16_t *col = book + (b * 65536 + id) * 3;
                for (int c = 0; c < 3; ++c) data[v * data_dim + c * n_basis + n_retain + b] = col[c];
            }
        }
        if (n_retain) {
            const npz::Array &rt = need(z, "data_retained");
            if (rt.word_size != 2 || rt.num_vals() != n_retain * qcap * N3_ * 3) throw std::runtime_error("data_retained has unexpected shape");
            const uint16_t *r = rt.data<uint16_t>();
            for (size_t b = 0; b < n_retain; ++b)
                for (size_t v = 0; v < qcap * N3_; ++v)
                    for (int c = 0; c < 3; ++c) data[v * data_dim + c * n_basis + b] = r[(b * qcap * N3_ + v) * 3 + c];
        }
        const npz::Array &sg = need(z, "sigma");
        if (sg.word_size != 2 || sg.num_vals() != qcap * N3_) throw std::runtime_error("sigma must be float16 [cap,N,N,N]");
        for (size_t v = 0; v < qcap * N3_; ++v) data[v * data_dim + data_dim - 1] = sg.data<uint16_t>()[v];
    } else {
        const npz::Array &d = need(z, "data");
        if (d.word_size != 2) throw std::runtime_error("data must be stored in half precision");
        const size_t dcap = d.shape.empty() ? 0 : d.shape[0];
        if (d.num_vals() != dcap * N3_ * (size_t)data_dim) throw std::runtime_error("data has unexpected shape");
        data.assign(d.data<uint16_t>(), d.data<uint16_t>() + d.num_vals());
    }
    check_sizes();
    std::cout << "Data size: " << capacity << std::endl;
}

void N3Tree::check_sizes() {
    if (N3_ == 0 || data_dim <= 0) throw std::runtime_error("tree has no shape");
    const size_t cap = data.size() / ((size_t)N3_ * data_dim);
    sample_counts.assign(cap * N3_, 8);  // n3tree.cpp:191-193
    if (cap != parent.size()) throw std::runtime_error("data and parent sizes not aligned");
    if (cap * N3_ != child.size()) throw std::runtime_error("data and child sizes not aligned");
    capacity = (int)cap;
}

void N3Tree::assign(const mnv_tree_view &v) {
    if (v.N <= 0 || v.capacity < 0 || v.data_dim <= 0 || !v.data || !v.child) throw std::runtime_error("invalid host tree view");
    free_device();
    N = v.N;
    N2_ = N * N;
    N3_ = N * N * N;
    data_dim = v.data_dim;
    data_format.format = v.format == MNV_FORMAT_SH ? DataFormat::SH : DataFormat::RGBA;
    data_format.basis_dim = v.basis_dim;
    for (int i = 0; i < 3; ++i) { scale[i] = v.scale[i]; offset[i] = v.offset[i]; }
    const size_t cap = (size_t)v.capacity;
    data.assign(v.data, v.data + cap * N3_ * data_dim);
    child.assign(v.child, v.child + cap * N3_);
    if (v.parent) parent.assign(v.parent, v.parent + cap);
    else {
        parent.assign(cap, -1);
        for (size_t c = 0; c < cap; ++c)
            for (int j = 0; j < N3_; ++j)
                if (child[c * N3_ + j] != 0) parent[c + child[c * N3_ + j]] = (int32_t)(c * N3_ + j);
    }
    check_sizes();
    if (v.sample_counts) sample_counts.assign(v.sample_counts, v.sample_counts + cap * N3_);
}

void N3Tree::adopt(const mnv_tree_view &meta, std::vector<uint16_t> &&data_in, std::vector<int32_t> &&child_in,
                   std::vector<int32_t> &&parent_in) {
    free_device();
    N = meta.N;
    N2_ = N * N;
    N3_ = N * N * N;
    data_dim = meta.data_dim;
    data_format.format = meta.format == MNV_FORMAT_SH ? DataFormat::SH : DataFormat::RGBA;
    data_format.basis_dim = meta.basis_dim;
    for (int i = 0; i < 3; ++i) { scale[i] = meta.scale[i]; offset[i] = meta.offset[i]; }
    data = std::move(data_in);
    child = std::move(child_in);
    parent = std::move(parent_in);
    check_sizes();
}

namespace {
void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace

void N3Tree::free_device() {
    if (device.accel) mnv_accel_destroy(device.accel);
    if (device.data) (void)hipFree(device.data);
    if (device.child) (void)hipFree(device.child);
    if (device.parent) (void)hipFree(device.parent);
    if (device.sample_counts) (void)hipFree(device.sample_counts);
    device = Device();
}

void N3Tree::move_to_device(long max_capacity, bool need_parent, bool need_sample_counts, void *hip_stream) {
    if (N <= 0) throw std::runtime_error("move_to_device on an empty tree");
    if (max_capacity < capacity) max_capacity = capacity;
    free_device();
    hipStream_t stream = (hipStream_t)hip_stream;
    const size_t row = (size_t)N3_ * data_dim * sizeof(uint16_t);
    hip_check(hipMalloc((void **)&device.data, (size_t)max_capacity * row), "hipMalloc(data)");
    hip_check(hipMemcpyAsync(device.data, data.data(), (size_t)capacity * row, hipMemcpyHostToDevice, stream), "copy data");
    hip_check(hipMalloc((void **)&device.child, (size_t)max_capacity * N3_ * sizeof(int32_t)), "hipMalloc(child)");
    hip_check(hipMemcpyAsync(device.child, child.data(), (size_t)capacity * N3_ * sizeof(int32_t), hipMemcpyHostToDevice, stream), "copy child");
    if (need_parent) {
        hip_check(hipMalloc((void **)&device.parent, (size_t)max_capacity * sizeof(int32_t)), "hipMalloc(parent)");
        hip_check(hipMemcpyAsync(device.parent, parent.data(), (size_t)capacity * sizeof(int32_t), hipMemcpyHostToDevice, stream), "copy parent");
    }
    if (need_sample_counts) {
        // The reference allocates this array without initialising it (n3tree.cpp:235-241);
        // here the host's 8s are uploaded so that the sample tracker is deterministic.
        hip_check(hipMalloc((void **)&device.sample_counts, (size_t)max_capacity * N3_ * sizeof(int16_t)), "hipMalloc(sample_counts)");
        hip_check(hipMemcpyAsync(device.sample_counts, sample_counts.data(), (size_t)capacity * N3_ * sizeof(int16_t), hipMemcpyHostToDevice, stream), "copy sample_counts");
    }
    device.max_capacity = max_capacity;
    hip_check(hipStreamSynchronize(stream), "upload");
    const mnv_tree_view dv = device_view();
    if (N == 2) {
        const int rc = mnv_accel_create_reserved(&dv, std::max<long>(max_capacity, capacity), hip_stream, &device.accel);
        if (rc != MNV_OK) throw std::runtime_error(std::string("mnv_accel_create: ") + mnv_last_error());
    }
}

void N3Tree::rebuild_accel(void *hip_stream) {
    if (!on_device() || N != 2) return;
    const mnv_tree_view dv = device_view();
    if (device.accel && mnv_accel_rebuild(device.accel, &dv, hip_stream) == MNV_OK) return;  // in place: the reserved arrays are reused
    if (device.accel) mnv_accel_destroy(device.accel);
    device.accel = nullptr;
    const int rc = mnv_accel_create_reserved(&dv, std::max<long>(device.max_capacity, capacity), hip_stream, &device.accel);
    if (rc != MNV_OK) throw std::runtime_error(std::string("mnv_accel_create: ") + mnv_last_error());
}

void N3Tree::refresh_accel(int old_capacity, const int32_t *changed_nodes, int n_changed, void *hip_stream) {
    if (!device.accel) return;
    const mnv_tree_view dv = device_view();
    const int rc = mnv_accel_refresh(device.accel, &dv, old_capacity, changed_nodes, n_changed, hip_stream);
    if (rc != MNV_OK) throw std::runtime_error(std::string("mnv_accel_refresh: ") + mnv_last_error());
}

void N3Tree::copy_from_device(void *hip_stream) {
    if (!on_device()) return;
    hipStream_t stream = (hipStream_t)hip_stream;
    const size_t cap = (size_t)capacity;
    data.resize(cap * N3_ * data_dim);
    child.resize(cap * N3_);
    hip_check(hipMemcpyAsync(data.data(), device.data, data.size() * sizeof(uint16_t), hipMemcpyDeviceToHost, stream), "download data");
    hip_check(hipMemcpyAsync(child.data(), device.child, child.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream), "download child");
    if (device.parent) {
        parent.resize(cap);
        hip_check(hipMemcpyAsync(parent.data(), device.parent, cap * sizeof(int32_t), hipMemcpyDeviceToHost, stream), "download parent");
    }
    if (device.sample_counts) {
        sample_counts.resize(cap * N3_);
        hip_check(hipMemcpyAsync(sample_counts.data(), device.sample_counts, sample_counts.size() * sizeof(int16_t), hipMemcpyDeviceToHost, stream),
                  "download sample_counts");
    }
    hip_check(hipStreamSynchronize(stream), "download tree");
}

mnv_tree_view N3Tree::host_view() const {
    mnv_tree_view v;
    std::memset(&v, 0, sizeof(v));
    v.data = data.data();
    v.child = child.data();
    v.parent = parent.data();
    v.sample_counts = sample_counts.data();
    for (int i = 0; i < 3; ++i) { v.offset[i] = offset[i]; v.scale[i] = scale[i]; }
    v.N = N;
    v.data_dim = data_dim;
    v.format = data_format.format == DataFormat::SH ? MNV_FORMAT_SH : MNV_FORMAT_RGBA;
    v.basis_dim = data_format.basis_dim;
    v.capacity = capacity;
    return v;
}

mnv_tree_view N3Tree::device_view() const {
    mnv_tree_view v = host_view();
    v.data = device.data;
    v.child = device.child;
    v.parent = device.parent;
    v.sample_counts = device.sample_counts;
    return v;
}

void N3Tree::save_npz(const std::string &path) const {
    npz::Writer w(path);
    const int64_t dd = data_dim;
    w.add("data_dim", "<i8", {}, &dd, 8);
    w.add_unicode("data_format", data_format.to_string());
    w.add("invradius3", "<f4", {3}, scale.data(), 12);
    w.add("offset", "<f4", {3}, offset.data(), 12);
    const size_t cap = (size_t)capacity, n = (size_t)N;
    w.add("child", "<i4", {cap, n, n, n}, child.data(), child.size() * 4);
    std::vector<int32_t> pd(cap * 2, 0);
    for (size_t c = 0; c < cap; ++c) {  // depth column: chunk level, root = 0
        pd[c * 2] = parent[c];
        pd[c * 2 + 1] = (c == 0 || parent[c] < 0) ? 0 : pd[(size_t)(parent[c] / N3_) * 2 + 1] + 1;
    }
    w.add("parent_depth", "<i4", {cap, 2}, pd.data(), pd.size() * 4);
    w.add("data", "<f2", {cap, n, n, n, (size_t)data_dim}, data.data(), data.size() * 2);
    w.close();
}

int64_t N3Tree::pack_index(int nd, int i, int j, int k) {
    assert(i < N && j < N && k < N && i >= 0 && j >= 0 && k >= 0);
    return (int64_t)nd * N3_ + i * N2_ + j * N + k;
}

std::tuple<int, int, int, int> N3Tree::unpack_index(int64_t packed) {
    const int k = (int)(packed % N);
    packed /= N;
    const int j = (int)(packed % N);
    packed /= N;
    const int i = (int)(packed % N);
    packed /= N;
    return std::tuple<int, int, int, int>{(int)packed, i, j, k};
}

}  // namespace viewer
