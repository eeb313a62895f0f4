#include "volume_renderer.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <stdexcept>
#include <string>

namespace viewer {

namespace {
void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void mnv_check(int rc, const char *what) {
    if (rc != MNV_OK) throw std::runtime_error(std::string(what) + ": " + mnv_last_error());
}
}  // namespace

struct VolumeRenderer::Impl {
    N3Tree *tree = nullptr;
    hipStream_t stream = nullptr;
    float *rgba = nullptr;
    uint8_t *rgba8 = nullptr;
    int width = 0, height = 0;
    bool initial_resize = true;

    Impl() { hip_check(hipStreamCreate(&stream), "hipStreamCreate"); }
    ~Impl() {
        free_frame();
        if (stream) (void)hipStreamDestroy(stream);
    }
    void free_frame() {
        if (rgba) (void)hipFree(rgba);
        if (rgba8) (void)hipFree(rgba8);
        rgba = nullptr;
        rgba8 = nullptr;
    }
};

VolumeRenderer::VolumeRenderer() : impl_(std::make_unique<Impl>()) {}
VolumeRenderer::~VolumeRenderer() {}

void VolumeRenderer::set(N3Tree &tree, long max_tree_capacity) {
    tree.move_to_device(max_tree_capacity, true, true, impl_->stream);
    impl_->tree = &tree;
    options.basis_minmax[0] = 0;
    options.basis_minmax[1] = std::max(tree.data_format.basis_dim - 1, 0);
}

void VolumeRenderer::clear() { impl_->tree = nullptr; }

void VolumeRenderer::resize(int width, int height) {
    if (impl_->width == width && impl_->height == height && impl_->rgba) return;
    if (!impl_->initial_resize && camera.width > 0 && camera.height > 0) {
        const float wr = (float)width / camera.width, hr = (float)height / camera.height;
        camera.fx *= wr;
        camera.default_fx *= wr;
        camera.fy *= hr;
        camera.default_fy *= hr;
        if (camera.default_cx != -1) camera.cx *= wr;
        if (camera.default_cy != -1) camera.cy *= hr;
    }
    impl_->initial_resize = false;
    camera.width = width;
    camera.height = height;
    if (camera.default_cx == -1) camera.cx = (float)(width / 2);   // "-1 = use width / 2" (main.cpp:496)
    if (camera.default_cy == -1) camera.cy = (float)(height / 2);
    impl_->free_frame();
    impl_->width = width;
    impl_->height = height;
    hip_check(hipMalloc((void **)&impl_->rgba, (size_t)width * height * 4 * sizeof(float)), "hipMalloc(frame)");
    hip_check(hipMalloc((void **)&impl_->rgba8, (size_t)width * height * 4), "hipMalloc(frame8)");
}

void VolumeRenderer::render() {
    if (!impl_->rgba) resize(camera.width, camera.height);
    camera._update();
    const mnv_camera cv = camera.c_abi();
    const mnv_rect full = {0, 0, impl_->width, impl_->height};
    if (impl_->tree == nullptr || impl_->tree->N <= 0) {
        mnv_tree_view empty = {};  // N == 0: background only (renderer_kernel.cu:266)
        mnv_check(mnv_render_voxels(&empty, &cv, options.c_abi(), full, impl_->rgba, impl_->rgba8, nullptr, nullptr, nullptr, 0, impl_->stream),
                  "mnv_render_voxels");
        return;
    }
    if (impl_->tree->device.accel) {
        mnv_check(mnv_render_voxels_accel(impl_->tree->device.accel, &cv, options.c_abi(), full, impl_->rgba, impl_->rgba8, impl_->stream),
                  "mnv_render_voxels_accel");
    } else {
        const mnv_tree_view dv = impl_->tree->device_view();
        mnv_check(mnv_render_voxels(&dv, &cv, options.c_abi(), full, impl_->rgba, impl_->rgba8, nullptr, nullptr, nullptr, 0, impl_->stream),
                  "mnv_render_voxels");
    }
}

const char *VolumeRenderer::get_backend() { return "HIP gfx950"; }

void VolumeRenderer::download(std::vector<float> *rgba, std::vector<uint8_t> *rgba8) {
    const size_t n = (size_t)impl_->width * impl_->height * 4;
    hip_check(hipStreamSynchronize(impl_->stream), "render");
    if (rgba) {
        rgba->resize(n);
        hip_check(hipMemcpy(rgba->data(), impl_->rgba, n * sizeof(float), hipMemcpyDeviceToHost), "download rgba");
    }
    if (rgba8) {
        rgba8->resize(n);
        hip_check(hipMemcpy(rgba8->data(), impl_->rgba8, n, hipMemcpyDeviceToHost), "download rgba8");
    }
}

const float *VolumeRenderer::device_rgba() const { return impl_->rgba; }
const uint8_t *VolumeRenderer::device_rgba8() const { return impl_->rgba8; }

double VolumeRenderer::take_average_ms() {
    double ms = 0.0;
    int32_t n = 0;
    if (mnv_take_timing(&ms, &n) != MNV_OK || n == 0) return 0.0;
    return ms / n;
}

}  // namespace viewer
