// volume_renderer.hpp -- viewer::VolumeRenderer for offline batch rendering.
//
// Same role and member names as the reference facade (reference include/renderer/renderer.hpp:9-39,
// src/renderer/cuda_renderer.cpp:68-163,383-516): it owns a Camera and RenderOptions, `set()` moves a
// tree to the device, `resize()` sizes the frame, `render()` launches the march for the current
// camera.  The GL framebuffers / CUDA-GL interop / blit of the reference are replaced by a linear
// device frame (float RGBA + RGBA8) that `download()` copies to the host.  Guided sampling and
// refinement (load_model, expand_voxels, prune_tree) are not part of this path.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "camera.hpp"
#include "n3tree.hpp"
#include "render_options.hpp"

namespace viewer {

struct VolumeRenderer {
    explicit VolumeRenderer();
    ~VolumeRenderer();

    // Render the currently set tree with `camera` and `options` (asynchronous on the internal stream).
    void render();
    // Set volumetric data to render: tree.move_to_device(max_tree_capacity, true, true) and
    // basis_minmax = {0, max(basis_dim - 1, 0)} as cuda_renderer.cpp:498-516.
    void set(N3Tree &tree, long max_tree_capacity);
    // Clear the volumetric data
    void clear();
    // Resize the frame.  The first call only sets the size; later calls rescale the intrinsics
    // (fx, fy, cx, cy each scaled once -- the reference scales cy twice, cuda_renderer.cpp:398,404-406).
    void resize(int width, int height);
    // Name identifying the renderer backend
    const char *get_backend();

    // Wait for the last render() and copy the frame to the host ([height][width][4]).
    void download(std::vector<float> *rgba, std::vector<uint8_t> *rgba8);
    // Device pointers of the current frame (valid until the next resize()).
    const float *device_rgba() const;
    const uint8_t *device_rgba8() const;
    // Average device time per render() since the last call, in ms (HIP events).
    double take_average_ms();

    // Camera instance
    Camera camera;
    // Rendering options
    RenderOptions options;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace viewer
