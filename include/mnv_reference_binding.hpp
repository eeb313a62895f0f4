// mnv_reference_binding.hpp -- the binding a maintainer of cmusatyalab/mega-nerf-viewer would add to call libmnv.so
// from the reference's own host data model (libtorch tensors, glm camera).  It is compiled only INSIDE a build of the
// reference tree (it includes the reference's headers); this repository builds it in oracle/Makefile.ref against
// /root/reference and runs it on the GPU (tests/test_parity_gpu.py::test_reference_binding_is_a_drop_in), which is the
// proof that the C ABI of include/mnv.h is a drop-in for
//     viewer::render_voxels(N3Tree&, const Camera&, const RenderOptions&, ...)      include/cuda/renderer_kernel.hpp:23-34
// Nothing of the reference is copied here: the functions below only read public members of its structs.
#pragma once

#include <glm/gtc/type_ptr.hpp>

#include <cstring>
#include <stdexcept>

#include "camera.hpp"           // reference include/camera.hpp
#include "n3tree/n3tree.hpp"    // reference include/n3tree/n3tree.hpp
#include "render_options.hpp"   // reference include/render_options.hpp

#include "mnv.h"

namespace viewer {

// was: the implicit conversion N3Tree& -> internal::TreeSpec (include/data_spec.hpp:38-49)
inline mnv_tree_view mnv_view(N3Tree &t) {
    mnv_tree_view v{};
    v.data = reinterpret_cast<const uint16_t *>(t.data.data_ptr<at::Half>());
    v.child = t.child.data_ptr<int32_t>();
    v.parent = (t.parent.defined() && t.parent.is_cuda()) ? t.parent.data_ptr<int32_t>() : nullptr;
    v.sample_counts = (t.sample_counts.defined() && t.sample_counts.is_cuda()) ? t.sample_counts.data_ptr<int16_t>() : nullptr;
    const torch::Tensor off = t.offset.cpu(), sc = t.scale.cpu();  // 3 floats each; a real integration caches these next to the tree
    for (int i = 0; i < 3; ++i) {
        v.offset[i] = off[i].item<float>();
        v.scale[i] = sc[i].item<float>();
    }
    v.N = t.N;
    v.data_dim = t.data_dim;
    v.capacity = t.capacity;
    v.format = t.data_format.format == DataFormat::SH ? MNV_FORMAT_SH : MNV_FORMAT_RGBA;
    v.basis_dim = t.data_format.basis_dim;
    return v;
}

// was: internal::CameraSpec(const Camera&) (include/data_spec.hpp:16-22) + the device copy of the matrix (src/camera.cpp:113-123)
inline mnv_camera mnv_view(const Camera &c) {
    mnv_camera m{};
    m.width = c.width;
    m.height = c.height;
    m.fx = c.fx;
    m.fy = c.fy;
    m.cx = c.cx;
    m.cy = c.cy;
    std::memcpy(m.c2w, glm::value_ptr(c.transform), sizeof(m.c2w));  // column-major right | up | back | center
    return m;
}

static_assert(sizeof(RenderOptions) == sizeof(mnv_render_options), "RenderOptions and mnv_render_options are the same POD");

// Replacement body of viewer::render_voxels (src/cuda/renderer_kernel.cu:396-437) with its ORIGINAL eleven parameters
// (include/cuda/renderer_kernel.hpp:23-34); what changes is the type of the two render targets: a build without CUDA-GL interop has no
// cudaArray_t, so image_arr is a linear RGBA8 image in device memory ([height][width][4], read AND written like the surface) and depth_arr
// a linear float depth image ([height][width]).  offscreen is honoured as the kernel honours it (renderer_kernel.cu:225-234,260-264,
// 277-280): true = composite over opt.background_brightness with t_max = 1e9f, false = over the pixels already in image_arr with
// t_max from depth_arr -- the only way cuda_renderer.cpp:141-142 calls it.
// NOTE for callers that switch on mnv_set_tree_cache(1): this call is then served from a cached re-layout of the arrays (tracker frames
// too), and the refinement loop edits tree.data / tree.child IN PLACE between frames (cuda_renderer.cpp:255-270,306-332,335-381) -- call
// mnv_tree_invalidate(tree.child.data_ptr()) after every such edit, or leave the cache off (the default), or use the accel + mnv_accel_refresh.
inline void render_voxels(N3Tree &tree, const Camera &cam, const RenderOptions &opt, uint8_t *&image_arr /* was cudaArray_t& */,
                          float *&depth_arr /* was cudaArray_t& */, hipStream_t &stream, const torch::Tensor &to_split,
                          const torch::Tensor &to_sample, const torch::Tensor &visited, const bool track_visit, const bool offscreen) {
    const mnv_tree_view tv = mnv_view(tree);
    const mnv_camera cv = mnv_view(cam);
    const mnv_rect full{0, 0, cam.width, cam.height};
    const mnv_frame_inputs in{offscreen ? nullptr : depth_arr, offscreen ? nullptr : image_arr};
    const int rc = mnv_render_voxels_ex(&tv, &cv, reinterpret_cast<const mnv_render_options *>(&opt), full, &in, nullptr, image_arr,
                                        to_split.defined() ? to_split.data_ptr<float>() : nullptr,
                                        to_sample.defined() ? to_sample.data_ptr<float>() : nullptr,
                                        visited.defined() ? visited.data_ptr<int32_t>() : nullptr, track_visit ? 1 : 0, (void *)stream);
    if (rc != MNV_OK) throw std::runtime_error(mnv_last_error());  // instead of cuda_assert's exit()
}

// The same launcher for the offline batch render (north star: no window): linear float RGBA (and / or RGBA8) out, offscreen.
inline void render_voxels(N3Tree &tree, const Camera &cam, const RenderOptions &opt, float *rgba_linear, uint8_t *rgba8_linear,
                          void *stream, float *to_split, float *to_sample, int32_t *visited, bool track_visit) {
    const mnv_tree_view tv = mnv_view(tree);
    const mnv_camera cv = mnv_view(cam);
    const mnv_rect full{0, 0, cam.width, cam.height};
    const int rc = mnv_render_voxels(&tv, &cv, reinterpret_cast<const mnv_render_options *>(&opt), full, rgba_linear, rgba8_linear, to_split,
                                     to_sample, visited, track_visit ? 1 : 0, stream);
    if (rc != MNV_OK) throw std::runtime_error(mnv_last_error());
}

// Replacement body of viewer::get_samples_from_voxels (src/cuda/renderer_kernel.cu:439-485) with its ORIGINAL sixteen parameters
// (include/cuda/renderer_kernel.hpp:36-52); depth_arr is the linear float depth image of render_voxels above, read when offscreen == false
// (renderer_kernel.cu:354-357).  The three cluster-grid tensors are the model attributes of cuda_renderer.cpp:524-539 (host or device).
inline void get_samples_from_voxels(N3Tree &tree, const Camera &cam, const RenderOptions &opt, float *&depth_arr /* was cudaArray_t& */,
                                    hipStream_t &stream, const torch::Tensor &to_split, const torch::Tensor &to_sample, const torch::Tensor &visited,
                                    const bool track_visit, const bool offscreen, const torch::Tensor &num_samples, const torch::Tensor &samples,
                                    const torch::Tensor &cluster_indices, const torch::Tensor &grid_dim, const torch::Tensor &min_position,
                                    const torch::Tensor &range) {
    const mnv_tree_view tv = mnv_view(tree);
    const mnv_camera cv = mnv_view(cam);
    const mnv_rect full{0, 0, cam.width, cam.height};
    const mnv_frame_inputs in{offscreen ? nullptr : depth_arr, nullptr};
    mnv_cluster_grid grid;
    const torch::Tensor gd = grid_dim.to(torch::kCPU, torch::kInt32), mp = min_position.to(torch::kCPU, torch::kFloat32), rg = range.to(torch::kCPU, torch::kFloat32);
    for (int i = 0; i < 2; ++i) grid.grid_dim[i] = gd.data_ptr<int32_t>()[i];
    for (int i = 0; i < 3; ++i) {
        grid.min_position[i] = mp.data_ptr<float>()[i];
        grid.range[i] = rg.data_ptr<float>()[i];
    }
    const int rc = mnv_get_samples_from_voxels_ex(&tv, &cv, reinterpret_cast<const mnv_render_options *>(&opt), full, &in,
                                                  to_split.defined() ? to_split.data_ptr<float>() : nullptr,
                                                  to_sample.defined() ? to_sample.data_ptr<float>() : nullptr,
                                                  visited.defined() ? visited.data_ptr<int32_t>() : nullptr, track_visit ? 1 : 0,
                                                  num_samples.data_ptr<int16_t>(), samples.data_ptr<float>(), (int32_t)samples.size(-1),
                                                  cluster_indices.data_ptr<int16_t>(), &grid, (void *)stream);
    if (rc != MNV_OK) throw std::runtime_error(mnv_last_error());
}

// Replacement body of viewer::render_nerf_results (src/cuda/renderer_kernel.cu:365-394) with its ORIGINAL nine parameters
// (include/cuda/renderer_kernel.hpp:12-21).  image_arr is the linear RGBA8 image of render_voxels above; it is WRITTEN only: the kernel
// starts every pixel at alpha 1 (renderer_kernel.cu:316) and composite_nerf_results never touches out[3], so composite_and_write adds
// (1 - 1) x whatever `offscreen` selects (:224-234) -- the frame does not depend on the flag, which is accepted and has no effect.
inline void render_nerf_results(N3Tree &tree, const Camera &cam, const RenderOptions &opt, uint8_t *&image_arr /* was cudaArray_t& */,
                                hipStream_t &stream, const torch::Tensor &sample_values, const torch::Tensor &z_vals, const torch::Tensor &offsets,
                                const bool offscreen) {
    (void)offscreen;
    const mnv_tree_view tv = mnv_view(tree);
    const mnv_camera cv = mnv_view(cam);
    const mnv_rect full{0, 0, cam.width, cam.height};
    const int rc = mnv_render_nerf_results(&tv, &cv, reinterpret_cast<const mnv_render_options *>(&opt), full, sample_values.data_ptr<float>(),
                                           (int32_t)sample_values.size(1), z_vals.data_ptr<float>(), offsets.data_ptr<int64_t>(), nullptr, image_arr,
                                           (void *)stream);
    if (rc != MNV_OK) throw std::runtime_error(mnv_last_error());
}

// was: the implicit conversion N3Tree& -> internal::TreeSpec for the kernels that WRITE the topology (include/data_spec.hpp:26-28, non-const
// child / parent); capacity is the chunk count before the call, as tree.capacity is when cuda_renderer.cpp:255,358 call in
inline mnv_tree_edit mnv_edit_view(N3Tree &t) {
    mnv_tree_edit e{};
    e.child = t.child.data_ptr<int32_t>();
    e.parent = t.parent.data_ptr<int32_t>();
    const torch::Tensor off = t.offset.cpu(), sc = t.scale.cpu();
    for (int i = 0; i < 3; ++i) {
        e.offset[i] = off[i].item<float>();
        e.scale[i] = sc[i].item<float>();
    }
    e.N = t.N;
    e.capacity = t.capacity;
    return e;
}

// the three cluster-grid tensors the refinement launchers take (model attributes, cuda_renderer.cpp:524-539): host or device
inline mnv_cluster_grid mnv_view(const torch::Tensor &grid_dim, const torch::Tensor &min_position, const torch::Tensor &range) {
    mnv_cluster_grid grid;
    const torch::Tensor gd = grid_dim.to(torch::kCPU, torch::kInt32), mp = min_position.to(torch::kCPU, torch::kFloat32), rg = range.to(torch::kCPU, torch::kFloat32);
    for (int i = 0; i < 2; ++i) grid.grid_dim[i] = gd.data_ptr<int32_t>()[i];
    for (int i = 0; i < 3; ++i) {
        grid.min_position[i] = mp.data_ptr<float>()[i];
        grid.range[i] = rg.data_ptr<float>()[i];
    }
    return grid;
}

// Replacement body of viewer::add_children_and_generate_samples (src/cuda/renderer_kernel.cu:487-510) with its ORIGINAL nine parameters
// (include/cuda/renderer_kernel.hpp:54-63): parent_nodes int32 [n][2], samples float [n*8][samples_per_corner][dim] (uniform numbers in,
// sample rows out), cluster_indices int16 [n*8][samples_per_corner], visited int32 [max_capacity].  Like the reference's launcher it runs on
// the default stream (no stream parameter exists).
inline void add_children_and_generate_samples(N3Tree &tree, const RenderOptions &opt, const torch::Tensor &parent_nodes, const torch::Tensor &samples,
                                              const torch::Tensor &cluster_indices, const torch::Tensor &visited, const torch::Tensor &grid_dim,
                                              const torch::Tensor &min_position, const torch::Tensor &range) {
    const mnv_tree_edit ev = mnv_edit_view(tree);
    const mnv_cluster_grid grid = mnv_view(grid_dim, min_position, range);
    const int rc = mnv_add_children_and_generate_samples(&ev, reinterpret_cast<const mnv_render_options *>(&opt), parent_nodes.data_ptr<int32_t>(),
                                                         (int32_t)parent_nodes.size(0), samples.data_ptr<float>(), (int32_t)samples.size(-1),
                                                         cluster_indices.data_ptr<int16_t>(), visited.data_ptr<int32_t>(), &grid, nullptr);
    if (rc != MNV_OK) throw std::runtime_error(mnv_last_error());
}

// Replacement body of viewer::generate_samples (src/cuda/renderer_kernel.cu:512-534), ORIGINAL eight parameters (renderer_kernel.hpp:65-73)
inline void generate_samples(N3Tree &tree, const RenderOptions &opt, const torch::Tensor &nodes, const torch::Tensor &samples,
                             const torch::Tensor &cluster_indices, const torch::Tensor &grid_dim, const torch::Tensor &min_position,
                             const torch::Tensor &range) {
    const mnv_tree_edit ev = mnv_edit_view(tree);
    const mnv_cluster_grid grid = mnv_view(grid_dim, min_position, range);
    const int rc = mnv_generate_samples(&ev, reinterpret_cast<const mnv_render_options *>(&opt), nodes.data_ptr<int32_t>(), (int32_t)nodes.size(0),
                                        samples.data_ptr<float>(), (int32_t)samples.size(-1), cluster_indices.data_ptr<int16_t>(), &grid, nullptr);
    if (rc != MNV_OK) throw std::runtime_error(mnv_last_error());
}

// Replacement body of viewer::adjust_parents_and_children (src/cuda/renderer_kernel.cu:536-549), ORIGINAL four parameters
// (renderer_kernel.hpp:75-79): to_delete is the bool tensor of cuda_renderer.cpp:337, index_shifts the int32 cumulative sum of :352
inline void adjust_parents_and_children(N3Tree &tree, const int first_shift_index, const torch::Tensor &to_delete, const torch::Tensor &index_shifts) {
    const mnv_tree_edit ev = mnv_edit_view(tree);
    static_assert(sizeof(bool) == 1, "to_delete is read as one byte per chunk");
    const int rc = mnv_adjust_parents_and_children(&ev, first_shift_index, reinterpret_cast<const uint8_t *>(to_delete.data_ptr<bool>()),
                                                   index_shifts.data_ptr<int32_t>(), nullptr);
    if (rc != MNV_OK) throw std::runtime_error(mnv_last_error());
}

// The tuned path: build once where the reference calls tree->move_to_device (cuda_renderer.cpp:498-505) ...
inline mnv_accel *make_accel(N3Tree &tree, long max_tree_capacity, void *stream) {
    const mnv_tree_view tv = mnv_view(tree);
    mnv_accel *accel = nullptr;
    if (mnv_accel_create_reserved(&tv, max_tree_capacity, stream, &accel) != MNV_OK) throw std::runtime_error(mnv_last_error());
    return accel;
}

// ... and render with it (no refinement active); the eleven-parameter shape on the packed layout:
inline void render_voxels(const mnv_accel *accel, const Camera &cam, const RenderOptions &opt, uint8_t *&image_arr, float *&depth_arr,
                          hipStream_t &stream, const bool offscreen) {
    const mnv_camera cv = mnv_view(cam);
    const mnv_rect full{0, 0, cam.width, cam.height};
    const mnv_frame_inputs in{offscreen ? nullptr : depth_arr, offscreen ? nullptr : image_arr};
    if (mnv_render_voxels_accel_ex(accel, &cv, reinterpret_cast<const mnv_render_options *>(&opt), full, &in, nullptr, image_arr, (void *)stream) != MNV_OK)
        throw std::runtime_error(mnv_last_error());
}

// the offline shape
inline void render_voxels(const mnv_accel *accel, const Camera &cam, const RenderOptions &opt, float *rgba_linear, uint8_t *rgba8_linear,
                          void *stream) {
    const mnv_camera cv = mnv_view(cam);
    const mnv_rect full{0, 0, cam.width, cam.height};
    if (mnv_render_voxels_accel(accel, &cv, reinterpret_cast<const mnv_render_options *>(&opt), full, rgba_linear, rgba8_linear, stream) != MNV_OK)
        throw std::runtime_error(mnv_last_error());
}

}  // namespace viewer
