// ref_driver.hip -- driver that runs the REFERENCE's own ray-march device code on the GPU.
//
// TEST INFRASTRUCTURE ONLY (see mnv_oracle.h).  This translation unit is compiled by
// oracle/Makefile.ref against the reference sources where they lie under /root/reference
// (hipify-perl'd into a scratch directory; nothing of them is copied into the repository).
// It includes the reference's include/cuda/rt_core.cuh and calls its
//     viewer::device::render_voxels_trace_ray<float>        (rt_core.cuh:162-332)
// through the reference's own N3Tree loader (src/n3tree/n3tree.cpp + 3rdparty/cnpy), its
// TreeSpec (include/data_spec.hpp:25-50) and its RenderOptions (include/render_options.hpp).
//
// The only restated piece is the body of the per-pixel kernels (render_voxels_kernel, renderer_kernel.cu:243-292, and its
// two guided-sampling siblings): they write through CUDA surface objects (surf2Dwrite), which gfx950 has no hardware for,
// so they cannot be built as they are.  The kernels below follow :272-275, :282-288 and the offscreen branch of
// composite_and_write (:224-229), call the reference's own screen2worlddir (:30-38) and rodrigues (:40-61) on the
// reference's own CameraSpec / Camera (include/data_spec.hpp:9-23, src/camera.cpp), and store the four floats the reference
// holds just before its u8 cast.
//
// screen2worlddir, rodrigues and the three refinement kernels (adjust_parents_and_children_kernel,
// add_children_and_generate_samples_kernel, generate_samples_kernel with generate_samples_inner; renderer_kernel.cu:30-213)
// sit in that same file but touch no surface: oracle/Makefile.ref cuts that contiguous block out of the reference file,
// verbatim, into the scratch directory (refine_kernels.inc) and it is included below, so these run as the reference wrote them.
#include <hip/hip_runtime.h>

#include <glm/gtc/type_ptr.hpp>

#include <memory>

#include "camera.hpp"
#include "cuda/common.cuh"
#include "cuda/rt_core.cuh"
#include "data_spec.hpp"
#include "n3tree/n3tree.hpp"
#include "render_options.hpp"

#include "mnv_reference_binding.hpp"  // include/ of this repository: the reference-side binding to libmnv.so

namespace viewer {
// cuda_assert is defined in the reference's src/cuda/common.cu (compiled alongside)

using internal::CameraSpec;
using internal::TreeSpec;
#include "refine_kernels.inc"  // the reference's own text: renderer_kernel.cu, screen2worlddir .. generate_samples_kernel

namespace {

// The reference's Camera with the given intrinsics and the caller's 12-float camera-to-world matrix (column-major right | up |
// back | center): _update(false) leaves the matrix alone and uploads it (src/camera.cpp:113-123).
std::unique_ptr<Camera> make_camera(int width, int height, float fx, float fy, float cx, float cy, const float *c2w12) {
    auto cam = std::make_unique<Camera>(width, height, fx, fy, cx, cy);
    memcpy(glm::value_ptr(cam->transform), c2w12, 12 * sizeof(float));
    cam->_update(false, true);
    (void)hipDeviceSynchronize();
    return cam;
}

// restated wrapper of render_voxels_kernel (renderer_kernel.cu:243-292) with composite_and_write (:215-241): the two surface reads of
// the offscreen == false branch (the image pixel :260-264, the depth pixel :277-280) come from linear arrays (NULL: offscreen == true)
__global__ void ref_render_voxels_kernel(const internal::TreeSpec tree, const CameraSpec cam, const RenderOptions opt,
                                         float *rgba, const float *tmax_px, const uint8_t *image_rgbx,
                                         torch::PackedTensorAccessor32<float, 2, torch::RestrictPtrTraits> to_split,
                                         torch::PackedTensorAccessor32<float, 2, torch::RestrictPtrTraits> to_sample,
                                         torch::PackedTensorAccessor32<int32_t, 1, torch::RestrictPtrTraits> visited,
                                         const bool track_visit) {
    CUDA_GET_THREAD_ID(idx, cam.width * cam.height);
    const int x = idx % cam.width, y = idx / cam.width;
    float dir[3], cen[3], out[4];
    bool enable_draw = tree.N > 0;
    out[0] = out[1] = out[2] = out[3] = 0.f;
    if (enable_draw) {
        screen2worlddir(x, y, cam, dir, cen);
        for (int i = 0; i < 3; ++i) cen[i] = tree.offset[i] + tree.scale[i] * cen[i];
        float t_max = 1e9f;
        if (tmax_px) t_max = tmax_px[idx];
        float vdir[3] = {dir[0], dir[1], dir[2]};
        float aa[3] = {opt.rot_dirs[0], opt.rot_dirs[1], opt.rot_dirs[2]};
        rodrigues(aa, vdir);
        device::render_voxels_trace_ray(tree, visited, dir, vdir, cen, opt, t_max, out, &to_split[idx][1],
                                        &to_split[idx][2], &to_split[idx][0], &to_sample[idx][1],
                                        &to_sample[idx][2], &to_sample[idx][0], track_visit);
    }
    const float nalpha = 1.f - out[3];
    if (image_rgbx) {
        const uint8_t *rgbx_init = image_rgbx + (size_t)idx * 4;
        out[0] += rgbx_init[0] / 255.f * nalpha;
        out[1] += rgbx_init[1] / 255.f * nalpha;
        out[2] += rgbx_init[2] / 255.f * nalpha;
    } else {
        const float remain = opt.background_brightness * nalpha;
        out[0] += remain;
        out[1] += remain;
        out[2] += remain;
    }
    rgba[idx * 4 + 0] = out[0];
    rgba[idx * 4 + 1] = out[1];
    rgba[idx * 4 + 2] = out[2];
    rgba[idx * 4 + 3] = out[3];
}

// restated wrapper of get_samples_from_voxels_kernel (renderer_kernel.cu:329-363); tmax_px: what the kernel reads from the depth surface when
// offscreen == false (:354-357), a linear [h][w] array here; nullptr = offscreen
__global__ void ref_get_samples_kernel(internal::TreeSpec tree, const CameraSpec cam, const RenderOptions opt, const float *tmax_px,
                                       torch::PackedTensorAccessor32<float, 2, torch::RestrictPtrTraits> to_split,
                                       torch::PackedTensorAccessor32<float, 2, torch::RestrictPtrTraits> to_sample,
                                       torch::PackedTensorAccessor32<int32_t, 1, torch::RestrictPtrTraits> visited,
                                       torch::PackedTensorAccessor32<short, 1, torch::RestrictPtrTraits> num_samples,
                                       torch::PackedTensorAccessor64<float, 3, torch::RestrictPtrTraits> samples,
                                       torch::PackedTensorAccessor64<short, 2, torch::RestrictPtrTraits> cluster_indices,
                                       const torch::PackedTensorAccessor32<int32_t, 1, torch::RestrictPtrTraits> grid_dim,
                                       const torch::PackedTensorAccessor32<float, 1, torch::RestrictPtrTraits> min_position,
                                       const torch::PackedTensorAccessor32<float, 1, torch::RestrictPtrTraits> range) {
    CUDA_GET_THREAD_ID(idx, cam.width * cam.height);
    const int x = idx % cam.width, y = idx / cam.width;
    float dir[3], cen[3];
    screen2worlddir(x, y, cam, dir, cen);
    float vdir[3] = {dir[0], dir[1], dir[2]};
    float aa[3] = {opt.rot_dirs[0], opt.rot_dirs[1], opt.rot_dirs[2]};
    rodrigues(aa, vdir);
    float t_max = 1e9f;
    if (tmax_px) t_max = tmax_px[idx];
    device::get_samples_trace_ray(tree, visited, dir, vdir, cen, opt, t_max, &to_split[idx][1], &to_split[idx][2],
                                  &to_split[idx][0], &to_sample[idx][1], &to_sample[idx][2], &to_sample[idx][0], false,
                                  &num_samples[idx], samples, cluster_indices, idx, grid_dim, min_position, range);
}

// restated wrapper of render_nerf_results_kernel (renderer_kernel.cu:294-327), offscreen composite
__global__ void ref_render_nerf_kernel(const internal::TreeSpec tree, const CameraSpec cam, const RenderOptions opt, float *rgba,
                                       const torch::PackedTensorAccessor64<float, 2, torch::RestrictPtrTraits> sample_values,
                                       const torch::PackedTensorAccessor64<float, 1, torch::RestrictPtrTraits> z_vals,
                                       const torch::PackedTensorAccessor32<int64_t, 1, torch::RestrictPtrTraits> offsets) {
    CUDA_GET_THREAD_ID(idx, cam.width * cam.height);
    const int x = idx % cam.width, y = idx / cam.width;
    float dir[3], cen[3], out[4];
    out[0] = out[1] = out[2] = 0.f;
    out[3] = 1.0f;
    screen2worlddir(x, y, cam, dir, cen);
    float vdir[3] = {dir[0], dir[1], dir[2]};
    float aa[3] = {opt.rot_dirs[0], opt.rot_dirs[1], opt.rot_dirs[2]};
    rodrigues(aa, vdir);
    device::composite_nerf_results(tree, vdir, opt, (idx == 0 ? 0 : offsets[idx - 1]), offsets[idx], sample_values, z_vals, out);
    const float nalpha = 1.f - out[3];
    const float remain = opt.background_brightness * nalpha;
    rgba[idx * 4 + 0] = out[0] + remain;
    rgba[idx * 4 + 1] = out[1] + remain;
    rgba[idx * 4 + 2] = out[2] + remain;
    rgba[idx * 4 + 3] = out[3];
}

}  // namespace
}  // namespace viewer

extern "C" {

// Opens `npz_path` with the reference's N3Tree::open, moves it to the device with the reference's
// move_to_device, renders the full frame and copies float RGBA [h][w][4] to `rgba_host`.
// `opt_bytes` is a reference-layout RenderOptions (render_options.hpp:9-56).  Also returns the first
// `n_probe` elements of the loaded data / child / parent arrays for loader parity checks.
int ref_render_npz(const char *npz_path, int width, int height, float fx, float fy, float cx, float cy,
                   const float *c2w12, const void *opt_bytes, int opt_size, float *rgba_host,
                   uint16_t *data_probe, int32_t *child_probe, int32_t *parent_probe, int n_probe,
                   int *meta /* N, data_dim, format, basis_dim, capacity */) {
    using namespace viewer;
    if (opt_size != (int)sizeof(RenderOptions)) return -2;
    RenderOptions opt;
    memcpy(&opt, opt_bytes, sizeof(opt));
    try {
        N3Tree tree;
        tree.open(npz_path);
        if (tree.N == 0) return -3;
        meta[0] = tree.N;
        meta[1] = tree.data_dim;
        meta[2] = (int)tree.data_format.format;
        meta[3] = tree.data_format.basis_dim;
        meta[4] = tree.capacity;
        if (n_probe > 0) {
            auto d = tree.data.flatten();
            auto c = tree.child.flatten();
            auto p = tree.parent.flatten();
            const int64_t nd = std::min<int64_t>(n_probe, d.numel()), nc = std::min<int64_t>(n_probe, c.numel()), np = std::min<int64_t>(n_probe, p.numel());
            memcpy(data_probe, d.data_ptr(), nd * 2);
            memcpy(child_probe, c.data_ptr(), nc * 4);
            memcpy(parent_probe, p.data_ptr(), np * 4);
        }
        tree.move_to_device(tree.capacity, true, true);
        tree.sample_counts.fill_(8);  // the reference leaves this array uninitialised (n3tree.cpp:235-241)
        auto camera = make_camera(width, height, fx, fy, cx, cy, c2w12);
        const CameraSpec cam(*camera);
        const int64_t n = (int64_t)width * height;
        auto fopt = torch::TensorOptions().device(torch::kCUDA).dtype(torch::kFloat32);
        torch::Tensor out = torch::zeros({n, 4}, fopt);
        torch::Tensor to_split = torch::full({n, 3}, -1.f, fopt), to_sample = torch::full({n, 3}, -1.f, fopt);
        torch::Tensor visited = torch::zeros({tree.capacity}, torch::TensorOptions().device(torch::kCUDA).dtype(torch::kInt32));
        const int threads = 512;  // auto_cuda_threads() picks 512 or 1024 (renderer_kernel.cu:14-28)
        const int blocks = N_BLOCKS_NEEDED(n, threads);
        hipLaunchKernelGGL(ref_render_voxels_kernel, dim3(blocks), dim3(threads), 0, 0, viewer::internal::TreeSpec(tree), cam, opt,
                           out.data_ptr<float>(), (const float *)nullptr, (const uint8_t *)nullptr, to_split.packed_accessor32<float, 2, torch::RestrictPtrTraits>(),
                           to_sample.packed_accessor32<float, 2, torch::RestrictPtrTraits>(),
                           visited.packed_accessor32<int32_t, 1, torch::RestrictPtrTraits>(), false);
        if (hipDeviceSynchronize() != hipSuccess) return -4;
        torch::Tensor h = out.cpu();
        memcpy(rgba_host, h.data_ptr<float>(), n * 4 * sizeof(float));
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_render_npz: %s\n", e.what());
        return -1;
    }
    return 0;
}

// The reference's march in its live call shape, offscreen == false (cuda_renderer.cpp:141-142): tmax_host [h][w] float = the depth
// attachment, image_host [h][w][4] uint8 = the image the volume is composited over (either may be NULL = that half offscreen).
int ref_render_onscreen_npz(const char *npz_path, int width, int height, float fx, float fy, float cx, float cy, const float *c2w12,
                            const void *opt_bytes, int opt_size, const float *tmax_host, const uint8_t *image_host, float *rgba_host) {
    using namespace viewer;
    if (opt_size != (int)sizeof(RenderOptions)) return -2;
    RenderOptions opt;
    memcpy(&opt, opt_bytes, sizeof(opt));
    try {
        N3Tree tree;
        tree.open(npz_path);
        if (tree.N == 0) return -3;
        tree.move_to_device(tree.capacity, true, true);
        tree.sample_counts.fill_(8);
        auto camera = make_camera(width, height, fx, fy, cx, cy, c2w12);
        const CameraSpec cam(*camera);
        const int64_t n = (int64_t)width * height;
        auto fopt = torch::TensorOptions().device(torch::kCUDA).dtype(torch::kFloat32);
        torch::Tensor out = torch::zeros({n, 4}, fopt);
        torch::Tensor to_split = torch::full({n, 3}, -1.f, fopt), to_sample = torch::full({n, 3}, -1.f, fopt);
        torch::Tensor visited = torch::zeros({tree.capacity}, torch::TensorOptions().device(torch::kCUDA).dtype(torch::kInt32));
        torch::Tensor tmax, image;
        if (tmax_host) tmax = torch::from_blob((void *)tmax_host, {n}, torch::kFloat32).clone().to(torch::kCUDA);
        if (image_host) image = torch::from_blob((void *)image_host, {n, 4}, torch::kUInt8).clone().to(torch::kCUDA);
        const int threads = 512;
        const int blocks = N_BLOCKS_NEEDED(n, threads);
        hipLaunchKernelGGL(ref_render_voxels_kernel, dim3(blocks), dim3(threads), 0, 0, viewer::internal::TreeSpec(tree), cam, opt,
                           out.data_ptr<float>(), tmax_host ? (const float *)tmax.data_ptr<float>() : (const float *)nullptr,
                           image_host ? (const uint8_t *)image.data_ptr<uint8_t>() : (const uint8_t *)nullptr,
                           to_split.packed_accessor32<float, 2, torch::RestrictPtrTraits>(),
                           to_sample.packed_accessor32<float, 2, torch::RestrictPtrTraits>(),
                           visited.packed_accessor32<int32_t, 1, torch::RestrictPtrTraits>(), false);
        if (hipDeviceSynchronize() != hipSuccess) return -4;
        torch::Tensor h = out.cpu();
        memcpy(rgba_host, h.data_ptr<float>(), n * 4 * sizeof(float));
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_render_onscreen_npz: %s\n", e.what());
        return -1;
    }
    return 0;
}

// The reference's render_voxels_trace_ray with its refinement trackers and visit marks (rt_core.cuh:132-134,179-180,237-252,
// 308-321) returned to the host.  sample_counts_host: [capacity][8] int16 uploaded into the tree's device array, or NULL for
// all 8 (the reference leaves the device array uninitialised, n3tree.cpp:235-241).
int ref_render_track_onscreen_npz(const char *npz_path, int width, int height, float fx, float fy, float cx, float cy, const float *c2w12,
                                  const void *opt_bytes, int opt_size, const int16_t *sample_counts_host, int track_visit, const float *tmax_host,
                                  const uint8_t *image_host, float *rgba_host, float *split_host, float *sample_host, int32_t *visited_host);

int ref_render_track_npz(const char *npz_path, int width, int height, float fx, float fy, float cx, float cy, const float *c2w12,
                         const void *opt_bytes, int opt_size, const int16_t *sample_counts_host, int track_visit, float *rgba_host,
                         float *split_host, float *sample_host, int32_t *visited_host) {
    return ref_render_track_onscreen_npz(npz_path, width, height, fx, fy, cx, cy, c2w12, opt_bytes, opt_size, sample_counts_host, track_visit, nullptr,
                                         nullptr, rgba_host, split_host, sample_host, visited_host);
}

// ... in the call shape of the render loop (cuda_renderer.cpp:141-142): trackers and marks AND offscreen == false -- tmax_host [h][w] and
// image_host [h][w][4] stand for the two surfaces the kernel reads (either may be NULL)
int ref_render_track_onscreen_npz(const char *npz_path, int width, int height, float fx, float fy, float cx, float cy, const float *c2w12,
                                  const void *opt_bytes, int opt_size, const int16_t *sample_counts_host, int track_visit, const float *tmax_host,
                                  const uint8_t *image_host, float *rgba_host, float *split_host, float *sample_host, int32_t *visited_host) {
    using namespace viewer;
    if (opt_size != (int)sizeof(RenderOptions)) return -2;
    RenderOptions opt;
    memcpy(&opt, opt_bytes, sizeof(opt));
    try {
        N3Tree tree;
        tree.open(npz_path);
        if (tree.N == 0) return -3;
        tree.move_to_device(tree.capacity, true, true);
        if (sample_counts_host) {
            torch::Tensor sc = torch::from_blob((void *)sample_counts_host, {(int64_t)tree.capacity, 8}, torch::kInt16).clone();
            tree.sample_counts.copy_(sc);
        } else {
            tree.sample_counts.fill_(8);
        }
        auto camera = make_camera(width, height, fx, fy, cx, cy, c2w12);
        const CameraSpec cam(*camera);
        const int64_t n = (int64_t)width * height;
        auto fopt = torch::TensorOptions().device(torch::kCUDA).dtype(torch::kFloat32);
        torch::Tensor out = torch::zeros({n, 4}, fopt);
        torch::Tensor to_split = torch::full({n, 3}, -1.f, fopt), to_sample = torch::full({n, 3}, -1.f, fopt);  // cuda_renderer.cpp:97-98
        torch::Tensor visited = torch::zeros({tree.capacity}, torch::TensorOptions().device(torch::kCUDA).dtype(torch::kInt32));
        torch::Tensor tmax, image;
        if (tmax_host) tmax = torch::from_blob((void *)tmax_host, {n}, torch::kFloat32).clone().to(torch::kCUDA);
        if (image_host) image = torch::from_blob((void *)image_host, {n, 4}, torch::kUInt8).clone().to(torch::kCUDA);
        const int threads = 512;
        const int blocks = N_BLOCKS_NEEDED(n, threads);
        hipLaunchKernelGGL(ref_render_voxels_kernel, dim3(blocks), dim3(threads), 0, 0, viewer::internal::TreeSpec(tree), cam, opt,
                           out.data_ptr<float>(), tmax_host ? (const float *)tmax.data_ptr<float>() : (const float *)nullptr,
                           image_host ? (const uint8_t *)image.data_ptr<uint8_t>() : (const uint8_t *)nullptr,
                           to_split.packed_accessor32<float, 2, torch::RestrictPtrTraits>(),
                           to_sample.packed_accessor32<float, 2, torch::RestrictPtrTraits>(),
                           visited.packed_accessor32<int32_t, 1, torch::RestrictPtrTraits>(), track_visit != 0);
        if (hipDeviceSynchronize() != hipSuccess) return -4;
        memcpy(rgba_host, out.cpu().data_ptr<float>(), n * 4 * sizeof(float));
        memcpy(split_host, to_split.cpu().data_ptr<float>(), n * 3 * sizeof(float));
        memcpy(sample_host, to_sample.cpu().data_ptr<float>(), n * 3 * sizeof(float));
        memcpy(visited_host, visited.cpu().data_ptr<int32_t>(), (size_t)tree.capacity * sizeof(int32_t));
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_render_track_npz: %s\n", e.what());
        return -1;
    }
    return 0;
}

// The reference's add_children_and_generate_samples_kernel (renderer_kernel.cu:170-198, launched as :487-510) on the tree in
// `npz_path` moved to the device with room for max_capacity chunks.  samples: in = the caller's uniform numbers, out = sample
// rows; child_out [max_capacity][8], parent_out [max_capacity], visited in/out [max_capacity].
int ref_add_children_dropin_npz(const char *npz_path, const void *opt_bytes, int opt_size, int max_capacity, const int32_t *parent_nodes,
                                int num_parents, float *samples, int samples_dim, int16_t *clusters, int32_t *visited, const int32_t *grid_dim2,
                                const float *min_position3, const float *range3, int32_t *child_out, int32_t *parent_out, int dropin);

int ref_add_children_npz(const char *npz_path, const void *opt_bytes, int opt_size, int max_capacity, const int32_t *parent_nodes,
                         int num_parents, float *samples, int samples_dim, int16_t *clusters, int32_t *visited,
                         const int32_t *grid_dim2, const float *min_position3, const float *range3, int32_t *child_out,
                         int32_t *parent_out) {
    return ref_add_children_dropin_npz(npz_path, opt_bytes, opt_size, max_capacity, parent_nodes, num_parents, samples, samples_dim, clusters, visited,
                                       grid_dim2, min_position3, range3, child_out, parent_out, 0);
}

// dropin != 0: not the reference's kernel but libmnv.so through the nine-parameter viewer::add_children_and_generate_samples of
// include/mnv_reference_binding.hpp, on the reference's own N3Tree and the same tensors -- the same arrays must come back
int ref_add_children_dropin_npz(const char *npz_path, const void *opt_bytes, int opt_size, int max_capacity, const int32_t *parent_nodes,
                                int num_parents, float *samples, int samples_dim, int16_t *clusters, int32_t *visited, const int32_t *grid_dim2,
                                const float *min_position3, const float *range3, int32_t *child_out, int32_t *parent_out, int dropin) {
    using namespace viewer;
    if (opt_size != (int)sizeof(RenderOptions)) return -2;
    RenderOptions opt;
    memcpy(&opt, opt_bytes, sizeof(opt));
    try {
        N3Tree tree;
        tree.open(npz_path);
        if (tree.N == 0) return -3;
        tree.move_to_device(max_capacity, true, true);
        auto dev = torch::kCUDA;
        const int64_t rows = (int64_t)num_parents * 8, spc = opt.samples_per_corner;
        torch::Tensor pn = torch::from_blob((void *)parent_nodes, {(int64_t)num_parents, 2}, torch::kInt32).clone().to(dev);
        torch::Tensor smp = torch::from_blob((void *)samples, {rows, spc, (int64_t)samples_dim}, torch::kFloat32).clone().to(dev);
        torch::Tensor cl = torch::full({rows, spc}, -1, torch::TensorOptions().device(dev).dtype(torch::kInt16));
        torch::Tensor vis = torch::from_blob((void *)visited, {(int64_t)max_capacity}, torch::kInt32).clone().to(dev);
        torch::Tensor gd = torch::from_blob((void *)grid_dim2, {2}, torch::kInt32).clone().to(dev);
        torch::Tensor mp = torch::from_blob((void *)min_position3, {3}, torch::kFloat32).clone().to(dev);
        torch::Tensor rg = torch::from_blob((void *)range3, {3}, torch::kFloat32).clone().to(dev);
        const int threads = 512, blocks = N_BLOCKS_NEEDED(rows, threads);
        if (dropin) {
            if (hipDeviceSynchronize() != hipSuccess) return -4;
            add_children_and_generate_samples(tree, opt, pn, smp, cl, vis, gd, mp, rg);
        } else
        hipLaunchKernelGGL(add_children_and_generate_samples_kernel, dim3(blocks), dim3(threads), 0, 0, viewer::internal::TreeSpec(tree), opt,
                           pn.packed_accessor32<int32_t, 2, torch::RestrictPtrTraits>(), smp.packed_accessor32<float, 3, torch::RestrictPtrTraits>(),
                           cl.packed_accessor32<short, 2, torch::RestrictPtrTraits>(), vis.packed_accessor32<int32_t, 1, torch::RestrictPtrTraits>(),
                           gd.packed_accessor32<int32_t, 1, torch::RestrictPtrTraits>(), mp.packed_accessor32<float, 1, torch::RestrictPtrTraits>(),
                           rg.packed_accessor32<float, 1, torch::RestrictPtrTraits>(), num_parents);
        if (hipDeviceSynchronize() != hipSuccess) return -4;
        memcpy(samples, smp.cpu().data_ptr(), rows * spc * samples_dim * 4);
        memcpy(clusters, cl.cpu().data_ptr(), rows * spc * 2);
        memcpy(visited, vis.cpu().data_ptr(), (size_t)max_capacity * 4);
        memcpy(child_out, tree.child.cpu().contiguous().data_ptr(), (size_t)max_capacity * 8 * 4);
        memcpy(parent_out, tree.parent.cpu().contiguous().data_ptr(), (size_t)max_capacity * 4);
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_add_children_npz: %s\n", e.what());
        return -1;
    }
    return 0;
}

// The reference's generate_samples_kernel (renderer_kernel.cu:200-213, launched as :512-534) for existing voxels nodes[i] = (chunk, child).
int ref_generate_samples_dropin_npz(const char *npz_path, const void *opt_bytes, int opt_size, const int32_t *nodes, int num_items, float *samples,
                                    int samples_dim, int16_t *clusters, const int32_t *grid_dim2, const float *min_position3, const float *range3,
                                    int dropin);

int ref_generate_samples_npz(const char *npz_path, const void *opt_bytes, int opt_size, const int32_t *nodes, int num_items, float *samples,
                             int samples_dim, int16_t *clusters, const int32_t *grid_dim2, const float *min_position3, const float *range3) {
    return ref_generate_samples_dropin_npz(npz_path, opt_bytes, opt_size, nodes, num_items, samples, samples_dim, clusters, grid_dim2, min_position3,
                                           range3, 0);
}

// dropin != 0: libmnv.so through the eight-parameter viewer::generate_samples of include/mnv_reference_binding.hpp
int ref_generate_samples_dropin_npz(const char *npz_path, const void *opt_bytes, int opt_size, const int32_t *nodes, int num_items, float *samples,
                                    int samples_dim, int16_t *clusters, const int32_t *grid_dim2, const float *min_position3, const float *range3,
                                    int dropin) {
    using namespace viewer;
    if (opt_size != (int)sizeof(RenderOptions)) return -2;
    RenderOptions opt;
    memcpy(&opt, opt_bytes, sizeof(opt));
    try {
        N3Tree tree;
        tree.open(npz_path);
        if (tree.N == 0) return -3;
        tree.move_to_device(tree.capacity, true, true);
        auto dev = torch::kCUDA;
        const int64_t rows = num_items, spc = opt.samples_per_corner;
        torch::Tensor nd = torch::from_blob((void *)nodes, {rows, 2}, torch::kInt32).clone().to(dev);
        torch::Tensor smp = torch::from_blob((void *)samples, {rows, spc, (int64_t)samples_dim}, torch::kFloat32).clone().to(dev);
        torch::Tensor cl = torch::full({rows, spc}, -1, torch::TensorOptions().device(dev).dtype(torch::kInt16));
        torch::Tensor gd = torch::from_blob((void *)grid_dim2, {2}, torch::kInt32).clone().to(dev);
        torch::Tensor mp = torch::from_blob((void *)min_position3, {3}, torch::kFloat32).clone().to(dev);
        torch::Tensor rg = torch::from_blob((void *)range3, {3}, torch::kFloat32).clone().to(dev);
        const int threads = 512, blocks = N_BLOCKS_NEEDED(rows, threads);
        if (dropin) {
            if (hipDeviceSynchronize() != hipSuccess) return -4;
            generate_samples(tree, opt, nd, smp, cl, gd, mp, rg);
        } else
        hipLaunchKernelGGL(generate_samples_kernel, dim3(blocks), dim3(threads), 0, 0, viewer::internal::TreeSpec(tree), opt,
                           nd.packed_accessor32<int32_t, 2, torch::RestrictPtrTraits>(), smp.packed_accessor32<float, 3, torch::RestrictPtrTraits>(),
                           cl.packed_accessor32<short, 2, torch::RestrictPtrTraits>(), gd.packed_accessor32<int32_t, 1, torch::RestrictPtrTraits>(),
                           mp.packed_accessor32<float, 1, torch::RestrictPtrTraits>(), rg.packed_accessor32<float, 1, torch::RestrictPtrTraits>(),
                           num_items);
        if (hipDeviceSynchronize() != hipSuccess) return -4;
        memcpy(samples, smp.cpu().data_ptr(), rows * spc * samples_dim * 4);
        memcpy(clusters, cl.cpu().data_ptr(), rows * spc * 2);
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_generate_samples_npz: %s\n", e.what());
        return -1;
    }
    return 0;
}

// The reference's adjust_parents_and_children_kernel (renderer_kernel.cu:63-86, launched as :536-549).
int ref_adjust_parents_dropin_npz(const char *npz_path, int first_shift_index, const uint8_t *to_delete, const int32_t *index_shifts,
                                  int32_t *child_out, int32_t *parent_out, int dropin);

int ref_adjust_parents_npz(const char *npz_path, int first_shift_index, const uint8_t *to_delete, const int32_t *index_shifts,
                           int32_t *child_out, int32_t *parent_out) {
    return ref_adjust_parents_dropin_npz(npz_path, first_shift_index, to_delete, index_shifts, child_out, parent_out, 0);
}

// dropin != 0: libmnv.so through the four-parameter viewer::adjust_parents_and_children of include/mnv_reference_binding.hpp
int ref_adjust_parents_dropin_npz(const char *npz_path, int first_shift_index, const uint8_t *to_delete, const int32_t *index_shifts,
                                  int32_t *child_out, int32_t *parent_out, int dropin) {
    using namespace viewer;
    try {
        N3Tree tree;
        tree.open(npz_path);
        if (tree.N == 0) return -3;
        const int64_t cap = tree.capacity;
        tree.move_to_device(cap, true, true);
        auto dev = torch::kCUDA;
        torch::Tensor td = torch::from_blob((void *)to_delete, {cap}, torch::kUInt8).clone().to(torch::kBool).to(dev);
        torch::Tensor sh = torch::from_blob((void *)index_shifts, {cap}, torch::kInt32).clone().to(dev);
        const int threads = 512, blocks = N_BLOCKS_NEEDED(cap - first_shift_index, threads);
        if (dropin) {
            if (hipDeviceSynchronize() != hipSuccess) return -4;
            adjust_parents_and_children(tree, first_shift_index, td, sh);
        } else
        hipLaunchKernelGGL(adjust_parents_and_children_kernel, dim3(blocks), dim3(threads), 0, 0, viewer::internal::TreeSpec(tree), first_shift_index,
                           td.packed_accessor32<bool, 1, torch::RestrictPtrTraits>(), sh.packed_accessor32<int32_t, 1, torch::RestrictPtrTraits>());
        if (hipDeviceSynchronize() != hipSuccess) return -4;
        memcpy(child_out, tree.child.cpu().contiguous().data_ptr(), (size_t)cap * 8 * 4);
        memcpy(parent_out, tree.parent.cpu().contiguous().data_ptr(), (size_t)cap * 4);
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_adjust_parents_npz: %s\n", e.what());
        return -1;
    }
    return 0;
}

// The reference's Camera constructor + _update pose math (src/camera.cpp:29-82, glm): pose vectors -> the 12-float
// camera-to-world matrix and the resolved intrinsics (fx, fy, cx, cy).  updates = how many times _update() runs after the
// vectors are set (the viewer calls it once per frame; it renormalises v_back each time).
int ref_camera_pose(int width, int height, float fx, float fy, float cx, float cy, const float *center3, const float *back3,
                    const float *up3, int updates, float *c2w12_out, float *intrinsics4_out) {
    using namespace viewer;
    try {
        Camera cam(width, height, fx, fy, cx, cy);
        cam.center = glm::vec3(center3[0], center3[1], center3[2]);
        cam.v_back = glm::vec3(back3[0], back3[1], back3[2]);
        cam.v_world_up = glm::vec3(up3[0], up3[1], up3[2]);
        for (int i = 0; i < updates; ++i) cam._update();
        (void)hipDeviceSynchronize();
        memcpy(c2w12_out, glm::value_ptr(cam.transform), 12 * sizeof(float));
        intrinsics4_out[0] = cam.fx;
        intrinsics4_out[1] = cam.fy;
        intrinsics4_out[2] = cam.cx;
        intrinsics4_out[3] = cam.cy;
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_camera_pose: %s\n", e.what());
        return -1;
    }
    return 0;
}

// The reference's Camera drag helpers (camera.hpp:22-25, camera.cpp:132-187): pose after begin_drag(x0, y0) .. drag_update(x1, y1) ..
// end_drag() and the next frame's _update().  pose9 = center, v_back, origin (in / out); c2w12_out = the new matrix.
int ref_camera_drag(int width, int height, float fx, const float *up3, float movement_speed, int is_pan, int about_origin, float x0, float y0,
                    float x1, float y1, float *pose9, float *c2w12_out) {
    using namespace viewer;
    try {
        Camera cam(width, height, fx);
        cam.center = glm::vec3(pose9[0], pose9[1], pose9[2]);
        cam.v_back = glm::vec3(pose9[3], pose9[4], pose9[5]);
        cam.origin = glm::vec3(pose9[6], pose9[7], pose9[8]);
        cam.v_world_up = glm::vec3(up3[0], up3[1], up3[2]);
        cam.movement_speed = movement_speed;
        cam._update();
        cam.begin_drag(x0, y0, is_pan != 0, about_origin != 0);
        cam.drag_update(x1, y1);
        cam.end_drag();
        cam._update();
        (void)hipDeviceSynchronize();
        for (int i = 0; i < 3; ++i) {
            pose9[i] = cam.center[i];
            pose9[3 + i] = cam.v_back[i];
            pose9[6 + i] = cam.origin[i];
        }
        memcpy(c2w12_out, glm::value_ptr(cam.transform), 12 * sizeof(float));
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_camera_drag: %s\n", e.what());
        return -1;
    }
    return 0;
}

// Drop-in demonstration: the reference's own loader, N3Tree (libtorch tensors on the device) and Camera (glm) feed libmnv.so
// through include/mnv_reference_binding.hpp -- what a ROCm build of the viewer would do per frame.  path: 0 = mnv_render_voxels on
// the reference's arrays (with trackers / visit marks), 1 = the packed accel.  Trackers / visited may be NULL.
int ref_dropin_render_npz(const char *npz_path, int width, int height, float fx, float fy, float cx, float cy, const float *center3,
                          const float *back3, const float *up3, const void *opt_bytes, int opt_size, int path, float *rgba_host,
                          uint8_t *rgba8_host, float *split_host, float *sample_host, int32_t *visited_host) {
    using namespace viewer;
    if (opt_size != (int)sizeof(RenderOptions)) return -2;
    RenderOptions opt;
    memcpy(&opt, opt_bytes, sizeof(opt));
    try {
        N3Tree tree;
        tree.open(npz_path);
        if (tree.N == 0) return -3;
        tree.move_to_device(tree.capacity, true, true);
        tree.sample_counts.fill_(8);
        Camera cam(width, height, fx, fy, cx, cy);  // the reference's camera: pose vectors -> _update() (glm)
        cam.center = glm::vec3(center3[0], center3[1], center3[2]);
        cam.v_back = glm::vec3(back3[0], back3[1], back3[2]);
        cam.v_world_up = glm::vec3(up3[0], up3[1], up3[2]);
        cam._update();
        const int64_t n = (int64_t)width * height;
        auto fopt = torch::TensorOptions().device(torch::kCUDA).dtype(torch::kFloat32);
        torch::Tensor out = torch::zeros({n, 4}, fopt), out8 = torch::zeros({n, 4}, torch::TensorOptions().device(torch::kCUDA).dtype(torch::kUInt8));
        torch::Tensor to_split = torch::full({n, 3}, -1.f, fopt), to_sample = torch::full({n, 3}, -1.f, fopt);
        torch::Tensor visited = torch::zeros({tree.capacity}, torch::TensorOptions().device(torch::kCUDA).dtype(torch::kInt32));
        if (hipDeviceSynchronize() != hipSuccess) return -4;
        if (path == 0) {
            render_voxels(tree, cam, opt, out.data_ptr<float>(), out8.data_ptr<uint8_t>(), nullptr, to_split.data_ptr<float>(),
                          to_sample.data_ptr<float>(), visited.data_ptr<int32_t>(), visited_host != nullptr);
        } else {
            mnv_accel *accel = make_accel(tree, tree.capacity, nullptr);
            render_voxels(accel, cam, opt, out.data_ptr<float>(), out8.data_ptr<uint8_t>(), nullptr);
            if (hipDeviceSynchronize() != hipSuccess) return -4;
            mnv_accel_destroy(accel);
        }
        if (hipDeviceSynchronize() != hipSuccess) return -4;
        memcpy(rgba_host, out.cpu().data_ptr<float>(), n * 4 * sizeof(float));
        if (rgba8_host) memcpy(rgba8_host, out8.cpu().data_ptr<uint8_t>(), n * 4);
        if (split_host) memcpy(split_host, to_split.cpu().data_ptr<float>(), n * 3 * sizeof(float));
        if (sample_host) memcpy(sample_host, to_sample.cpu().data_ptr<float>(), n * 3 * sizeof(float));
        if (visited_host) memcpy(visited_host, visited.cpu().data_ptr<int32_t>(), (size_t)tree.capacity * sizeof(int32_t));
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_dropin_render_npz: %s\n", e.what());
        return -1;
    }
    return 0;
}

// The same demonstration for the reference's LIVE call: render_voxels(tree, cam, opt, image_arr, depth_arr, stream, to_split, to_sample,
// visited, track_visit, offscreen = false) through the eleven-parameter binding.  image_host [h][w][4] is read (the image under the
// volume) and overwritten (the frame); depth_host [h][w].  path: 0 = reference arrays, 1 = packed accel.
int ref_dropin_onscreen_npz(const char *npz_path, int width, int height, float fx, float fy, float cx, float cy, const float *center3,
                            const float *back3, const float *up3, const void *opt_bytes, int opt_size, int path, int offscreen,
                            uint8_t *image_host, const float *depth_host) {
    using namespace viewer;
    if (opt_size != (int)sizeof(RenderOptions)) return -2;
    RenderOptions opt;
    memcpy(&opt, opt_bytes, sizeof(opt));
    try {
        N3Tree tree;
        tree.open(npz_path);
        if (tree.N == 0) return -3;
        tree.move_to_device(tree.capacity, true, true);
        tree.sample_counts.fill_(8);
        Camera cam(width, height, fx, fy, cx, cy);
        cam.center = glm::vec3(center3[0], center3[1], center3[2]);
        cam.v_back = glm::vec3(back3[0], back3[1], back3[2]);
        cam.v_world_up = glm::vec3(up3[0], up3[1], up3[2]);
        cam._update();
        const int64_t n = (int64_t)width * height;
        auto fopt = torch::TensorOptions().device(torch::kCUDA).dtype(torch::kFloat32);
        torch::Tensor image = torch::from_blob((void *)image_host, {n, 4}, torch::kUInt8).clone().to(torch::kCUDA);
        torch::Tensor depth = torch::from_blob((void *)depth_host, {n}, torch::kFloat32).clone().to(torch::kCUDA);
        torch::Tensor to_split = torch::full({n, 3}, -1.f, fopt), to_sample = torch::full({n, 3}, -1.f, fopt);
        torch::Tensor visited = torch::zeros({tree.capacity}, torch::TensorOptions().device(torch::kCUDA).dtype(torch::kInt32));
        uint8_t *image_arr = image.data_ptr<uint8_t>();
        float *depth_arr = depth.data_ptr<float>();
        hipStream_t stream = nullptr;
        if (hipDeviceSynchronize() != hipSuccess) return -4;
        if (path == 0) {
            render_voxels(tree, cam, opt, image_arr, depth_arr, stream, to_split, to_sample, visited, false, offscreen != 0);
        } else {
            mnv_accel *accel = make_accel(tree, tree.capacity, nullptr);
            render_voxels(accel, cam, opt, image_arr, depth_arr, stream, offscreen != 0);
            if (hipDeviceSynchronize() != hipSuccess) return -4;
            mnv_accel_destroy(accel);
        }
        if (hipDeviceSynchronize() != hipSuccess) return -4;
        memcpy(image_host, image.cpu().data_ptr<uint8_t>(), n * 4);
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_dropin_onscreen_npz: %s\n", e.what());
        return -1;
    }
    return 0;
}

int ref_get_samples_onscreen_npz(const char *npz_path, int width, int height, float fx, float fy, float cx, float cy, const float *c2w12,
                                 const void *opt_bytes, int opt_size, const int32_t *grid_dim2, const float *min_position3,
                                 const float *range3, int samples_dim, const float *tmax_host, int16_t *num_samples_host, float *samples_host,
                                 int16_t *cluster_host, float *split_host, float *sample_host, int dropin);

// The reference's get_samples_trace_ray on the tree in `npz_path` (full frame, offscreen).
int ref_get_samples_npz(const char *npz_path, int width, int height, float fx, float fy, float cx, float cy, const float *c2w12,
                        const void *opt_bytes, int opt_size, const int32_t *grid_dim2, const float *min_position3,
                        const float *range3, int samples_dim, int16_t *num_samples_host, float *samples_host,
                        int16_t *cluster_host) {
    return ref_get_samples_onscreen_npz(npz_path, width, height, fx, fy, cx, cy, c2w12, opt_bytes, opt_size, grid_dim2, min_position3, range3, samples_dim,
                                        nullptr, num_samples_host, samples_host, cluster_host, nullptr, nullptr, 0);
}

// ... with the depth attachment of offscreen == false (tmax_host [h][w], or NULL) and, when asked for, the two tracker arrays [h*w][3].
// dropin != 0: not the reference's device code but libmnv.so through the sixteen-parameter binding of include/mnv_reference_binding.hpp
// (viewer::get_samples_from_voxels on the reference's own N3Tree / Camera / tensors) -- the same arrays must come back.
int ref_get_samples_onscreen_npz(const char *npz_path, int width, int height, float fx, float fy, float cx, float cy, const float *c2w12,
                                 const void *opt_bytes, int opt_size, const int32_t *grid_dim2, const float *min_position3,
                                 const float *range3, int samples_dim, const float *tmax_host, int16_t *num_samples_host, float *samples_host,
                                 int16_t *cluster_host, float *split_host, float *sample_host, int dropin) {
    using namespace viewer;
    if (opt_size != (int)sizeof(RenderOptions)) return -2;
    RenderOptions opt;
    memcpy(&opt, opt_bytes, sizeof(opt));
    try {
        N3Tree tree;
        tree.open(npz_path);
        if (tree.N == 0) return -3;
        tree.move_to_device(tree.capacity, true, true);
        tree.sample_counts.fill_(8);
        auto camera = make_camera(width, height, fx, fy, cx, cy, c2w12);
        const CameraSpec cam(*camera);
        const int64_t n = (int64_t)width * height;
        auto dev = torch::kCUDA;
        auto fopt = torch::TensorOptions().device(dev).dtype(torch::kFloat32);
        torch::Tensor to_split = torch::full({n, 3}, -1.f, fopt), to_sample = torch::full({n, 3}, -1.f, fopt);
        torch::Tensor visited = torch::zeros({tree.capacity}, torch::TensorOptions().device(dev).dtype(torch::kInt32));
        torch::Tensor num_samples = torch::zeros({n}, torch::TensorOptions().device(dev).dtype(torch::kInt16));
        torch::Tensor samples = torch::full({n, (int64_t)opt.max_guided_samples, (int64_t)samples_dim}, -1.f, fopt);
        torch::Tensor clusters = torch::full({n, (int64_t)opt.max_guided_samples}, -1, torch::TensorOptions().device(dev).dtype(torch::kInt16));
        torch::Tensor gd = torch::from_blob((void *)grid_dim2, {2}, torch::kInt32).clone().to(dev);
        torch::Tensor mp = torch::from_blob((void *)min_position3, {3}, torch::kFloat32).clone().to(dev);
        torch::Tensor rg = torch::from_blob((void *)range3, {3}, torch::kFloat32).clone().to(dev);
        torch::Tensor tmax_dev;
        if (tmax_host) tmax_dev = torch::from_blob((void *)tmax_host, {n}, torch::kFloat32).clone().to(dev);
        const int threads = 512, blocks = N_BLOCKS_NEEDED(n, threads);
        if (dropin) {
            float *depth_arr = tmax_host ? tmax_dev.data_ptr<float>() : nullptr;
            hipStream_t stream = nullptr;
            if (hipDeviceSynchronize() != hipSuccess) return -4;
            get_samples_from_voxels(tree, *camera, opt, depth_arr, stream, to_split, to_sample, visited, false, tmax_host == nullptr, num_samples, samples,
                                    clusters, gd, mp, rg);
        } else
        hipLaunchKernelGGL(ref_get_samples_kernel, dim3(blocks), dim3(threads), 0, 0, viewer::internal::TreeSpec(tree), cam, opt,
                           tmax_host ? tmax_dev.data_ptr<float>() : (const float *)nullptr,
                           to_split.packed_accessor32<float, 2, torch::RestrictPtrTraits>(),
                           to_sample.packed_accessor32<float, 2, torch::RestrictPtrTraits>(),
                           visited.packed_accessor32<int32_t, 1, torch::RestrictPtrTraits>(),
                           num_samples.packed_accessor32<short, 1, torch::RestrictPtrTraits>(),
                           samples.packed_accessor64<float, 3, torch::RestrictPtrTraits>(),
                           clusters.packed_accessor64<short, 2, torch::RestrictPtrTraits>(),
                           gd.packed_accessor32<int32_t, 1, torch::RestrictPtrTraits>(),
                           mp.packed_accessor32<float, 1, torch::RestrictPtrTraits>(),
                           rg.packed_accessor32<float, 1, torch::RestrictPtrTraits>());
        if (hipDeviceSynchronize() != hipSuccess) return -4;
        memcpy(num_samples_host, num_samples.cpu().data_ptr(), n * 2);
        memcpy(samples_host, samples.cpu().data_ptr(), n * opt.max_guided_samples * samples_dim * 4);
        memcpy(cluster_host, clusters.cpu().data_ptr(), n * opt.max_guided_samples * 2);
        if (split_host) memcpy(split_host, to_split.cpu().data_ptr(), n * 3 * 4);
        if (sample_host) memcpy(sample_host, to_sample.cpu().data_ptr(), n * 3 * 4);
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_get_samples_npz: %s\n", e.what());
        return -1;
    }
    return 0;
}

// The reference's composite_nerf_results; the tree in `npz_path` supplies format / basis_dim.
int ref_render_nerf_results_npz(const char *npz_path, int width, int height, float fx, float fy, float cx, float cy,
                                const float *c2w12, const void *opt_bytes, int opt_size, const float *sample_values,
                                int64_t n_samples, int value_stride, const float *z_vals, const int64_t *offsets,
                                float *rgba_host) {
    using namespace viewer;
    if (opt_size != (int)sizeof(RenderOptions)) return -2;
    RenderOptions opt;
    memcpy(&opt, opt_bytes, sizeof(opt));
    try {
        N3Tree tree;
        tree.open(npz_path);
        if (tree.N == 0) return -3;
        tree.move_to_device(tree.capacity, true, true);
        auto camera = make_camera(width, height, fx, fy, cx, cy, c2w12);
        const CameraSpec cam(*camera);
        const int64_t n = (int64_t)width * height;
        auto dev = torch::kCUDA;
        torch::Tensor sv = torch::from_blob((void *)sample_values, {n_samples, (int64_t)value_stride}, torch::kFloat32).clone().to(dev);
        torch::Tensor zv = torch::from_blob((void *)z_vals, {n_samples}, torch::kFloat32).clone().to(dev);
        torch::Tensor of = torch::from_blob((void *)offsets, {n}, torch::kInt64).clone().to(dev);
        torch::Tensor out = torch::zeros({n, 4}, torch::TensorOptions().device(dev).dtype(torch::kFloat32));
        const int threads = 512, blocks = N_BLOCKS_NEEDED(n, threads);
        hipLaunchKernelGGL(ref_render_nerf_kernel, dim3(blocks), dim3(threads), 0, 0, viewer::internal::TreeSpec(tree), cam, opt,
                           out.data_ptr<float>(), sv.packed_accessor64<float, 2, torch::RestrictPtrTraits>(),
                           zv.packed_accessor64<float, 1, torch::RestrictPtrTraits>(),
                           of.packed_accessor32<int64_t, 1, torch::RestrictPtrTraits>());
        if (hipDeviceSynchronize() != hipSuccess) return -4;
        memcpy(rgba_host, out.cpu().data_ptr(), n * 16);
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_render_nerf_results_npz: %s\n", e.what());
        return -1;
    }
    return 0;
}

// libmnv.so through the nine-parameter viewer::render_nerf_results of include/mnv_reference_binding.hpp on the reference's own N3Tree,
// Camera and tensors: the RGBA8 image the reference's launcher writes through its surface (renderer_kernel.cu:237).  image_host [h][w][4]
// is uploaded first (what is "under" the frame; the result must not depend on it, renderer_kernel.cu:316) and returned overwritten.
int ref_render_nerf_results_dropin_npz(const char *npz_path, int width, int height, float fx, float fy, float cx, float cy, const float *c2w12,
                                       const void *opt_bytes, int opt_size, const float *sample_values, int64_t n_samples, int value_stride,
                                       const float *z_vals, const int64_t *offsets, int offscreen, uint8_t *image_host) {
    using namespace viewer;
    if (opt_size != (int)sizeof(RenderOptions)) return -2;
    RenderOptions opt;
    memcpy(&opt, opt_bytes, sizeof(opt));
    try {
        N3Tree tree;
        tree.open(npz_path);
        if (tree.N == 0) return -3;
        tree.move_to_device(tree.capacity, true, true);
        auto camera = make_camera(width, height, fx, fy, cx, cy, c2w12);
        const int64_t n = (int64_t)width * height;
        auto dev = torch::kCUDA;
        torch::Tensor sv = torch::from_blob((void *)sample_values, {n_samples, (int64_t)value_stride}, torch::kFloat32).clone().to(dev);
        torch::Tensor zv = torch::from_blob((void *)z_vals, {n_samples}, torch::kFloat32).clone().to(dev);
        torch::Tensor of = torch::from_blob((void *)offsets, {n}, torch::kInt64).clone().to(dev);
        torch::Tensor image = torch::from_blob((void *)image_host, {n, 4}, torch::kUInt8).clone().to(dev);
        uint8_t *image_arr = image.data_ptr<uint8_t>();
        hipStream_t stream = nullptr;
        if (hipDeviceSynchronize() != hipSuccess) return -4;
        render_nerf_results(tree, *camera, opt, image_arr, stream, sv, zv, of, offscreen != 0);
        if (hipDeviceSynchronize() != hipSuccess) return -4;
        memcpy(image_host, image.cpu().data_ptr<uint8_t>(), n * 4);
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_render_nerf_results_dropin_npz: %s\n", e.what());
        return -1;
    }
    return 0;
}

int ref_render_options_size(void) { return (int)sizeof(viewer::RenderOptions); }

}  // extern "C"
