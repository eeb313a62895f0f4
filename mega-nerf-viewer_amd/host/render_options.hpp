// render_options.hpp -- viewer::RenderOptions, the drop-in API struct of the reference
// (reference include/render_options.hpp:9-56), field for field with the same defaults.
// It is layout-identical to the C-ABI's mnv_render_options (checked below), so a
// RenderOptions can be handed to mnv_render_voxels by address.
#pragma once

#include "../../include/mnv.h"

#define VIEWER_GLOBAL_BASIS_MAX MNV_BASIS_MAX

namespace viewer {

struct RenderOptions {
    // * BASIC RENDERING
    float step_size = 1e-4f;            // epsilon added to each step
    float sigma_thresh = 1e-2f;         // sigma below this counts as empty
    float stop_thresh = 1e-2f;          // stop when remaining light falls below this
    float background_brightness = 1.f;
    // * VISUALIZATION
    float render_bbox[6] = {0.f, 0.f, 0.f, 1.f, 1.f, 1.f};  // in tree space [0,1]^3
    int basis_minmax[2] = {0, VIEWER_GLOBAL_BASIS_MAX - 1};
    float rot_dirs[3] = {0.f, 0.f, 0.f};                    // axis-angle applied to view dirs
    // * ADVANCED VISUALIZATION
    bool show_grid = false;
    int grid_max_depth = 4;
    bool render_depth = false;
    bool use_splitting = false;
    bool use_guided_sampling = false;
    int max_depth = 16;
    int samples_per_corner = 8;
    int split_batch_size = 4192;
    int nerf_batch_size = 1024;
    int max_sample_count = 256;
    bool need_viewdir = false;
    int appearance_embedding = -1;
    int max_guided_samples = 128;

    const mnv_render_options *c_abi() const { return reinterpret_cast<const mnv_render_options *>(this); }
};

static_assert(sizeof(RenderOptions) == sizeof(mnv_render_options), "RenderOptions must mirror mnv_render_options");
static_assert(offsetof(RenderOptions, render_bbox) == offsetof(mnv_render_options, render_bbox), "layout");
static_assert(offsetof(RenderOptions, render_depth) == offsetof(mnv_render_options, render_depth), "layout");
static_assert(offsetof(RenderOptions, max_guided_samples) == offsetof(mnv_render_options, max_guided_samples), "layout");

}  // namespace viewer
