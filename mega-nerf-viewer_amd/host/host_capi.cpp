// host_capi.cpp -- extern "C" wrappers of include/mnv.h over the C++ host data model.
#include <cstring>
#include <exception>
#include <string>

#include "../../include/mnv.h"
#include "../csrc/mnv_error.h"
#include "camera.hpp"
#include "n3tree.hpp"
#include "render_options.hpp"
#include "volume_renderer.hpp"

namespace viewer::synth {
void random_tree(const mnv_synth_random_params &p, N3Tree &out);
void shell_tree(const mnv_synth_shell_params &p, N3Tree &out);
void terrain_tree(const mnv_synth_terrain_params &p, N3Tree &out);
}  // namespace viewer::synth

struct mnv_n3tree {
    viewer::N3Tree tree;
};

struct mnv_renderer {
    viewer::VolumeRenderer rend;
};

namespace {
template <typename F>
int guarded(F f) {
    try {
        return f();
    } catch (const std::exception &e) {
        return mnv::set_error(MNV_E_IO, e.what());
    } catch (...) {
        return mnv::set_error(MNV_E_IO, "unknown C++ exception");
    }
}
}  // namespace

extern "C" {

void mnv_default_render_options(mnv_render_options *opt) {
    const viewer::RenderOptions d;
    std::memcpy(opt, &d, sizeof(*opt));
}

void mnv_cli_render_options(mnv_render_options *opt) {
    // src/opts.cpp:17-32 defaults mapped as in :49-67
    viewer::RenderOptions d;
    d.background_brightness = 0.f;
    d.step_size = 1e-4f;
    d.stop_thresh = 1e-2f;
    d.sigma_thresh = 1e-2f;
    d.split_batch_size = 4096;
    d.nerf_batch_size = 4096;
    d.samples_per_corner = 8;
    d.appearance_embedding = -1;
    d.max_guided_samples = 128;
    std::memcpy(opt, &d, sizeof(*opt));
}

void mnv_camera_init(mnv_camera *cam, int32_t width, int32_t height, float fx, float fy, float cx, float cy) {
    const viewer::Camera c(width, height, fx, fy, cx, cy);
    *cam = c.c_abi();
}

void mnv_camera_set_pose(mnv_camera *cam, const float center[3], const float v_back[3], const float v_world_up[3]) {
    viewer::Camera c(cam->width, cam->height, cam->fx, cam->fy, cam->cx, cam->cy);
    c.center = {center[0], center[1], center[2]};
    c.v_back = {v_back[0], v_back[1], v_back[2]};
    c.v_world_up = {v_world_up[0], v_world_up[1], v_world_up[2]};
    c._update();
    std::memcpy(cam->c2w, c.transform, sizeof(cam->c2w));
}

void mnv_camera_drag(mnv_camera *cam, float center[3], float v_back[3], const float v_world_up[3], float origin[3], float movement_speed,
                     int is_pan, int about_origin, float x0, float y0, float x1, float y1) {
    viewer::Camera c(cam->width, cam->height, cam->fx, cam->fy, cam->cx, cam->cy);
    c.center = {center[0], center[1], center[2]};
    c.v_back = {v_back[0], v_back[1], v_back[2]};
    c.v_world_up = {v_world_up[0], v_world_up[1], v_world_up[2]};
    c.origin = {origin[0], origin[1], origin[2]};
    c.movement_speed = movement_speed;
    c._update();
    c.begin_drag(x0, y0, is_pan != 0, about_origin != 0);
    c.drag_update(x1, y1);
    c.end_drag();
    c._update();  // what the next frame does (VolumeRenderer::render -> camera._update)
    for (int i = 0; i < 3; ++i) {
        center[i] = c.center[i];
        v_back[i] = c.v_back[i];
        origin[i] = c.origin[i];
    }
    std::memcpy(cam->c2w, c.transform, sizeof(cam->c2w));
}

int mnv_n3tree_open(const char *npz_path, mnv_n3tree **out) {
    if (!npz_path || !out) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        auto *t = new mnv_n3tree();
        try {
            t->tree.open(npz_path);
        } catch (...) {
            delete t;
            throw;
        }
        *out = t;
        return MNV_OK;
    });
}

int mnv_n3tree_from_arrays(const mnv_tree_view *host_view, mnv_n3tree **out) {
    if (!host_view || !out) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        auto *t = new mnv_n3tree();
        try {
            t->tree.assign(*host_view);
        } catch (...) {
            delete t;
            throw;
        }
        *out = t;
        return MNV_OK;
    });
}

void mnv_n3tree_free(mnv_n3tree *t) { delete t; }

int mnv_n3tree_host_view(const mnv_n3tree *t, mnv_tree_view *view) {
    if (!t || !view) return mnv::set_error(MNV_E_INVALID, "null argument");
    *view = t->tree.host_view();
    return MNV_OK;
}

int mnv_n3tree_move_to_device(mnv_n3tree *t, int64_t max_capacity, int need_parent, int need_sample_counts,
                              void *hip_stream) {
    if (!t) return mnv::set_error(MNV_E_INVALID, "null argument");
    if (mnv_device_count() <= 0) return mnv::set_error(MNV_E_NO_DEVICE, "no HIP device visible");
    return guarded([&] {
        t->tree.move_to_device((long)max_capacity, need_parent != 0, need_sample_counts != 0, hip_stream);
        return MNV_OK;
    });
}

int mnv_n3tree_device_view(const mnv_n3tree *t, mnv_tree_view *view) {
    if (!t || !view) return mnv::set_error(MNV_E_INVALID, "null argument");
    if (!t->tree.on_device()) return mnv::set_error(MNV_E_INVALID, "tree is not on the device");
    *view = t->tree.device_view();
    return MNV_OK;
}

const mnv_accel *mnv_n3tree_accel(const mnv_n3tree *t) { return t ? t->tree.device.accel : nullptr; }

int mnv_n3tree_save_npz(const mnv_n3tree *t, const char *npz_path) {
    if (!t || !npz_path) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        t->tree.save_npz(npz_path);
        return MNV_OK;
    });
}

void mnv_data_format_parse(const char *str, int32_t *format, int32_t *basis_dim) {
    viewer::DataFormat f;
    f.parse(str ? str : "");
    if (format) *format = f.format == viewer::DataFormat::SH ? MNV_FORMAT_SH : MNV_FORMAT_RGBA;
    if (basis_dim) *basis_dim = f.basis_dim;
}

int mnv_data_format_to_string(int32_t format, int32_t basis_dim, char *buf, size_t buflen) {
    viewer::DataFormat f;
    f.format = format == MNV_FORMAT_SH ? viewer::DataFormat::SH : viewer::DataFormat::RGBA;
    f.basis_dim = basis_dim;
    const std::string s = f.to_string();
    if (!buf || buflen <= s.size()) return mnv::set_error(MNV_E_INVALID, "buffer too small");
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return MNV_OK;
}

int mnv_synth_random_tree(const mnv_synth_random_params *p, mnv_n3tree **out) {
    if (!p || !out) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        auto *t = new mnv_n3tree();
        try {
            viewer::synth::random_tree(*p, t->tree);
        } catch (...) {
            delete t;
            throw;
        }
        *out = t;
        return MNV_OK;
    });
}

int mnv_synth_terrain_tree(const mnv_synth_terrain_params *p, mnv_n3tree **out) {
    if (!p || !out) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        auto *t = new mnv_n3tree();
        try {
            viewer::synth::terrain_tree(*p, t->tree);
        } catch (...) {
            delete t;
            throw;
        }
        *out = t;
        return MNV_OK;
    });
}

int mnv_synth_shell_tree(const mnv_synth_shell_params *p, mnv_n3tree **out) {
    if (!p || !out) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        auto *t = new mnv_n3tree();
        try {
            viewer::synth::shell_tree(*p, t->tree);
        } catch (...) {
            delete t;
            throw;
        }
        *out = t;
        return MNV_OK;
    });
}

/* ---- viewer::VolumeRenderer behind the C ABI (include/renderer/renderer.hpp:9-39) */

int mnv_renderer_create(mnv_renderer **out) {
    if (!out) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        *out = new mnv_renderer();
        return MNV_OK;
    });
}

void mnv_renderer_destroy(mnv_renderer *r) { delete r; }

int mnv_renderer_set(mnv_renderer *r, mnv_n3tree *tree, int64_t max_tree_capacity) {
    if (!r || !tree) return mnv::set_error(MNV_E_INVALID, "null argument");
    if (max_tree_capacity < tree->tree.capacity) return mnv::set_error(MNV_E_INVALID, "max_tree_capacity is smaller than the tree");
    return guarded([&] {
        r->rend.set(tree->tree, (long)max_tree_capacity);
        return MNV_OK;
    });
}

int mnv_renderer_load_model(mnv_renderer *r, const char *npz_path) {
    if (!r || !npz_path) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        r->rend.load_model(npz_path);
        return MNV_OK;
    });
}

int mnv_renderer_set_model(mnv_renderer *r, const mnv_mlp_desc *desc, const uint16_t *params, size_t n_halfs, const mnv_cluster_grid *grid) {
    if (!r || !desc || !params || !grid) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        r->rend.set_model(*desc, params, n_halfs, *grid);
        return MNV_OK;
    });
}

int mnv_renderer_resize(mnv_renderer *r, int32_t width, int32_t height) {
    if (!r || width < 1 || height < 1) return mnv::set_error(MNV_E_INVALID, "invalid size");
    return guarded([&] {
        r->rend.resize(width, height);
        return MNV_OK;
    });
}

mnv_render_options *mnv_renderer_options(mnv_renderer *r) { return r ? reinterpret_cast<mnv_render_options *>(&r->rend.options) : nullptr; }

int mnv_renderer_set_camera(mnv_renderer *r, float fx, float fy, const float center[3], const float v_back[3], const float v_world_up[3]) {
    if (!r || !center || !v_back || !v_world_up) return mnv::set_error(MNV_E_INVALID, "null argument");
    viewer::Camera &c = r->rend.camera;
    if (fx > 0.f) c.fx = c.default_fx = fx;
    if (fy > 0.f) c.fy = c.default_fy = fy;
    else if (fx > 0.f) c.fy = c.default_fy = fx;
    c.center = {center[0], center[1], center[2]};
    c.v_back = {v_back[0], v_back[1], v_back[2]};
    c.v_world_up = {v_world_up[0], v_world_up[1], v_world_up[2]};
    return MNV_OK;
}

int mnv_renderer_set_seed(mnv_renderer *r, uint64_t seed, int32_t accel_rebuild_after) {
    if (!r) return mnv::set_error(MNV_E_INVALID, "null argument");
    r->rend.seed = seed;
    if (accel_rebuild_after >= 0) r->rend.accel_rebuild_after = accel_rebuild_after;
    return MNV_OK;
}

int mnv_renderer_render(mnv_renderer *r, mnv_renderer_stats *stats) {
    if (!r) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        r->rend.render();
        if (stats) {
            const auto &s = r->rend.stats;
            stats->track_visit = s.track_visit;
            stats->used_accel = s.used_accel;
            stats->full = s.full;
            stats->split_candidates = s.split_candidates;
            stats->added = s.added;
            stats->sample_candidates = s.sample_candidates;
            stats->resampled = s.resampled;
            stats->pruned = s.pruned;
            stats->guided_samples = s.guided_samples;
            stats->capacity = s.capacity;
            stats->fused = s.fused;
            stats->reserved = 0;
        }
        return MNV_OK;
    });
}

int mnv_renderer_download(mnv_renderer *r, float *rgba, uint8_t *rgba8) {
    if (!r) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        std::vector<float> f;
        std::vector<uint8_t> u;
        r->rend.download(rgba ? &f : nullptr, rgba8 ? &u : nullptr);
        if (rgba) std::memcpy(rgba, f.data(), f.size() * sizeof(float));
        if (rgba8) std::memcpy(rgba8, u.data(), u.size());
        return MNV_OK;
    });
}

int mnv_renderer_set_frames_in_flight(mnv_renderer *r, int32_t count) {
    if (!r || count < 1 || count > 16) return mnv::set_error(MNV_E_INVALID, "frames in flight: 1 .. 16");
    return guarded([&] {
        r->rend.sync_tree_streams();
        r->rend.frames_in_flight = count;
        return MNV_OK;
    });
}

int mnv_renderer_set_guided_in_flight(mnv_renderer *r, int enable) {
    if (!r) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        r->rend.sync_tree_streams();
        r->rend.guided_in_flight = enable != 0;
        return MNV_OK;
    });
}

int mnv_renderer_slot_guided_samples(mnv_renderer *r, int32_t slot, int64_t *count_out) {
    if (!r || !count_out) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        *count_out = (int64_t)r->rend.slot_guided_samples(slot);
        return MNV_OK;
    });
}

int mnv_renderer_set_fused_guided(mnv_renderer *r, int enable) {
    if (!r) return mnv::set_error(MNV_E_INVALID, "null argument");
    r->rend.use_fused_guided = enable != 0;
    return MNV_OK;
}

int mnv_renderer_set_frame_inputs(mnv_renderer *r, const float *tmax_px, const uint8_t *rgba8_init) {
    if (!r) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        r->rend.set_frame_inputs(tmax_px, rgba8_init);
        return MNV_OK;
    });
}

int mnv_renderer_set_ranks(mnv_renderer *r, mnv_comm *comm, int32_t tile_w, int32_t tile_h) {
    if (!r) return mnv::set_error(MNV_E_INVALID, "null argument");
    if (comm && (tile_w < 8 || tile_h < 8 || tile_w % 8 || tile_h % 8)) return mnv::set_error(MNV_E_INVALID, "macro tiles are multiples of 8 pixels");
    return guarded([&] {
        r->rend.set_ranks(comm, tile_w, tile_h);
        return MNV_OK;
    });
}

int32_t mnv_renderer_last_slot(const mnv_renderer *r) { return r ? r->rend.last_slot() : -1; }

int mnv_renderer_download_slot(mnv_renderer *r, int32_t slot, float *rgba, uint8_t *rgba8) {
    if (!r) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        std::vector<float> f;
        std::vector<uint8_t> u;
        r->rend.download_slot(slot, rgba ? &f : nullptr, rgba8 ? &u : nullptr);
        if (rgba) std::memcpy(rgba, f.data(), f.size() * sizeof(float));
        if (rgba8) std::memcpy(rgba8, u.data(), u.size());
        return MNV_OK;
    });
}

int mnv_renderer_sync_tree(mnv_renderer *r) {
    if (!r) return mnv::set_error(MNV_E_INVALID, "null argument");
    return guarded([&] {
        r->rend.sync_tree();
        return MNV_OK;
    });
}

}  // extern "C"
