#include "volume_renderer.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "npz.hpp"

namespace viewer {

namespace {
void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void mnv_check(int rc, const char *what) {
    if (rc != MNV_OK) throw std::runtime_error(std::string(what) + ": " + mnv_last_error());
}
}  // namespace

// grow-only device buffer
struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    ~DeviceBuffer() { release(); }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    template <typename T>
    T *get(size_t count) {
        const size_t need = count * sizeof(T);
        if (need > bytes) {
            release();
            hip_check(hipMalloc(&ptr, need + need / 8), "hipMalloc(refinement buffer)");
            bytes = need + need / 8;
        }
        return static_cast<T *>(ptr);
    }
};

struct VolumeRenderer::Impl {
    N3Tree *tree = nullptr;
    long max_tree_capacity = 0;
    mnv_frame_inputs inputs = {nullptr, nullptr};  // offscreen == false: the caller's depth image and image (set_frame_inputs); the _ex entry
                                                   // points take a block of two nulls as offscreen == true
    // Frames in flight.  The reference calls render_voxels once per frame on one stream (cuda_renderer.cpp:141-142); a single
    // 1080p launch of the march ends in a tail of a few wavefronts finishing the longest rays (a third of the wave slots idle
    // on average), so plain frames rotate over `slots` -- each with its own stream and frame buffers -- and the tail of frame k
    // overlaps frames k+1, k+2.  Frames that refine the tree run one at a time on slot 0 (they mutate what the next one reads).
    struct Slot {
        hipStream_t stream = nullptr;
        float *rgba = nullptr;
        uint8_t *rgba8 = nullptr;
        unsigned long long *count_dev = nullptr, *count_host = nullptr;  // guided frames in flight: this slot's sample counter and its pinned copy
        bool counted = false;                                             // ... which a frame on this slot has written (or is about to)
    };
    std::vector<Slot> slots;
    int cur = 0;            // slot of the most recent render()
    bool overlapped = false;  // plain frames may still be running on slots other than 0
    bool tree_stream_dirty = false;  // slot 0's stream may still be editing the tree / accel (a frame other than a plain one ran last)
    hipStream_t stream = nullptr;  // = slots[0].stream: refinement, tree upload, accel rebuild
    float *rgba = nullptr;         // = slots[cur]
    uint8_t *rgba8 = nullptr;
    int width = 0, height = 0;
    bool initial_resize = true;

    // refinement state (cuda_renderer.cpp:441-468,571-600)
    mnv_mlp *mlp = nullptr;
    mnv_mlp_desc mlp_desc{};
    mnv_cluster_grid grid{};
    DeviceBuffer split_tracker, sample_tracker, visit_tracker, num_samples, cluster_indices, guided_samples;
    DeviceBuffer offsets, z_vals, sample_rows, sample_clusters, nerf_results;
    DeviceBuffer nodes, rand_sample, rand_clusters, results, fused_counter;
    bool patch_after_prune = true;
    // several ranks (set_ranks): this rank's share of the frame in compact tile-major order, the gather table on rank 0, the marks table
    mnv_comm *comm = nullptr;
    mnv_partition part = {0, 1, 0, 0, 0};
    int64_t rank_px = 0;  // pixels per rank in the compact layout: j_max macro tiles
    DeviceBuffer local, local8, table, table8, marks_table;
    int64_t tracker_rows() const { return comm ? rank_px * part.world : (int64_t)width * height; }
    bool fused_inputs_ok = false;  // the model's encoded input fits the fused guided kernel (<= 64 features)
    bool prune_happened = false, can_reuse_results = false, accel_stale = false;
    bool marks_fresh = false, want_marks = false;  // see render(): prune only after a track_visit frame
    int quiet_frames = 0;
    long reuse_total = 0;
    uint64_t frame = 0;
    // the fused guided kernel's sample count: copied to pinned memory behind the kernel and read at the frame's LAST wait (the vote's,
    // the tree edit's) instead of at a wait of its own right behind the march -- 20-40 us of a refinement frame
    unsigned long long *count_host = nullptr;
    bool count_pending = false;
    void read_count_later(const unsigned long long *counter) {
        if (!count_host) hip_check(hipHostMalloc((void **)&count_host, sizeof(unsigned long long), hipHostMallocDefault), "hipHostMalloc(sample count)");
        hip_check(hipMemcpyAsync(count_host, counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream), "read sample counter");
        count_pending = true;
    }
    void finish_count(FrameStats &st) {
        if (!count_pending) return;
        hip_check(hipStreamSynchronize(stream), "guided frame");
        st.guided_samples = (long)*count_host;
        count_pending = false;
    }

    Impl() {
        slots.resize(1);
        hip_check(hipStreamCreate(&slots[0].stream), "hipStreamCreate");
        stream = slots[0].stream;
    }
    ~Impl() {
        free_frame();
        if (count_host) (void)hipHostFree(count_host);
        if (mlp) mnv_mlp_destroy(mlp);
        for (Slot &s : slots) {
            if (s.count_dev) (void)hipFree(s.count_dev);
            if (s.count_host) (void)hipHostFree(s.count_host);
            if (s.stream) (void)hipStreamDestroy(s.stream);
        }
    }
    void sync_all() {
        for (Slot &s : slots) hip_check(hipStreamSynchronize(s.stream), "hipStreamSynchronize");
        overlapped = false;
    }
    void free_frame() {
        for (Slot &s : slots) {
            if (s.stream) (void)hipStreamSynchronize(s.stream);
            if (s.rgba) (void)hipFree(s.rgba);
            if (s.rgba8) (void)hipFree(s.rgba8);
            s.rgba = nullptr;
            s.rgba8 = nullptr;
        }
        rgba = nullptr;
        rgba8 = nullptr;
    }
    // make `n` slots exist, each with a stream and frame buffers of the current size
    void ensure_slots(int n) {
        if (n < 1) n = 1;
        while ((int)slots.size() < n) {
            Slot s;
            hip_check(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking), "hipStreamCreate");
            slots.push_back(s);
        }
        for (int i = 0; i < n; ++i) {
            Slot &s = slots[i];
            if (!s.rgba && width > 0 && height > 0) {
                hip_check(hipMalloc((void **)&s.rgba, (size_t)width * height * 4 * sizeof(float)), "hipMalloc(frame)");
                hip_check(hipMalloc((void **)&s.rgba8, (size_t)width * height * 4), "hipMalloc(frame8)");
            }
        }
    }
    void use_slot(int i) {
        cur = i;
        rgba = slots[i].rgba;
        rgba8 = slots[i].rgba8;
    }
    void fill_f32(float *p, size_t n, float v) {
        uint32_t bits;
        std::memcpy(&bits, &v, 4);
        hip_check(hipMemsetD32Async((hipDeviceptr_t)p, (int)bits, n, stream), "fill");
    }
    mnv_tree_edit edit() const {
        mnv_tree_edit e{};
        e.child = tree->device.child;
        e.parent = tree->device.parent;
        for (int i = 0; i < 3; ++i) {
            e.offset[i] = tree->offset[i];
            e.scale[i] = tree->scale[i];
        }
        e.N = tree->N;
        e.capacity = tree->capacity;
        return e;
    }
    int sample_cols(const RenderOptions &o) const { return 3 + (o.need_viewdir ? 3 : 0) + (o.appearance_embedding != -1 ? 1 : 0); }

    void tree_changed() {
        accel_stale = true;
        quiet_frames = 0;
        can_reuse_results = false;
    }

    // cuda_renderer.cpp:205-272
    void expand_voxels(RenderOptions &o, FrameStats &st, uint64_t seed);
    // cuda_renderer.cpp:274-333
    void get_more_samples(RenderOptions &o, FrameStats &st, uint64_t seed);
    // cuda_renderer.cpp:335-381
    void prune_tree(FrameStats &st);
    // query_submodules for the samples of a refinement step.  On several ranks (set_ranks) every rank evaluates its share of the rows
    // -- the samples are a pure function of (seed, frame), so every rank holds them all -- and the results are all-gathered: what
    // stays replicated of the step is the vote and the tree edit, not the networks ("broadcast new rows").
    float *query_rows(const int16_t *d_clusters, const float *d_rand, int dim, int64_t rows, int stride);
    void refine_after_frame(RenderOptions &o, FrameStats &st, uint64_t seed, bool track_visit);
};

float *VolumeRenderer::Impl::query_rows(const int16_t *d_clusters, const float *d_rand, int dim, int64_t rows, int stride) {
    const int world = comm ? part.world : 1, rank = comm ? part.rank : 0;
    if (world <= 1) {
        float *d_results = results.get<float>((size_t)rows * stride);
        mnv_check(mnv_query_submodules(mlp, d_clusters, d_rand, dim, rows, d_results, stride, stream), "mnv_query_submodules");
        return d_results;
    }
    const int64_t per = (rows + world - 1) / world;  // equal blocks (the last one ragged): the all-gather's table is [world][per rows]
    float *d_results = results.get<float>((size_t)per * world * stride);
    const int64_t lo = std::min<int64_t>(rows, per * rank), hi = std::min<int64_t>(rows, per * (rank + 1));
    if (hi > lo)
        mnv_check(mnv_query_submodules(mlp, d_clusters + lo, d_rand + lo * dim, dim, hi - lo, d_results + lo * stride, stride, stream), "mnv_query_submodules");
    mnv_check(mnv_allgather(comm, d_results, (size_t)per * stride * sizeof(float), stream), "mnv_allgather");
    return d_results;
}

void VolumeRenderer::Impl::expand_voxels(RenderOptions &o, FrameStats &st, uint64_t seed) {
    const int64_t n_px = tracker_rows();
    int32_t n = 0, n_cand = 0;
    int32_t *d_nodes = nodes.get<int32_t>((size_t)std::max(o.split_batch_size, 1) * 2);
    mnv_check(mnv_select_split_candidates(split_tracker.get<float>(n_px * 3), n_px, o.split_batch_size, d_nodes, &n, &n_cand, stream),
              "mnv_select_split_candidates");
    st.split_candidates = n_cand;
    if (n_cand == 0) {
        get_more_samples(o, st, seed);
        return;
    }
    if (n == 0) return;
    if (tree->capacity + n > max_tree_capacity) {
        st.full = true;  // "Full"
        return;
    }
    const int dim = sample_cols(o), spc = o.samples_per_corner, dd = tree->data_dim;
    const int64_t children = (int64_t)n * 8, rows = children * spc;
    float *d_rand = rand_sample.get<float>((size_t)rows * dim);
    int16_t *d_clusters = rand_clusters.get<int16_t>((size_t)rows);
    mnv_check(mnv_fill_uniform(d_rand, rows * dim, seed, stream), "mnv_fill_uniform");
    const mnv_tree_edit e = edit();
    mnv_check(mnv_add_children_and_generate_samples(&e, o.c_abi(), d_nodes, n, d_rand, dim, d_clusters, visit_tracker.get<int32_t>(max_tree_capacity),
                                                    &grid, stream),
              "mnv_add_children_and_generate_samples");
    float *d_results = query_rows(d_clusters, d_rand, dim, rows, dd + 1);
    mnv_check(mnv_apply_split_results(tree->device.data, tree->device.sample_counts, tree->capacity, n, d_results, dd + 1, spc, dd, stream),
              "mnv_apply_split_results");
    const int old_capacity = tree->capacity;
    tree->capacity += n;
    st.added = n;
    can_reuse_results = false;
    if (!accel_stale) tree->refresh_accel(old_capacity, nullptr, 0, stream);  // the packed layout follows the split
}

void VolumeRenderer::Impl::get_more_samples(RenderOptions &o, FrameStats &st, uint64_t seed) {
    const int64_t n_px = tracker_rows();
    int32_t n = 0, n_cand = 0;
    int32_t *d_nodes = nodes.get<int32_t>((size_t)std::max(o.split_batch_size, 1) * 2);
    mnv_check(mnv_select_sample_candidates(sample_tracker.get<float>(n_px * 3), n_px, o.split_batch_size, d_nodes, &n, &n_cand, stream),
              "mnv_select_sample_candidates");
    st.sample_candidates = n_cand;
    if (n == 0) return;
    // The reference draws 3 columns here whatever the model needs (cuda_renderer.cpp:298-301) while its kernel
    // writes the view-direction / embedding columns as well; the row is sized like expand_voxels' instead.
    const int dim = sample_cols(o), spc = o.samples_per_corner, dd = tree->data_dim;
    const int64_t rows = (int64_t)n * spc;
    float *d_rand = rand_sample.get<float>((size_t)rows * dim);
    int16_t *d_clusters = rand_clusters.get<int16_t>((size_t)rows);
    mnv_check(mnv_fill_uniform(d_rand, rows * dim, seed ^ 0x5a5a5a5a5a5a5a5aull, stream), "mnv_fill_uniform");
    const mnv_tree_edit e = edit();
    mnv_check(mnv_generate_samples(&e, o.c_abi(), d_nodes, n, d_rand, dim, d_clusters, &grid, stream), "mnv_generate_samples");
    float *d_results = query_rows(d_clusters, d_rand, dim, rows, dd + 1);
    mnv_check(mnv_apply_sample_results(tree->device.data, tree->device.sample_counts, d_nodes, n, d_results, dd + 1, spc, dd, stream),
              "mnv_apply_sample_results");
    st.resampled = n;
    can_reuse_results = false;
    if (!accel_stale) tree->refresh_accel(tree->capacity, d_nodes, n, stream);  // new sigma / colour rows of the resampled leaves
}

void VolumeRenderer::Impl::prune_tree(FrameStats &st) {
    const mnv_tree_edit e = edit();
    int32_t new_cap = tree->capacity, n_del = 0;
    // sample_counts is compacted with the other arrays; the reference forgets it (cuda_renderer.cpp:357-369)
    // A prune renumbers the chunks; the packed accel follows in place (mnv_prune_tree_accel: renumbered node words and lookup grids,
    // compacted colour rows), so that the next frame -- a visit-mark frame, cuda_renderer.cpp:101-102 -- runs on the tuned kernel as well.
    mnv_accel *follow = (tree->device.accel && !accel_stale && patch_after_prune) ? tree->device.accel : nullptr;
    mnv_check(mnv_prune_tree_accel(&e, tree->device.data, tree->data_dim, tree->device.sample_counts, visit_tracker.get<int32_t>(max_tree_capacity),
                                   (int32_t)max_tree_capacity, follow, &new_cap, &n_del, stream),
              "mnv_prune_tree_accel");
    st.pruned = n_del > 0 ? n_del : -1;
    if (n_del > 0) {
        tree->capacity = new_cap;
        tree_changed();
        if (follow) {
            accel_stale = false;
        } else if (tree->device.accel) {  // no patch (switched off, or the accel was stale already): rebuild in place, 2.9 ms on the 1.5 M-chunk tree
            tree->rebuild_accel(stream);
            accel_stale = false;
        }
    }
}

// cuda_renderer.cpp:144-155: what follows the march of a refinement frame
void VolumeRenderer::Impl::refine_after_frame(RenderOptions &options, FrameStats &stats, uint64_t seed, bool track_visit) {
    Impl &I = *this;
    N3Tree &tree = *I.tree;
    ++I.quiet_frames;
    // The capacity check runs only while splitting is on: the reference's unconditional
    // check would prune a tree that was merely loaded with max_tree_capacity close to its size.
    if (options.use_splitting) {
        I.expand_voxels(options, stats, seed + 0x9e3779b97f4a7c15ull * I.frame);
        if (track_visit) I.marks_fresh = true;
        I.want_marks = false;
        if (I.max_tree_capacity - tree.capacity < options.split_batch_size) {
            // The reference prunes here with whatever marks exist (cuda_renderer.cpp:148-150); with a camera that
            // has not moved since the tree was small that is no marks at all and the whole tree goes.  Here a
            // prune waits for one track_visit frame since the marks were last cleared.
            if (I.marks_fresh) {
                I.prune_tree(stats);
                I.prune_happened = true;
                I.marks_fresh = false;
            } else {
                I.want_marks = true;
                I.prune_happened = false;
            }
        } else {
            I.prune_happened = false;
        }
    }
}

VolumeRenderer::VolumeRenderer() : impl_(std::make_unique<Impl>()) {}
VolumeRenderer::~VolumeRenderer() {}

void VolumeRenderer::set(N3Tree &tree, long max_tree_capacity) {
    impl_->sync_all();
    tree.move_to_device(max_tree_capacity, true, true, impl_->stream);
    impl_->tree = &tree;
    impl_->max_tree_capacity = max_tree_capacity;
    // visit_tracker = zeros, [0] = 1 (cuda_renderer.cpp:504-506)
    int32_t *visited = impl_->visit_tracker.get<int32_t>((size_t)max_tree_capacity);
    hip_check(hipMemsetAsync(visited, 0, (size_t)max_tree_capacity * 4, impl_->stream), "clear visit marks");
    const int32_t one = 1;
    hip_check(hipMemcpyAsync(visited, &one, 4, hipMemcpyHostToDevice, impl_->stream), "mark root");
    hip_check(hipStreamSynchronize(impl_->stream), "set");
    impl_->prune_happened = impl_->can_reuse_results = impl_->accel_stale = impl_->marks_fresh = impl_->want_marks = false;
    impl_->quiet_frames = 0;
    options.basis_minmax[0] = 0;
    options.basis_minmax[1] = std::max(tree.data_format.basis_dim - 1, 0);
}

void VolumeRenderer::set_model(const mnv_mlp_desc &desc, const uint16_t *params, size_t n_halfs, const mnv_cluster_grid &grid) {
    mnv_mlp *m = nullptr;
    mnv_check(mnv_mlp_create(&desc, params, n_halfs, impl_->stream, &m), "mnv_mlp_create");
    if (impl_->mlp) mnv_mlp_destroy(impl_->mlp);
    impl_->mlp = m;
    impl_->mlp_desc = desc;
    {
        const int in_dim = 3 + 6 * desc.pos_octaves + (desc.need_viewdir ? 3 + 6 * desc.dir_octaves : 0) + (desc.n_embeddings > 0 ? desc.embedding_dim : 0);
        impl_->fused_inputs_ok = in_dim <= 64;
    }
    impl_->grid = grid;
    // cuda_renderer.cpp:534-537
    options.need_viewdir = desc.need_viewdir != 0;
    if (options.appearance_embedding == -1 && desc.n_embeddings > 0) options.appearance_embedding = 0;
    if (desc.n_embeddings <= 0) options.appearance_embedding = -1;
    impl_->can_reuse_results = false;
}

bool VolumeRenderer::has_model() const { return impl_->mlp != nullptr; }

void VolumeRenderer::load_model(const std::string &npz_path) {
    npz::Archive a = npz::load(npz_path);
    auto need = [&](const char *name) -> npz::Array & {
        auto it = a.find(name);
        if (it == a.end()) throw std::runtime_error(std::string("model container lacks '") + name + "'");
        return it->second;
    };
    auto ints = [&](const char *name, size_t count, int32_t *out) {
        npz::Array &arr = need(name);
        if (arr.num_vals() != count || (arr.kind != 'i' && arr.kind != 'u')) throw std::runtime_error(std::string(name) + " has an unexpected shape");
        for (size_t i = 0; i < count; ++i) out[i] = arr.word_size == 8 ? (int32_t)arr.data<int64_t>()[i] : arr.data<int32_t>()[i];
    };
    auto floats = [&](const char *name, size_t count, float *out) {
        npz::Array &arr = need(name);
        if (arr.num_vals() != count || arr.kind != 'f') throw std::runtime_error(std::string(name) + " has an unexpected shape");
        for (size_t i = 0; i < count; ++i) out[i] = arr.word_size == 8 ? (float)arr.data<double>()[i] : arr.data<float>()[i];
    };
    int32_t d[9];
    ints("mlp_desc", 9, d);
    mnv_mlp_desc desc{};
    desc.n_clusters = d[0];
    desc.pos_octaves = d[1];
    desc.dir_octaves = d[2];
    desc.need_viewdir = d[3];
    desc.n_embeddings = d[4];
    desc.embedding_dim = d[5];
    desc.hidden_width = d[6];
    desc.hidden_layers = d[7];
    desc.out_dim = d[8];
    floats("mlp_center", 3, desc.center);
    floats("mlp_inv_extent", 3, desc.inv_extent);
    mnv_cluster_grid grid{};
    ints("grid_dim", 2, grid.grid_dim);
    float max_position[3];
    floats("min_position", 3, grid.min_position);
    floats("max_position", 3, max_position);
    for (int i = 0; i < 3; ++i) grid.range[i] = max_position[i] - grid.min_position[i];  // cuda_renderer.cpp:527
    npz::Array &p = need("mlp_params");
    if (p.word_size != 2) throw std::runtime_error("mlp_params must be binary16");
    set_model(desc, p.data<uint16_t>(), p.num_vals(), grid);
}

void VolumeRenderer::clear() { impl_->tree = nullptr; }

const mnv_mlp *VolumeRenderer::model() const { return impl_->mlp; }
const mnv_cluster_grid &VolumeRenderer::cluster_grid() const { return impl_->grid; }

void VolumeRenderer::set_frame_inputs(const float *tmax_px_device, const uint8_t *rgba8_init_device) {
    Impl &I = *impl_;
    if (I.comm && (tmax_px_device || rgba8_init_device)) throw std::runtime_error("frame inputs (offscreen == false) are for one rank");
    I.sync_all();
    I.inputs.tmax_px = tmax_px_device;
    I.inputs.rgba8_init = rgba8_init_device;
    I.can_reuse_results = false;  // the samples of the last guided frame were limited by the previous depth image
}

void VolumeRenderer::set_ranks(mnv_comm *comm, int tile_w, int tile_h) {
    Impl &I = *impl_;
    if (comm && (I.inputs.tmax_px || I.inputs.rgba8_init)) throw std::runtime_error("frame inputs (offscreen == false) are for one rank");
    I.sync_all();
    I.comm = comm;
    if (!comm) {
        I.part = {0, 1, 0, 0, 0};
        return;
    }
    I.part = {mnv_comm_rank(comm), mnv_comm_world(comm), tile_w, tile_h, 0};
}

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
    impl_->ensure_slots(1);
    impl_->use_slot(0);
}

void VolumeRenderer::render() {
    Impl &I = *impl_;
    if (!I.slots[0].rgba) resize(camera.width, camera.height);
    camera._update();
    const mnv_camera cv = camera.c_abi();
    const mnv_rect full = {0, 0, I.width, I.height};
    stats = FrameStats();
    {
        // a plain frame of a tree with a current accel takes the next slot; everything else runs alone on slot 0
        if (overlaps_next()) {
            // Slot streams are not ordered among each other.  Whatever slot 0's stream still holds from a frame that changed the tree
            // or its accel (splits, resampled leaves, accel patches, a prune's renumbering: asynchronous on that stream) must be
            // finished before a frame on another stream reads those arrays: once, on the way from such frames to plain ones.
            if (I.tree_stream_dirty) {
                hip_check(hipStreamSynchronize(I.stream), "hipStreamSynchronize");
                I.tree_stream_dirty = false;
            }
            I.ensure_slots(frames_in_flight);
            I.use_slot((I.cur + 1) % frames_in_flight);
            I.overlapped = true;
            Impl::Slot &S = I.slots[I.cur];
            if (I.mlp != nullptr && options.use_guided_sampling) {  // (overlaps_next: a guided-sampling frame that changes nothing, guided_in_flight)
                if (!S.count_dev) {
                    hip_check(hipMalloc((void **)&S.count_dev, sizeof(unsigned long long)), "hipMalloc(sample counter)");
                    hip_check(hipHostMalloc((void **)&S.count_host, sizeof(unsigned long long), hipHostMallocDefault), "hipHostMalloc(sample count)");
                }
                hip_check(hipMemsetAsync(S.count_dev, 0, sizeof(unsigned long long), S.stream), "clear sample counter");
                mnv_check(mnv_render_guided_fused_track_ex(I.tree->device.accel, &cv, options.c_abi(), full, &I.inputs, I.mlp, &I.grid, I.rgba, I.rgba8, nullptr,
                                                           nullptr, nullptr, nullptr, nullptr, S.count_dev, S.stream),
                          "mnv_render_guided_fused_track_ex");
                hip_check(hipMemcpyAsync(S.count_host, S.count_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, S.stream), "read sample counter");
                S.counted = true;
                stats.fused = true;
                stats.guided_samples = -1;  // in flight: slot_guided_samples(last_slot())
                ++I.quiet_frames;           // (what refine_after_frame does for a frame without splitting)
            } else {
                S.counted = false;
                mnv_check(mnv_render_voxels_accel_ex(I.tree->device.accel, &cv, options.c_abi(), full, &I.inputs, I.rgba, I.rgba8, S.stream),
                          "mnv_render_voxels_accel_ex");
            }
            stats.used_accel = true;
            stats.capacity = I.tree->capacity;
            ++I.frame;
            return;
        }
        if (I.overlapped) I.sync_all();
        I.use_slot(0);
        I.slots[0].counted = false;  // (this frame's sample count is in `stats`)
        I.tree_stream_dirty = true;  // what follows runs on slot 0's stream and may edit the tree
    }
    if (I.comm) {
        render_ranks();
        return;
    }
    if (I.tree == nullptr || I.tree->N <= 0) {
        mnv_tree_view empty = {};  // N == 0: background only (renderer_kernel.cu:266)
        mnv_check(mnv_render_voxels_ex(&empty, &cv, options.c_abi(), full, &I.inputs, I.rgba, I.rgba8, nullptr, nullptr, nullptr, 0, I.stream),
                  "mnv_render_voxels_ex");
        return;
    }
    N3Tree &tree = *I.tree;
    const int64_t n_px = (int64_t)I.width * I.height;
    const bool refine = I.mlp != nullptr && (options.use_splitting || options.use_guided_sampling);
    if (refine && I.mlp_desc.out_dim != tree.data_dim + 1)
        throw std::runtime_error("the model's out_dim must be the tree's data_dim + 1 (cuda_renderer.cpp:255-257)");

    // cuda_renderer.cpp:98-107
    const bool camera_has_changed = camera.has_changed();
    const bool track_visit = refine && ((camera_has_changed && tree.capacity > I.max_tree_capacity * 3 / 4) || I.prune_happened || I.want_marks);
    if (camera_has_changed) I.can_reuse_results = false;
    stats.track_visit = track_visit;
    float *split = nullptr, *sample = nullptr;
    int32_t *visited = I.visit_tracker.get<int32_t>((size_t)std::max<long>(I.max_tree_capacity, 1));
    if (refine && options.use_splitting) {
        split = I.split_tracker.get<float>(n_px * 3);
        sample = I.sample_tracker.get<float>(n_px * 3);
    }
    // splits and resampled leaves are patched into the packed accel as they happen (mnv_accel_refresh); a prune renumbers
    // the chunks, after which the march reads the reference-layout arrays until the accel has been rebuilt -- once the
    // tree has been left alone for accel_rebuild_after frames
    if (I.accel_stale && I.quiet_frames >= accel_rebuild_after) {
        tree.rebuild_accel(I.stream);
        I.accel_stale = false;
    }
    const mnv_tree_view dv = tree.device_view();

    // the guided-sampling frame as ONE kernel (march + networks + composite [+ trackers + visit marks], no sample buffer) whenever
    // the accel is current and the network is one the fused kernel covers
    const bool fuse = refine && options.use_guided_sampling && use_fused_guided && tree.device.accel && !I.accel_stale &&
                      (!track_visit || tree.device.parent) && !options.render_depth && I.mlp_desc.hidden_width == 64 && I.fused_inputs_ok &&
                      (tree.data_format.format != DataFormat::SH || tree.data_format.basis_dim == 1 || tree.data_format.basis_dim == 4 ||
                       tree.data_format.basis_dim == 9 || tree.data_format.basis_dim == 16);
    if (split) {
        // cuda_renderer.cpp:97-98.  The fused kernel writes all three words of both rows for every pixel it renders, so a healthy fused
        // frame does not need the fill (50 us of a 3 ms configs[4] frame) -- but a wavefront that its watchdog abandoned (MNV_E_FAULT, reported
        // by the NEXT fused call) leaves its pixels' rows unwritten, and refine_after_frame edits the tree from these rows in this same
        // frame: stale (chunk, child) pairs of an earlier frame may name chunks a prune has renumbered since.  -1 rows name nothing.
        I.fill_f32(split, n_px * 3, -1.f);
        I.fill_f32(sample, n_px * 3, -1.f);
    }
    if (fuse) {
        unsigned long long *counter = I.fused_counter.get<unsigned long long>(1);
        hip_check(hipMemsetAsync(counter, 0, sizeof(unsigned long long), I.stream), "clear sample counter");
        // with refinement on as well (configs[4]) the same kernel also writes the trackers and the visit marks
        mnv_check(mnv_render_guided_fused_track_ex(tree.device.accel, &cv, options.c_abi(), full, &I.inputs, I.mlp, &I.grid, I.rgba, I.rgba8, split, sample,
                                                   split ? tree.device.sample_counts : nullptr, track_visit ? visited : nullptr, tree.device.parent,
                                                   counter, I.stream),
                  "mnv_render_guided_fused_track_ex");
        I.read_count_later(counter);
        stats.used_accel = true;
        stats.fused = true;
        I.can_reuse_results = false;
    } else if (refine && options.use_guided_sampling) {
        // cuda_renderer.cpp:109-139
        const int samples_dim = 1 + I.sample_cols(options), max_g = options.max_guided_samples, dd = tree.data_dim;
        int64_t *offsets = I.offsets.get<int64_t>(n_px);
        if (!I.can_reuse_results) {
            int16_t *num = I.num_samples.get<int16_t>(n_px);
            int16_t *clusters = I.cluster_indices.get<int16_t>(n_px * max_g);
            float *guided = I.guided_samples.get<float>((size_t)n_px * max_g * samples_dim);
            hip_check(hipMemsetAsync(num, 0, n_px * 2, I.stream), "clear num_samples");
            if (tree.device.accel && !I.accel_stale && (!track_visit || tree.device.parent)) {
                // visit marks on the packed layout: the march marks leaf chunks, a closure pass adds their ancestors
                stats.used_accel = true;
                mnv_check(mnv_get_samples_from_voxels_accel_visit_ex(tree.device.accel, &cv, options.c_abi(), full, &I.inputs, split, sample,
                                                                     tree.device.sample_counts, track_visit ? visited : nullptr, tree.device.parent, num, guided,
                                                                     samples_dim, clusters, &I.grid, I.stream),
                          "mnv_get_samples_from_voxels_accel_visit_ex");
            } else {
                mnv_check(mnv_get_samples_from_voxels_ex(&dv, &cv, options.c_abi(), full, &I.inputs, split, sample, visited, track_visit, num, guided,
                                                         samples_dim, clusters, &I.grid, I.stream),
                          "mnv_get_samples_from_voxels_ex");
            }
            // one call in the steady state: the packed buffers keep the size of the previous frames and grow when a frame
            // emits more (the call then reports the total it needs)
            int64_t total = 0, cap_rows = (int64_t)(I.z_vals.bytes / sizeof(float));
            float *z = nullptr, *rows = nullptr, *values = nullptr;
            int16_t *row_clusters = nullptr;
            for (int attempt = 0; attempt < 2; ++attempt) {
                z = I.z_vals.get<float>((size_t)std::max<int64_t>(cap_rows, 1));
                rows = I.sample_rows.get<float>((size_t)std::max<int64_t>(cap_rows, 1) * (samples_dim - 1));
                row_clusters = I.sample_clusters.get<int16_t>((size_t)std::max<int64_t>(cap_rows, 1));
                const int rc = mnv_compact_guided_samples(num, guided, clusters, n_px, max_g, samples_dim, offsets, z, rows, row_clusters, cap_rows, &total,
                                                          I.stream);
                if (rc == MNV_OK) break;
                if (total <= cap_rows || attempt == 1) mnv_check(rc, "mnv_compact_guided_samples");
                cap_rows = total + total / 8;
            }
            values = I.nerf_results.get<float>((size_t)std::max<int64_t>(total, 1) * (dd + 1));
            if (total > 0)
                mnv_check(mnv_query_submodules(I.mlp, row_clusters, rows, samples_dim - 1, total, values, dd + 1, I.stream), "mnv_query_submodules");
            I.reuse_total = total;
            I.can_reuse_results = true;
        }
        stats.guided_samples = I.reuse_total;
        mnv_check(mnv_render_nerf_results(&dv, &cv, options.c_abi(), full, I.nerf_results.get<float>(1), tree.data_dim + 1, I.z_vals.get<float>(1), offsets,
                                          I.rgba, I.rgba8, I.stream),
                  "mnv_render_nerf_results");
    } else if (tree.device.accel && !I.accel_stale && (!track_visit || tree.device.parent)) {
        stats.used_accel = true;
        mnv_check(mnv_render_voxels_accel_visit_ex(tree.device.accel, &cv, options.c_abi(), full, &I.inputs, I.rgba, I.rgba8, split, sample,
                                                   tree.device.sample_counts, track_visit ? visited : nullptr, tree.device.parent, I.stream),
                  "mnv_render_voxels_accel_visit_ex");
    } else {
        mnv_check(mnv_render_voxels_ex(&dv, &cv, options.c_abi(), full, &I.inputs, I.rgba, I.rgba8, split, sample, visited, track_visit, I.stream),
                  "mnv_render_voxels_ex");
    }

    if (refine) I.refine_after_frame(options, stats, seed, track_visit);
    I.finish_count(stats);
    ++I.frame;
    stats.capacity = tree.capacity;
}

// render() of one rank of several (set_ranks)
void VolumeRenderer::render_ranks() {
    Impl &I = *impl_;
    if (I.tree == nullptr || I.tree->N <= 0 || !I.tree->device.accel)
        throw std::runtime_error("several ranks need a tree with the packed accel (N == 2, RGBA or SH1/4/9/16/25 rows)");
    N3Tree &tree = *I.tree;
    const mnv_camera cv = camera.c_abi();
    const mnv_rect full = {0, 0, I.width, I.height};
    const int world = I.part.world, rank = I.part.rank;
    int32_t j_max = 0;
    for (int r = 0; r < world; ++r) {
        const mnv_partition pr = {r, world, I.part.tile_w, I.part.tile_h, 0};
        j_max = std::max(j_max, mnv_partition_local_tiles(full, pr));
    }
    I.rank_px = (int64_t)j_max * I.part.tile_w * I.part.tile_h;
    const int64_t rows = I.rank_px * world;
    const bool refine = I.mlp != nullptr && (options.use_splitting || options.use_guided_sampling);
    if (refine && I.mlp_desc.out_dim != tree.data_dim + 1)
        throw std::runtime_error("the model's out_dim must be the tree's data_dim + 1 (cuda_renderer.cpp:255-257)");
    if (options.render_depth && refine) throw std::runtime_error("several ranks: no depth mode while refining");

    // cuda_renderer.cpp:98-107 -- the same decisions on every rank: they depend on the camera and on the (replicated) tree only
    const bool camera_has_changed = camera.has_changed();
    const bool track_visit = refine && ((camera_has_changed && tree.capacity > I.max_tree_capacity * 3 / 4) || I.prune_happened || I.want_marks);
    if (camera_has_changed) I.can_reuse_results = false;
    stats.track_visit = track_visit;
    float *local = I.local.get<float>((size_t)I.rank_px * 4);
    uint8_t *local8 = I.local8.get<uint8_t>((size_t)I.rank_px * 4);
    float *split = nullptr, *sample = nullptr;  // [world][rank_px][3]: block r holds rank r's rows
    int32_t *visited = I.visit_tracker.get<int32_t>((size_t)std::max<long>(I.max_tree_capacity, 1));
    if (refine && options.use_splitting) {
        split = I.split_tracker.get<float>(rows * 3);
        sample = I.sample_tracker.get<float>(rows * 3);
        I.fill_f32(split, rows * 3, -1.f);
        I.fill_f32(sample, rows * 3, -1.f);
    }
    float *my_split = split ? split + (size_t)rank * I.rank_px * 3 : nullptr, *my_sample = sample ? sample + (size_t)rank * I.rank_px * 3 : nullptr;
    if (I.accel_stale) {  // every frame of a rank runs on the packed layout
        tree.rebuild_accel(I.stream);
        I.accel_stale = false;
    }
    if (track_visit && !tree.device.parent) throw std::runtime_error("several ranks: visit marks need the tree's parent array");
    stats.used_accel = true;
    if (refine && options.use_guided_sampling) {
        const bool fits = I.mlp_desc.hidden_width == 64 && I.fused_inputs_ok &&
                          (tree.data_format.format != DataFormat::SH || tree.data_format.basis_dim == 1 || tree.data_format.basis_dim == 4 ||
                           tree.data_format.basis_dim == 9 || tree.data_format.basis_dim == 16);
        if (!fits) throw std::runtime_error("several ranks: guided sampling needs a network the fused kernel covers (64 wide, at most 64 encoded inputs)");
        unsigned long long *counter = I.fused_counter.get<unsigned long long>(1);
        hip_check(hipMemsetAsync(counter, 0, sizeof(unsigned long long), I.stream), "clear sample counter");
        mnv_check(mnv_render_guided_fused_track_part(tree.device.accel, &cv, options.c_abi(), full, I.part, I.mlp, &I.grid, local, local8, my_split, my_sample,
                                                     split ? tree.device.sample_counts : nullptr, track_visit ? visited : nullptr, tree.device.parent, counter,
                                                     I.stream),
                  "mnv_render_guided_fused_track_part");
        I.read_count_later(counter);  // (this rank's share)
        stats.fused = true;
    } else {
        mnv_check(mnv_render_voxels_accel_visit_part(tree.device.accel, &cv, options.c_abi(), full, I.part, local, local8, my_split, my_sample,
                                                     tree.device.sample_counts, track_visit ? visited : nullptr, tree.device.parent, I.stream),
                  "mnv_render_voxels_accel_visit_part");
    }
    // the picture: tiles to rank 0, un-permuted there
    float *table = rank == 0 ? I.table.get<float>((size_t)rows * 4) : nullptr;
    uint8_t *table8 = rank == 0 ? I.table8.get<uint8_t>((size_t)rows * 4) : nullptr;
    mnv_check(mnv_gather_tiles(I.comm, local, table, (size_t)I.rank_px * 16, 0, I.stream), "mnv_gather_tiles");
    mnv_check(mnv_gather_tiles(I.comm, local8, table8, (size_t)I.rank_px * 4, 0, I.stream), "mnv_gather_tiles");
    if (rank == 0) {
        mnv_check(mnv_assemble_tiles(table, I.rgba, I.width, I.height, I.part, 1, 16, I.stream), "mnv_assemble_tiles");
        mnv_check(mnv_assemble_tiles(table8, I.rgba8, I.width, I.height, I.part, 1, 4, I.stream), "mnv_assemble_tiles");
    }
    // the votes need every rank's tracker rows, a prune every rank's marks
    if (split) {
        mnv_check(mnv_allgather(I.comm, split, (size_t)I.rank_px * 12, I.stream), "mnv_allgather");
        mnv_check(mnv_allgather(I.comm, sample, (size_t)I.rank_px * 12, I.stream), "mnv_allgather");
    }
    if (track_visit && world > 1) {
        // the marks of the chunks that exist: the replicas agree on tree.capacity, and a chunk beyond it cannot be marked (the table is
        // sized once for the largest tree -- reallocation would stall the frame -- but only cap words per rank travel and merge)
        const size_t cap = (size_t)tree.capacity;
        int32_t *marks = I.marks_table.get<int32_t>((size_t)I.max_tree_capacity * world);
        hip_check(hipMemcpyAsync(marks + cap * rank, visited, cap * 4, hipMemcpyDeviceToDevice, I.stream), "copy marks");
        mnv_check(mnv_allgather(I.comm, marks, cap * 4, I.stream), "mnv_allgather");
        mnv_check(mnv_merge_visit_marks(marks, world, (int32_t)cap, visited, I.stream), "mnv_merge_visit_marks");
    }
    if (refine) I.refine_after_frame(options, stats, seed, track_visit);
    I.finish_count(stats);
    ++I.frame;
    stats.capacity = tree.capacity;
}

const char *VolumeRenderer::get_backend() { return "HIP gfx950"; }

void VolumeRenderer::download(std::vector<float> *rgba, std::vector<uint8_t> *rgba8) {
    download_slot(impl_->cur, rgba, rgba8);
}

int VolumeRenderer::last_slot() const { return impl_->cur; }

bool VolumeRenderer::overlaps_next() const {
    const Impl &I = *impl_;
    if (I.comm || frames_in_flight <= 1 || I.tree == nullptr || I.tree->N <= 0 || !I.tree->device.accel || I.accel_stale) return false;
    const bool refine_now = I.mlp != nullptr && (options.use_splitting || options.use_guided_sampling);
    if (!refine_now) return true;  // a plain frame on the packed accel
    // a guided-sampling frame that changes nothing and wants nothing but the picture: the fused kernel, no splitting, no visit marks (a tree
    // below 3/4 of its capacity and no prune pending: render()'s track_visit is then false whatever the camera did)
    const N3Tree &tree = *I.tree;
    const bool covered = I.mlp_desc.hidden_width == 64 && I.fused_inputs_ok && I.mlp_desc.out_dim == tree.data_dim + 1 &&
                         (tree.data_format.format != DataFormat::SH || tree.data_format.basis_dim == 1 || tree.data_format.basis_dim == 4 ||
                          tree.data_format.basis_dim == 9 || tree.data_format.basis_dim == 16);
    return guided_in_flight && use_fused_guided && options.use_guided_sampling && !options.use_splitting && !options.render_depth && covered &&
           !I.prune_happened && !I.want_marks && tree.capacity <= I.max_tree_capacity * 3 / 4;
}

long VolumeRenderer::slot_guided_samples(int slot) {
    Impl &I = *impl_;
    if (slot < 0 || slot >= (int)I.slots.size()) throw std::runtime_error("slot_guided_samples: no such frame slot");
    Impl::Slot &S = I.slots[slot];
    if (!S.counted) return slot == I.cur ? stats.guided_samples : 0;
    hip_check(hipStreamSynchronize(S.stream), "slot_guided_samples");
    return (long)*S.count_host;
}

int VolumeRenderer::next_slot() const { return overlaps_next() ? (impl_->cur + 1) % frames_in_flight : 0; }

void VolumeRenderer::download_slot(int slot, std::vector<float> *rgba, std::vector<uint8_t> *rgba8) {
    if (slot < 0 || slot >= (int)impl_->slots.size() || !impl_->slots[slot].rgba) throw std::runtime_error("download_slot: no such frame slot");
    const size_t n = (size_t)impl_->width * impl_->height * 4;
    hip_check(hipStreamSynchronize(impl_->slots[slot].stream), "render");
    if (rgba) {
        rgba->resize(n);
        hip_check(hipMemcpy(rgba->data(), impl_->slots[slot].rgba, n * sizeof(float), hipMemcpyDeviceToHost), "download rgba");
    }
    if (rgba8) {
        rgba8->resize(n);
        hip_check(hipMemcpy(rgba8->data(), impl_->slots[slot].rgba8, n, hipMemcpyDeviceToHost), "download rgba8");
    }
}

void VolumeRenderer::sync_tree_streams() { impl_->sync_all(); }

void VolumeRenderer::sync_tree() {
    impl_->sync_all();
    if (impl_->tree) impl_->tree->copy_from_device(impl_->stream);
}

const float *VolumeRenderer::device_rgba() const { return impl_->rgba; }
const uint8_t *VolumeRenderer::device_rgba8() const { return impl_->rgba8; }

}  // namespace viewer
