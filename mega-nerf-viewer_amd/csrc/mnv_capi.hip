// mnv_capi.hip -- the extern "C" entry points of include/mnv.h that touch the device.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "mnv_internal.h"

namespace mnv {

static thread_local std::string g_last_error;

int set_error(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return MNV_OK;
    return set_error((int)e, std::string(what) + ": " + hipGetErrorString(e));
}

// ---- argument block ----------------------------------------------------------------
void fill_camera(CamBlock &C, const mnv_camera *cam);
int fill_params(FrameParams &P, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile) {
    if (!cam || !opt) return set_error(MNV_E_INVALID, "camera/options pointer is null");
    if (cam->width <= 0 || cam->height <= 0) return set_error(MNV_E_INVALID, "camera has no pixels");
    if (tile.w < 0 || tile.h < 0) return set_error(MNV_E_INVALID, "negative tile extent");
    fill_camera(P.cam, cam);
    P.x0 = tile.x0;
    P.y0 = tile.y0;
    P.tw = tile.w;
    P.th = tile.h;
    P.step_size = opt->step_size;
    P.sigma_thresh = opt->sigma_thresh;
    P.stop_thresh = opt->stop_thresh;
    P.background_brightness = opt->background_brightness;
    std::memcpy(P.render_bbox, opt->render_bbox, sizeof(P.render_bbox));
    P.basis_min = opt->basis_minmax[0];
    P.basis_max = opt->basis_minmax[1];
    P.render_depth = opt->render_depth ? 1 : 0;
    // rodrigues(opt.rot_dirs, .) frame constants, reference renderer_kernel.cu:43-51
    const float *aa = opt->rot_dirs;
    const float angle = sqrtf(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
    P.rot_enabled = !((double)angle < 1e-6);
    P.rot_k[0] = P.rot_k[1] = P.rot_k[2] = 0.f;
    P.rot_cos = 1.f;
    P.rot_sin = 0.f;
    if (P.rot_enabled) {
        for (int i = 0; i < 3; ++i) P.rot_k[i] = aa[i] / angle;
        P.rot_cos = cosf(angle);
        P.rot_sin = sinf(angle);
    }
    return MNV_OK;
}

// renderer_kernel.cu:272-275: cen = offset + scale * c2w[9..11]; identical for every ray, so it is
// computed once on the host (this file is compiled with -ffp-contract=off: mul then add).
void fill_origin(CamBlock &C, const float offset[3], const float scale[3]) {
    for (int i = 0; i < 3; ++i) {
        const float prod = scale[i] * C.c2w[9 + i];
        C.cen[i] = offset[i] + prod;
    }
    C.pad = 0.f;
}

void fill_camera(CamBlock &C, const mnv_camera *cam) {
    C.fx = cam->fx;
    C.fy = cam->fy;
    C.cx = cam->cx;
    C.cy = cam->cy;
    std::memcpy(C.c2w, cam->c2w, sizeof(C.c2w));
}

int fill_tree_params(MarchParams &P, const mnv_tree_view *t) {
    if (!t) return set_error(MNV_E_INVALID, "tree view is null");
    if (t->N > 16) return set_error(MNV_E_UNSUPPORTED, "branching factors above 16 per axis are not supported");
    if (t->N > 0 && (!t->data || !t->child)) return set_error(MNV_E_INVALID, "tree arrays are null");
    if (t->N > 0 && t->data_dim < 1) return set_error(MNV_E_INVALID, "data_dim < 1");
    if (t->N > 0 && t->format == MNV_FORMAT_SH && t->basis_dim >= 0 && 3 * t->basis_dim + 1 > t->data_dim)
        return set_error(MNV_E_INVALID, "data_dim too small for 3 * basis_dim + sigma");
    if (t->N > 0 && (t->format != MNV_FORMAT_SH || t->basis_dim < 0) && t->data_dim < 4)
        return set_error(MNV_E_INVALID, "RGBA rows need data_dim >= 4");
    P.data = t->data;
    P.child = t->child;
    P.sample_counts = t->sample_counts;
    std::memcpy(P.offset, t->offset, sizeof(P.offset));
    std::memcpy(P.scale, t->scale, sizeof(P.scale));
    fill_origin(P.cam, P.offset, P.scale);
    P.data_dim = t->data_dim;
    P.basis_dim = t->basis_dim;
    P.format = t->format;
    P.capacity = t->capacity;
    P.N = t->N;
    return MNV_OK;
}

__global__ void fill_background_kernel(const FrameParams P) {
    // tree.N <= 0: "draw nothing" (renderer_kernel.cu:266-269) -> background only
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)P.tw * P.th) return;
    composite_and_write(P, p, 0.f, 0.f, 0.f, 0.f);
}

int launch_background(const FrameParams &P, hipStream_t stream) {
    const int64_t n = (int64_t)P.tw * P.th;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(fill_background_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, P);
    return (int)hipGetLastError();
}

}  // namespace mnv

using namespace mnv;

extern "C" {

int mnv_version(void) { return MNV_VERSION; }

const char *mnv_last_error(void) { return g_last_error.c_str(); }

int mnv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mnv_stream_create_reserved(int32_t reserve_cus, void **stream_out, int32_t *enabled_cus) {
    if (!stream_out) return set_error(MNV_E_INVALID, "stream_out is null");
    int dev = 0;
    int rc = check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    hipDeviceProp_t prop;
    if ((rc = check_hip(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties"))) return rc;
    const int n = prop.multiProcessorCount;
    if (reserve_cus < 0 || reserve_cus >= n) return set_error(MNV_E_INVALID, "reserve_cus must be in [0, compute units of the device)");
    hipStream_t s = nullptr;
    if (reserve_cus == 0) {
        if ((rc = check_hip(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreateWithFlags"))) return rc;
    } else {
        // low (n - reserve) bits: on gfx942 / gfx950 mask bit i enables a unit of XCD i % 8 and, within the XCD, of shader
        // engine (i / 8) % 4 (tools/cumask/map_bits.py), so a multiple of 32 removes one unit from every shader engine
        uint32_t mask[32] = {0};
        const int keep = n - reserve_cus;
        if (n > 1024) return set_error(MNV_E_UNSUPPORTED, "more than 1024 compute units");
        for (int i = 0; i < keep; ++i) mask[i >> 5] |= 1u << (i & 31);
        if ((rc = check_hip(hipExtStreamCreateWithCUMask(&s, (uint32_t)((n + 31) / 32), mask), "hipExtStreamCreateWithCUMask"))) return rc;
    }
    *stream_out = (void *)s;
    if (enabled_cus) *enabled_cus = n - reserve_cus;
    return MNV_OK;
}

int mnv_stream_destroy(void *stream) { return check_hip(hipStreamDestroy((hipStream_t)stream), "hipStreamDestroy"); }

// ---- mnv_set_tree_cache: the stateless entry point with a memory (see include/mnv.h) ------------------------------------------------
namespace {
struct CachedTree {
    const void *child = nullptr, *data = nullptr;
    int32_t capacity = 0, data_dim = 0, basis_dim = 0, format = 0;
    int device = 0;
    bool used = false;            // the entry names a tree
    mnv_accel *accel = nullptr;   // NULL in a used entry: the re-layout of this tree could not be built (deeper than the packed layout goes,
                                  // out of memory) -- remembered, so that the build is not retried with every frame (mnv_tree_invalidate forgets)
    hipEvent_t built = nullptr;   // recorded on the stream that built the re-layout; launches on other streams wait for it
    hipStream_t build_stream = nullptr;
    uint64_t stamp = 0;
};
constexpr int kCachedTrees = 4;
std::mutex g_cache_mu;
CachedTree g_cache[kCachedTrees];
std::atomic<int> g_cache_on{0};
uint64_t g_cache_clock = 0;

void drop(CachedTree &e) {
    if (e.accel) {
        int cur = 0;
        const bool other = hipGetDevice(&cur) == hipSuccess && cur != e.device && hipSetDevice(e.device) == hipSuccess;
        (void)hipDeviceSynchronize();  // frames that still read the re-layout (launched under g_cache_mu, so all of them are queued by now)
        mnv_accel_destroy(e.accel);
        if (other) (void)hipSetDevice(cur);
    }
    if (e.built) (void)hipEventDestroy(e.built);
    e = CachedTree();
}
}  // namespace

void mnv_set_tree_cache(int enable) {
    g_cache_on.store(enable ? 1 : 0, std::memory_order_relaxed);
    if (!enable) mnv_tree_invalidate(nullptr);
}

void mnv_tree_invalidate(const void *child) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (auto &e : g_cache)
        if (e.used && (!child || e.child == child)) drop(e);
}

// the cached re-layout of `tree` (built on `stream` at the first call), or NULL when this frame must take the stateless path.
// Called with g_cache_mu held, and the caller launches its frame before releasing it: an eviction on another thread (drop: device-wide
// wait, then destroy) can then never fall between the look-up and the launch.  The key is the arrays' identity (addresses, capacity, row
// format, device); offset and scale are NOT part of it -- the frame takes them from the call's tree view (render_accel_for_tree), so the
// same arrays under another transform render with that transform.
static const mnv_accel *cached_accel(const mnv_tree_view *tree, hipStream_t stream) {
    const int b = (tree->format == MNV_FORMAT_SH && tree->basis_dim >= 0) ? tree->basis_dim : -1;
    if (!(b == -1 || b == 1 || b == 4 || b == 9 || b == 16 || b == 25)) return nullptr;  // what the packed layout has rows for
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    CachedTree *hit = nullptr, *victim = nullptr;
    for (auto &e : g_cache)
        if (e.used && e.child == tree->child && e.data == tree->data && e.capacity == tree->capacity && e.data_dim == tree->data_dim &&
            e.basis_dim == tree->basis_dim && e.format == tree->format && e.device == dev)
            hit = &e;
    if (!hit) {
        // build first, evict afterwards: a tree whose re-layout cannot be built must not cost another tree its entry
        mnv_accel *a = nullptr;
        hipEvent_t ev = nullptr;
        if (mnv_accel_create(tree, (void *)stream, &a) != MNV_OK) a = nullptr;  // e.g. deeper than the packed layout goes: stateless path
        if (a && (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, stream) != hipSuccess)) {
            mnv_accel_destroy(a);
            if (ev) (void)hipEventDestroy(ev);
            return nullptr;
        }
        for (auto &e : g_cache)  // a free entry, else a remembered failure, else the least recently used one
            if (!victim || (victim->used && (!e.used || (!e.accel && victim->accel) || ((!e.accel == !victim->accel) && e.stamp < victim->stamp)))) victim = &e;
        if (!a && victim->used && victim->accel) return nullptr;  // nothing but live re-layouts to evict: do not remember the failure at their cost
        if (victim->used) drop(*victim);
        victim->used = true;
        victim->child = tree->child; victim->data = tree->data; victim->capacity = tree->capacity; victim->data_dim = tree->data_dim;
        victim->basis_dim = tree->basis_dim; victim->format = tree->format; victim->device = dev;
        victim->accel = a; victim->built = ev; victim->build_stream = stream;
        hit = victim;
    }
    hit->stamp = ++g_cache_clock;
    if (!hit->accel) return nullptr;  // remembered failure
    if (hit->build_stream != stream && hipStreamWaitEvent(stream, hit->built, 0) != hipSuccess) return nullptr;
    return hit->accel;
}

int mnv_render_voxels(const mnv_tree_view *tree, const mnv_camera *cam, const mnv_render_options *opt,
                      mnv_rect tile, float *rgba_out, uint8_t *rgba8_out, float *split_track,
                      float *sample_track, int32_t *visited, int track_visit, void *hip_stream) {
    return mnv_render_voxels_ex(tree, cam, opt, tile, nullptr, rgba_out, rgba8_out, split_track, sample_track, visited, track_visit, hip_stream);
}

int mnv_render_voxels_ex(const mnv_tree_view *tree, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                         const mnv_frame_inputs *inputs, float *rgba_out, uint8_t *rgba8_out, float *split_track, float *sample_track,
                         int32_t *visited, int track_visit, void *hip_stream) {
    MarchParams P;
    std::memset(static_cast<void *>(&P), 0, sizeof(P));
    int rc = fill_params(P, cam, opt, tile);
    if (rc) return rc;
    rc = fill_tree_params(P, tree);
    if (rc) return rc;
    if (track_visit && !visited) return set_error(MNV_E_INVALID, "track_visit set but visited is null");
    hipStream_t stream = (hipStream_t)hip_stream;
    if (g_cache_on.load(std::memory_order_relaxed) && tree->N == 2 && !track_visit) {
        // frames of a tree this process has seen before run on the packed re-layout kept from that call (mnv_set_tree_cache) -- with the
        // refinement trackers as well (the reference passes them with every call); visit marks need the parent array and walk the arrays
        std::lock_guard<std::mutex> lk(g_cache_mu);
        if (const mnv_accel *a = cached_accel(tree, stream))
            return render_accel_for_tree(a, tree, cam, opt, tile, inputs, rgba_out, rgba8_out, split_track, sample_track, stream);
    }
    P.max_depth = opt->max_depth;
    P.max_sample_count = opt->max_sample_count;
    P.rgba = rgba_out;
    P.rgba8 = rgba8_out;
    if (inputs) {
        P.tmax_px = inputs->tmax_px;
        P.rgba8_init = inputs->rgba8_init;
    }
    P.split_track = split_track;
    P.sample_track = sample_track;
    P.visited = visited;
    P.track_visit = track_visit ? 1 : 0;
    if (tree->N <= 0) return check_hip((hipError_t)launch_background(P, stream), "fill_background_kernel");
    return check_hip((hipError_t)launch_ref_layout(P, stream), "march_ref_layout_kernel");
}

}  // extern "C"
