// mnv_render -- offline batch renderer: the reference's `nerf-viewer` command line without the window.
//
// Flag names and defaults follow the reference (src/opts.cpp:17-32 common flags, main.cpp:491-505
// viewer flags); --grid (wireframe overlay) is accepted and ignored.  --model_path names a model container
// (an .npz, see VolumeRenderer::load_model) and enables the refinement flags, which the reference exposes as
// window check boxes (main.cpp:270-300) rather than flags:
//   --use_splitting          grow / resample / prune the tree from the per-ray trackers while rendering
//   --use_guided_sampling    composite per-sample network outputs instead of the tree's colours
//   --max_depth D  --max_sample_count C  --seed S
//   --save_tree FILE.npz     write the refined tree after the last frame
// Added for batch use:
//   --out PREFIX     write PREFIX_%04d.ppm (RGB from the RGBA8 output) per frame
//   --raw            also write PREFIX_%04d.f32 (float RGBA, row-major, little endian)
//   --frames N       render N frames; with --orbit DEG the camera is rotated about --origin around
//                    --world_up by DEG degrees per frame (default 0: N identical frames, for timing)
//   --gpu ID         HIP device
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "volume_renderer.hpp"

namespace {

struct Args {
    std::map<std::string, std::string> kv;
    std::string file;
    bool has(const std::string &k) const { return kv.count(k) != 0; }
    std::string get(const std::string &k, const std::string &d) const { return has(k) ? kv.at(k) : d; }
    float f(const std::string &k, float d) const { return has(k) ? std::strtof(kv.at(k).c_str(), nullptr) : d; }
    long l(const std::string &k, long d) const { return has(k) ? std::strtol(kv.at(k).c_str(), nullptr, 10) : d; }
    std::vector<float> vec(const std::string &k, std::vector<float> d) const {
        if (!has(k)) return d;
        std::vector<float> out;
        const std::string s = kv.at(k);
        size_t i = 0;
        while (i < s.size()) {
            size_t j = s.find(',', i);
            if (j == std::string::npos) j = s.size();
            out.push_back(std::strtof(s.substr(i, j - i).c_str(), nullptr));
            i = j + 1;
        }
        return out;
    }
};

Args parse(int argc, char **argv) {
    static const std::map<std::string, std::string> shorts = {
        {"s", "step_size"}, {"e", "stop_thresh"}, {"a", "sigma_thresh"}, {"c", "max_tree_capacity"}, {"x", "split_batch_size"},
        {"n", "nerf_batch_size"}, {"v", "samples_per_voxel"}, {"b", "bounds_only"}, {"y", "appearance_embedding"},
        {"z", "max_guided_samples"}, {"w", "width"}, {"h", "height"}};
    static const char *flags[] = {"bounds_only", "raw", "help", "use_splitting", "use_guided_sampling"};
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string t = argv[i];
        if (t.size() > 1 && t[0] == '-' && !(t[1] >= '0' && t[1] <= '9') && t[1] != '.') {
            std::string name = t.substr(t[1] == '-' ? 2 : 1), val;
            const size_t eq = name.find('=');
            bool has_val = false;
            if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has_val = true; }
            if (shorts.count(name)) name = shorts.at(name);
            bool is_flag = false;
            for (const char *f : flags) is_flag |= name == f;
            if (!has_val && !is_flag) {
                if (i + 1 >= argc) throw std::runtime_error("missing value for --" + name);
                val = argv[++i];
            }
            a.kv[name] = is_flag && !has_val ? "1" : val;
        } else {
            a.file = t;
        }
    }
    if (a.has("file")) a.file = a.kv["file"];
    return a;
}

void usage() {
    std::puts("usage: mnv_render npz_file [--bg 0.0] [-s step_size] [-e stop_thresh] [-a sigma_thresh] [-c max_tree_capacity]\n"
              "                  [-w width] [-h height] [--fx 1111] [--fy -1] [--cx -1] [--cy -1] [--center x,y,z] [--back x,y,z]\n"
              "                  [--origin x,y,z] [--world_up x,y,z] [-b] [--out PREFIX] [--raw] [--frames N] [--orbit DEG] [--gpu ID]\n"
              "                  [--model_path MODEL.npz [--use_splitting] [--use_guided_sampling] [-x split_batch_size] [-v samples_per_voxel]\n"
              "                   [-y appearance_embedding] [-z max_guided_samples] [--max_depth D] [--max_sample_count C] [--seed S]\n"
              "                   [--save_tree FILE.npz]]");
}

// rotate v about unit axis k by angle (Rodrigues), double precision
void rotate(float v[3], const float k[3], double ang) {
    const double c = std::cos(ang), s = std::sin(ang);
    const double x = v[0], y = v[1], z = v[2], kx = k[0], ky = k[1], kz = k[2];
    const double dot = kx * x + ky * y + kz * z;
    v[0] = (float)(x * c + (ky * z - kz * y) * s + kx * dot * (1 - c));
    v[1] = (float)(y * c + (kz * x - kx * z) * s + ky * dot * (1 - c));
    v[2] = (float)(z * c + (kx * y - ky * x) * s + kz * dot * (1 - c));
}

}  // namespace

int main(int argc, char **argv) {
    try {
        const Args args = parse(argc, argv);
        if (args.has("help") || args.file.empty()) {
            usage();
            return args.has("help") ? 0 : 2;
        }
        if (hipSetDevice((int)args.l("gpu", 0)) != hipSuccess) throw std::runtime_error("no usable HIP device");

        viewer::N3Tree tree(args.file);  // main.cpp:528
        if (args.has("bounds_only") && tree.N > 0) {  // main.cpp:529-538: keep one empty root chunk
            tree.capacity = 1;
            tree.data.assign((size_t)8 * tree.data_dim, 0);
            tree.child.assign(8, 0);
            tree.parent.assign(1, -1);
            tree.sample_counts.assign(8, 8);
        }
        const int width = (int)args.l("width", 800), height = (int)args.l("height", 800);  // main.cpp:491-492
        viewer::VolumeRenderer rend;
        // render_options_from_args, src/opts.cpp:49-67
        rend.options.background_brightness = args.f("bg", 0.0f);
        rend.options.step_size = args.f("step_size", 1e-4f);
        rend.options.stop_thresh = args.f("stop_thresh", 1e-2f);
        rend.options.sigma_thresh = args.f("sigma_thresh", 1e-2f);
        rend.options.split_batch_size = (int)args.l("split_batch_size", 4096);
        rend.options.nerf_batch_size = (int)args.l("nerf_batch_size", 4096);
        rend.options.samples_per_corner = (int)args.l("samples_per_voxel", 8);
        rend.options.appearance_embedding = (int)args.l("appearance_embedding", -1);
        rend.options.max_guided_samples = (int)args.l("max_guided_samples", 128);
        // camera from flags, main.cpp:550-582
        rend.camera = viewer::Camera(width, height, args.f("fx", 1111.f), args.f("fy", -1.f), args.f("cx", -1.f), args.f("cy", -1.f));
        const std::vector<float> center = args.vec("center", {-3.5f, 0.f, 3.5f}), back = args.vec("back", {-0.7071068f, 0.f, 0.7071068f}),
                                 origin = args.vec("origin", {0.f, 0.f, 0.f}), up = args.vec("world_up", {0.f, 0.f, 1.f});
        if (center.size() != 3 || back.size() != 3 || origin.size() != 3 || up.size() != 3) throw std::runtime_error("vector flags need 3 components");
        rend.camera.center = {center[0], center[1], center[2]};
        rend.camera.v_back = {back[0], back[1], back[2]};
        rend.camera.origin = {origin[0], origin[1], origin[2]};
        rend.camera.v_world_up = {up[0], up[1], up[2]};

        const bool refine = args.has("model_path") && (args.has("use_splitting") || args.has("use_guided_sampling"));
        // the reference reserves max_tree_capacity (default 20M chunks) up front; without refinement the tree cannot grow
        const long max_capacity = refine ? std::max<long>(tree.capacity, args.l("max_tree_capacity", 20000000)) : tree.capacity;
        if (tree.N > 0) rend.set(tree, max_capacity);
        rend.resize(width, height);
        if (args.has("model_path")) {  // main.cpp:585-589
            rend.load_model(args.get("model_path", ""));
            rend.options.use_splitting = args.has("use_splitting");
            rend.options.use_guided_sampling = args.has("use_guided_sampling");
            rend.options.max_depth = (int)args.l("max_depth", rend.options.max_depth);
            rend.options.max_sample_count = (int)args.l("max_sample_count", rend.options.max_sample_count);
            rend.seed = (uint64_t)args.l("seed", 0);
        }

        const long frames = args.l("frames", 1);
        const double orbit = args.f("orbit", 0.f) * M_PI / 180.0;
        const std::string out = args.get("out", "");
        float axis[3] = {up[0], up[1], up[2]};
        const float an = std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
        for (float &v : axis) v /= an;
        std::vector<float> rgba;
        std::vector<uint8_t> rgba8;
        mnv_set_timing(1);
        for (long f = 0; f < frames; ++f) {
            rend.render();
            if (refine) {
                const auto &st = rend.stats;
                std::printf("frame %ld: capacity %ld", f, st.capacity);
                if (st.split_candidates || st.added) std::printf("  split candidates %d, added %d%s", st.split_candidates, st.added, st.full ? " (full)" : "");
                if (st.sample_candidates) std::printf("  sample candidates %d, resampled %d", st.sample_candidates, st.resampled);
                if (st.pruned) std::printf("  pruned %d", st.pruned > 0 ? st.pruned : 0);
                if (st.guided_samples) std::printf("  guided samples %ld", st.guided_samples);
                std::printf("\n");
            }
            if (!out.empty()) {
                rend.download(args.has("raw") ? &rgba : nullptr, &rgba8);
                char name[4096];
                std::snprintf(name, sizeof(name), "%s_%04ld.ppm", out.c_str(), f);
                if (std::FILE *fp = std::fopen(name, "wb")) {
                    std::fprintf(fp, "P6\n%d %d\n255\n", width, height);
                    for (size_t p = 0; p < (size_t)width * height; ++p) std::fwrite(&rgba8[p * 4], 1, 3, fp);
                    std::fclose(fp);
                } else {
                    throw std::runtime_error(std::string("cannot write ") + name);
                }
                if (args.has("raw")) {
                    std::snprintf(name, sizeof(name), "%s_%04ld.f32", out.c_str(), f);
                    std::FILE *fp = std::fopen(name, "wb");
                    if (!fp) throw std::runtime_error(std::string("cannot write ") + name);
                    std::fwrite(rgba.data(), sizeof(float), rgba.size(), fp);
                    std::fclose(fp);
                }
            }
            if (orbit != 0.0) {  // rotate the camera about `origin` around world_up
                float c[3] = {rend.camera.center.x - origin[0], rend.camera.center.y - origin[1], rend.camera.center.z - origin[2]};
                float b[3] = {rend.camera.v_back.x, rend.camera.v_back.y, rend.camera.v_back.z};
                rotate(c, axis, orbit);
                rotate(b, axis, orbit);
                rend.camera.center = {c[0] + origin[0], c[1] + origin[1], c[2] + origin[2]};
                rend.camera.v_back = {b[0], b[1], b[2]};
            }
        }
        rend.download(nullptr, nullptr);
        if (args.has("save_tree") && tree.N > 0) {
            rend.sync_tree();
            tree.save_npz(args.get("save_tree", ""));
        }
        const double ms = rend.take_average_ms();
        std::printf("%s: %ld frame(s) %dx%d, %.3f ms/frame on the device, %.1f Mrays/s\n", rend.get_backend(), frames, width, height, ms,
                    ms > 0 ? (double)width * height / ms / 1e3 : 0.0);
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "mnv_render: %s\n", e.what());
        return 1;
    }
}
