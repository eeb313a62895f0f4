#include "camera.hpp"

#include <cmath>
#include <cstring>

namespace viewer {

namespace {
// glm semantics used by the reference (3rdparty/glm/glm/detail/func_geometric.inl):
// dot = x*x + y*y + z*z summed left to right, normalize(v) = v * (1 / sqrt(dot(v, v))).
vec3 normalize(const vec3 &v) {
    const float d = v.x * v.x + v.y * v.y + v.z * v.z;
    const float inv = 1.0f / std::sqrt(d);
    return {v.x * inv, v.y * inv, v.z * inv};
}
vec3 cross(const vec3 &a, const vec3 &b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
}  // namespace

Camera::Camera(int width, int height, float fx, float fy, float cx, float cy)
    : width(width),
      height(height),
      fx(fx),
      fy(fy < 0.f ? fx : fy),
      cx(cx < 0.f ? (float)(width / 2) : cx),    // integer halving, as camera.cpp:35
      cy(cy < 0.f ? (float)(height / 2) : cy),
      default_fx(fx),
      default_fy(fy < 0.f ? fx : fy),
      default_cx(cx),
      default_cy(cy) {
    std::memset(transform, 0, sizeof(transform));
    center = {-3.55f, 0.0f, 3.55f};
    v_back = {-0.7071068f, 0.0f, 0.7071068f};
    v_world_up = {0.0f, 0.0f, 1.0f};
    origin = {0.0f, 0.0f, 0.0f};
    _update();
}

void Camera::_update(bool transform_from_vecs) {
    if (transform_from_vecs) {
        v_back = normalize(v_back);
        v_right = normalize(cross(v_world_up, v_back));
        v_up = cross(v_back, v_right);
        const vec3 *cols[4] = {&v_right, &v_up, &v_back, &center};
        for (int c = 0; c < 4; ++c) {
            for (int i = 0; i < 3; ++i) {
                if (transform[c * 3 + i] != (*cols[c])[i]) transform_changed_ = true;
                transform[c * 3 + i] = (*cols[c])[i];
            }
        }
    }
    if (last_fx != fx || last_fy != fy || last_width != width || last_height != height) {
        transform_changed_ = true;
        last_fx = fx;
        last_fy = fy;
        last_width = width;
        last_height = height;
    }
    if (transform_changed_) {
        has_changed_ = true;
        transform_changed_ = false;
    }
}

void Camera::move(const vec3 &xyz) {
    center.x += xyz.x * movement_speed;
    center.y += xyz.y * movement_speed;
    center.z += xyz.z * movement_speed;
}

bool Camera::has_changed() {
    const bool r = has_changed_;
    has_changed_ = false;
    return r;
}

mnv_camera Camera::c_abi() const {
    mnv_camera c;
    c.width = width;
    c.height = height;
    c.fx = fx;
    c.fy = fy;
    c.cx = cx;
    c.cy = cy;
    std::memcpy(c.c2w, transform, sizeof(c.c2w));
    return c;
}

}  // namespace viewer
