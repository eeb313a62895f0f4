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

namespace {
// 3x3 rotation about `axis` by `angle`, as glm::rotate builds it (3rdparty/glm/glm/ext/matrix_transform.inl: the axis is normalised, the
// matrix is c * I + (1 - c) * a a^T + s * [a]x), column-major: m[col][row]
struct mat3 {
    float m[3][3];
};
mat3 rotation(float angle, const vec3 &v) {
    const float c = std::cos(angle), s = std::sin(angle);
    const vec3 a = normalize(v);
    const vec3 t = {(1.0f - c) * a.x, (1.0f - c) * a.y, (1.0f - c) * a.z};
    mat3 r;
    r.m[0][0] = c + t.x * a.x;
    r.m[0][1] = t.x * a.y + s * a.z;
    r.m[0][2] = t.x * a.z - s * a.y;
    r.m[1][0] = t.y * a.x - s * a.z;
    r.m[1][1] = c + t.y * a.y;
    r.m[1][2] = t.y * a.z + s * a.x;
    r.m[2][0] = t.z * a.x + s * a.y;
    r.m[2][1] = t.z * a.y - s * a.x;
    r.m[2][2] = c + t.z * a.z;
    return r;
}
mat3 mul(const mat3 &a, const mat3 &b) {  // a * b: column j of the product = a's columns weighted by column j of b
    mat3 r;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) r.m[j][i] = a.m[0][i] * b.m[j][0] + a.m[1][i] * b.m[j][1] + a.m[2][i] * b.m[j][2];
    return r;
}
vec3 apply(const mat3 &a, const vec3 &v) {
    return {a.m[0][0] * v.x + a.m[1][0] * v.y + a.m[2][0] * v.z, a.m[0][1] * v.x + a.m[1][1] * v.y + a.m[2][1] * v.z,
            a.m[0][2] * v.x + a.m[1][2] * v.y + a.m[2][2] * v.z};
}
}  // namespace

void Camera::begin_drag(float x, float y, bool is_pan, bool about_origin) {
    drag_.is_dragging = true;
    drag_.start_x = x;
    drag_.start_y = y;
    drag_.start_back = v_back;
    drag_.start_right = v_right;
    drag_.start_up = v_up;
    drag_.start_center = center;
    drag_.start_origin = origin;
    drag_.is_panning = is_pan;
    drag_.about_origin = about_origin;
}

void Camera::drag_update(float x, float y) {
    if (!drag_.is_dragging) return;
    // screen delta -> angle / distance: -2 * movement_speed / max(width, height) per pixel (camera.cpp:146-148)
    const float k = -2.f * movement_speed / (float)(width > height ? width : height);
    float dx = (x - drag_.start_x) * k, dy = (y - drag_.start_y) * k;
    if (drag_.is_panning) {
        auto slide = [&](const vec3 &from) -> vec3 {
            return {from.x + dx * drag_.start_right.x - dy * drag_.start_up.x, from.y + dx * drag_.start_right.y - dy * drag_.start_up.y,
                    from.z + dx * drag_.start_right.z - dy * drag_.start_up.z};
        };
        center = slide(drag_.start_center);
        if (drag_.about_origin) origin = slide(drag_.start_origin);
        return;  // (the reference does not call _update here: the next frame's does)
    }
    if (drag_.about_origin) {
        dx = -dx;
        dy = -dy;
    }
    // tilt about the start's right vector; refuse to flip over the pole (camera.cpp:164-170)
    const mat3 tilt = rotation(-dy, drag_.start_right);
    const vec3 back_tilted = apply(tilt, drag_.start_back);
    const vec3 cr = cross(v_world_up, back_tilted);
    if (cr.x * drag_.start_right.x + cr.y * drag_.start_right.y + cr.z * drag_.start_right.z < 0.f) return;
    const float two_pi = 2.f * (float)M_PI;
    const mat3 m = mul(rotation(std::fmod(-dx, two_pi), v_world_up), tilt);
    v_back = normalize(apply(m, drag_.start_back));
    if (drag_.about_origin) {
        const vec3 rel = {drag_.start_center.x - origin.x, drag_.start_center.y - origin.y, drag_.start_center.z - origin.z};
        const vec3 r = apply(m, rel);
        center = {r.x + origin.x, r.y + origin.y, r.z + origin.z};
    }
    _update(true);
}

bool Camera::is_dragging() const { return drag_.is_dragging; }

void Camera::end_drag() { drag_.is_dragging = false; }

void Camera::move(const vec3 &xyz) {
    center.x += xyz.x * movement_speed;
    center.y += xyz.y * movement_speed;
    center.z += xyz.z * movement_speed;
    if (drag_.is_dragging) {
        drag_.start_center.x += xyz.x * movement_speed;
        drag_.start_center.y += xyz.y * movement_speed;
        drag_.start_center.z += xyz.z * movement_speed;
    }
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
