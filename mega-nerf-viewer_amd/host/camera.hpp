// camera.hpp -- viewer::Camera, the drop-in camera struct of the reference
// (reference include/camera.hpp:12-87, src/camera.cpp:29-130).
//
// Kept: the pose model (v_back / v_world_up / center / origin), the derived v_up /
// v_right, the 4x3 column-major camera-to-world `transform`, intrinsics, _update(),
// move() and the has_changed() latch.  Changed on purpose: there is no device copy
// of the transform -- the 12 floats travel by value inside mnv_camera, which removes
// the reference's hidden default-stream cudaMemcpyAsync (camera.cpp:113-123).
// The drag helpers (camera.hpp:22-25) are kept for API compatibility -- they feed only the interactive window, which is out of scope,
// but a host that scripts a camera path with them gets the reference's poses (tests/golden/ref_camera_drag.npz).
#pragma once

#include "../../include/mnv.h"

namespace viewer {

struct vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};

struct Camera {
    Camera(int width = 256, int height = 256, float fx = 1111.f, float fy = -1.f, float cx = -1.f,
           float cy = -1.f);

    /** Drag helpers (camera.hpp:22-25, camera.cpp:132-187): screen-space drag from (x, y) of begin_drag to (x, y) of drag_update --
        a pan along v_right / v_up, or a rotation about v_world_up and the start's v_right (about the camera, or about `origin`) **/
    void begin_drag(float x, float y, bool is_pan, bool about_origin);
    void drag_update(float x, float y);
    void end_drag();
    bool is_dragging() const;
    /** Move center by += xyz * movement_speed, correctly handling drag **/
    void move(const vec3 &xyz);
    bool has_changed();

    // Camera pose model, modify these then call _update()
    vec3 v_back, v_world_up, center;
    vec3 origin;
    // Updated by _update()
    vec3 v_up, v_right;
    // 4x3 C2W transform, column-major [right | up | back | center]
    float transform[12];

    int width, height;
    float fx, fy;
    float cx, cy;
    float default_fx, default_fy;
    float default_cx, default_cy;
    float movement_speed = 1.f;

    // Recompute v_right / v_up / transform from the pose vectors (camera.cpp:54-82)
    void _update(bool transform_from_vecs = true);

    // The by-value camera block handed to the C ABI (replaces CameraSpec, data_spec.hpp:9-23)
    mnv_camera c_abi() const;

private:
    struct DragState {
        bool is_dragging = false, is_panning = false, about_origin = false;
        float start_x = 0.f, start_y = 0.f;
        vec3 start_back, start_right, start_up, start_center, start_origin;
    } drag_;
    bool has_changed_ = true;
    bool transform_changed_ = false;
    float last_fx = 0.f, last_fy = 0.f;
    int last_width = 0, last_height = 0;
};

}  // namespace viewer
