// mnv_internal.h -- host-side declarations shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/mnv.h"
#include "mnv_device.h"
#include "mnv_error.h"

namespace mnv {

int check_hip(hipError_t e, const char *what);

// Fill the camera / option / rodrigues part of the kernel argument block.
int fill_params(FrameParams &P, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile);

void fill_origin(CamBlock &C, const float offset[3], const float scale[3]);
void fill_camera(CamBlock &C, const mnv_camera *cam);
int fill_tree_params(MarchParams &P, const mnv_tree_view *t);
int launch_ref_layout(const MarchParams &P, hipStream_t stream);
int launch_background(const FrameParams &P, hipStream_t stream);
// the packed layout follows a prune; called by mnv_prune_tree_accel with the OLD numbering (before fix-up and compaction)
int accel_apply_prune(mnv_accel *a, const int32_t *parent, const uint16_t *data, int32_t data_dim, const uint8_t *to_delete, const int32_t *shifts,
                      int32_t old_capacity, int32_t n_deleted, hipStream_t stream);

// a frame (plain, or with the refinement trackers) on `accel` with the offset / scale of the caller's tree view (the accel's arrays are what was cached; the transform is the call's)
int render_accel_for_tree(const mnv_accel *accel, const mnv_tree_view *tree, const mnv_camera *cam, const mnv_render_options *opt, mnv_rect tile,
                          const mnv_frame_inputs *inputs, float *rgba_out, uint8_t *rgba8_out, float *split_track, float *sample_track,
                          hipStream_t stream);

}  // namespace mnv
