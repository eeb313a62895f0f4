// mnv_knobs.h -- measurement / ablation / test knobs of the library.  The SHIPPED libmnv.so reads no environment variable: every knob
// answers with its default (mnv_knobs.cpp compiled without MNV_TEST_HOOKS holds neither getenv nor a variable name).  The test-hook
// build (testhooks/libmnv.so, selected with MNV_LIB_PATH by tools/ and a few tests) compiles the same unit with the name table and reads
// MNV_<NAME> from the environment -- once per process, callers keep the value in a function-local static.
#pragma once

namespace mnv {

enum Knob {
    KNOB_TILE_WLOG,        // log2 of the ray-tile width (3: 8x8 tiles)
    KNOB_QUEUES,           // ray queues (8: one per XCD)
    KNOB_LDS_LEVEL,        // levels of the lookup grid staged in LDS
    KNOB_BLOCKS_PER_CU,    // persistent workgroups per compute unit
    KNOB_REFILL_MIN,       // idle lanes before a wavefront refills (single-frame launches)
    KNOB_ABLATE,           // diagnostics instantiation: bit mask of ablations (break results)
    KNOB_STATS,            // diagnostics instantiation: counters (1) / phase clocks (2)
    KNOB_TIMELINE,         // string: file for the tile / wavefront time stamps of the last launch
    KNOB_GRID2_LEVEL,      // level of the second lookup grid
    KNOB_BRICK_LEVELS,     // levels below the second lookup grid held in bricks (0: none)
    KNOB_F2_BLOCKS_PER_CU, // guided_fused2_kernel workgroups per compute unit
    KNOB_F2_SWITCH_MIN,    // guided_fused2_kernel: samples of the last sub-module that keep a consumer with it
    KNOB_FUSED_BATCH_MIN,  // guided_fused_kernel: pooled samples that start a network run
    KNOB_VOTE_WIDE_KEYS,   // the vote's 52-bit key path on ordinary inputs
    KNOB_VOTE_FULL_SORT,   // the vote's sort-all-counts path on small batches
    KNOB_ASSEMBLE_NARROW,  // one RGBA8 pixel per thread in mnv_assemble_tiles
    KNOB_REFRESH_DEBUG,    // mnv_accel_refresh prints what it patched
    KNOB_SYNTH_TIMING,     // the synthetic-tree generators print their phases
    KNOB_FOOTPRINT,        // string: file for the unique 128-byte lines the accel's launches touched, per array (written when the accel is destroyed)
    KNOB_SHADOW,           // allocate the copies the -DMNV_SHADOW_MASK variants of the march read (16 nodes, 32 rows, 64 bricks)
    KNOB_COUNT
};

// rays from which a mnv_render_voxels launch derives its level-7 lookup table first (mnv_march_ref_layout.hip): 65536.  The test-hook build
// exports mnv_hook_set_ref_table_min_rays(int64) so that the tests can run either path at any size (negative: never).
long long ref_table_min_rays();
// value of MNV_<NAME> as an integer (test-hook build) or `dflt` (shipped build, or the variable is not set)
int knob_int(Knob k, int dflt);
// whether MNV_<NAME> is set at all (test-hook build); false in the shipped build
bool knob_set(Knob k);
// the variable's text, or NULL
const char *knob_str(Knob k);

}  // namespace mnv
