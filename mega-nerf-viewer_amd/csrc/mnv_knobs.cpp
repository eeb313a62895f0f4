// mnv_knobs.cpp -- see mnv_knobs.h.  Compiled twice: plain for libmnv.so (no getenv, no names), with MNV_TEST_HOOKS for testhooks/libmnv.so.
#include "mnv_knobs.h"

#ifdef MNV_TEST_HOOKS
#include <atomic>
#include <cstdlib>
#endif

namespace mnv {

#ifdef MNV_TEST_HOOKS
static const char *const kKnobNames[KNOB_COUNT] = {
    "MNV_TILE_WLOG",      "MNV_QUEUES",           "MNV_LDS_LEVEL",      "MNV_BLOCKS_PER_CU",   "MNV_REFILL_MIN",     "MNV_ABLATE",
    "MNV_STATS",          "MNV_TIMELINE",         "MNV_GRID2_LEVEL",    "MNV_BRICK_LEVELS",    "MNV_F2_BLOCKS_PER_CU", "MNV_F2_SWITCH_MIN",
    "MNV_FUSED_BATCH_MIN", "MNV_VOTE_WIDE_KEYS",  "MNV_VOTE_FULL_SORT", "MNV_ASSEMBLE_NARROW", "MNV_REFRESH_DEBUG",  "MNV_SYNTH_TIMING",
    "MNV_FOOTPRINT",      "MNV_SHADOW",
};
const char *knob_str(Knob k) { return k >= 0 && k < KNOB_COUNT ? std::getenv(kKnobNames[k]) : nullptr; }
int knob_int(Knob k, int dflt) {
    const char *v = knob_str(k);
    return v ? std::atoi(v) : dflt;
}
bool knob_set(Knob k) { return knob_str(k) != nullptr; }
static std::atomic<long long> g_ref_table_min_rays{1 << 16};
long long ref_table_min_rays() { return g_ref_table_min_rays.load(std::memory_order_relaxed); }
#else
long long ref_table_min_rays() { return 1 << 16; }
const char *knob_str(Knob) { return nullptr; }
int knob_int(Knob, int dflt) { return dflt; }
bool knob_set(Knob) { return false; }
#endif

}  // namespace mnv

#ifdef MNV_TEST_HOOKS
extern "C" void mnv_hook_set_ref_table_min_rays(long long min_rays) { mnv::g_ref_table_min_rays.store(min_rays, std::memory_order_relaxed); }
#endif
