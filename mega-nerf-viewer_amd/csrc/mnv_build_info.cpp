// mnv_build_info.cpp -- which sources this libmnv.so was built from (the Makefile passes the hash; not part of the hash itself).
#ifndef MNV_SOURCE_SHA
#define MNV_SOURCE_SHA "unknown"
#endif

// "mnv-source-sha:<hash>" is also findable with strings(1), i.e. without loading the library (__graft_entry__._prebuilt_sha)
static const char kStamp[] = "mnv-source-sha:" MNV_SOURCE_SHA;

extern "C" const char *mnv_source_sha(void) { return kStamp + 15; }
