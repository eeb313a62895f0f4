// link-only stand-ins for the device entry points n3tree.cpp references (never called by the fuzz driver)
#include "mnv.h"
extern "C" {
const char *mnv_last_error(void) { return ""; }
int mnv_accel_create_reserved(const mnv_tree_view *, int64_t, void *, mnv_accel **) { return -3; }
int mnv_accel_rebuild(mnv_accel *, const mnv_tree_view *, void *) { return -3; }
int mnv_accel_refresh(mnv_accel *, const mnv_tree_view *, int32_t, const int32_t *, int32_t, void *) { return -3; }
void mnv_accel_destroy(mnv_accel *) {}
}
