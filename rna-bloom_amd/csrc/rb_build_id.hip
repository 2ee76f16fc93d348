// rb_build_id(): which kernel sources this library was built from (tools/csrc_id.py: sha256 over csrc/* and include/rb_capi.h).
// The Makefile passes the value; bench.py compares it with the id stored beside the committed PMC summaries.
#include "../../include/rb_capi.h"
#ifndef RB_BUILD_ID
#define RB_BUILD_ID "unknown"
#endif
extern "C" const char *rb_build_id(void) { return RB_BUILD_ID; }
