// Identity of the build: a hash of every source the library was compiled from (rpsmf_amd/build.py passes it in).
// __graft_entry__.build() and rpsmf_amd.build compare it with the hash of the sources on disk.
#include "../../include/psmf_hip.h"
#ifndef PSMF_BUILD_ID
#define PSMF_BUILD_ID "unknown"
#endif
// (the marker lets build.py read the id from the file without dlopen()ing a library that may be stale)
static const char psmf_build_marker[] = "PSMF_BUILD_ID=" PSMF_BUILD_ID;
extern "C" const char* psmf_build_id(void) { return psmf_build_marker + 14; }
