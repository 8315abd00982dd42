// dev_knobs.cpp -- compiled ONLY into the developer build of the library (build.sh dev -> regard3d_amd/libr3dm_dev.so,
// -DR3DM_DEVTOOLS).  The product library never reads the environment (regard3d_amd/csrc/r3dm_internal.hpp).
#include <cstdlib>

int r3dm_dev_knob(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

const char* r3dm_dev_str(const char* name) { return getenv(name); }
