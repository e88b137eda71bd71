// tensorrec_amd/csrc/core.hip -- library-level entry points (version, last error, device query).
#include "common.hpp"
#include <string.h>

static thread_local char g_last_error[512] = "";

extern "C" void trec_set_last_error(const char* msg)
{
    strncpy(g_last_error, msg ? msg : "", sizeof(g_last_error) - 1);
    g_last_error[sizeof(g_last_error) - 1] = 0;
}

extern "C" const char* trec_last_error(void) { return g_last_error; }

extern "C" int trec_abi_version(void) { return 1; }

// number of compute units of the current device (256 on MI355X); used by the host side to size grids
extern "C" int trec_device_cu_count(void)
{
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return n;
}
