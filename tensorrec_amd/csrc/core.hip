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

// ---- tuning knobs: name -> int, set from the host side (bench / tests) to A/B kernel variants in one process ----
#include <map>
#include <string>
static std::map<std::string, int>& tuning_map() { static std::map<std::string, int> m; return m; }

extern "C" int trec_set_tuning(const char* name, int32_t value)
{
    if (!name) return TREC_ERR_INVALID;
    tuning_map()[name] = value;
    return TREC_OK;
}

extern "C" int trec_get_tuning(const char* name, int dflt)
{
    auto it = tuning_map().find(name ? name : "");
    return it == tuning_map().end() ? dflt : it->second;
}
