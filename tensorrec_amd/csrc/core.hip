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

// ---- CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing-by-8 on the host ----
// TFRecord framing (tensorrec/input_utils.py:103-105 writes through tf.python_io.TFRecordWriter; :141 reads through
// tf.data.TFRecordDataset) checksums the length header and the payload with a masked CRC-32C.  Interaction files run to
// hundreds of MB, so the checksum is native; tensorrec_amd/input_utils.py does the framing and the protobuf wire format.
static uint32_t g_crc_tab[8][256];
static bool g_crc_ready = false;
static void crc32c_init()
{
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
        g_crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xffu];
    g_crc_ready = true;
}

// crc: 0 for a fresh checksum, or the value returned for the preceding bytes (streaming); returns the CRC-32C as int32 bits
extern "C" int trec_crc32c(const void* data, uint64_t n, uint32_t crc)
{
    if (!g_crc_ready) crc32c_init();
    const unsigned char* p = (const unsigned char*)data;
    uint32_t c = ~crc;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = g_crc_tab[7][lo & 0xffu] ^ g_crc_tab[6][(lo >> 8) & 0xffu] ^ g_crc_tab[5][(lo >> 16) & 0xffu] ^
            g_crc_tab[4][lo >> 24] ^ g_crc_tab[3][hi & 0xffu] ^ g_crc_tab[2][(hi >> 8) & 0xffu] ^
            g_crc_tab[1][(hi >> 16) & 0xffu] ^ g_crc_tab[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ g_crc_tab[0][(c ^ *p++) & 0xffu];
    return (int)(~c);
}
