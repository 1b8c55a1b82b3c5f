/* AddressSanitizer run of the device front end's kernel source on the CPU (tests only): every output array is
 * allocated at its exact size, so a write or read outside what aacg_parse_batch promises aborts the program.
 *   asan_parse <case dir> <name> <sample index> <max units> <max channels> <options> <want tns> */
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/aacgpu.h"

extern "C" int emu_parse(int sample_index, const aacg_code_entry* entries, const uint32_t* counts,
                         const uint8_t* bytes, size_t n_bytes, const aacg_parse_frame* frames, uint32_t n_frames,
                         uint32_t max_units, uint32_t max_channels, uint32_t options,
                         aacg_unit_desc* units, int16_t* q, aacg_band_meta* meta, aacg_tns_info* tns, aacg_parse_result* results);
extern "C" const char* emu_last_error();

static std::vector<char> slurp(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { std::fprintf(stderr, "cannot read %s\n", path.c_str()); std::exit(2); }
    return std::vector<char>(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv)
{
    if (argc != 8) return 2;
    const std::string dir = argv[1], name = argv[2];
    const int si = std::atoi(argv[3]);
    const uint32_t U = (uint32_t)std::atoi(argv[4]), C = (uint32_t)std::atoi(argv[5]), options = (uint32_t)std::atoi(argv[6]);
    const bool want_tns = std::atoi(argv[7]) != 0;
    const std::vector<char> entries = slurp(dir + "/codebooks.entries"), counts = slurp(dir + "/codebooks.counts");
    const std::vector<char> bytes = slurp(dir + "/" + name + ".bytes"), frames = slurp(dir + "/" + name + ".frames");
    const uint32_t n = (uint32_t)(frames.size() / sizeof(aacg_parse_frame));
    std::vector<aacg_unit_desc> units((size_t)n * U);
    std::vector<int16_t> q((size_t)n * C * 1024);
    std::vector<aacg_band_meta> meta((size_t)n * C);
    std::vector<aacg_tns_info> tns(want_tns ? (size_t)n * C : 0);
    std::vector<aacg_parse_result> results(n);
    const int rc = emu_parse(si, (const aacg_code_entry*)entries.data(), (const uint32_t*)counts.data(), (const uint8_t*)bytes.data(), bytes.size(),
                             (const aacg_parse_frame*)frames.data(), n, U, C, options, units.data(), q.data(), meta.data(),
                             want_tns ? tns.data() : nullptr, results.data());
    if (rc) { std::fprintf(stderr, "emu_parse: %d %s\n", rc, emu_last_error()); return 1; }
    unsigned ok = 0;
    for (auto& r : results) ok += r.status == 0;
    std::printf("asan_parse %s: %u frames, %u parsed\n", name.c_str(), n, ok);
    return 0;
}
