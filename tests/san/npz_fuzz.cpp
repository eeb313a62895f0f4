// AddressSanitizer / UBSan driver for the .npz loader (host/npz.cpp, host/n3tree.cpp): opens mutated copies of a valid file --
// bit flips, truncations, random runs, random words -- and expects either a tree or a C++ exception, never a memory error.
// Built and run by tests/test_sanitizers.py (CPU only: GPU sanitizers are not available on this pool).
// usage: npz_fuzz <valid.npz> <iterations> <scratch.npz>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "n3tree.hpp"
#include "npz.hpp"
int main(int argc, char **argv) {
    std::string path = argv[1];
    if (argc < 4) return 2;
    int iters = atoi(argv[2]);
    std::ifstream f(path, std::ios::binary);
    std::vector<char> orig((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::mt19937_64 rng(1);
    int ok = 0, bad = 0;
    for (int it = 0; it < iters; ++it) {
        std::vector<char> b = orig;
        int kind = it % 4;
        if (kind == 0) { for (int k = 0; k < 1 + (int)(rng() % 8); ++k) b[rng() % b.size()] ^= (char)(1 << (rng() % 8)); }
        else if (kind == 1) { b.resize(rng() % b.size()); }
        else if (kind == 2) { size_t p = rng() % b.size(); for (int k = 0; k < 16 && p + k < b.size(); ++k) b[p + k] = (char)rng(); }
        else { size_t p = rng() % (b.size() - 4); uint32_t v = (uint32_t)rng(); memcpy(&b[p], &v, 4); }
        std::string tmp = argv[3];
        { std::ofstream o(tmp, std::ios::binary); o.write(b.data(), b.size()); }
        try { viewer::N3Tree t(tmp); (void)t.capacity; ++ok; } catch (const std::exception &e) { ++bad; }
    }
    printf("opened %d, rejected %d\n", ok, bad);
    return 0;
}
