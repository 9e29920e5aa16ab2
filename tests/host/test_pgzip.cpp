// test_pgzip -- dsk_amd/host/pgzip.cpp against zlib on one file:  test_pgzip <file.gz> <threads> <chunk bytes>
// prints "OK <bytes> <slabs> <ms parallel> <ms zlib>" when the parallel inflate reproduces zlib's bytes, "NA" when it declines the file
// (several members, too small, ..), "MISMATCH" / "ERROR <what>" otherwise (exit code 1).
#include "../../dsk_amd/host/pgzip.hpp"

#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: test_pgzip file.gz threads chunk_bytes\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    std::vector<unsigned char> z;
    { unsigned char buf[1 << 16]; size_t g; while ((g = fread(buf, 1, sizeof buf, f)) > 0) z.insert(z.end(), buf, buf + g); }
    fclose(f);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<char> ref;
    { gzFile g = gzopen(argv[1], "rb"); gzbuffer(g, 1 << 20); std::vector<char> buf(1 << 22); int got; while ((got = gzread(g, buf.data(), (unsigned)buf.size())) > 0) ref.insert(ref.end(), buf.begin(), buf.begin() + got); gzclose(g); }
    auto t1 = std::chrono::steady_clock::now();
    std::vector<char> out; size_t slabs = 0; bool saw_last = false;
    bool ok = false;
    try {
        ok = dsk::pgz_inflate(z.data(), z.size(), (unsigned)atoi(argv[2]), (size_t)atoll(argv[3]), 4096,
                              [&](char* d, size_t n, bool last) { d[-1] = 'x'; d[-4096] = 'y';      // (the headroom is writable)
                              out.insert(out.end(), d, d + n); ++slabs; if (saw_last) throw std::runtime_error("consume after last"); saw_last = last; });
    } catch (const std::exception& e) { printf("ERROR %s\n", e.what()); return 1; }
    auto t2 = std::chrono::steady_clock::now();
    if (!ok) { if (!out.empty()) { printf("ERROR declined after consuming\n"); return 1; } printf("NA\n"); return 0; }
    if (!saw_last || out.size() != ref.size() || memcmp(out.data(), ref.data(), ref.size()) != 0) { printf("MISMATCH %zu vs %zu\n", out.size(), ref.size()); return 1; }
    printf("OK %zu %zu %.1f %.1f\n", out.size(), slabs, std::chrono::duration<double, std::milli>(t2 - t1).count(), std::chrono::duration<double, std::milli>(t1 - t0).count());
    return 0;
}
