// test_stream_raw -- IBank::streamRaw (dsk_amd/host/bank.cpp): the bank's TEXT as it lies in the (inflated) file, for an engine that
// parses on the device.  test_stream_raw <uri>  prints "RAW <format> <pieces> <bytes> <fnv1a of the bytes> <first piece flags ok>" or "NO"
// (the bank does not offer its text: album, BGZF, text that does not start like FASTA / FASTQ) -- nothing may have been handed on then.
#include "../../dsk_amd/host/bank.hpp"
#include "../../dsk_amd/host/tool.hpp"

#include <cstdint>
#include <cstdio>
#include <memory>

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: test_stream_raw <uri>\n"); return 2; }
    try {
        std::unique_ptr<dsk::IBank> bank(dsk::Bank::open(argv[1]));
        uint64_t h = 1469598103934665603ull, bytes = 0, pieces = 0; int fmt = 0; bool flags_ok = true;
        for (dsk::IBank* sub : bank->banks()) {
            uint64_t before = pieces;
            const bool ok = sub->streamRaw([&](const char* d, size_t n, int f, bool first) {
                if ((pieces == before) != first) flags_ok = false;          // new_file exactly on a file's first piece
                if (fmt && f != fmt && !first) flags_ok = false;
                fmt = f; ++pieces; bytes += n;
                for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)d[i]; h *= 1099511628211ull; }
            });
            if (!ok) { if (pieces != before) { printf("ERROR declined after handing on\n"); return 1; } printf("NO\n"); return 0; }
        }
        printf("RAW %d %llu %llu %016llx %d\n", fmt, (unsigned long long)pieces, (unsigned long long)bytes, (unsigned long long)h, flags_ok ? 1 : 0);
    } catch (dsk::Exception& e) { printf("EXCEPTION %s\n", e.getMessage()); return 1; }
    return 0;
}
