/*
 * dsk_oracle_cli.c -- CPU ORACLE command line (TEST INFRASTRUCTURE).
 * Mirrors the two reference commands used by scripts/simple_test.sh:36-37,88-89:
 *   dsk_oracle_cli -file <uri> -kmer-size K [-abundance-min A] [-abundance-max B]
 *                  [-histo-max H] [-nb-cores T] [-histo out.histo] [-ascii out.txt]
 * -histo writes "<i>\t<count>\n" rows 1..H (the text simple_test.sh:37 extracts);
 * -ascii writes "<kmer> <count>\n" rows (utils/dsk2ascii.cpp:104), ascending.
 */
#include "dsk_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char** argv) {
    const char *file = NULL, *histo = NULL, *ascii = NULL;
    int k = 31, threads = 1; unsigned amin = 2, amax = 2147483647u, hmax = 10000;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "-file")) file = argv[i + 1];
        else if (!strcmp(argv[i], "-kmer-size")) k = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-abundance-min")) amin = (unsigned)atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-abundance-max")) amax = (unsigned)strtoul(argv[i + 1], 0, 10);
        else if (!strcmp(argv[i], "-histo-max")) hmax = (unsigned)atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-nb-cores")) threads = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "-histo")) histo = argv[i + 1];
        else if (!strcmp(argv[i], "-ascii")) ascii = argv[i + 1];
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (!file) { fprintf(stderr, "usage: %s -file <uri> -kmer-size K ...\n", argv[0]); return 2; }
    uint8_t* s; uint64_t n, nreads;
    double t0 = now();
    if (dsko_load_bank(file, &s, &n, &nreads)) { fprintf(stderr, "EXCEPTION: cannot read %s\n", file); return 1; }
    double t1 = now();
    dsko_result* r = dsko_count(s, n, k, threads);
    double t2 = now();
    if (!r) { fprintf(stderr, "EXCEPTION: bad kmer size\n"); return 1; }
    uint64_t d = dsko_num_distinct(r);
    fprintf(stderr, "reads %llu bytes %llu kmers %llu distinct %llu solid %llu parse_s %.3f count_s %.3f\n",
            (unsigned long long)nreads, (unsigned long long)n, (unsigned long long)dsko_total_kmers(r),
            (unsigned long long)d, (unsigned long long)dsko_num_solid(r, amin, amax), t1 - t0, t2 - t1);
    if (histo) {
        uint64_t* h = (uint64_t*)calloc(hmax + 1, sizeof(uint64_t));
        dsko_histogram(r, h, hmax);
        FILE* f = fopen(histo, "wb");
        for (unsigned i = 1; i <= hmax; i++) fprintf(f, "%u\t%llu\n", i, (unsigned long long)h[i]);
        fclose(f); free(h);
    }
    if (ascii) {
        uint64_t* lo = malloc((d ? d : 1) * 8); uint64_t* hi = malloc((d ? d : 1) * 8); uint32_t* ab = malloc((d ? d : 1) * 4);
        dsko_rows(r, lo, hi, ab);
        FILE* f = fopen(ascii, "wb");
        char buf[80];
        for (uint64_t i = 0; i < d; i++) if (ab[i] >= amin && ab[i] <= amax) {
            dsko_kmer_to_string(lo[i], hi[i], k, buf);
            fprintf(f, "%s %u\n", buf, ab[i]);
        }
        fclose(f); free(lo); free(hi); free(ab);
    }
    dsko_free(r); dsko_free_stream(s);
    return 0;
}
