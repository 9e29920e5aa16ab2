/*
 * dsk_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See dsk_oracle.h for the reference file:line each rule follows and for the
 * parity-pinning statement.  Build: see oracle/Makefile (gcc + zlib + pthread).
 */
#define _GNU_SOURCE
#include "dsk_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* 4096 value-prefix partitions: canonical k-mers are densest at small values (up to twice the mean), so partitions far
 * outnumber the threads and are handed out by a ticket -- with 256 partitions on 256 threads the slowest one set the pace */
#define PART_BITS 12
#define NPART (1 << PART_BITS)

struct dsko_result {
    int k;
    uint64_t total, distinct;
    uint64_t *lo, *hi;      /* words 0 and 1 of every row */
    uint64_t *w2, *w3;      /* words 2 and 3 (k > 64 only, else NULL) */
    uint32_t* ab;
};

/* README.md:111 -- A=0, C=1, T=2, G=3; anything else is not a nucleotide. */
static uint8_t g_code[256];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_code(void) {
    memset(g_code, 255, sizeof(g_code));
    g_code['A'] = g_code['a'] = 0;
    g_code['C'] = g_code['c'] = 1;
    g_code['T'] = g_code['t'] = 2;
    g_code['G'] = g_code['g'] = 3;
}

#include <sys/mman.h>
/* large arrays: 2 MB-aligned and marked for transparent huge pages (what free() releases again) -- 4 KB first-touch faults of a
 * multi-GB array do not scale over threads (the kernel serialises them): 0.4 s for the 3.8 GB key array of 4.8e8 k-mers on 8 threads
 * and on 256 alike */
static void* dsko_big_alloc(size_t bytes) {
    void* p = NULL;
    if (bytes < ((size_t)4 << 20)) return malloc(bytes ? bytes : 1);
    if (posix_memalign(&p, (size_t)2 << 20, bytes)) return NULL;
    (void)madvise(p, bytes, MADV_HUGEPAGE);
    return p;
}

/* The key array of a count (8 / 16 / 32 bytes per k-mer: the largest allocation by far) is kept for the next call instead of being
 * given back: releasing 3.8 GB takes the kernel 0.19 s (pages are cleared on free) and faulting them in again costs as much -- a
 * third of a count on 32 threads.  One buffer, grow-only, handed out under a mutex; it lives until the process ends. */
static pthread_mutex_t g_arena_mu = PTHREAD_MUTEX_INITIALIZER;
static void* g_arena = NULL;
static size_t g_arena_bytes = 0;
static void* dsko_arena_take(size_t bytes, size_t* cap) {
    void* p = NULL;
    pthread_mutex_lock(&g_arena_mu);
    if (g_arena && g_arena_bytes >= bytes) { p = g_arena; *cap = g_arena_bytes; g_arena = NULL; g_arena_bytes = 0; }
    pthread_mutex_unlock(&g_arena_mu);
    if (p) return p;
    *cap = bytes;
    return dsko_big_alloc(bytes);
}
static void dsko_arena_give(void* p, size_t cap) {
    void* old = NULL;
    if (!p) return;
    pthread_mutex_lock(&g_arena_mu);
    if (!g_arena || cap > g_arena_bytes) { old = g_arena; g_arena = p; g_arena_bytes = cap; } else old = p;
    pthread_mutex_unlock(&g_arena_mu);
    free(old);
}

#include <time.h>
static double dsko_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

#define KT uint64_t
#define SFX 64
#include "dsk_oracle_core.inc"
#undef KT
#undef SFX
#define KT unsigned __int128
#define SFX 128
#include "dsk_oracle_core.inc"
/* k in 65..128: 256-bit keys.  C23 _BitInt needs clang (ROCm's is in the image); a gcc-11 build of the
 * oracle simply lacks this width and dsko_count returns NULL for k > 64. */
#if defined(__clang__) && defined(__BITINT_MAXWIDTH__) && __BITINT_MAXWIDTH__ >= 256
#define DSKO_HAVE_256 1
#undef KT
#undef SFX
#define KT unsigned _BitInt(256)
#define SFX 256
#include "dsk_oracle_core.inc"
#endif
#undef KT
#undef SFX

dsko_result* dsko_count(const uint8_t* stream, uint64_t nbytes, int k, int nthreads) {
    pthread_once(&g_once, init_code);
    if (k < 1 || k > 128) return NULL;
#ifdef DSKO_HAVE_256
    if (k > 64) return count256(stream, nbytes, k, nthreads);
#else
    if (k > 64) return NULL;
#endif
    return k <= 32 ? count64(stream, nbytes, k, nthreads) : count128(stream, nbytes, k, nthreads);
}

int dsko_max_kmer_size(void) {
#ifdef DSKO_HAVE_256
    return 128;
#else
    return 64;
#endif
}

void dsko_free(dsko_result* r) {
    if (!r) return;
    free(r->lo); free(r->hi); free(r->w2); free(r->w3); free(r->ab); free(r);
}

void dsko_rows4(const dsko_result* r, uint64_t* w0, uint64_t* w1, uint64_t* w2, uint64_t* w3, uint32_t* abundance) {
    dsko_rows(r, w0, w1, abundance);
    if (w2) { if (r->w2) memcpy(w2, r->w2, r->distinct * sizeof(uint64_t)); else memset(w2, 0, r->distinct * sizeof(uint64_t)); }
    if (w3) { if (r->w3) memcpy(w3, r->w3, r->distinct * sizeof(uint64_t)); else memset(w3, 0, r->distinct * sizeof(uint64_t)); }
}

uint64_t dsko_total_kmers(const dsko_result* r) { return r->total; }
uint64_t dsko_num_distinct(const dsko_result* r) { return r->distinct; }

void dsko_rows(const dsko_result* r, uint64_t* lo, uint64_t* hi, uint32_t* abundance) {
    if (lo) memcpy(lo, r->lo, r->distinct * sizeof(uint64_t));
    if (hi) memcpy(hi, r->hi, r->distinct * sizeof(uint64_t));
    if (abundance) memcpy(abundance, r->ab, r->distinct * sizeof(uint32_t));
}

/* test/k27.histo: one row per abundance 1..10000, value = number of distinct
 * k-mers seen that many times (all k-mers, before the solidity filter); the
 * last row absorbs larger abundances. */
void dsko_histogram(const dsko_result* r, uint64_t* out, uint32_t histo_max) {
    memset(out, 0, ((size_t)histo_max + 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < r->distinct; i++) {
        uint32_t c = r->ab[i];
        out[c > histo_max ? histo_max : c]++;
    }
}

uint64_t dsko_num_solid(const dsko_result* r, uint32_t amin, uint32_t amax) {
    uint64_t n = 0;
    for (uint64_t i = 0; i < r->distinct; i++) n += (r->ab[i] >= amin && r->ab[i] <= amax);
    return n;
}

/* utils/dsk2ascii.cpp:104 prints model.toString(value): first base = most
 * significant 2 bits (test/short.parse_results:1 round-trips under this). */
void dsko_kmer_to_string(uint64_t lo, uint64_t hi, int k, char* out) {
    static const char L[4] = {'A', 'C', 'T', 'G'};
    unsigned __int128 v = ((unsigned __int128)hi << 64) | lo;
    for (int i = 0; i < k; i++) out[i] = L[(unsigned)(v >> (2 * (k - 1 - i))) & 3];
    out[k] = 0;
}

void dsko_enumerate(const uint8_t* s, uint64_t n, int k, uint64_t* lo, uint64_t* hi, uint8_t* valid) {
    pthread_once(&g_once, init_code);
    typedef unsigned __int128 K;
    const K mask = (k == 64) ? ~(K)0 : (((K)1 << (2 * k)) - 1);
    K fwd = 0, rc = 0; int run = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint8_t c = g_code[s[i]];
        valid[i] = 0;
        if (lo) lo[i] = 0;
        if (hi) hi[i] = 0;
        if (c > 3) { run = 0; fwd = rc = 0; continue; }
        fwd = ((fwd << 2) | c) & mask;
        rc = (rc >> 2) | ((K)(c ^ 2) << (2 * (k - 1)));
        if (++run >= k) {
            K canon = fwd < rc ? fwd : rc;
            valid[i] = 1;
            if (lo) lo[i] = (uint64_t)canon;
            if (hi) hi[i] = (uint64_t)(canon >> 64);
        }
    }
}

/* Same for k up to 128: words[4*i .. 4*i+3] = the canonical k-mer ending at byte i, least significant word first. */
int dsko_enumerate4(const uint8_t* s, uint64_t n, int k, uint64_t* words, uint8_t* valid) {
#ifdef DSKO_HAVE_256
    pthread_once(&g_once, init_code);
    typedef unsigned _BitInt(256) K;
    const K mask = (k == 128) ? ~(K)0 : (((K)1 << (2 * k)) - 1);
    K fwd = 0, rc = 0; int run = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint8_t c = g_code[s[i]];
        valid[i] = 0;
        for (int x = 0; x < 4; x++) words[4 * i + x] = 0;
        if (c > 3) { run = 0; fwd = rc = 0; continue; }
        fwd = ((fwd << 2) | c) & mask;
        rc = (rc >> 2) | ((K)(c ^ 2) << (2 * (k - 1)));
        if (++run >= k) {
            K canon = fwd < rc ? fwd : rc;
            valid[i] = 1;
            for (int x = 0; x < 4; x++) words[4 * i + x] = (uint64_t)(canon >> (64 * x));
        }
    }
    return 0;
#else
    (void)s; (void)n; (void)k; (void)words; (void)valid;
    return -1;
#endif
}

void dsko_minimizers(const uint8_t* s, uint64_t n, int k, int m, uint32_t* minim, uint8_t* valid) {
    pthread_once(&g_once, init_code);
    const uint32_t mmask = (m == 16) ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1);
    int run = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint8_t c = g_code[s[i]];
        valid[i] = 0; minim[i] = 0;
        if (c > 3) { run = 0; continue; }
        if (++run < k) continue;
        /* brute force over the k-m+1 m-mers of the window ending at i */
        uint32_t best = 0xFFFFFFFFu;
        for (uint64_t e = i - (uint64_t)(k - m); e <= i; e++) {
            uint32_t f = 0, r = 0;
            for (int t = 0; t < m; t++) {
                uint8_t b = g_code[s[e - (uint64_t)(m - 1) + (uint64_t)t]];
                f = ((f << 2) | b) & mmask;
                r = (r >> 2) | ((uint32_t)(b ^ 2) << (2 * (m - 1)));
            }
            uint32_t cm = f < r ? f : r;
            if (cm < best) best = cm;
        }
        valid[i] = 1; minim[i] = best;
    }
}

/* ------------------------------------------------------------------ bank */

typedef struct { uint8_t* p; uint64_t n, cap; } buf_t;
static void buf_reserve(buf_t* b, uint64_t extra) {
    if (b->n + extra <= b->cap) return;
    uint64_t c = b->cap ? b->cap : (1u << 20);
    while (c < b->n + extra) c *= 2;
    b->p = (uint8_t*)realloc(b->p, c); b->cap = c;
}

static int slurp_gz(const char* path, buf_t* raw) {
    gzFile f = gzopen(path, "rb");   /* transparently reads plain files too */
    if (!f) return -1;
    gzbuffer(f, 1 << 20);
    for (;;) {
        buf_reserve(raw, 1 << 22);
        int got = gzread(f, raw->p + raw->n, 1 << 22);
        if (got < 0) { gzclose(f); return -2; }
        if (got == 0) break;
        raw->n += (uint64_t)got;
    }
    gzclose(f);
    return 0;
}

/* README.md:52-61: FASTA ('>' header, sequence possibly over several lines,
 * test/longread.fasta) or FASTQ ('@' header, sequence, '+', quality). */
static void parse_records(const buf_t* raw, buf_t* out, uint64_t* nreads) {
    const uint8_t* s = raw->p; uint64_t n = raw->n, i = 0;
    while (i < n) {
        while (i < n && (s[i] == '\n' || s[i] == '\r' || s[i] == ' ' || s[i] == '\t')) i++;
        if (i >= n) break;
        uint8_t kind = s[i];
        while (i < n && s[i] != '\n') i++;          /* header line */
        if (i < n) i++;
        uint64_t seq_begin = out->n;
        if (kind == '>') {
            while (i < n && s[i] != '>') {
                uint64_t e = i; while (e < n && s[e] != '\n') e++;
                buf_reserve(out, e - i + 1);
                for (uint64_t t = i; t < e; t++) if (s[t] != '\r' && s[t] != ' ' && s[t] != '\t') out->p[out->n++] = s[t];
                i = e < n ? e + 1 : e;
            }
        } else if (kind == '@') {
            while (i < n && s[i] != '+') {
                uint64_t e = i; while (e < n && s[e] != '\n') e++;
                buf_reserve(out, e - i + 1);
                for (uint64_t t = i; t < e; t++) if (s[t] != '\r' && s[t] != ' ' && s[t] != '\t') out->p[out->n++] = s[t];
                i = e < n ? e + 1 : e;
            }
            uint64_t len = out->n - seq_begin;
            while (i < n && s[i] != '\n') i++;      /* '+' line */
            if (i < n) i++;
            uint64_t q = 0;                          /* quality: len symbols */
            while (i < n && q < len) { if (s[i] != '\n' && s[i] != '\r') q++; i++; }
            while (i < n && s[i] != '\n') i++;
        } else {
            continue;                                /* junk line: skipped */
        }
        buf_reserve(out, 1);
        out->p[out->n++] = '\n';
        (*nreads)++;
    }
}

int dsko_load_bank(const char* uri, uint8_t** stream, uint64_t* nbytes, uint64_t* nreads) {
    buf_t out = {0, 0, 0};
    uint64_t nr = 0;
    char* list = strdup(uri);
    int rc = 0;
    for (char* tok = strtok(list, ","); tok; tok = strtok(NULL, ",")) {
        buf_t raw = {0, 0, 0};
        if (slurp_gz(tok, &raw) != 0) { rc = -1; free(raw.p); break; }
        parse_records(&raw, &out, &nr);
        free(raw.p);
    }
    free(list);
    if (rc) { free(out.p); return rc; }
    buf_reserve(&out, 1);
    *stream = out.p; *nbytes = out.n;
    if (nreads) *nreads = nr;
    return 0;
}

void dsko_free_stream(uint8_t* stream) { free(stream); }
