/*
 * vbz_oracle_bench.c -- CPU ORACLE (test infrastructure): threaded timing harness for the CPU
 * baseline leg of bench.py.  It drives the oracle's vbo_compress / vbo_decompress (the restated
 * reference path -- scalar svb, not the SSSE3 worker -- plus the pinned libzstd) from N persistent
 * pthreads over independent reads, the way the reference is parallelised in practice (one
 * process/thread per file: reference README.md:36-40).
 *
 * Workers live for the whole call and meet at a barrier around every pass; reads are claimed from an
 * atomic counter (no static imbalance); the first pass is an untimed verification pass (decoded
 * samples memcmp'ed against the input), the timed passes contain vbo_compress + vbo_decompress only.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "vbz_oracle.h"

typedef struct {
    uint32_t n_reads;
    int threads;
    int16_t** reads;
    uint32_t* nsamples;        /* values per read (int16 samples, or uint32 elements when elem == 4) */
    uint32_t max_samples;
    uint32_t elem;             /* bytes per value: 2 = the int16 signal generator, 4 = the config-4 uint32 generator */
    const VboOptions* opts;
    pthread_barrier_t bar;
    atomic_uint next;          /* work queue: next unclaimed read */
    atomic_int stop, failed;
    int verify;                /* this pass compares the decoded samples (untimed pass) */
} shared_t;

typedef struct {
    shared_t* sh;
    int tid;
    uint64_t comp_bytes;       /* of the last pass */
    double enc_s, dec_s;       /* of the last pass */
} worker_t;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* worker(void* arg)
{
    worker_t* w = (worker_t*)arg;
    shared_t* sh = w->sh;
    vbo_size_t cap = vbo_max_compressed_size(sh->max_samples * sh->elem, sh->opts);
    uint8_t* cbuf = (uint8_t*)malloc((size_t)cap + 64);
    int16_t* dbuf = (int16_t*)malloc((size_t)sh->max_samples * sh->elem + 64);
    if (!cbuf || !dbuf) atomic_store(&sh->failed, 1);
    for (uint32_t i = (uint32_t)w->tid; i < sh->n_reads; i += (uint32_t)sh->threads) {  /* generation, untimed */
        if (sh->elem == 4) vbo_synth_u32(5, i, (uint32_t*)sh->reads[i], sh->nsamples[i]);
        else vbo_synth_signal(5, i, sh->reads[i], sh->nsamples[i]);
    }
    for (;;) {
        pthread_barrier_wait(&sh->bar);  /* pass start (the main thread resets the queue before it) */
        if (atomic_load(&sh->stop)) break;
        uint64_t comp = 0;
        double enc = 0, dec = 0;
        for (;;) {
            const uint32_t i = atomic_fetch_add(&sh->next, 1u);
            if (i >= sh->n_reads || atomic_load(&sh->failed)) break;
            const uint32_t bytes = sh->nsamples[i] * sh->elem;
            const double t0 = now_s();
            const vbo_size_t c = vbo_compress(sh->reads[i], bytes, cbuf, cap, sh->opts);
            const double t1 = now_s();
            if (vbo_is_error(c)) { atomic_store(&sh->failed, 1); break; }
            const vbo_size_t d = vbo_decompress(cbuf, c, dbuf, bytes, sh->opts);
            const double t2 = now_s();
            if (d != bytes || (sh->verify && memcmp(dbuf, sh->reads[i], bytes) != 0)) { atomic_store(&sh->failed, 1); break; }
            comp += c;
            enc += t1 - t0;
            dec += t2 - t1;
        }
        w->comp_bytes = comp;
        w->enc_s = enc;
        w->dec_s = dec;
        pthread_barrier_wait(&sh->bar);  /* pass end */
    }
    free(cbuf);
    free(dbuf);
    return NULL;
}

/* Generates reads [0, n_reads) of the SURVEY 8(d) workload (seed 5), runs one untimed verification pass, then
 * timed encode+decode passes on `threads` persistent threads until at least `min_seconds` have elapsed (at least
 * one).  out[0]=raw bytes per pass, out[1]=compressed bytes per pass, out[2]=best pass wall seconds, out[3]=sum of
 * per-thread encode seconds of that pass, out[4]=same for decode, out[5]=timed passes run.  Returns 0 on success. */
static int bench_roundtrip(uint32_t n_reads, uint32_t elem, uint32_t fixed_count, int threads, double min_seconds, const VboOptions* opts, double* out);

int vbo_bench_roundtrip(uint32_t n_reads, int threads, double min_seconds, const VboOptions* opts, double* out)
{
    return bench_roundtrip(n_reads, 2, 0, threads, min_seconds, opts, out);
}

/* The same harness over `n_buffers` buffers of `count` uint32 values of the config-4 generator (seed 5, buffer index i). */
int vbo_bench_roundtrip_u32(uint32_t n_buffers, uint32_t count, int threads, double min_seconds, const VboOptions* opts, double* out)
{
    return bench_roundtrip(n_buffers, 4, count, threads, min_seconds, opts, out);
}

static int bench_roundtrip(uint32_t n_reads, uint32_t elem, uint32_t fixed_count, int threads, double min_seconds, const VboOptions* opts, double* out)
{
    if (threads < 1) threads = 1;
    shared_t sh;
    memset(&sh, 0, sizeof sh);
    sh.n_reads = n_reads;
    sh.threads = threads;
    sh.opts = opts;
    sh.elem = elem;
    sh.reads = (int16_t**)calloc(n_reads ? n_reads : 1, sizeof(*sh.reads));
    sh.nsamples = (uint32_t*)calloc(n_reads ? n_reads : 1, sizeof(*sh.nsamples));
    worker_t* ws = (worker_t*)calloc((size_t)threads, sizeof(*ws));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(*th));
    if (!sh.reads || !sh.nsamples || !ws || !th) return -1;
    uint64_t raw = 0;
    for (uint32_t i = 0; i < n_reads; ++i) {
        sh.nsamples[i] = fixed_count ? fixed_count : vbo_synth_read_length(5, i);
        if (sh.nsamples[i] > sh.max_samples) sh.max_samples = sh.nsamples[i];
        sh.reads[i] = (int16_t*)malloc((size_t)sh.nsamples[i] * elem);
        if (!sh.reads[i]) return -1;
        raw += (uint64_t)sh.nsamples[i] * elem;
    }
    pthread_barrier_init(&sh.bar, NULL, (unsigned)threads + 1u);
    atomic_init(&sh.next, 0u);
    atomic_init(&sh.stop, 0);
    atomic_init(&sh.failed, 0);
    for (int t = 0; t < threads; ++t) {
        ws[t].sh = &sh;
        ws[t].tid = t;
        pthread_create(&th[t], NULL, worker, &ws[t]);
    }
    double best = 1e30, enc = 0, dec = 0, total = 0;
    uint64_t comp = 0;
    int passes = 0;
    /* pass 0: verification, untimed */
    sh.verify = 1;
    atomic_store(&sh.next, 0u);
    pthread_barrier_wait(&sh.bar);
    pthread_barrier_wait(&sh.bar);
    sh.verify = 0;
    while ((total < min_seconds || passes < 1) && !atomic_load(&sh.failed)) {
        atomic_store(&sh.next, 0u);
        const double t0 = now_s();
        pthread_barrier_wait(&sh.bar);
        pthread_barrier_wait(&sh.bar);
        const double dt = now_s() - t0;
        total += dt;
        ++passes;
        if (dt < best) {
            best = dt;
            enc = dec = 0;
            comp = 0;
            for (int t = 0; t < threads; ++t) { enc += ws[t].enc_s; dec += ws[t].dec_s; comp += ws[t].comp_bytes; }
        }
    }
    atomic_store(&sh.stop, 1);
    pthread_barrier_wait(&sh.bar);
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&sh.bar);
    const int ok = !atomic_load(&sh.failed);
    out[0] = (double)raw; out[1] = (double)comp; out[2] = best; out[3] = enc; out[4] = dec; out[5] = passes;
    for (uint32_t i = 0; i < n_reads; ++i) free(sh.reads[i]);
    free(sh.reads); free(sh.nsamples); free(ws); free(th);
    return ok ? 0 : -2;
}
