/*
 * vbz_oracle_bench.c -- CPU ORACLE (test infrastructure): threaded timing harness for the CPU
 * baseline leg of bench.py.  It drives the oracle's vbo_compress / vbo_decompress (the restated
 * reference path + the pinned libzstd) from N pthreads over independent reads, the way the
 * reference is parallelised in practice (one process/thread per file: reference README.md:36-40).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "vbz_oracle.h"

typedef struct {
    int tid, threads;
    uint32_t n_reads;
    int16_t** reads;
    uint32_t* nsamples;
    const VboOptions* opts;
    int passes;
    uint64_t comp_bytes; /* of one pass */
    double enc_s, dec_s;
    int ok;
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
    uint32_t maxn = 0;
    for (uint32_t i = (uint32_t)w->tid; i < w->n_reads; i += (uint32_t)w->threads)
        if (w->nsamples[i] > maxn) maxn = w->nsamples[i];
    vbo_size_t cap = vbo_max_compressed_size(maxn * 2, w->opts);
    uint8_t* cbuf = (uint8_t*)malloc(cap + 64);
    int16_t* dbuf = (int16_t*)malloc((size_t)maxn * 2 + 64);
    w->ok = cbuf && dbuf;
    w->enc_s = w->dec_s = 0;
    for (int p = 0; p < w->passes && w->ok; ++p) {
        uint64_t comp = 0;
        for (uint32_t i = (uint32_t)w->tid; i < w->n_reads; i += (uint32_t)w->threads) {
            const uint32_t bytes = w->nsamples[i] * 2;
            double t0 = now_s();
            vbo_size_t c = vbo_compress(w->reads[i], bytes, cbuf, cap, w->opts);
            double t1 = now_s();
            if (vbo_is_error(c)) { w->ok = 0; break; }
            vbo_size_t d = vbo_decompress(cbuf, c, dbuf, bytes, w->opts);
            double t2 = now_s();
            if (d != bytes || memcmp(dbuf, w->reads[i], bytes) != 0) { w->ok = 0; break; }
            comp += c;
            w->enc_s += t1 - t0;
            w->dec_s += t2 - t1;
        }
        w->comp_bytes = comp;
    }
    free(cbuf);
    free(dbuf);
    return NULL;
}

typedef struct {
    uint32_t n_reads, first, threads_used;
    uint64_t seed;
    int16_t** reads;
    uint32_t* nsamples;
} gen_t;

static void* gen_worker(void* arg)
{
    worker_t* w = (worker_t*)arg;
    for (uint32_t i = (uint32_t)w->tid; i < w->n_reads; i += (uint32_t)w->threads)
        vbo_synth_signal(5, i, w->reads[i], w->nsamples[i]);
    return NULL;
}

/* Generates reads [0, n_reads) of the SURVEY 8(d) workload (seed 5), then runs encode+decode passes
 * on `threads` threads until at least `min_seconds` have elapsed.  out[0]=raw bytes per pass,
 * out[1]=compressed bytes per pass, out[2]=best pass wall seconds, out[3]=sum of per-thread encode
 * seconds per pass, out[4]=same for decode, out[5]=passes run.  Returns 0 on success. */
int vbo_bench_roundtrip(uint32_t n_reads, int threads, double min_seconds, const VboOptions* opts, double* out)
{
    if (threads < 1) threads = 1;
    int16_t** reads = (int16_t**)calloc(n_reads, sizeof(*reads));
    uint32_t* ns = (uint32_t*)calloc(n_reads, sizeof(*ns));
    worker_t* ws = (worker_t*)calloc((size_t)threads, sizeof(*ws));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(*th));
    if (!reads || !ns || !ws || !th) return -1;
    uint64_t raw = 0;
    for (uint32_t i = 0; i < n_reads; ++i) {
        ns[i] = vbo_synth_read_length(5, i);
        reads[i] = (int16_t*)malloc((size_t)ns[i] * 2);
        if (!reads[i]) return -1;
        raw += (uint64_t)ns[i] * 2;
    }
    for (int t = 0; t < threads; ++t) {
        ws[t].tid = t; ws[t].threads = threads; ws[t].n_reads = n_reads; ws[t].reads = reads; ws[t].nsamples = ns;
        ws[t].opts = opts; ws[t].passes = 1;
        pthread_create(&th[t], NULL, gen_worker, &ws[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    double best = 1e30, enc = 0, dec = 0, total = 0;
    uint64_t comp = 0;
    int passes = 0, ok = 1;
    while ((total < min_seconds || passes < 2) && ok) {
        double t0 = now_s();
        for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, worker, &ws[t]);
        for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
        double dt = now_s() - t0;
        total += dt;
        ++passes;
        if (dt < best) {
            best = dt;
            enc = dec = 0;
            comp = 0;
            for (int t = 0; t < threads; ++t) { enc += ws[t].enc_s; dec += ws[t].dec_s; comp += ws[t].comp_bytes; }
        }
        for (int t = 0; t < threads; ++t) ok &= ws[t].ok;
    }
    out[0] = (double)raw; out[1] = (double)comp; out[2] = best; out[3] = enc; out[4] = dec; out[5] = passes;
    for (uint32_t i = 0; i < n_reads; ++i) free(reads[i]);
    free(reads); free(ns); free(ws); free(th);
    return ok ? 0 : -2;
}
