/*
 * orc_bench.c — timing harness for the oracle as the CPU baseline (bench.py's cpu_baseline leg; test infrastructure,
 * never linked into the product).  "The reference's CPU path timed beside it": the oracle restates aac.js's
 * process() + interleave (decoder.js:201-215, 218-334), which is single-threaded JavaScript; SURVEY.md §8d asks for
 * one thread and for all host cores.  Frames of different streams are independent, so T threads each decode their
 * own copy of the batch (own overlap state, own PCM buffer: T independent stream sets) until the time is up.
 */
#define _POSIX_C_SOURCE 200809L
#include "aac_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    int sample_index, input_kind, max_streams, max_channels;
    const aacg_unit_desc* units; uint32_t n_units;
    const void* coeffs; size_t coeff_bytes;
    const aacg_band_meta* meta; size_t n_meta;
    size_t n_pcm;
    double seconds;
    uint64_t batches;
    double elapsed;
    int rc;
} job_t;

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static void* worker(void* arg)
{
    job_t* j = (job_t*)arg;
    /* private copies: every thread is its own set of streams */
    void* coeffs = malloc(j->coeff_bytes);
    aacg_band_meta* meta = j->meta ? (aacg_band_meta*)malloc(j->n_meta * sizeof(aacg_band_meta)) : NULL;
    float* pcm = (float*)malloc(j->n_pcm * sizeof(float));
    float* ov = (float*)calloc((size_t)j->max_streams * (size_t)j->max_channels * 1024u, sizeof(float));
    if (!coeffs || !pcm || !ov || (j->meta && !meta)) { j->rc = -3; goto done; }
    memcpy(coeffs, j->coeffs, j->coeff_bytes);
    if (meta) memcpy(meta, j->meta, j->n_meta * sizeof(aacg_band_meta));
    const double t0 = now_s();
    do {
        j->rc = orc_decode_batch(j->sample_index, j->input_kind, j->max_streams, j->max_channels, j->units, j->n_units,
                                 coeffs, meta, pcm, ov, NULL);
        if (j->rc) break;
        j->batches++;
    } while (now_s() - t0 < j->seconds);
    j->elapsed = now_s() - t0;
done:
    free(coeffs); free(meta); free(pcm); free(ov);
    return NULL;
}

/* Runs n_threads workers for ~seconds each; returns the number of whole batches decoded by all of them (< 0: error)
 * and the longest worker time in *elapsed. */
long long orc_bench_threads(int n_threads, double seconds, int sample_index, int input_kind, int max_streams, int max_channels,
                            const aacg_unit_desc* units, uint32_t n_units, const void* coeffs, size_t coeff_bytes,
                            const aacg_band_meta* meta, size_t n_meta, size_t n_pcm_floats, double* elapsed)
{
    if (n_threads < 1 || n_threads > 4096) return -1;
    orc_init();
    job_t* jobs = (job_t*)calloc((size_t)n_threads, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    if (!jobs || !th) { free(jobs); free(th); return -3; }
    int started = 0;
    for (int i = 0; i < n_threads; i++) {
        job_t j = {sample_index, input_kind, max_streams, max_channels, units, n_units, coeffs, coeff_bytes, meta, n_meta, n_pcm_floats, seconds, 0, 0.0, 0};
        jobs[i] = j;
        if (pthread_create(&th[i], NULL, worker, &jobs[i]) != 0) break;
        started++;
    }
    long long total = 0;
    double longest = 0;
    int rc = started == n_threads ? 0 : -2;
    for (int i = 0; i < started; i++) {
        pthread_join(th[i], NULL);
        if (jobs[i].rc) rc = jobs[i].rc;
        total += (long long)jobs[i].batches;
        if (jobs[i].elapsed > longest) longest = jobs[i].elapsed;
    }
    if (elapsed) *elapsed = longest;
    free(jobs); free(th);
    return rc ? (long long)rc : total;
}
