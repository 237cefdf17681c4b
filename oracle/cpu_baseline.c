/*
 * cpu_baseline.c -- timing driver for the CPU ORACLE (test / measurement infrastructure, NOT product code).
 *
 * bench.py's `cpu_baseline` leg: the "reference CPU path" of SURVEY.md section 8(d) -- the single-threaded fp64 restatement
 * of loadPointCloud + computeNDTCells (x2) + NDTMatcherD2D::match (oracle/ndt_oracle.c), compiled -O3 -march=native on
 * the box that runs the bench, pinned with taskset by the caller, no Python in the timed loop: one warm-up pass,
 * then `reps` timed passes over the sample (CLOCK_MONOTONIC), median reported.  With -fopenmp (cpu_baseline_omp) the
 * pairs of a pass are spread over `threads` threads: the labelled all-cores figure (upstream's derivativesNDT runs an
 * OpenMP team; here whole registrations run in parallel, which is the more favourable way to use the cores).
 *
 * usage: cpu_baseline <sample.bin> <reps> <threads>
 * sample.bin: int32 n_pairs, int32 n_points, double res, double size[3], double range, double delta_score,
 *             int32 n_neighbours, int32 itr_max; then per pair: float xyz[n_points*3] fixed, same moving, double T[16]
 *             (column-major initial guess).  Writes <sample.bin>.out: n_pairs x 16 doubles (registered poses).
 */
#define _GNU_SOURCE
#include "ndt_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static int cmp_d(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

typedef struct {
    int n_pairs, n_points, n_neighbours, itr_max;
    double res, size[3], range, delta;
    const float *scans;   /* per pair: fixed then moving */
    const double *T0;
} sample;

static void register_pair(const sample *s, int k, double T[16], int *iters)
{
    const double centre[3] = {0, 0, 0};
    const size_t np = (size_t)s->n_points;
    const float *fixed = s->scans + (size_t)k * 2 * np * 3, *moving = fixed + np * 3;
    oracle_map *a = oracle_map_create(s->res, centre, s->size), *b = oracle_map_create(s->res, centre, s->size);
    oracle_map_load_points(a, fixed, np, 3, s->range, NULL);
    oracle_map_compute_cells(a, 3, 1000.0);
    oracle_map_load_points(b, moving, np, 3, s->range, NULL);
    oracle_map_compute_cells(b, 3, 1000.0);
    oracle_match_params p = {s->n_neighbours, s->itr_max, s->delta, 1, 1.0, 0.05, 0x3f, 1};
    oracle_match_result r;
    memcpy(T, s->T0 + 16 * (size_t)k, 16 * sizeof(double));
    oracle_match_d2d(a, b, T, &p, &r);
    *iters = r.iterations;
    oracle_map_destroy(a);
    oracle_map_destroy(b);
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s sample.bin reps threads\n", argv[0]); return 2; }
    const int reps = atoi(argv[2]);
    int threads = atoi(argv[3]);
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    sample s;
    if (fread(&s.n_pairs, 4, 1, f) != 1 || fread(&s.n_points, 4, 1, f) != 1 || fread(&s.res, 8, 1, f) != 1 ||
        fread(s.size, 8, 3, f) != 3 || fread(&s.range, 8, 1, f) != 1 || fread(&s.delta, 8, 1, f) != 1 ||
        fread(&s.n_neighbours, 4, 1, f) != 1 || fread(&s.itr_max, 4, 1, f) != 1) { fprintf(stderr, "short header\n"); return 1; }
    const size_t per_pair = (size_t)s.n_points * 6;
    float *scans = (float *)malloc((size_t)s.n_pairs * per_pair * sizeof(float));
    double *T0 = (double *)malloc((size_t)s.n_pairs * 16 * sizeof(double));
    double *Tout = (double *)malloc((size_t)s.n_pairs * 16 * sizeof(double));
    if (!scans || !T0 || !Tout) { fprintf(stderr, "out of memory\n"); return 1; }
    for (int k = 0; k < s.n_pairs; k++)
        if (fread(scans + (size_t)k * per_pair, sizeof(float), per_pair, f) != per_pair || fread(T0 + 16 * (size_t)k, 8, 16, f) != 16) {
            fprintf(stderr, "short sample\n");
            return 1;
        }
    fclose(f);
    s.scans = scans;
    s.T0 = T0;
#ifdef _OPENMP
    if (threads < 1) threads = omp_get_max_threads();
    omp_set_num_threads(threads);
#else
    threads = 1;
#endif
    long iters_total = 0;
    double *pass = (double *)malloc((size_t)(reps > 0 ? reps : 1) * sizeof(double));
    for (int rep = -1; rep < reps; rep++) {                 /* rep -1: warm-up (page faults, caches, clocks) */
        const int n = (rep < 0) ? (s.n_pairs < 8 * threads ? s.n_pairs : 8 * threads) : s.n_pairs;
        long it_sum = 0;
        const double t0 = now_s();
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : it_sum)
#endif
        for (int k = 0; k < n; k++) {
            int it = 0;
            register_pair(&s, k, Tout + 16 * (size_t)k, &it);
            it_sum += it;
        }
        const double dt = now_s() - t0;
        if (rep >= 0) { pass[rep] = dt; iters_total = it_sum; }
    }
    double *sorted = (double *)malloc((size_t)reps * sizeof(double));
    memcpy(sorted, pass, (size_t)reps * sizeof(double));
    qsort(sorted, (size_t)reps, sizeof(double), cmp_d);
    const double med = (reps % 2) ? sorted[reps / 2] : 0.5 * (sorted[reps / 2 - 1] + sorted[reps / 2]);
    char outname[4096];
    snprintf(outname, sizeof outname, "%s.out", argv[1]);
    FILE *o = fopen(outname, "wb");
    if (o) { fwrite(Tout, 8, (size_t)s.n_pairs * 16, o); fclose(o); }
    printf("{\"pairs\": %d, \"points\": %d, \"reps\": %d, \"threads\": %d, \"median_pass_s\": %.6f, \"min_pass_s\": %.6f, "
           "\"max_pass_s\": %.6f, \"registrations_per_s\": %.4f, \"mean_iterations\": %.3f}\n",
           s.n_pairs, s.n_points, reps, threads, med, sorted[0], sorted[reps - 1], (double)s.n_pairs / med,
           (double)iters_total / (double)s.n_pairs);
    return 0;
}
