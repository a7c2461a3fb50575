/*
 * c_host.c — the C ABI from plain C (no Python, no torch, no JavaScript): plan a batch, run it on host PCM, print
 * the callback rows.  Shows what any FFI (cgo, JNI, ...) would bind; used by tests/test_c_host.py.
 *
 *   cc -std=c99 -Iinclude examples/c_host.c -Lwebspeechanalyzer_amd/lib -lwsa -Wl,-rpath,$PWD/webspeechanalyzer_amd/lib -o c_host
 *   ./c_host level fs clip0.f32 [clip1.f32 ...]        (raw little-endian float32 mono files)
 * Output: one line per feature row: clip si t_start t_len feature[0] feature[1] ... (level 5 / 13: 53 features).
 */
#include <stdio.h>
#include <stdlib.h>
#include "wsa.h"

static float *read_f32(const char *path, uint32_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    float *p = (float *)malloc(sz > 0 ? (size_t)sz : 4);
    if (fread(p, 1, (size_t)sz, f) != (size_t)sz) { perror("read"); exit(2); }
    fclose(f);
    *n = (uint32_t)(sz / 4);
    return p;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s level fs clip.f32...\n", argv[0]); return 2; }
    wsa_config cfg;
    wsa_config_default(&cfg);
    cfg.output_level = atoi(argv[1]);
    const double fs = atof(argv[2]);
    const uint32_t n_clips = (uint32_t)(argc - 3);
    uint32_t *n_samples = (uint32_t *)malloc(sizeof(uint32_t) * n_clips);
    const float **pcm = (const float **)malloc(sizeof(float *) * n_clips);
    for (uint32_t i = 0; i < n_clips; i++) pcm[i] = read_f32(argv[3 + i], &n_samples[i]);

    wsa_ctx *ctx = NULL;
    if (wsa_create(&cfg, 0, &ctx) != WSA_OK) { fprintf(stderr, "wsa_create: %s\n", wsa_last_error(NULL)); return 1; }
    wsa_batch *b = NULL;
    if (wsa_batch_create(ctx, n_clips, n_samples, fs, &b) != WSA_OK) { fprintf(stderr, "wsa_batch_create: %s\n", wsa_last_error(ctx)); return 1; }
    if (wsa_batch_run_host(b, pcm, NULL) != WSA_OK) { fprintf(stderr, "wsa_batch_run_host: %s\n", wsa_last_error(ctx)); return 1; }
    wsa_device_result r;
    if (wsa_batch_result(b, NULL, &r) != WSA_OK) { fprintf(stderr, "wsa_batch_result: %s\n", wsa_last_error(ctx)); return 1; }
    int32_t *meta = (int32_t *)malloc(sizeof(int32_t) * 8 * (r.n_rows ? r.n_rows : 1));
    double *feat = (double *)malloc(sizeof(double) * WSA_NFEAT * (r.n_rows ? r.n_rows : 1));
    if (wsa_batch_copy_rows(b, NULL, meta, feat, r.n_rows ? r.n_rows : 1, NULL, 0, NULL, NULL) != WSA_OK) {
        fprintf(stderr, "wsa_batch_copy_rows: %s\n", wsa_last_error(ctx)); return 1;
    }
    for (uint32_t k = 0; k < r.n_rows; k++) {
        printf("%d %d %d %d", meta[8 * k], meta[8 * k + 1], meta[8 * k + 2], meta[8 * k + 3]);
        for (int j = 0; j < WSA_NFEAT; j++) printf(" %.17g", feat[(size_t)k * WSA_NFEAT + j]);
        printf("\n");
    }
    fprintf(stderr, "%u clips, %u frames, %u segments, %u rows\n", r.n_clips, r.n_frames_total, r.n_segments, r.n_rows);
    wsa_batch_destroy(b);
    wsa_destroy(ctx);
    return 0;
}
