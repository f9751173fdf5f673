/* screen_windows.c -- the drop-in boundary used from plain C99 (no C++, no Python, no PyTorch): open a KMC database on a
 * GPU, cut a newline-free sequence into tiling windows (GetVariants.java:307-320), screen them in one call
 * (GetVariants.java:129-159, 203-258) and print what Window.addTotalKmers / setEffLength / addData would receive, plus the
 * identity score (Data.java:95-107).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/screen_windows.c -Lkcftools_amd -lkcfx -Wl,-rpath,$PWD/kcftools_amd -o screen_windows
 *   ./screen_windows <kmc prefix> <sequence file: raw bases, no header, newlines ignored> <window> [min count]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kcfx.h"

static int die(const char* what) {
    fprintf(stderr, "%s: %s\n", what, kcfx_last_error());
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <kmc prefix> <sequence file> <window> [min count]\n", argv[0]);
        return 2;
    }
    const int window = atoi(argv[3]);
    const int min_count = argc > 4 ? atoi(argv[4]) : 1;

    FILE* f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 1; }
    fseek(f, 0, SEEK_END);
    long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char* text = (unsigned char*)malloc((size_t)size + 1);
    long n = 0;
    for (int c; (c = fgetc(f)) != EOF;)
        if (c != '\n' && c != '\r') text[n++] = (unsigned char)c;
    fclose(f);

    kcfx_db_info_t info;
    if (kcfx_db_header(argv[1], &info)) return die("kcfx_db_header");
    const int k = info.kmer_length;

    int64_t n_win = 0;
    if (kcfx_windows((int32_t)n, window, 0, k, NULL, NULL, 0, &n_win)) return die("kcfx_windows");
    int32_t* starts = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_win + 1));
    int32_t* ends = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_win + 1));
    if (n_win && kcfx_windows((int32_t)n, window, 0, k, starts, ends, n_win, &n_win)) return die("kcfx_windows");
    int64_t* begin = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_win + 1));
    int32_t* len = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_win + 1));
    for (int64_t i = 0; i < n_win; ++i) {
        begin[i] = starts[i];
        len[i] = ends[i] - starts[i];
    }

    kcfx_db* db = NULL;
    if (kcfx_db_open(argv[1], 0, &db)) return die("kcfx_db_open");
    kcfx_result* res = (kcfx_result*)calloc((size_t)(n_win + 1), sizeof(kcfx_result));
    double* score = (double*)malloc(sizeof(double) * (size_t)(n_win + 1));
    double* depth = (double*)malloc(sizeof(double) * (size_t)(n_win + 1));
    if (kcfx_screen(db, text, n, begin, len, (int32_t)n_win, min_count, res)) return die("kcfx_screen");
    if (kcfx_score(res, n_win, 0.3, 0.3, 0.4, score, depth)) return die("kcfx_score");

    printf("start,end,total,efflen,obs,var,inner,left,right,count_sum,score,mean_count\n");
    for (int64_t i = 0; i < n_win; ++i)
        printf("%d,%d,%d,%d,%d,%d,%d,%d,%d,%lld,%.17g,%.17g\n", starts[i], ends[i], res[i].total_kmers, res[i].eff_length, res[i].obs,
               res[i].var, res[i].inner, res[i].left, res[i].right, (long long)res[i].kmer_count_sum, score[i], depth[i]);

    kcfx_db_close(db);
    kcfx_release_cached(-1);
    free(res); free(score); free(depth); free(begin); free(len); free(starts); free(ends); free(text);
    return 0;
}
