/*
 * frame_index_oracle.c -- plain-C restatement of the reference's frame-index selection.
 * *** TEST INFRASTRUCTURE ONLY *** (see oracle/merv_oracle.py header). Follows
 * merv/preprocessing/datasets/datasets.py:46-52 (NaN guards) and :126-141 (np.linspace(..., dtype=int)).
 * Written independently of merv_amd/csrc/sampler.cpp (different loop structure: it builds the float ramp first,
 * the way numpy does, then floors) so that the two can check each other.
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* clip_end_sec / end_frame "None" are passed as NaN / -1 */
int oracle_frame_indices(int64_t n_frames, double fps, double clip_start, double clip_end, int64_t end_frame,
                         int num, int64_t *out) {
    if (num <= 0) return 0;
    if (isnan(clip_start)) clip_start = 0.0;                 /* :46-48 */
    int end_is_none = isnan(clip_end);                       /* :50-52 */
    double total_secs = (double)n_frames / fps;              /* :128 */
    double start, stop;
    if (end_frame < 0) {                                     /* :131 */
        if (end_is_none) clip_end = total_secs;              /* :132-133 */
        start = clip_start * fps;
        double b = clip_end * fps - 1.0;
        double a = (double)(n_frames - 1);
        stop = b < a ? b : a;                                /* Python min(a, b) */
    } else {
        start = 0.0;
        double b = (double)end_frame;
        double a = (double)(n_frames - 1);
        stop = b < a ? b : a;                                /* :139-140 */
    }
    double *y = (double *)malloc(sizeof(double) * (size_t)num);
    if (!y) return -1;
    for (int k = 0; k < num; ++k) y[k] = (double)k;          /* arange(0, num) */
    int div = num - 1;
    double delta = stop - start;
    if (div > 0) {
        double step = delta / (double)div;
        if (step == 0.0) {
            for (int k = 0; k < num; ++k) { y[k] = y[k] / (double)div; y[k] = y[k] * delta; }
        } else {
            for (int k = 0; k < num; ++k) y[k] = y[k] * step;
        }
    } else {
        for (int k = 0; k < num; ++k) y[k] = y[k] * delta;
    }
    for (int k = 0; k < num; ++k) y[k] = y[k] + start;
    if (num > 1) y[num - 1] = stop;
    for (int k = 0; k < num; ++k) out[k] = (int64_t)floor(y[k]);
    free(y);
    return 0;
}
