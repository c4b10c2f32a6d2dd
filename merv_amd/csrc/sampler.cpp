// Frame-index sampler: host-side, float64, bit-exact with numpy.linspace(..., dtype=int) as used by
// merv/preprocessing/datasets/datasets.py:131-141 of the reference. Compiled with -ffp-contract=off so that
// "k*step + start" and "total*fps - 1" are two roundings each, exactly like numpy / CPython.
#include <math.h>
#include <stdint.h>

#include "../../include/merv_hip.h"

extern "C" void merv_set_error(const char* msg);  // capi.hip

namespace {

// numpy.linspace(start, stop, num, endpoint=True, dtype=int): float64 ramp, last sample forced to `stop`,
// floor, cast. (numpy/_core/function_base.py; "any_step_zero" branch kept for denormal steps.)
void linspace_int(double start, double stop, int32_t num, int64_t* out) {
    if (num <= 0) return;
    const int32_t div = num - 1;
    const double delta = stop - start;
    if (div > 0) {
        const double step = delta / (double)div;
        for (int32_t k = 0; k < num; ++k) {
            double y;
            if (step == 0.0) {
                y = (double)k / (double)div;
                y = y * delta;
            } else {
                y = (double)k * step;
            }
            y = y + start;
            if (k == num - 1) y = stop;
            out[k] = (int64_t)floor(y);
        }
    } else {
        double y = 0.0 * delta;
        y = y + start;
        out[0] = (int64_t)floor(y);
    }
}

}  // namespace

extern "C" int merv_frame_indices(int64_t video_num_frames, double avg_fps, double clip_start_sec, double clip_end_sec,
                                  int64_t end_frame, int32_t num_frames, int64_t* out_ids) {
    if (!out_ids || num_frames < 0) {
        merv_set_error("merv_frame_indices: bad arguments");
        return 1;
    }
    if (video_num_frames <= 0) {
        merv_set_error("merv_frame_indices: empty video");
        return 1;
    }
    // datasets.py:46-52: NaN start -> 0, NaN end -> None
    if (isnan(clip_start_sec)) clip_start_sec = 0.0;
    const double last = (double)(video_num_frames - 1);
    double start, stop;
    if (end_frame < 0) {  // datasets.py:131-137 ("end_frame is None or end_frame < 0")
        if (isnan(clip_end_sec)) {
            const double total_secs = (double)video_num_frames / avg_fps;  // :128
            clip_end_sec = total_secs;
        }
        start = clip_start_sec * avg_fps;
        const double cand = clip_end_sec * avg_fps - 1.0;
        stop = (cand < last) ? cand : last;  // Python min(N-1, cand): returns N-1 unless cand is strictly smaller
    } else {  // :138-141
        start = 0.0;
        const double cand = (double)end_frame;
        stop = (cand < last) ? cand : last;
    }
    linspace_int(start, stop, num_frames, out_ids);
    return 0;
}

extern "C" int merv_temporal_subsample(int32_t loaded_frames, int32_t max_nf, int32_t nf, int32_t* out_idx,
                                       int32_t* out_n) {
    if (nf <= 0 || max_nf <= 0 || loaded_frames < 0 || !out_n) {
        merv_set_error("merv_temporal_subsample: bad arguments");
        return 1;
    }
    const int32_t stride = max_nf / nf;  // merv.py:804  video[:: max(num_frames) // nf]
    if (stride == 0) {
        merv_set_error("merv_temporal_subsample: slice step cannot be zero (nf > max_nf)");
        return 1;
    }
    int32_t n = 0;
    for (int32_t i = 0; i < loaded_frames; i += stride) {
        if (out_idx) out_idx[n] = i;
        ++n;
    }
    *out_n = n;
    return 0;
}
