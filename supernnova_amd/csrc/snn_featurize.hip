// snn_featurize.hip -- host-only: the row-building half of the on-the-fly (broker) featuriser, one native pass per
// stage instead of numpy sorts over all observations (validation/validate_onthefly.py -> the reference's
// utils/data_utils.compute_delta_time :364-390 and data/make_dataset.pivot_dataframe_single_from_df :800-938).
//   snn_featurize_plan   observations -> time order inside each light curve, 8-hour groups, output row of every observation
//   snn_featurize_fill   per output row and filter: flux / error of the observation with the lowest error
// No device code; compiled into libsnn_hip.so with the rest.  Light curves are independent: both stages run over
// ranges of curves on a few threads.
#include <stdint.h>
#include <algorithm>
#include <thread>
#include <vector>

#include "../../include/snn_hip.h"
#include "snn_host.h"

namespace {
template <class F> void parallel_ranges(int64_t n_items, F&& fn)
{
    const int64_t nt = std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), 16, n_items / 2048}));
    if (nt == 1) { fn(0, n_items); return; }
    std::vector<std::thread> th;
    for (int64_t k = 0; k < nt; ++k) th.emplace_back([&, k] { fn(n_items * k / nt, n_items * (k + 1) / nt); });
    for (auto& t : th) t.join();
}
}  // namespace

// order[n]: observations sorted by (curve, MJD), stable.  row_of[n]: for the j-th observation IN THAT ORDER, its output
// row (rows are numbered curve by curve, group by group).  rows_per_curve[n_curves].  Returns the number of rows.
// Group rule (make_dataset.py:818-832, sequential): an observation opens a new group when it is the first of its
// curve, when its time equals the previous observation's (delta_time == 0), or when it lies more than min_dt after
// the first time of the running group; a new group whose first time equals the running group's joins its row.
extern "C" int64_t snn_featurize_plan(int64_t n, int64_t n_curves, const int64_t* curve, const double* mjd, double min_dt,
                                      int64_t* order, int64_t* row_of, double* row_mjd_of, int64_t* rows_per_curve)
{
    if (n < 0 || n_curves < 0 || (n > 0 && (!curve || !mjd || !order || !row_of || !row_mjd_of || !rows_per_curve))) {
        snn_set_error("snn_featurize_plan: null pointer or negative size");
        return SNN_ERR_ARG;
    }
    std::vector<int64_t> start(n_curves + 1, 0);
    for (int64_t i = 0; i < n; ++i) {
        if (curve[i] < 0 || curve[i] >= n_curves) { snn_set_error("snn_featurize_plan: curve index out of range"); return SNN_ERR_ARG; }
        ++start[curve[i] + 1];
    }
    for (int64_t c = 0; c < n_curves; ++c) start[c + 1] += start[c];
    {
        std::vector<int64_t> pos(start.begin(), start.end() - 1);
        for (int64_t i = 0; i < n; ++i) order[pos[curve[i]]++] = i;     // counting sort by curve: stable
    }
    parallel_ranges(n_curves, [&](int64_t c0, int64_t c1) {
        for (int64_t c = c0; c < c1; ++c) {
            int64_t* o = order + start[c];
            const int64_t m = start[c + 1] - start[c];
            bool sorted = true;
            for (int64_t j = 1; j < m && sorted; ++j) sorted = mjd[o[j]] >= mjd[o[j - 1]];
            if (!sorted) std::stable_sort(o, o + m, [&](int64_t a, int64_t b) { return mjd[a] < mjd[b]; });
            int64_t rows = 0;
            double last = 0.0, prev_group = 0.0;
            for (int64_t j = 0; j < m; ++j) {
                const double t = mjd[o[j]];
                const double dt = j == 0 ? 0.0 : t - mjd[o[j - 1]];
                double g;
                if (dt == 0.0 || (t - last) > min_dt) g = last = t;
                else g = prev_group;
                if (j == 0 || g != prev_group) ++rows;
                prev_group = g;
                row_of[start[c] + j] = rows - 1;          // row inside the curve; made global below
                row_mjd_of[start[c] + j] = g;
            }
            rows_per_curve[c] = rows;
        }
    });
    int64_t total = 0;
    std::vector<int64_t> base(n_curves);
    for (int64_t c = 0; c < n_curves; ++c) { base[c] = total; total += rows_per_curve[c]; }
    parallel_ranges(n_curves, [&](int64_t c0, int64_t c1) {
        for (int64_t c = c0; c < c1; ++c)
            for (int64_t j = start[c]; j < start[c + 1]; ++j) row_of[j] += base[c];
    });
    return total;
}

// flux_out / err_out [n_rows][F] (zero-initialised by the caller), present_out [n_rows][F] bytes (zeroed), row_mjd[n_rows]:
// for every (row, filter) the observation with the lowest error, the earliest one on ties; filt < 0 (a filter the
// model does not know) fills nothing.
extern "C" int snn_featurize_fill(int64_t n, int64_t n_rows, int32_t F, const int64_t* order, const int64_t* row_of,
                                  const double* row_mjd_of, const double* flux, const double* err, const int64_t* filt,
                                  double* flux_out, double* err_out, uint8_t* present_out, double* row_mjd)
{
    if (n < 0 || n_rows < 0 || F < 1 || (n > 0 && (!order || !row_of || !row_mjd_of || !flux || !err || !filt || !flux_out ||
                                                  !err_out || !present_out || !row_mjd))) {
        snn_set_error("snn_featurize_fill: null pointer or bad size");
        return SNN_ERR_ARG;
    }
    for (int64_t j = 0; j < n; ++j) {                 // rows of one light curve are contiguous: no two threads share a row
        const int64_t i = order[j], r = row_of[j];
        if (i < 0 || i >= n) { snn_set_error("snn_featurize_fill: order[%lld] = %lld outside [0, n)", (long long)j, (long long)i); return SNN_ERR_ARG; }
        const int64_t k = filt[i];
        if (r < 0 || r >= n_rows || k >= F) { snn_set_error("snn_featurize_fill: row or filter index out of range"); return SNN_ERR_ARG; }
        row_mjd[r] = row_mjd_of[j];
        if (k < 0) continue;
        const int64_t cell = r * F + k;
        if (!present_out[cell] || err[i] < err_out[cell]) {
            present_out[cell] = 1;
            flux_out[cell] = flux[i];
            err_out[cell] = err[i];
        }
    }
    return SNN_OK;
}
