// Host side of the ground-truth evaluation: best-match assignment and the seven scores of
// Testing::eval_performance (reference src/testing.cpp:88-136, 239-406), computed from the
// contingency table the GPU kernels d_contingency / d_contingency_ghost produce.
// table[i*M + j] = voxels shared by segment i and truth label j; ssize[i], tsize[j] the cloud sizes.
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
#include <vector>
#include "../../include/f3ds.h"

inline f3ds_performance f3ds_scores_from_table(const std::vector<uint32_t>& table, const std::vector<uint32_t>& ssize,
                                               const std::vector<uint32_t>& tsize, uint32_t n_truth_points) {
    const size_t K = ssize.size(), M = tsize.size();
    // truth labels are visited by descending size; labels of equal size share one map slot, so only
    // the lowest of them is ever matched (std::map::insert keeps the first, testing.cpp:97-100)
    std::map<uint32_t, uint32_t> by_size;
    if (K) for (size_t j = 0; j < M; ++j) by_size.insert({tsize[j], (uint32_t)j});
    std::vector<int64_t> match(M, -1);
    std::vector<unsigned char> used(K, 0);
    std::vector<uint32_t> col(K);
    for (auto it = by_size.rbegin(); it != by_size.rend(); ++it) {
        const uint32_t j = it->second;
        for (size_t i = 0; i < K; ++i) col[i] = table[i * M + j];
        int64_t row = -1;
        for (;;) {
            int64_t best = 0;
            for (size_t i = 1; i < K; ++i) if (col[i] > col[best]) best = (int64_t)i;    // first maximum
            if (!used[best]) { row = best; break; }
            col[best] = 0;
            bool any = false;
            for (size_t i = 0; i < K && !any; ++i) any = col[i] != 0;
            if (!any) break;
        }
        match[j] = row;
        if (row >= 0) used[row] = 1;
    }
    f3ds_performance out;
    const float N = (float)n_truth_points;
    float h_s = 0, h_t = 0, mi = 0;
    for (size_t i = 0; i < K; ++i) {
        const float p = (float)ssize[i];
        h_s -= std::log(p / N) * p / N;
        for (size_t j = 0; j < M; ++j) {
            const float q = (float)tsize[j];
            if (i == 0) h_t -= std::log(q / N) * q / N;
            const float r = (float)table[i * M + j];
            if (r != 0) mi += std::log((N * r) / (p * q)) * r / N;
        }
    }
    out.voi = h_s + h_t - 2 * mi;
    float p = 0, r = 0, fp = 0, fn = 0, w = 0;
    for (size_t j = 0; j < M; ++j) {
        const float g = (float)tsize[j];
        if (match[j] < 0) { fn += g; continue; }
        const size_t i = (size_t)match[j];
        const float in = (float)table[i * M + j], s = (float)ssize[i];
        p += in * g / s; r += in; fp += (s - in); fn += (g - in);
        const float un = (float)(ssize[i] + tsize[j] - table[i * M + j]);      // |A u B| of the two multisets
        w += in * g / un;
    }
    out.precision = p / N; out.recall = r / N; out.fpr = fp / N; out.fnr = fn / N;
    out.fscore = (out.precision == 0 && out.recall == 0) ? 0.0f : 2 * (out.precision * out.recall) / (out.precision + out.recall);
    out.wov = w / N;
    return out;
}
