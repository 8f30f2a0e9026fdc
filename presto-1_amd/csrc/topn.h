// topn.h -- TopN on the GPU: the n first rows of the input in the order of the sort channels.
// Reference: M/operator/TopNOperator.java:47-62, TopNProcessor.java:45-66 (a GroupedTopNRowNumberBuilder without grouping) and the
// row order of SimplePageWithPositionComparator.java:58-79 + TypeOperators.java:578-596 (nulls placed by the SortOrder, values by
// the type's COMPARISON operator, negated for DESC).
#pragma once

#include "common.h"
#include "join.h"

namespace tgpu {

class TopNGpu {
public:
    TopNGpu(Context *ctx, std::vector<int32_t> types, int64_t n, std::vector<int32_t> sort_channels, std::vector<int32_t> sort_orders);
    // keeps the page's n first rows (in sort order); the page itself is not retained
    void add_page(const DevicePage &page);
    // the n first rows of everything added so far, in sort order (rows that compare equal keep their input order)
    DevicePage result();
    int64_t estimated_size() const { return kept_.estimated_size(); }

    // row numbers of `page`'s first min(limit, rows) rows in the order of the sort channels (rows that compare equal keep their
    // input order); limit >= rows sorts the whole page (OrderByOperator)
    static BufferPtr sorted_positions(Context *ctx, const DevicePage &page, const std::vector<int32_t> &sort_channels, const std::vector<int32_t> &sort_orders,
                                      int64_t limit, int64_t &count);

private:
    BufferPtr top_positions(const DevicePage &page, int64_t &count) { return sorted_positions(ctx_, page, sort_channels_, sort_orders_, n_, count); }
    Context *ctx_;
    std::vector<int32_t> types_, sort_channels_, sort_orders_;
    int64_t n_;
    PagesIndexGpu kept_;   // candidates: the per-page winners, appended page after page
    bool sorted_ = false;  // the store holds one selection result (at most n rows, in order)
};

}  // namespace tgpu
