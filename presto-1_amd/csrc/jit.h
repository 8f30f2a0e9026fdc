// jit.h -- the GPU counterpart of the reference's expression codegen (M/sql/gen/ExpressionCompiler.java:94-122,
// PageFunctionCompiler.java:164-212,367-404): a RowExpression filter + projections is compiled (hiprtc, gfx950) into one
// fused, row-selective kernel pair and cached on disk by source hash.
#pragma once

#include "common.h"

namespace tgpu {

constexpr int kFpMaxCols = 24;
constexpr int kFpMaxProj = 16;

struct FpArgs {  // must match the struct declared in the generated source
    const void *col_values[kFpMaxCols];
    const uint8_t *col_nulls[kFpMaxCols];
    const int32_t *col_offsets[kFpMaxCols];
    void *out_values[kFpMaxProj];
    uint8_t *out_nulls[kFpMaxProj];
    int32_t *positions;
    const int32_t *tile_offsets;
    int32_t *tile_counts;
    unsigned long long *error;
    long long n;
};

struct JitModule;

// PageProcessor (M/operator/project/PageProcessor.java:111-137): filter, then projections on the selected positions only.
class PageProcessorGpu {
public:
    PageProcessorGpu(std::vector<int32_t> input_types, const tgpu_page_processor_spec *spec);
    ~PageProcessorGpu();
    // generates the kernel source and compiles it into the on-disk cache; needs no GPU (used by build() to pre-warm)
    void precompile();
    // returns false when no row is selected (no output page); `out` gets one column per projection
    bool process(Context *ctx, const DevicePage &in, DevicePage &out);
    const std::vector<int32_t> &output_types() const { return output_types_; }
    const std::string &source() const { return source_; }

private:
    enum class ProjKind { COMPUTED, IDENTITY, CONSTANT_NULL };
    struct Proj {
        ProjKind kind;
        int channel = -1;   // IDENTITY
        int slot = -1;      // COMPUTED: index into FpArgs.out_*
        int32_t type = 0;
    };
    void generate();
    void ensure_loaded(Context *ctx);

    std::vector<int32_t> input_types_;
    std::vector<tgpu_expr_node> nodes_;
    std::string pool_;
    int filter_root_;
    std::vector<int32_t> proj_roots_;
    std::vector<Proj> projs_;
    std::vector<int32_t> output_types_;
    int computed_count_ = 0;
    std::string source_;
    std::shared_ptr<JitModule> module_;
};

std::string resource_dir();
void set_resource_dir(const std::string &dir);

}  // namespace tgpu
