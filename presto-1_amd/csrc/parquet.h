// parquet.h -- scan-side decode, second format: the data pages of a flat Parquet column -> flat device columns.  parquet.hip
#pragma once

#include "common.h"

namespace tgpu {
namespace parquet {

// PrimitiveColumnReader.readPageV1 / readPageV2 + the type's reader (lib/trino-parquet/src/main/java/io/trino/parquet/reader/): one data page of a
// FLAT column (no repetition levels, definition level 0 or 1) -> a column of n positions.  `def_levels`: the page's definition levels as an
// RLE / bit-packed hybrid of bit width 1 WITHOUT the 4-byte length a V1 page puts in front (null: a required column); `values`: the value
// section; `dictionary` / `dictionary_count`: the chunk's PLAIN dictionary page for the dictionary encodings.
DeviceColumn decode_data_page(Context *ctx, int32_t type, int32_t physical, int32_t encoding, int64_t n, const uint8_t *def_levels, int64_t def_len, const uint8_t *values,
                              int64_t values_len, const uint8_t *dictionary, int64_t dictionary_len, int32_t dictionary_count);

}  // namespace parquet
}  // namespace tgpu
