// orc.h -- scan-side decode, first slice: ORC stripe streams -> flat device columns.  orc.hip
#pragma once

#include "common.h"

namespace tgpu {
namespace orc {

// LongColumnReader (lib/trino-orc/src/main/java/io/trino/orc/reader/LongColumnReader.java:100-230): PRESENT (optional) + DATA (signed RLEv2) of
// one column of one stripe / row group -> a BIGINT / INTEGER / DATE column of n positions
DeviceColumn decode_long_column(Context *ctx, int32_t type, int32_t encoding, int64_t n, const uint8_t *present, int64_t present_len, const uint8_t *data, int64_t data_len);
// BooleanColumnReader: PRESENT (optional) + DATA (a boolean stream: byte RLE, bits most significant first)
DeviceColumn decode_boolean_column(Context *ctx, int64_t n, const uint8_t *present, int64_t present_len, const uint8_t *data, int64_t data_len);
// SliceDictionaryColumnReader (reader/SliceDictionaryColumnReader.java:120-330): DATA (unsigned RLEv2 ids) + LENGTH (unsigned RLEv2 entry lengths) +
// DICTIONARY_DATA (the entries' bytes back to back) -> a flat VARCHAR column
DeviceColumn decode_direct_string_column(Context *ctx, int32_t encoding, int64_t n, const uint8_t *present, int64_t present_len, const uint8_t *data, int64_t data_len,
                                         const uint8_t *length_stream, int64_t length_len);
DeviceColumn decode_double_column(Context *ctx, int64_t n, const uint8_t *present, int64_t present_len, const uint8_t *data, int64_t data_len);
DeviceColumn decode_dictionary_string_column(Context *ctx, int32_t encoding, int64_t n, const uint8_t *present, int64_t present_len, const uint8_t *data, int64_t data_len,
                                             int32_t dictionary_size, const uint8_t *length_stream, int64_t length_len, const uint8_t *dictionary_data, int64_t dictionary_data_len);

}  // namespace orc
}  // namespace tgpu
