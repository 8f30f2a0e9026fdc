// serde.h -- SerializedPage (the reference's exchange / spill page format) <-> device pages.  serde.hip
#pragma once

#include "common.h"

namespace tgpu {
namespace serde {

// PagesSerde.deserialize (M/execution/buffer/PagesSerde.java:117-160) of one uncompressed, unencrypted SerializedPage held in host
// memory, straight into flat HBM columns.  `types` are the channel types the consumer expects (the encodings do not tell BIGINT from
// DOUBLE or INTEGER from DATE).
DevicePage deserialize(Context *ctx, const uint8_t *bytes, int64_t len, const int32_t *types, int32_t type_count);

// PagesSerde.serialize (PagesSerde.java:64-115) + PagesSerdeUtil.writeSerializedPage of a device page into `out` (host);
// returns the number of bytes written.  out == nullptr: returns an upper bound of that number without doing any work.
int64_t serialize(Context *ctx, const DevicePage &page, uint8_t *out, int64_t capacity);

}  // namespace serde
}  // namespace tgpu
