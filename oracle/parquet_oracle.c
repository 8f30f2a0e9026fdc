/* parquet_oracle.c -- TEST INFRASTRUCTURE ONLY (like the rest of oracle/): a plain-C, value-at-a-time restatement of the Parquet byte-level
 * decoders the reference's page readers call (lib/trino-parquet/src/main/java/io/trino/parquet/reader/PrimitiveColumnReader.java readPageV1 /
 * initDataReader, LevelRLEReader.java, dictionary/DictionaryReader.java).  Those decoders are NOT in the reference tree: they are parquet-mr's
 * (org.apache.parquet classes -- RunLengthBitPackingHybridDecoder, PlainValuesReader, BinaryPlainValuesReader, BooleanPlainValuesReader -- that the
 * reference takes from io.prestosql.hive:hive-apache 3.1.2-6, a shaded bundle its root pom.xml:537-538 pins; absent from /root/reference), a third-party dependency; restated from the public Parquet format specification (Encodings.md: "Run Length Encoding / Bit-Packing Hybrid",
 * "Plain").  Pinned (tests/test_parquet_oracle_cpu.py) on pages written by Apache Arrow's Parquet writer and the values its reader returns
 * (an independent implementation of the same specification) and on the one Parquet file the reference's tests hold that these types cover
 * (testing/trino-product-tests/.../hive/data/single_int_column/data.parquet).  The reference holds no decoded golden values of its own for
 * these decoders: beyond those two pins the parity of this restatement is "format-pinned".
 */
#include <stdint.h>
#include <string.h>

/* RLE / bit-packed hybrid: `want` values of `bit_width` bits; returns the number decoded or -1 (corrupt / truncated) */
int64_t o_pq_hybrid(const uint8_t *bytes, int64_t len, int32_t bit_width, int32_t *out, int64_t want)
{
    int64_t at = 0, n = 0;
    if (bit_width < 0 || bit_width > 32) return -1;
    if (bit_width == 0) {
        for (; n < want; n++) out[n] = 0;
        return n;
    }
    while (n < want) {
        uint64_t header = 0;
        int shift = 0;
        for (;;) {   /* ULEB128 */
            if (at >= len || shift > 35) return -1;
            const int b = bytes[at++];
            header |= (uint64_t)(b & 0x7f) << shift;
            shift += 7;
            if (!(b & 0x80)) break;
        }
        if (header & 1) {   /* bit-packed: (header >> 1) groups of 8 values, least significant bit first */
            const int64_t values = (int64_t)(header >> 1) * 8;
            int64_t bit = 0;
            for (int64_t i = 0; i < values && n < want; i++, bit += bit_width) {
                uint64_t v = 0;
                for (int k = 0; k < bit_width; k++) {
                    const int64_t b = bit + k, byte = at + (b >> 3);
                    if (byte >= len) return -1;
                    v |= (uint64_t)((bytes[byte] >> (b & 7)) & 1) << k;
                }
                out[n++] = (int32_t)v;
            }
            at += (int64_t)(header >> 1) * bit_width;
            if (at > len) at = len;
        }
        else {   /* RLE: header >> 1 copies of a value of ceil(bit_width / 8) little-endian bytes */
            const int64_t count = (int64_t)(header >> 1);
            const int width_bytes = (bit_width + 7) / 8;
            uint64_t v = 0;
            if (count == 0 || at + width_bytes > len) return -1;
            for (int i = 0; i < width_bytes; i++) v |= (uint64_t)bytes[at++] << (8 * i);
            for (int64_t i = 0; i < count && n < want; i++) out[n++] = (int32_t)v;
        }
    }
    return n;
}

/* PLAIN BYTE_ARRAY: `count` values, each a 4-byte little-endian length and its bytes: offsets[count + 1] into `pool`; returns bytes or -1 */
int64_t o_pq_plain_byte_array(const uint8_t *bytes, int64_t len, int64_t count, int32_t *offsets, uint8_t *pool, int64_t pool_cap)
{
    int64_t at = 0, used = 0;
    offsets[0] = 0;
    for (int64_t i = 0; i < count; i++) {
        uint32_t l;
        if (at + 4 > len) return -1;
        memcpy(&l, bytes + at, 4);
        at += 4;
        if (l > 0x7fffffffu || at + (int64_t)l > len || used + (int64_t)l > pool_cap) return -1;
        memcpy(pool + used, bytes + at, l);
        at += l;
        used += l;
        offsets[i + 1] = (int32_t)used;
    }
    return used;
}

/* PLAIN BOOLEAN: one bit per value, least significant bit first */
int64_t o_pq_plain_boolean(const uint8_t *bytes, int64_t len, int64_t count, uint8_t *out)
{
    if (len < (count + 7) / 8) return -1;
    for (int64_t i = 0; i < count; i++) out[i] = (bytes[i >> 3] >> (i & 7)) & 1;
    return count;
}

/* DELTA_BINARY_PACKED (Encodings.md "Delta Encoding"; parquet-mr DeltaBinaryPackingValuesReader, which ParquetEncoding.java:146-154 hands INT32 /
 * INT64 pages to): header = block size, miniblocks per block, total value count (ULEB128), first value (zigzag ULEB128); then blocks: min delta
 * (zigzag ULEB128), one bit-width byte per miniblock, the miniblocks' deltas minus the min delta, bit-packed least significant bit first (a
 * miniblock is always written whole; miniblocks past the last value are not written).  value[i] = value[i - 1] + min delta + packed delta, in
 * wrapping 64-bit arithmetic (`bits` = 32: the INT32 reader keeps the low 32 bits).  Returns the number of values or -1. */
static int pq_uleb(const uint8_t *bytes, int64_t len, int64_t *at, uint64_t *out)
{
    uint64_t v = 0;
    int shift = 0;
    for (;;) {
        if (*at >= len || shift > 63) return -1;
        const int b = bytes[(*at)++];
        v |= (uint64_t)(b & 0x7f) << shift;
        shift += 7;
        if (!(b & 0x80)) break;
    }
    *out = v;
    return 0;
}

/* *consumed (may be null) = the bytes the section takes when it holds exactly `want` values: what DELTA_LENGTH_BYTE_ARRAY needs to find the strings */
int64_t o_pq_delta_binary_packed(const uint8_t *bytes, int64_t len, int64_t want, int32_t bits, int64_t *out, int64_t *consumed)
{
    int64_t at = 0;
    uint64_t block_size, miniblocks, total, zz;
    if (pq_uleb(bytes, len, &at, &block_size) || pq_uleb(bytes, len, &at, &miniblocks) || pq_uleb(bytes, len, &at, &total) || pq_uleb(bytes, len, &at, &zz)) return -1;
    if (miniblocks == 0 || miniblocks > 4096 || block_size == 0 || block_size % 128 || block_size % miniblocks || (block_size / miniblocks) % 32) return -1;
    if ((int64_t)total < want) return -1;
    if (consumed) *consumed = at;
    if (want == 0) return 0;
    const int64_t mini = (int64_t)(block_size / miniblocks);
    uint64_t value = (zz >> 1) ^ (~(zz & 1) + 1);   /* zigzag */
    int64_t n = 0;
    out[n++] = bits == 32 ? (int64_t)(int32_t)value : (int64_t)value;
    while (n < want) {
        uint64_t zmin;
        if (pq_uleb(bytes, len, &at, &zmin)) return -1;
        const uint64_t min_delta = (zmin >> 1) ^ (~(zmin & 1) + 1);
        if (at + (int64_t)miniblocks > len) return -1;
        const uint8_t *widths = bytes + at;
        at += (int64_t)miniblocks;
        for (uint64_t m = 0; m < miniblocks && n < want; m++) {
            const int w = widths[m];
            if (w > 64 || at + mini * w / 8 > len) return -1;
            for (int64_t i = 0; i < mini && n < want; i++) {
                uint64_t d = 0;
                for (int k = 0; k < w; k++) {
                    const int64_t b = i * w + k;
                    d |= (uint64_t)((bytes[at + (b >> 3)] >> (b & 7)) & 1) << k;
                }
                value += min_delta + d;
                out[n++] = bits == 32 ? (int64_t)(int32_t)value : (int64_t)value;
            }
            at += mini * w / 8;
        }
    }
    if (consumed) *consumed = at;
    return n;
}

/* DELTA_BYTE_ARRAY (Encodings.md "Delta Strings"; ParquetEncoding.java:165-173 -> parquet-mr's DeltaByteArrayReader): the prefix lengths as a
 * DELTA_BINARY_PACKED section, then the suffixes as a DELTA_LENGTH_BYTE_ARRAY section; value[i] = value[i - 1][0 .. prefix[i]) ++ suffix[i].
 * offsets[count + 1] into pool; returns the bytes written or -1. */
int64_t o_pq_delta_byte_array(const uint8_t *bytes, int64_t len, int64_t count, int64_t *scratch /* 2 * count */, int32_t *offsets, uint8_t *pool, int64_t pool_cap)
{
    int64_t used_a = 0, used_b = 0;
    int64_t *prefix = scratch, *slen = scratch + count;
    offsets[0] = 0;
    if (count == 0) return 0;
    if (o_pq_delta_binary_packed(bytes, len, count, 32, prefix, &used_a) != count) return -1;
    if (o_pq_delta_binary_packed(bytes + used_a, len - used_a, count, 32, slen, &used_b) != count) return -1;
    int64_t at = used_a + used_b, out = 0, prev = 0, prev_len = 0;
    for (int64_t i = 0; i < count; i++) {
        if (prefix[i] < 0 || prefix[i] > prev_len || slen[i] < 0 || at + slen[i] > len || out + prefix[i] + slen[i] > pool_cap) return -1;
        memmove(pool + out, pool + prev, (size_t)prefix[i]);
        memcpy(pool + out + prefix[i], bytes + at, (size_t)slen[i]);
        at += slen[i];
        prev = out;
        prev_len = prefix[i] + slen[i];
        out += prev_len;
        if (out > 0x7fffffffLL) return -1;
        offsets[i + 1] = (int32_t)out;
    }
    return out;
}
