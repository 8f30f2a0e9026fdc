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
