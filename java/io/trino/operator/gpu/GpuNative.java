/*
 * Java glue of the MI355X operator hot path: binds libtgpu.so (include/tgpu.h) through the JNI shim jni/tgpu_jni.c.
 * Not compiled in this repository (its build image has no JDK); sources a maintainer adds to core/trino-main.
 */
package io.trino.operator.gpu;

import io.trino.spi.StandardErrorCode;
import io.trino.spi.TrinoException;

/** One static native method per C-ABI entry point the glue uses (jni/tgpu_jni.c). Handles are the library's opaque pointers. */
public final class GpuNative
{
    static {
        System.loadLibrary("tgpu_jni");     // links libtgpu.so
    }

    private GpuNative() {}

    /** Thrown by the shim for a negative status code; {@link #toTrinoException} maps it to the reference's error codes (tgpu.h). */
    public static final class NativeError
            extends RuntimeException
    {
        public final int code;

        public NativeError(int code, String message)
        {
            super(message);
            this.code = code;
        }
    }

    public static TrinoException toTrinoException(NativeError e)
    {
        StandardErrorCode code;
        switch (e.code) {
            case -2: code = StandardErrorCode.NUMERIC_VALUE_OUT_OF_RANGE; break;
            case -3: code = StandardErrorCode.GENERIC_INSUFFICIENT_RESOURCES; break;
            case -4: code = StandardErrorCode.COMPILER_ERROR; break;
            case -7: code = StandardErrorCode.DIVISION_BY_ZERO; break;
            case -8: code = StandardErrorCode.NOT_SUPPORTED; break;
            case -9: code = StandardErrorCode.INVALID_CAST_ARGUMENT; break;
            default: code = StandardErrorCode.GENERIC_INTERNAL_ERROR;   // -1, -5; -6 = the GPU / driver failed (no CPU fallback inside the library)
        }
        return new TrinoException(code, e.getMessage(), e);
    }

    // ---- context: one per worker and GPU (tgpu_context_*) ----
    public static native long createContext(int device);
    public static native void destroyContext(long context);
    public static native void synchronizeContext(long context);
    /** output pages cut at PageBuilder.isFull granularity in front of Java operators (tgpu_context_set_max_output_page); 0 = no limit */
    public static native void setMaxOutputPage(long context, long maxBytes, long maxRows);
    /** 0 = EXACT (correctly rounded sums), 1 = JAVA (row order, bit-identical to DoubleSumAggregation); tgpu_context_set_double_sum_order */
    public static native void setDoubleSumOrder(long context, int order);
    /** the embedding's promise that device blocks it passes with addInput stay untouched until the operator has finished: lets the fused
     * operators keep such pages by reference after the call (tgpu_context_set_device_input_stable); pages of this library never need it */
    public static native void setDeviceInputStable(long context, boolean stable);
    public static native void profileEnable(long context, boolean enabled);
    public static native String profileDump(long context);

    // ---- Operator protocol (io.trino.operator.Operator) ----
    public static native void addInput(long operator, int positions, int[] types, int[] encodings, int[] arrayOffsets, int[] dictionaryPositions,
            Object[] values, Object[] nulls, Object[] offsets, Object[] ids, Object[] dictionaryValues, Object[] dictionaryNulls, Object[] dictionaryOffsets);
    public static native void addInputDevicePage(long operator, long outputPage);
    public static native boolean needsInput(long operator);
    public static native boolean isFinished(long operator);
    public static native boolean isBlocked(long operator);
    public static native void finish(long operator);
    public static native long memoryBytes(long operator);
    public static native void close(long operator);
    public static native long revocableMemoryBytes(long operator);
    public static native void startMemoryRevoke(long operator);
    public static native void finishMemoryRevoke(long operator);
    public static native void setSpillEnabled(long factory, boolean enabled);
    public static native void setMaxPartialMemory(long factory, long bytes);
    public static native void spillStats(long operator, long[] countAndBytes);
    public static native long getOutput(long operator, boolean[] wouldBlock);

    // ---- output pages ----
    public static native int pagePositionCount(long page);
    public static native int pageChannelCount(long page);
    public static native void blockInfo(long page, int channel, long[] typeBytesNulls);
    public static native void copyBlocks(long page, Object[] values, Object[] nulls, Object[] offsets);
    public static native void releasePage(long page);
    public static native long deserializePage(long context, byte[] bytes, int offset, int length, int[] types);
    /** PagesSerde.serialize of a device-resident page; out == null returns an upper bound of the size */
    public static native long serializePage(long context, long page, byte[] out);

    // ---- factories (io.trino.operator.OperatorFactory).  An expression program travels as the six arrays of GpuRowExpressions.Program. ----
    public static native long createFilterProjectFactory(long context, int operatorId, int[] inputTypes, int[][] nodes, long[] longValues, double[] doubleValues,
            byte[] stringPool, int filterRoot, int[] projectionRoots);
    public static native long createScanFilterProjectFactory(long context, int operatorId, int[] types, int[][] nodes, long[] longValues, double[] doubleValues,
            byte[] stringPool, int filterRoot, int[] projectionRoots);
    public static native long createHashAggregationFactory(long context, int operatorId, int[] groupByTypes, int[] groupByChannels, int hashChannel, int step,
            int[] aggregates, int expectedGroups, boolean produceDefaultOutput);
    /** FilterAndProject feeding HashAggregation as one fused pipeline (tgpu_filter_project_hash_aggregation_factory_create) */
    public static native long createFilterProjectHashAggregationFactory(long context, int operatorId, int[] inputTypes, int[][] nodes, long[] longValues, double[] doubleValues,
            byte[] stringPool, int filterRoot, int[] projectionRoots, int[] groupByTypes, int[] groupByChannels, int hashChannel, int step, int[] aggregates, int expectedGroups);
    /** returns {factory, bridge}; partitionCount > 1: the PartitionedLookupSourceFactory protocol (one build operator per partition) */
    public static native long[] createHashBuilderFactory(long context, int operatorId, int[] types, int[] outputChannels, int[] hashChannels, int precomputedHashChannel,
            int expectedPositions, int partitionCount);
    public static native void setJoinFilter(long bridge, int[] probeTypes, int[][] nodes, long[] longValues, double[] doubleValues, byte[] stringPool, int filterRoot,
            int[] projectionRoots);
    public static native void lookupSourceStats(long bridge, long[] positionsSlotsLinks);
    public static native long createLookupJoinFactory(long context, int operatorId, long bridge, int[] probeTypes, int[] probeJoinChannels, int probeHashChannel,
            int[] probeOutputChannels, int joinType);
    /** FilterAndProject feeding LookupJoin as one fused pipeline (tgpu_filter_project_lookup_join_factory_create) */
    public static native long createFilterProjectLookupJoinFactory(long context, int operatorId, long bridge, int[] inputTypes, int[][] nodes, long[] longValues,
            double[] doubleValues, byte[] stringPool, int filterRoot, int[] projectionRoots, int[] probeJoinChannels, int probeHashChannel, int[] probeOutputChannels, int joinType);
    public static native long createLookupOuterFactory(long context, int operatorId, long bridge, int[] probeOutputTypes);
    public static native void destroyBridge(long bridge);
    public static native long createTopNFactory(long context, int operatorId, int[] types, long n, int[] sortChannels, int[] sortOrders);
    public static native long createOrderByFactory(long context, int operatorId, int[] types, int[] outputChannels, int expectedPositions, int[] sortChannels, int[] sortOrders);
    public static native long createMergePagesFactory(long context, int operatorId, int[] types, long minPageSizeInBytes, int minRowCount, long maxPageSizeInBytes);
    public static native long createPartitionedOutputFactory(long context, int operatorId, int[] types, int[] partitionChannels, int hashChannel, int partitionCount,
            boolean replicatesAnyRow, int nullChannel, int partitionFunction);
    public static native long partitionedOutputPoll(long operator, int[] partition);
    public static native void partitionedOutputInfo(long operator, long[] rowsAndPages);
    public static native long createDynamicFilterSourceFactory(long context, int operatorId, int[] types, int[] channels, int maxDistinctValues, long maxFilterSizeInBytes,
            int minMaxCollectionLimit);
    public static native long dynamicFilterSourceResult(long operator, int filterChannel, long[] kindMinMax);
    public static native long createOperator(long factory);
    public static native void noMoreOperators(long factory);
    public static native long duplicateFactory(long factory);
    public static native void destroyFactory(long factory);

    // ---- the scan side (ScanFilterAndProjectOperator over a ConnectorPageSource): upcalls into GpuPageSource ----
    public static native void scanAddPageSource(long operator, GpuPageSource source, int[] types);
    public static native void scanNoMoreSplits(long operator);
    public static native void scanStats(long operator, long[] positionsLoadedSkipped);

    // ---- scan-side decode (tgpu_orc_decode_*): the decompressed streams of one ORC column of one stripe / row group -> a device-resident page ----
    /** LongColumnReader: PRESENT (or null) + DATA; type = GpuPages.T_BIGINT / T_INTEGER / T_DATE, encoding = the ColumnEncoding kind's ordinal */
    public static native long orcDecodeLongColumn(long context, int type, int encoding, int positionCount, byte[] present, byte[] data);
    public static native long orcDecodeBooleanColumn(long context, int positionCount, byte[] present, byte[] data);
    /** SliceDictionaryColumnReader: DATA ids + the dictionary as LENGTH stream and DICTIONARY_DATA bytes */
    public static native long orcDecodeDictionaryStringColumn(long context, int encoding, int positionCount, byte[] present, byte[] data, int dictionarySize, byte[] lengthStream,
            byte[] dictionaryData);

    /** DoubleColumnReader: DATA = the non-null rows' doubles, 8 little-endian bytes each */
    public static native long orcDecodeDoubleColumn(long context, int positionCount, byte[] present, byte[] data);
    /** SliceDirectColumnReader: LENGTH (one length per non-null row) + DATA (their bytes) */
    public static native long orcDecodeDirectStringColumn(long context, int encoding, int positionCount, byte[] present, byte[] data, byte[] lengthStream);

    /**
     * PrimitiveColumnReader.readPageV1 / readPageV2 of a flat Parquet column: the page's definition levels (RLE / bit-packed hybrid, no length prefix;
     * null for a required column), its value section and, for the dictionary encodings, the chunk's PLAIN dictionary page.  physical / encoding =
     * the parquet.thrift ordinals (Type: 0 BOOLEAN, 1 INT32, 2 INT64, 5 DOUBLE, 6 BYTE_ARRAY; Encoding: 0 PLAIN, 2 PLAIN_DICTIONARY, 3 RLE, 8 RLE_DICTIONARY)
     */
    public static native long parquetDecodeDataPage(long context, int type, int physical, int encoding, int positionCount, byte[] definitionLevels, byte[] values,
            byte[] dictionary, int dictionaryCount);

    // ---- exchange between the GPUs of one node (tgpu_exchange_*); pages are output-page handles: they never leave HBM ----
    public static native byte[] exchangeUniqueId();
    public static native long createExchange(long context, byte[] uniqueId, int rank, int world);
    public static native void destroyExchange(long exchange);
    public static native long exchangeRepartition(long exchange, long page, int[] keyChannels, int hashChannel);
    public static native long exchangePartitionedOutput(long exchange, long partitionedOutputOperator, int[] types);
    public static native long exchangeAllGather(long exchange, long page);
    public static native long exchangeBytesSent(long exchange);
}
