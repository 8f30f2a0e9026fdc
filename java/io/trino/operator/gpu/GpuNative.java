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

    // context: one per worker and GPU
    public static native long createContext(int device);
    public static native void destroyContext(long context);

    // Operator protocol (io.trino.operator.Operator)
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
    public static native long getOutput(long operator, boolean[] wouldBlock);

    // output pages
    public static native int pagePositionCount(long page);
    public static native int pageChannelCount(long page);
    public static native void blockInfo(long page, int channel, long[] typeBytesNulls);
    public static native void copyBlocks(long page, Object[] values, Object[] nulls, Object[] offsets);
    public static native void releasePage(long page);
    public static native long deserializePage(long context, byte[] bytes, int offset, int length, int[] types);

    // factories (io.trino.operator.OperatorFactory)
    public static native long createFilterProjectFactory(long context, int operatorId, int[] inputTypes, int[][] nodes, long[] longValues, double[] doubleValues,
            byte[] stringPool, int filterRoot, int[] projectionRoots);
    public static native long createHashAggregationFactory(long context, int operatorId, int[] groupByTypes, int[] groupByChannels, int hashChannel, int step,
            int[] aggregates, int expectedGroups, boolean produceDefaultOutput);
    public static native long[] createHashBuilderFactory(long context, int operatorId, int[] types, int[] outputChannels, int[] hashChannels, int precomputedHashChannel,
            int expectedPositions);
    public static native long createLookupJoinFactory(long context, int operatorId, long bridge, int[] probeTypes, int[] probeJoinChannels, int probeHashChannel,
            int[] probeOutputChannels, int joinType);
    public static native long createLookupOuterFactory(long context, int operatorId, long bridge, int[] probeOutputTypes);
    public static native void destroyBridge(long bridge);
    public static native long createOperator(long factory);
    public static native void noMoreOperators(long factory);
    public static native long duplicateFactory(long factory);
    public static native void destroyFactory(long factory);
}
