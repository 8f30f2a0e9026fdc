package io.trino.operator.gpu;

import io.airlift.slice.Slice;
import io.airlift.slice.Slices;
import io.trino.spi.Page;
import io.trino.spi.block.Block;
import io.trino.spi.block.ByteArrayBlock;
import io.trino.spi.block.DictionaryBlock;
import io.trino.spi.block.GpuBlockAccess;
import io.trino.spi.block.IntArrayBlock;
import io.trino.spi.block.LongArrayBlock;
import io.trino.spi.block.RunLengthEncodedBlock;
import io.trino.spi.block.VariableWidthBlock;
import io.trino.spi.type.Type;

import java.util.List;
import java.util.Optional;

import static io.trino.spi.type.BigintType.BIGINT;
import static io.trino.spi.type.BooleanType.BOOLEAN;
import static io.trino.spi.type.DateType.DATE;
import static io.trino.spi.type.DoubleType.DOUBLE;
import static io.trino.spi.type.IntegerType.INTEGER;

/**
 * io.trino.spi.Page / Block (core/trino-spi/src/main/java/io/trino/spi/Page.java:33-73, block/LongArrayBlock.java:38-75, VariableWidthBlock.java:38-83,
 * DictionaryBlock.java:40-100, RunLengthEncodedBlock.java:30-70) <-> tgpu_block: the blocks' own primitive arrays travel, pinned by the shim for
 * the duration of one call; dictionary and RLE blocks keep their encoding (the library evaluates once per dictionary entry where it can).
 */
public final class GpuPages
{
    // tgpu_type / tgpu_encoding (include/tgpu.h)
    public static final int T_BIGINT = 1, T_INTEGER = 2, T_DATE = 3, T_DOUBLE = 4, T_BOOLEAN = 5, T_VARCHAR = 6;
    public static final int FLAT = 0, DICTIONARY = 1, RLE = 2;

    private GpuPages() {}

    public static int typeCode(Type type)
    {
        if (type.equals(BIGINT)) return T_BIGINT;
        if (type.equals(INTEGER)) return T_INTEGER;
        if (type.equals(DATE)) return T_DATE;
        if (type.equals(DOUBLE)) return T_DOUBLE;
        if (type.equals(BOOLEAN)) return T_BOOLEAN;
        if (type instanceof io.trino.spi.type.VarcharType) return T_VARCHAR;
        throw new IllegalArgumentException("type not supported by the GPU operators: " + type);   // the planner keeps the Java operator for such plan nodes
    }

    public static int[] typeCodes(List<Type> types)
    {
        return types.stream().mapToInt(GpuPages::typeCode).toArray();
    }

    /** a Page whose blocks still live in HBM: handed from one GPU operator to the next without materialising heap blocks */
    public static final class DeviceResidentPage
            extends Page
    {
        private final long handle;       // tgpu_output_page*

        DeviceResidentPage(long handle, Block[] lazyBlocks)
        {
            super(GpuNative.pagePositionCount(handle), lazyBlocks);
            this.handle = handle;
        }

        long handle()
        {
            return handle;
        }
    }

    /** wraps an output-page handle: every channel is a LazyBlock that copies itself to the heap only if a Java operator touches it */
    public static Page deviceResident(long page)
    {
        int channels = GpuNative.pageChannelCount(page);
        Block[][] loaded = new Block[1][];
        Block[] lazy = new Block[channels];
        int positions = GpuNative.pagePositionCount(page);
        for (int ch = 0; ch < channels; ch++) {
            int channel = ch;
            lazy[ch] = new io.trino.spi.block.LazyBlock(positions, () -> {
                if (loaded[0] == null) {
                    loaded[0] = toHeapBlocks(page);      // all channels in one native call, one stream synchronisation
                }
                return loaded[0][channel];
            });
        }
        return new DeviceResidentPage(page, lazy);      // released by the consumer glue (GpuNative.releasePage) once the page is consumed
    }

    /** tgpu_output_page_block_info + tgpu_output_page_copy_blocks -> LongArrayBlock / IntArrayBlock / ByteArrayBlock / VariableWidthBlock */
    public static Block[] toHeapBlocks(long page)
    {
        int channels = GpuNative.pageChannelCount(page);
        int positions = GpuNative.pagePositionCount(page);
        int[] types = new int[channels];
        Object[] values = new Object[channels];
        Object[] nulls = new Object[channels];
        Object[] offsets = new Object[channels];
        long[] info = new long[3];
        for (int ch = 0; ch < channels; ch++) {
            GpuNative.blockInfo(page, ch, info);
            types[ch] = (int) info[0];
            switch (types[ch]) {
                case T_BIGINT: case T_DOUBLE: values[ch] = new long[positions]; break;
                case T_INTEGER: case T_DATE: values[ch] = new int[positions]; break;
                case T_BOOLEAN: values[ch] = new byte[positions]; break;
                default: values[ch] = new byte[(int) info[1]]; offsets[ch] = new int[positions + 1];
            }
            nulls[ch] = info[2] != 0 ? new boolean[positions] : null;
        }
        GpuNative.copyBlocks(page, values, nulls, offsets);
        Block[] blocks = new Block[channels];
        for (int ch = 0; ch < channels; ch++) {
            Optional<boolean[]> isNull = Optional.ofNullable((boolean[]) nulls[ch]);
            switch (types[ch]) {
                case T_BIGINT: case T_DOUBLE: blocks[ch] = new LongArrayBlock(positions, isNull, (long[]) values[ch]); break;
                case T_INTEGER: case T_DATE: blocks[ch] = new IntArrayBlock(positions, isNull, (int[]) values[ch]); break;
                case T_BOOLEAN: blocks[ch] = new ByteArrayBlock(positions, isNull, (byte[]) values[ch]); break;
                default: blocks[ch] = new VariableWidthBlock(positions, Slices.wrappedBuffer((byte[]) values[ch]), (int[]) offsets[ch], isNull);
            }
        }
        return blocks;
    }

    /** Operator.addInput for a heap page: the blocks' raw arrays, encodings kept */
    public static void addInput(long operator, Page page, int[] channelTypes)
    {
        int n = page.getChannelCount();
        int[] types = new int[n], encodings = new int[n], arrayOffsets = new int[n], dictionaryPositions = new int[n];
        Object[] values = new Object[n], nulls = new Object[n], offsets = new Object[n], ids = new Object[n];
        Object[] dValues = new Object[n], dNulls = new Object[n], dOffsets = new Object[n];
        for (int ch = 0; ch < n; ch++) {
            Block block = page.getBlock(ch);
            Block flat = block;
            if (block instanceof DictionaryBlock) {
                encodings[ch] = DICTIONARY;
                ids[ch] = GpuBlockAccess.ids((DictionaryBlock) block);
                arrayOffsets[ch] = GpuBlockAccess.idsOffset((DictionaryBlock) block);
                flat = ((DictionaryBlock) block).getDictionary();
            }
            else if (block instanceof RunLengthEncodedBlock) {
                encodings[ch] = RLE;
                flat = ((RunLengthEncodedBlock) block).getValue();
            }
            GpuBlockAccess.Raw raw = GpuBlockAccess.raw(flat);      // copies a region view into compact arrays when the block is not array-backed
            types[ch] = channelTypes[ch];                           // LongArrayBlock holds BIGINT or DOUBLE, IntArrayBlock INTEGER or DATE: the operator's input types decide
            if (encodings[ch] == FLAT) {
                values[ch] = raw.values;
                nulls[ch] = raw.nulls;
                offsets[ch] = raw.offsets;
                arrayOffsets[ch] = raw.arrayOffset;
            }
            else {
                dValues[ch] = raw.compactValues();
                dNulls[ch] = raw.compactNulls();
                dOffsets[ch] = raw.compactOffsets();
                dictionaryPositions[ch] = flat.getPositionCount();
            }
        }
        GpuNative.addInput(operator, page.getPositionCount(), types, encodings, arrayOffsets, dictionaryPositions, values, nulls, offsets, ids, dValues, dNulls, dOffsets);
    }

    /** ExchangeOperator / spill read-back: the bytes of an uncompressed SerializedPage go to HBM as they came off the wire (PagesSerde.java:117-160) */
    public static Page deserialize(long context, Slice serializedPageBytes, List<Type> types)
    {
        byte[] bytes = serializedPageBytes.byteArray();
        return deviceResident(GpuNative.deserializePage(context, bytes, serializedPageBytes.byteArrayOffset(), serializedPageBytes.length(), typeCodes(types)));
    }
}
