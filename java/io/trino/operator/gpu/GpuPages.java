package io.trino.operator.gpu;

import io.airlift.slice.Slice;
import io.airlift.slice.Slices;
import io.trino.spi.Page;
import io.trino.spi.block.Block;
import io.trino.spi.block.ByteArrayBlock;
import io.trino.spi.block.DictionaryBlock;
import io.trino.spi.block.GpuBlockAccess;
import io.trino.spi.block.IntArrayBlock;
import io.trino.spi.block.LazyBlock;
import io.trino.spi.block.LongArrayBlock;
import io.trino.spi.block.RunLengthEncodedBlock;
import io.trino.spi.block.VariableWidthBlock;
import io.trino.spi.type.Type;

import java.lang.ref.Cleaner;
import java.util.List;
import java.util.Optional;
import java.util.concurrent.atomic.AtomicBoolean;

import static io.trino.spi.type.BigintType.BIGINT;
import static io.trino.spi.type.BooleanType.BOOLEAN;
import static io.trino.spi.type.DateType.DATE;
import static io.trino.spi.type.DoubleType.DOUBLE;
import static io.trino.spi.type.IntegerType.INTEGER;

/**
 * io.trino.spi.Page / Block (core/trino-spi/src/main/java/io/trino/spi/Page.java:33-73, block/LongArrayBlock.java:38-75, VariableWidthBlock.java:38-83,
 * DictionaryBlock.java:40-100, RunLengthEncodedBlock.java:30-70) <-> tgpu_block: the blocks' own primitive arrays travel, pinned by the shim for
 * the duration of one call; dictionary and RLE blocks keep their encoding (the library evaluates once per dictionary entry where it can).
 *
 * A page that is still in HBM travels between GPU operators as an ordinary {@link Page} -- Page is final (Page.java:33), nothing subclasses it --
 * whose blocks are {@link DeviceBlock}s: LazyBlock subclasses (LazyBlock.java:34 is not final) that carry the output-page handle.  A Java operator that
 * touches such a block loads it to the heap like any lazy block; a GPU operator recognises its own kind ({@link #deviceHandle}) and hands the handle on.
 */
public final class GpuPages
{
    // tgpu_type / tgpu_encoding (include/tgpu.h)
    public static final int T_BIGINT = 1, T_INTEGER = 2, T_DATE = 3, T_DOUBLE = 4, T_BOOLEAN = 5, T_VARCHAR = 6;
    public static final int FLAT = 0, DICTIONARY = 1, RLE = 2;

    private static final Cleaner CLEANER = Cleaner.create();

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

    /** the native output page behind the blocks of one device-resident Page: released once, by whoever gets there first -- the GPU consumer, or the GC */
    static final class DevicePageHandle
    {
        final long handle;       // tgpu_output_page*
        final int positionCount;
        final int channelCount;
        private final AtomicBoolean released = new AtomicBoolean();
        private Block[] heapBlocks;      // all channels, loaded by ONE native call the first time a Java operator touches any of them

        DevicePageHandle(long handle)
        {
            this.handle = handle;
            this.positionCount = GpuNative.pagePositionCount(handle);
            this.channelCount = GpuNative.pageChannelCount(handle);
            AtomicBoolean flag = released;
            long nativePage = handle;                    // (the cleaning action must not capture `this`)
            CLEANER.register(this, () -> {
                if (flag.compareAndSet(false, true)) {
                    GpuNative.releasePage(nativePage);
                }
            });
        }

        synchronized Block heapBlock(int channel)
        {
            if (heapBlocks == null) {
                if (released.get()) {
                    throw new IllegalStateException("device page was consumed by a GPU operator before its blocks were read");
                }
                heapBlocks = toHeapBlocks(handle);
            }
            return heapBlocks[channel];
        }

        /** the consumer is done with the device copy (a GPU operator has taken it over, or the heap copy exists and nothing else needs HBM) */
        void release()
        {
            if (released.compareAndSet(false, true)) {
                GpuNative.releasePage(handle);
            }
        }

        boolean isReleased()
        {
            return released.get();
        }
    }

    /** channel `channel` of a page that still lives in HBM; loads itself to the heap only if a Java operator reads it */
    public static final class DeviceBlock
            extends LazyBlock
    {
        final DevicePageHandle page;
        final int channel;

        DeviceBlock(DevicePageHandle page, int channel)
        {
            super(page.positionCount, () -> page.heapBlock(channel));
            this.page = page;
            this.channel = channel;
        }
    }

    /** wraps an output-page handle as a Page of {@link DeviceBlock}s */
    public static Page deviceResident(long handle)
    {
        DevicePageHandle page = new DevicePageHandle(handle);
        Block[] blocks = new Block[page.channelCount];
        for (int channel = 0; channel < blocks.length; channel++) {
            blocks[channel] = new DeviceBlock(page, channel);
        }
        return new Page(page.positionCount, blocks);
    }

    /**
     * The output-page handle of `page` if it is, unchanged, what {@link #deviceResident} built -- every channel the DeviceBlock of that channel of ONE
     * device page, none of them loaded or consumed yet -- else null: a page some Java operator has re-arranged (getColumns, getRegion, appended channels)
     * takes the heap path, its blocks load themselves.
     */
    static DevicePageHandle deviceHandle(Page page)
    {
        int channels = page.getChannelCount();
        if (channels == 0 || !(page.getBlock(0) instanceof DeviceBlock)) {
            return null;
        }
        DevicePageHandle handle = ((DeviceBlock) page.getBlock(0)).page;
        if (handle.isReleased() || handle.channelCount != channels || handle.positionCount != page.getPositionCount()) {
            return null;
        }
        for (int channel = 0; channel < channels; channel++) {
            Block block = page.getBlock(channel);
            if (!(block instanceof DeviceBlock) || ((DeviceBlock) block).page != handle || ((DeviceBlock) block).channel != channel) {
                return null;
            }
        }
        return handle;
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

    /** the arrays of one block as the shim takes them (GpuNative.addInput's parallel arrays, GpuPageSource.loadBlock's Object[8]) */
    static final class BlockArrays
    {
        int encoding;
        int arrayOffset;
        int dictionaryPositions;
        Object values;
        Object nulls;
        Object offsets;
        Object ids;
        Object dictionaryValues;
        Object dictionaryNulls;
        Object dictionaryOffsets;
    }

    /**
     * One loaded block -> its raw arrays, encodings kept one level deep: a DictionaryBlock over a flat dictionary and an RLE block over a flat value stay
     * encoded (the library evaluates once per entry); anything nested deeper (a dictionary of a dictionary, S/block/DictionaryBlock.java:351-357
     * copyRegion returns a DictionaryBlock again) is materialised through the type's BlockBuilder into a flat array block -- no recursion on block kinds.
     */
    static BlockArrays arraysOf(Block block, Type type)
    {
        BlockArrays a = new BlockArrays();
        Block loaded = block.getLoadedBlock();
        Block flat = loaded;
        if (loaded instanceof DictionaryBlock && GpuBlockAccess.isArrayBacked(((DictionaryBlock) loaded).getDictionary().getLoadedBlock())) {
            a.encoding = DICTIONARY;
            a.ids = GpuBlockAccess.ids((DictionaryBlock) loaded);
            a.arrayOffset = GpuBlockAccess.idsOffset((DictionaryBlock) loaded);
            flat = ((DictionaryBlock) loaded).getDictionary().getLoadedBlock();
        }
        else if (loaded instanceof RunLengthEncodedBlock && GpuBlockAccess.isArrayBacked(((RunLengthEncodedBlock) loaded).getValue().getLoadedBlock())) {
            a.encoding = RLE;
            flat = ((RunLengthEncodedBlock) loaded).getValue().getLoadedBlock();
        }
        GpuBlockAccess.Raw raw = GpuBlockAccess.raw(flat, type);
        if (a.encoding == FLAT) {
            a.values = raw.values;
            a.nulls = raw.nulls;
            a.offsets = raw.offsets;
            a.arrayOffset = raw.arrayOffset;
        }
        else {
            a.dictionaryValues = raw.compactValues();
            a.dictionaryNulls = raw.compactNulls();
            a.dictionaryOffsets = raw.compactOffsets();
            a.dictionaryPositions = flat.getPositionCount();
        }
        return a;
    }

    /** Operator.addInput for a heap page: the blocks' raw arrays, encodings kept */
    public static void addInput(long operator, Page page, List<Type> channelTypes)
    {
        int n = page.getChannelCount();
        int[] types = new int[n], encodings = new int[n], arrayOffsets = new int[n], dictionaryPositions = new int[n];
        Object[] values = new Object[n], nulls = new Object[n], offsets = new Object[n], ids = new Object[n];
        Object[] dValues = new Object[n], dNulls = new Object[n], dOffsets = new Object[n];
        for (int ch = 0; ch < n; ch++) {
            BlockArrays a = arraysOf(page.getBlock(ch), channelTypes.get(ch));
            types[ch] = typeCode(channelTypes.get(ch));     // LongArrayBlock holds BIGINT or DOUBLE, IntArrayBlock INTEGER or DATE: the operator's input types decide
            encodings[ch] = a.encoding;
            arrayOffsets[ch] = a.arrayOffset;
            dictionaryPositions[ch] = a.dictionaryPositions;
            values[ch] = a.values;
            nulls[ch] = a.nulls;
            offsets[ch] = a.offsets;
            ids[ch] = a.ids;
            dValues[ch] = a.dictionaryValues;
            dNulls[ch] = a.dictionaryNulls;
            dOffsets[ch] = a.dictionaryOffsets;
        }
        GpuNative.addInput(operator, page.getPositionCount(), types, encodings, arrayOffsets, dictionaryPositions, values, nulls, offsets, ids, dValues, dNulls, dOffsets);
    }

    /** ExchangeOperator / spill read-back: the bytes of a SerializedPage go to HBM as they came off the wire (PagesSerde.java:117-160) */
    public static Page deserialize(long context, Slice serializedPageBytes, List<Type> types)
    {
        byte[] bytes = serializedPageBytes.byteArray();
        return deviceResident(GpuNative.deserializePage(context, bytes, serializedPageBytes.byteArrayOffset(), serializedPageBytes.length(), typeCodes(types)));
    }

    /** PagesSerde.serialize of a page that is still in HBM (the producer side of an exchange towards Java workers) */
    public static Slice serialize(long context, Page page)
    {
        DevicePageHandle handle = deviceHandle(page);
        if (handle == null) {
            throw new IllegalArgumentException("not a device-resident page");
        }
        byte[] out = new byte[(int) GpuNative.serializePage(context, handle.handle, null)];
        int written = (int) GpuNative.serializePage(context, handle.handle, out);
        return Slices.wrappedBuffer(out, 0, written);
    }
}
