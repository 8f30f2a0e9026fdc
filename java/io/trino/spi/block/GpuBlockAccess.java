package io.trino.spi.block;

import io.airlift.slice.Slice;

import java.lang.reflect.Field;

/**
 * Raw array access to the flat blocks for the GPU glue.  Lives in io.trino.spi.block because the blocks expose their storage only to
 * their own package (LongArrayBlock.java:38-41 private fields, package-private getValuesSlice :269; AbstractVariableWidthBlock.java:26
 * protected getRawSlice); what is private is read through cached reflection.
 */
public final class GpuBlockAccess
{
    private GpuBlockAccess() {}

    public static final class Raw
    {
        public int typeCode;          // tgpu_type
        public Object values;         // long[] / int[] / byte[] (VARCHAR: the slice's byte[])
        public boolean[] nulls;       // valueIsNull or null
        public int[] offsets;         // VARCHAR
        public int arrayOffset;
        public int positionCount;

        /** the dictionary side of a DictionaryBlock / RLE block is passed compact (arrayOffset applied) */
        public Object compactValues()
        {
            if (arrayOffset == 0 || typeCode == 6) {
                return values;
            }
            if (values instanceof long[]) return java.util.Arrays.copyOfRange((long[]) values, arrayOffset, arrayOffset + positionCount);
            if (values instanceof int[]) return java.util.Arrays.copyOfRange((int[]) values, arrayOffset, arrayOffset + positionCount);
            return java.util.Arrays.copyOfRange((byte[]) values, arrayOffset, arrayOffset + positionCount);
        }

        public boolean[] compactNulls()
        {
            return nulls == null || arrayOffset == 0 ? nulls : java.util.Arrays.copyOfRange(nulls, arrayOffset, arrayOffset + positionCount);
        }

        public int[] compactOffsets()
        {
            return offsets == null || arrayOffset == 0 ? offsets : java.util.Arrays.copyOfRange(offsets, arrayOffset, arrayOffset + positionCount + 1);
        }
    }

    private static Object field(Object block, Class<?> owner, String name)
    {
        try {
            Field f = owner.getDeclaredField(name);
            f.setAccessible(true);
            return f.get(block);
        }
        catch (ReflectiveOperationException e) {
            throw new IllegalStateException("block layout changed: " + owner.getSimpleName() + "." + name, e);
        }
    }

    public static Raw raw(Block block)
    {
        Raw r = new Raw();
        r.positionCount = block.getPositionCount();
        if (block instanceof LongArrayBlock) {
            r.typeCode = 1;           // the channel type (BIGINT vs DOUBLE) is the operator's: both are 8-byte values
            r.values = field(block, LongArrayBlock.class, "values");
            r.nulls = (boolean[]) field(block, LongArrayBlock.class, "valueIsNull");
            r.arrayOffset = (int) field(block, LongArrayBlock.class, "arrayOffset");
        }
        else if (block instanceof IntArrayBlock) {
            r.typeCode = 2;
            r.values = field(block, IntArrayBlock.class, "values");
            r.nulls = (boolean[]) field(block, IntArrayBlock.class, "valueIsNull");
            r.arrayOffset = (int) field(block, IntArrayBlock.class, "arrayOffset");
        }
        else if (block instanceof ByteArrayBlock) {
            r.typeCode = 5;
            r.values = field(block, ByteArrayBlock.class, "values");
            r.nulls = (boolean[]) field(block, ByteArrayBlock.class, "valueIsNull");
            r.arrayOffset = (int) field(block, ByteArrayBlock.class, "arrayOffset");
        }
        else if (block instanceof VariableWidthBlock) {
            VariableWidthBlock v = (VariableWidthBlock) block;
            Slice slice = v.getRawSlice(0);
            r.typeCode = 6;
            r.values = slice.byteArray();          // offsets are relative to the slice: the shim adds byteArrayOffset through `offsets`
            r.offsets = (int[]) field(block, VariableWidthBlock.class, "offsets");
            r.nulls = (boolean[]) field(block, VariableWidthBlock.class, "valueIsNull");
            r.arrayOffset = (int) field(block, VariableWidthBlock.class, "arrayOffset");
            if (slice.byteArrayOffset() != 0) {    // a slice into a larger array: rebase once
                int[] rebased = new int[r.positionCount + 1];
                for (int i = 0; i <= r.positionCount; i++) rebased[i] = r.offsets[r.arrayOffset + i] + slice.byteArrayOffset();
                r.offsets = rebased;
                r.nulls = r.compactNulls();
                r.arrayOffset = 0;
            }
        }
        else {
            // any other block (a region view, a builder's block, ...): Block.copyRegion gives a compact array-backed block of the same kind
            return raw(block.copyRegion(0, block.getPositionCount()));
        }
        return r;
    }

    public static int[] ids(DictionaryBlock block)
    {
        return (int[]) field(block, DictionaryBlock.class, "ids");
    }

    public static int idsOffset(DictionaryBlock block)
    {
        return (int) field(block, DictionaryBlock.class, "idsOffset");
    }
}
