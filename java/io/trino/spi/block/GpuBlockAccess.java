package io.trino.spi.block;

import io.airlift.slice.Slice;
import io.trino.spi.type.Type;

import java.lang.reflect.Field;

/**
 * Raw array access to the flat blocks for the GPU glue.  Lives in io.trino.spi.block because the blocks expose their storage only to
 * their own package (LongArrayBlock.java:38-41 private fields; AbstractVariableWidthBlock.java:26 protected getRawSlice); what is private is
 * read through reflection handles looked up ONCE (static finals), not per block.
 */
public final class GpuBlockAccess
{
    private GpuBlockAccess() {}

    private static Field handle(Class<?> owner, String name)
    {
        try {
            Field f = owner.getDeclaredField(name);
            f.setAccessible(true);
            return f;
        }
        catch (ReflectiveOperationException e) {
            throw new ExceptionInInitializerError("block layout changed: " + owner.getSimpleName() + "." + name);
        }
    }

    private static final Field LONG_VALUES = handle(LongArrayBlock.class, "values");
    private static final Field LONG_NULLS = handle(LongArrayBlock.class, "valueIsNull");
    private static final Field LONG_OFFSET = handle(LongArrayBlock.class, "arrayOffset");
    private static final Field INT_VALUES = handle(IntArrayBlock.class, "values");
    private static final Field INT_NULLS = handle(IntArrayBlock.class, "valueIsNull");
    private static final Field INT_OFFSET = handle(IntArrayBlock.class, "arrayOffset");
    private static final Field BYTE_VALUES = handle(ByteArrayBlock.class, "values");
    private static final Field BYTE_NULLS = handle(ByteArrayBlock.class, "valueIsNull");
    private static final Field BYTE_OFFSET = handle(ByteArrayBlock.class, "arrayOffset");
    private static final Field VAR_OFFSETS = handle(VariableWidthBlock.class, "offsets");
    private static final Field VAR_NULLS = handle(VariableWidthBlock.class, "valueIsNull");
    private static final Field VAR_OFFSET = handle(VariableWidthBlock.class, "arrayOffset");
    private static final Field DICT_IDS = handle(DictionaryBlock.class, "ids");
    private static final Field DICT_IDS_OFFSET = handle(DictionaryBlock.class, "idsOffset");

    private static Object get(Field f, Object block)
    {
        try {
            return f.get(block);
        }
        catch (IllegalAccessException e) {
            throw new IllegalStateException(e);
        }
    }

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

    /** LongArrayBlock / IntArrayBlock / ByteArrayBlock / VariableWidthBlock: the four block classes whose arrays travel as they are */
    public static boolean isArrayBacked(Block block)
    {
        return block instanceof LongArrayBlock || block instanceof IntArrayBlock || block instanceof ByteArrayBlock || block instanceof VariableWidthBlock;
    }

    /**
     * The raw arrays of a flat block.  A block of any other class (a builder's block, a dictionary of a dictionary, ...) is first materialised through
     * the type's BlockBuilder (Type.appendTo, S/type/Type.java:165), whose build() is one of the four array-backed classes for the types the GPU
     * operators accept; if it is not, that is an error -- there is no second attempt and no recursion.
     */
    public static Raw raw(Block block, Type type)
    {
        if (!isArrayBacked(block)) {
            BlockBuilder builder = type.createBlockBuilder(null, block.getPositionCount());
            for (int position = 0; position < block.getPositionCount(); position++) {
                type.appendTo(block, position, builder);
            }
            block = builder.build();
            if (!isArrayBacked(block)) {
                throw new IllegalArgumentException("no array-backed form of " + block.getClass().getSimpleName() + " for type " + type);
            }
        }
        Raw r = new Raw();
        r.positionCount = block.getPositionCount();
        if (block instanceof LongArrayBlock) {
            r.typeCode = 1;           // the channel type (BIGINT vs DOUBLE) is the operator's: both are 8-byte values
            r.values = get(LONG_VALUES, block);
            r.nulls = (boolean[]) get(LONG_NULLS, block);
            r.arrayOffset = (int) get(LONG_OFFSET, block);
        }
        else if (block instanceof IntArrayBlock) {
            r.typeCode = 2;
            r.values = get(INT_VALUES, block);
            r.nulls = (boolean[]) get(INT_NULLS, block);
            r.arrayOffset = (int) get(INT_OFFSET, block);
        }
        else if (block instanceof ByteArrayBlock) {
            r.typeCode = 5;
            r.values = get(BYTE_VALUES, block);
            r.nulls = (boolean[]) get(BYTE_NULLS, block);
            r.arrayOffset = (int) get(BYTE_OFFSET, block);
        }
        else {
            VariableWidthBlock v = (VariableWidthBlock) block;
            Slice slice = v.getRawSlice(0);
            r.typeCode = 6;
            r.values = slice.byteArray();
            r.offsets = (int[]) get(VAR_OFFSETS, block);
            r.nulls = (boolean[]) get(VAR_NULLS, block);
            r.arrayOffset = (int) get(VAR_OFFSET, block);
            if (slice.byteArrayOffset() != 0) {    // a slice into a larger array: the offsets are relative to the slice, rebase them once
                int[] rebased = new int[r.positionCount + 1];
                for (int i = 0; i <= r.positionCount; i++) rebased[i] = r.offsets[r.arrayOffset + i] + slice.byteArrayOffset();
                r.offsets = rebased;
                r.nulls = r.compactNulls();
                r.arrayOffset = 0;
            }
        }
        return r;
    }

    public static int[] ids(DictionaryBlock block)
    {
        return (int[]) get(DICT_IDS, block);
    }

    public static int idsOffset(DictionaryBlock block)
    {
        return (int) get(DICT_IDS_OFFSET, block);
    }
}
