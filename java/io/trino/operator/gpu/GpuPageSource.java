package io.trino.operator.gpu;

import io.trino.spi.Page;
import io.trino.spi.connector.ConnectorPageSource;
import io.trino.spi.type.Type;

import java.io.IOException;
import java.io.UncheckedIOException;
import java.util.List;

/**
 * The split's ConnectorPageSource (core/trino-spi/src/main/java/io/trino/spi/connector/ConnectorPageSource.java) as the library's tgpu_page_source
 * callbacks see it: the JNI shim calls these five methods from inside tgpu_operator_get_output / _is_blocked / _is_finished on the driver thread
 * (jni/tgpu_jni.c "the scan side").  Every channel is announced lazy; the library asks for the channels its page processor reads -- the filter's first,
 * the projections' only when a row survived (PageProcessor.java:111-137) -- through {@link #loadBlock}, which is where a LazyBlock of the connector
 * actually loads.
 */
public final class GpuPageSource
{
    private final ConnectorPageSource source;
    private final List<Type> types;
    private Page current;

    public GpuPageSource(ConnectorPageSource source, List<Type> types)
    {
        this.source = source;
        this.types = types;
    }

    /** ConnectorPageSource.getNextPage: the page's position count, -1 when there is no page right now */
    public int nextPage()
    {
        current = source.getNextPage();
        return current == null ? -1 : current.getPositionCount();
    }

    public boolean isFinished()
    {
        return source.isFinished();
    }

    public boolean isBlocked()
    {
        return !source.isBlocked().isDone();
    }

    /** {int[3]{encoding, arrayOffset, dictionaryPositions}, values, nulls, offsets, ids, dictionaryValues, dictionaryNulls, dictionaryOffsets} of channel `channel` */
    public Object[] loadBlock(int channel)
    {
        GpuPages.BlockArrays a = GpuPages.arraysOf(current.getBlock(channel), types.get(channel));
        return new Object[] {new int[] {a.encoding, a.arrayOffset, a.dictionaryPositions}, a.values, a.nulls, a.offsets, a.ids, a.dictionaryValues, a.dictionaryNulls,
                a.dictionaryOffsets};
    }

    public void close()
    {
        try {
            source.close();
        }
        catch (IOException e) {
            throw new UncheckedIOException(e);
        }
    }

    ConnectorPageSource connectorPageSource()
    {
        return source;
    }
}
