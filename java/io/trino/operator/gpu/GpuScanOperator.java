package io.trino.operator.gpu;

import io.trino.metadata.Split;
import io.trino.operator.OperatorContext;
import io.trino.operator.SourceOperator;
import io.trino.spi.Page;
import io.trino.spi.connector.ConnectorPageSource;
import io.trino.spi.connector.UpdatablePageSource;
import io.trino.spi.type.Type;
import io.trino.sql.planner.plan.PlanNodeId;

import java.util.List;
import java.util.Optional;
import java.util.concurrent.ScheduledExecutorService;
import java.util.function.Function;
import java.util.function.Supplier;

/**
 * ScanFilterAndProjectOperator (core/trino-main/src/main/java/io/trino/operator/ScanFilterAndProjectOperator.java:66-447), page-source flavour
 * (processPageSource :275-287): a SourceOperator (SourceOperator.java:24-33) whose splits' ConnectorPageSources are pulled by the library through
 * {@link GpuPageSource}.  `pageSourceForSplit` is the planner's PageSourceProvider.createPageSource bound to this scan's table and columns
 * (ScanFilterAndProjectOperator.java:232-263).
 */
public class GpuScanOperator
        extends GpuOperator
        implements SourceOperator
{
    private final PlanNodeId sourceId;
    private final List<Type> sourceTypes;
    private final Function<Split, ConnectorPageSource> pageSourceForSplit;

    public GpuScanOperator(OperatorContext operatorContext, long handle, PlanNodeId sourceId, List<Type> sourceTypes, Function<Split, ConnectorPageSource> pageSourceForSplit,
            ScheduledExecutorService poller)
    {
        super(operatorContext, handle, sourceTypes, poller);
        this.sourceId = sourceId;
        this.sourceTypes = sourceTypes;
        this.pageSourceForSplit = pageSourceForSplit;
    }

    @Override
    public PlanNodeId getSourceId()
    {
        return sourceId;
    }

    @Override
    public Supplier<Optional<UpdatablePageSource>> addSplit(Split split)
    {
        ConnectorPageSource source = pageSourceForSplit.apply(split);
        try {
            GpuNative.scanAddPageSource(handle, new GpuPageSource(source, sourceTypes), GpuPages.typeCodes(sourceTypes));
        }
        catch (GpuNative.NativeError e) {
            throw GpuNative.toTrinoException(e);
        }
        return () -> source instanceof UpdatablePageSource ? Optional.of((UpdatablePageSource) source) : Optional.empty();
    }

    @Override
    public void noMoreSplits()
    {
        try {
            GpuNative.scanNoMoreSplits(handle);
        }
        catch (GpuNative.NativeError e) {
            throw GpuNative.toTrinoException(e);
        }
    }

    @Override
    public boolean needsInput()
    {
        return false;      // a source operator takes no input (ScanFilterAndProjectOperator / WorkProcessorSourceOperatorAdapter)
    }

    @Override
    public void addInput(Page page)
    {
        throw new UnsupportedOperationException(getClass().getName() + " cannot take input");
    }
}
