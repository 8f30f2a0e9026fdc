package io.trino.operator.gpu;

import io.trino.metadata.Split;
import io.trino.operator.DriverContext;
import io.trino.operator.OperatorContext;
import io.trino.operator.SourceOperator;
import io.trino.operator.SourceOperatorFactory;
import io.trino.spi.connector.ConnectorPageSource;
import io.trino.spi.type.Type;
import io.trino.sql.planner.plan.PlanNodeId;

import java.util.List;
import java.util.concurrent.ScheduledExecutorService;
import java.util.function.Function;

/**
 * ScanFilterAndProjectOperatorFactory (core/trino-main/src/main/java/io/trino/operator/ScanFilterAndProjectOperator.java:449-560) as a
 * SourceOperatorFactory (SourceOperatorFactory.java:18-32; duplicate() keeps the interface's default: source factories are not duplicated).
 */
public class GpuScanOperatorFactory
        implements SourceOperatorFactory
{
    private final int operatorId;
    private final PlanNodeId planNodeId;
    private final PlanNodeId sourceId;
    private final List<Type> sourceTypes;
    private final Function<Split, ConnectorPageSource> pageSourceForSplit;
    private final ScheduledExecutorService poller;
    private final long factory;
    private boolean closed;

    GpuScanOperatorFactory(int operatorId, PlanNodeId planNodeId, PlanNodeId sourceId, List<Type> sourceTypes, Function<Split, ConnectorPageSource> pageSourceForSplit,
            ScheduledExecutorService poller, long factory)
    {
        this.operatorId = operatorId;
        this.planNodeId = planNodeId;
        this.sourceId = sourceId;
        this.sourceTypes = sourceTypes;
        this.pageSourceForSplit = pageSourceForSplit;
        this.poller = poller;
        this.factory = factory;
    }

    @Override
    public PlanNodeId getSourceId()
    {
        return sourceId;
    }

    @Override
    public SourceOperator createOperator(DriverContext driverContext)
    {
        if (closed) {
            throw new IllegalStateException("Factory is already closed");
        }
        OperatorContext operatorContext = driverContext.addOperatorContext(operatorId, planNodeId, "GpuScanFilterAndProjectOperator");
        try {
            return new GpuScanOperator(operatorContext, GpuNative.createOperator(factory), sourceId, sourceTypes, pageSourceForSplit, poller);
        }
        catch (GpuNative.NativeError e) {
            throw GpuNative.toTrinoException(e);
        }
    }

    @Override
    public void noMoreOperators()
    {
        if (!closed) {
            closed = true;
            GpuNative.noMoreOperators(factory);
        }
    }
}
