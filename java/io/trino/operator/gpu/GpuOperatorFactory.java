package io.trino.operator.gpu;

import io.trino.operator.DriverContext;
import io.trino.operator.Operator;
import io.trino.operator.OperatorContext;
import io.trino.operator.OperatorFactory;
import io.trino.spi.type.Type;
import io.trino.sql.planner.plan.PlanNodeId;

import java.util.List;
import java.util.concurrent.ScheduledExecutorService;

/**
 * io.trino.operator.OperatorFactory (core/trino-main/src/main/java/io/trino/operator/OperatorFactory.java:18-50) over a tgpu_operator_factory
 * handle; constructed by {@link GpuOperatorFactories} at the LocalExecutionPlanner call sites listed in INTEGRATION.md.
 */
public class GpuOperatorFactory
        implements OperatorFactory
{
    protected final int operatorId;
    protected final PlanNodeId planNodeId;
    protected final String operatorType;
    protected final List<Type> inputTypes;
    protected final ScheduledExecutorService poller;
    protected long factory;                                // tgpu_operator_factory*
    protected boolean closed;

    GpuOperatorFactory(int operatorId, PlanNodeId planNodeId, String operatorType, List<Type> inputTypes, ScheduledExecutorService poller, long factory)
    {
        this.operatorId = operatorId;
        this.planNodeId = planNodeId;
        this.operatorType = operatorType;
        this.inputTypes = inputTypes;
        this.poller = poller;
        this.factory = factory;
    }

    @Override
    public Operator createOperator(DriverContext driverContext)
    {
        if (closed) {
            throw new IllegalStateException("Factory is already closed");
        }
        OperatorContext operatorContext = driverContext.addOperatorContext(operatorId, planNodeId, operatorType);
        try {
            return new GpuOperator(operatorContext, GpuNative.createOperator(factory), inputTypes, poller);
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
            GpuNative.noMoreOperators(factory);       // probe factories: lets the build side release its table once the probes are done
        }
    }

    @Override
    public OperatorFactory duplicate()
    {
        try {
            return new GpuOperatorFactory(operatorId, planNodeId, operatorType, inputTypes, poller, GpuNative.duplicateFactory(factory));
        }
        catch (GpuNative.NativeError e) {
            throw GpuNative.toTrinoException(e);      // NOT_SUPPORTED for a hash builder, like HashBuilderOperator.java:150-152
        }
    }
}
