package io.trino.operator.gpu;

import io.airlift.units.DataSize;
import io.trino.operator.LookupJoinOperators.JoinType;
import io.trino.operator.OperatorFactory;
import io.trino.spi.type.Type;
import io.trino.sql.planner.plan.AggregationNode.Step;
import io.trino.sql.planner.plan.PlanNodeId;
import io.trino.sql.relational.RowExpression;

import java.util.List;
import java.util.Optional;
import java.util.OptionalInt;
import java.util.concurrent.ScheduledExecutorService;

/**
 * What LocalExecutionPlanner constructs instead of the Java factories when the session enables the GPU operators
 * (core/trino-main/src/main/java/io/trino/sql/planner/LocalExecutionPlanner.java:1235-1384 filter/project, :2965-3056 aggregation,
 * :2104-2311 join build + probe).  Every method returns Optional.empty() when the plan node is outside what the library covers
 * (a type, function or aggregate it does not have): the planner keeps the Java operator for that node -- the fallback is at the
 * planner, never inside the library.
 */
public final class GpuOperatorFactories
{
    private final long context;                         // tgpu_context*: one per worker and GPU
    private final ScheduledExecutorService poller;

    public GpuOperatorFactories(int device, ScheduledExecutorService poller)
    {
        this.context = GpuNative.createContext(device);
        this.poller = poller;
    }

    /** FilterAndProjectOperator.createOperatorFactory (operator/FilterAndProjectOperator.java:73-88) */
    public Optional<OperatorFactory> filterAndProject(int operatorId, PlanNodeId planNodeId, List<Type> inputTypes, Optional<RowExpression> filter, List<RowExpression> projections)
    {
        int[] types;
        Optional<GpuRowExpressions.Program> program;
        try {
            types = GpuPages.typeCodes(inputTypes);
            program = GpuRowExpressions.serialize(filter, projections);
        }
        catch (IllegalArgumentException unsupportedType) {
            return Optional.empty();
        }
        if (program.isEmpty()) {
            return Optional.empty();
        }
        GpuRowExpressions.Program p = program.get();
        long factory = GpuNative.createFilterProjectFactory(context, operatorId, types, p.nodeArray(), p.longArray(), p.doubleArray(), p.stringPool.toByteArray(), p.filterRoot,
                p.projectionRoots);
        return Optional.of(new GpuOperatorFactory(operatorId, planNodeId, "GpuFilterAndProjectOperator", types, poller, factory));
    }

    /**
     * HashAggregationOperatorFactory (operator/HashAggregationOperator.java:54-262).  aggregates: {tgpu_agg_function, input channel, mask channel} triples
     * resolved by the caller from the AccumulatorFactories' bound signatures (count / sum / avg over BIGINT and DOUBLE; anything else -> Optional.empty()).
     */
    public Optional<OperatorFactory> hashAggregation(int operatorId, PlanNodeId planNodeId, List<Type> inputTypes, List<Type> groupByTypes, List<Integer> groupByChannels, Step step,
            Optional<int[]> aggregates, Optional<Integer> hashChannel, int expectedGroups, boolean produceDefaultOutput, Optional<DataSize> maxPartialMemory, boolean spillEnabled)
    {
        if (aggregates.isEmpty()) {
            return Optional.empty();
        }
        int[] types;
        int[] keyTypes;
        try {
            types = GpuPages.typeCodes(inputTypes);
            keyTypes = GpuPages.typeCodes(groupByTypes);
        }
        catch (IllegalArgumentException unsupportedType) {
            return Optional.empty();
        }
        int stepCode = step == Step.SINGLE ? 0 : step == Step.PARTIAL ? 1 : step == Step.FINAL ? 2 : -1;
        if (stepCode < 0) {
            return Optional.empty();    // INTERMEDIATE
        }
        long factory = GpuNative.createHashAggregationFactory(context, operatorId, keyTypes, groupByChannels.stream().mapToInt(Integer::intValue).toArray(), hashChannel.orElse(-1),
                stepCode, aggregates.get(), expectedGroups, produceDefaultOutput);
        // HashAggregationOperatorFactory(..., maxPartialMemory, spillEnabled, ...) (operator/HashAggregationOperator.java:128-154)
        maxPartialMemory.ifPresent(limit -> GpuNative.setMaxPartialMemory(factory, limit.toBytes()));
        if (spillEnabled) {
            GpuNative.setSpillEnabled(factory, true);
        }
        return Optional.of(new GpuOperatorFactory(operatorId, planNodeId, "GpuHashAggregationOperator", types, poller, factory));
    }

    /** the join bridge handle (tgpu_lookup_source_factory*) plays the JoinBridgeManager's role (operator/PartitionedLookupSourceFactory.java:146-205) */
    public static final class JoinBuild
    {
        public final OperatorFactory buildFactory;
        public final long bridge;

        JoinBuild(OperatorFactory buildFactory, long bridge)
        {
            this.buildFactory = buildFactory;
            this.bridge = bridge;
        }
    }

    /** HashBuilderOperatorFactory (operator/HashBuilderOperator.java:54-152) */
    public Optional<JoinBuild> hashBuilder(int operatorId, PlanNodeId planNodeId, List<Type> types, List<Integer> outputChannels, List<Integer> hashChannels,
            OptionalInt preComputedHashChannel, int expectedPositions)
    {
        int[] codes;
        try {
            codes = GpuPages.typeCodes(types);
        }
        catch (IllegalArgumentException unsupportedType) {
            return Optional.empty();
        }
        long[] handles = GpuNative.createHashBuilderFactory(context, operatorId, codes, outputChannels.stream().mapToInt(Integer::intValue).toArray(),
                hashChannels.stream().mapToInt(Integer::intValue).toArray(), preComputedHashChannel.orElse(-1), expectedPositions);
        return Optional.of(new JoinBuild(new GpuOperatorFactory(operatorId, planNodeId, "GpuHashBuilderOperator", codes, poller, handles[0]), handles[1]));
    }

    /** LookupJoinOperators.innerJoin / probeOuterJoin / lookupOuterJoin / fullOuterJoin (operator/LookupJoinOperators.java:30-63) */
    public Optional<OperatorFactory> lookupJoin(int operatorId, PlanNodeId planNodeId, JoinBuild build, List<Type> probeTypes, List<Integer> probeJoinChannels,
            OptionalInt probeHashChannel, List<Integer> probeOutputChannels, JoinType joinType)
    {
        int[] codes;
        try {
            codes = GpuPages.typeCodes(probeTypes);
        }
        catch (IllegalArgumentException unsupportedType) {
            return Optional.empty();
        }
        long factory = GpuNative.createLookupJoinFactory(context, operatorId, build.bridge, codes, probeJoinChannels.stream().mapToInt(Integer::intValue).toArray(),
                probeHashChannel.orElse(-1), probeOutputChannels.stream().mapToInt(Integer::intValue).toArray(), joinType.ordinal());
        return Optional.of(new GpuOperatorFactory(operatorId, planNodeId, "GpuLookupJoinOperator", codes, poller, factory));
    }

    /** LookupJoinOperatorFactory.createOuterOperatorFactory (operator/LookupJoinOperatorFactory.java:88-103,141-146) */
    public OperatorFactory lookupOuter(int operatorId, PlanNodeId planNodeId, JoinBuild build, List<Type> probeOutputTypes)
    {
        int[] codes = GpuPages.typeCodes(probeOutputTypes);
        return new GpuOperatorFactory(operatorId, planNodeId, "GpuLookupOuterOperator", new int[0], poller, GpuNative.createLookupOuterFactory(context, operatorId, build.bridge, codes));
    }
}
