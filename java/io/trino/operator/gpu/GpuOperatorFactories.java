package io.trino.operator.gpu;

import io.airlift.units.DataSize;
import io.trino.metadata.Split;
import io.trino.operator.LookupJoinOperators.JoinType;
import io.trino.operator.OperatorFactory;
import io.trino.operator.SourceOperatorFactory;
import io.trino.spi.connector.ConnectorPageSource;
import io.trino.spi.connector.SortOrder;
import io.trino.spi.type.Type;
import io.trino.sql.planner.plan.AggregationNode.Step;
import io.trino.sql.planner.plan.PlanNodeId;
import io.trino.sql.relational.RowExpression;

import java.util.List;
import java.util.Optional;
import java.util.OptionalInt;
import java.util.concurrent.ScheduledExecutorService;
import java.util.function.Function;

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

    public long context()
    {
        return context;
    }

    /** operators in front of Java operators: output cut at PageBuilder.isFull granularity (S/block/PageBuilderStatus.java:49-60); 0 = one page per call */
    public void setMaxOutputPage(long maxBytes, long maxRows)
    {
        GpuNative.setMaxOutputPage(context, maxBytes, maxRows);
    }

    /** strict parity runs: sum(double) / avg(double) in the Java row order (DESIGN.md "DOUBLE aggregate policy") */
    public void setJavaDoubleSumOrder(boolean javaOrder)
    {
        GpuNative.setDoubleSumOrder(context, javaOrder ? 1 : 0);
    }

    /**
     * The promise that device blocks this embedding passes to addInput stay untouched until the operator that took them has finished
     * (a device-resident connector whose stripes live as long as the split): lets the fused aggregation / join keep such pages by reference.
     * Pages of the library itself (DevicePageHandle) never need it.
     */
    public void setDeviceInputStable(boolean stable)
    {
        GpuNative.setDeviceInputStable(context, stable);
    }

    private static int[] ints(List<Integer> values)
    {
        return values.stream().mapToInt(Integer::intValue).toArray();
    }

    private static int stepCode(Step step)
    {
        return step == Step.SINGLE ? 0 : step == Step.PARTIAL ? 1 : step == Step.FINAL ? 2 : -1;     // INTERMEDIATE: not built
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
        return Optional.of(new GpuOperatorFactory(operatorId, planNodeId, "GpuFilterAndProjectOperator", inputTypes, poller, factory));
    }

    /**
     * ScanFilterAndProjectOperatorFactory (operator/ScanFilterAndProjectOperator.java:449-560; LocalExecutionPlanner.java:1343-1384): the page-source flavour;
     * `pageSourceForSplit` = pageSourceProvider.createPageSource(session, split, table, columns, dynamicFilter) bound by the planner
     */
    public Optional<SourceOperatorFactory> scanFilterAndProject(int operatorId, PlanNodeId planNodeId, PlanNodeId sourceId, List<Type> sourceTypes,
            Function<Split, ConnectorPageSource> pageSourceForSplit, Optional<RowExpression> filter, List<RowExpression> projections)
    {
        int[] types;
        Optional<GpuRowExpressions.Program> program;
        try {
            types = GpuPages.typeCodes(sourceTypes);
            program = GpuRowExpressions.serialize(filter, projections);
        }
        catch (IllegalArgumentException unsupportedType) {
            return Optional.empty();
        }
        if (program.isEmpty()) {
            return Optional.empty();
        }
        GpuRowExpressions.Program p = program.get();
        long factory = GpuNative.createScanFilterProjectFactory(context, operatorId, types, p.nodeArray(), p.longArray(), p.doubleArray(), p.stringPool.toByteArray(), p.filterRoot,
                p.projectionRoots);
        return Optional.of(new GpuScanOperatorFactory(operatorId, planNodeId, sourceId, sourceTypes, pageSourceForSplit, poller, factory));
    }

    /**
     * HashAggregationOperatorFactory (operator/HashAggregationOperator.java:54-262).  aggregates: {tgpu_agg_function, input channel, mask channel} triples
     * resolved by the caller from the AccumulatorFactories' bound signatures (count / sum / avg over BIGINT and DOUBLE, min / max over BIGINT and DOUBLE; anything else -> Optional.empty()).
     */
    public Optional<OperatorFactory> hashAggregation(int operatorId, PlanNodeId planNodeId, List<Type> inputTypes, List<Type> groupByTypes, List<Integer> groupByChannels, Step step,
            Optional<int[]> aggregates, Optional<Integer> hashChannel, int expectedGroups, boolean produceDefaultOutput, Optional<DataSize> maxPartialMemory, boolean spillEnabled)
    {
        if (aggregates.isEmpty() || stepCode(step) < 0) {
            return Optional.empty();
        }
        int[] keyTypes;
        try {
            GpuPages.typeCodes(inputTypes);
            keyTypes = GpuPages.typeCodes(groupByTypes);
        }
        catch (IllegalArgumentException unsupportedType) {
            return Optional.empty();
        }
        long factory = GpuNative.createHashAggregationFactory(context, operatorId, keyTypes, ints(groupByChannels), hashChannel.orElse(-1), stepCode(step), aggregates.get(),
                expectedGroups, produceDefaultOutput);
        // HashAggregationOperatorFactory(..., maxPartialMemory, spillEnabled, ...) (operator/HashAggregationOperator.java:128-154)
        maxPartialMemory.ifPresent(limit -> GpuNative.setMaxPartialMemory(factory, limit.toBytes()));
        if (spillEnabled) {
            GpuNative.setSpillEnabled(factory, true);
        }
        return Optional.of(new GpuOperatorFactory(operatorId, planNodeId, "GpuHashAggregationOperator", inputTypes, poller, factory));
    }

    /**
     * A FilterNode / ProjectNode directly under an AggregationNode (LocalExecutionPlanner.visitAggregation over a filter/project source, :1198,2965-3056) as ONE
     * fused pipeline: tgpu_filter_project_hash_aggregation_factory_create, the operator bench.py's Q1 line times.  groupByChannels, hashChannel and the
     * aggregates' channels index the PROJECTIONS.  Same results as filterAndProject(...) feeding hashAggregation(...); SINGLE and PARTIAL steps.
     */
    public Optional<OperatorFactory> filterProjectHashAggregation(int operatorId, PlanNodeId planNodeId, List<Type> inputTypes, Optional<RowExpression> filter,
            List<RowExpression> projections, List<Type> groupByTypes, List<Integer> groupByChannels, Step step, Optional<int[]> aggregates, Optional<Integer> hashChannel,
            int expectedGroups, Optional<DataSize> maxPartialMemory, boolean spillEnabled)
    {
        if (aggregates.isEmpty() || (step != Step.SINGLE && step != Step.PARTIAL)) {
            return Optional.empty();
        }
        int[] types;
        int[] keyTypes;
        Optional<GpuRowExpressions.Program> program;
        try {
            types = GpuPages.typeCodes(inputTypes);
            keyTypes = GpuPages.typeCodes(groupByTypes);
            program = GpuRowExpressions.serialize(filter, projections);
        }
        catch (IllegalArgumentException unsupportedType) {
            return Optional.empty();
        }
        if (program.isEmpty()) {
            return Optional.empty();
        }
        GpuRowExpressions.Program p = program.get();
        long factory = GpuNative.createFilterProjectHashAggregationFactory(context, operatorId, types, p.nodeArray(), p.longArray(), p.doubleArray(), p.stringPool.toByteArray(),
                p.filterRoot, p.projectionRoots, keyTypes, ints(groupByChannels), hashChannel.orElse(-1), stepCode(step), aggregates.get(), expectedGroups);
        maxPartialMemory.ifPresent(limit -> GpuNative.setMaxPartialMemory(factory, limit.toBytes()));
        if (spillEnabled) {
            GpuNative.setSpillEnabled(factory, true);
        }
        return Optional.of(new GpuOperatorFactory(operatorId, planNodeId, "GpuFilterProjectHashAggregationOperator", inputTypes, poller, factory));
    }

    /** the join bridge handle (tgpu_lookup_source_factory*) plays the JoinBridgeManager's role (operator/PartitionedLookupSourceFactory.java:146-205) */
    public static final class JoinBuild
    {
        public final OperatorFactory buildFactory;
        public final long bridge;
        public final List<Type> buildTypes;

        JoinBuild(OperatorFactory buildFactory, long bridge, List<Type> buildTypes)
        {
            this.buildFactory = buildFactory;
            this.bridge = bridge;
            this.buildTypes = buildTypes;
        }

        /** {positions, table slots, position links} of the built table */
        public long[] stats()
        {
            long[] out = new long[3];
            GpuNative.lookupSourceStats(bridge, out);
            return out;
        }

        public void destroy()
        {
            GpuNative.destroyBridge(bridge);
        }
    }

    /**
     * HashBuilderOperatorFactory (operator/HashBuilderOperator.java:54-152; LocalExecutionPlanner.java:2104-2214).  partitionCount = the build side's driver
     * count (PartitionedLookupSourceFactory.java:110-124): 1 = one HashBuilderOperator, else one per local-exchange partition (a power of two).
     */
    public Optional<JoinBuild> hashBuilder(int operatorId, PlanNodeId planNodeId, List<Type> types, List<Integer> outputChannels, List<Integer> hashChannels,
            OptionalInt preComputedHashChannel, int expectedPositions, int partitionCount)
    {
        int[] codes;
        try {
            codes = GpuPages.typeCodes(types);
        }
        catch (IllegalArgumentException unsupportedType) {
            return Optional.empty();
        }
        long[] handles = GpuNative.createHashBuilderFactory(context, operatorId, codes, ints(outputChannels), ints(hashChannels), preComputedHashChannel.orElse(-1), expectedPositions,
                partitionCount);
        return Optional.of(new JoinBuild(new GpuOperatorFactory(operatorId, planNodeId, "GpuHashBuilderOperator", types, poller, handles[0]), handles[1], types));
    }

    /**
     * JoinFilterFunction (operator/JoinHash.java:44-47; sql/gen/JoinFilterFunctionCompiler.java; handed to the build side like JoinHashSupplier.java:54-70): `filter`
     * reads build channels [0, buildTypes) and probe channel k as channel buildTypes + k.  Before the probe factories are created; false = outside the IR.
     */
    public boolean joinFilter(JoinBuild build, List<Type> probeTypes, RowExpression filter)
    {
        Optional<GpuRowExpressions.Program> program = GpuRowExpressions.serialize(Optional.of(filter), List.of());
        if (program.isEmpty()) {
            return false;
        }
        GpuRowExpressions.Program p = program.get();
        GpuNative.setJoinFilter(build.bridge, GpuPages.typeCodes(probeTypes), p.nodeArray(), p.longArray(), p.doubleArray(), p.stringPool.toByteArray(), p.filterRoot, p.projectionRoots);
        return true;
    }

    /** LookupJoinOperators.innerJoin / probeOuterJoin / lookupOuterJoin / fullOuterJoin (operator/LookupJoinOperators.java:30-63; LocalExecutionPlanner.java:2284-2311) */
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
        long factory = GpuNative.createLookupJoinFactory(context, operatorId, build.bridge, codes, ints(probeJoinChannels), probeHashChannel.orElse(-1), ints(probeOutputChannels),
                joinType.ordinal());
        return Optional.of(new GpuOperatorFactory(operatorId, planNodeId, "GpuLookupJoinOperator", probeTypes, poller, factory));
    }

    /**
     * A FilterNode / ProjectNode directly under the probe side of a JoinNode as ONE fused pipeline (tgpu_filter_project_lookup_join_factory_create, the operator
     * bench.py's headline times): the projections form the probe page; probeJoinChannels / probeHashChannel / probeOutputChannels index them.
     */
    public Optional<OperatorFactory> filterProjectLookupJoin(int operatorId, PlanNodeId planNodeId, JoinBuild build, List<Type> inputTypes, Optional<RowExpression> filter,
            List<RowExpression> projections, List<Integer> probeJoinChannels, OptionalInt probeHashChannel, List<Integer> probeOutputChannels, JoinType joinType)
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
        long factory = GpuNative.createFilterProjectLookupJoinFactory(context, operatorId, build.bridge, types, p.nodeArray(), p.longArray(), p.doubleArray(),
                p.stringPool.toByteArray(), p.filterRoot, p.projectionRoots, ints(probeJoinChannels), probeHashChannel.orElse(-1), ints(probeOutputChannels), joinType.ordinal());
        return Optional.of(new GpuOperatorFactory(operatorId, planNodeId, "GpuFilterProjectLookupJoinOperator", inputTypes, poller, factory));
    }

    /** LookupJoinOperatorFactory.createOuterOperatorFactory (operator/LookupJoinOperatorFactory.java:88-103,141-146) */
    public OperatorFactory lookupOuter(int operatorId, PlanNodeId planNodeId, JoinBuild build, List<Type> probeOutputTypes)
    {
        int[] codes = GpuPages.typeCodes(probeOutputTypes);
        return new GpuOperatorFactory(operatorId, planNodeId, "GpuLookupOuterOperator", List.of(), poller, GpuNative.createLookupOuterFactory(context, operatorId, build.bridge, codes));
    }

    /** TopNOperator.createOperatorFactory (operator/TopNOperator.java:47-62; LocalExecutionPlanner.visitTopN) */
    public Optional<OperatorFactory> topN(int operatorId, PlanNodeId planNodeId, List<Type> types, long n, List<Integer> sortChannels, List<SortOrder> sortOrders)
    {
        int[] codes;
        try {
            codes = GpuPages.typeCodes(types);
        }
        catch (IllegalArgumentException unsupportedType) {
            return Optional.empty();
        }
        long factory = GpuNative.createTopNFactory(context, operatorId, codes, n, ints(sortChannels), sortOrders.stream().mapToInt(SortOrder::ordinal).toArray());
        return Optional.of(new GpuOperatorFactory(operatorId, planNodeId, "GpuTopNOperator", types, poller, factory));
    }

    /** OrderByOperator.OrderByOperatorFactory (operator/OrderByOperator.java:48-131; LocalExecutionPlanner.visitSort) */
    public Optional<OperatorFactory> orderBy(int operatorId, PlanNodeId planNodeId, List<Type> types, List<Integer> outputChannels, int expectedPositions,
            List<Integer> sortChannels, List<SortOrder> sortOrders)
    {
        int[] codes;
        try {
            codes = GpuPages.typeCodes(types);
        }
        catch (IllegalArgumentException unsupportedType) {
            return Optional.empty();
        }
        long factory = GpuNative.createOrderByFactory(context, operatorId, codes, ints(outputChannels), expectedPositions, ints(sortChannels),
                sortOrders.stream().mapToInt(SortOrder::ordinal).toArray());
        return Optional.of(new GpuOperatorFactory(operatorId, planNodeId, "GpuOrderByOperator", types, poller, factory));
    }

    /** MergePages (operator/project/MergePages.java:64-190) as an operator in front of GPU operators that want large pages (DESIGN.md "Page granularity") */
    public OperatorFactory mergePages(int operatorId, PlanNodeId planNodeId, List<Type> types, DataSize minPageSize, int minRowCount, DataSize maxPageSize)
    {
        long factory = GpuNative.createMergePagesFactory(context, operatorId, GpuPages.typeCodes(types), minPageSize.toBytes(), minRowCount, maxPageSize.toBytes());
        return new GpuOperatorFactory(operatorId, planNodeId, "GpuMergePagesOperator", types, poller, factory);
    }

    /**
     * PartitionedOutputOperator (operator/PartitionedOutputOperator.java:46-300): the shuffle producer.  The pending (partition, page) pairs are taken with
     * {@link #pollPartitionedOutput} (what PagePartitioner.flush hands to outputBuffer.enqueue, :451-470) or sent to the other GPUs by {@link GpuExchange}.
     * localPartitionFunction: the LocalPartitionGenerator function of local exchanges instead of (rawHash & 0x7fff...) % partitionCount.
     */
    public Optional<OperatorFactory> partitionedOutput(int operatorId, PlanNodeId planNodeId, List<Type> types, List<Integer> partitionChannels, OptionalInt hashChannel,
            int partitionCount, boolean replicatesAnyRow, OptionalInt nullChannel, boolean localPartitionFunction)
    {
        int[] codes;
        try {
            codes = GpuPages.typeCodes(types);
        }
        catch (IllegalArgumentException unsupportedType) {
            return Optional.empty();
        }
        long factory = GpuNative.createPartitionedOutputFactory(context, operatorId, codes, ints(partitionChannels), hashChannel.orElse(-1), partitionCount, replicatesAnyRow,
                nullChannel.orElse(-1), localPartitionFunction ? 1 : 0);
        return Optional.of(new GpuOperatorFactory(operatorId, planNodeId, "GpuPartitionedOutputOperator", types, poller, factory));
    }

    /** the next pending page of a partitioned-output operator as a device-resident Page, partition[0] = its partition; null when nothing is pending */
    public static io.trino.spi.Page pollPartitionedOutput(GpuOperator operator, int[] partition)
    {
        long page = GpuNative.partitionedOutputPoll(operator.nativeHandle(), partition);
        return page == 0 ? null : GpuPages.deviceResident(page);
    }

    /** DynamicFilterSourceOperatorFactory (operator/DynamicFilterSourceOperator.java:74-143; LocalExecutionPlanner.java:2216-2260) */
    public Optional<OperatorFactory> dynamicFilterSource(int operatorId, PlanNodeId planNodeId, List<Type> types, List<Integer> channels, int maxDistinctValues,
            DataSize maxFilterSize, int minMaxCollectionLimit)
    {
        int[] codes;
        try {
            codes = GpuPages.typeCodes(types);
        }
        catch (IllegalArgumentException unsupportedType) {
            return Optional.empty();
        }
        long factory = GpuNative.createDynamicFilterSourceFactory(context, operatorId, codes, ints(channels), maxDistinctValues, maxFilterSize.toBytes(), minMaxCollectionLimit);
        return Optional.of(new GpuOperatorFactory(operatorId, planNodeId, "GpuDynamicFilterSourceOperator", types, poller, factory));
    }

    /** the Domain of filter channel k after finish() (DynamicFilterSourceOperator.java:383-424): kindMinMax = {ALL 0 | VALUES 1 | RANGE 2 | NONE 3, min, max}; the values / VARCHAR range page or null */
    public static io.trino.spi.Page dynamicFilterResult(GpuOperator operator, int filterChannel, long[] kindMinMax)
    {
        long page = GpuNative.dynamicFilterSourceResult(operator.nativeHandle(), filterChannel, kindMinMax);
        return page == 0 ? null : GpuPages.deviceResident(page);
    }

    /**
     * The hop between two stages whose tasks are the GPUs of one node (tgpu_exchange_*: K10 partition kernels + one grouped RCCL all-to-all-v over xGMI)
     * instead of PartitionedOutputOperator -> OutputBuffer -> HTTP -> ExchangeOperator.  Collective: every rank makes the same calls in the same order.
     */
    public static final class GpuExchange
    {
        private final long exchange;

        /** rank 0 creates the id ({@link #uniqueId}) and ships it with the task's exchange locations */
        public GpuExchange(GpuOperatorFactories factories, byte[] uniqueId, int rank, int world)
        {
            this.exchange = GpuNative.createExchange(factories.context, uniqueId, rank, world);
        }

        public static byte[] uniqueId()
        {
            return GpuNative.exchangeUniqueId();
        }

        private static long handleOf(io.trino.spi.Page page)
        {
            GpuPages.DevicePageHandle handle = GpuPages.deviceHandle(page);
            if (handle == null) {
                throw new IllegalArgumentException("the GPU exchange moves device-resident pages");
            }
            return handle.handle;
        }

        /** FIXED_HASH_DISTRIBUTION: this rank's rows of every rank's page */
        public io.trino.spi.Page repartition(io.trino.spi.Page page, List<Integer> keyChannels, OptionalInt hashChannel)
        {
            return GpuPages.deviceResident(GpuNative.exchangeRepartition(exchange, handleOf(page), ints(keyChannels), hashChannel.orElse(-1)));
        }

        /** what a partitioned-output operator (partitionCount == world) has pending after one addInput, shuffled to its ranks */
        public io.trino.spi.Page partitionedOutput(GpuOperator partitionedOutputOperator, List<Type> types)
        {
            return GpuPages.deviceResident(GpuNative.exchangePartitionedOutput(exchange, partitionedOutputOperator.nativeHandle(), GpuPages.typeCodes(types)));
        }

        /** FIXED_BROADCAST_DISTRIBUTION (a replicated join build side) */
        public io.trino.spi.Page allGather(io.trino.spi.Page page)
        {
            return GpuPages.deviceResident(GpuNative.exchangeAllGather(exchange, handleOf(page)));
        }

        public long bytesSent()
        {
            return GpuNative.exchangeBytesSent(exchange);
        }

        public void close()
        {
            GpuNative.destroyExchange(exchange);
        }
    }
}
