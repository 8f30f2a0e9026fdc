package io.trino.operator.gpu;

import com.google.common.util.concurrent.ListenableFuture;
import com.google.common.util.concurrent.SettableFuture;
import io.trino.memory.context.LocalMemoryContext;
import io.trino.operator.Operator;
import io.trino.operator.OperatorContext;
import io.trino.spi.Page;
import io.trino.spi.type.Type;

import java.util.List;
import java.util.concurrent.ScheduledExecutorService;
import java.util.concurrent.TimeUnit;

/**
 * io.trino.operator.Operator (core/trino-main/src/main/java/io/trino/operator/Operator.java:20-102) over a tgpu_operator handle.
 * Same call protocol and error behaviour as the Java operator it replaces; one thread at a time per instance (Driver.java:55-62).
 */
public class GpuOperator
        implements Operator
{
    private final OperatorContext operatorContext;
    private final LocalMemoryContext memory;
    private final LocalMemoryContext revocableMemory;
    private final ScheduledExecutorService poller;
    protected long handle;                                 // tgpu_operator*
    private final boolean[] wouldBlock = new boolean[1];

    private final List<Type> inputTypes;                   // the type of every input channel (LongArrayBlock holds BIGINT or DOUBLE: the channel type decides)

    public GpuOperator(OperatorContext operatorContext, long handle, List<Type> inputTypes, ScheduledExecutorService poller)
    {
        this.inputTypes = inputTypes;
        this.operatorContext = operatorContext;
        this.memory = operatorContext.localUserMemoryContext();
        this.revocableMemory = operatorContext.localRevocableMemoryContext();
        this.handle = handle;
        this.poller = poller;
    }

    @Override
    public OperatorContext getOperatorContext()
    {
        return operatorContext;
    }

    /** the build -> probe dependency (LookupJoinOperator.java:235-243) and probes -> outer operator: polled, the driver thread never blocks */
    @Override
    public ListenableFuture<?> isBlocked()
    {
        if (!call(() -> GpuNative.isBlocked(handle))) {
            return NOT_BLOCKED;
        }
        SettableFuture<?> unblocked = SettableFuture.create();
        Runnable check = new Runnable()
        {
            @Override
            public void run()
            {
                if (handle == 0 || !GpuNative.isBlocked(handle)) {
                    unblocked.set(null);
                }
                else {
                    poller.schedule(this, 1, TimeUnit.MILLISECONDS);
                }
            }
        };
        poller.schedule(check, 1, TimeUnit.MILLISECONDS);
        return unblocked;
    }

    @Override
    public boolean needsInput()
    {
        return call(() -> GpuNative.needsInput(handle));
    }

    @Override
    public void addInput(Page page)
    {
        try {
            GpuPages.DevicePageHandle device = GpuPages.deviceHandle(page);
            if (device != null) {
                // the producer was a GPU operator and nothing touched the page in between: it never left HBM (tgpu_operator_add_input_output_page
                // shares the buffers, so the producer's handle can go at once)
                GpuNative.addInputDevicePage(handle, device.handle);
                device.release();
            }
            else {
                GpuPages.addInput(handle, page.getLoadedPage(), inputTypes);
            }
        }
        catch (GpuNative.NativeError e) {
            throw GpuNative.toTrinoException(e);
        }
        updateMemory();
    }

    // OperatorContext.java:263-275; SpillableHashAggregationBuilder.updateMemory :117-128 (user vs revocable)
    protected void updateMemory()
    {
        memory.setBytes(GpuNative.memoryBytes(handle));
        revocableMemory.setBytes(GpuNative.revocableMemoryBytes(handle));
    }

    @Override
    public Page getOutput()
    {
        try {
            long page = GpuNative.getOutput(handle, wouldBlock);
            return page == 0 ? null : GpuPages.deviceResident(page);
        }
        catch (GpuNative.NativeError e) {
            throw GpuNative.toTrinoException(e);
        }
    }

    @Override
    public void finish()
    {
        try {
            GpuNative.finish(handle);
        }
        catch (GpuNative.NativeError e) {
            throw GpuNative.toTrinoException(e);
        }
    }

    @Override
    public boolean isFinished()
    {
        return call(() -> GpuNative.isFinished(handle));
    }

    // Only a spill-enabled SINGLE / FINAL hash aggregation holds revocable memory (GpuNative.setSpillEnabled on its factory): the revoke moves
    // its groups to host memory and is complete when the native call returns (SpillableHashAggregationBuilder.startMemoryRevoke)
    @Override
    public ListenableFuture<?> startMemoryRevoke()
    {
        try {
            GpuNative.startMemoryRevoke(handle);
        }
        catch (GpuNative.NativeError e) {
            throw GpuNative.toTrinoException(e);
        }
        return NOT_BLOCKED;
    }

    @Override
    public void finishMemoryRevoke()
    {
        GpuNative.finishMemoryRevoke(handle);
        updateMemory();
    }

    @Override
    public void close()
    {
        if (handle != 0) {
            GpuNative.close(handle);
            handle = 0;
            memory.setBytes(0);
            revocableMemory.setBytes(0);
        }
    }

    /** the operator's native handle, for the pieces that are not part of the Operator interface (PartitionedOutput's pending pages, DynamicFilterSource's result) */
    public long nativeHandle()
    {
        return handle;
    }

    protected interface BooleanCall
    {
        boolean get();
    }

    protected static boolean call(BooleanCall c)
    {
        try {
            return c.get();
        }
        catch (GpuNative.NativeError e) {
            throw GpuNative.toTrinoException(e);
        }
    }
}
