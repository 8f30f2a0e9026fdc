package io.trino.operator.gpu;

import io.airlift.slice.Slice;
import io.trino.spi.type.Type;
import io.trino.sql.relational.CallExpression;
import io.trino.sql.relational.ConstantExpression;
import io.trino.sql.relational.InputReferenceExpression;
import io.trino.sql.relational.RowExpression;
import io.trino.sql.relational.SpecialForm;

import java.io.ByteArrayOutputStream;
import java.util.ArrayList;
import java.util.List;
import java.util.Optional;

/**
 * RowExpression (core/trino-main/src/main/java/io/trino/sql/relational) -> the tgpu_expr_node array of tgpu_page_processor_spec: what the
 * reference hands to ExpressionCompiler.compilePageProcessor (sql/gen/ExpressionCompiler.java:94-122) goes to the library's own compiler
 * (hiprtc, gfx950) instead.  An expression outside the library's IR makes {@link #serialize} return empty: the planner then keeps the Java operator.
 */
public final class GpuRowExpressions
{
    // tgpu_expr_kind / tgpu_expr_op / tgpu_special_form (include/tgpu.h)
    private static final int INPUT = 0, CONST = 1, CALL = 2, SPECIAL = 3;
    private static final String[] OPERATORS = {null, "$operator$ADD", "$operator$SUBTRACT", "$operator$MULTIPLY", "$operator$DIVIDE", "$operator$MODULUS", "$operator$NEGATION",
            "$operator$EQUAL", "$operator$NOT_EQUAL", "$operator$LESS_THAN", "$operator$LESS_THAN_OR_EQUAL", "$operator$GREATER_THAN", "$operator$GREATER_THAN_OR_EQUAL", "not",
            "$operator$CAST"};

    public static final class Program
    {
        public final List<int[]> nodes = new ArrayList<>();      // {kind, type, op, n_args, arg0, arg1, arg2, is_null, slen}
        public final List<Long> longValues = new ArrayList<>();
        public final List<Double> doubleValues = new ArrayList<>();
        public final ByteArrayOutputStream stringPool = new ByteArrayOutputStream();
        public int filterRoot = -1;
        public int[] projectionRoots = new int[0];

        public int[][] nodeArray()
        {
            return nodes.toArray(new int[0][]);
        }

        public long[] longArray()
        {
            return longValues.stream().mapToLong(Long::longValue).toArray();
        }

        public double[] doubleArray()
        {
            return doubleValues.stream().mapToDouble(Double::doubleValue).toArray();
        }
    }

    private GpuRowExpressions() {}

    public static Optional<Program> serialize(Optional<RowExpression> filter, List<RowExpression> projections)
    {
        try {
            Program program = new Program();
            if (filter.isPresent()) {
                program.filterRoot = add(program, filter.get());
            }
            program.projectionRoots = projections.stream().mapToInt(p -> add(program, p)).toArray();
            return Optional.of(program);
        }
        catch (IllegalArgumentException unsupported) {
            return Optional.empty();
        }
    }

    private static int emit(Program p, int kind, int type, int op, int[] args, boolean isNull, long longValue, double doubleValue, int stringLength)
    {
        int[] node = {kind, type, op, args.length, args.length > 0 ? args[0] : 0, args.length > 1 ? args[1] : 0, args.length > 2 ? args[2] : 0, isNull ? 1 : 0, stringLength};
        p.nodes.add(node);
        p.longValues.add(longValue);
        p.doubleValues.add(doubleValue);
        return p.nodes.size() - 1;
    }

    private static int add(Program p, RowExpression e)
    {
        int type = GpuPages.typeCode(e.getType());
        if (e instanceof InputReferenceExpression) {
            return emit(p, INPUT, type, ((InputReferenceExpression) e).getField(), new int[0], false, 0, 0, 0);
        }
        if (e instanceof ConstantExpression) {
            Object value = ((ConstantExpression) e).getValue();
            if (value == null) {
                return emit(p, CONST, type, 0, new int[0], true, 0, 0, 0);
            }
            if (value instanceof Slice) {
                byte[] bytes = ((Slice) value).getBytes();
                int at = p.stringPool.size();
                p.stringPool.write(bytes, 0, bytes.length);
                return emit(p, CONST, type, 0, new int[0], false, at, 0, bytes.length);
            }
            if (value instanceof Double) {
                return emit(p, CONST, type, 0, new int[0], false, 0, (Double) value, 0);
            }
            if (value instanceof Boolean) {
                return emit(p, CONST, type, 0, new int[0], false, (Boolean) value ? 1 : 0, 0, 0);
            }
            return emit(p, CONST, type, 0, new int[0], false, ((Number) value).longValue(), 0, 0);
        }
        if (e instanceof CallExpression) {
            CallExpression call = (CallExpression) e;
            String name = call.getResolvedFunction().getSignature().getName();
            for (int op = 1; op < OPERATORS.length; op++) {
                if (OPERATORS[op].equalsIgnoreCase(name)) {
                    int[] args = call.getArguments().stream().mapToInt(a -> add(p, a)).toArray();
                    return emit(p, CALL, type, op, args, false, 0, 0, 0);
                }
            }
            throw new IllegalArgumentException("function not in the GPU IR: " + name);
        }
        if (e instanceof SpecialForm) {
            SpecialForm form = (SpecialForm) e;
            int op;
            switch (form.getForm()) {
                case AND: op = 1; break;
                case OR: op = 2; break;
                case IF: op = 3; break;
                case IS_NULL: op = 4; break;
                case COALESCE: op = 5; break;
                case BETWEEN: op = 6; break;
                default: throw new IllegalArgumentException("special form not in the GPU IR: " + form.getForm());
            }
            List<RowExpression> arguments = form.getArguments();
            if (arguments.size() > 3) {
                throw new IllegalArgumentException("special form with more than 3 arguments");
            }
            int[] args = arguments.stream().mapToInt(a -> add(p, a)).toArray();
            return emit(p, SPECIAL, type, op, args, false, 0, 0, 0);
        }
        throw new IllegalArgumentException("expression not in the GPU IR: " + e);
    }
}
