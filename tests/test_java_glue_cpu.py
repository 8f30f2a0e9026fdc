"""Static checks of the Java glue (java/io/trino/...) against the reference's SOURCES -- there is no JDK in the image, so nothing compiles the glue;
these tests catch the class of mistake round 2 shipped (a glue class extending the final io.trino.spi.Page).  They read /root/reference as text and
are skipped where it does not exist (the GPU box).  Checked: every io.trino import resolves to a reference class (nested ones included); no glue class
extends a final class or implements a class; every @Override method is declared by a supertype (the reference's interface / class, or another glue
class); every reference constructor the glue calls exists with that number of arguments; the methods and private fields of reference classes the glue
relies on exist under those names; the handles GpuNative declares are the ones the glue calls."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SRC_ROOTS = [os.path.join(REF, "core/trino-main/src/main/java"), os.path.join(REF, "core/trino-spi/src/main/java"), os.path.join(REF, "lib/trino-memory-context/src/main/java")]
GLUE = os.path.join(ROOT, "java")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference sources are only present in the build container")


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def glue_files():
    out = {}
    for dirpath, _, files in os.walk(GLUE):
        for f in files:
            if f.endswith(".java"):
                out[os.path.join(dirpath, f)] = strip_comments(open(os.path.join(dirpath, f)).read())
    return out


def ref_source(qualified):
    """source text of the file that declares `qualified` (a.b.Outer or a.b.Outer.Inner), or None"""
    parts = qualified.split(".")
    for cut in (len(parts), len(parts) - 1):
        rel = os.path.join(*parts[:cut]) + ".java"
        for root in SRC_ROOTS:
            p = os.path.join(root, rel)
            if os.path.exists(p):
                text = strip_comments(open(p).read())
                if cut == len(parts) or re.search(r"\b(class|interface|enum)\s+%s\b" % parts[-1], text):
                    return text
    return None


def imports_of(text):
    return re.findall(r"^import (?:static )?([\w.]+);", text, flags=re.M)


def resolve(name, text, package):
    """simple name -> qualified name through the file's imports, its own package, or the glue itself"""
    for imp in imports_of(text):
        if imp.endswith("." + name):
            return imp
    return package + "." + name


def test_every_reference_import_exists():
    for path, text in glue_files().items():
        for imp in imports_of(text):
            if imp.startswith("io.trino.") and not imp.startswith(("io.trino.operator.gpu.", "io.trino.spi.block.GpuBlockAccess")):
                base = imp
                if re.search(r"import static " + re.escape(imp), text):   # static import of a member: drop the member
                    base = imp.rsplit(".", 1)[0]
                assert ref_source(base) is not None, f"{os.path.relpath(path, ROOT)} imports {imp}, which the reference does not have"


def declared_types(text):
    """(kind, name, extends, implements list) of every class / interface declared in a file"""
    out = []
    for m in re.finditer(r"\b(class|interface)\s+(\w+)\s*(?:extends\s+([\w.]+))?\s*(?:implements\s+([\w.,\s]+?))?\s*\{", text):
        impl = [x.strip() for x in (m.group(4) or "").split(",") if x.strip()]
        out.append((m.group(1), m.group(2), m.group(3), impl))
    return out


def glue_type_sources():
    out = {}
    for path, text in glue_files().items():
        pkg = re.search(r"^package ([\w.]+);", text, flags=re.M).group(1)
        for kind, name, ext, impl in declared_types(text):
            out[name] = (pkg, text, ext, impl)
    return out


def test_no_glue_class_extends_a_final_class_or_implements_a_class():
    glue = glue_type_sources()
    checked = 0
    for path, text in glue_files().items():
        pkg = re.search(r"^package ([\w.]+);", text, flags=re.M).group(1)
        for kind, name, ext, impl in declared_types(text):
            for sup, want_interface in [(ext, False)] + [(i, True) for i in impl]:
                if not sup or sup in ("RuntimeException",):
                    continue
                simple = sup.split(".")[-1]
                if simple in glue:
                    continue
                q = resolve(simple, text, pkg)
                src = ref_source(q)
                assert src is not None, f"{name}: supertype {sup} not found in the reference"
                decl = re.search(r"([\w\s]*?)\b(class|interface)\s+%s\b" % simple, src)
                assert decl, (name, sup)
                checked += 1
                if want_interface:
                    assert decl.group(2) == "interface", f"{name} implements {sup}, which is a class"
                else:
                    assert decl.group(2) == "class" and "final" not in decl.group(1).split(), f"{name} extends {sup}, which is final or not a class: '{decl.group(0).strip()}'"
    assert checked >= 5


def supertype_texts(name, glue, seen=None):
    """the source texts of every supertype of glue class `name` (transitively, through the glue and the reference)"""
    seen = seen if seen is not None else set()
    pkg, text, ext, impl = glue[name]
    out = []
    for sup in [ext] + impl:
        if not sup:
            continue
        simple = sup.split(".")[-1]
        if simple in seen:
            continue
        seen.add(simple)
        if simple in glue:
            out.append(glue[simple][1])
            out += supertype_texts(simple, glue, seen)
            continue
        src = ref_source(resolve(simple, text, pkg))
        if src is None:
            continue
        out.append(src)
        # one more level inside the reference: interface Foo extends Bar
        for m in re.finditer(r"\b(?:class|interface)\s+%s\b[^{]*?\b(?:extends|implements)\s+([\w.,\s]+?)\s*\{" % simple, src):
            for up in m.group(1).replace("implements", ",").split(","):
                up = up.strip().split(".")[-1]
                if up and up not in seen:
                    seen.add(up)
                    pk = re.search(r"^package ([\w.]+);", src, flags=re.M).group(1)
                    s2 = ref_source(resolve(up, src, pk))
                    if s2:
                        out.append(s2)
    return out


def test_every_override_is_declared_by_a_supertype():
    glue = glue_type_sources()
    checked = 0
    for path, text in glue_files().items():
        # the class each @Override sits in: the innermost type declaration in front of it (anonymous Runnable / lambdas aside)
        for m in re.finditer(r"@Override\s+(?:public|protected)?\s*(?:[\w<>\[\],.?\s]+?)\s+(\w+)\s*\(([^)]*)\)", text):
            method, params = m.group(1), [p for p in m.group(2).split(",") if p.strip()]
            before = text[:m.start()]
            owner = None
            for t in re.finditer(r"\b(?:class|interface)\s+(\w+)", before):
                owner = t.group(1)
            anon = re.search(r"new (\w+)\(\)\s*\{[^{}]*$", before[-400:])
            if anon:                       # e.g. new Runnable() { @Override public void run() ... }
                if anon.group(1) == "Runnable":
                    assert method == "run" and not params
                    continue
            assert owner in glue, (path, method)
            sups = supertype_texts(owner, glue)
            assert sups, f"{owner}.{method} is marked @Override but {owner} has no supertype"
            pattern = re.compile(r"\b%s\s*\(([^)]*)\)" % method)
            found = False
            for src in sups:
                for d in pattern.finditer(src):
                    if len([p for p in d.group(1).split(",") if p.strip()]) == len(params):
                        found = True
            assert found, f"{owner}.{method}({len(params)} args) overrides nothing its supertypes declare"
            checked += 1
    assert checked >= 20


def test_reference_constructors_and_members_the_glue_relies_on_exist():
    S = "io.trino.spi."
    M = "io.trino."
    wants = [
        # (qualified class, regex that must match its source)
        (S + "Page", r"public Page\(int positionCount, Block\.\.\. blocks\)"),
        (S + "Page", r"public Page getLoadedPage\(\)"),
        (S + "Page", r"public final class Page"),                       # why nothing in the glue subclasses it
        (S + "block.LazyBlock", r"public class LazyBlock"),
        (S + "block.LazyBlock", r"public LazyBlock\(int positionCount, LazyBlockLoader loader\)"),
        (S + "block.LazyBlockLoader", r"Block load\(\);"),
        (S + "block.Block", r"default Block getLoadedBlock\(\)"),
        (S + "block.LongArrayBlock", r"public LongArrayBlock\(int positionCount, Optional<boolean\[\]> valueIsNull, long\[\] values\)"),
        (S + "block.IntArrayBlock", r"public IntArrayBlock\(int positionCount, Optional<boolean\[\]> valueIsNull, int\[\] values\)"),
        (S + "block.ByteArrayBlock", r"public ByteArrayBlock\(int positionCount, Optional<boolean\[\]> valueIsNull, byte\[\] values\)"),
        (S + "block.VariableWidthBlock", r"public VariableWidthBlock\(int positionCount, Slice slice, int\[\] offsets, Optional<boolean\[\]> valueIsNull\)"),
        (S + "block.VariableWidthBlock", r"protected Slice getRawSlice\(int position\)"),
        (S + "block.DictionaryBlock", r"public Block getDictionary\(\)"),
        (S + "block.RunLengthEncodedBlock", r"public Block getValue\(\)"),
        (S + "block.BlockBuilder", r"Block build\(\);"),
        (S + "type.Type", r"BlockBuilder createBlockBuilder\(BlockBuilderStatus blockBuilderStatus, int expectedEntries\);"),
        (S + "type.Type", r"void appendTo\(Block block, int position, BlockBuilder blockBuilder\);"),
        (S + "connector.ConnectorPageSource", r"Page getNextPage\(\);"),
        (S + "connector.ConnectorPageSource", r"default CompletableFuture<\?> isBlocked\(\)"),
        (S + "connector.ConnectorPageSource", r"void close\(\)\s+throws IOException;"),
        (S + "connector.SortOrder", r"ASC_NULLS_FIRST\(true, true\),\s+ASC_NULLS_LAST\(true, false\),\s+DESC_NULLS_FIRST\(false, true\),\s+DESC_NULLS_LAST\(false, false\)"),
        (M + "operator.LookupJoinOperators", r"INNER,\s+PROBE_OUTER,[^\n]*\s+LOOKUP_OUTER,[^\n]*\s+FULL_OUTER"),   # ordinals = tgpu_join_type
        (M + "operator.OperatorContext", r"public LocalMemoryContext localUserMemoryContext\(\)"),
        (M + "operator.OperatorContext", r"public LocalMemoryContext localRevocableMemoryContext\(\)"),
        (M + "operator.DriverContext", r"public OperatorContext addOperatorContext\(int operatorId, PlanNodeId planNodeId, String operatorType\)"),
        (M + "operator.SourceOperator", r"Supplier<Optional<UpdatablePageSource>> addSplit\(Split split\);"),
        (M + "operator.SourceOperatorFactory", r"SourceOperator createOperator\(DriverContext driverContext\);"),
        (M + "sql.relational.CallExpression", r"public ResolvedFunction getResolvedFunction\(\)"),
        (M + "sql.relational.InputReferenceExpression", r"public int getField\(\)"),
        (M + "sql.relational.ConstantExpression", r"public Object getValue\(\)"),
        (M + "sql.relational.SpecialForm", r"public Form getForm\(\)"),
        (M + "sql.relational.SpecialForm", r"\bBETWEEN\b"),
        (M + "sql.planner.plan.AggregationNode", r"enum Step"),
    ]
    for cls, pattern in wants:
        src = ref_source(cls)
        assert src is not None, cls
        assert re.search(pattern, src), f"{cls}: /{pattern}/ not found"
    # the private fields GpuBlockAccess reads through reflection
    for cls, fields in (("LongArrayBlock", ["values", "valueIsNull", "arrayOffset"]), ("IntArrayBlock", ["values", "valueIsNull", "arrayOffset"]),
                        ("ByteArrayBlock", ["values", "valueIsNull", "arrayOffset"]), ("VariableWidthBlock", ["offsets", "valueIsNull", "arrayOffset"]),
                        ("DictionaryBlock", ["ids", "idsOffset"])):
        src = ref_source(S + "block." + cls)
        for f in fields:
            assert re.search(r"private final [\w\[\]<>]+ %s;" % f, src), (cls, f)
    access = open(os.path.join(GLUE, "io/trino/spi/block/GpuBlockAccess.java")).read()
    assert len(re.findall(r"getDeclaredField", access)) == 1 and "static final Field" in access      # looked up once, cached
    assert "copyRegion" not in strip_comments(access)                                                # the recursion of round 2 is gone


def test_the_glue_calls_only_natives_gpu_native_declares_and_uses_the_fused_factories():
    files = glue_files()
    native = [t for p, t in files.items() if p.endswith("GpuNative.java")][0]
    declared = set(re.findall(r"public static native [\w\[\]]+ (\w+)\(", native))
    called = set()
    for p, t in files.items():
        called |= set(re.findall(r"GpuNative\.(\w+)\(", t))
    called -= {"toTrinoException"}
    assert called <= declared, called - declared
    for must in ("createFilterProjectLookupJoinFactory", "createFilterProjectHashAggregationFactory", "createScanFilterProjectFactory", "createTopNFactory", "createOrderByFactory",
                 "createMergePagesFactory", "createPartitionedOutputFactory", "createDynamicFilterSourceFactory", "setJoinFilter", "createExchange", "setMaxOutputPage",
                 "setDoubleSumOrder", "scanAddPageSource"):
        assert must in called, must
    unused = declared - called
    assert unused <= {"synchronizeContext", "destroyContext", "profileEnable", "profileDump", "spillStats", "partitionedOutputInfo", "scanStats", "destroyFactory", "orcDecodeLongColumn", "orcDecodeBooleanColumn", "orcDecodeDictionaryStringColumn", "orcDecodeDirectStringColumn", "orcDecodeDoubleColumn", "parquetDecodeDataPage"}, unused
