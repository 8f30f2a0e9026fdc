// jit.cpp -- RowExpression -> HIP source -> hiprtc (gfx950) -> cached code object -> hipModule.
//
// Semantics generated (all from the reference's bytecode generators):
//   * calls propagate null and evaluate arguments left to right, stopping at the first null argument
//     (M/sql/gen/BytecodeUtils.java:189-356);
//   * AND / OR short-circuit with three-valued logic (AndCodeGenerator.java:44-105, OrCodeGenerator.java);
//   * filter: selected = !wasNull && value (PageFunctionCompiler.java:502-544); selected positions keep input order
//     (M/operator/project/PageFilter.java:27-50);
//   * projections are evaluated on the selected positions only (PageProcessor.java:120-136), so an arithmetic error in
//     a projection is raised only for selected rows; BIGINT / INTEGER arithmetic is checked (M/type/BigintOperators.java:47-113).
#include "jit.h"

#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <fstream>
#include <set>
#include <sstream>

#include "kernels.h"

namespace tgpu {

// ---------------------------------------------------------------------------------------------------------------------
// resource dir + code-object cache
// ---------------------------------------------------------------------------------------------------------------------
static std::string g_resource_dir;
static std::mutex g_jit_mu;

std::string resource_dir()
{
    std::lock_guard<std::mutex> lk(g_jit_mu);
    if (g_resource_dir.empty()) {
        Dl_info info;
        if (dladdr((void *)&resource_dir, &info) && info.dli_fname) {
            std::string p = info.dli_fname;
            size_t slash = p.find_last_of('/');
            g_resource_dir = slash == std::string::npos ? "." : p.substr(0, slash);
        }
        else g_resource_dir = ".";
    }
    return g_resource_dir;
}

void set_resource_dir(const std::string &dir)
{
    std::lock_guard<std::mutex> lk(g_jit_mu);
    g_resource_dir = dir;
}

static uint64_t fnv1a(const std::string &s)
{
    uint64_t h = 1469598103934665603ULL;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ULL; }
    return h;
}

static std::vector<char> compile_source(const std::string &source)
{
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, source.c_str(), "tgpu_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        fail(TGPU_ERR_COMPILER, "hiprtcCreateProgram failed");
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, ' ');
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        fail(TGPU_ERR_COMPILER, "kernel compilation failed: " + log);
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    return code;
}

static std::string cache_path(const std::string &source)
{
    char name[64];
    snprintf(name, sizeof(name), "%016llx.hsaco", (unsigned long long)fnv1a(source));
    return resource_dir() + "/_kcache/" + name;
}

// compile (or fetch from the disk cache) the code object of `source`
static std::vector<char> code_object_for(const std::string &source)
{
    const std::string path = cache_path(source);
    {
        std::ifstream f(path, std::ios::binary);
        if (f) {
            std::vector<char> code((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            if (!code.empty()) return code;
        }
    }
    std::vector<char> code = compile_source(source);
    mkdir((resource_dir() + "/_kcache").c_str(), 0755);
    const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
    {
        std::ofstream f(tmp, std::ios::binary);
        if (f) {
            f.write(code.data(), (std::streamsize)code.size());
            f.close();
            rename(tmp.c_str(), path.c_str());  // atomic publish; a read-only tree just skips the cache
        }
    }
    return code;
}

struct JitModule {
    hipModule_t mod = nullptr;
    hipFunction_t count = nullptr, emit = nullptr;
    ~JitModule()
    {
        if (mod) hipModuleUnload(mod);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// code generation
// ---------------------------------------------------------------------------------------------------------------------
namespace {

const char *ctype(int32_t t)
{
    switch (t) {
    case TGPU_BIGINT: return "long long";
    case TGPU_INTEGER: case TGPU_DATE: return "int";
    case TGPU_DOUBLE: return "double";
    case TGPU_BOOLEAN: return "bool";
    default: return "?";
    }
}

struct Val {
    std::string v, n, len;  // value var, null-flag var, (VARCHAR) length var
    int32_t type;
};

struct Gen {
    const std::vector<tgpu_expr_node> &nodes;
    const std::string &pool;
    const std::vector<int32_t> &in_types;
    std::ostringstream os;
    std::ostringstream consts;
    int tmp = 0;
    std::set<int> used_cols;

    Gen(const std::vector<tgpu_expr_node> &n, const std::string &p, const std::vector<int32_t> &t) : nodes(n), pool(p), in_types(t) {}

    std::string ind(int d) { return std::string((size_t)d * 2, ' '); }

    [[noreturn]] void bad(const std::string &m) { fail(TGPU_ERR_COMPILER, "expression compiler: " + m); }

    const tgpu_expr_node &node(int i)
    {
        if (i < 0 || i >= (int)nodes.size()) bad("node index out of range");
        return nodes[i];
    }

    Val declare(int32_t type, int d)
    {
        Val r;
        r.type = type;
        int k = tmp++;
        r.v = "v" + std::to_string(k);
        r.n = "n" + std::to_string(k);
        if (type == TGPU_VARCHAR) {
            r.len = "l" + std::to_string(k);
            os << ind(d) << "const unsigned char* " << r.v << " = 0; int " << r.len << " = 0; bool " << r.n << " = false;\n";
        }
        else {
            if (!valid_type(type)) bad("bad expression type");
            os << ind(d) << ctype(type) << " " << r.v << " = 0; bool " << r.n << " = false;\n";
        }
        return r;
    }

    std::string fmt_double(double x)
    {
        // exact round trip through the bit pattern
        unsigned long long bits;
        memcpy(&bits, &x, 8);
        std::ostringstream s;
        s << "__longlong_as_double((long long)0x" << std::hex << bits << "ULL)";
        return s.str();
    }

    void error_stmt(int d, const char *code, const Val &r)
    {
        os << ind(d) << "{ tg_error(A.error, row, " << code << "); " << r.n << " = true; }\n";
    }

    // emits code that computes node idx into a fresh (value, null) pair declared at depth d
    Val gen(int idx, int d)
    {
        const tgpu_expr_node &nd = node(idx);
        switch (nd.kind) {
        case TGPU_EX_INPUT: {
            int ch = nd.op;
            if (ch < 0 || ch >= (int)in_types.size()) bad("input channel out of range");
            if (ch >= kFpMaxCols) bad("too many input channels");
            if (in_types[ch] != nd.type) bad("input reference type does not match the channel type");
            used_cols.insert(ch);
            Val r = declare(nd.type, d);
            std::string c = "c" + std::to_string(ch), cn = "cn" + std::to_string(ch), co = "co" + std::to_string(ch);
            os << ind(d) << r.n << " = " << cn << " && " << cn << "[row];\n";
            if (nd.type == TGPU_VARCHAR)
                os << ind(d) << "if (!" << r.n << ") { int a_ = " << co << "[row]; " << r.v << " = " << c << " + a_; " << r.len << " = " << co << "[row + 1] - a_; }\n";
            else if (nd.type == TGPU_BOOLEAN)
                os << ind(d) << "if (!" << r.n << ") " << r.v << " = " << c << "[row] != 0;\n";
            else
                os << ind(d) << "if (!" << r.n << ") " << r.v << " = " << c << "[row];\n";
            return r;
        }
        case TGPU_EX_CONST: {
            Val r = declare(nd.type, d);
            if (nd.is_null) {
                os << ind(d) << r.n << " = true;\n";
                return r;
            }
            switch (nd.type) {
            case TGPU_BIGINT: os << ind(d) << r.v << " = (long long)" << (long long)nd.ival << "LL;\n"; break;
            case TGPU_INTEGER: case TGPU_DATE: os << ind(d) << r.v << " = (int)" << (long long)nd.ival << "LL;\n"; break;
            case TGPU_BOOLEAN: os << ind(d) << r.v << " = " << (nd.ival ? "true" : "false") << ";\n"; break;
            case TGPU_DOUBLE: os << ind(d) << r.v << " = " << fmt_double(nd.dval) << ";\n"; break;
            case TGPU_VARCHAR: {
                if (nd.ival < 0 || nd.slen < 0 || (size_t)(nd.ival + nd.slen) > pool.size()) bad("string constant outside the pool");
                std::string name = "K" + std::to_string(idx);
                consts << "__device__ const unsigned char " << name << "[" << (nd.slen > 0 ? nd.slen : 1) << "] = {";
                for (int i = 0; i < nd.slen; i++) consts << (i ? "," : "") << (int)(unsigned char)pool[(size_t)nd.ival + i];
                if (nd.slen == 0) consts << "0";
                consts << "};\n";
                os << ind(d) << r.v << " = " << name << "; " << r.len << " = " << nd.slen << ";\n";
                break;
            }
            default: bad("bad constant type");
            }
            return r;
        }
        case TGPU_EX_CALL: return gen_call(nd, d);
        case TGPU_EX_SPECIAL: return gen_special(nd, d);
        default: bad("unknown node kind");
        }
    }

    static bool is_cmp(int op) { return op >= TGPU_OP_EQUAL && op <= TGPU_OP_GREATER_THAN_OR_EQUAL; }

    std::string cmp_expr(int32_t t, int op, const Val &a, const Val &b)
    {
        const char *sym = op == TGPU_OP_EQUAL ? "==" : op == TGPU_OP_NOT_EQUAL ? "!=" : op == TGPU_OP_LESS_THAN ? "<" :
                          op == TGPU_OP_LESS_THAN_OR_EQUAL ? "<=" : op == TGPU_OP_GREATER_THAN ? ">" : ">=";
        if (t == TGPU_VARCHAR) return "(tg_strcmp(" + a.v + ", " + a.len + ", " + b.v + ", " + b.len + ") " + sym + " 0)";
        if (t == TGPU_BOOLEAN) return "((int)" + a.v + " " + sym + " (int)" + b.v + ")";
        return "(" + a.v + " " + sym + " " + b.v + ")";
    }

    Val gen_call(const tgpu_expr_node &nd, int d)
    {
        if (nd.n_args < 1 || nd.n_args > 2) bad("calls take 1 or 2 arguments");
        Val r = declare(nd.type, d);
        // arguments left to right; the first null argument skips the rest and the call
        os << ind(d) << "{\n";
        Val a = gen(nd.args[0], d + 1);
        os << ind(d + 1) << "if (" << a.n << ") " << r.n << " = true; else {\n";
        int dd = d + 2;
        Val b;
        if (nd.n_args == 2) {
            b = gen(nd.args[1], dd);
            os << ind(dd) << "if (" << b.n << ") " << r.n << " = true; else {\n";
            dd++;
        }
        const int32_t at = a.type;
        switch (nd.op) {
        case TGPU_OP_ADD: case TGPU_OP_SUBTRACT: case TGPU_OP_MULTIPLY: case TGPU_OP_DIVIDE: case TGPU_OP_MODULUS: {
            if (nd.n_args != 2 || a.type != nd.type || b.type != nd.type) bad("arithmetic operand types must equal the result type");
            const char *sym = nd.op == TGPU_OP_ADD ? "+" : nd.op == TGPU_OP_SUBTRACT ? "-" : nd.op == TGPU_OP_MULTIPLY ? "*" : nd.op == TGPU_OP_DIVIDE ? "/" : "%";
            if (nd.type == TGPU_DOUBLE) {
                if (nd.op == TGPU_OP_MODULUS) os << ind(dd) << r.v << " = fmod(" << a.v << ", " << b.v << ");\n";
                else os << ind(dd) << r.v << " = " << a.v << " " << sym << " " << b.v << ";\n";
            }
            else if (nd.type == TGPU_BIGINT || nd.type == TGPU_INTEGER) {
                const char *T = nd.type == TGPU_BIGINT ? "long long" : "int";
                const char *MINV = nd.type == TGPU_BIGINT ? "(-9223372036854775807LL - 1)" : "(-2147483647 - 1)";
                if (nd.op == TGPU_OP_ADD || nd.op == TGPU_OP_SUBTRACT || nd.op == TGPU_OP_MULTIPLY) {
                    const char *fn = nd.op == TGPU_OP_ADD ? "__builtin_add_overflow" : nd.op == TGPU_OP_SUBTRACT ? "__builtin_sub_overflow" : "__builtin_mul_overflow";
                    os << ind(dd) << T << " t_; if (" << fn << "(" << a.v << ", " << b.v << ", &t_))";
                    os << " { tg_error(A.error, row, TG_E_RANGE); " << r.n << " = true; } else " << r.v << " = t_;\n";
                }
                else {
                    os << ind(dd) << "if (" << b.v << " == 0) { tg_error(A.error, row, TG_E_DIV0); " << r.n << " = true; }\n";
                    if (nd.op == TGPU_OP_DIVIDE) {
                        os << ind(dd) << "else if (" << a.v << " == " << MINV << " && " << b.v << " == -1) { tg_error(A.error, row, TG_E_RANGE); " << r.n << " = true; }\n";
                        os << ind(dd) << "else " << r.v << " = " << a.v << " / " << b.v << ";\n";
                    }
                    else {
                        os << ind(dd) << "else " << r.v << " = (" << b.v << " == -1) ? 0 : " << a.v << " % " << b.v << ";\n";
                    }
                }
            }
            else bad("arithmetic on unsupported type");
            break;
        }
        case TGPU_OP_NEGATE:
            if (nd.n_args != 1 || a.type != nd.type) bad("negate operand type");
            if (nd.type == TGPU_DOUBLE) os << ind(dd) << r.v << " = -" << a.v << ";\n";
            else if (nd.type == TGPU_BIGINT)
                os << ind(dd) << "if (" << a.v << " == (-9223372036854775807LL - 1)) { tg_error(A.error, row, TG_E_RANGE); " << r.n << " = true; } else " << r.v << " = -" << a.v << ";\n";
            else if (nd.type == TGPU_INTEGER)
                os << ind(dd) << "if (" << a.v << " == (-2147483647 - 1)) { tg_error(A.error, row, TG_E_RANGE); " << r.n << " = true; } else " << r.v << " = -" << a.v << ";\n";
            else bad("negate on unsupported type");
            break;
        case TGPU_OP_NOT:
            if (nd.n_args != 1 || a.type != TGPU_BOOLEAN || nd.type != TGPU_BOOLEAN) bad("NOT needs a boolean");
            os << ind(dd) << r.v << " = !" << a.v << ";\n";
            break;
        case TGPU_OP_CAST:
            if (nd.n_args != 1) bad("cast takes one argument");
            if (nd.type == at) os << ind(dd) << r.v << " = " << a.v << ";" << (at == TGPU_VARCHAR ? (" " + r.len + " = " + a.len + ";") : "") << "\n";
            else if (nd.type == TGPU_DOUBLE && (at == TGPU_BIGINT || at == TGPU_INTEGER)) os << ind(dd) << r.v << " = (double)" << a.v << ";\n";
            else if (nd.type == TGPU_BIGINT && at == TGPU_INTEGER) os << ind(dd) << r.v << " = (long long)" << a.v << ";\n";
            else if (nd.type == TGPU_INTEGER && at == TGPU_BIGINT)
                os << ind(dd) << "if (" << a.v << " > 2147483647LL || " << a.v << " < -2147483648LL) { tg_error(A.error, row, TG_E_RANGE); " << r.n << " = true; } else " << r.v << " = (int)" << a.v << ";\n";
            else bad("unsupported cast");
            break;
        default:
            if (is_cmp(nd.op)) {
                if (nd.n_args != 2 || a.type != b.type || nd.type != TGPU_BOOLEAN) bad("comparison operand types must match");
                os << ind(dd) << r.v << " = " << cmp_expr(at, nd.op, a, b) << ";\n";
            }
            else bad("unknown call op");
        }
        if (nd.n_args == 2) os << ind(d + 2) << "}\n";
        os << ind(d + 1) << "}\n";
        os << ind(d) << "}\n";
        return r;
    }

    Val gen_special(const tgpu_expr_node &nd, int d)
    {
        switch (nd.op) {
        case TGPU_SF_AND:
        case TGPU_SF_OR: {
            if (nd.n_args != 2 || nd.type != TGPU_BOOLEAN) bad("AND/OR take two booleans");
            const bool is_and = nd.op == TGPU_SF_AND;
            Val r = declare(TGPU_BOOLEAN, d);
            os << ind(d) << "{\n";
            Val l = gen(nd.args[0], d + 1);
            if (l.type != TGPU_BOOLEAN) bad("AND/OR operand must be boolean");
            // left decides alone when it is FALSE (AND) / TRUE (OR): right is not evaluated
            os << ind(d + 1) << "if (!" << l.n << " && " << (is_and ? "!" : "") << l.v << ") " << r.v << " = " << (is_and ? "false" : "true") << "; else {\n";
            Val rt = gen(nd.args[1], d + 2);
            if (rt.type != TGPU_BOOLEAN) bad("AND/OR operand must be boolean");
            os << ind(d + 2) << "if (" << rt.n << ") " << r.n << " = true;\n";
            os << ind(d + 2) << "else if (" << (is_and ? "!" : "") << rt.v << ") " << r.v << " = " << (is_and ? "false" : "true") << ";\n";
            os << ind(d + 2) << "else { " << r.n << " = " << l.n << "; " << r.v << " = " << (is_and ? "true" : "false") << "; }\n";
            os << ind(d + 1) << "}\n";
            os << ind(d) << "}\n";
            return r;
        }
        case TGPU_SF_IF: {
            if (nd.n_args != 3) bad("IF takes three arguments");
            Val r = declare(nd.type, d);
            os << ind(d) << "{\n";
            Val c = gen(nd.args[0], d + 1);
            if (c.type != TGPU_BOOLEAN) bad("IF condition must be boolean");
            os << ind(d + 1) << "if (!" << c.n << " && " << c.v << ") {\n";
            Val t = gen(nd.args[1], d + 2);
            if (t.type != nd.type) bad("IF branch type");
            assign(r, t, d + 2);
            os << ind(d + 1) << "} else {\n";
            Val f = gen(nd.args[2], d + 2);
            if (f.type != nd.type) bad("IF branch type");
            assign(r, f, d + 2);
            os << ind(d + 1) << "}\n";
            os << ind(d) << "}\n";
            return r;
        }
        case TGPU_SF_IS_NULL: {
            if (nd.n_args != 1 || nd.type != TGPU_BOOLEAN) bad("IS_NULL takes one argument");
            Val r = declare(TGPU_BOOLEAN, d);
            os << ind(d) << "{\n";
            Val a = gen(nd.args[0], d + 1);
            os << ind(d + 1) << r.v << " = " << a.n << ";\n";
            os << ind(d) << "}\n";
            return r;
        }
        case TGPU_SF_COALESCE: {
            if (nd.n_args < 1) bad("COALESCE needs arguments");
            Val r = declare(nd.type, d);
            os << ind(d) << r.n << " = true;\n";
            int depth = d;
            for (int k = 0; k < nd.n_args; k++) {
                os << ind(depth) << "if (" << r.n << ") {\n";
                depth++;
                Val a = gen(nd.args[k], depth);
                if (a.type != nd.type) bad("COALESCE argument type");
                assign(r, a, depth);
            }
            for (int k = 0; k < nd.n_args; k++) {
                depth--;
                os << ind(depth) << "}\n";
            }
            return r;
        }
        case TGPU_SF_BETWEEN: {
            // value >= min AND value <= max with AND's three-valued logic; value / min are evaluated once
            if (nd.n_args != 3 || nd.type != TGPU_BOOLEAN) bad("BETWEEN takes three arguments");
            Val r = declare(TGPU_BOOLEAN, d);
            os << ind(d) << "{\n";
            Val v = gen(nd.args[0], d + 1);
            Val lo = gen(nd.args[1], d + 1);
            if (v.type != lo.type) bad("BETWEEN operand types");
            os << ind(d + 1) << "bool ln_ = " << v.n << " || " << lo.n << "; bool lv_ = false; if (!ln_) lv_ = " << cmp_expr(v.type, TGPU_OP_GREATER_THAN_OR_EQUAL, v, lo) << ";\n";
            os << ind(d + 1) << "if (!ln_ && !lv_) " << r.v << " = false; else {\n";
            Val hi = gen(nd.args[2], d + 2);
            if (v.type != hi.type) bad("BETWEEN operand types");
            os << ind(d + 2) << "bool rn_ = " << v.n << " || " << hi.n << "; bool rv_ = false; if (!rn_) rv_ = " << cmp_expr(v.type, TGPU_OP_LESS_THAN_OR_EQUAL, v, hi) << ";\n";
            os << ind(d + 2) << "if (rn_) " << r.n << " = true; else if (!rv_) " << r.v << " = false; else { " << r.n << " = ln_; " << r.v << " = true; }\n";
            os << ind(d + 1) << "}\n";
            os << ind(d) << "}\n";
            return r;
        }
        default: bad("unknown special form");
        }
    }

    void assign(const Val &dst, const Val &src, int d)
    {
        os << ind(d) << dst.n << " = " << src.n << "; " << dst.v << " = " << src.v << ";";
        if (dst.type == TGPU_VARCHAR) os << " " << dst.len << " = " << src.len << ";";
        os << "\n";
    }
};

const char *kPrelude = R"SRC(
// generated by libtgpu (jit.cpp): fused filter + project kernels for gfx950
#define TG_E_RANGE 2
#define TG_E_DIV0 7
#define TG_MAXC 24
#define TG_MAXP 16
struct FpArgs {
  const void* col_values[TG_MAXC];
  const unsigned char* col_nulls[TG_MAXC];
  const int* col_offsets[TG_MAXC];
  void* out_values[TG_MAXP];
  unsigned char* out_nulls[TG_MAXP];
  int* positions;
  const int* tile_offsets;
  int* tile_counts;
  unsigned long long* error;
  long long n;
};
// first failing row wins (the reference throws at the first failing position)
__device__ inline void tg_error(unsigned long long* e, long long row, int code) {
  atomicMin(e, ((unsigned long long)row << 8) | (unsigned long long)code);
}
__device__ inline int tg_strcmp(const unsigned char* a, int la, const unsigned char* b, int lb) {
  int m = la < lb ? la : lb;
  for (int i = 0; i < m; i++) { int d = (int)a[i] - (int)b[i]; if (d) return d; }
  return la - lb;
}
#define TG_TILE 1024
#define TG_STRIPES 4
)SRC";

}  // namespace

PageProcessorGpu::PageProcessorGpu(std::vector<int32_t> input_types, const tgpu_page_processor_spec *spec) : input_types_(std::move(input_types))
{
    TG_CHECK_ARG(spec != nullptr, "page processor spec is null");
    TG_CHECK_ARG(spec->node_count >= 0 && (spec->node_count == 0 || spec->nodes != nullptr), "bad node array");
    nodes_.assign(spec->nodes, spec->nodes + spec->node_count);
    if (spec->string_pool && spec->string_pool_len > 0) pool_.assign(spec->string_pool, spec->string_pool + spec->string_pool_len);
    filter_root_ = spec->filter_root;
    TG_CHECK_ARG(spec->projection_count >= 0 && spec->projection_count <= 64, "bad projection count");
    proj_roots_.assign(spec->projection_roots, spec->projection_roots + spec->projection_count);
    for (int32_t t : input_types_) TG_CHECK_ARG(valid_type(t), "unknown input type");
    generate();
}

PageProcessorGpu::~PageProcessorGpu() {}

void PageProcessorGpu::generate()
{
    Gen g(nodes_, pool_, input_types_);
    // classify projections: identity (InputPageProjection) needs no codegen (PageFunctionCompiler.java:176-186)
    for (int32_t root : proj_roots_) {
        const tgpu_expr_node &nd = g.node(root);
        Proj p;
        p.type = nd.type;
        if (nd.kind == TGPU_EX_INPUT) {
            if (nd.op < 0 || nd.op >= (int)input_types_.size() || input_types_[nd.op] != nd.type) g.bad("identity projection channel/type mismatch");
            p.kind = ProjKind::IDENTITY;
            p.channel = nd.op;
        }
        else {
            if (nd.type == TGPU_VARCHAR) g.bad("computed VARCHAR projections are not supported");
            p.kind = ProjKind::COMPUTED;
            p.slot = computed_count_++;
            if (computed_count_ > kFpMaxProj) g.bad("too many computed projections");
        }
        projs_.push_back(p);
        output_types_.push_back(nd.type);
    }

    std::ostringstream filter_body, proj_body;
    if (filter_root_ >= 0) {
        Val f = g.gen(filter_root_, 2);
        if (f.type != TGPU_BOOLEAN) g.bad("filter must be boolean");
        g.os << "    return !" << f.n << " && " << f.v << ";\n";
        filter_body << g.os.str();
        g.os.str("");
    }
    for (size_t i = 0; i < projs_.size(); i++) {
        if (projs_[i].kind != ProjKind::COMPUTED) continue;
        g.os << "    {\n";
        Val v = g.gen(proj_roots_[i], 3);
        const int slot = projs_[i].slot;
        const char *T = projs_[i].type == TGPU_BOOLEAN ? "unsigned char" : ctype(projs_[i].type);
        g.os << "      ((" << T << "*)A.out_values[" << slot << "])[o] = " << v.n << " ? (" << T << ")0 : (" << T << ")" << v.v << ";\n";
        g.os << "      A.out_nulls[" << slot << "][o] = " << v.n << " ? 1 : 0;\n";
        g.os << "    }\n";
    }
    proj_body << g.os.str();

    std::ostringstream cols;
    for (int ch : g.used_cols) {
        const int32_t t = input_types_[ch];
        const char *T = t == TGPU_VARCHAR ? "unsigned char" : (t == TGPU_BOOLEAN ? "unsigned char" : ctype(t));
        cols << "  const " << T << "* c" << ch << " = (const " << T << "*)A.col_values[" << ch << "];\n";
        cols << "  const unsigned char* cn" << ch << " = A.col_nulls[" << ch << "];\n";
        if (t == TGPU_VARCHAR) cols << "  const int* co" << ch << " = A.col_offsets[" << ch << "];\n";
        cols << "  (void)c" << ch << "; (void)cn" << ch << ";\n";
    }

    std::ostringstream src;
    src << kPrelude << g.consts.str();
    const bool has_filter = filter_root_ >= 0;
    if (has_filter) {
        src << "__device__ inline bool tg_filter(const FpArgs& A, long long row) {\n" << cols.str() << filter_body.str() << "}\n";
        // pass 1: selected rows per tile
        src << R"SRC(
extern "C" __global__ void __launch_bounds__(256) fp_count(FpArgs A) {
  const long long tile_base = (long long)blockIdx.x * TG_TILE;
  int cnt = 0;
#pragma unroll
  for (int s = 0; s < TG_STRIPES; s++) {
    long long row = tile_base + s * 256 + threadIdx.x;
    if (row < A.n && tg_filter(A, row)) cnt++;
  }
  __shared__ int wsum[4];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_down(cnt, d, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) A.tile_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
)SRC";
    }
    src << "__device__ inline void tg_project(const FpArgs& A, long long row, long long o) {\n" << cols.str() << proj_body.str() << "  (void)row; (void)o;\n}\n";
    if (has_filter) {
        // pass 2: ballot + prefix compaction inside the tile, tile offsets from the scan of pass 1
        src << R"SRC(
extern "C" __global__ void __launch_bounds__(256) fp_emit(FpArgs A) {
  const long long tile_base = (long long)blockIdx.x * TG_TILE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __shared__ int C[4][TG_STRIPES];
  bool sel[TG_STRIPES];
  unsigned long long b[TG_STRIPES];
#pragma unroll
  for (int s = 0; s < TG_STRIPES; s++) {
    long long row = tile_base + s * 256 + threadIdx.x;
    sel[s] = row < A.n && tg_filter(A, row);
    b[s] = __ballot(sel[s]);
    if (lane == 0) C[w][s] = __popcll(b[s]);
  }
  __syncthreads();
  long long base = A.tile_offsets[blockIdx.x];
#pragma unroll
  for (int s = 0; s < TG_STRIPES; s++) {
    int before = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < 4; w2++) { int c = C[w2][s]; if (w2 < w) before += c; total += c; }
    if (sel[s]) {
      long long row = tile_base + s * 256 + threadIdx.x;
      long long o = base + before + __builtin_amdgcn_mbcnt_hi((unsigned)(b[s] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b[s], 0u));
      A.positions[o] = (int)row;
      tg_project(A, row, o);
    }
    base += total;
  }
}
)SRC";
    }
    else {
        src << R"SRC(
extern "C" __global__ void __launch_bounds__(256) fp_emit(FpArgs A) {
  const long long tile_base = (long long)blockIdx.x * TG_TILE;
#pragma unroll
  for (int s = 0; s < TG_STRIPES; s++) {
    long long row = tile_base + s * 256 + threadIdx.x;
    if (row < A.n) tg_project(A, row, row);
  }
}
)SRC";
    }
    source_ = src.str();
}

void PageProcessorGpu::precompile() { (void)code_object_for(source_); }

void PageProcessorGpu::ensure_loaded(Context *ctx)
{
    (void)ctx;
    if (module_) return;
    std::vector<char> code = code_object_for(source_);
    auto m = std::make_shared<JitModule>();
    hipError_t e = hipModuleLoadData(&m->mod, code.data());
    if (e != hipSuccess) fail(TGPU_ERR_COMPILER, std::string("hipModuleLoadData failed: ") + hipGetErrorString(e));
    if (filter_root_ >= 0) {
        e = hipModuleGetFunction(&m->count, m->mod, "fp_count");
        if (e != hipSuccess) fail(TGPU_ERR_COMPILER, "generated module lacks fp_count");
    }
    e = hipModuleGetFunction(&m->emit, m->mod, "fp_emit");
    if (e != hipSuccess) fail(TGPU_ERR_COMPILER, "generated module lacks fp_emit");
    module_ = m;
}

static void launch(hipFunction_t f, int grid, FpArgs &args, hipStream_t stream)
{
    size_t size = sizeof(FpArgs);
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    HIP_CHECK(hipModuleLaunchKernel(f, (unsigned)grid, 1, 1, 256, 1, 1, 0, stream, nullptr, config));
}

bool PageProcessorGpu::process(Context *ctx, const DevicePage &in, DevicePage &out)
{
    TG_CHECK_ARG(in.cols.size() == input_types_.size(), "page channel count differs from the operator's input types");
    for (size_t i = 0; i < in.cols.size(); i++) TG_CHECK_ARG(in.cols[i].type == input_types_[i], "page channel type differs from the operator's input types");
    out.cols.clear();
    out.n = 0;
    const int64_t n = in.n;
    if (n == 0) return false;  // PageProcessor.java:116-118
    ensure_loaded(ctx);

    FpArgs args{};
    for (size_t i = 0; i < in.cols.size() && i < (size_t)kFpMaxCols; i++) {
        args.col_values[i] = in.cols[i].values;
        args.col_nulls[i] = in.cols[i].nulls;
        args.col_offsets[i] = in.cols[i].offsets;
    }
    args.n = n;
    const int64_t tiles = ceil_div(n, 1024);
    TG_CHECK_ARG(tiles <= 0x7fffffffLL, "page too large");
    BufferPtr err = ctx->alloc(8);
    HIP_CHECK(hipMemsetAsync(err->ptr(), 0xff, 8, ctx->stream()));
    args.error = err->as<unsigned long long>();

    auto check_error = [&]() {
        unsigned long long e = ctx->read_scalar(err->as<unsigned long long>());
        if (e == ~0ull) return;
        const long long row = (long long)(e >> 8);
        const int code = (int)(e & 0xff);
        if (code == 7) fail(TGPU_ERR_DIVISION_BY_ZERO, "Division by zero (position " + std::to_string(row) + ")");
        fail(TGPU_ERR_NUMERIC_VALUE_OUT_OF_RANGE, "numeric value out of range: arithmetic overflow (position " + std::to_string(row) + ")");
    };

    int64_t n_sel = n;
    BufferPtr positions, tile_counts, tile_offsets;
    const bool has_filter = filter_root_ >= 0;
    if (has_filter) {
        tile_counts = ctx->alloc((size_t)tiles * 4);
        tile_offsets = ctx->alloc((size_t)tiles * 4);
        BufferPtr total = ctx->alloc(8);
        args.tile_counts = tile_counts->as<int32_t>();
        {
            ProfileScope ps(ctx, "filter_count");
            launch(module_->count, (int)tiles, args, ctx->stream());
        }
        k::exclusive_scan_i32(ctx, tile_counts->as<int32_t>(), tile_offsets->as<int32_t>(), tiles, total->as<int64_t>());
        n_sel = ctx->read_scalar(total->as<int64_t>());
        check_error();  // filter errors surface before any projection runs
        if (n_sel == 0) return false;  // PageProcessor.java:122-124
        positions = ctx->alloc((size_t)n_sel * 4);
        args.positions = positions->as<int32_t>();
        args.tile_offsets = tile_offsets->as<int32_t>();
    }
    // outputs of computed projections
    std::vector<DeviceColumn> computed((size_t)computed_count_);
    for (auto &p : projs_) {
        if (p.kind != ProjKind::COMPUTED) continue;
        DeviceColumn c;
        c.type = p.type;
        c.n = n_sel;
        c.values_buf = ctx->alloc((size_t)n_sel * type_width(p.type));
        c.values = c.values_buf->ptr();
        c.nulls_buf = ctx->alloc((size_t)n_sel);
        c.nulls = c.nulls_buf->as<uint8_t>();
        args.out_values[p.slot] = c.values_buf->ptr();
        args.out_nulls[p.slot] = c.nulls_buf->as<uint8_t>();
        computed[(size_t)p.slot] = c;
    }
    if (has_filter || computed_count_ > 0) {
        ProfileScope ps(ctx, has_filter ? "filter_project_emit" : "project_emit");
        launch(module_->emit, (int)tiles, args, ctx->stream());
    }
    if (computed_count_ > 0) check_error();
    out.n = n_sel;
    for (auto &p : projs_) {
        if (p.kind == ProjKind::COMPUTED) out.cols.push_back(computed[(size_t)p.slot]);
        else if (!has_filter || n_sel == n) out.cols.push_back(in.cols[(size_t)p.channel]);  // all rows selected: pass the block through
        else out.cols.push_back(k::gather_column(ctx, in.cols[(size_t)p.channel], positions->as<int32_t>(), n_sel, false));
    }
    return true;
}

}  // namespace tgpu
