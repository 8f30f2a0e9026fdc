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

#include <algorithm>
#include <fstream>
#include <functional>
#include <list>
#include <tuple>
#include <set>
#include <sstream>

#include "agg.h"
#include "groupby.h"
#include "join.h"
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

// -ffp-contract=off: the JVM never fuses a multiply into an add; a projection's product feeding an accumulator's `sum += v` must be
// rounded first, like DoubleSumAggregation.java:34-38 sees it (hipcc's default, fp-contract=fast, would emit an FMA there)
static const char *kCompileOptions[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};
constexpr int kCompileOptionCount = (int)(sizeof(kCompileOptions) / sizeof(kCompileOptions[0]));

static std::vector<char> compile_source(const std::string &source)
{
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, source.c_str(), "tgpu_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        fail(TGPU_ERR_COMPILER, "hiprtcCreateProgram failed");
    hiprtcResult r = hiprtcCompileProgram(prog, kCompileOptionCount, kCompileOptions);
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

// What a cached code object depends on besides the source text: the compiler (hiprtc version), its options and the target.
static const std::string &toolchain_tag()
{
    static const std::string tag = [] {
        int major = 0, minor = 0;
        hiprtcVersion(&major, &minor);
        std::string t = "hiprtc " + std::to_string(major) + "." + std::to_string(minor);
        for (const char *o : kCompileOptions) t += std::string(" ") + o;
        return t;
    }();
    return tag;
}

static uint64_t fnv1a_seeded(const std::string &s, uint64_t h)
{
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ULL; }
    return h;
}

static std::string cache_path(const std::string &source)
{
    char name[64];
    snprintf(name, sizeof(name), "%016llx.hsaco", (unsigned long long)fnv1a(toolchain_tag() + "\n" + source));
    return resource_dir() + "/_kcache/" + name;
}

// A cache file = header {magic, a second independent hash of (toolchain, source), source length} + the code object: a file that was
// written for another source with the same 64-bit name, by another toolchain, or cut short, is recompiled instead of loaded.
struct CacheHeader {
    char magic[8];
    uint64_t check, source_len, code_len;
};
static CacheHeader header_for(const std::string &source, size_t code_len)
{
    CacheHeader h{};
    memcpy(h.magic, "TGPUJIT2", 8);
    h.check = fnv1a_seeded(source + "\n" + toolchain_tag(), 0x9E3779B97F4A7C15ULL);
    h.source_len = source.size();
    h.code_len = code_len;
    return h;
}

// compile (or fetch from the disk cache) the code object of `source`
static std::vector<char> code_object_for(const std::string &source)
{
    const std::string path = cache_path(source);
    if (getenv("TGPU_JIT_DUMP")) {  // keep the generated source next to its code object (kernel studies)
        mkdir((resource_dir() + "/_kcache").c_str(), 0755);
        std::ofstream f(path.substr(0, path.size() - 6) + ".hip");
        f << source;
    }
    {
        std::ifstream f(path, std::ios::binary);
        if (f) {
            std::vector<char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            if (file.size() > sizeof(CacheHeader)) {
                CacheHeader got, want = header_for(source, file.size() - sizeof(CacheHeader));
                memcpy(&got, file.data(), sizeof(got));
                if (memcmp(&got, &want, sizeof(got)) == 0) return std::vector<char>(file.begin() + sizeof(CacheHeader), file.end());
            }
        }
    }
    std::vector<char> code = compile_source(source);
    mkdir((resource_dir() + "/_kcache").c_str(), 0755);
    const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
    {
        std::ofstream f(tmp, std::ios::binary);
        if (f) {
            const CacheHeader h = header_for(source, code.size());
            f.write(reinterpret_cast<const char *>(&h), sizeof(h));
            f.write(code.data(), (std::streamsize)code.size());
            f.close();
            rename(tmp.c_str(), path.c_str());  // atomic publish; a read-only tree just skips the cache
        }
    }
    return code;
}

// A process-wide cache bounded like the reference's compiled-class caches (PageFunctionCompiler's expression caches hold a
// configured maximum, M/sql/gen/PageFunctionCompiler.java:101-139): least recently used entries are dropped; whoever still
// uses an evicted object keeps it alive through its shared_ptr.
template <typename V> class LruCache {
public:
    explicit LruCache(size_t capacity) : capacity_(capacity) {}
    template <typename Make> std::shared_ptr<V> get(const std::string &key, Make make)
    {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = map_.find(key);
        if (it != map_.end()) {
            order_.splice(order_.begin(), order_, it->second.second);
            return it->second.first;
        }
        std::shared_ptr<V> obj = make();
        order_.push_front(key);
        map_[key] = {obj, order_.begin()};
        while (map_.size() > capacity_) {
            map_.erase(order_.back());
            order_.pop_back();
        }
        return obj;
    }
    size_t size()
    {
        std::lock_guard<std::mutex> lk(mu_);
        return map_.size();
    }

private:
    size_t capacity_;
    std::mutex mu_;
    std::list<std::string> order_;
    std::map<std::string, std::pair<std::shared_ptr<V>, std::list<std::string>::iterator>> map_;
};
static size_t jit_cache_capacity()
{
    const char *v = getenv("TGPU_JIT_CACHE_ENTRIES");
    const long n = v ? atol(v) : 0;
    return n > 0 ? (size_t)n : 1000;   // the reference's expression-cache-size default is 1000 entries
}

struct JitModule {
    hipModule_t mod = nullptr;
    hipFunction_t count = nullptr, emit = nullptr;
    std::map<std::string, hipFunction_t> fns;
    ~JitModule()
    {
        if (mod) hipModuleUnload(mod);
    }
    std::mutex mu;   // modules are shared between operators (load_module), possibly on different driver threads
    std::map<std::string, int> resident;
    // workgroups of 256 threads of `name` that fit on one CU (registers / LDS): the grid of a persistent kernel
    int blocks_per_cu(const char *name)
    {
        hipFunction_t f = fn(name);
        std::lock_guard<std::mutex> lk(mu);
        auto it = resident.find(name);
        if (it != resident.end()) return it->second;
        int nb = 0;
        if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, 256, 0) != hipSuccess || nb < 1) nb = 1;
        resident[name] = nb;
        return nb;
    }
    hipFunction_t fn(const char *name)
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = fns.find(name);
        if (it != fns.end()) return it->second;
        hipFunction_t f = nullptr;
        if (hipModuleGetFunction(&f, mod, name) != hipSuccess) fail(TGPU_ERR_COMPILER, std::string("generated module lacks ") + name);
        fns[name] = f;
        return f;
    }
};

// Loaded modules are shared process-wide per device, keyed by their source: operator factories are created per query, the
// kernels they generate repeat (the counterpart of the reference's compiled-class caches, M/sql/gen/PageFunctionCompiler.java
// :101-139 and JoinCompiler.java:89-107).
static std::shared_ptr<JitModule> load_module(const std::string &source)
{
    static LruCache<JitModule> loaded(jit_cache_capacity());
    int device = 0;
    HIP_CHECK(hipGetDevice(&device));   // the calling thread is bound to the context's device (c_api.cpp bind_thread)
    const std::string key = std::to_string(device) + ":" + std::to_string(source.size()) + ":" + cache_path(source);
    return loaded.get(key, [&] {
        std::vector<char> code = code_object_for(source);
        auto m = std::make_shared<JitModule>();
        hipError_t e = hipModuleLoadData(&m->mod, code.data());
        if (e != hipSuccess) fail(TGPU_ERR_COMPILER, std::string("hipModuleLoadData failed: ") + hipGetErrorString(e));
        return m;
    });
}

// text of a device header shipped next to the library (csrc/), with its #pragma once removed, for embedding in JIT sources
static std::string device_header(const char *name)
{
    const std::string path = resource_dir() + "/csrc/" + name;
    std::ifstream f(path);
    if (!f) fail(TGPU_ERR_COMPILER, "cannot read " + path + " (set the resource dir with tgpu_set_resource_dir)");
    std::stringstream ss;
    ss << f.rdbuf();
    std::string t = ss.str();
    size_t p = t.find("#pragma once");
    if (p != std::string::npos) t.erase(p, 12);
    return t;
}

// p[0 .. n_ones) = ~0 (error words: "none"), p[n_ones .. n_ones + n_zeros) = 0 (counters): one launch instead of two memsets
static __global__ void init_words_kernel(unsigned long long *p, int n_ones, int n_zeros)
{
    for (int i = (int)threadIdx.x; i < n_ones + n_zeros; i += (int)blockDim.x) p[i] = i < n_ones ? ~0ull : 0ull;
}

template <typename Args> static void launch_args(hipFunction_t f, int grid, Args &args, hipStream_t stream, int block = 256)
{
    size_t size = sizeof(Args);
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    HIP_CHECK(hipModuleLaunchKernel(f, (unsigned)grid, 1, 1, (unsigned)block, 1, 1, 0, stream, nullptr, config));
}


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
    // register-row mode: fixed-width INPUT references read a pre-loaded row struct `R` (fields c<ch>, n<ch>) instead of
    // memory, so that the kernel can issue the loads of the NEXT tile before it evaluates the current one
    bool reg_mode = false;
    std::set<int> reg_cols;

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
            Val r = declare(nd.type, d);
            if (reg_mode && nd.type != TGPU_VARCHAR) {
                reg_cols.insert(ch);
                os << ind(d) << r.n << " = R.n" << ch << " != 0;\n";
                os << ind(d) << "if (!" << r.n << ") " << r.v << " = R.c" << ch << (nd.type == TGPU_BOOLEAN ? " != 0" : "") << ";\n";
                return r;
            }
            used_cols.insert(ch);
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
            // castToBoolean: M/type/BigintOperators.java:115-120, IntegerOperators.java:146-151, DoubleOperators.java:101-106 (NaN -> true)
            else if (nd.type == TGPU_BOOLEAN && (at == TGPU_BIGINT || at == TGPU_INTEGER || at == TGPU_DOUBLE)) os << ind(dd) << r.v << " = " << a.v << " != 0;\n";
            // M/type/BooleanOperators.java:37-63
            else if (at == TGPU_BOOLEAN && (nd.type == TGPU_BIGINT || nd.type == TGPU_INTEGER)) os << ind(dd) << r.v << " = " << a.v << " ? 1 : 0;\n";
            else if (at == TGPU_BOOLEAN && nd.type == TGPU_DOUBLE) os << ind(dd) << r.v << " = " << a.v << " ? 1.0 : 0.0;\n";
            else if (nd.type == TGPU_BIGINT && at == TGPU_DOUBLE) {
                // DoubleOperators.castToLong :153-163 (DoubleMath.roundToLong HALF_UP): ties away from zero; NaN / inf / out of long range -> INVALID_CAST_ARGUMENT
                os << ind(dd) << "double z_ = tg_round_half_away(" << a.v << ");\n";
                os << ind(dd) << "if (!(z_ >= -9223372036854775808.0 && z_ < 9223372036854775808.0)) { tg_error(A.error, row, TG_E_CAST); " << r.n << " = true; } else " << r.v << " = (long long)z_;\n";
            }
            else if (nd.type == TGPU_INTEGER && at == TGPU_DOUBLE) {
                // DoubleOperators.castToInteger :108-121: NaN -> INVALID_CAST_ARGUMENT, else toIntExact((long) round(value))
                os << ind(dd) << "double z_ = tg_round_half_away(" << a.v << ");\n";
                os << ind(dd) << "if (" << a.v << " != " << a.v << ") { tg_error(A.error, row, TG_E_CAST); " << r.n << " = true; }\n";
                os << ind(dd) << "else if (!(z_ >= -2147483648.0 && z_ <= 2147483647.0)) { tg_error(A.error, row, TG_E_RANGE); " << r.n << " = true; } else " << r.v << " = (int)z_;\n";
            }
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
#define TG_E_CAST 9
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
  unsigned long long* sel_mask;   // filter verdicts of pass 1, one 64-bit ballot per (tile, stripe, wave)
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
// nearest integer, ties away from zero (java Math.round mirrored around zero = guava HALF_UP); NaN / inf pass through
__device__ inline double tg_round_half_away(double x) {
  if (!(fabs(x) < 4503599627370496.0)) return x;
  const double t = trunc(x);
  return (fabs(x - t) >= 0.5) ? t + (x < 0 ? -1.0 : 1.0) : t;
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
    // which input channels do the computed expressions read?
    std::set<int> read;
    std::function<void(int)> walk = [&](int idx) {
        if (idx < 0 || idx >= (int)nodes_.size()) return;
        const tgpu_expr_node &nd = nodes_[(size_t)idx];
        if (nd.kind == TGPU_EX_INPUT) read.insert(nd.op);
        if (nd.kind == TGPU_EX_CALL || nd.kind == TGPU_EX_SPECIAL)
            for (int k = 0; k < nd.n_args && k < 3; k++) walk(nd.args[k]);
    };
    if (filter_root_ >= 0) walk(filter_root_);
    filter_channels_.assign(read.begin(), read.end());
    for (size_t i = 0; i < projs_.size(); i++)
        if (projs_[i].kind == ProjKind::COMPUTED) walk(proj_roots_[i]);
    single_input_ = read.size() == 1 ? *read.begin() : -1;
    std::set<int> pread;
    std::swap(read, pread);
    for (size_t i = 0; i < projs_.size(); i++) walk(proj_roots_[i]);
    projection_channels_.assign(read.begin(), read.end());
}

std::shared_ptr<PageProcessorGpu> PageProcessorGpu::filter_only()
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!filter_only_) {
        tgpu_page_processor_spec spec{nodes_.data(), (int32_t)nodes_.size(), pool_.data(), (int32_t)pool_.size(), filter_root_, 0, nullptr};
        filter_only_ = PageProcessorGpu::shared(input_types_, &spec);
    }
    return filter_only_;
}

PageProcessorGpu::~PageProcessorGpu() {}

void PageProcessorGpu::process_dictionary(Context *ctx, const DeviceColumn &dictionary, DevicePage &out)
{
    TG_CHECK_STATE(single_input_ >= 0, "the processor's expressions do not read a single channel");
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (!dict_processor_) {
            // the same node array with every input reference pointing at channel 0; no filter; projections = [filter, computed...]
            std::vector<tgpu_expr_node> nodes = nodes_;
            for (auto &nd : nodes)
                if (nd.kind == TGPU_EX_INPUT) nd.op = 0;
            std::vector<int32_t> roots;
            if (filter_root_ >= 0) roots.push_back(filter_root_);
            std::vector<int32_t> by_slot((size_t)computed_count_, -1);
            for (size_t i = 0; i < projs_.size(); i++)
                if (projs_[i].kind == ProjKind::COMPUTED) by_slot[(size_t)projs_[i].slot] = proj_roots_[i];
            for (int32_t r : by_slot) roots.push_back(r);
            tgpu_page_processor_spec spec{nodes.data(), (int32_t)nodes.size(), pool_.data(), (int32_t)pool_.size(), -1, (int32_t)roots.size(), roots.data()};
            dict_processor_ = PageProcessorGpu::shared({input_types_[(size_t)single_input_]}, &spec);
        }
    }
    DevicePage in;
    in.n = dictionary.n;
    in.cols.push_back(dictionary);
    dict_processor_->process(ctx, in, out);
}

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
    const bool sel = row < A.n && tg_filter(A, row);
    if (sel) cnt++;
    // the verdicts are kept (1 bit per row) so that pass 2 neither re-reads the filter's columns nor re-evaluates it
    const unsigned long long b = __ballot(sel);
    if ((threadIdx.x & 63) == 0) A.sel_mask[((long long)blockIdx.x * TG_STRIPES + s) * 4 + (threadIdx.x >> 6)] = b;
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
    b[s] = A.sel_mask[((long long)blockIdx.x * TG_STRIPES + s) * 4 + w];   // pass 1's verdicts (wave-uniform load)
    sel[s] = (b[s] >> lane) & 1ULL;
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

// ---- process-wide cache of the generated objects ----------------------------------------------------------------------
namespace {
void put_i32(std::string &k, int32_t v) { k.append(reinterpret_cast<const char *>(&v), 4); }
void put_i64(std::string &k, int64_t v) { k.append(reinterpret_cast<const char *>(&v), 8); }

// every field the generators read (field by field: the structs' padding is the caller's business), plus the kernel-study
// environment switches that change the generated source
std::string spec_key(const char *what, const std::vector<int32_t> &input_types, const tgpu_page_processor_spec *spec, const std::vector<int32_t> &extra)
{
    TG_CHECK_ARG(spec != nullptr, "page processor spec is null");
    std::string k(what);
    put_i32(k, (int32_t)input_types.size());
    for (int32_t t : input_types) put_i32(k, t);
    put_i32(k, spec->node_count);
    for (int32_t i = 0; i < spec->node_count; i++) {
        const tgpu_expr_node &n = spec->nodes[i];
        put_i32(k, n.kind); put_i32(k, n.type); put_i32(k, n.op); put_i32(k, n.n_args);
        put_i32(k, n.args[0]); put_i32(k, n.args[1]); put_i32(k, n.args[2]); put_i32(k, n.is_null);
        put_i64(k, n.ival);
        k.append(reinterpret_cast<const char *>(&n.dval), 8);
        put_i32(k, n.slen);
    }
    put_i32(k, spec->string_pool_len);
    if (spec->string_pool && spec->string_pool_len > 0) k.append(spec->string_pool, (size_t)spec->string_pool_len);
    put_i32(k, spec->filter_root);
    put_i32(k, spec->projection_count);
    for (int32_t i = 0; i < spec->projection_count; i++) put_i32(k, spec->projection_roots[i]);
    put_i32(k, (int32_t)extra.size());
    for (int32_t v : extra) put_i32(k, v);
    for (const char *env : {"TGPU_FG_EXP", "TGPU_FJ_EXP", "TGPU_FA_STRIPES", "TGPU_FJ_STRIPES", "TGPU_FG_STRIPES"}) {
        const char *v = getenv(env);
        k += '|';
        if (v) k += v;
    }
    return k;
}

template <typename T, typename Make> std::shared_ptr<T> cached_object(const std::string &key, Make make)
{
    static LruCache<T> objects(jit_cache_capacity());
    return objects.get(key, make);
}
}  // namespace

std::shared_ptr<PageProcessorGpu> PageProcessorGpu::shared(const std::vector<int32_t> &input_types, const tgpu_page_processor_spec *spec)
{
    return cached_object<PageProcessorGpu>(spec_key("pp", input_types, spec, {}), [&] { return std::make_shared<PageProcessorGpu>(input_types, spec); });
}

std::shared_ptr<FusedProbeGpu> FusedProbeGpu::shared(const std::vector<int32_t> &input_types, const tgpu_page_processor_spec *spec, int32_t join_channel,
                                                     const std::vector<int32_t> &output_channels)
{
    std::vector<int32_t> extra{join_channel};
    extra.insert(extra.end(), output_channels.begin(), output_channels.end());
    return cached_object<FusedProbeGpu>(spec_key("fj", input_types, spec, extra),
                                        [&] { return std::make_shared<FusedProbeGpu>(input_types, spec, join_channel, output_channels); });
}

std::shared_ptr<FusedAggGpu> FusedAggGpu::shared(const std::vector<int32_t> &input_types, const tgpu_page_processor_spec *spec,
                                                 const std::vector<tgpu_agg_spec> &aggs, const std::vector<int32_t> &group_by_channels)
{
    std::vector<int32_t> extra{(int32_t)aggs.size()};
    for (const tgpu_agg_spec &a : aggs) {
        extra.push_back(a.function);
        extra.push_back(a.input_channel);
        extra.push_back(a.mask_channel);
    }
    extra.insert(extra.end(), group_by_channels.begin(), group_by_channels.end());
    return cached_object<FusedAggGpu>(spec_key("fa", input_types, spec, extra),
                                      [&] { return std::make_shared<FusedAggGpu>(input_types, spec, aggs, group_by_channels); });
}

void PageProcessorGpu::ensure_loaded(Context *ctx)
{
    (void)ctx;
    std::lock_guard<std::mutex> lk(mu_);
    if (module_) return;
    module_ = load_module(source_);
    if (filter_root_ >= 0) fn_count_ = module_->fn("fp_count");
    fn_emit_ = module_->fn("fp_emit");
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

    auto raise_error = [&](unsigned long long e) { raise_expression_error(e); };
    auto check_error = [&]() { raise_error(ctx->read_scalar(err->as<unsigned long long>())); };

    int64_t n_sel = n;
    BufferPtr positions, tile_counts, tile_offsets, sel_mask;
    const bool has_filter = filter_root_ >= 0;
    if (has_filter) {
        sel_mask = ctx->alloc((size_t)tiles * 4 * 4 * 8);   // TG_STRIPES (4) x 4 waves ballots per tile
        args.sel_mask = sel_mask->as<unsigned long long>();
        tile_counts = ctx->alloc((size_t)tiles * 4);
        tile_offsets = ctx->alloc((size_t)tiles * 4);
        BufferPtr total = ctx->alloc(8);
        args.tile_counts = tile_counts->as<int32_t>();
        {
            ProfileScope ps(ctx, "filter_count");
            launch(fn_count_, (int)tiles, args, ctx->stream());
        }
        k::exclusive_scan_i32(ctx, tile_counts->as<int32_t>(), tile_offsets->as<int32_t>(), tiles, total->as<int64_t>());
        // the selected-row count and the filter's error word come back in one round trip
        unsigned long long filter_error = ~0ull;
        ctx->download_batch({{&n_sel, total->ptr(), 8}, {&filter_error, err->ptr(), 8}});
        raise_error(filter_error);  // filter errors surface before any projection runs
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
        launch(fn_emit_, (int)tiles, args, ctx->stream());
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

// =====================================================================================================================
// FusedProbeGpu: filter + project + hash-join probe in one kernel
// =====================================================================================================================
namespace {

// rows per lane and tile in the fused probe kernel
static int fj_stripes()
{
    const char *e = getenv("TGPU_FJ_STRIPES");
    // 3 rows per lane and tile: measured best on the streaming (lineitem) launch of Q3 with the DIRECT layout -- 1.48-1.55 ms against
    // 1.57-1.62 at 4, 1.60 at 6, 1.67 at 8 and 1.88 at 2 (ABAB on one box): fewer registers = more resident waves with loads in flight
    const int v = e ? atoi(e) : 3;
    return v >= 1 && v <= 8 ? v : 3;
}

const char *kFjKernels = R"SRC(
struct FjArgs {
  FpArgs fp;
  const TgSlot16* slots;
  unsigned long long mask;
  TgPrefilter pf;
  int* tile_cnt;            // pairs produced by each CHUNK of tiles (see fj_probe)
  int* tile_src;            // where the chunk's pairs start inside its block's private region
  const int* tile_dst;      // pass 2: exclusive scan of tile_cnt = final output offset of the chunk
  int* pair_probe;          // block-private regions, capacity = rows the block owns
  int* pair_build;
  int* out_build;           // pass 2: build positions in final order
  unsigned long long* counters;
  long long tiles;
  long long grid1;          // grid size of pass 1 (defines the region layout)
  int outer;                // bit 0: PROBE_OUTER / FULL_OUTER rows; bit 1: nobody reads the build positions (no build output channels,
                            // no outer tracking): the DIRECT layout then skips the rank and position lookups
  int chunk_shift;          // a workgroup takes 2^chunk_shift consecutive tiles at a time
  struct { const void* values; const unsigned char* nulls; void* out_values; unsigned char* out_nulls; int width; int pad; } bcol[4];   // build-side
  int n_bcol;               // output channels pass 2 gathers itself (fixed width)
  int pad2;
  void* carry[4];           // FJ_CARRY: block-private regions of the probe-side output VALUES (pass 1 evaluates them from its row
  unsigned char* carry_nulls; // registers for every emitted pair; pass 2 moves them instead of re-reading the input columns sparsely)
  unsigned long long* host_out; // epilogue (small pages): host-visible result words, written by the LAST workgroup of pass 1 (null: none)
  unsigned int* done;           //                         workgroups that have finished (rests at 0)
  const FpArgs* pages;          // FJ_EPILOGUE == 2 (a launch over a LIST of pages): one FpArgs per page, in device memory
  const int* page_tile0;        //   first tile of each page in the launch's tile sequence, [n_pages] = all tiles
  int n_pages;
  int pad3;
};
#define FJ_STRIPES @FJ_STRIPES@
#define FJ_TILE (FJ_STRIPES * 256)
#define FJ_COUNT_SLOTS 64
#ifndef FJ_EPILOGUE
#define FJ_EPILOGUE 0   // the variant for pages (few, one-tile chunks): pass 1 ends with the scan and hands its totals to the host itself
#endif

// rows-capacity offset of block b's private pair region: CHUNKS of 2^chunk_shift consecutive tiles are dealt round-robin, block b
// owns ceil((chunks - b) / grid) of them
__device__ inline long long fj_region_base(long long b, long long tiles, long long grid, int chunk_shift) {
  const long long chunks = (tiles + (1LL << chunk_shift) - 1) >> chunk_shift;
  const long long q = chunks / grid, r = chunks % grid;
  return ((b * q + (b < r ? b : r)) << chunk_shift) * FJ_TILE;
}

@CARRY_FUNCS@
// pass 1: one lane per row.  Matches are compacted in input order inside the tile (ballot + popcount) and appended to the
// block's PRIVATE region, so workgroups never communicate; a scan over the per-tile counts (in tile = input order) then
// gives every tile its final offset and pass 2 moves the pairs there -- probe positions come out ascending exactly as
// LookupJoinPageBuilder.java:144-153 requires.
//
// A probe is a chain of three dependent loads (row -> pre-filter word -> table slot).  The loop is software-pipelined over
// the block's tiles so that the three kinds of load of one iteration belong to three different tiles and are issued back to
// back: iteration `it` issues the row loads of tile it, the pre-filter loads of tile it-1, the slot loads of tile it-2 and
// compacts / writes the pairs of tile it-3, whose slots arrived during the previous iteration.  One memory round trip per
// iteration instead of three (vmcnt is in-order on CDNA: a wait for a dependent load would also wait for every prefetch
// issued before it, so "prefetch, then probe" inside one iteration cannot overlap).
__device__ __forceinline__ void fj_probe_body(const FjArgs& J, const unsigned int bid, const unsigned int nblk) {
  const FpArgs& A = J.fp;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __shared__ int C[2][4];   // pairs of each wave, double-buffered by tile parity (one barrier per tile)
  const long long region = fj_region_base(bid, J.tiles, nblk, J.chunk_shift);
  long long local = 0;      // pairs this block has written so far (uniform across the block)
  unsigned int selected = 0;   // per lane: rows that passed the filter (a lane sees fewer than 2^31 rows)
  // Row layout of a tile: wave w owns the contiguous rows [w * 64 * FJ_STRIPES, (w + 1) * 64 * FJ_STRIPES) of the tile and its
  // lane reads row s * 64 + lane of them in stripe s (512 contiguous bytes per wave and load).  Input order inside the tile is
  // then (wave, stripe, lane): the compaction needs the other waves' TOTALS only, everything else is wave-local scalar work.
  // Rows are 32-bit here (a page has fewer than 2^31 positions).
#if FJ_EPILOGUE != 2
  const unsigned int n_rows = (unsigned int)A.n;
#endif
  const unsigned int wave_row = (unsigned int)w * (64u * FJ_STRIPES) + (unsigned int)lane;
  // The j-th tile this workgroup processes: chunks of 2^chunk_shift CONSECUTIVE tiles are dealt round-robin to the workgroups, so
  // the pairs a workgroup produces for one chunk are contiguous in its region AND in the final (input) order: pass 2 then moves
  // whole chunks (thousands of pairs, every lane busy) instead of single tiles (a handful of pairs per wave).
  const int csh = J.chunk_shift;
  const long long cmask = (1LL << csh) - 1;
  const long long chunks = (J.tiles + cmask) >> csh;
  const long long my_tiles = chunks > (long long)bid ? ((chunks - bid + nblk - 1) / nblk) << csh : 0;   // incl. tiles past the end
  auto tile_of = [&](long long j) -> long long { return ((((j >> csh) * nblk) + bid) << csh) + (j & cmask); };
#if FJ_EPILOGUE == 2
  // MULTI: the launch covers a list of pages taken as one sequence of tiles (a page starts a tile: J.page_tile0[p] = its first tile); rows
  // travel through the stages and into the pairs as VIRTUAL rows (tile * FJ_TILE + ...), loads and expressions use the page's own rows.
  // A workgroup's tiles ascend, so the page cursors of stage A / B only move forward (workgroup-uniform: scalar loads).
  const FpArgs* __restrict__ pgs = J.pages;
  const int* __restrict__ pt0 = J.page_tile0;
  // the descriptor of the page stage A will load from is fetched ONE ITERATION AHEAD (scalar loads: the search of the tile's page and the
  // page's column pointers are two dependent round trips that would otherwise sit in front of every iteration's row loads)
  int pN = 0;
  long long tN = 0;   // tile of the next iteration's stage A (my_tiles == 0: unused)
  {
    const long long t0 = tile_of(0);
    tN = (my_tiles > 0 && t0 < J.tiles) ? t0 : 0;
    while (tN >= (long long)pt0[pN + 1]) pN++;
  }
  FpArgs AN = pgs[pN];    // next stage A's page
  FpArgs AB = AN;         // this iteration's stage B page (= the previous iteration's stage A page)
  long long tB0 = 0;      // first tile of that page
  long long tN0 = pt0[pN];
#endif
  long long chunk_local0 = 0;   // `local` at the start of the chunk being compacted
  TgRow rw[FJ_STRIPES];                                                                                            // stage A -> B
  long long pkey[FJ_STRIPES]; unsigned int psidx[FJ_STRIPES]; unsigned long long pbw[FJ_STRIPES];                      // B -> C
#if FJ_PF == 2
  unsigned long long pbits[FJ_STRIPES];     // Bloom mask of the key
#else
  unsigned int pbits[FJ_STRIPES];           // bit of the key inside its bitmap word
#endif
  unsigned char pfl[FJ_STRIPES];                                                                                   // 1 = probe, 2 = passed the filter
#if FJ_PF == 1 || FJ_PF == 3
  const unsigned long long key_range = (unsigned long long)J.pf.key_max - (unsigned long long)J.pf.key_min;
#endif
  long long skey[FJ_STRIPES]; unsigned int ssidx[FJ_STRIPES]; unsigned char sfl[FJ_STRIPES]; TgSlot16 ssl[FJ_STRIPES]; // C -> D
#if FJ_CARRY
  TgOut pov[FJ_STRIPES], sov[FJ_STRIPES];   // the row's output values, travelling B -> C -> D with its key
#pragma unroll
  for (int s = 0; s < FJ_STRIPES; s++) { tg_zero_out(pov[s]); tg_zero_out(sov[s]); }
#endif
#pragma unroll
  for (int s = 0; s < FJ_STRIPES; s++) {
    tg_zero_row(rw[s]);
    pkey[s] = 0; psidx[s] = 0; pbits[s] = 0; pbw[s] = 0; pfl[s] = 0;
    skey[s] = 0; ssidx[s] = 0; sfl[s] = 0; ssl[s].key = 0; ssl[s].head = -1; ssl[s].count = 0;
  }
  for (long long it = 0; it < my_tiles + 3; it++) {
    const long long jD = it - 3, jC = it - 2, jB = it - 1, jA = it;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): everything the previous iteration issued (consumed right below anyway)
    // stage D, part 1: resolve tile jD from the slots loaded by the previous iteration; only a collision with another key
    // walks further
    int head[FJ_STRIPES];
    bool emit[FJ_STRIPES];
#pragma unroll
    for (int s = 0; s < FJ_STRIPES; s++) { head[s] = -1; emit[s] = false; }
    const bool doD = jD >= 0 && tile_of(jD) < J.tiles;
    if (doD) {
#pragma unroll
      for (int s = 0; s < FJ_STRIPES; s++) {
#if FJ_PF == 3
        // the bitmap is exact: the key is in the build side.  The pair carries the key's rank among the build keys (present keys
        // in front of its bitmap word, loaded by the previous iteration, + set bits below its own); fj_emit reads the build
        // position at that rank, for the matching rows only
        // (nobody reads the build positions: the rank structure may not even be built -- rank_base[0] is then whatever the buffer held)
        if (sfl[s] & 1) head[s] = (J.outer & 2) ? 0 : ssl[s].head + (int)ssidx[s];
#else
        if ((sfl[s] & 1) && ssl[s].head >= 0) {
          if (ssl[s].key == skey[s]) head[s] = ssl[s].head;
          else {
            unsigned long long pos = ((unsigned long long)ssidx[s] + 1) & J.mask;
            for (unsigned long long k = 0; k < J.mask; k++) {
              const TgSlot16 t = J.slots[pos];
              if (t.head < 0) break;
              if (t.key == skey[s]) { head[s] = t.head; break; }
              pos = (pos + 1) & J.mask;
            }
          }
        }
#endif
        emit[s] = head[s] >= 0 || ((J.outer & 1) && (sfl[s] & 2));   // PROBE_OUTER: every row that passed the filter (LookupJoinOperator.java:354-361)
      }
    }
#if FJ_CARRY
    // stage D, part 2 (carry variant: here, while the stage registers still hold tile jD's output values -- stage C below overwrites them)
    if (doD) {
      const long long tile = tile_of(jD);
      const unsigned int row0 = (unsigned int)(tile * FJ_TILE) + wave_row;
      if ((jD & cmask) == 0) chunk_local0 = local;
      unsigned long long b[FJ_STRIPES];
      int wave_total = 0;
#pragma unroll
      for (int s = 0; s < FJ_STRIPES; s++) {
        b[s] = __ballot(emit[s]);
        wave_total += __popcll(b[s]);
      }
      int* Cw = C[jD & 1];
      if (lane == 0) Cw[w] = wave_total;
      __syncthreads();
      int before = 0, tile_total = 0;
#pragma unroll
      for (int w2 = 0; w2 < 4; w2++) { const int c = Cw[w2]; if (w2 < w) before += c; tile_total += c; }
      int* pp = J.pair_probe + (region + local);   // uniform bases, 32-bit lane offsets
      int* pb = J.pair_build + (region + local);
      unsigned int o = (unsigned int)before;
#pragma unroll
      for (int s = 0; s < FJ_STRIPES; s++) {
        if (emit[s]) {
          const unsigned int at = o + __builtin_amdgcn_mbcnt_hi((unsigned)(b[s] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b[s], 0u));
          pp[at] = (int)(row0 + s * 64);
          pb[at] = head[s];
#if FJ_CARRY
          tg_carry_store(J, region + local + at, sov[s]);
#endif
        }
        o += (unsigned int)__popcll(b[s]);
      }
      local += tile_total;
      // per-chunk bookkeeping, rewritten after every tile of the chunk (the last one stands)
      if (threadIdx.x == 0) {
#if FJ_EPILOGUE   // (agent-scope store: the epilogue's reader may sit on another XCD)
        __hip_atomic_store(&J.tile_cnt[tile >> csh], (int)(local - chunk_local0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        J.tile_cnt[tile >> csh] = (int)(local - chunk_local0);
#endif
        J.tile_src[tile >> csh] = (int)chunk_local0;
      }
    }
#endif
    // stage C: tile jC -- pre-filter verdicts; the survivors' first table slot is loaded below (lanes without a survivor read
    // slot 0: always-valid addresses, no branches around the loads)
    const bool doC = jC >= 0 && jC < my_tiles && tile_of(jC) < J.tiles, doB = jB >= 0 && jB < my_tiles && tile_of(jB) < J.tiles,
               doA = jA < my_tiles && tile_of(jA) < J.tiles;
    unsigned int cidx[FJ_STRIPES];
#pragma unroll
    for (int s = 0; s < FJ_STRIPES; s++) cidx[s] = 0;
    if (doC) {
#pragma unroll
      for (int s = 0; s < FJ_STRIPES; s++) {
#if FJ_PF == 2
        bool maybe = (pfl[s] & 1) && (pbw[s] & pbits[s]) == pbits[s];
#elif FJ_PF == 1 || FJ_PF == 3
        bool maybe = (pfl[s] & 1) && ((unsigned int)(pbw[s] >> pbits[s]) & 1u);
#else
        bool maybe = (pfl[s] & 1) != 0;
#endif
#ifdef FJ_EXP_NOPROBE
        maybe = false;
#endif
#if FJ_PF == 3
        skey[s] = pkey[s];
        ssidx[s] = (J.outer & 2) ? 0u : (unsigned int)__popcll(pbw[s] & ((1ULL << pbits[s]) - 1ULL));   // set bits below the key's own (unless nobody reads positions)
        sfl[s] = (unsigned char)((maybe ? 1 : 0) | (pfl[s] & 2));
        cidx[s] = (maybe && !(J.outer & 2)) ? (psidx[s] >> 6) : 0u;                                     // its bitmap word (entry 0 when the rank is not needed)
#else
        skey[s] = pkey[s]; ssidx[s] = psidx[s];
        sfl[s] = (unsigned char)((maybe ? 1 : 0) | (pfl[s] & 2));
        cidx[s] = maybe ? psidx[s] : 0u;
#endif
#if FJ_CARRY
        sov[s] = pov[s];
#endif
      }
    }
    // stage B: tile jB -- filter + key from the rows loaded by the previous iteration; the pre-filter word (exact key bitmap
    // for dense key domains, else the blocked Bloom filter) is loaded below, again from an always-valid address
#if FJ_PF == 2
    unsigned long long bidx[FJ_STRIPES];   // Bloom filter word
#else
    unsigned int bidx[FJ_STRIPES];         // bitmap word: the key range of the bitmap layouts is below 2^32 (32-bit offsets from a scalar base)
#endif
#pragma unroll
    for (int s = 0; s < FJ_STRIPES; s++) bidx[s] = 0;
    if (doB) {
#if FJ_EPILOGUE == 2
      const unsigned int n_rows = (unsigned int)AB.n;
      const unsigned int row0 = (unsigned int)((tile_of(jB) - tB0) * FJ_TILE) + wave_row;   // the page's own rows
#else
      const FpArgs& AB = A;
      const unsigned int row0 = (unsigned int)(tile_of(jB) * FJ_TILE) + wave_row;
#endif
#pragma unroll
      for (int s = 0; s < FJ_STRIPES; s++) {
        const unsigned int row = row0 + s * 64;
        bool sel = false, passed = false;
        long long key = 0;
        if (row < n_rows && tg_filter(AB, row, rw[s])) {
          selected++;
          passed = true;
          const bool kn = tg_key(AB, row, rw[s], key);   // JoinProbe.java:87-97: a null probe key never matches
          sel = !kn;
#if FJ_CARRY
          tg_carry_eval(AB, row, rw[s], pov[s]);
#endif
        }
        bidx[s] = 0;
#if FJ_PF == 1 || FJ_PF == 3
        {
          // key in [key_min, key_max] <=> (key - key_min) mod 2^64 <= key_max - key_min; a key outside cannot match: no probe
          const unsigned long long d = (unsigned long long)key - (unsigned long long)J.pf.key_min;
          sel = sel && d <= key_range;
          pbits[s] = (unsigned int)d & 63u;
          bidx[s] = sel ? (unsigned int)(d >> 6) : 0u;
        }
#elif FJ_PF == 2
        {
          const unsigned long long hm = tg_fmix64((unsigned long long)tg_hash_long(key));
          pbits[s] = tg_bloom_mask(hm);
          bidx[s] = tg_bloom_word(hm, J.pf.bloom_word_mask);
        }
#else
        pbits[s] = 0;
#endif
#if FJ_PF == 3
        pkey[s] = key; psidx[s] = (unsigned int)((unsigned long long)key - (unsigned long long)J.pf.key_min);   // DIRECT: the slot is the key's offset
#else
        pkey[s] = key; psidx[s] = (unsigned int)tg_slot_of(key, J.mask);
#endif
        pfl[s] = (unsigned char)((sel ? 1 : 0) | (passed ? 2 : 0));
      }
    }
    // Every value loaded by the previous iteration has been consumed above; now issue this iteration's loads back to back:
    // slots of tile jC, pre-filter words of tile jB, rows of tile jA.  They are unconditional (always-valid addresses: slot 0
    // / word 0 / the page's last row when a stage is idle or a row is past the end): a branch around a load makes the
    // compiler's in-order wait-count bookkeeping conservative and serialises the three groups.
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < FJ_STRIPES; s++) {
#if FJ_PF == 3
      ssl[s].head = J.pf.rank_base[cidx[s]];   // DIRECT layout: present keys in front of the survivor's bitmap word (entry 0 otherwise: unused)
#else
      ssl[s] = J.slots[cidx[s]];
#endif
    }
#pragma unroll
    for (int s = 0; s < FJ_STRIPES; s++) {
#if FJ_PF == 1 || FJ_PF == 3
      pbw[s] = J.pf.bitmap[bidx[s]];
#elif FJ_PF == 2
      pbw[s] = J.pf.bloom[bidx[s]];
#else
      pbw[s] = ~0ULL;
#endif
    }
    {
#if FJ_EPILOGUE == 2
      // this iteration's page was fetched by the previous one (idle stage: the same page again, its first tile)
      const FpArgs AA = AN;
      const long long tA = doA ? tN : tN0;
      const unsigned int n_rows = (unsigned int)AA.n;
      const unsigned int tile_row0 = (unsigned int)((tA - tN0) * FJ_TILE);
      const unsigned int row0 = tile_row0 + wave_row;
      AB = AA;      // (stage B of the NEXT iteration works on the rows loaded here; this iteration's stage B has run already)
      tB0 = tN0;
      {             // and the page of the next iteration's tile: issued now, needed in an iteration's time
        const long long jn = jA + 1;
        if (jn < my_tiles && tile_of(jn) < J.tiles) {
          tN = tile_of(jn);
          while (tN >= (long long)pt0[pN + 1]) pN++;
          tN0 = pt0[pN];
          AN = pgs[pN];
        }
      }
#else
      const FpArgs& AA = A;
      const unsigned int row0 = (unsigned int)((doA ? tile_of(jA) : 0LL) * FJ_TILE) + wave_row;
      const unsigned int tile_row0 = (unsigned int)((doA ? tile_of(jA) : 0LL) * FJ_TILE);
#endif
      if (tile_row0 + FJ_TILE <= n_rows) {   // interior tile: one address per column, the stripes are constant offsets from it
#pragma unroll
        for (int s = 0; s < FJ_STRIPES; s++) tg_load_row(AA, row0 + s * 64, rw[s]);
      }
      else {
#pragma unroll
        for (int s = 0; s < FJ_STRIPES; s++) {
          const unsigned int row = row0 + s * 64;
          tg_load_row(AA, row < n_rows ? row : n_rows - 1, rw[s]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#if !FJ_CARRY
    // stage D, part 2: compact the pairs of tile jD in input order and append them to the block's region
    if (doD) {
      const long long tile = tile_of(jD);
      const unsigned int row0 = (unsigned int)(tile * FJ_TILE) + wave_row;
      if ((jD & cmask) == 0) chunk_local0 = local;
      unsigned long long b[FJ_STRIPES];
      int wave_total = 0;
#pragma unroll
      for (int s = 0; s < FJ_STRIPES; s++) {
        b[s] = __ballot(emit[s]);
        wave_total += __popcll(b[s]);
      }
      int* Cw = C[jD & 1];
      if (lane == 0) Cw[w] = wave_total;
      __syncthreads();
      int before = 0, tile_total = 0;
#pragma unroll
      for (int w2 = 0; w2 < 4; w2++) { const int c = Cw[w2]; if (w2 < w) before += c; tile_total += c; }
      int* pp = J.pair_probe + (region + local);   // uniform bases, 32-bit lane offsets
      int* pb = J.pair_build + (region + local);
      unsigned int o = (unsigned int)before;
#pragma unroll
      for (int s = 0; s < FJ_STRIPES; s++) {
        if (emit[s]) {
          const unsigned int at = o + __builtin_amdgcn_mbcnt_hi((unsigned)(b[s] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b[s], 0u));
          pp[at] = (int)(row0 + s * 64);
          pb[at] = head[s];
        }
        o += (unsigned int)__popcll(b[s]);
      }
      local += tile_total;
      // per-chunk bookkeeping, rewritten after every tile of the chunk (the last one stands)
      if (threadIdx.x == 0) {
#if FJ_EPILOGUE   // (agent-scope store: the epilogue's reader may sit on another XCD)
        __hip_atomic_store(&J.tile_cnt[tile >> csh], (int)(local - chunk_local0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        J.tile_cnt[tile >> csh] = (int)(local - chunk_local0);
#endif
        J.tile_src[tile >> csh] = (int)chunk_local0;
      }
    }
#endif
  }
  // rows that passed the filter: ONE atomic per workgroup, spread over FJ_COUNT_SLOTS words on separate cache lines (the host adds
  // them up).  One atomic per wave onto a single word serialised at ~13 ns each: 50 us for the 1024 workgroups of a 2^20-row page,
  // 80 us at the end of every full-size launch.
  unsigned long long selected_wave = selected;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) selected_wave += __shfl_down(selected_wave, d, 64);
  __shared__ unsigned long long S[4];
  if (lane == 0) S[w] = selected_wave;
  __syncthreads();
#if !FJ_EPILOGUE
  if (threadIdx.x == 0) {
    const unsigned long long block_total = S[0] + S[1] + S[2] + S[3];
    if (block_total) atomicAdd(&J.counters[(bid % FJ_COUNT_SLOTS) * 16], block_total);
  }
}
#else
  // Epilogue of a small page (J.host_out): the workgroup that finishes LAST scans the chunk counts into the chunks' output offsets (what
  // a scan launch would do) and stores {error word, pairs, selected rows} straight into host memory, then a flag the host polls (what a
  // device-to-host copy and an event would do); it also puts the counters it read back to their rest values, so the next launch needs
  // no launch that zeroes them.  Three launches and a copy per page become one launch.
  // Who is last: the workgroup's one atomic on its count slot also carries an arrival (bits 48+); the workgroup that completes a slot
  // bumps `done`, the one that completes `done` is last -- 64 atomics on one word instead of one per workgroup (13 ns each, serialised).
  // No fences: chunk counts are published with agent-scope stores (write-through) that the barrier's wait has seen acknowledged before the
  // arrival atomic is issued, and the last workgroup reads them with agent-scope loads.
  __shared__ int S_last;
  if (threadIdx.x == 0) {
    const unsigned long long block_total = S[0] + S[1] + S[2] + S[3];
    S_last = 0;
    if (!J.host_out) {
      if (block_total) atomicAdd(&J.counters[(bid % FJ_COUNT_SLOTS) * 16], block_total);
    } else {
      const unsigned int slot = bid % FJ_COUNT_SLOTS;
      const unsigned int in_slot = (nblk - slot + FJ_COUNT_SLOTS - 1) / FJ_COUNT_SLOTS;          // workgroups that count into this slot
      const unsigned int live_slots = nblk < FJ_COUNT_SLOTS ? nblk : FJ_COUNT_SLOTS;
      const unsigned long long before = atomicAdd(&J.counters[slot * 16], block_total + (1ULL << 48));
      if ((unsigned int)(before >> 48) + 1u == in_slot) S_last = (atomicAdd(J.done, 1u) + 1u == live_slots) ? 1 : 0;
    }
  }
  if (J.host_out) {
    __shared__ int S_part[4];
    __syncthreads();
    if (S_last) {
      // (acquire at agent scope = invalidate this XCD's view: the plain, batched loads below then see the other XCDs' write-through stores)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      const long long chunks = (J.tiles + (1LL << J.chunk_shift) - 1) >> J.chunk_shift;
      int* dst = (int*)J.tile_dst;
      int carry = 0;
      for (long long base = 0; base < chunks; base += 2048) {   // 8 consecutive counts per lane: 2 KB per wave and pass, coalesced
        const long long i0 = base + (long long)threadIdx.x * 8;
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = i0 + k < chunks ? J.tile_cnt[i0 + k] : 0;
        int mine = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) mine += v[k];
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
        if (lane == 63) S_part[w] = incl;
        __syncthreads();
        int run = carry + incl - mine;
        for (int k = 0; k < w; k++) run += S_part[k];
        carry += S_part[0] + S_part[1] + S_part[2] + S_part[3];
#pragma unroll
        for (int k = 0; k < 8; k++) if (i0 + k < chunks) { dst[i0 + k] = run; run += v[k]; }
        __syncthreads();
      }
      const int total = carry;
      if (w == 0) {
        unsigned long long sel = __hip_atomic_load(&J.counters[lane * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ((1ULL << 48) - 1);
        __hip_atomic_store(&J.counters[lane * 16], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sel += __shfl_down(sel, d, 64);
        if (lane == 0) {
          const unsigned long long err = __hip_atomic_load(A.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(A.error, ~0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(J.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          J.host_out[0] = err;
          J.host_out[1] = (unsigned long long)total;
          J.host_out[2] = sel;
          __threadfence_system();
          __hip_atomic_store(&J.host_out[7], 1ULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  }
}
#endif

extern "C" __global__ void __launch_bounds__(256) fj_probe(FjArgs J) { fj_probe_body(J, blockIdx.x, gridDim.x); }

// pass 2: one workgroup per chunk: moves the chunk's pairs to their final position and evaluates the probe-side output
// projections for the matching rows only
__device__ __forceinline__ void fj_emit_body(const FjArgs& J, const unsigned int bid, const unsigned int nblk) {
  const FpArgs& A = J.fp;
  const long long chunks = (J.tiles + (1LL << J.chunk_shift) - 1) >> J.chunk_shift;
#if FJ_EPILOGUE == 2
  int pE = 0;   // page of the chunk (= tile: the multi variant runs one-tile chunks); a wave's chunks ascend
#endif
#if FJ_EPILOGUE
  // page variants: chunks are single tiles with a few dozen pairs each -- one WAVE per chunk (four independent chains of dependent loads
  // per workgroup instead of one, no idle lanes waiting for 38 pairs); nothing below synchronises across the workgroup
  const long long first = (long long)bid * 4 + (threadIdx.x >> 6), stride = (long long)nblk * 4;
  const int lane0 = threadIdx.x & 63, lanes = 64;
#else
  const long long first = bid, stride = nblk;
  const int lane0 = threadIdx.x, lanes = 256;
#endif
  for (long long chunk = first; chunk < chunks; chunk += stride) {
    const int cnt = J.tile_cnt[chunk];
    if (cnt == 0) continue;
    const long long src = fj_region_base(chunk % J.grid1, J.tiles, J.grid1, J.chunk_shift) + J.tile_src[chunk];
    const long long dst = J.tile_dst[chunk];
#if FJ_EPILOGUE == 2
    while (chunk >= (long long)J.page_tile0[pE + 1]) pE++;
    FpArgs AE = J.pages[pE];
#pragma unroll
    for (int k = 0; k < TG_MAXP; k++) { AE.out_values[k] = J.fp.out_values[k]; AE.out_nulls[k] = J.fp.out_nulls[k]; }   // (allocated after the descriptors went up)
    const long long row_base = (long long)J.page_tile0[pE] * FJ_TILE;
#else
    const FpArgs& AE = A;
    const long long row_base = 0;
#endif
    for (int i = lane0; i < cnt; i += lanes) {
      const long long row = (long long)J.pair_probe[src + i] - row_base;
#if FJ_PF == 3
      if (!(J.outer & 2)) {
        const int rank = J.pair_build[src + i];   // rank of the key among the build keys (-1: unmatched row of an outer probe)
        J.out_build[dst + i] = (rank < 0 || !J.pf.direct) ? rank : J.pf.direct[rank];
      }
#else
      if (!(J.outer & 2)) J.out_build[dst + i] = J.pair_build[src + i];
#endif
      if (J.n_bcol > 0) {
        const int pos = J.out_build[dst + i];   // (written by this lane just above)
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (k >= J.n_bcol) break;
          const bool nl = pos < 0 || (J.bcol[k].nulls && J.bcol[k].nulls[pos]);
          if (J.bcol[k].out_nulls) J.bcol[k].out_nulls[dst + i] = nl ? 1 : 0;
          if (J.bcol[k].width == 8) ((long long*)J.bcol[k].out_values)[dst + i] = pos < 0 ? 0LL : ((const long long*)J.bcol[k].values)[pos];
          else if (J.bcol[k].width == 4) ((int*)J.bcol[k].out_values)[dst + i] = pos < 0 ? 0 : ((const int*)J.bcol[k].values)[pos];
          else ((unsigned char*)J.bcol[k].out_values)[dst + i] = pos < 0 ? (unsigned char)0 : ((const unsigned char*)J.bcol[k].values)[pos];
        }
      }
#if FJ_CARRY
      (void)row;
      tg_carry_move(J, src + i, dst + i);
#else
      tg_emit_outputs(AE, row, dst + i);
#endif
    }
  }
}
extern "C" __global__ void __launch_bounds__(256) fj_emit(FjArgs J) { fj_emit_body(J, blockIdx.x, gridDim.x); }
#if FJ_EPILOGUE
// One launch for two pages of a page-at-a-time probe: pass 2 of an EARLIER page (its totals have reached the host) next to pass 1 of the NEW
// page.  Both are chains of dependent memory round trips that one page cannot fill the chip with (17.7 and 9.6 us for a 2^20-row page);
// side by side in one grid they overlap.  The probe's workgroups come first (they run longer).
struct FjPairArgs { FjArgs probe; FjArgs emit; unsigned int probe_blocks; unsigned int pad; };
extern "C" __global__ void __launch_bounds__(256) fj_pair(FjPairArgs P) {
  if (blockIdx.x < P.probe_blocks) fj_probe_body(P.probe, blockIdx.x, P.probe_blocks);
  else fj_emit_body(P.emit, blockIdx.x - P.probe_blocks, gridDim.x - P.probe_blocks);
}
#endif
)SRC";

}  // namespace

FusedProbeGpu::FusedProbeGpu(std::vector<int32_t> input_types, const tgpu_page_processor_spec *spec, int32_t join_channel, std::vector<int32_t> output_channels)
    : input_types_(std::move(input_types)), output_channels_(std::move(output_channels)), join_channel_(join_channel)
{
    TG_CHECK_ARG(spec != nullptr, "page processor spec is null");
    nodes_.assign(spec->nodes, spec->nodes + spec->node_count);
    if (spec->string_pool && spec->string_pool_len > 0) pool_.assign(spec->string_pool, spec->string_pool + spec->string_pool_len);
    filter_root_ = spec->filter_root;
    proj_roots_.assign(spec->projection_roots, spec->projection_roots + spec->projection_count);
    TG_CHECK_ARG(join_channel_ >= 0 && join_channel_ < (int)proj_roots_.size(), "join channel out of range");
    for (int32_t ch : output_channels_) TG_CHECK_ARG(ch >= 0 && ch < (int)proj_roots_.size(), "probe output channel out of range");
    for (int32_t r : proj_roots_) {
        TG_CHECK_ARG(r >= 0 && r < (int)nodes_.size(), "projection root out of range");
        proj_types_.push_back(nodes_[(size_t)r].type);
    }
    const int32_t kt = proj_types_[(size_t)join_channel_];
    supported_ = (kt == TGPU_BIGINT || kt == TGPU_INTEGER || kt == TGPU_DATE) && (int)output_channels_.size() <= kFpMaxProj;
    for (int32_t ch : output_channels_) supported_ = supported_ && proj_types_[(size_t)ch] != TGPU_VARCHAR;
    // The fused kernel evaluates the non-key projections lazily (for matching rows only) or not at all (channels the join
    // drops), whereas FilterAndProjectOperator evaluates every projection on every selected row.  That is only equivalent
    // when those projections cannot raise: anything with checked integer arithmetic keeps the unfused composition.
    std::function<bool(int)> can_raise = [&](int idx) -> bool {
        const tgpu_expr_node &nd = nodes_[(size_t)idx];
        if (nd.kind == TGPU_EX_CALL) {
            const bool int_result = nd.type == TGPU_BIGINT || nd.type == TGPU_INTEGER;
            if (int_result && nd.op >= TGPU_OP_ADD && nd.op <= TGPU_OP_NEGATE) return true;
            if (nd.op == TGPU_OP_CAST && nd.n_args == 1 && int_result) {   // narrowing casts are checked (BIGINT -> INTEGER, DOUBLE -> BIGINT / INTEGER)
                const int32_t from = nodes_[(size_t)nd.args[0]].type;
                if (from == TGPU_DOUBLE || (from == TGPU_BIGINT && nd.type == TGPU_INTEGER)) return true;
            }
        }
        if (nd.kind == TGPU_EX_CALL || nd.kind == TGPU_EX_SPECIAL)
            for (int k = 0; k < nd.n_args; k++)
                if (can_raise(nd.args[k])) return true;
        return false;
    };
    for (size_t ch = 0; ch < proj_roots_.size(); ch++)
        if ((int)ch != join_channel_ && can_raise(proj_roots_[ch])) supported_ = false;
    if (supported_) generate();
}

FusedProbeGpu::~FusedProbeGpu() {}

void FusedProbeGpu::generate()
{
    auto cols_decl = [&](const Gen &g) {
        std::ostringstream cols;
        for (int ch : g.used_cols) {
            const int32_t t = input_types_[(size_t)ch];
            const char *T = (t == TGPU_VARCHAR || t == TGPU_BOOLEAN) ? "unsigned char" : ctype(t);
            cols << "  const " << T << "* c" << ch << " = (const " << T << "*)A.col_values[" << ch << "]; (void)c" << ch << ";\n";
            cols << "  const unsigned char* cn" << ch << " = FJ_NO_NULLS ? (const unsigned char*)0 : A.col_nulls[" << ch << "]; (void)cn" << ch << ";\n";
            if (t == TGPU_VARCHAR) cols << "  const int* co" << ch << " = A.col_offsets[" << ch << "]; (void)co" << ch << ";\n";
        }
        return cols.str();
    };
    auto splice = [](std::string t, const std::string &cols) {
        size_t p = t.find("@COLS@");
        if (p != std::string::npos) t.replace(p, 6, cols);
        return t;
    };
    const int32_t kt = proj_types_[(size_t)join_channel_];

    // (a) filter + join key in register-row mode: evaluated for every row of pass 1 from pre-loaded column values
    Gen gr(nodes_, pool_, input_types_);
    gr.reg_mode = true;
    std::vector<std::string> reg_bodies;
    if (filter_root_ >= 0) {
        Val f = gr.gen(filter_root_, 1);
        if (f.type != TGPU_BOOLEAN) gr.bad("filter must be boolean");
        std::ostringstream fb;
        fb << "__device__ inline bool tg_filter(const FpArgs& A, long long row, const TgRow& R) {\n@COLS@" << gr.os.str() << "  return !" << f.n << " && " << f.v << ";\n}\n";
        reg_bodies.push_back(fb.str());
    }
    else reg_bodies.push_back("__device__ inline bool tg_filter(const FpArgs& A, long long row, const TgRow& R) { return true; }\n");
    {
        gr.os.str("");
        Val v = gr.gen(proj_roots_[(size_t)join_channel_], 1);
        std::ostringstream f;
        f << "__device__ inline bool tg_key(const FpArgs& A, long long row, const TgRow& R, long long& key) {\n@COLS@" << gr.os.str() << "  key = (long long)" << v.v
          << ";\n  return " << v.n << ";\n}\n";
        reg_bodies.push_back(f.str());
    }
    // (b) output projections in memory mode: evaluated lazily in pass 2 for the matching rows only
    Gen gm(nodes_, pool_, input_types_);
    gm.tmp = gr.tmp;
    std::vector<std::string> mem_bodies;
    std::set<int> outs(output_channels_.begin(), output_channels_.end());
    for (int ch : outs) {
        gm.os.str("");
        Val v = gm.gen(proj_roots_[(size_t)ch], 1);
        std::ostringstream f;
        f << "__device__ inline bool tg_p" << ch << "(const FpArgs& A, long long row, " << ctype(proj_types_[(size_t)ch]) << "& out) {\n@COLS@" << gm.os.str()
          << "  out = " << v.v << ";\n  return " << v.n << ";\n}\n";
        mem_bodies.push_back(f.str());
    }

    // (c) the same output projections in register-row mode, for the CARRY variant (FJ_CARRY 1): pass 1 evaluates them for every row
    // that passes the filter, from column values it pre-loads with the filter's (coalesced), and writes them next to the pair; pass 2
    // then moves dense values instead of gathering the input columns at the matching rows -- at >= 1 match in 16 rows a sparse gather
    // touches most 128-byte lines of those columns anyway, and a probe launch that is bound by its table lookups (the orders launch
    // of Q3: L2 requests) has the HBM bandwidth to stream them.  Eligible: up to 4 fixed-width outputs of at most 24 bytes together that
    // read fixed-width columns only.
    Gen gc(nodes_, pool_, input_types_);
    gc.reg_mode = true;
    gc.tmp = gm.tmp + 1000;
    std::vector<std::string> carry_bodies;
    int carry_bytes = 0;
    for (size_t i = 0; i < output_channels_.size(); i++) {
        const int ch = output_channels_[i];
        gc.os.str("");
        Val v = gc.gen(proj_roots_[(size_t)ch], 1);
        std::ostringstream f;
        f << "__device__ inline bool tg_c" << i << "(const FpArgs& A, long long row, const TgRow& R, " << ctype(proj_types_[(size_t)ch]) << "& out) {\n@COLS@" << gc.os.str()
          << "  out = " << v.v << ";\n  return " << v.n << ";\n}\n";
        carry_bodies.push_back(f.str());
        carry_bytes += type_width(proj_types_[(size_t)ch]);
    }
    carry_supported_ = !output_channels_.empty() && output_channels_.size() <= 4 && carry_bytes <= 24 && gc.used_cols.empty();   // (used_cols: columns read from memory = VARCHAR inputs)
    std::vector<int> carry_only_cols;   // columns only the carried outputs read
    for (int ch : gc.reg_cols)
        if (!gr.reg_cols.count(ch)) carry_only_cols.push_back(ch);

    std::ostringstream src;
    src << "#ifndef FJ_CARRY\n#define FJ_CARRY 0\n#endif\n";
    src << kPrelude << device_header("device_hash.h") << device_header("device_join.h") << gr.consts.str() << gm.consts.str() << gc.consts.str();
    // the pre-loaded row: one field pair per fixed-width column the filter / key (and, in the carry variant, the outputs) read
    auto field = [&](int ch) {
        const int32_t t = input_types_[(size_t)ch];
        return std::string("  ") + (t == TGPU_BOOLEAN ? "unsigned char" : ctype(t)) + " c" + std::to_string(ch) + "; unsigned char n" + std::to_string(ch) + ";\n";
    };
    auto load = [&](int ch) {
        const int32_t t = input_types_[(size_t)ch];
        const std::string T = t == TGPU_BOOLEAN ? "unsigned char" : ctype(t), C = std::to_string(ch);
        return "  R.c" + C + " = ((const " + T + "*)A.col_values[" + C + "])[row]; R.n" + C + " = (!FJ_NO_NULLS && A.col_nulls[" + C + "]) ? A.col_nulls[" + C + "][row] : 0;\n";
    };
    src << "struct TgRow {\n";
    for (int ch : gr.reg_cols) src << field(ch);
    src << "#if FJ_CARRY\n";
    for (int ch : carry_only_cols) src << field(ch);
    src << "#endif\n";
    if (gr.reg_cols.empty()) src << "  int unused;\n";
    src << "};\n__device__ inline void tg_load_row(const FpArgs& A, long long row, TgRow& R) {\n";
    for (int ch : gr.reg_cols) src << load(ch);
    src << "#if FJ_CARRY\n";
    for (int ch : carry_only_cols) src << load(ch);
    src << "#endif\n";
    src << "  (void)A; (void)row; (void)R;\n}\n__device__ inline void tg_zero_row(TgRow& R) {\n";
    for (int ch : gr.reg_cols) src << "  R.c" << ch << " = 0; R.n" << ch << " = 0;\n";
    src << "#if FJ_CARRY\n";
    for (int ch : carry_only_cols) src << "  R.c" << ch << " = 0; R.n" << ch << " = 0;\n";
    src << "#endif\n";
    src << "  (void)R;\n}\n";
    const std::string rc = cols_decl(gr), mc = cols_decl(gm);
    for (auto &b : reg_bodies) src << splice(b, rc);
    for (auto &b : mem_bodies) src << splice(b, mc);
    // the carried outputs of one row (register form), their evaluation, and the two stores
    src << "struct FjArgs;\nstruct TgOut {\n";
    for (size_t i = 0; i < output_channels_.size(); i++) src << "  " << ctype(proj_types_[(size_t)output_channels_[i]]) << " v" << i << ";\n";
    src << "#if !FJ_NO_NULLS\n  unsigned char nulls;\n#endif\n};\n__device__ inline void tg_zero_out(TgOut& O) {\n";
    for (size_t i = 0; i < output_channels_.size(); i++) src << "  O.v" << i << " = 0;\n";
    src << "#if !FJ_NO_NULLS\n  O.nulls = 0;\n#endif\n}\n#if FJ_CARRY\n";
    {
        const std::string cc = cols_decl(gc);
        for (auto &b : carry_bodies) src << splice(b, cc);
        src << "__device__ inline void tg_carry_eval(const FpArgs& A, long long row, const TgRow& R, TgOut& O) {\n  unsigned char nl = 0;\n";
        for (size_t i = 0; i < output_channels_.size(); i++) {
            const int32_t t = proj_types_[(size_t)output_channels_[i]];
            src << "  { " << ctype(t) << " v = 0; const bool n = tg_c" << i << "(A, row, R, v); O.v" << i << " = n ? (" << ctype(t) << ")0 : v; nl |= n ? " << (1 << i) << " : 0; }\n";
        }
        src << "#if !FJ_NO_NULLS\n  O.nulls = nl;\n#else\n  (void)nl;\n#endif\n}\n";
    }
    src << "#endif\n";
    std::ostringstream carry_funcs;   // need FjArgs: spliced in behind its definition (@CARRY_FUNCS@ in kFjKernels)
    carry_funcs << "#if FJ_CARRY\n__device__ inline void tg_carry_store(const FjArgs& J, long long at, const TgOut& O) {\n";
    for (size_t i = 0; i < output_channels_.size(); i++) {
        const int32_t t = proj_types_[(size_t)output_channels_[i]];
        const std::string T = t == TGPU_BOOLEAN ? "unsigned char" : ctype(t);
        carry_funcs << "  ((" << T << "*)J.carry[" << i << "])[at] = (" << T << ")O.v" << i << ";\n";
    }
    carry_funcs << "#if !FJ_NO_NULLS\n  if (J.carry_nulls) J.carry_nulls[at] = O.nulls;\n#endif\n}\n"
                << "__device__ inline void tg_carry_move(const FjArgs& J, long long from, long long to) {\n  const FpArgs& A = J.fp;\n"
                << "  const unsigned int nl = J.carry_nulls ? J.carry_nulls[from] : 0u; (void)nl;\n";
    for (size_t i = 0; i < output_channels_.size(); i++) {
        const int32_t t = proj_types_[(size_t)output_channels_[i]];
        const std::string T = t == TGPU_BOOLEAN ? "unsigned char" : ctype(t);
        carry_funcs << "  ((" << T << "*)A.out_values[" << i << "])[to] = ((const " << T << "*)J.carry[" << i << "])[from]; if (A.out_nulls[" << i << "]) A.out_nulls[" << i
                    << "][to] = (unsigned char)((nl >> " << i << ") & 1u);\n";
    }
    carry_funcs << "}\n#endif\n";
    (void)kt;
    src << "__device__ inline void tg_emit_outputs(const FpArgs& A, long long row, long long o) {\n";
    for (size_t i = 0; i < output_channels_.size(); i++) {
        const int ch = output_channels_[i];
        const int32_t t = proj_types_[(size_t)ch];
        const char *T = t == TGPU_BOOLEAN ? "unsigned char" : ctype(t);
        src << "  { " << ctype(t) << " v = 0; const bool n = tg_p" << ch << "(A, row, v); ((" << T << "*)A.out_values[" << i << "])[o] = n ? (" << T << ")0 : (" << T
            << ")v; if (A.out_nulls[" << i << "]) A.out_nulls[" << i << "][o] = n ? 1 : 0; }\n";   // (no null vector: the host proved the channel null-free)
    }
    src << "}\n";
    // experiment switches for kernel studies (tools/exp_fused.py); never set in production
    if (const char *exp = getenv("TGPU_FJ_EXP")) src << "#define FJ_EXP_" << exp << " 1\n";
    {
        std::string kernels = kFjKernels;
        const std::string tag = "@FJ_STRIPES@";
        kernels.replace(kernels.find(tag), tag.size(), std::to_string(fj_stripes()));
        const std::string ctag = "@CARRY_FUNCS@";
        kernels.replace(kernels.find(ctag), ctag.size(), carry_funcs.str());
        src << kernels;
    }
    source_ = src.str();
}

// one specialisation per layout of the lookup source (0 no pre-filter, 1 exact key bitmap, 2 blocked Bloom filter, 3 DIRECT:
// bitmap + build position by key offset, no hash table)
// x pages with / without null vectors (FJ_NO_NULLS: the null loads and tests fold away)
static std::string prefilter_source(const std::string &src, int variant)
{
    // the kernels carry their table layout in their name (fj_probe_direct, ..._bitmap, ..._bloom, ..._plain): a profiler's per-kernel
    // statistics then keep the DIRECT-layout launches (TPCH keys) apart from the open-address ones
    static const char *layout[4] = {"plain", "bitmap", "bloom", "direct"};
    const std::string l = layout[variant % 4];
    return "#define FJ_PF " + std::to_string(variant % 4) + "\n#define FJ_NO_NULLS " + std::to_string((variant / 4) % 2) + "\n#define FJ_CARRY " + std::to_string((variant / 8) % 2) +
           "\n#define FJ_EPILOGUE " + std::to_string(variant / 16) + "\n#define fj_pair fj_pair_" + l + "\n#define fj_probe fj_probe_" + l +
           "\n#define fj_emit fj_emit_" + l + "\n" + src;
}

void FusedProbeGpu::precompile()
{
    if (!supported_) return;
    for (int variant = 0; variant < 8; variant++) {
        (void)code_object_for(prefilter_source(source_, variant));
        (void)code_object_for(prefilter_source(source_, 16 + variant));   // the page variants (FJ_EPILOGUE 1: one page, 2: a list of pages)
        (void)code_object_for(prefilter_source(source_, 32 + variant));
    }
    // (the opt-in carry variants, 8 + ..., are compiled on first use)
}

JitModule *FusedProbeGpu::module_for(int kind, bool no_nulls, bool carry, int epilogue)
{
    std::lock_guard<std::mutex> lk(mu_);
    const int variant = kind + (no_nulls ? 4 : 0) + (carry ? 8 : 0) + epilogue * 16;   // epilogue: 0 plain, 1 page, 2 list of pages
    if (!modules_[variant]) modules_[variant] = load_module(prefilter_source(source_, variant));
    return modules_[variant].get();
}

// What a probe page has in flight between its two passes: pass 1 (filter + probe + in-tile compaction) and the scan of the chunk counts are
// launched by begin(), which also starts the read-back of the totals; finish() waits for THAT read only, sizes the output and launches pass 2.
// An operator that keeps the input page alive can put the next page's begin() between the two (operators.cpp), so that nobody waits for
// a read-back on the common path.
struct FusedProbeGpu::Pending {
    FjArgs J{};
    JitModule *module = nullptr;
    int pf_kind = 0;
    int64_t chunks = 0;
    bool outer = false, need_build_positions = false, carry = false;
    BufferPtr tile_cnt, tile_src, tile_dst, misc, pair_probe, pair_build;
    std::vector<BufferPtr> carry_regions;
    Context::AsyncRead read;
    Context::Signal signal;   // epilogue path: the kernel delivers the totals itself
    int64_t grid1 = 0, emit_blocks = 0;
    bool probe_launched = false;
    int kernel_variant = 0;        // FJ_EPILOGUE of the module: 0 whole table, 1 one page, 2 list of pages
    std::vector<bool> col_nulls;   // per input channel: some page of the launch carries a null vector
    BufferPtr descriptors;         // multi-page launches: the pages' FpArgs + first tiles
};

void FusedProbeGpu::process(Context *ctx, const DevicePage &in, const LookupSourceGpu &source, bool outer, bool need_build_positions,
                            std::vector<DeviceColumn> &probe_out, BufferPtr &build_idx, int64_t &count, int64_t &selected_rows,
                            const std::vector<DeviceColumn> *build_cols, std::vector<DeviceColumn> *build_out)
{
    std::shared_ptr<Pending> p = begin(ctx, in, source, outer, need_build_positions);
    finish(ctx, p, in, probe_out, build_idx, count, selected_rows, build_cols, build_out);
}

std::shared_ptr<FusedProbeGpu::Pending> FusedProbeGpu::begin(Context *ctx, const DevicePage &in, const LookupSourceGpu &source, bool outer, bool need_build_positions, bool launch)
{
    return begin(ctx, std::vector<const DevicePage *>{&in}, source, outer, need_build_positions, launch);
}

// rows one multi-page launch may cover: 32 768 tiles (the epilogue's scan takes them 2 048 at a time)
int64_t FusedProbeGpu::multi_page_row_limit() { return kFjMultiMaxTiles * (int64_t)fj_stripes() * 256; }

// `pages`: one page, or several (each non-empty) that are probed as ONE sequence of rows in page order -- one launch, one output page: the
// kernels' multi variant (FJ_EPILOGUE == 2) reads the pages' column pointers from an array of descriptors.  Several pages need the page
// variant's conditions (at most kFjMultiMaxTiles tiles in all, a free signal slot); callers check can_batch() first.
std::shared_ptr<FusedProbeGpu::Pending> FusedProbeGpu::begin(Context *ctx, const std::vector<const DevicePage *> &pages, const LookupSourceGpu &source, bool outer,
                                                             bool need_build_positions, bool launch)
{
    TG_CHECK_STATE(supported_, "fused probe not supported for this configuration");
    TG_CHECK_ARG(!pages.empty(), "no page");
    const DevicePage &in = *pages[0];
    const bool multi = pages.size() > 1;
    for (const DevicePage *pg : pages) TG_CHECK_ARG(pg->cols.size() == input_types_.size(), "page channel count differs from the operator's input types");
    if (need_build_positions) source.ensure_rank();
    IntTableView tv;
    TG_CHECK_STATE(source.int_table(tv) && tv.links == nullptr, "fused probe needs the int-key table without duplicate build keys");
    bool any_nulls = false;
    std::vector<bool> col_nulls(in.cols.size(), false);
    for (const DevicePage *pg : pages)
        for (size_t i = 0; i < pg->cols.size(); i++)
            if (pg->cols[i].nulls != nullptr) any_nulls = col_nulls[i] = true;
    // CARRY (generate(), (c)): pass 1 evaluates the probe-side outputs for every row that passes the filter and stores them with the
    // pair, pass 2 moves dense values instead of gathering the input columns at the matching rows.  Built for VERDICT r1 item 4(d) and
    // MEASURED on Q3's orders launch (1 match in 10 rows, where the gather touches 80-97 % of the output columns' lines anyway): the
    // emit pass falls from 0.63 to 0.28 ms, but the probe launch rises from 0.69 to 1.06 ms (+16 B per row in flight, 118 instead of
    // 80 VGPRs = 4 instead of 6 waves per SIMD under an L2-request-bound lookup) -- a wash (4.31-4.40 ms per step either way).  So the
    // variant is opt-in (TGPU_FJ_CARRY=1; exact key bitmaps, inner joins) and stays tested; the default keeps the two-pass gather.
    bool carry = false;
    if (const char *f = getenv("TGPU_FJ_CARRY")) carry = atoi(f) != 0 && carry_supported_ && (tv.rank_base || tv.bitmap) && !output_channels_.empty() && !outer;
    const int64_t tile_rows = (int64_t)fj_stripes() * 256;
    // the launch's rows: a page's own rows, or -- several pages -- VIRTUAL rows: every page starts a tile
    std::vector<int32_t> page_tile0;
    int64_t n = in.n;
    if (multi) {
        int64_t t = 0;
        for (const DevicePage *pg : pages) {
            TG_CHECK_ARG(pg->n > 0, "empty page in a multi-page launch");
            page_tile0.push_back((int32_t)t);
            t += ceil_div(pg->n, tile_rows);
        }
        page_tile0.push_back((int32_t)t);
        TG_CHECK_ARG(t <= kFjMultiMaxTiles, "too many rows for one multi-page launch");
        n = t * tile_rows;
        carry = false;
    }
    if (n == 0) return nullptr;
    // pages of up to kFjEpilogueMaxChunks tiles (always one-tile chunks, see below) run the kernel variant whose pass 1 ends with the scan and
    // the hand-over of the totals; whole tables keep the variant without it (measured on the SF100 tables: the epilogue's write-through
    // count stores and its extra state cost the lineitem launch 1.50 -> 1.82 ms)
    const bool page_variant = multi || (ceil_div(n, tile_rows) <= kFjEpilogueMaxChunks && getenv("TGPU_DISABLE_PROBE_EPILOGUE") == nullptr);
    JitModule *module = module_for(tv.rank_base ? 3 : (tv.bitmap ? 1 : (tv.bloom ? 2 : 0)), !any_nulls, carry, multi ? 2 : (page_variant ? 1 : 0));
    std::shared_ptr<Pending> pend = std::make_shared<Pending>();
    pend->col_nulls = col_nulls;
    pend->kernel_variant = multi ? 2 : (page_variant ? 1 : 0);
    // the page variants hand their totals over through a signal slot of the context.  None free (many operators of one context inside such a
    // launch at once): one page falls back to the scan launch and the copy; a list of pages cannot -- the caller probes them one by one
    if (page_variant) pend->signal = ctx->begin_signal();
    if (multi && pend->signal.slot < 0) return nullptr;
    struct SignalGuard {   // an allocation that fails between here and the launch must not leave the slot waiting for a kernel that never runs
        Context *ctx;
        Context::Signal *signal;
        bool armed = true;
        ~SignalGuard()
        {
            if (armed && signal->slot >= 0) {
                ctx->abandon_signal(*signal);
                signal->slot = -1;
            }
        }
    } signal_guard{ctx, &pend->signal};
    pend->module = module;
    pend->outer = outer;
    pend->need_build_positions = need_build_positions;
    pend->carry = carry;
    FjArgs &J = pend->J;
    for (size_t i = 0; i < in.cols.size() && i < (size_t)kFpMaxCols; i++) {
        J.fp.col_values[i] = in.cols[i].values;
        J.fp.col_nulls[i] = in.cols[i].nulls;
        J.fp.col_offsets[i] = in.cols[i].offsets;
    }
    J.fp.n = multi ? in.n : n;
    J.slots = tv.slots;
    J.mask = tv.mask;
    J.bitmap = tv.bitmap;
    J.key_min = tv.key_min;
    J.key_max = tv.key_max;
    J.bloom = tv.bloom;
    J.bloom_word_mask = tv.bloom_word_mask;
    J.direct = tv.direct;
    J.rank_base = tv.rank_base;
    J.outer = (outer ? 1 : 0) | (need_build_positions ? 0 : 2);
    J.tiles = ceil_div(n, tile_rows);
    TG_CHECK_ARG(n <= 0x7fffffffLL && J.tiles <= 0x7fffffffLL, "page too large");
    // persistent workgroups: exactly as many as are resident at once (a second round of workgroups would only add a tail)
    const int pf_kind = tv.rank_base ? 3 : (tv.bitmap ? 1 : (tv.bloom ? 2 : 0));
    static const char *probe_names[4] = {"fj_probe_plain", "fj_probe_bitmap", "fj_probe_bloom", "fj_probe_direct"};
    const int64_t resident = (int64_t)ctx->cu_count() * module->blocks_per_cu(probe_names[pf_kind]);
    // chunks of consecutive tiles per workgroup (fj_probe): up to 64 tiles, but at least ~4 chunks per resident workgroup
    int chunk_shift = 0;
    const int max_chunk_shift = getenv("TGPU_FJ_CHUNK_SHIFT") ? atoi(getenv("TGPU_FJ_CHUNK_SHIFT")) : 6;
    while (!multi && chunk_shift < max_chunk_shift && (J.tiles >> (chunk_shift + 1)) >= 4 * resident) chunk_shift++;   // (multi: a chunk must not straddle pages)
    J.chunk_shift = chunk_shift;
    const int64_t chunks = (J.tiles + (1ll << chunk_shift) - 1) >> chunk_shift;
    pend->chunks = chunks;
    pend->pf_kind = pf_kind;
    const int64_t grid1 = std::min<int64_t>(chunks, resident);
    J.grid1 = grid1;
    BufferPtr tile_cnt = ctx->alloc((size_t)chunks * 4), tile_src = ctx->alloc((size_t)chunks * 4), tile_dst = ctx->alloc((size_t)chunks * 4);
    // [0] expression error word, [2] total pairs, [16 + 16 i] rows selected by the filter as counted by workgroups i mod 64 (FJ_COUNT_SLOTS)
    constexpr int kMiscWords = kFjMiscWords;
    // small pages: the probe kernel's last workgroup does the scan and hands the totals to the host itself (fj_probe's epilogue): its
    // counters live in the context's persistent scratch words ([0] error word, [1] finished workgroups, [16 + 16 i] selected rows), which
    // every launch leaves at rest
    static_assert((size_t)kFjMiscWords * 8 <= Context::kZeroedScratchBytes, "the probe's counters fit the context's scratch words");
    if (pend->signal.slot >= 0 && chunk_shift != 0) {   // (cannot happen for at most kFjEpilogueMaxChunks tiles: fewer than 8 x the resident workgroups)
        ctx->abandon_signal(pend->signal);
        pend->signal.slot = -1;
    }
    const bool epilogue = pend->signal.slot >= 0;
    BufferPtr misc;
    if (epilogue) {
        unsigned long long *z = static_cast<unsigned long long *>(ctx->zeroed_scratch());
        J.fp.error = z;
        J.done = reinterpret_cast<unsigned int *>(z + 1);
        J.counters = z + 16;
        J.host_out = pend->signal.device;
    } else {
        misc = ctx->alloc((size_t)kMiscWords * 8);
        init_words_kernel<<<1, 256, 0, ctx->stream()>>>(misc->as<unsigned long long>(), 1, kMiscWords - 1);   // one launch: [0] = ~0 (no error), the rest 0
        J.fp.error = misc->as<unsigned long long>();
        J.counters = misc->as<unsigned long long>() + 16;
    }
    if (multi) {
        // the pages' descriptors and their first tiles, one upload: [FpArgs x pages][int32 x (pages + 1)]
        const size_t desc_bytes = pages.size() * sizeof(FpArgs), tile_bytes = page_tile0.size() * 4;
        std::vector<uint8_t> host(desc_bytes + tile_bytes);
        for (size_t i = 0; i < pages.size(); i++) {
            FpArgs d{};
            for (size_t c = 0; c < pages[i]->cols.size() && c < (size_t)kFpMaxCols; c++) {
                d.col_values[c] = pages[i]->cols[c].values;
                d.col_nulls[c] = pages[i]->cols[c].nulls;
                d.col_offsets[c] = pages[i]->cols[c].offsets;
            }
            d.n = pages[i]->n;
            d.error = J.fp.error;
            memcpy(host.data() + i * sizeof(FpArgs), &d, sizeof(FpArgs));
        }
        memcpy(host.data() + desc_bytes, page_tile0.data(), tile_bytes);
        pend->descriptors = ctx->alloc(host.size());
        ctx->upload(pend->descriptors->ptr(), host.data(), host.size());
        J.pages = pend->descriptors->as<FpArgs>();
        J.page_tile0 = reinterpret_cast<const int32_t *>(pend->descriptors->as<uint8_t>() + desc_bytes);
        J.n_pages = (int32_t)pages.size();
    }
    J.tile_cnt = tile_cnt->as<int32_t>();
    J.tile_src = tile_src->as<int32_t>();
    J.tile_dst = tile_dst->as<int32_t>();
    // without duplicate build keys a probe row yields at most one pair: the private regions hold tiles x 2048 rows in total
    const int64_t cap = (chunks << chunk_shift) * tile_rows;
    BufferPtr pair_probe = ctx->alloc((size_t)cap * 4), pair_build = ctx->alloc((size_t)cap * 4);
    J.pair_probe = pair_probe->as<int32_t>();
    J.pair_build = pair_build->as<int32_t>();
    std::vector<BufferPtr> &carry_regions = pend->carry_regions;
    if (carry) {
        bool nullable = false;
        for (size_t i = 0; i < output_channels_.size(); i++) {
            carry_regions.push_back(ctx->alloc((size_t)cap * type_width(proj_types_[(size_t)output_channels_[i]])));
            J.carry[i] = carry_regions.back()->ptr();
            const tgpu_expr_node &root = nodes_[(size_t)proj_roots_[(size_t)output_channels_[i]]];
            nullable = nullable || !(root.kind == TGPU_EX_INPUT && root.op >= 0 && root.op < (int)in.cols.size() && in.cols[(size_t)root.op].nulls == nullptr);
        }
        if (nullable) {
            carry_regions.push_back(ctx->alloc((size_t)cap));
            J.carry_nulls = carry_regions.back()->as<uint8_t>();
        }
    }
    pend->grid1 = grid1;
    pend->tile_cnt = tile_cnt; pend->tile_src = tile_src; pend->tile_dst = tile_dst; pend->misc = misc; pend->pair_probe = pair_probe; pend->pair_build = pair_build;
    // (launch == false: the caller puts pass 1 into one launch with another page's pass 2, launch_pair; only the page variant can)
    if (launch || !epilogue) launch_probe(ctx, pend);
    signal_guard.armed = false;
    return pend;
}

void FusedProbeGpu::launch_probe(Context *ctx, const std::shared_ptr<Pending> &pend)
{
    static const char *probe_names[4] = {"fj_probe_plain", "fj_probe_bitmap", "fj_probe_bloom", "fj_probe_direct"};
    if (!pend || pend->probe_launched) return;
    FjArgs &J = pend->J;
    {
        ProfileScope ps(ctx, "fused_filter_probe");
        launch_args(pend->module->fn(probe_names[pend->pf_kind]), (int)pend->grid1, J, ctx->stream());
    }
    pend->probe_launched = true;
    if (pend->signal.slot < 0) {
        {
            ProfileScope ps(ctx, "fused_probe_scan");
            k::exclusive_scan_i32(ctx, J.tile_cnt, pend->tile_dst->as<int32_t>(), pend->chunks, (int64_t *)(pend->misc->as<unsigned long long>() + 2));
        }
        pend->read = ctx->begin_read(pend->misc->ptr(), (size_t)kFjMiscWords * 8);
    }
}

bool FusedProbeGpu::can_pair(const std::shared_ptr<Pending> &emit_side, const std::shared_ptr<Pending> &probe_side) const
{
    return emit_side && probe_side && emit_side->module == probe_side->module && emit_side->pf_kind == probe_side->pf_kind && emit_side->emit_blocks > 0 &&
           !probe_side->probe_launched && probe_side->signal.slot >= 0 && getenv("TGPU_DISABLE_PROBE_PAIRING") == nullptr;
}

// pass 2 of `emit_side` (prepared: finish(..., launch = false)) and pass 1 of `probe_side` (prepared: begin(..., launch = false)) in one launch
void FusedProbeGpu::launch_pair(Context *ctx, const std::shared_ptr<Pending> &emit_side, const std::shared_ptr<Pending> &probe_side)
{
    static const char *pair_names[4] = {"fj_pair_plain", "fj_pair_bitmap", "fj_pair_bloom", "fj_pair_direct"};
    TG_CHECK_STATE(can_pair(emit_side, probe_side), "these two pages cannot share a launch");
    struct FjPairArgs {
        FjArgs probe, emit;
        unsigned int probe_blocks, pad;
    } P{probe_side->J, emit_side->J, (unsigned int)probe_side->grid1, 0u};
    {
        ProfileScope ps(ctx, "fused_probe_pair");
        launch_args(probe_side->module->fn(pair_names[probe_side->pf_kind]), (int)(probe_side->grid1 + emit_side->emit_blocks), P, ctx->stream());
    }
    probe_side->probe_launched = true;
    emit_side->emit_blocks = 0;   // (launched)
}

void FusedProbeGpu::launch_emit(Context *ctx, const std::shared_ptr<Pending> &pend)
{
    static const char *emit_names[4] = {"fj_emit_plain", "fj_emit_bitmap", "fj_emit_bloom", "fj_emit_direct"};
    if (!pend || pend->emit_blocks <= 0) return;
    ProfileScope ps(ctx, "fused_probe_emit");
    launch_args(pend->module->fn(emit_names[pend->pf_kind]), (int)pend->emit_blocks, pend->J, ctx->stream());
    pend->emit_blocks = 0;
}

void FusedProbeGpu::cancel(Context *ctx, const std::shared_ptr<Pending> &pend)
{
    if (!pend) return;
    if (!pend->probe_launched) {   // prepared only: no kernel will ever write the slot
        ctx->abandon_signal(pend->signal);
        pend->signal.slot = -1;
        return;
    }
    if (pend->signal.slot >= 0) {
        unsigned long long words[Context::kSignalWords];
        ctx->finish_signal(pend->signal, words);   // waits: the kernel still owns the slot until it has written it
        pend->signal.slot = -1;
        return;
    }
    std::vector<unsigned long long> h((size_t)kFjMiscWords);
    ctx->finish_read(pend->read, h.data());   // gives the read-back slot back; the page's buffers go with `pend`
}

void FusedProbeGpu::finish(Context *ctx, const std::shared_ptr<Pending> &pend, const DevicePage &in, std::vector<DeviceColumn> &probe_out, BufferPtr &build_idx, int64_t &count,
                           int64_t &selected_rows, const std::vector<DeviceColumn> *build_cols, std::vector<DeviceColumn> *build_out, bool launch)
{
    count = 0;
    selected_rows = 0;
    probe_out.clear();
    if (!pend) return;
    FjArgs &J = pend->J;
    const int64_t chunks = pend->chunks;
    const bool outer = pend->outer, need_build_positions = pend->need_build_positions;
    TG_CHECK_STATE(pend->probe_launched, "pass 1 of this page has not been launched");
    if (pend->signal.slot >= 0) {
        unsigned long long words[Context::kSignalWords];
        Context::Signal sig = pend->signal;
        pend->signal.slot = -1;
        ctx->finish_signal(sig, words);
        raise_expression_error(words[0]);
        count = (int64_t)words[1];
        selected_rows = (int64_t)words[2];
    } else {
        std::vector<unsigned long long> h((size_t)kFjMiscWords);
        ctx->finish_read(pend->read, h.data());
        raise_expression_error(h[0]);
        for (int i = 0; i < kFjCountSlots; i++) selected_rows += (int64_t)h[(size_t)(16 + i * 16)];
        count = (int64_t)h[2];
    }
    if (count == 0) return;
    if (count > 0x7fffffffLL) fail(TGPU_ERR_INSUFFICIENT_RESOURCES, "join output of one probe page cannot exceed 2 billion rows");
    build_idx = ctx->alloc((size_t)count * 4);
    J.out_build = build_idx->as<int32_t>();
    for (size_t i = 0; i < output_channels_.size(); i++) {
        DeviceColumn c;
        c.type = proj_types_[(size_t)output_channels_[i]];
        c.n = count;
        c.values_buf = ctx->alloc((size_t)count * type_width(c.type));
        c.values = c.values_buf->ptr();
        // an output that is a plain reference to an input column without a null vector cannot hold a null: no null vector is
        // produced for it (what Block.mayHaveNull() == false gives the reference's downstream operators), so the build / group-by
        // kernels behind the join skip their null checks for the channel
        const tgpu_expr_node &root = nodes_[(size_t)proj_roots_[(size_t)output_channels_[i]]];
        const bool null_free = root.kind == TGPU_EX_INPUT && root.op >= 0 && root.op < (int)pend->col_nulls.size() && !pend->col_nulls[(size_t)root.op];
        if (!null_free) {
            c.nulls_buf = ctx->alloc((size_t)count);
            c.nulls = c.nulls_buf->as<uint8_t>();
        }
        J.fp.out_values[i] = c.values_buf->ptr();
        J.fp.out_nulls[i] = null_free ? nullptr : c.nulls_buf->as<uint8_t>();
        probe_out.push_back(c);
    }
    J.n_bcol = 0;
    if (build_cols && build_out && need_build_positions) {
        TG_CHECK_ARG((int)build_cols->size() <= kFjMaxBuildCols, "too many build channels for the fused gather");
        for (const DeviceColumn &src : *build_cols) {
            TG_CHECK_ARG(src.type != TGPU_VARCHAR, "the fused gather takes fixed-width build channels");
            DeviceColumn c;
            c.type = src.type;
            c.n = count;
            c.values_buf = ctx->alloc((size_t)count * type_width(c.type));
            c.values = c.values_buf->ptr();
            if (src.nulls != nullptr || outer) {
                c.nulls_buf = ctx->alloc((size_t)count);
                c.nulls = c.nulls_buf->as<uint8_t>();
            }
            FjArgs::BuildCol &b = J.bcol[J.n_bcol++];
            b.values = src.values;
            b.nulls = src.nulls;
            b.out_values = c.values_buf->ptr();
            b.out_nulls = c.nulls_buf ? c.nulls_buf->as<uint8_t>() : nullptr;
            b.width = type_width(c.type);
            build_out->push_back(c);
        }
    }
    // (page variants of the kernels: one wave per chunk, four chunks per workgroup)
    pend->emit_blocks = std::min<int64_t>(pend->kernel_variant ? ceil_div(chunks, (int64_t)4) : chunks, (int64_t)ctx->cu_count() * 8);
    if (launch) launch_emit(ctx, pend);   // (else: the caller launches it, alone or paired with the next page's pass 1)
    // no second error read-back: the projections evaluated by pass 2 cannot raise (the constructor keeps anything with checked
    // integer arithmetic on the unfused path), and pass 1's filter / key errors were raised above
}

// =====================================================================================================================
// FusedAggGpu: filter -> row mask; projections + accumulate in one kernel
// =====================================================================================================================
namespace {


// rows per lane and tile in the fused accumulate kernels (register rows: 2 x stripes x row dwords live across the LDS phase)
static int fa_stripes()
{
    const char *e = getenv("TGPU_FA_STRIPES");
    const int v = e ? atoi(e) : 8;
    return v >= 1 && v <= 8 ? v : 8;
}

const char *kFaKernels = R"SRC(
struct FaArgs {
  FpArgs fp;
  const int* gids;
  unsigned char* mask_out;
  TgAggState st[TG_MAX_AGGS];
  TgLowCardPlan plan;
  long long tiles;
  int lowcard;
  int pad;
  const unsigned int* ord_keys;   // ORDERED mode: (group id + 1) of the page's rows in (group, row) order, 0 = filtered row
  const int* ord_rows;            //               and their row numbers
  const unsigned char* gids8;     // compact group ids (id + 1, 0 = filtered row) instead of gids
  TgFoldScratch fold;             // low-cardinality launches: the workgroups' folded partials
  const unsigned long long* gate; // speculative launch behind a group-by probe: its counters; anything but clean = do nothing
  const int* ord_stretch;         // ORDERED mode, chained kernel over all groups: {first, end} of every group's stretch of ord_keys; or
  long long* ord_list;            // the groups the lane-per-group kernel handed over after ord_handoff rows: {group, next index} pairs,
  unsigned int* ord_list_count;   // their number (starts at 0); ord_handoff 0 = lanes walk their groups to the end
  int ord_handoff;
  int pad4;
};
// clean = no row met a new group ([0]), no table overflow ([2]), no expression error ([7] == ~0): groupby.h GbhSpeculateFn
#define FA_GATE_CLOSED(F) ((F).gate && (((F).gate[0] | (F).gate[2] | ~(F).gate[7]) != 0ULL))
// group id of a row: compact byte ids when the group-by table delivered them, else int32 ids, else the single global group
// (FA_GID8 is a compile-time variant: a run-time choice would put branches around the pipelined loads)
#ifndef FA_GID8
#define FA_GID8 0
#endif
// FA_GID_RAW is what the pipelined load keeps in a register (no arithmetic on it before the tile is processed: that would
// make the prefetch wait for its own load); FA_GID_ID / FA_GID_LIVE decode it at accumulation time
#if FA_GID8
#define FA_GID_RAW(F, row) ((int)(F).gids8[row])
#define FA_GID_NONE 0
#define FA_GID_LIVE(g) ((g) > 0)
#define FA_GID_ID(g) ((g) - 1)
#else
#define FA_GID_RAW(F, row) ((F).gids ? (F).gids[row] : 0)
#define FA_GID_NONE (-1)
#define FA_GID_LIVE(g) ((g) >= 0)
#define FA_GID_ID(g) (g)
#endif
#define FA_STRIPES @FA_STRIPES@
#define FA_TILE (FA_STRIPES * 256)

// pass A: the filter as a row mask (one byte per row) for the group-by table
extern "C" __global__ void __launch_bounds__(256) fa_mask(FaArgs F) {
  const FpArgs& A = F.fp;
  for (long long row = (long long)blockIdx.x * 256 + threadIdx.x; row < A.n; row += (long long)gridDim.x * 256)
    F.mask_out[row] = tg_filter_mem(A, row) ? 1 : 0;
}

// pass C: group id + the aggregates' input projections (evaluated in registers from pre-loaded column values, next tile's
// loads in flight while the current tile accumulates) -> lane-private LDS accumulators (few groups) or exact global atomics
template <bool LC> __device__ inline void fa_accumulate_body(const FaArgs& F, unsigned char* lds) {
  const FpArgs& A = F.fp;
  TgRow cur[FA_STRIPES], nxt[FA_STRIPES];
  int gcur[FA_STRIPES], gnxt[FA_STRIPES];
#pragma unroll
  for (int s = 0; s < FA_STRIPES; s++) {
    const long long row = (long long)blockIdx.x * FA_TILE + threadIdx.x + s * 256;
    tg_zero_row(cur[s]);
    gcur[s] = FA_GID_NONE;
    if (blockIdx.x < F.tiles && row < A.n) { gcur[s] = FA_GID_RAW(F, row); tg_load_row(A, row, cur[s]); }
  }
  // drain the prologue loads here: otherwise their pending state flows into the loop header and the compiler's waitcnt
  // insertion (vmcnt is in-order) puts vmcnt(<=4) waits inside the tile processing, which also wait for the prefetch
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
  for (long long tile = blockIdx.x; tile < F.tiles; tile += gridDim.x) {
    const long long row0 = tile * FA_TILE + threadIdx.x;
    const long long ntile = tile + gridDim.x;
#pragma unroll
    for (int s = 0; s < FA_STRIPES; s++) {
      const long long row = ntile * FA_TILE + threadIdx.x + s * 256;
      tg_zero_row(nxt[s]);
      gnxt[s] = FA_GID_NONE;
      if (ntile < F.tiles && row < A.n) { gnxt[s] = FA_GID_RAW(F, row); tg_load_row(A, row, nxt[s]); }
    }
    // keep the prefetch loads up here (issued before the LDS phase, landing while it runs): without the fence the scheduler
    // sinks them next to their first use to save registers, which serialises HBM latency with the accumulation
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < FA_STRIPES; s++) {
      if (FA_GID_LIVE(gcur[s])) {
        if (LC) tg_accumulate_row_lc(F, A, row0 + s * 256, cur[s], FA_GID_ID(gcur[s]), lds);
        else tg_accumulate_row_gl(F, A, row0 + s * 256, cur[s], FA_GID_ID(gcur[s]));
      }
    }
#pragma unroll
    for (int s = 0; s < FA_STRIPES; s++) { cur[s] = nxt[s]; gcur[s] = gnxt[s]; }
  }
}

extern "C" __global__ void __launch_bounds__(256) fa_accumulate_lowcard(FaArgs F) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[FA_LDS_BYTES];
  if (FA_GATE_CLOSED(F)) return;
#ifndef FA_DEBUG_SKIP
#define FA_DEBUG_SKIP 0
#endif
  if (!(FA_DEBUG_SKIP & 1)) tg_lc_zero(lds, F.plan);
  if (!(FA_DEBUG_SKIP & 2)) fa_accumulate_body<true>(F, lds);
  if (!(FA_DEBUG_SKIP & 4)) tg_lc_fold(lds, F.plan, F.st, F.fold);
}

extern "C" __global__ void __launch_bounds__(256) fa_accumulate_global(FaArgs F) {
  if (FA_GATE_CLOSED(F)) return;
  fa_accumulate_body<false>(F, (unsigned char*)0);
}

// ORDERED mode, row order through chains (device_agg.h): workgroup b adds the rows of group b (ord_stretch: every group of the page), or the
// workgroups share the list of groups that fa_accumulate_ordered's lanes handed over after their first ord_handoff rows (a group far
// longer than the others: its lane would be the whole launch)
extern "C" __global__ void __launch_bounds__(TG_ORD_WAVES * 64) fa_ordered_chain(FaArgs F) {
  __shared__ __attribute__((aligned(16))) double vals[2 * TG_ORD_MAX_DOUBLES * TG_ORD_STRIDE];
  const FpArgs& A = F.fp;
  if (FA_GATE_CLOSED(F)) return;
  if (F.ord_stretch) {
    const long long s = F.ord_stretch[(size_t)blockIdx.x * 2], e = F.ord_stretch[(size_t)blockIdx.x * 2 + 1];
    if (e == s) return;
    tg_accumulate_group_chained(F, A, (long long)blockIdx.x, s, e, vals);
    return;
  }
  const unsigned int count = *F.ord_list_count;
  for (unsigned int b = blockIdx.x; b < count; b += gridDim.x) {
    const long long g = F.ord_list[(size_t)b * 2], s = F.ord_list[(size_t)b * 2 + 1];
    long long lo = s, hi = A.n;          // the end of the group's stretch: first index whose key is larger
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (F.ord_keys[mid] <= (unsigned int)(g + 1)) lo = mid + 1; else hi = mid; }
    if (lo > s) tg_accumulate_group_chained(F, A, g, s, lo, vals);
  }
}

extern "C" __global__ void __launch_bounds__(256) fa_accumulate_ordered(FaArgs F) {
  const FpArgs& A = F.fp;
  if (FA_GATE_CLOSED(F)) return;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < A.n; i += (long long)gridDim.x * 256) {
    const unsigned int key = F.ord_keys[i];
    if (key == 0 || (i > 0 && F.ord_keys[i - 1] == key)) continue;
    tg_accumulate_group_ordered(F, A, i, A.n);
  }
}
)SRC";

static const char *kFqKernel = R"SRC(
// ---- ONE pass per page (DESIGN.md "Page granularity"): filter + group lookup + the aggregates' input projections + accumulate, no group ids
// in memory, no read-back in front of the next page.  For the steady state of few groups: a row is matched against the register
// signatures / LDS records of the groups the table holds (at most FG_LDS_GROUPS); a row that matches none only COUNTS (counters[0]) -- the
// page is then "dirty": its totals stay pending and are dropped, the host re-runs it through the insert protocol.  A clean page's totals are
// made final by the NEXT one-pass launch (stream order), or by the host for the last one.
struct FqArgs {
  FaArgs fa;
  TgKeyCols store;
  int store_groups;
  int pad;
  unsigned long long* counters;        // this page: [0] rows of unknown groups, [7] expression error word (~0 = none)
  const unsigned long long* prev;      // the previous one-pass page's counters (its totals are in fold.pending), or null
  unsigned long long* host_out;        // host-visible words for the counters ({[0], [2], [7]}, then a flag in word 7), written by the
  unsigned int* done;                  // workgroup that finishes last (`done` = finished workgroups, rests at 0); null: the host copies
  const FpArgs* pages;                 // fq_onepass_multi: the launch's pages in order (device memory; error words all = fa.fp.error)
  int n_pages;
  int pad3;
};
#define FQ_STRIPES 8
#define FQ_TILE (FQ_STRIPES * 256)
// the raw loads of one row: filter inputs, key cells (phase A: null flags, fixed-width values, varchar offsets) and the aggregates' inputs
struct TgQRow { TgFRow f; TgKeyRow k; TgRow r; };
__device__ inline void fq_load_a(const FpArgs& A, long long row, TgQRow& R) {
  // unconditional (a row past the end re-reads the page's last row): scalar tile base 0 + the row as the lane offset
  fg_zero_key(R.k);
  tg_load_frow(A, 0, (unsigned int)row, R.f);
  fg_load_key_a(A, 0, (unsigned int)row, R.k);
  tg_load_row(A, row, R.r);
}
// MULTI: the launch covers a LIST of pages (Q.pages: one FpArgs per page, in device memory), taken as one sequence of tiles in page order --
// a blocking operator that is handed small pages need not launch per page (the fixed cost of a launch, ~18 us, is 1.3 M rows' worth of
// streaming).  The page of a tile is workgroup-uniform: its column pointers are scalar loads.
template <bool MULTI> __device__ __forceinline__ void fq_body(const FqArgs& Q) {
  const FaArgs& F = Q.fa;
#define FQ_PAGE(i) (MULTI ? Q.pages[i] : F.fp)
#define FQ_TILES(n) (((n) + FQ_TILE - 1) / FQ_TILE)
  __shared__ __attribute__((aligned(16))) unsigned char lds[FQ_LDS_BYTES];   // the lane-private states of the groups a one-pass launch may meet
  __shared__ __attribute__((aligned(16))) unsigned char rec[FG_LDS_GROUPS * FG_NKEYS * 32];
  const int np = MULTI ? Q.n_pages : 1;
  // this workgroup's tiles: global tile numbers blockIdx.x, + gridDim.x, ...; (pc, base) = the page of tile g and that page's first tile
  long long g = blockIdx.x, base = 0;
  int pc = 0;
  while (pc < np) { const long long t = FQ_TILES(FQ_PAGE(pc).n); if (g < base + t) break; base += t; pc++; }
  // the first tile's loads go out before anything else: they land under the set-up below
  TgQRow cur[FQ_STRIPES], nxt[FQ_STRIPES];
  {
    const FpArgs A = FQ_PAGE(pc < np ? pc : 0);
    const long long last = A.n - 1;
#pragma unroll
    for (int s = 0; s < FQ_STRIPES; s++) {
      const long long row = (g - base) * FQ_TILE + threadIdx.x + s * 256;
      fq_load_a(A, (pc < np && row < last) ? row : last, cur[s]);
    }
  }
  // (1) the previous page's totals of this workgroup row: final if that page was clean, else dropped (overwritten below either way)
  if (Q.prev && (Q.prev[0] | Q.prev[2] | ~Q.prev[7]) == 0ULL) tg_commit_pending(F.fold, F.plan.n_aggs, F.st, false);
  tg_lc_zero(lds, F.plan);
  const int lg = Q.store_groups < FG_LDS_GROUPS ? Q.store_groups : FG_LDS_GROUPS;
  fg_build_records(Q.store, lg, rec);   // ends with a workgroup barrier (also between the commit above and the fold below)
  TgRecReg rr[FG_REG_GROUPS];
#pragma unroll
  for (int g2 = 0; g2 < FG_REG_GROUPS; g2++) {
    fg_load_recreg(rec, g2 < lg ? g2 : 0, rr[g2]);
    if (g2 >= lg) rr[g2].s[0] = ~(FG_SIG_T)0;
  }
  const int rg = lg < FG_REG_GROUPS ? lg : FG_REG_GROUPS;
  unsigned long long unknown = 0;
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the prologue's loads do not flow into the loop header (see fa_accumulate_body)
  while (pc < np) {
    const FpArgs A = FQ_PAGE(pc);
    const long long tb = (g - base) * FQ_TILE;
    const long long left = A.n - tb;
    const unsigned int lim = (unsigned int)(left < FQ_TILE ? left : FQ_TILE) - 1u;
    bool sel[FQ_STRIPES];
    // this tile's first varchar bytes (their offsets have landed), then the NEXT tile's row loads: both in flight under the accumulation
#pragma unroll
    for (int s = 0; s < FQ_STRIPES; s++) {
      const unsigned int o = threadIdx.x + s * 256;
      sel[s] = o <= lim && tg_filter_f(A, tb + o, cur[s].f);
      fg_key_lengths(cur[s].k);
    }
#pragma unroll
    for (int s = 0; s < FQ_STRIPES; s++) fg_load_key_b(A, cur[s].k, sel[s]);
    const long long gn = g + gridDim.x;
    long long bn = base;
    int pn = pc;
    while (pn < np) { const long long t = FQ_TILES(FQ_PAGE(pn).n); if (gn < bn + t) break; bn += t; pn++; }
    {
      // (past the last tile: the loads re-read this page's last row)
      const FpArgs N = FQ_PAGE(pn < np ? pn : pc);
      const long long nlast = N.n - 1;
#pragma unroll
      for (int s = 0; s < FQ_STRIPES; s++) {
        const long long row = (gn - bn) * FQ_TILE + threadIdx.x + s * 256;
        fq_load_a(N, (pn < np && row < nlast) ? row : nlast, nxt[s]);
      }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < FQ_STRIPES; s++) {
      const unsigned int o = threadIdx.x + s * 256;
      FG_SIG_T sig[FG_SIG_WORDS];
      bool open;
      fg_row_sig(cur[s].k, sig, open);
      int result = -1;
#pragma unroll
      for (int g2 = FG_REG_GROUPS - 1; g2 >= 0; g2--) {
        bool same = true;
#pragma unroll
        for (int j = 0; j < FG_SIG_WORDS; j++) same = same && sig[j] == rr[g2].s[j];
        result = same ? g2 : result;
      }
      const bool redo = open && result >= 0;
      result = (redo || !sel[s]) ? -1 : result;
      if (sel[s] && result < 0) {   // not decided by the signatures: the byte-wise LDS records; a row no record matches is left to the host
        for (int g2 = redo ? 0 : rg; g2 < lg; g2++)
          if (fg_eq_record(A, cur[s].k, rec, g2) > 0) { result = g2; break; }
        unknown += result < 0 ? 1ULL : 0ULL;
      }
      if (sel[s] && result >= 0) tg_accumulate_row_lc(F, A, tb + o, cur[s].r, result, lds);
    }
#pragma unroll
    for (int s = 0; s < FQ_STRIPES; s++) cur[s] = nxt[s];
    g = gn; base = bn; pc = pn;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) unknown += __shfl_down(unknown, d, 64);
  if ((threadIdx.x & 63) == 0 && unknown) atomicAdd(&Q.counters[0], unknown);
  tg_lc_fold_to<true>(lds, F.plan, F.st, F.fold.pending + (size_t)blockIdx.x * F.fold.stride * 3);
  // the page's verdict goes to the host without a copy in the stream (a copy between two pages' launches keeps the second one waiting):
  // the workgroup that arrives last stores the counters into host memory and raises the flag the host polls
  if (Q.host_out) {
    __syncthreads();   // (every wave's counter atomics have been acknowledged)
    if (threadIdx.x == 0 && atomicAdd(Q.done, 1u) + 1u == gridDim.x) {
      __hip_atomic_store(Q.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      Q.host_out[0] = __hip_atomic_load(&Q.counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      Q.host_out[1] = __hip_atomic_load(&Q.counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      Q.host_out[2] = __hip_atomic_load(&Q.counters[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      __hip_atomic_store(&Q.host_out[7], 1ULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
#undef FQ_PAGE
#undef FQ_TILES
}
extern "C" __global__ void __launch_bounds__(256) fq_onepass(FqArgs Q) { fq_body<false>(Q); }
extern "C" __global__ void __launch_bounds__(256) fq_onepass_multi(FqArgs Q) { fq_body<true>(Q); }
)SRC";

// host mirror of the generated FaArgs
struct FaArgsHost {
    FpArgs fp;
    const int32_t *gids;
    unsigned char *mask_out;
    struct St {
        int32_t function;
        int32_t pad;
        long long *counts;
        long long *limbs;
        unsigned int *special;
        unsigned long long *i128;
        double *dsum;
    } st[16];
    struct Plan {
        int32_t n_aggs, n_wide;
        int32_t wide_slot[16];
        int32_t n_cnt;
        int32_t cnt_slot[16];
        int32_t count_from_rows[16];
        int32_t rows_slot;
        int32_t per_group_bytes, n_groups;
    } plan;
    long long tiles;
    int32_t lowcard;
    int32_t pad;
    const unsigned int *ord_keys;
    const int *ord_rows;
    const unsigned char *gids8;
    struct {
        unsigned long long *partials;
        int32_t stride;
        unsigned long long *pending;
    } fold;
    const unsigned long long *gate;
    const int *ord_stretch;
    long long *ord_list;
    unsigned int *ord_list_count;
    int32_t ord_handoff;
    int32_t pad4;
};
// host mirror of the generated FqArgs (fq_onepass)
struct FqArgsHost {
    FaArgsHost fa;
    KeyCols store;
    int32_t store_groups;
    int32_t pad;
    unsigned long long *counters;
    const unsigned long long *prev;
    unsigned long long *host_out;
    unsigned int *done;
    const FpArgs *pages;
    int32_t n_pages;
    int32_t pad3;
};

}  // namespace

FusedAggGpu::FusedAggGpu(std::vector<int32_t> input_types, const tgpu_page_processor_spec *spec, std::vector<tgpu_agg_spec> aggs,
                         std::vector<int32_t> group_by_channels)
    : input_types_(std::move(input_types)), aggs_(std::move(aggs))
{
    TG_CHECK_ARG(spec != nullptr, "page processor spec is null");
    nodes_.assign(spec->nodes, spec->nodes + spec->node_count);
    if (spec->string_pool && spec->string_pool_len > 0) pool_.assign(spec->string_pool, spec->string_pool + spec->string_pool_len);
    filter_root_ = spec->filter_root;
    proj_roots_.assign(spec->projection_roots, spec->projection_roots + spec->projection_count);
    for (int32_t r : proj_roots_) {
        TG_CHECK_ARG(r >= 0 && r < (int)nodes_.size(), "projection root out of range");
        proj_types_.push_back(nodes_[(size_t)r].type);
    }
    TG_CHECK_ARG((int)aggs_.size() <= 16, "at most 16 aggregates");
    supported_ = true;
    const int np = (int)proj_roots_.size();
    for (auto &a : aggs_) {
        TG_CHECK_ARG(a.function >= TGPU_AGG_COUNT_ALL && a.function <= TGPU_AGG_MAX_DOUBLE, "unknown aggregate function");
        if (a.function != TGPU_AGG_COUNT_ALL) {
            TG_CHECK_ARG(a.input_channel >= 0 && a.input_channel < np, "aggregate input channel out of range");
            const int32_t t = proj_types_[(size_t)a.input_channel];
            const bool want_bigint = a.function == TGPU_AGG_SUM_BIGINT || a.function == TGPU_AGG_AVG_BIGINT || a.function == TGPU_AGG_MIN_BIGINT || a.function == TGPU_AGG_MAX_BIGINT;
            const bool want_double = a.function == TGPU_AGG_SUM_DOUBLE || a.function == TGPU_AGG_AVG_DOUBLE || a.function == TGPU_AGG_MIN_DOUBLE || a.function == TGPU_AGG_MAX_DOUBLE;
            TG_CHECK_ARG(!(want_bigint && t != TGPU_BIGINT) && !(want_double && t != TGPU_DOUBLE), "aggregate input type mismatch");
            if (a.function == TGPU_AGG_COUNT_COLUMN && t == TGPU_VARCHAR) supported_ = supported_ && nodes_[(size_t)proj_roots_[(size_t)a.input_channel]].kind == TGPU_EX_INPUT;
        }
        if (a.mask_channel >= 0) TG_CHECK_ARG(a.mask_channel < np && proj_types_[(size_t)a.mask_channel] == TGPU_BOOLEAN, "aggregate mask must be a BOOLEAN channel");
    }
    // Equivalence with FilterAndProject -> HashAggregation needs every projection that can raise to be evaluated on every
    // selected row; the fused kernel evaluates exactly the aggregates' inputs on exactly those rows, so any OTHER projection
    // that can raise (dropped by the aggregation) keeps the unfused composition.
    std::function<bool(int)> can_raise = [&](int idx) -> bool {
        const tgpu_expr_node &nd = nodes_[(size_t)idx];
        if (nd.kind == TGPU_EX_CALL) {
            const bool int_result = nd.type == TGPU_BIGINT || nd.type == TGPU_INTEGER;
            if (int_result && nd.op >= TGPU_OP_ADD && nd.op <= TGPU_OP_NEGATE) return true;
            if (nd.op == TGPU_OP_CAST && nd.n_args == 1 && int_result) {   // narrowing casts are checked (BIGINT -> INTEGER, DOUBLE -> BIGINT / INTEGER)
                const int32_t from = nodes_[(size_t)nd.args[0]].type;
                if (from == TGPU_DOUBLE || (from == TGPU_BIGINT && nd.type == TGPU_INTEGER)) return true;
            }
        }
        if (nd.kind == TGPU_EX_CALL || nd.kind == TGPU_EX_SPECIAL)
            for (int k = 0; k < nd.n_args; k++)
                if (can_raise(nd.args[k])) return true;
        return false;
    };
    std::set<int> used;
    for (auto &a : aggs_) {
        if (a.function != TGPU_AGG_COUNT_ALL) used.insert(a.input_channel);
        if (a.mask_channel >= 0) used.insert(a.mask_channel);
    }
    // nothing the accumulate kernels evaluate can raise (TPCH Q1: double arithmetic only): their error word is never written and
    // the per-page read-back of it is skipped
    accumulate_can_raise_ = false;
    for (int ch : used) accumulate_can_raise_ = accumulate_can_raise_ || can_raise(proj_roots_[(size_t)ch]);
    for (int ch = 0; ch < np; ch++)
        if (can_raise(proj_roots_[(size_t)ch])) {
            // a masked aggregate would skip evaluating its input on masked-out rows; an unused channel is never evaluated
            bool always_evaluated = used.count(ch) > 0;
            for (auto &a : aggs_)
                if (a.function != TGPU_AGG_COUNT_ALL && a.input_channel == ch && a.mask_channel >= 0) always_evaluated = false;
            if (!always_evaluated) supported_ = false;
        }
    // state sharing: aggregates over the same (input expression, mask) share their lane-private partials -- sum(x) and avg(x)
    // need one (hi, lo) pair and one count between them (they still own separate global states)
    n_wide_ = 0;
    n_cnt_ = 0;
    std::map<std::tuple<int, int, int>, int> wide_of, cnt_of;
    std::function<void(int, std::set<int> &)> inputs_of = [&](int idx, std::set<int> &out) {
        const tgpu_expr_node &nd = nodes_[(size_t)idx];
        if (nd.kind == TGPU_EX_INPUT) out.insert(nd.op);
        if (nd.kind == TGPU_EX_CALL || nd.kind == TGPU_EX_SPECIAL)
            for (int k = 0; k < nd.n_args; k++) inputs_of(nd.args[k], out);
    };
    std::function<bool(int)> never_null_given_inputs = [&](int idx) -> bool {
        // true when the expression is null only if one of its input columns is (no null literal, no IF / COALESCE subtleties)
        const tgpu_expr_node &nd = nodes_[(size_t)idx];
        if (nd.kind == TGPU_EX_INPUT) return true;
        if (nd.kind == TGPU_EX_CONST) return !nd.is_null;
        if (nd.kind == TGPU_EX_SPECIAL) return false;
        for (int k = 0; k < nd.n_args; k++)
            if (!never_null_given_inputs(nd.args[k])) return false;
        return true;
    };
    for (size_t k = 0; k < aggs_.size(); k++) {
        const tgpu_agg_spec &a = aggs_[k];
        // (min / max keep no lane-private sum either: their rows go straight to the state word, device_agg.h tg_minmax_update)
        const bool count_only = a.function == TGPU_AGG_COUNT_ALL || a.function == TGPU_AGG_COUNT_COLUMN || (a.function >= TGPU_AGG_MIN_BIGINT && a.function <= TGPU_AGG_MAX_DOUBLE);
        const int in_root = a.function == TGPU_AGG_COUNT_ALL ? -1 : proj_roots_[(size_t)a.input_channel];
        const int mask_root = a.mask_channel >= 0 ? proj_roots_[(size_t)a.mask_channel] : -1;
        const int kind = a.function == TGPU_AGG_SUM_BIGINT ? 1 : (a.function == TGPU_AGG_AVG_BIGINT ? 2 : 0);
        if (count_only) wide_slot_[k] = -1;
        else {
            auto key = std::make_tuple(in_root, mask_root, kind);
            auto it = wide_of.find(key);
            if (it == wide_of.end()) it = wide_of.emplace(key, n_wide_++).first;
            wide_slot_[k] = it->second;
        }
        auto ckey = std::make_tuple(in_root, mask_root, 0);
        auto ct = cnt_of.find(ckey);
        if (ct == cnt_of.end()) {
            ct = cnt_of.emplace(ckey, n_cnt_++).first;
            std::set<int> ins;
            bool simple = true;
            if (in_root >= 0) {
                inputs_of(in_root, ins);
                simple = never_null_given_inputs(in_root);
            }
            cnt_inputs_.push_back(simple ? std::vector<int>(ins.begin(), ins.end()) : std::vector<int>{-1});
            cnt_masked_.push_back(mask_root >= 0);
        }
        cnt_slot_[k] = ct->second;
    }
    rows_slot_ = n_cnt_;   // one extra count slot: rows of the group
    per_group_bytes_ = n_wide_ * 2 * 256 * 8 + (n_cnt_ + 1) * 256 * 4;
    max_groups_ = per_group_bytes_ > 0 ? (160 * 1024 - 64) / per_group_bytes_ : 0;  // 64 B: headroom for the compiler's own LDS use
    if (aggs_.empty()) supported_ = false;
    for (int32_t ch : group_by_channels) {
        TG_CHECK_ARG(ch >= 0 && ch < np, "group-by channel out of range");
        const int raw = identity_channel(ch);
        if (raw < 0) supported_ = false;   // computed group-by keys keep the unfused composition
        key_inputs_.push_back(raw);
    }
    if ((int)key_inputs_.size() > kMaxKeyChannels) supported_ = false;
    if (!supported_) key_inputs_.clear();
    if (supported_) generate();
}

FusedAggGpu::~FusedAggGpu() {}

int FusedAggGpu::identity_channel(int ch) const
{
    const tgpu_expr_node &nd = nodes_[(size_t)proj_roots_[(size_t)ch]];
    return nd.kind == TGPU_EX_INPUT ? nd.op : -1;
}

void FusedAggGpu::generate()
{
    auto cols_decl = [&](const Gen &g) {
        std::ostringstream cols;
        for (int ch : g.used_cols) {
            const int32_t t = input_types_[(size_t)ch];
            const char *T = (t == TGPU_VARCHAR || t == TGPU_BOOLEAN) ? "unsigned char" : ctype(t);
            cols << "  const " << T << "* c" << ch << " = (const " << T << "*)A.col_values[" << ch << "]; (void)c" << ch << ";\n";
            cols << "  const unsigned char* cn" << ch << " = FA_NO_NULLS ? (const unsigned char*)0 : A.col_nulls[" << ch << "]; (void)cn" << ch << ";\n";
            if (t == TGPU_VARCHAR) cols << "  const int* co" << ch << " = A.col_offsets[" << ch << "]; (void)co" << ch << ";\n";
        }
        return cols.str();
    };
    // (a) the filter in memory mode (pass A)
    Gen gm(nodes_, pool_, input_types_);
    std::ostringstream filter_fn;
    if (filter_root_ >= 0) {
        Val f = gm.gen(filter_root_, 1);
        if (f.type != TGPU_BOOLEAN) gm.bad("filter must be boolean");
        filter_fn << "__device__ inline bool tg_filter_mem(const FpArgs& A, long long row) {\n" << cols_decl(gm) << gm.os.str() << "  return !" << f.n << " && " << f.v << ";\n}\n";
    }
    else filter_fn << "__device__ inline bool tg_filter_mem(const FpArgs& A, long long row) { return true; }\n";

    // (b) the aggregates' masks and inputs in register-row mode (pass C).  Phase 1 evaluates every aggregate's (take, value)
    // pair into registers; the low-cardinality variant then reads ALL the lane's LDS slots, updates them in registers and
    // writes them back (one LDS round trip per row instead of one per aggregate); the global variant issues the exact atomics.
    Gen gr(nodes_, pool_, input_types_);
    gr.reg_mode = true;
    gr.tmp = gm.tmp;
    std::ostringstream eval, lc_read, lc_upd, lc_write, gl, nf_any, nf_slow, nf_clear, ord_decl, ord_upd, ord_write, ch_decl, ch_upd, ch_write, ch_sum;
    int ord_doubles = 0;   // DOUBLE sums = sequential chains of the ORDERED mode's few-group kernel (fa_ordered_chain)
    for (size_t k = 0; k < aggs_.size(); k++) {
        const tgpu_agg_spec &a = aggs_[k];
        const int w = wide_slot_[k];
        const bool is_dbl = a.function == TGPU_AGG_SUM_DOUBLE || a.function == TGPU_AGG_AVG_DOUBLE || a.function == TGPU_AGG_AVG_BIGINT;
        const bool is_big = a.function == TGPU_AGG_SUM_BIGINT;
        const bool is_mm = a.function >= TGPU_AGG_MIN_BIGINT && a.function <= TGPU_AGG_MAX_DOUBLE;
        const bool mm_dbl = a.function == TGPU_AGG_MIN_DOUBLE || a.function == TGPU_AGG_MAX_DOUBLE;
        const std::string mm_word = "&F.st[" + std::to_string(k) + "].i128[(size_t)g * 2]";
        const std::string mm_code = "tg_minmax_encode(" + std::to_string(a.function) + ", (unsigned long long)y" + std::to_string(k) + ")";
        eval << "  bool t" << k << " = true; double x" << k << " = 0.0; long long y" << k << " = 0; (void)x" << k << "; (void)y" << k << ";\n  {\n";
        gr.os.str("");
        if (a.mask_channel >= 0) {
            Val m = gr.gen(proj_roots_[(size_t)a.mask_channel], 2);
            eval << gr.os.str() << "    t" << k << " = !" << m.n << " && " << m.v << ";\n";
            gr.os.str("");
        }
        if (a.function != TGPU_AGG_COUNT_ALL) {
            eval << "    if (t" << k << ") {\n";
            Val v = gr.gen(proj_roots_[(size_t)a.input_channel], 3);
            eval << gr.os.str();
            gr.os.str("");
            eval << "      if (" << v.n << ") t" << k << " = false;\n";
            if (is_dbl) eval << "      else x" << k << " = " << (a.function == TGPU_AGG_AVG_BIGINT ? "(double)" : "") << v.v << ";\n";
            else if (mm_dbl) eval << "      else y" << k << " = __double_as_longlong(" << v.v << ");\n";
            else if (is_big || is_mm) eval << "      else y" << k << " = " << v.v << ";\n";
            eval << "    }\n";
        }
        eval << "  }\n";
        if (is_dbl) {  // NaN / +-inf are flagged on the group, not summed (the flags decide the result at evaluation time)
            eval << "  const bool nf" << k << " = t" << k << " && !(fabs(x" << k << ") <= 1.7976931348623157e308);\n";
            nf_any << (nf_any.str().empty() ? "" : " || ") << "nf" << k;
            nf_slow << "    if (nf" << k << ") tg_flag_special(&F.st[" << k << "].special[g], x" << k << ");\n";
            nf_clear << "  x" << k << " = nf" << k << " ? 0.0 : x" << k << ";\n";
        }
        // low-cardinality: lane-private slots.  Shared states are updated once (by the first aggregate that owns them).
        const int cs = cnt_slot_[k];
        bool first_cnt = true, first_wide = true;
        for (size_t j = 0; j < k; j++) {
            if (cnt_slot_[j] == cs) first_cnt = false;
            if (w >= 0 && wide_slot_[j] == w) first_wide = false;
        }
        if (first_cnt) {
            // without nulls in the page the count of an unmasked aggregate over a never-null expression IS the row count: static
            const bool static_rows = !cnt_masked_[(size_t)cs] && !(cnt_inputs_[(size_t)cs].size() == 1 && cnt_inputs_[(size_t)cs][0] < 0);
            const std::string cfr = "(FA_NO_NULLS ? " + std::string(static_rows ? "1" : "0") + " : F.plan.count_from_rows[" + std::to_string(k) + "])";
            lc_read << "  unsigned int c" << cs << " = 0; if (!" << cfr << ") c" << cs << " = cnt_base[" << cs << " * 256 + threadIdx.x];\n";
            lc_upd << "  c" << cs << " += t" << k << " ? 1u : 0u;\n";
            lc_write << "  if (!" << cfr << ") cnt_base[" << cs << " * 256 + threadIdx.x] = c" << cs << ";\n";
        }
        if (is_mm) lc_upd << "  if (t" << k << ") tg_minmax_update(" << mm_word << ", " << mm_code << ");\n";
        if (is_dbl && first_wide) {
            lc_read << "  double h" << w << " = hi_base[" << w << " * 256 + threadIdx.x], l" << w << " = lo_base[" << w << " * 256 + threadIdx.x];\n";
            lc_upd << "  { const double v_ = t" << k << " ? x" << k << " : 0.0; const double s_ = h" << w << " + v_; const double bb_ = s_ - h" << w << "; l" << w
                   << " += (h" << w << " - (s_ - bb_)) + (v_ - bb_); h" << w << " = s_; }\n";
            lc_write << "  hi_base[" << w << " * 256 + threadIdx.x] = h" << w << "; lo_base[" << w << " * 256 + threadIdx.x] = l" << w << ";\n";
        }
        else if (is_big && first_wide) {
            lc_read << "  unsigned long long bl" << w << " = ((unsigned long long*)hi_base)[" << w << " * 256 + threadIdx.x]; long long bh" << w << " = ((long long*)lo_base)[" << w
                    << " * 256 + threadIdx.x];\n";
            lc_upd << "  { const long long v_ = t" << k << " ? y" << k << " : 0; const unsigned long long n_ = bl" << w << " + (unsigned long long)v_; bh" << w
                   << " += (v_ < 0 ? -1 : 0) + (n_ < bl" << w << " ? 1 : 0); bl" << w << " = n_; }\n";
            lc_write << "  ((unsigned long long*)hi_base)[" << w << " * 256 + threadIdx.x] = bl" << w << "; ((long long*)lo_base)[" << w << " * 256 + threadIdx.x] = bh" << w << ";\n";
        }
        // ORDERED mode: the rows of one group are added by one lane, in row order, into plain locals
        ord_decl << "    long long oc" << k << " = 0;";
        if (is_dbl) ord_decl << " double os" << k << " = F.st[" << k << "].dsum[g];";
        if (is_big) ord_decl << " __int128 ob" << k << " = 0;";
        if (is_mm) ord_decl << " unsigned long long om" << k << " = 0;";
        ord_decl << "\n";
        ord_upd << "      if (t" << k << ") { oc" << k << "++;";
        if (is_dbl) ord_upd << " os" << k << " += x" << k << ";";
        if (is_big) ord_upd << " ob" << k << " += y" << k << ";";
        if (is_mm) ord_upd << " { const unsigned long long c_ = " << mm_code << "; om" << k << " = c_ > om" << k << " ? c_ : om" << k << "; }";
        ord_upd << " }\n";
        ord_write << "    if (oc" << k << ") F.st[" << k << "].counts[g] += oc" << k << ";\n";
        if (is_dbl) ord_write << "    F.st[" << k << "].dsum[g] = os" << k << ";\n";
        if (is_mm) ord_write << "    if (om" << k << ") tg_minmax_update(" << mm_word << ", om" << k << ");\n";
        if (is_big)
            ord_write << "    if (ob" << k << " != 0) { unsigned long long* p_ = &F.st[" << k << "].i128[(size_t)g * 2]; const unsigned __int128 n_ = (((unsigned __int128)p_[1] << 64) | p_[0]) + (unsigned __int128)ob"
                      << k << "; p_[0] = (unsigned long long)n_; p_[1] = (unsigned long long)(n_ >> 64); }\n";
        // ORDERED mode, few groups (fa_ordered_chain): the producer lanes count and add the integers (any order gives the same bits) and
        // hand the doubles to the chain lane of the aggregate -- a row the aggregate skips travels as -0.0, the identity of IEEE addition
        ch_decl << "  long long oc" << k << " = 0;";
        if (is_big) ch_decl << " __int128 ob" << k << " = 0;";
        if (is_mm) ch_decl << " unsigned long long om" << k << " = 0;";
        ch_decl << "\n";
        ch_upd << "      oc" << k << " += t" << k << " ? 1 : 0;";
        if (is_big) ch_upd << " if (t" << k << ") ob" << k << " += y" << k << ";";
        if (is_mm) ch_upd << " if (t" << k << ") { const unsigned long long c_ = " << mm_code << "; om" << k << " = c_ > om" << k << " ? c_ : om" << k << "; }";
        if (is_dbl) ch_upd << " out[" << ord_doubles << " * TG_ORD_STRIDE] = t" << k << " ? x" << k << " : -0.0;";
        ch_upd << "\n";
        ch_write << "    if (oc" << k << ") atomicAdd((unsigned long long*)&F.st[" << k << "].counts[g], (unsigned long long)oc" << k << ");\n";
        if (is_big) ch_write << "    if (ob" << k << " != 0) tg_i128_add_wide(&F.st[" << k << "].i128[(size_t)g * 2], ob" << k << ");\n";
        if (is_mm) ch_write << "    if (om" << k << ") tg_minmax_update(" << mm_word << ", om" << k << ");\n";
        if (is_dbl) {
            ch_sum << "    if (lane == " << ord_doubles << ") sum = F.st[" << k << "].dsum;\n";
            ord_doubles++;
        }
        // global: exact atomics per row
        gl << "  if (t" << k << ") {\n    atomicAdd((unsigned long long*)&F.st[" << k << "].counts[g], 1ULL);\n";
        if (is_dbl) gl << "    tg_kulisch_add(&F.st[" << k << "].limbs[(size_t)g * TG_LIMBS], &F.st[" << k << "].special[g], x" << k << ");\n";
        else if (is_big) gl << "    tg_i128_add(&F.st[" << k << "].i128[(size_t)g * 2], y" << k << ");\n";
        else if (is_mm) gl << "    tg_minmax_update(" << mm_word << ", " << mm_code << ");\n";
        gl << "  }\n";
    }

    // non-finite inputs are rare: one branch per row guards the per-aggregate flag updates, the values are cleared by selects
    std::string eval_all = eval.str();
    if (!nf_any.str().empty()) eval_all += "  if (" + nf_any.str() + ") {\n" + nf_slow.str() + "  }\n" + nf_clear.str();

    std::ostringstream src;
    src << kPrelude;
    if (const char *exp = getenv("TGPU_FG_EXP")) src << "#define FG_EXP_" << exp << " 1\n";  // kernel-study switch, never set in production
    if (const char *st = getenv("TGPU_FG_STRIPES")) src << "#define FG_STRIPES " << std::max(1, std::min(16, atoi(st))) << "\n";  // rows per lane per tile of fg_probe
    if (getenv("TGPU_FA_DEBUG_SKIP")) src << "#define FA_DEBUG_SKIP " << atoi(getenv("TGPU_FA_DEBUG_SKIP")) << "\n";   // kernel study only
    src << device_header("device_hash.h") << device_header("device_agg.h") << gm.consts.str() << gr.consts.str();
    src << "#define FA_LDS_BYTES " << std::max(2048, max_groups_ * per_group_bytes_) << "\n";   // >= the fold's exchange area (device_agg.h)
    // FA_NO_NULLS 1: the specialisation for pages without null vectors (null loads and per-aggregate count slots fold away)
    src << "#ifndef FA_NO_NULLS\n#define FA_NO_NULLS 0\n#endif\n";
    src << "struct TgRow {\n";
    for (int ch : gr.reg_cols) {
        const int32_t t = input_types_[(size_t)ch];
        src << "  " << (t == TGPU_BOOLEAN ? "unsigned char" : ctype(t)) << " c" << ch << "; unsigned char n" << ch << ";\n";
    }
    if (gr.reg_cols.empty()) src << "  int unused;\n";
    src << "};\n__device__ inline void tg_load_row(const FpArgs& A, long long row, TgRow& R) {\n";
    for (int ch : gr.reg_cols) {
        const int32_t t = input_types_[(size_t)ch];
        const char *T = t == TGPU_BOOLEAN ? "unsigned char" : ctype(t);
        src << "  R.c" << ch << " = ((const " << T << "*)A.col_values[" << ch << "])[row]; R.n" << ch << " = (!FA_NO_NULLS && A.col_nulls[" << ch << "]) ? A.col_nulls[" << ch << "][row] : 0;\n";
    }
    src << "  (void)A; (void)row; (void)R;\n}\n__device__ inline void tg_zero_row(TgRow& R) {\n";
    for (int ch : gr.reg_cols) src << "  R.c" << ch << " = 0; R.n" << ch << " = 0;\n";
    src << "  (void)R;\n}\n";
    src << filter_fn.str();
    if (!key_inputs_.empty()) {
        // key accessor generated for this key schema (the GPU counterpart of JoinCompiler's hashRow / positionNotDistinctFromRow)
        src << device_header("device_cols.h") << device_header("device_groupby.h");
        src << "struct FgArgs {\n  FpArgs fp;\n  TgKeyCols store;\n  unsigned long long* words;\n  unsigned long long mask;\n  int* out;\n  unsigned long long* counters;\n"
               "  long long row0;\n  long long n;\n  int store_groups;\n  int pad;\n  const FgArgs* self;\n  unsigned char* out8;\n};\n";
        auto cell = [&](int i, const std::string &row, const std::string &pfx) {
            // declares <pfx>n (null flag) and the cell's value variables for key column i of the raw input at `row`
            const int ch = key_inputs_[(size_t)i];
            const int32_t t = input_types_[(size_t)ch];
            std::ostringstream o;
            o << "    const bool " << pfx << "n = !FA_NO_NULLS && A.col_nulls[" << ch << "] && A.col_nulls[" << ch << "][" << row << "];\n";
            if (t == TGPU_VARCHAR)
                o << "    const int " << pfx << "a = A.col_offsets[" << ch << "][" << row << "]; const int " << pfx << "l = A.col_offsets[" << ch << "][" << row << " + 1] - " << pfx
                  << "a; const unsigned char* " << pfx << "p = (const unsigned char*)A.col_values[" << ch << "] + " << pfx << "a;\n";
            else {
                const char *T = t == TGPU_BOOLEAN ? "unsigned char" : (t == TGPU_DOUBLE ? "unsigned long long" : ctype(t));
                o << "    const " << T << " " << pfx << "v = ((const " << T << "*)A.col_values[" << ch << "])[" << row << "];\n";
            }
            return o.str();
        };
        auto eq = [&](int i, const std::string &x, const std::string &y) {
            // value equality of two non-null cells (DOUBLE compared as IS NOT DISTINCT: NaN == NaN, -0 == +0)
            const int32_t t = input_types_[(size_t)key_inputs_[(size_t)i]];
            std::ostringstream o;
            if (t == TGPU_VARCHAR)
                o << "    { if (" << x << "l != " << y << "l) return false; for (int j_ = 0; j_ < " << x << "l; j_++) if (" << x << "p[j_] != " << y << "p[j_]) return false; }\n";
            else if (t == TGPU_DOUBLE)
                o << "    { const double u_ = __longlong_as_double((long long)" << x << "v), w_ = __longlong_as_double((long long)" << y << "v); if (!((u_ != u_ && w_ != w_) || u_ == w_)) return false; }\n";
            else if (t == TGPU_BOOLEAN) o << "    if ((" << x << "v != 0) != (" << y << "v != 0)) return false;\n";
            else o << "    if (" << x << "v != " << y << "v) return false;\n";
            return o.str();
        };
        src << "struct SpecKeys {\n  const FpArgs& A; const TgKeyCols& S; long long row0;\n";
        src << "  __device__ long long hash(long long r) const {\n    const long long row = row0 + r; tg_i64 h = 0;\n";
        for (size_t i = 0; i < key_inputs_.size(); i++) {
            const int32_t t = input_types_[(size_t)key_inputs_[i]];
            src << "   {\n" << cell((int)i, "row", "x") << "    tg_i64 c = 0;\n    if (!xn) c = ";
            switch (t) {
            case TGPU_BIGINT: src << "tg_hash_long(xv)"; break;
            case TGPU_INTEGER: case TGPU_DATE: src << "tg_hash_int(xv)"; break;
            case TGPU_DOUBLE: src << "tg_hash_double_bits(xv)"; break;
            case TGPU_BOOLEAN: src << "tg_hash_boolean(xv)"; break;
            default: src << "(tg_i64)tg_xxh64(xp, xl)"; break;
            }
            src << ";\n    h = tg_combine_hash(h, c);\n   }\n";
        }
        src << "    return h;\n  }\n";
        src << "  __device__ bool eq_store(long long r, int g) const {\n    const long long row = row0 + r;\n";
        for (size_t i = 0; i < key_inputs_.size(); i++) {
            const int32_t t = input_types_[(size_t)key_inputs_[i]];
            src << "   {\n" << cell((int)i, "row", "x");
            src << "    const bool yn = S.c[" << i << "].nulls && S.c[" << i << "].nulls[g];\n";
            if (t == TGPU_VARCHAR)
                src << "    const int ya = S.c[" << i << "].offsets[g]; const int yl = S.c[" << i << "].offsets[g + 1] - ya; const unsigned char* yp = (const unsigned char*)S.c[" << i << "].values + ya;\n";
            else {
                const char *T = t == TGPU_BOOLEAN ? "unsigned char" : (t == TGPU_DOUBLE ? "unsigned long long" : ctype(t));
                src << "    const " << T << " yv = ((const " << T << "*)S.c[" << i << "].values)[g];\n";
            }
            src << "    if (xn || yn) { if (xn != yn) return false; } else\n" << eq((int)i, "x", "y") << "   }\n";
        }
        src << "    return true;\n  }\n";
        src << "  __device__ bool eq_row(long long r, long long r2) const {\n    const long long row = row0 + r, row2 = row0 + r2;\n";
        for (size_t i = 0; i < key_inputs_.size(); i++) {
            src << "   {\n" << cell((int)i, "row", "x") << cell((int)i, "row2", "y");
            src << "    if (xn || yn) { if (xn != yn) return false; } else\n" << eq((int)i, "x", "y") << "   }\n";
        }
        src << "    return true;\n  }\n};\n";
        // block-local copy of the first groups' keys in LDS (few groups = TPCH Q1): a row is compared against those records
        // (LDS broadcast reads) before it ever touches the table.  Record per (group, key column): 32 bytes =
        // data[16] (fixed-width value / first 16 varchar bytes) | len int | null byte | long byte (varchar longer than 16 bytes)
        src << "#define FG_LDS_GROUPS 16\n#define FG_NKEYS " << key_inputs_.size() << "\n";
        src << "__device__ inline void fg_build_records(const TgKeyCols& S, int groups, unsigned char* rec) {\n"
               "  for (int idx = threadIdx.x; idx < groups * FG_NKEYS; idx += 256) {\n    const int g = idx / FG_NKEYS, i = idx % FG_NKEYS;\n"
               "    unsigned char* r = rec + (size_t)idx * 32;\n    const TgColView& c = S.c[i];\n"
               "    const bool nl = c.nulls && c.nulls[g];\n    r[20] = nl ? 1 : 0; r[21] = 0; *(int*)(r + 16) = 0;\n"
               "    for (int j = 0; j < 16; j++) r[j] = 0;\n    if (nl) continue;\n"
               "    if (c.type == 6) {\n      const int a = c.offsets[g], l = c.offsets[g + 1] - a;\n      *(int*)(r + 16) = l;\n      if (l > 16) r[21] = 1;\n"
               "      else for (int j = 0; j < l; j++) r[j] = ((const unsigned char*)c.values)[a + j];\n    }\n"
               "    else if (c.type == 1 || c.type == 4) *(unsigned long long*)r = ((const unsigned long long*)c.values)[g];\n"
               "    else if (c.type == 2 || c.type == 3) *(long long*)r = (long long)((const int*)c.values)[g];\n"
               "    else *(unsigned long long*)r = ((const unsigned char*)c.values)[g] != 0 ? 1ULL : 0ULL;\n  }\n  __syncthreads();\n}\n";
        // pre-loaded key cells of one row (phase A: null flags, fixed-width values, varchar offset/length; phase B: the first
        // varchar byte), so that the loads of all the stripes of a tile are in flight together
        src << "struct TgKeyRow {\n";
        for (size_t i = 0; i < key_inputs_.size(); i++) {
            const int32_t t = input_types_[(size_t)key_inputs_[i]];
            if (t == TGPU_VARCHAR) src << "  int a" << i << ", e" << i << ", l" << i << "; unsigned char n" << i << ", b" << i << ";\n";
            else src << "  " << (t == TGPU_BOOLEAN ? "unsigned char" : (t == TGPU_DOUBLE ? "unsigned long long" : ctype(t))) << " v" << i << "; unsigned char n" << i << ";\n";
        }
        // phase A is loads only (no arithmetic on the loaded values): the loads of all stripes stay in flight together
        // (addresses = uniform tile base + 32-bit lane offset: scalar base + 32-bit VGPR offset addressing, no 64-bit lane math)
        src << "};\n__device__ inline void fg_load_key_a(const FpArgs& A, long long tb, unsigned int o, TgKeyRow& K) {\n";
        for (size_t i = 0; i < key_inputs_.size(); i++) {
            const int ch = key_inputs_[i];
            const int32_t t = input_types_[(size_t)ch];
            src << "  K.n" << i << " = (!FA_NO_NULLS && A.col_nulls[" << ch << "]) ? (A.col_nulls[" << ch << "] + tb)[o] : 0;\n";
            if (t == TGPU_VARCHAR)
                src << "  K.a" << i << " = (A.col_offsets[" << ch << "] + tb)[o]; K.e" << i << " = (A.col_offsets[" << ch << "] + tb)[o + 1u];\n";
            else {
                const char *T = t == TGPU_BOOLEAN ? "unsigned char" : (t == TGPU_DOUBLE ? "unsigned long long" : ctype(t));
                src << "  K.v" << i << " = ((const " << T << "*)A.col_values[" << ch << "] + tb)[o];\n";
            }
        }
        src << "}\n__device__ inline void fg_key_lengths(TgKeyRow& K) {\n";
        for (size_t i = 0; i < key_inputs_.size(); i++)
            if (input_types_[(size_t)key_inputs_[i]] == TGPU_VARCHAR) src << "  K.l" << i << " = K.e" << i << " - K.a" << i << ";\n";
        src << "  (void)K;\n}\n__device__ inline void fg_zero_key(TgKeyRow& K) {\n";
        for (size_t i = 0; i < key_inputs_.size(); i++) {
            const int32_t t = input_types_[(size_t)key_inputs_[i]];
            if (t == TGPU_VARCHAR) src << "  K.a" << i << " = 0; K.e" << i << " = 0; K.l" << i << " = 0; K.n" << i << " = 1; K.b" << i << " = 0;\n";
            else src << "  K.v" << i << " = 0; K.n" << i << " = 1;\n";
        }
        // phase B is unconditional too (a branch around a load makes the compiler wait for every load in flight): a row that
        // has no first byte to fetch reads byte 0 of the column (the host guarantees one readable byte) and ignores it
        src << "}\n__device__ inline void fg_load_key_b(const FpArgs& A, TgKeyRow& K, bool live) {\n";
        for (size_t i = 0; i < key_inputs_.size(); i++) {
            const int ch = key_inputs_[i];
            if (input_types_[(size_t)ch] == TGPU_VARCHAR)
                src << "  K.b" << i << " = ((const unsigned char*)A.col_values[" << ch << "])[(live && !K.n" << i << " && K.l" << i << " > 0) ? (unsigned int)K.a" << i << " : 0u];\n";
        }
        src << "  (void)live;\n";
        src << "  (void)A; (void)K;\n}\n";
        // row vs LDS record: 1 equal, 0 different, -1 cannot decide here (long varchar: the table path decides)
        src << "__device__ inline int fg_eq_record(const FpArgs& A, const TgKeyRow& K, const unsigned char* rec, int g) {\n";
        for (size_t i = 0; i < key_inputs_.size(); i++) {
            const int ch = key_inputs_[i];
            const int32_t t = input_types_[(size_t)ch];
            src << "  {\n    const unsigned char* r = rec + ((size_t)g * FG_NKEYS + " << i << ") * 32;\n";
            src << "    const bool xn = K.n" << i << " != 0, yn = r[20] != 0;\n    if (xn || yn) { if (xn != yn) return 0; } else {\n";
            if (t == TGPU_VARCHAR) {
                src << "      if (K.l" << i << " != *(const int*)(r + 16)) return 0;\n      if (r[21]) return -1;\n      if (K.l" << i << " > 0 && K.b" << i << " != r[0]) return 0;\n"
                    << "      for (int j_ = 1; j_ < K.l" << i << "; j_++) if (((const unsigned char*)A.col_values[" << ch << "])[K.a" << i << " + j_] != r[j_]) return 0;\n";
            }
            else if (t == TGPU_DOUBLE)
                src << "      { const double u_ = __longlong_as_double((long long)K.v" << i << "), w_ = __longlong_as_double(*(const long long*)r); if (!((u_ != u_ && w_ != w_) || u_ == w_)) return 0; }\n";
            else if (t == TGPU_BOOLEAN) src << "      if ((K.v" << i << " != 0) != (*(const unsigned long long*)r != 0)) return 0;\n";
            else src << "      if ((long long)K.v" << i << " != *(const long long*)r) return 0;\n";
            src << "    }\n  }\n";
        }
        src << "  (void)A;\n  return 1;\n}\n";
        // The first FG_REG_GROUPS records additionally live in registers as SIGNATURES: bit fields [null mask | 32 bits per
        // INTEGER / DATE key | 1 per BOOLEAN | 64 per BIGINT / DOUBLE key (DOUBLE canonicalised: one NaN, -0 -> +0) | 10 per VARCHAR
        // key = min(length, 2) << 8 | first byte] packed into as few words as they need (one 32-bit word for two varchar keys).
        // Equal signatures decide the row unless it has a varchar key longer than one byte ("open": the byte-wise LDS comparison
        // decides -- which is also why lengths above one need not be told apart here).  A record that cannot be summarised
        // (varchar longer than 16 bytes) has first byte 0 in the record: at worst the row comes out "open" and the byte-wise path
        // sends it to the table.
        {
            struct Field { std::string expr; int bits; };
            // greedy packing, a field never straddles a 64-bit word
            auto pack = [&](const std::vector<Field> &f, const std::string &dst, int &words, bool &narrow) {
                std::vector<std::string> w;
                int used = 64, total = 0;
                for (const Field &x : f) {
                    total += x.bits;
                    if (used + x.bits > 64) { w.push_back(""); used = 0; }
                    std::string &cur = w.back();
                    if (!cur.empty()) cur += " | ";
                    cur += "((unsigned long long)(" + x.expr + ") << " + std::to_string(used) + ")";
                    used += x.bits;
                }
                words = (int)w.size();
                narrow = words == 1 && total <= 32;
                std::ostringstream o;
                for (size_t j = 0; j < w.size(); j++) o << "  " << dst << "[" << j << "] = (FG_SIG_T)(" << w[j] << ");\n";
                return o.str();
            };
            std::ostringstream row_sig, rec_sig;
            std::vector<Field> fr{{"nm", (int)key_inputs_.size()}}, fg{{"nm", (int)key_inputs_.size()}};
            row_sig << "  open = false;\n  unsigned int nm = 0;\n";
            rec_sig << "  unsigned int nm = 0;\n";
            for (size_t i = 0; i < key_inputs_.size(); i++) {
                const int32_t t = input_types_[(size_t)key_inputs_[i]];
                const std::string I = std::to_string(i);
                row_sig << "  const bool n" << I << " = K.n" << I << " != 0; nm |= n" << I << " ? " << (1u << i) << "u : 0u;\n";
                rec_sig << "  const unsigned char* r" << I << " = rec + ((size_t)g * FG_NKEYS + " << I << ") * 32;\n  const bool n" << I << " = r" << I << "[20] != 0; nm |= n" << I
                        << " ? " << (1u << i) << "u : 0u;\n";
                if (t == TGPU_VARCHAR) {
                    row_sig << "  const unsigned int w" << I << " = n" << I << " ? 0u : (((unsigned int)(K.l" << I << " < 2 ? K.l" << I << " : 2) << 8) | (K.l" << I << " > 0 ? (unsigned int)K.b" << I
                            << " : 0u));\n  open = open || (!n" << I << " && K.l" << I << " > 1);\n";
                    rec_sig << "  const int l" << I << " = *(const int*)(r" << I << " + 16);\n  const unsigned int w" << I << " = n" << I << " ? 0u : (((unsigned int)(l" << I << " < 2 ? l" << I
                            << " : 2) << 8) | (l" << I << " > 0 ? (unsigned int)r" << I << "[0] : 0u));\n";
                    fr.push_back({"w" + I, 10});
                    fg.push_back({"w" + I, 10});
                }
                else if (t == TGPU_BIGINT || t == TGPU_DOUBLE) {
                    row_sig << "  const unsigned long long q" << I << " = n" << I << " ? 0ULL : "
                            << (t == TGPU_DOUBLE ? "fg_canon_double((unsigned long long)K.v" + I + ")" : "(unsigned long long)K.v" + I) << ";\n";
                    rec_sig << "  const unsigned long long q" << I << " = n" << I << " ? 0ULL : "
                            << (t == TGPU_DOUBLE ? "fg_canon_double(*(const unsigned long long*)r" + I + ")" : "*(const unsigned long long*)r" + I) << ";\n";
                    fr.push_back({"q" + I, 64});
                    fg.push_back({"q" + I, 64});
                }
                else if (t == TGPU_BOOLEAN) {
                    row_sig << "  const unsigned int w" << I << " = n" << I << " ? 0u : (K.v" << I << " != 0 ? 1u : 0u);\n";
                    rec_sig << "  const unsigned int w" << I << " = n" << I << " ? 0u : (*(const unsigned long long*)r" << I << " != 0 ? 1u : 0u);\n";
                    fr.push_back({"w" + I, 1});
                    fg.push_back({"w" + I, 1});
                }
                else {
                    row_sig << "  const unsigned int w" << I << " = n" << I << " ? 0u : (unsigned int)K.v" << I << ";\n";
                    rec_sig << "  const unsigned int w" << I << " = n" << I << " ? 0u : (unsigned int)*(const unsigned long long*)r" << I << ";\n";
                    fr.push_back({"w" + I, 32});
                    fg.push_back({"w" + I, 32});
                }
            }
            int words = 1;
            bool narrow = false;
            const std::string row_pack = pack(fr, "sig", words, narrow), rec_pack = pack(fg, "R.s", words, narrow);
            src << "#define FG_REG_GROUPS 4\n#define FG_SIG_WORDS " << words << "\ntypedef " << (narrow ? "unsigned int" : "unsigned long long")
                << " FG_SIG_T;\nstruct TgRecReg { FG_SIG_T s[FG_SIG_WORDS]; };\n";
            src << "__device__ inline unsigned long long fg_canon_double(unsigned long long bits) {\n  const double u = __longlong_as_double((long long)bits);\n"
                   "  return u != u ? 0x7ff8000000000000ULL : (u == 0.0 ? 0ULL : bits);\n}\n";
            src << "__device__ inline void fg_row_sig(const TgKeyRow& K, FG_SIG_T* sig, bool& open) {\n" << row_sig.str() << row_pack << "}\n";
            src << "__device__ inline void fg_load_recreg(const unsigned char* rec, int g, TgRecReg& R) {\n" << rec_sig.str() << rec_pack << "}\n";
        }
        // (c) the filter in register-row mode for the group probe: its fixed-width input columns are loaded in phase A
        {
            Gen gf(nodes_, pool_, input_types_);
            gf.reg_mode = true;
            gf.tmp = gr.tmp + 1000;
            std::ostringstream body;
            if (filter_root_ >= 0) {
                Val f = gf.gen(filter_root_, 1);
                body << cols_decl(gf) << gf.os.str() << "  return !" << f.n << " && " << f.v << ";\n";
            }
            else body << "  return true;\n";
            src << gf.consts.str() << "struct TgFRow {\n";
            for (int ch : gf.reg_cols) {
                const int32_t t = input_types_[(size_t)ch];
                src << "  " << (t == TGPU_BOOLEAN ? "unsigned char" : ctype(t)) << " c" << ch << "; unsigned char n" << ch << ";\n";
            }
            if (gf.reg_cols.empty()) src << "  int unused;\n";
            src << "};\n__device__ inline void tg_load_frow(const FpArgs& A, long long tb, unsigned int o, TgFRow& R) {\n";
            for (int ch : gf.reg_cols) {
                const int32_t t = input_types_[(size_t)ch];
                const char *T = t == TGPU_BOOLEAN ? "unsigned char" : ctype(t);
                src << "  R.c" << ch << " = ((const " << T << "*)A.col_values[" << ch << "] + tb)[o]; R.n" << ch << " = (!FA_NO_NULLS && A.col_nulls[" << ch << "]) ? (A.col_nulls[" << ch
                    << "] + tb)[o] : 0;\n";
            }
            src << "  (void)A; (void)tb; (void)o; (void)R;\n}\n__device__ inline void tg_zero_frow(TgFRow& R) {\n";
            for (int ch : gf.reg_cols) src << "  R.c" << ch << " = 0; R.n" << ch << " = 0;\n";
            src << "  (void)R;\n}\n__device__ inline bool tg_filter_f(const FpArgs& A, long long row, const TgFRow& R) {\n" << body.str() << "}\n";
        }
        src << R"SRC(
#ifndef FG_STRIPES
#define FG_STRIPES 8
#endif
#define FG_TILE (FG_STRIPES * 256)
// the table path (hashing + probe / insert protocol) is rare once the first groups are cached in LDS: it is kept out of line
// so that the hot loop stays small enough for the instruction cache
// It reads its arguments from the kernel-argument segment in memory (__builtin_amdgcn_kernarg_segment_ptr): taking the address of the by-value
// kernel parameter instead would make the compiler spill the whole argument block to scratch and turn every column access
// of the hot loop into scratch + flat loads.  Returns the group result in the low word, the "pending" flag in bit 32.
__device__ __attribute__((noinline)) long long fg_table_path(const FgArgs* Gm, long long r, int store_groups) {
  const FgArgs& G = *Gm;
  SpecKeys K{G.fp, G.store, G.row0};
  bool pending = false;
  const int result = tg_gbh_probe<true>(K, r, G.words, G.mask, store_groups, G.counters, pending);
  return (long long)(unsigned int)result | (pending ? (1LL << 32) : 0LL);
}
// a row the register signatures did not decide: the byte-wise comparison against the LDS records of groups [from, lg), then
// the table (store_groups 0 skips its own key-store scan unless a long varchar key left a cached group undecided).  Out of
// line for the same reason; the key cells travel by value.
__device__ __attribute__((noinline)) long long fg_slow_path(const FgArgs* Gm, const unsigned char* rec, TgKeyRow K, long long r, int from, int lg, bool redo) {
  const FgArgs& G = *Gm;
  bool undecided = redo;
  for (int g = from; g < lg; g++) {
    const int e = fg_eq_record(G.fp, K, rec, g);
    if (e > 0) return (long long)(unsigned int)g;
    undecided = undecided || e < 0;
  }
  return fg_table_path(Gm, r, undecided ? G.store_groups : 0);
}
// pass B: filter + group lookup / insert (protocol: device_groupby.h).  Eight rows per lane per tile, processed in phases so
// that the loads of all eight rows are in flight together: (A) filter column + key cells, (B) first varchar bytes,
// (C) compare against the LDS copies of the first groups' keys; only a row that matches none of them touches the table.
extern "C" __global__ void __launch_bounds__(256) fg_probe(FgArgs G) {
  const FpArgs& A = G.fp;
  // the out-of-line paths read the arguments from memory: the kernel-argument segment itself (FgArgs is the only parameter, so it
  // starts the segment) -- no host-side copy of the block per launch
  const FgArgs* fg_self = (const FgArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ __attribute__((aligned(16))) unsigned char rec[FG_LDS_GROUPS * FG_NKEYS * 32];
  const int lg = G.store_groups < FG_LDS_GROUPS ? G.store_groups : FG_LDS_GROUPS;
  fg_build_records(G.store, lg, rec);
  TgRecReg rr[FG_REG_GROUPS];
#pragma unroll
  for (int g = 0; g < FG_REG_GROUPS; g++) {
    fg_load_recreg(rec, g < lg ? g : 0, rr[g]);
    // an unused register slot gets a signature no row has (every key null AND value bits set): no "g < rg" test per row
    if (g >= lg) rr[g].s[0] = ~(FG_SIG_T)0;
  }
  const int rg = lg < FG_REG_GROUPS ? lg : FG_REG_GROUPS;
  const long long tiles = (G.n + FG_TILE - 1) / FG_TILE;
  unsigned long long npending = 0;
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    // rows of the tile = uniform base + 32-bit lane offset (scalar-base addressing: no 64-bit lane arithmetic in the hot loop)
    const long long t0 = tile * FG_TILE;
    const long long tb = G.row0 + t0;
    const long long left = G.n - t0;
    const unsigned int lim = (unsigned int)(left < FG_TILE ? left : FG_TILE) - 1u;   // the tile's last row
    bool sel[FG_STRIPES];
    TgKeyRow kr[FG_STRIPES];
    {
      TgFRow fr[FG_STRIPES];
#pragma unroll
      for (int s = 0; s < FG_STRIPES; s++) {   // phase A: loads only, unconditional (rows past the end re-read the last row)
        const unsigned int o = threadIdx.x + s * 256;
        const unsigned int oc = o < lim ? o : lim;
        fg_zero_key(kr[s]);
        tg_load_frow(A, tb, oc, fr[s]);
        fg_load_key_a(A, tb, oc, kr[s]);
      }
#pragma unroll
      for (int s = 0; s < FG_STRIPES; s++) {
        const unsigned int o = threadIdx.x + s * 256;
        sel[s] = o <= lim && tg_filter_f(A, tb + o, fr[s]);
        fg_key_lengths(kr[s]);
      }
    }
#pragma unroll
    for (int s = 0; s < FG_STRIPES; s++) fg_load_key_b(A, kr[s], sel[s]);   // phase B
#pragma unroll
    for (int s = 0; s < FG_STRIPES; s++) {
      const unsigned int o = threadIdx.x + s * 256;
      // phase C, branch-free for the common case: the row's signature against the register copies of the first groups
      FG_SIG_T sig[FG_SIG_WORDS];
      bool open;
      fg_row_sig(kr[s], sig, open);
      int result = -1;
#pragma unroll
      for (int g = FG_REG_GROUPS - 1; g >= 0; g--) {   // distinct groups have distinct keys: at most one signature matches
        bool same = true;
#pragma unroll
        for (int j = 0; j < FG_SIG_WORDS; j++) same = same && sig[j] == rr[g].s[j];
        result = same ? g : result;
      }
      const bool redo = open && result >= 0;   // a longer varchar key must be compared byte-wise: from group 0 below
      result = (redo || !sel[s]) ? -1 : result;
      if (sel[s] && result < 0) {
        const long long tp = fg_slow_path(fg_self, rec, kr[s], t0 + o, redo ? 0 : rg, lg, redo);
        result = (int)(unsigned int)(tp & 0xffffffffLL);
        npending += (unsigned long long)(tp >> 32);
      }
      if (o <= lim) {
        // compact mode: group id + 1 in one byte (0 = filtered row); 255 = a group that is new in this sub-batch (the host
        // re-runs the sub-batch in int32 mode then)
        if (G.out8) (G.out8 + t0)[o] = result < -1 ? (unsigned char)255 : (unsigned char)(result + 1);
        else (G.out + t0)[o] = result;
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) npending += __shfl_down(npending, d, 64);
  if ((threadIdx.x & 63) == 0 && npending) atomicAdd(&G.counters[0], npending);
}
)SRC";
    }
    src << "struct FaArgs;\n";
    // forward declare FaArgs fields used by the generated accumulate function: emit the struct first
    std::string kernels = kFaKernels;
    const size_t split = kernels.find("#define FA_STRIPES");
    src << kernels.substr(0, split);
    src << "__device__ inline void tg_accumulate_row_lc(const FaArgs& F, const FpArgs& A, long long row, const TgRow& R, int g, unsigned char* lds) {\n" << cols_decl(gr)
        << "  double* hi_base = tg_lc_hi(lds, F.plan, g); double* lo_base = tg_lc_lo(lds, F.plan, g);\n"
        << "  unsigned int* cnt_base = tg_lc_cnt(lds, F.plan, g);\n  (void)hi_base; (void)lo_base; (void)row;\n"
        << eval_all << lc_read.str() << "  unsigned int rows_ = cnt_base[" << rows_slot_ << " * 256 + threadIdx.x];\n" << lc_upd.str() << "  rows_ += 1u;\n" << lc_write.str()
        << "  cnt_base[" << rows_slot_ << " * 256 + threadIdx.x] = rows_;\n}\n";
    src << "__device__ inline void tg_accumulate_row_gl(const FaArgs& F, const FpArgs& A, long long row, const TgRow& R, int g) {\n" << cols_decl(gr) << "  (void)row;\n"
        << eval_all << gl.str() << "}\n";
    // many groups, ORDERED mode (agg.h): the lane that owns a group's first row in (group, row) order walks the group's rows and
    // adds them in row order -- the order of the reference's per-position loop -- into plain doubles (NaN / inf flow through
    // the additions as in Java: no special flags here)
    src << "__device__ inline void tg_accumulate_group_ordered(const FaArgs& F, const FpArgs& A, long long i, long long n) {\n" << cols_decl(gr)
        << "  const unsigned int key = F.ord_keys[i];\n  const long long g = (long long)key - 1;\n" << ord_decl.str()
        << "  for (long long j = i; j < n && F.ord_keys[j] == key; j++) {\n"
        << "    if (F.ord_handoff && j - i == F.ord_handoff) {   // a long group: the rest goes to a workgroup of fa_ordered_chain (state so far: below)\n"
        << "      const unsigned int at = atomicAdd(F.ord_list_count, 1u);\n      F.ord_list[(size_t)at * 2] = g;\n      F.ord_list[(size_t)at * 2 + 1] = j;\n      break;\n    }\n"
        << "    const long long row = F.ord_rows[j];\n    TgRow R;\n    tg_load_row(A, row, R);\n"
        << eval.str() << ord_upd.str() << "  }\n" << ord_write.str() << "}\n";
    // few groups, ORDERED mode: one workgroup per group.  Waves 1..15 evaluate the group's rows, 960 at a time, into LDS; wave 0 holds one
    // chain per DOUBLE sum (lane d = the d-th such aggregate) and adds the tile before, value after value in row order, while the next one
    // is produced.  The additions of one sum are a dependent chain whatever the machine: this keeps that chain free of loads and
    // expression work (the old lane-per-group loop paid a row load and the projections between two additions).
    ord_doubles_ = ord_doubles;
    src << "__device__ inline void tg_accumulate_group_chained(const FaArgs& F, const FpArgs& A, long long g, long long s, long long e, double* vals) {\n" << cols_decl(gr)
        << "  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;\n" << ch_decl.str()
        << "  double* sum = (double*)0;\n  if (wave == 0) {\n    __builtin_amdgcn_s_setprio(3);\n" << ch_sum.str() << "  }\n  double os = sum ? sum[g] : 0.0;\n"
        << "  const long long tiles = (e - s + TG_ORD_TILE - 1) / TG_ORD_TILE;\n"
        << "  const int r = (wave - 1) * 64 + lane;\n"
        << "  long long row_next = (wave > 0 && s + r < e) ? F.ord_rows[s + r] : 0;   // (the row number of the next tile is on its way while this one is evaluated)\n"
        << "  for (long long t = 0; t <= tiles; t++) {\n"
        << "    if (wave > 0 && t < tiles) {\n      const long long j = s + t * TG_ORD_TILE + r;\n      const long long row = row_next;\n"
        << "      if (j + TG_ORD_TILE < e) row_next = F.ord_rows[j + TG_ORD_TILE];\n"
        << "      double* out = vals + (t & 1) * (TG_ORD_MAX_DOUBLES * TG_ORD_STRIDE) + r; (void)out;\n"
        << "      if (j < e) {\n      TgRow R;\n      tg_load_row(A, row, R);\n" << eval.str() << ch_upd.str() << "      }\n    }\n"
        << "    else if (wave == 0 && t > 0 && sum) {\n      const long long left = e - (s + (t - 1) * TG_ORD_TILE);\n"
        << "      os = tg_chain_add_tile(vals + ((t - 1) & 1) * (TG_ORD_MAX_DOUBLES * TG_ORD_STRIDE) + lane * TG_ORD_STRIDE, left < TG_ORD_TILE ? (int)left : TG_ORD_TILE, os);\n    }\n"
        << "    __syncthreads();\n  }\n  if (sum) sum[g] = os;\n  if (wave > 0) {\n" << ch_write.str() << "  }\n}\n";
    std::string tail = kernels.substr(split);
    const std::string tag = "@FA_STRIPES@";
    tail.replace(tail.find(tag), tag.size(), std::to_string(fa_stripes()));
    src << tail;
    if (!key_inputs_.empty()) src << "#define FQ_LDS_BYTES " << std::max(2048, onepass_groups() * per_group_bytes_) << "\n" << kFqKernel;
    source_ = src.str();
}

// the specialisation for pages without null vectors: same source, FA_NO_NULLS 1
static std::string no_nulls_source(const std::string &src) { return "#define FA_NO_NULLS 1\n" + src; }
// the specialisation for compact (one byte per row) group ids
static std::string gid8_source(const std::string &src) { return "#define FA_GID8 1\n" + src; }

void FusedAggGpu::precompile()
{
    if (!supported_) return;
    (void)code_object_for(source_);
    (void)code_object_for(no_nulls_source(source_));
    (void)code_object_for(gid8_source(no_nulls_source(source_)));
}

void FusedAggGpu::ensure_loaded()
{
    if (!module_) module_ = load_module(source_);
}

JitModule *FusedAggGpu::module_for(const DevicePage &in, bool gid8)
{
    bool nulls = false;
    for (const DeviceColumn &c : in.cols) nulls = nulls || c.nulls != nullptr;
    return module_variant(nulls, gid8);
}

JitModule *FusedAggGpu::module_variant(bool nulls, bool gid8)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (gid8) {
        std::shared_ptr<JitModule> &m = nulls ? module_g8_ : module_nn_g8_;
        if (!m) m = load_module(gid8_source(nulls ? source_ : no_nulls_source(source_)));
        return m.get();
    }
    if (nulls) {
        ensure_loaded();
        return module_.get();
    }
    if (!module_nn_) module_nn_ = load_module(no_nulls_source(source_));
    return module_nn_.get();
}

void FusedAggGpu::raise_if_error(Context *ctx, BufferPtr &err)
{
    raise_expression_error(ctx->read_scalar(err->as<unsigned long long>()));
}

static void fill_fp_cols(FpArgs &fp, const DevicePage &in)
{
    for (size_t i = 0; i < in.cols.size() && i < (size_t)kFpMaxCols; i++) {
        fp.col_values[i] = in.cols[i].values;
        fp.col_nulls[i] = in.cols[i].nulls;
        fp.col_offsets[i] = in.cols[i].offsets;
    }
    fp.n = in.n;
}

void FusedAggGpu::filter_mask(Context *ctx, const DevicePage &in, uint8_t *mask_out)
{
    TG_CHECK_STATE(supported_, "fused aggregation not supported for this configuration");
    JitModule *module = module_for(in);
    if (in.n == 0) return;
    FaArgsHost F{};
    fill_fp_cols(F.fp, in);
    BufferPtr err = ctx->alloc(8);
    HIP_CHECK(hipMemsetAsync(err->ptr(), 0xff, 8, ctx->stream()));
    F.fp.error = err->as<unsigned long long>();
    F.mask_out = mask_out;
    {
        ProfileScope ps(ctx, "fused_filter_mask");
        int64_t blocks = std::min<int64_t>(ceil_div(in.n, 256), (int64_t)ctx->cu_count() * 16);
        launch_args(module->fn("fa_mask"), (int)blocks, F, ctx->stream());
    }
    raise_if_error(ctx, err);
}

namespace {
struct FgArgsHost {
    FpArgs fp;
    KeyCols store;
    unsigned long long *words;
    unsigned long long mask;
    int32_t *out;
    unsigned long long *counters;
    long long row0;
    long long n;
    int32_t store_groups;
    int32_t pad;
    const void *self;   // device copy of this block (read by the out-of-line table path)
    unsigned char *out8;   // compact mode: one byte per row instead of `out` (GbhProbeLaunch)
};
}  // namespace

void FusedAggGpu::probe_groups(Context *ctx, const DevicePage &in, const GbhProbeLaunch &l)
{
    TG_CHECK_STATE(supported_ && !key_inputs_.empty(), "fused group lookup not available for this configuration");
    JitModule *module = module_for(in);
    FgArgsHost G{};
    fill_fp_cols(G.fp, in);
    G.fp.error = l.counters + 7;   // the group-by table reads it back together with its own counters (GbhProbeLaunch)
    // the kernel reads byte 0 of a varchar key column for rows without a first byte: give an all-empty column one to read
    BufferPtr dummy;
    for (int ch : key_inputs_)
        if (input_types_[(size_t)ch] == TGPU_VARCHAR && G.fp.col_values[ch] == nullptr) {
            if (!dummy) {
                dummy = ctx->alloc(16);
                HIP_CHECK(hipMemsetAsync(dummy->ptr(), 0, 16, ctx->stream()));
            }
            G.fp.col_values[ch] = dummy->ptr();
        }
    G.store = l.store;
    G.words = (unsigned long long *)l.words;
    G.mask = l.mask;
    G.out = l.out;
    G.out8 = l.out8;
    G.counters = l.counters;
    G.row0 = l.row0;
    G.n = l.n;
    G.store_groups = l.store_groups;
    G.self = nullptr;   // (the kernel takes its argument block from the kernarg segment)
    {
        ProfileScope ps(ctx, "fused_filter_group_probe");
        // persistent grid: exactly the resident workgroups, each walking tiles with a grid stride (no second wave of blocks)
        const int64_t blocks = std::min<int64_t>(ceil_div(l.n, 256), (int64_t)ctx->cu_count() * module->blocks_per_cu("fg_probe"));
        launch_args(module->fn("fg_probe"), (int)blocks, G, ctx->stream());
    }
}

// One pass per launch (fq_onepass / fq_onepass_multi): see the kernel.  `blocks` is fixed per operator (one workgroup row of pending totals per
// workgroup).  Several pages: their column pointers travel as an array of FpArgs behind one small upload (kept alive by `keep`).
void FusedAggGpu::onepass(Context *ctx, const std::vector<const DevicePage *> &pages, GroupedAccumulators &accs, const KeyCols &store, int64_t groups,
                          unsigned long long *counters, const unsigned long long *prev, int64_t blocks, unsigned long long *host_out, BufferPtr *keep)
{
    TG_CHECK_STATE(supported_ && !key_inputs_.empty() && groups > 0 && groups <= max_groups_ && !accumulate_can_raise_, "one-pass aggregation not available for this configuration");
    TG_CHECK_ARG(!pages.empty(), "one-pass launch without a page");
    const DevicePage &in = *pages[0];
    bool any_nulls = false;
    for (const DevicePage *pg : pages)
        for (const DeviceColumn &c : pg->cols) any_nulls = any_nulls || c.nulls != nullptr;
    JitModule *module = module_variant(any_nulls, false);   // (the variant without null handling only when no page of the launch carries a null vector)
    accs.reserve(max_groups_);
    FqArgsHost Q{};
    FaArgsHost &F = Q.fa;
    BufferPtr dummy;
    auto fill = [&](FpArgs &fp, const DevicePage &pg) {
        fill_fp_cols(fp, pg);
        fp.error = counters + 7;
        for (int ch : key_inputs_)
            if (input_types_[(size_t)ch] == TGPU_VARCHAR && fp.col_values[ch] == nullptr) {
                if (!dummy) {
                    dummy = ctx->alloc(16);
                    HIP_CHECK(hipMemsetAsync(dummy->ptr(), 0, 16, ctx->stream()));
                }
                fp.col_values[ch] = dummy->ptr();
            }
    };
    fill(F.fp, in);
    for (size_t k = 0; k < aggs_.size(); k++) {
        GroupedAccumulators::DeviceState d = accs.device_state((int)k);
        F.st[k].function = d.function;
        F.st[k].counts = d.counts;
        F.st[k].limbs = d.limbs;
        F.st[k].special = d.special;
        F.st[k].i128 = d.i128;
        F.st[k].dsum = d.dsum;
    }
    F.plan.n_aggs = (int32_t)aggs_.size();
    F.plan.n_wide = n_wide_;
    F.plan.n_cnt = n_cnt_ + 1;
    F.plan.rows_slot = rows_slot_;
    for (size_t k = 0; k < aggs_.size(); k++) {
        F.plan.wide_slot[k] = wide_slot_[k];
        F.plan.cnt_slot[k] = cnt_slot_[k];
        const int cs = cnt_slot_[k];
        bool from_rows = !cnt_masked_[(size_t)cs];
        for (const DevicePage *pg : pages)
            for (int ch : cnt_inputs_[(size_t)cs]) from_rows = from_rows && ch >= 0 && pg->cols[(size_t)ch].nulls == nullptr;
        F.plan.count_from_rows[k] = from_rows ? 1 : 0;
    }
    F.plan.per_group_bytes = per_group_bytes_;
    F.lowcard = 1;
    F.plan.n_groups = (int32_t)groups;
    F.tiles = ceil_div(in.n, 8 * 256);
    const GroupedAccumulators::FoldScratch fs = accs.fold_scratch(blocks, max_groups_);
    F.fold.partials = fs.partials;
    F.fold.stride = fs.stride;
    F.fold.pending = fs.pending;
    Q.store = store;
    Q.store_groups = (int32_t)groups;
    Q.counters = counters;
    Q.prev = prev;
    Q.host_out = host_out;
    Q.done = host_out ? reinterpret_cast<unsigned int *>(static_cast<unsigned long long *>(ctx->zeroed_scratch()) + 1) : nullptr;
    const bool multi = pages.size() > 1;
    if (multi) {
        std::vector<FpArgs> desc(pages.size());
        for (size_t i = 0; i < pages.size(); i++) {
            TG_CHECK_ARG(pages[i]->n > 0, "empty page in a multi-page launch");
            desc[i] = FpArgs{};
            fill(desc[i], *pages[i]);
        }
        BufferPtr d = ctx->alloc(desc.size() * sizeof(FpArgs));
        ctx->upload(d->ptr(), desc.data(), desc.size() * sizeof(FpArgs));
        Q.pages = d->as<FpArgs>();
        Q.n_pages = (int32_t)pages.size();
        if (keep) *keep = d;
    }
    if (getenv("TGPU_DEBUG_LAUNCH")) {   // kernel studies: what a launch was given (stderr)
        fprintf(stderr, "[tgpu] fq_onepass%s pages=%zu groups=%lld blocks=%lld prev=%p counters=%p host_out=%p desc=%p\n", multi ? "_multi" : "", pages.size(), (long long)groups,
                (long long)blocks, (const void *)prev, (void *)counters, (void *)host_out, (const void *)Q.pages);
        for (size_t i = 0; i < pages.size(); i++) {
            fprintf(stderr, "[tgpu]   page %zu n=%lld", i, (long long)pages[i]->n);
            for (const DeviceColumn &c : pages[i]->cols) fprintf(stderr, " (%p %p %p)", c.values, (const void *)c.nulls, (const void *)c.offsets);
            fprintf(stderr, "\n");
        }
    }
    ProfileScope ps(ctx, "fused_filter_group_accumulate_onepass");
    launch_args(module->fn(multi ? "fq_onepass_multi" : "fq_onepass"), (int)blocks, Q, ctx->stream());
}

// gids8 -> gids, for the (rare) page whose ids arrived compact but whose groups do not fit the lane-private LDS path
static __global__ void __launch_bounds__(256) widen_gids_kernel(const unsigned char *gids8, long long n, int *gids)
{
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < n; r += (long long)gridDim.x * 256) gids[r] = (int)gids8[r] - 1;
}

void FusedAggGpu::accumulate(Context *ctx, const DevicePage &in, const int32_t *gids, const uint8_t *gids8, int64_t groups, GroupedAccumulators &accs,
                             const unsigned long long *gate)
{
    TG_CHECK_STATE(supported_, "fused aggregation not supported for this configuration");
    if (in.n == 0) return;
    BufferPtr widened;
    if (gids8 && (groups > max_groups_ || accs.force_ordered())) {   // the ORDERED / exact-global paths take int32 ids
        widened = ctx->alloc((size_t)in.n * 4);
        widen_gids_kernel<<<(int)std::min<int64_t>(ceil_div(in.n, 256), (int64_t)ctx->cu_count() * 8), 256, 0, ctx->stream()>>>(gids8, in.n, widened->as<int>());
        check_launch("widen_gids");
        gids = widened->as<int32_t>();
        gids8 = nullptr;
    }
    JitModule *module = module_for(in, gids8 != nullptr);
    BufferPtr ord_keys, ord_rows;
    const bool ordered = accs.begin_ordered(gids, in.n, groups, max_groups_, ord_keys, ord_rows);
    // (low-cardinality launches: the states of every group the folded partials have room for, see GroupedAccumulators::fold_scratch)
    if (!ordered) accs.reserve(groups <= max_groups_ ? max_groups_ : groups);
    FaArgsHost F{};
    fill_fp_cols(F.fp, in);
    BufferPtr err = ctx->alloc(8);
    if (accumulate_can_raise_) HIP_CHECK(hipMemsetAsync(err->ptr(), 0xff, 8, ctx->stream()));   // (never written, never read otherwise)
    F.fp.error = err->as<unsigned long long>();
    F.gids = gids;
    F.gids8 = gids8;
    F.gate = gate;
    for (size_t k = 0; k < aggs_.size(); k++) {
        GroupedAccumulators::DeviceState d = accs.device_state((int)k);
        F.st[k].function = d.function;
        F.st[k].counts = d.counts;
        F.st[k].limbs = d.limbs;
        F.st[k].special = d.special;
        F.st[k].i128 = d.i128;
        F.st[k].dsum = d.dsum;
    }
    if (ordered) {
        F.ord_keys = ord_keys->as<unsigned int>();
        F.ord_rows = ord_rows->as<int>();
        // few groups with many rows each: one workgroup per group, the sums as chains fed from LDS (same bits as the lane-per-group kernel)
        const int64_t ids = groups > 0 ? groups : 1;
        if (ord_doubles_ <= kOrdChainMaxDoubles && ids <= ord_chain_max_groups() && in.n >= ids * kOrdChainMinRows && getenv("TGPU_DISABLE_ORDERED_CHAIN") == nullptr) {
            BufferPtr stretches = accs.group_stretches(F.ord_keys, in.n, ids);
            F.ord_stretch = stretches->as<int>();
            ProfileScope ps(ctx, "fused_project_accumulate_ordered_chain");
            launch_args(module->fn("fa_ordered_chain"), (int)ids, F, ctx->stream(), kOrdChainWaves * 64);
            if (accumulate_can_raise_) raise_if_error(ctx, err);
            return;
        }
        // one lane per group; a group of more than kOrdHandoffRows rows (skew: one key far more frequent than the others) is handed over to
        // the chained kernel after that many rows -- its lane would otherwise be the whole launch (128 ns per row)
        BufferPtr list, list_count;
        const int64_t long_groups = in.n / kOrdHandoffRows;   // at most this many groups can be that long
        const bool handoff = long_groups > 0 && ord_doubles_ <= kOrdChainMaxDoubles && getenv("TGPU_DISABLE_ORDERED_CHAIN") == nullptr;
        if (handoff) {
            list = ctx->alloc((size_t)long_groups * 16);
            list_count = ctx->alloc_zero(8);
            F.ord_list = list->as<long long>();
            F.ord_list_count = list_count->as<unsigned int>();
            F.ord_handoff = (int32_t)kOrdHandoffRows;
        }
        {
            ProfileScope ps(ctx, "fused_project_accumulate_ordered");
            const int64_t blocks = std::min<int64_t>(ceil_div(in.n, 256), (int64_t)ctx->cu_count() * 8);
            launch_args(module->fn("fa_accumulate_ordered"), (int)blocks, F, ctx->stream());
        }
        if (handoff) {
            ProfileScope ps(ctx, "fused_project_accumulate_ordered_handoff");
            launch_args(module->fn("fa_ordered_chain"), (int)std::min<int64_t>(long_groups, ctx->cu_count()), F, ctx->stream(), kOrdChainWaves * 64);
        }
        if (accumulate_can_raise_) raise_if_error(ctx, err);
        return;
    }
    F.plan.n_aggs = (int32_t)aggs_.size();
    F.plan.n_wide = n_wide_;
    F.plan.n_cnt = n_cnt_ + 1;
    F.plan.rows_slot = rows_slot_;
    for (size_t k = 0; k < aggs_.size(); k++) {
        F.plan.wide_slot[k] = wide_slot_[k];
        F.plan.cnt_slot[k] = cnt_slot_[k];
        // the aggregate counts every row of its group when it has no mask and none of the columns its input reads has nulls
        // in this page: its count is then the group's row count (one shared LDS counter instead of one per aggregate)
        const int cs = cnt_slot_[k];
        bool from_rows = !cnt_masked_[(size_t)cs];
        for (int ch : cnt_inputs_[(size_t)cs]) from_rows = from_rows && ch >= 0 && in.cols[(size_t)ch].nulls == nullptr;
        F.plan.count_from_rows[k] = from_rows ? 1 : 0;
    }
    F.plan.per_group_bytes = per_group_bytes_;
    const int64_t g = groups > 0 ? groups : 1;
    F.lowcard = (g <= max_groups_ && getenv("TGPU_DISABLE_LOWCARD") == nullptr) ? 1 : 0;
    F.plan.n_groups = F.lowcard ? (int32_t)g : 0;
    F.tiles = ceil_div(in.n, fa_stripes() * 256);
    // the LDS array is static (sized for max_groups_), so one block per CU is resident; the 8-stripe software pipeline keeps
    // ~8 x row-bytes in flight per lane, which is what saturates HBM at 4 waves per CU
    const int64_t blocks = std::min<int64_t>(F.tiles, (int64_t)ctx->cu_count() * (F.lowcard ? 1 : 4));
    if (F.lowcard) {
        const GroupedAccumulators::FoldScratch fs = accs.fold_scratch(blocks, max_groups_);
        F.fold.partials = fs.partials;
        F.fold.stride = fs.stride;
    }
    {
        ProfileScope ps(ctx, F.lowcard ? "fused_project_accumulate_lowcard" : "fused_project_accumulate");
        launch_args(module->fn(F.lowcard ? "fa_accumulate_lowcard" : "fa_accumulate_global"), (int)blocks, F, ctx->stream());
    }
    if (accumulate_can_raise_) raise_if_error(ctx, err);
}

}  // namespace tgpu
