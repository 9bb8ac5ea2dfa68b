// expr_jit.hip — SURVEY §8(f) rank 1: the expression DAG of a query step compiled at run time into ONE HIP kernel.
//
// Replaces ExpressionActions::execute (src/Interpreters/ExpressionActions.cpp:595-747), which runs every action of the DAG as a
// separate IFunction::executeImpl and materialises every intermediate column, and stands where the reference's own run-time
// compiler stands (setting compile_expressions: src/Interpreters/JIT/CHJIT.cpp, src/Interpreters/JIT/compileFunction.cpp,
// ExpressionJIT.cpp — LLVM IR per fused sub-DAG, cached by DAG hash).  Here the DAG becomes the body of a hand-written
// streaming kernel skeleton (contiguous chunk per workgroup iteration, 16..64-byte nontemporal loads issued before first use,
// wave64 shuffle reduction — the geometry of k_filter_sum), compiled for gfx950 with hiprtc and cached by source text.
//   k_map   : any set of DAG nodes -> materialised columns (ActionsDAG outputs)
//   k_fsum  : WHERE <node> + sum(<node>), count() in one pass, nothing materialised (FilterTransform + AggregatingTransform
//             without key fused behind the expression)
// Semantics restated per function (types: src/DataTypes/NumberTraits.h:40-215; comparisons: src/Core/AccurateComparison.h:20-204;
// logical: src/Functions/FunctionsLogical.h:82-140 on static_cast<bool>; arithmetic: src/Functions/FunctionBinaryArithmetic.h with
// `static_cast<Result>(a) OP b`; dates: DateTimeTransforms.h ToYearImpl/ToMonthImpl/ToYYYYMMImpl/ToDayOfMonthImpl over DayNum).
#include "chgpu_internal.h"

#include <dlfcn.h>

#include <mutex>
#include <string>

namespace
{

// ---- hiprtc, resolved at first use (no link-time dependency: the process may already hold PyTorch's copy of the library) ----
struct Rtc
{
    void * h = nullptr;
    int (*create)(void **, const char *, const char *, int, const char * const *, const char * const *) = nullptr;
    int (*compile)(void *, int, const char * const *) = nullptr;
    int (*log_size)(void *, size_t *) = nullptr;
    int (*log)(void *, char *) = nullptr;
    int (*code_size)(void *, size_t *) = nullptr;
    int (*code)(void *, char *) = nullptr;
    int (*destroy)(void **) = nullptr;
};
Rtc g_rtc;
std::mutex g_jit_mutex;
std::map<std::string, std::vector<char>> g_code_cache;                    // source text -> code object
std::map<std::pair<int, std::string>, hipModule_t> g_module_cache;        // (device, source text) -> loaded module

int rtc_load()
{
    if (g_rtc.h)
        return CHGPU_OK;
    const char * names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
    void * h = nullptr;
    for (const char * n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
            break;
    if (!h)
        return chgpu_set_error(CHGPU_ERR_NOT_IMPLEMENTED, "cannot load libhiprtc (%s): expressions stay on the CPU", dlerror());
#define SYM(field, name)                                              \
    *(void **)&g_rtc.field = dlsym(h, name);                          \
    if (!g_rtc.field)                                                 \
        return chgpu_set_error(CHGPU_ERR_DEVICE, "libhiprtc lacks %s", name);
    SYM(create, "hiprtcCreateProgram")
    SYM(compile, "hiprtcCompileProgram")
    SYM(log_size, "hiprtcGetProgramLogSize")
    SYM(log, "hiprtcGetProgramLog")
    SYM(code_size, "hiprtcGetCodeSize")
    SYM(code, "hiprtcGetCode")
    SYM(destroy, "hiprtcDestroyProgram")
#undef SYM
    g_rtc.h = h;
    return CHGPU_OK;
}

const char * ctype(int t)
{
    switch (t)
    {
        case CHGPU_I64: return "i64";
        case CHGPU_U64: return "u64";
        case CHGPU_I32: return "i32";
        case CHGPU_U32: return "u32";
        case CHGPU_I16: return "i16";
        case CHGPU_U16: return "u16";
        case CHGPU_I8: return "i8";
        case CHGPU_U8: return "u8";
        case CHGPU_F64: return "f64";
        case CHGPU_F32: return "f32";
        default: return nullptr;
    }
}

// NumberTraits::Construct<is_signed, is_floating, size> (NumberTraits.h:40-63); -1 = a type this path does not carry
int construct(bool sgn, bool flt, size_t size)
{
    if (flt)
        return size == 8 ? CHGPU_F64 : size == 4 ? CHGPU_F32 : -1;
    switch (size)
    {
        case 1: return sgn ? CHGPU_I8 : CHGPU_U8;
        case 2: return sgn ? CHGPU_I16 : CHGPU_U16;
        case 4: return sgn ? CHGPU_I32 : CHGPU_U32;
        case 8: return sgn ? CHGPU_I64 : CHGPU_U64;
        default: return -1;
    }
}
size_t next_size(size_t s) { return s < 8 ? s * 2 : s; } // NumberTraits.h:32-37
bool is_sgn(int t) { return chgpu_type_is_signed(t) || chgpu_type_is_float(t); } // is_signed_v<Float> is true

// result type of a function applied to argument types; -1 = not carried / illegal
int infer_type(int fn, int a, int b, int c)
{
    const size_t sa = a >= 0 ? chgpu_type_size(a) : 0, sb = b >= 0 ? chgpu_type_size(b) : 0;
    const bool fa = a >= 0 && chgpu_type_is_float(a), fb = b >= 0 && chgpu_type_is_float(b);
    if (fn >= CHGPU_FN_EQUALS && fn <= CHGPU_FN_GREATER_OR_EQUALS)
        return (sa && sb) ? CHGPU_U8 : -1;
    switch (fn)
    {
        case CHGPU_FN_PLUS:
        case CHGPU_FN_MULTIPLY: // ResultOfAdditionMultiplication
            return (sa && sb) ? construct(is_sgn(a) || is_sgn(b), fa || fb, next_size(sa > sb ? sa : sb)) : -1;
        case CHGPU_FN_MINUS: // ResultOfSubtraction
            return (sa && sb) ? construct(true, fa || fb, next_size(sa > sb ? sa : sb)) : -1;
        case CHGPU_FN_DIVIDE: // ResultOfFloatingPointDivision
            return (sa && sb) ? CHGPU_F64 : -1;
        case CHGPU_FN_NEGATE: // ResultOfNegate
            return sa ? construct(true, fa, is_sgn(a) ? sa : next_size(sa)) : -1;
        case CHGPU_FN_AND:
        case CHGPU_FN_OR:
        case CHGPU_FN_XOR:
            return (sa && sb) ? CHGPU_U8 : -1;
        case CHGPU_FN_NOT:
            return sa ? CHGPU_U8 : -1;
        case CHGPU_FN_BIT_AND:
        case CHGPU_FN_BIT_OR:
        case CHGPU_FN_BIT_XOR: // ResultOfBit, integers only here
            return (sa && sb && !fa && !fb) ? construct(is_sgn(a) || is_sgn(b), false, sa > sb ? sa : sb) : -1;
        case CHGPU_FN_IF: // ResultOfIf (NumberTraits.h:159-199) for the branches b, c; condition a is an integer
        {
            const size_t sc = c >= 0 ? chgpu_type_size(c) : 0;
            if (!sa || fa || !sb || !sc)
                return -1;
            if (b == c)
                return b;
            const bool fc = chgpu_type_is_float(c);
            const bool has_float = fb || fc, has_integer = !fb || !fc, has_signed = is_sgn(b) || is_sgn(c), has_unsigned = !is_sgn(b) || !is_sgn(c);
            const size_t max_u = std::max(is_sgn(b) ? (size_t)0 : sb, is_sgn(c) ? (size_t)0 : sc);
            const size_t max_s = std::max(is_sgn(b) ? sb : (size_t)0, is_sgn(c) ? sc : (size_t)0);
            const size_t max_i = std::max(fb ? (size_t)0 : sb, fc ? (size_t)0 : sc);
            const size_t max_f = std::max(fb ? sb : (size_t)0, fc ? sc : (size_t)0);
            const size_t m = std::max(sb, sc);
            const bool dbl = (has_float && has_integer && max_i >= max_f) || (has_signed && has_unsigned && max_u >= max_s);
            return construct(has_signed, has_float, dbl ? m * 2 : m); // UInt64 with Int<x>, Float<x> with [U]Int64 -> size 16 -> -1 (Error)
        }
        case CHGPU_FN_INT_DIV: // ResultOfIntegerDivision; integers only here (the float forms throw on NaN / infinities)
            return (sa && sb && !fa && !fb) ? construct(is_sgn(a) || is_sgn(b), false, sa) : -1;
        case CHGPU_FN_MODULO: // ResultOfModulo; integers only here
            return (sa && sb && !fa && !fb) ? construct(is_sgn(a), false, is_sgn(a) ? next_size(sb) : sb) : -1;
        case CHGPU_FN_TO_YEAR: return a == CHGPU_U16 ? CHGPU_U16 : -1;
        case CHGPU_FN_TO_MONTH: return a == CHGPU_U16 ? CHGPU_U8 : -1;
        case CHGPU_FN_TO_DAY_OF_MONTH: return a == CHGPU_U16 ? CHGPU_U8 : -1;
        case CHGPU_FN_TO_YYYYMM: return a == CHGPU_U16 ? CHGPU_U32 : -1;
        case CHGPU_FN_TO_YYYYMMDD: return a == CHGPU_U16 ? CHGPU_U32 : -1;
        case CHGPU_FN_TO_DAY_OF_WEEK: return a == CHGPU_U16 ? CHGPU_U8 : -1;
        case CHGPU_FN_TO_QUARTER: return a == CHGPU_U16 ? CHGPU_U8 : -1;
        case CHGPU_FN_TO_START_OF_MONTH: return a == CHGPU_U16 ? CHGPU_U16 : -1; // a Date again
        default: break;
    }
    if (fn >= CHGPU_FN_CAST && fn < CHGPU_FN_CAST + 16)
    {
        const int to = fn - CHGPU_FN_CAST;
        if (!sa || !chgpu_type_size(to))
            return -1;
        if (fa && !chgpu_type_is_float(to))
            return -1; // Float -> integer: out-of-range values are target-defined in the reference (x86 cvttsd2si); not carried
        return to;
    }
    return -1;
}

int fn_arity(int fn)
{
    if (fn == CHGPU_FN_IF)
        return 3;
    if (fn == CHGPU_FN_NEGATE || fn == CHGPU_FN_NOT || (fn >= CHGPU_FN_TO_YEAR && fn <= CHGPU_FN_TO_START_OF_MONTH) || (fn >= CHGPU_FN_CAST && fn < CHGPU_FN_CAST + 16))
        return 1;
    return 2;
}

// the operand widened without loss to the 64-bit class the exact comparison helpers take
std::string wide(int t, const std::string & e)
{
    if (chgpu_type_is_float(t))
        return "(f64)" + e;
    return (chgpu_type_is_signed(t) ? "(i64)" : "(u64)") + e;
}
std::string truth(const std::string & e) { return "(" + e + " != 0)"; }

const char * PRELUDE = R"SRC(
typedef unsigned long long u64; typedef long long i64; typedef unsigned int u32; typedef int i32;
typedef unsigned short u16; typedef short i16; typedef unsigned char u8; typedef signed char i8;
typedef float f32; typedef double f64;
#define DEV static __device__ __forceinline__
// accurate::lessOp / equalsOp (AccurateComparison.h:20-204) on operands widened to i64 / u64 / f64: mathematically exact
DEV bool nan_(i64) { return false; }
DEV bool nan_(u64) { return false; }
DEV bool nan_(f64 x) { return x != x; }
DEV bool lt_(i64 a, i64 b) { return a < b; }
DEV bool lt_(u64 a, u64 b) { return a < b; }
DEV bool lt_(f64 a, f64 b) { return a < b; }
DEV bool lt_(i64 a, u64 b) { return a < 0 || (u64)a < b; }
DEV bool lt_(u64 a, i64 b) { return b >= 0 && a < (u64)b; }
DEV bool lt_(i64 a, f64 b)
{
    if (b != b) return false;
    if (b >= 9223372036854775808.0) return true;
    if (b < -9223372036854775808.0) return false;
    const i64 t = (i64)b; // truncation, exact in range
    if (a != t) return a < t;
    return b - (f64)t > 0;
}
DEV bool lt_(f64 a, i64 b)
{
    if (a != a) return false;
    if (a >= 9223372036854775808.0) return false;
    if (a < -9223372036854775808.0) return true;
    const i64 t = (i64)a;
    if (t != b) return t < b;
    return a - (f64)t < 0;
}
DEV bool lt_(u64 a, f64 b)
{
    if (b != b) return false;
    if (b >= 18446744073709551616.0) return true;
    if (b <= 0) return false;
    const u64 t = (u64)b;
    if (a != t) return a < t;
    return b - (f64)t > 0;
}
DEV bool lt_(f64 a, u64 b)
{
    if (a != a) return false;
    if (a >= 18446744073709551616.0) return false;
    if (a < 0) return true;
    const u64 t = (u64)a;
    if (t != b) return t < b;
    return a - (f64)t < 0;
}
DEV bool eq_(i64 a, i64 b) { return a == b; }
DEV bool eq_(u64 a, u64 b) { return a == b; }
DEV bool eq_(f64 a, f64 b) { return a == b; }
DEV bool eq_(i64 a, u64 b) { return a >= 0 && (u64)a == b; }
DEV bool eq_(u64 a, i64 b) { return b >= 0 && a == (u64)b; }
DEV bool eq_(i64 a, f64 b)
{
    if (!(b >= -9223372036854775808.0 && b < 9223372036854775808.0)) return false;
    const i64 t = (i64)b;
    return t == a && (f64)t == b;
}
DEV bool eq_(f64 a, i64 b) { return eq_(b, a); }
DEV bool eq_(u64 a, f64 b)
{
    if (!(b >= 0 && b < 18446744073709551616.0)) return false;
    const u64 t = (u64)b;
    return t == a && (f64)t == b;
}
DEV bool eq_(f64 a, u64 b) { return eq_(b, a); }
template <typename A, typename B> DEV bool le_(A a, B b) { return !nan_(a) && !nan_(b) && !lt_(b, a); }
template <typename A, typename B> DEV bool ge_(A a, B b) { return !nan_(a) && !nan_(b) && !lt_(a, b); }
// DayNum -> civil date (proleptic Gregorian; what DateLUTImpl's day table holds for DayNum, Common/DateLUTImpl.h)
struct Civil { u32 y, m, d; };
DEV Civil civil_(u32 days)
{
    const u32 z = days + 719468u;
    const u32 era = z / 146097u;
    const u32 doe = z - era * 146097u;
    const u32 yoe = (doe - doe / 1460u + doe / 36524u - doe / 146096u) / 365u;
    const u32 doy = doe - (365u * yoe + yoe / 4u - yoe / 100u);
    const u32 mp = (5u * doy + 2u) / 153u;
    Civil c;
    c.d = doy - (153u * mp + 2u) / 5u + 1u;
    c.m = mp < 10u ? mp + 3u : mp - 9u;
    c.y = yoe + era * 400u + (c.m <= 2u ? 1u : 0u);
    return c;
}
)SRC";

} // namespace

struct chgpu_expr
{
    std::vector<chgpu_expr_node> nodes;
    std::vector<int> types;       // resolved type of every node
    std::vector<int> input_types; // type of cols[j] (from the INPUT nodes), -1 = unused slot
    std::string body;             // statements computing n0..nK from `r`
};

namespace
{

int build_body(chgpu_expr * e)
{
    std::string s;
    char buf[256];
    for (size_t k = 0; k < e->nodes.size(); ++k)
    {
        const chgpu_expr_node & nd = e->nodes[k];
        const int t = e->types[k];
        const char * ct = ctype(t);
        std::string rhs;
        auto N = [&](int j) { return "n" + std::to_string(nd.args[j]); };
        auto T = [&](int j) { return e->types[nd.args[j]]; };
        if (nd.kind == CHGPU_EX_INPUT)
            rhs = "r.c" + std::to_string(nd.code);
        else if (nd.kind == CHGPU_EX_CONST)
        {
            if (t == CHGPU_F64)
                snprintf(buf, sizeof(buf), "__longlong_as_double((long long)0x%llxull)", (unsigned long long)nd.bits);
            else if (t == CHGPU_F32)
                snprintf(buf, sizeof(buf), "__uint_as_float(0x%xu)", (unsigned)(nd.bits & 0xffffffffu));
            else
                snprintf(buf, sizeof(buf), "(%s)0x%llxull", ct, (unsigned long long)nd.bits);
            rhs = buf;
        }
        else
        {
            const int fn = nd.code;
            if (fn >= CHGPU_FN_EQUALS && fn <= CHGPU_FN_GREATER_OR_EQUALS)
            {
                const std::string a = wide(T(0), N(0)), b = wide(T(1), N(1));
                switch (fn)
                {
                    case CHGPU_FN_EQUALS: rhs = "eq_(" + a + ", " + b + ")"; break;
                    case CHGPU_FN_NOT_EQUALS: rhs = "!eq_(" + a + ", " + b + ")"; break;
                    case CHGPU_FN_LESS: rhs = "lt_(" + a + ", " + b + ")"; break;
                    case CHGPU_FN_GREATER: rhs = "lt_(" + b + ", " + a + ")"; break;
                    case CHGPU_FN_LESS_OR_EQUALS: rhs = "le_(" + a + ", " + b + ")"; break;
                    default: rhs = "ge_(" + a + ", " + b + ")"; break;
                }
                rhs = "(u8)(" + rhs + ")";
            }
            else if (fn == CHGPU_FN_PLUS || fn == CHGPU_FN_MINUS || fn == CHGPU_FN_MULTIPLY)
            {
                const char * op = fn == CHGPU_FN_PLUS ? "+" : fn == CHGPU_FN_MINUS ? "-" : "*";
                if (chgpu_type_is_float(t)) // always Float64: nextSize of a >= 4-byte operand
                    rhs = "(f64)" + N(0) + " " + op + " (f64)" + N(1);
                else // the result type holds both operands: two's complement arithmetic in 64 bits, truncated, is exact
                    rhs = std::string("(") + ct + ")((u64)" + N(0) + " " + op + " (u64)" + N(1) + ")";
            }
            else if (fn == CHGPU_FN_DIVIDE)
                rhs = "(f64)" + N(0) + " / (f64)" + N(1);
            else if (fn == CHGPU_FN_NEGATE)
                rhs = chgpu_type_is_float(t) ? "-" + N(0) : std::string("(") + ct + ")(0ull - (u64)" + N(0) + ")";
            else if (fn == CHGPU_FN_AND)
                rhs = "(u8)(" + truth(N(0)) + " & " + truth(N(1)) + ")";
            else if (fn == CHGPU_FN_OR)
                rhs = "(u8)(" + truth(N(0)) + " | " + truth(N(1)) + ")";
            else if (fn == CHGPU_FN_XOR)
                rhs = "(u8)(" + truth(N(0)) + " ^ " + truth(N(1)) + ")";
            else if (fn == CHGPU_FN_NOT)
                rhs = "(u8)!" + truth(N(0));
            else if (fn == CHGPU_FN_BIT_AND || fn == CHGPU_FN_BIT_OR || fn == CHGPU_FN_BIT_XOR)
            {
                const char * op = fn == CHGPU_FN_BIT_AND ? "&" : fn == CHGPU_FN_BIT_OR ? "|" : "^";
                rhs = std::string("(") + ct + ")((u64)" + N(0) + " " + op + " (u64)" + N(1) + ")";
            }
            else if (fn == CHGPU_FN_INT_DIV)
            {
                // DivideIntegralImpl::apply (src/Functions/DivisionUtils.h:66-105).  The operands keep their exact C types, so the
                // division is performed in the same promoted type as on the host (usual arithmetic conversions, LP64).
                const int ta = T(0), tb = T(1);
                if (chgpu_type_is_signed(ta) || chgpu_type_is_signed(tb))
                {
                    const int sa_t = construct(true, false, chgpu_type_size(ta));
                    const int sb_t = chgpu_type_size(ta) <= chgpu_type_size(tb) ? construct(true, false, chgpu_type_size(tb)) : sa_t;
                    rhs = std::string("(") + ct + ")((" + ctype(sa_t) + ")" + N(0) + " / (" + ctype(sb_t) + ")" + N(1) + ")";
                }
                else
                    rhs = std::string("(") + ct + ")(" + N(0) + " / " + N(1) + ")";
            }
            else if (fn == CHGPU_FN_MODULO) // ModuloImpl::apply (:126-170): IntegerAType(a) % IntegerBType(b), then the cast
                rhs = std::string("(") + ct + ")(" + N(0) + " % " + N(1) + ")";
            else if (fn == CHGPU_FN_IF)
                rhs = truth(N(0)) + " ? (" + ct + ")" + N(1) + " : (" + ct + ")" + N(2);
            else if (fn == CHGPU_FN_TO_YEAR)
                rhs = "(u16)civil_(" + N(0) + ").y";
            else if (fn == CHGPU_FN_TO_MONTH)
                rhs = "(u8)civil_(" + N(0) + ").m";
            else if (fn == CHGPU_FN_TO_DAY_OF_MONTH)
                rhs = "(u8)civil_(" + N(0) + ").d";
            else if (fn == CHGPU_FN_TO_YYYYMM)
                rhs = "(u32)(civil_(" + N(0) + ").y * 100u + civil_(" + N(0) + ").m)";
            else if (fn == CHGPU_FN_TO_YYYYMMDD)
                rhs = "(u32)(civil_(" + N(0) + ").y * 10000u + civil_(" + N(0) + ").m * 100u + civil_(" + N(0) + ").d)";
            else if (fn == CHGPU_FN_TO_DAY_OF_WEEK) // ToDayOfWeekImpl, mode 0: Monday = 1 ... Sunday = 7; 1970-01-01 was a Thursday
                rhs = "(u8)(((u32)" + N(0) + " + 3u) % 7u + 1u)";
            else if (fn == CHGPU_FN_TO_QUARTER)
                rhs = "(u8)((civil_(" + N(0) + ").m - 1u) / 3u + 1u)";
            else if (fn == CHGPU_FN_TO_START_OF_MONTH)
                rhs = "(u16)((u32)" + N(0) + " - (civil_(" + N(0) + ").d - 1u))";
            else // cast
                rhs = std::string("(") + ct + ")" + N(0);
        }
        s += std::string("    const ") + ct + " n" + std::to_string(k) + " = " + rhs + ";\n";
    }
    e->body = s;
    return CHGPU_OK;
}

// rows per lane and vector: 16-byte loads of the widest column, at least 4 bytes of the narrowest
u32 vec_rows(const std::vector<int> & types)
{
    size_t wmax = 1, wmin = 8;
    for (int t : types)
    {
        const size_t w = chgpu_type_size(t);
        wmax = std::max(wmax, w);
        wmin = std::min(wmin, w);
    }
    u32 v = (u32)std::max((size_t)16 / wmax, (size_t)4 / wmin);
    return v < 1 ? 1 : v > 16 ? 16 : v;
}

struct KernelSpec
{
    std::vector<u32> out_nodes; // k_map
    int filter_node = -1;       // k_fsum
    int value_node = -1;        // k_fsum (-1: count only)
    bool fsum = false;
    u32 vec = 1;
};

static int jit_env(const chgpu_ctx * ctx, const char * name, int dflt) { return (int)chgpu_opt(ctx, name, dflt); } // developer knobs (chgpu_ctx_set_option)
// vectors in flight per lane and column / workgroups per CU: developer overrides for A/B runs (CHGPU_TUNE_JIT_UNROLL, _WG_MAP, _WG_SUM)
// (the source generator has no context: the process-wide default, chgpu_ctx_set_option(NULL, ...), fixed at the first compilation)
static int jit_unroll()
{
    static const int v = jit_env(nullptr, "tune_jit_unroll", 4);
    return v;
}
#define JIT_UNROLL jit_unroll()
constexpr u32 JIT_MAX_COLS = 8;

struct JitArgs
{
    const void * in[JIT_MAX_COLS];
    void * out[JIT_MAX_COLS];
    u64 n;
    u64 * part; // k_fsum: {sum bits, count} per workgroup
    u32 n_parts;
    u32 pad;
};

std::string gen_source(const chgpu_expr * e, const KernelSpec & ks)
{
    std::string s = PRELUDE;
    const u32 V = ks.vec;
    auto vtype = [&](int t) { return std::string("v") + ctype(t) + "_t"; };
    // vector typedefs for every element type
    for (int t = 0; t <= CHGPU_F32; ++t)
        s += std::string("typedef ") + ctype(t) + " " + vtype(t) + " __attribute__((ext_vector_type(" + std::to_string(V) + ")));\n";
    s += "struct Args { const void * in[8]; void * out[8]; u64 n; u64 * part; u32 n_parts; u32 pad; };\n";
    s += "struct Row {";
    for (size_t j = 0; j < e->input_types.size(); ++j)
        if (e->input_types[j] >= 0)
            s += std::string(" ") + ctype(e->input_types[j]) + " c" + std::to_string(j) + ";";
    s += " };\n";
    s += "struct Res {";
    if (ks.fsum)
    {
        s += " bool keep;";
        if (ks.value_node >= 0)
            s += std::string(" ") + ctype(e->types[ks.value_node]) + " val;";
    }
    else
        for (size_t o = 0; o < ks.out_nodes.size(); ++o)
            s += std::string(" ") + ctype(e->types[ks.out_nodes[o]]) + " o" + std::to_string(o) + ";";
    s += " };\n";
    s += "DEV void eval(const Row & r, Res & o)\n{\n" + e->body;
    if (ks.fsum)
    {
        s += ks.filter_node >= 0 ? "    o.keep = n" + std::to_string(ks.filter_node) + " != 0;\n" : std::string("    o.keep = true;\n");
        if (ks.value_node >= 0)
            s += "    o.val = n" + std::to_string(ks.value_node) + ";\n";
    }
    else
        for (size_t o = 0; o < ks.out_nodes.size(); ++o)
            s += "    o.o" + std::to_string(o) + " = n" + std::to_string(ks.out_nodes[o]) + ";\n";
    s += "}\n";

    const bool facc = ks.fsum && ks.value_node >= 0 && chgpu_type_is_float(e->types[ks.value_node]);
    const std::string U = std::to_string(JIT_UNROLL), VS = std::to_string(V);
    s += "extern \"C\" __global__ __launch_bounds__(256) void k_run(Args a)\n{\n";
    if (ks.fsum)
        s += facc ? "    f64 acc = 0; u64 cnt = 0;\n" : "    u64 acc = 0; u64 cnt = 0;\n";
    s += "    const u64 nvec = a.n / " + VS + ";\n    constexpr u64 CH = 256ull * " + U + ";\n    const u64 nch = nvec / CH;\n";
    // every wave owns a contiguous strip of U * 64 vectors of the chunk.  One-byte outputs (masks) whose V * U bytes per lane make 16
    // are transposed through a wave-private LDS strip so that each lane stores 16 CONTIGUOUS bytes once per chunk instead of U
    // narrow vectors (k_cmp_mask's scheme: the 4-byte stores cost the generated mask kernel 0.57 of peak against 0.72).
    const bool tr_ok = !ks.fsum && V * (u32)JIT_UNROLL == 16;
    std::vector<int> tr_slot(ks.out_nodes.size(), -1);
    int n_tr = 0;
    if (tr_ok)
        for (size_t o = 0; o < ks.out_nodes.size(); ++o)
            if (chgpu_type_size(e->types[ks.out_nodes[o]]) == 1)
                tr_slot[o] = n_tr++;
    if (n_tr)
        s += "    typedef u8 v16b_t __attribute__((ext_vector_type(16)));\n    __shared__ __attribute__((aligned(16))) u8 tr[" + std::to_string(n_tr) + "][4][1024];\n";
    s += "    for (u64 ch = blockIdx.x; ch < nch; ch += gridDim.x)\n    {\n        const u64 sb = ch * CH + (u64)(threadIdx.x >> 6) * (64 * " + U +
         ");\n        const u64 vb = sb + (threadIdx.x & 63);\n";
    for (size_t j = 0; j < e->input_types.size(); ++j)
        if (e->input_types[j] >= 0)
        {
            const std::string vt = vtype(e->input_types[j]), J = std::to_string(j);
            s += "        " + vt + " x" + J + "[" + U + "];\n";
            s += "#pragma unroll\n        for (int k = 0; k < " + U + "; ++k) x" + J + "[k] = __builtin_nontemporal_load((const " + vt + " *)a.in[" + J + "] + vb + (u64)k * 64);\n";
        }
    s += "#pragma unroll\n        for (int k = 0; k < " + U + "; ++k)\n        {\n";
    if (!ks.fsum)
        for (size_t o = 0; o < ks.out_nodes.size(); ++o)
            s += "            " + vtype(e->types[ks.out_nodes[o]]) + " y" + std::to_string(o) + ";\n";
    s += "#pragma unroll\n            for (int q = 0; q < " + VS + "; ++q)\n            {\n                Row r; Res o;\n";
    for (size_t j = 0; j < e->input_types.size(); ++j)
        if (e->input_types[j] >= 0)
            s += "                r.c" + std::to_string(j) + " = x" + std::to_string(j) + "[k][q];\n";
    s += "                eval(r, o);\n";
    if (ks.fsum)
    {
        if (ks.value_node >= 0)
            s += facc ? "                acc += o.keep ? (f64)o.val : 0.0;\n" : "                acc += o.keep ? (u64)o.val : 0ull;\n";
        s += "                cnt += o.keep ? 1u : 0u;\n";
    }
    else
        for (size_t o = 0; o < ks.out_nodes.size(); ++o)
            s += "                y" + std::to_string(o) + "[q] = o.o" + std::to_string(o) + ";\n";
    s += "            }\n";
    if (!ks.fsum)
        for (size_t o = 0; o < ks.out_nodes.size(); ++o)
        {
            if (tr_slot[o] >= 0)
                s += "            *(" + vtype(e->types[ks.out_nodes[o]]) + " *)&tr[" + std::to_string(tr_slot[o]) + "][threadIdx.x >> 6][(k * 64 + (threadIdx.x & 63)) * " + VS +
                     "] = y" + std::to_string(o) + ";\n";
            else
                s += "            ((" + vtype(e->types[ks.out_nodes[o]]) + " *)a.out[" + std::to_string(o) + "])[vb + (u64)k * 64] = y" + std::to_string(o) + ";\n";
        }
    s += "        }\n";
    if (n_tr)
    {
        s += "        __builtin_amdgcn_wave_barrier();\n"; // LDS operations of one wave complete in order
        for (size_t o = 0; o < ks.out_nodes.size(); ++o)
            if (tr_slot[o] >= 0)
                s += "        *((v16b_t *)((u8 *)a.out[" + std::to_string(o) + "] + sb * " + VS + ") + (threadIdx.x & 63)) = *(const v16b_t *)&tr[" +
                     std::to_string(tr_slot[o]) + "][threadIdx.x >> 6][(threadIdx.x & 63) * 16];\n";
        s += "        __builtin_amdgcn_wave_barrier();\n";
    }
    s += "    }\n";
    // ragged tail, one row per lane
    s += "    for (u64 i = nch * CH * " + VS + " + (u64)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (u64)gridDim.x * 256)\n    {\n        Row r; Res o;\n";
    for (size_t j = 0; j < e->input_types.size(); ++j)
        if (e->input_types[j] >= 0)
            s += "        r.c" + std::to_string(j) + " = ((const " + ctype(e->input_types[j]) + " *)a.in[" + std::to_string(j) + "])[i];\n";
    s += "        eval(r, o);\n";
    if (ks.fsum)
    {
        if (ks.value_node >= 0)
            s += facc ? "        acc += o.keep ? (f64)o.val : 0.0;\n" : "        acc += o.keep ? (u64)o.val : 0ull;\n";
        s += "        cnt += o.keep ? 1u : 0u;\n";
    }
    else
        for (size_t o = 0; o < ks.out_nodes.size(); ++o)
            s += "        ((" + std::string(ctype(e->types[ks.out_nodes[o]])) + " *)a.out[" + std::to_string(o) + "])[i] = o.o" + std::to_string(o) + ";\n";
    s += "    }\n";
    if (ks.fsum)
    {
        // wave64 shuffle reduce -> LDS -> one partial per workgroup; k_fin adds the partials in a fixed order
        s += R"SRC(
    __shared__ u64 sh_a[4], sh_c[4];
    u64 ab = ACC_BITS(acc);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
    {
        const u64 oa = ((u64)__shfl_down((u32)(ab >> 32), d, 64) << 32) | __shfl_down((u32)ab, d, 64);
        const u64 oc = ((u64)__shfl_down((u32)(cnt >> 32), d, 64) << 32) | __shfl_down((u32)cnt, d, 64);
        ab = ACC_ADD(ab, oa);
        cnt += oc;
    }
    if ((threadIdx.x & 63) == 0) { sh_a[threadIdx.x >> 6] = ab; sh_c[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        u64 ta = sh_a[0], tc = sh_c[0];
        for (int w = 1; w < 4; ++w) { ta = ACC_ADD(ta, sh_a[w]); tc += sh_c[w]; }
        a.part[2 * blockIdx.x] = ta;
        a.part[2 * blockIdx.x + 1] = tc;
    }
}
extern "C" __global__ __launch_bounds__(64) void k_fin(Args a)
{
    if (threadIdx.x != 0) return;
    u64 ta = ACC_ZERO, tc = 0;
    for (u32 p = 0; p < a.n_parts; ++p) { ta = ACC_ADD(ta, a.part[2 * p]); tc += a.part[2 * p + 1]; }
    a.part[2 * a.n_parts] = ta;
    a.part[2 * a.n_parts + 1] = tc;
}
)SRC";
        const std::string defs = facc ? "#define ACC_BITS(x) ((u64)__double_as_longlong(x))\n#define ACC_ADD(x, y) ((u64)__double_as_longlong(__longlong_as_double((long long)(x)) + __longlong_as_double((long long)(y))))\n#define ACC_ZERO 0ull\n"
                                      : "#define ACC_BITS(x) (x)\n#define ACC_ADD(x, y) ((x) + (y))\n#define ACC_ZERO 0ull\n";
        s = defs + s;
    }
    else
        s += "}\n";
    return s;
}

// WHERE + projection in one step (FilterTransform behind an ExpressionTransform): k_fcount evaluates the filter node and counts
// the surviving rows of every 1024-row chunk (one wave per chunk, 64 consecutive rows per step); after the scan of the counts
// k_femit evaluates filter and outputs again and writes the survivors compacted, in order (rank inside a step from the wave
// ballot).  The inputs are read twice; no mask, no unfiltered intermediate column is ever written.
constexpr u32 FE_CHUNK = 1024;

std::string gen_filter_source(const chgpu_expr * e, const KernelSpec & ks)
{
    std::string s = PRELUDE;
    s += "struct Args { const void * in[8]; void * out[8]; u64 n; u64 * part; u32 n_parts; u32 pad; };\n";
    s += "struct Row {";
    for (size_t j = 0; j < e->input_types.size(); ++j)
        if (e->input_types[j] >= 0)
            s += std::string(" ") + ctype(e->input_types[j]) + " c" + std::to_string(j) + ";";
    s += " };\nstruct Res { bool keep;";
    for (size_t o = 0; o < ks.out_nodes.size(); ++o)
        s += std::string(" ") + ctype(e->types[ks.out_nodes[o]]) + " o" + std::to_string(o) + ";";
    s += " };\nDEV void eval(const Row & r, Res & o)\n{\n" + e->body;
    s += "    o.keep = n" + std::to_string(ks.filter_node) + " != 0;\n";
    for (size_t o = 0; o < ks.out_nodes.size(); ++o)
        s += "    o.o" + std::to_string(o) + " = n" + std::to_string(ks.out_nodes[o]) + ";\n";
    s += "}\n";
    std::string load;
    for (size_t j = 0; j < e->input_types.size(); ++j)
        if (e->input_types[j] >= 0)
            load += "            r.c" + std::to_string(j) + " = ((const " + ctype(e->input_types[j]) + " *)a.in[" + std::to_string(j) + "])[in ? i : 0];\n";
    const std::string C = std::to_string(FE_CHUNK);
    // a.part: u32 counts[n_chunks] (k_fcount writes) ; a.out[7]: const u64 offsets[n_chunks] (k_femit reads)
    s += "extern \"C\" __global__ __launch_bounds__(256) void k_fcount(Args a)\n{\n"
         "    const u32 lane = threadIdx.x & 63;\n"
         "    const u64 n_chunks = (a.n + " + C + " - 1) / " + C + ";\n"
         "    for (u64 ch = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6; ch < n_chunks; ch += ((u64)gridDim.x * 256) >> 6)\n    {\n"
         "        u32 cnt = 0;\n"
         "#pragma unroll 4\n"
         "        for (u32 st = 0; st < " + C + " / 64; ++st)\n        {\n"
         "            const u64 i = ch * " + C + " + st * 64 + lane;\n            const bool in = i < a.n;\n            Row r; Res o;\n" + load +
         "            eval(r, o);\n            cnt += (u32)__popcll(__ballot(in && o.keep));\n        }\n"
         "        if (lane == 0) ((u32 *)a.part)[ch] = cnt;\n    }\n}\n";
    s += "extern \"C\" __global__ __launch_bounds__(256) void k_femit(Args a)\n{\n"
         "    const u32 lane = threadIdx.x & 63;\n"
         "    const u64 n_chunks = (a.n + " + C + " - 1) / " + C + ";\n"
         "    const u64 * offsets = (const u64 *)a.part;\n"
         "    for (u64 ch = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6; ch < n_chunks; ch += ((u64)gridDim.x * 256) >> 6)\n    {\n"
         "        u64 pos = offsets[ch];\n"
         "#pragma unroll 4\n"
         "        for (u32 st = 0; st < " + C + " / 64; ++st)\n        {\n"
         "            const u64 i = ch * " + C + " + st * 64 + lane;\n            const bool in = i < a.n;\n            Row r; Res o;\n" + load +
         "            eval(r, o);\n            const bool keep = in && o.keep;\n            const u64 b = __ballot(keep);\n"
         "            const u64 dst = pos + __builtin_amdgcn_mbcnt_hi((u32)(b >> 32), __builtin_amdgcn_mbcnt_lo((u32)b, 0u));\n"
         "            if (keep)\n            {\n";
    for (size_t o = 0; o < ks.out_nodes.size(); ++o)
        s += "                ((" + std::string(ctype(e->types[ks.out_nodes[o]])) + " *)a.out[" + std::to_string(o) + "])[dst] = o.o" + std::to_string(o) + ";\n";
    s += "            }\n            pos += (u64)__popcll(b);\n        }\n    }\n}\n";
    return s;
}

// min(value), max(value), count() over the rows that pass the filter (executeWithoutKeyImpl with AggregateFunctionMin / Max,
// src/AggregateFunctions/AggregateFunctionMinMax.h; SingleValueDataFixed::setIfSmaller / setIfGreater) for INTEGER values: the value is
// widened to 64 bits and mapped to an order-preserving unsigned key, lanes keep a running (lowest, highest), waves reduce by shuffles.
// Float values are not carried: the reference keeps a NaN that arrives first (setIfSmaller compares with <), an order-dependent result.
std::string gen_minmax_source(const chgpu_expr * e, int filter_node, int value_node)
{
    std::string s = PRELUDE;
    s += "struct Args { const void * in[8]; void * out[8]; u64 n; u64 * part; u32 n_parts; u32 pad; };\n";
    s += "struct Row {";
    for (size_t j = 0; j < e->input_types.size(); ++j)
        if (e->input_types[j] >= 0)
            s += std::string(" ") + ctype(e->input_types[j]) + " c" + std::to_string(j) + ";";
    s += " };\nstruct Res { bool keep; u64 key; };\nDEV void eval(const Row & r, Res & o)\n{\n" + e->body;
    s += filter_node >= 0 ? "    o.keep = n" + std::to_string(filter_node) + " != 0;\n" : std::string("    o.keep = true;\n");
    const bool sg = chgpu_type_is_signed(e->types[value_node]);
    s += std::string("    o.key = ") + (sg ? "(u64)(i64)n" + std::to_string(value_node) + " ^ 0x8000000000000000ull" : "(u64)n" + std::to_string(value_node)) + ";\n}\n";
    std::string load;
    for (size_t j = 0; j < e->input_types.size(); ++j)
        if (e->input_types[j] >= 0)
            load += "            r.c" + std::to_string(j) + " = ((const " + ctype(e->input_types[j]) + " *)a.in[" + std::to_string(j) + "])[in ? i : 0];\n";
    s += R"SRC(
DEV u64 shfl_u64(u64 v, int d) { return ((u64)__shfl_down((u32)(v >> 32), d, 64) << 32) | __shfl_down((u32)v, d, 64); }
extern "C" __global__ __launch_bounds__(256) void k_mm(Args a)
{
    u64 lo = ~0ull, hi = 0ull, cnt = 0;
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i0 = (u64)blockIdx.x * 256 + threadIdx.x; i0 < a.n; i0 += stride * 4)
    {
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
            const u64 i = i0 + (u64)q * stride;
            const bool in = i < a.n;
            Row r; Res o;
)SRC" + load + R"SRC(
            eval(r, o);
            if (in && o.keep) { lo = o.key < lo ? o.key : lo; hi = o.key > hi ? o.key : hi; ++cnt; }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
    {
        const u64 l2 = shfl_u64(lo, d), h2 = shfl_u64(hi, d), c2 = shfl_u64(cnt, d);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; cnt += c2;
    }
    __shared__ u64 sl[4], sh[4], sc[4];
    if ((threadIdx.x & 63) == 0) { sl[threadIdx.x >> 6] = lo; sh[threadIdx.x >> 6] = hi; sc[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        for (int w = 1; w < 4; ++w) { lo = sl[w] < lo ? sl[w] : lo; hi = sh[w] > hi ? sh[w] : hi; cnt += sc[w]; }
        a.part[3 * blockIdx.x] = lo; a.part[3 * blockIdx.x + 1] = hi; a.part[3 * blockIdx.x + 2] = cnt;
    }
}
extern "C" __global__ __launch_bounds__(64) void k_mm_fin(Args a)
{
    if (threadIdx.x != 0) return;
    u64 lo = ~0ull, hi = 0ull, cnt = 0;
    for (u32 p = 0; p < a.n_parts; ++p)
    {
        const u64 l2 = a.part[3 * p], h2 = a.part[3 * p + 1];
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; cnt += a.part[3 * p + 2];
    }
    a.part[3 * a.n_parts] = lo; a.part[3 * a.n_parts + 1] = hi; a.part[3 * a.n_parts + 2] = cnt;
}
)SRC";
    return s;
}

int jit_compile(const std::string & src, const std::vector<char> ** code_out)
{
    std::lock_guard<std::mutex> g(g_jit_mutex);
    auto it = g_code_cache.find(src);
    if (it == g_code_cache.end())
    {
        CHGPU_TRY(rtc_load());
        void * prog = nullptr;
        if (g_rtc.create(&prog, src.c_str(), "chgpu_expr.hip", 0, nullptr, nullptr) != 0)
            return chgpu_set_error(CHGPU_ERR_DEVICE, "hiprtcCreateProgram failed");
        const char * opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off"};
        const int rc = g_rtc.compile(prog, 3, opts);
        if (rc != 0)
        {
            size_t ls = 0;
            g_rtc.log_size(prog, &ls);
            std::string log(ls + 1, '\0');
            if (ls)
                g_rtc.log(prog, &log[0]);
            g_rtc.destroy(&prog);
            if (getenv("CHGPU_JIT_DUMP"))
                fprintf(stderr, "%s\n", src.c_str());
            return chgpu_set_error(CHGPU_ERR_LOGICAL, "hiprtc compile failed (%d): %.400s", rc, log.c_str());
        }
        size_t cs = 0;
        g_rtc.code_size(prog, &cs);
        std::vector<char> code(cs);
        g_rtc.code(prog, code.data());
        g_rtc.destroy(&prog);
        it = g_code_cache.emplace(src, std::move(code)).first;
    }
    *code_out = &it->second;
    return CHGPU_OK;
}

int jit_module(chgpu_ctx * ctx, const std::string & src, hipModule_t * mod)
{
    const std::vector<char> * code = nullptr;
    CHGPU_TRY(jit_compile(src, &code));
    std::lock_guard<std::mutex> g(g_jit_mutex);
    auto key = std::make_pair(ctx->device, src);
    auto it = g_module_cache.find(key);
    if (it == g_module_cache.end())
    {
        hipModule_t m = nullptr;
        CHGPU_HIP(hipModuleLoadData(&m, code->data()));
        it = g_module_cache.emplace(key, m).first;
    }
    *mod = it->second;
    return CHGPU_OK;
}

int check_spec(const chgpu_expr * e, uint32_t n_cols, const chgpu_col * const * cols, u64 * rows_out)
{
    CHGPU_REQUIRE(n_cols >= e->input_types.size(), CHGPU_ERR_BAD_ARGUMENTS, "expression reads column %zu, %u columns given", e->input_types.size() - 1, n_cols);
    u64 rows = 0;
    bool first = true;
    for (size_t j = 0; j < e->input_types.size(); ++j)
    {
        if (e->input_types[j] < 0)
            continue;
        CHGPU_REQUIRE(cols[j], CHGPU_ERR_BAD_ARGUMENTS, "column %zu is NULL", j);
        CHGPU_REQUIRE(cols[j]->type == e->input_types[j], CHGPU_ERR_BAD_ARGUMENTS, "column %zu has type %d, the expression was compiled for %d", j, cols[j]->type, e->input_types[j]);
        if (first)
            rows = cols[j]->rows, first = false;
        CHGPU_REQUIRE(cols[j]->rows == rows, CHGPU_ERR_SIZES_MISMATCH, "Sizes of columns doesn't match: %llu and %llu", (unsigned long long)cols[j]->rows, (unsigned long long)rows);
    }
    CHGPU_REQUIRE(!first, CHGPU_ERR_BAD_ARGUMENTS, "an expression without input columns has no row count");
    *rows_out = rows;
    return CHGPU_OK;
}

} // namespace

extern "C" int chgpu_expr_compile(uint32_t n_nodes, const chgpu_expr_node * nodes, chgpu_expr ** out)
{
    CHGPU_REQUIRE(nodes && out && n_nodes > 0, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(n_nodes <= 256, CHGPU_ERR_NOT_IMPLEMENTED, "expression of %u nodes", n_nodes);
    chgpu_expr * e = new chgpu_expr;
    e->nodes.assign(nodes, nodes + n_nodes);
    e->types.assign(n_nodes, -1);
    int rc = CHGPU_OK;
    for (uint32_t k = 0; k < n_nodes && rc == CHGPU_OK; ++k)
    {
        const chgpu_expr_node & nd = nodes[k];
        if (nd.kind == CHGPU_EX_INPUT)
        {
            if (nd.code < 0 || (u32)nd.code >= JIT_MAX_COLS || !chgpu_type_size(nd.type))
                rc = chgpu_set_error(CHGPU_ERR_BAD_ARGUMENTS, "node %u: input column %d of type %d", k, nd.code, nd.type);
            else
            {
                if (e->input_types.size() <= (size_t)nd.code)
                    e->input_types.resize(nd.code + 1, -1);
                if (e->input_types[nd.code] >= 0 && e->input_types[nd.code] != nd.type)
                    rc = chgpu_set_error(CHGPU_ERR_BAD_ARGUMENTS, "node %u: column %d declared with two types", k, nd.code);
                e->input_types[nd.code] = nd.type;
                e->types[k] = nd.type;
            }
        }
        else if (nd.kind == CHGPU_EX_CONST)
        {
            if (!chgpu_type_size(nd.type))
                rc = chgpu_set_error(CHGPU_ERR_BAD_ARGUMENTS, "node %u: constant of type %d", k, nd.type);
            e->types[k] = nd.type;
        }
        else if (nd.kind == CHGPU_EX_FUNC)
        {
            const int ar = fn_arity(nd.code);
            int at[3] = {-1, -1, -1};
            for (int j = 0; j < ar && rc == CHGPU_OK; ++j)
            {
                if (nd.args[j] < 0 || (u32)nd.args[j] >= k)
                    rc = chgpu_set_error(CHGPU_ERR_BAD_ARGUMENTS, "node %u: operand %d is not an earlier node", k, j);
                else
                    at[j] = e->types[nd.args[j]];
            }
            if (rc == CHGPU_OK && (nd.code == CHGPU_FN_INT_DIV || nd.code == CHGPU_FN_MODULO))
            {
                // intDiv / modulo throw ILLEGAL_DIVISION on a zero divisor and on min / -1 (throwIfDivisionLeadsToFPE,
                // DivisionUtils.h:15-26): a kernel cannot throw per row, so only constant divisors that can never throw are
                // compiled; everything else stays on the CPU
                const chgpu_expr_node & dn = nodes[nd.args[1]];
                bool ok = dn.kind == CHGPU_EX_CONST && at[0] >= 0 && at[1] >= 0 && chgpu_type_is_int(at[0]) && chgpu_type_is_int(at[1]);
                if (ok)
                {
                    const size_t sb = chgpu_type_size(at[1]);
                    const u64 mask = sb == 8 ? ~0ull : ((1ull << (8 * sb)) - 1);
                    const u64 bits = dn.bits & mask;
                    if (bits == 0)
                        ok = false; // division by zero
                    // all-ones divisor: -1 once it is (or is cast to) a signed type of its own width -- min / -1 would throw
                    if (bits == mask && (chgpu_type_is_signed(at[1]) || (nd.code == CHGPU_FN_INT_DIV && chgpu_type_is_signed(at[0]) && chgpu_type_size(at[0]) <= sb)))
                        ok = false;
                    // the most negative value of a signed divisor type: the reference's constant-divisor path (ModuloByConstantImpl::vectorConstant,
                    // src/Functions/modulo.cpp:56-80) throws ILLEGAL_DIVISION "Division by the most negative number" where ModuloImpl::apply would
                    // compute a % b -- e.g. Int64 % toInt64(-9223372036854775808), Int32 % toInt8(-128).  Refused for every operand pair (a superset).
                    if (nd.code == CHGPU_FN_MODULO && chgpu_type_is_signed(at[1]) && bits == (1ull << (8 * sb - 1)))
                        ok = false; // (intDiv by that constant does not throw: DivideIntegralByConstantImpl, src/Functions/intDiv.cpp:55-78)
                }
                if (!ok)
                    rc = chgpu_set_error(CHGPU_ERR_NOT_IMPLEMENTED, "node %u: intDiv / modulo need a constant integer divisor that cannot raise ILLEGAL_DIVISION", k);
            }
            if (rc == CHGPU_OK)
            {
                e->types[k] = infer_type(nd.code, at[0], at[1], at[2]);
                if (e->types[k] < 0)
                    rc = chgpu_set_error(CHGPU_ERR_NOT_IMPLEMENTED, "node %u: function %d over types (%d, %d, %d)", k, nd.code, at[0], at[1], at[2]);
            }
        }
        else
            rc = chgpu_set_error(CHGPU_ERR_BAD_ARGUMENTS, "node %u: kind %d", k, nd.kind);
    }
    if (rc == CHGPU_OK)
        rc = build_body(e);
    if (rc != CHGPU_OK)
    {
        delete e;
        return rc;
    }
    *out = e;
    return CHGPU_OK;
}

extern "C" int chgpu_expr_node_type(const chgpu_expr * e, uint32_t node, int * type_out)
{
    CHGPU_REQUIRE(e && type_out && node < e->types.size(), CHGPU_ERR_BAD_ARGUMENTS, "bad node");
    *type_out = e->types[node];
    return CHGPU_OK;
}

extern "C" int chgpu_expr_free(chgpu_expr * e)
{
    delete e;
    return CHGPU_OK;
}

static int make_spec(const chgpu_expr * e, bool fsum, uint32_t n_outputs, const uint32_t * out_nodes, int filter_node, int value_node, bool aligned, KernelSpec * ks)
{
    ks->fsum = fsum;
    std::vector<int> touched;
    for (int t : e->input_types)
        if (t >= 0)
            touched.push_back(t);
    if (fsum)
    {
        CHGPU_REQUIRE(filter_node < (int)e->types.size() && value_node < (int)e->types.size(), CHGPU_ERR_BAD_ARGUMENTS, "bad node");
        CHGPU_REQUIRE(filter_node < 0 || chgpu_type_is_int(e->types[filter_node]), CHGPU_ERR_BAD_ARGUMENTS,
                      "Illegal type for filter: the WHERE node must be an integer (FilterDescription.cpp:86-92)");
        ks->filter_node = filter_node;
        ks->value_node = value_node;
    }
    else
    {
        CHGPU_REQUIRE(n_outputs > 0 && n_outputs <= JIT_MAX_COLS && out_nodes, CHGPU_ERR_BAD_ARGUMENTS, "1..8 outputs");
        for (uint32_t o = 0; o < n_outputs; ++o)
        {
            CHGPU_REQUIRE(out_nodes[o] < e->types.size(), CHGPU_ERR_BAD_ARGUMENTS, "bad output node");
            ks->out_nodes.push_back(out_nodes[o]);
            touched.push_back(e->types[out_nodes[o]]);
        }
    }
    ks->vec = aligned ? vec_rows(touched) : 1;
    return CHGPU_OK;
}

/* run hiprtc only (no device needed): the "does it compile for gfx950" check of a DAG */
extern "C" int chgpu_expr_precompile(const chgpu_expr * e, uint32_t n_outputs, const uint32_t * out_nodes, int filter_node, int value_node,
                                     uint64_t * code_bytes_out)
{
    CHGPU_REQUIRE(e, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    KernelSpec ks;
    CHGPU_TRY(make_spec(e, n_outputs == 0, n_outputs, out_nodes, filter_node, value_node, true, &ks));
    const std::vector<char> * code = nullptr;
    if (n_outputs > 0 && filter_node >= 0) // the WHERE + projection pair (k_fcount, k_femit)
    {
        CHGPU_REQUIRE((size_t)filter_node < e->types.size() && n_outputs <= 7, CHGPU_ERR_BAD_ARGUMENTS, "bad filter node / more than 7 outputs");
        ks.filter_node = filter_node;
        CHGPU_TRY(jit_compile(gen_filter_source(e, ks), &code));
    }
    else
        CHGPU_TRY(jit_compile(gen_source(e, ks), &code));
    if (code_bytes_out)
        *code_bytes_out = code->size();
    return CHGPU_OK;
}

static bool cols_aligned(const chgpu_expr * e, const chgpu_col * const * cols)
{
    for (size_t j = 0; j < e->input_types.size(); ++j)
        if (e->input_types[j] >= 0 && ((uintptr_t)cols[j]->data & 63) != 0)
            return false;
    return true;
}

extern "C" int chgpu_expr_execute(chgpu_ctx * ctx, const chgpu_expr * e, uint32_t n_cols, const chgpu_col * const * cols, uint32_t n_outputs,
                                  const uint32_t * out_nodes, chgpu_col ** outs)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && e && cols && outs, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    u64 rows = 0;
    CHGPU_TRY(check_spec(e, n_cols, cols, &rows));
    KernelSpec ks;
    CHGPU_TRY(make_spec(e, false, n_outputs, out_nodes, -1, -1, cols_aligned(e, cols), &ks));
    hipModule_t mod = nullptr;
    CHGPU_TRY(jit_module(ctx, gen_source(e, ks), &mod));
    hipFunction_t fn = nullptr;
    CHGPU_HIP(hipModuleGetFunction(&fn, mod, "k_run"));
    JitArgs a;
    memset(&a, 0, sizeof(a));
    for (size_t j = 0; j < e->input_types.size(); ++j)
        a.in[j] = e->input_types[j] >= 0 ? cols[j]->data : nullptr;
    std::vector<chgpu_col *> res(n_outputs, nullptr);
    for (uint32_t o = 0; o < n_outputs; ++o)
    {
        const int rc = chgpu_col_new(ctx, e->types[out_nodes[o]], rows, &res[o]);
        if (rc != CHGPU_OK)
        {
            for (chgpu_col * c : res)
                if (c)
                    chgpu_col_free(c);
            return rc;
        }
        a.out[o] = res[o]->data;
    }
    a.n = rows;
    if (rows)
    {
        const int wg_map = jit_env(ctx, "tune_jit_wg_map", 4);
        const u32 grid = chgpu_grid_for(ctx, (rows + ks.vec * JIT_UNROLL - 1) / (ks.vec * JIT_UNROLL), 256, wg_map);
        void * params[] = {&a};
        const hipError_t le = hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, ctx->stream, params, nullptr);
        if (le != hipSuccess)
        {
            for (chgpu_col * c : res)
                chgpu_col_free(c);
            return chgpu_set_error(CHGPU_ERR_DEVICE, "expression kernel launch: %s", hipGetErrorString(le));
        }
        ctx->counters[6] += 1;
    }
    for (uint32_t o = 0; o < n_outputs; ++o)
        outs[o] = res[o];
    return CHGPU_OK;
}

extern "C" int chgpu_expr_filter_sum_node(chgpu_ctx * ctx, const chgpu_expr * e, uint32_t n_cols, const chgpu_col * const * cols, int filter_node,
                                          int value_node, int * result_type_out, void * sum_out, uint64_t * count_out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && e && cols, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    u64 rows = 0;
    CHGPU_TRY(check_spec(e, n_cols, cols, &rows));
    KernelSpec ks;
    CHGPU_TRY(make_spec(e, true, 0, nullptr, filter_node, value_node, cols_aligned(e, cols), &ks));
    if (result_type_out)
        *result_type_out = value_node >= 0 ? chgpu_sum_result_type(e->types[value_node]) : CHGPU_U64;
    u64 res[2] = {0, 0};
    if (rows)
    {
        hipModule_t mod = nullptr;
        CHGPU_TRY(jit_module(ctx, gen_source(e, ks), &mod));
        hipFunction_t fn = nullptr, fin = nullptr;
        CHGPU_HIP(hipModuleGetFunction(&fn, mod, "k_run"));
        CHGPU_HIP(hipModuleGetFunction(&fin, mod, "k_fin"));
        const int wg_sum = jit_env(ctx, "tune_jit_wg_sum", 2);
        const u32 grid = chgpu_grid_for(ctx, (rows + ks.vec * JIT_UNROLL - 1) / (ks.vec * JIT_UNROLL), 256, wg_sum);
        void * scratch = nullptr;
        CHGPU_TRY(chgpu_scratch(ctx, ((size_t)grid + 1) * 2 * sizeof(u64), &scratch));
        JitArgs a;
        memset(&a, 0, sizeof(a));
        for (size_t j = 0; j < e->input_types.size(); ++j)
            a.in[j] = e->input_types[j] >= 0 ? cols[j]->data : nullptr;
        a.n = rows;
        a.part = (u64 *)scratch;
        a.n_parts = grid;
        void * params[] = {&a};
        CHGPU_HIP(hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, ctx->stream, params, nullptr));
        CHGPU_HIP(hipModuleLaunchKernel(fin, 1, 1, 1, 64, 1, 1, 0, ctx->stream, params, nullptr));
        ctx->counters[6] += 2;
        CHGPU_TRY(chgpu_read_back(ctx, a.part + 2 * (size_t)grid, res, sizeof(res)));
    }
    if (sum_out)
        memcpy(sum_out, &res[0], 8);
    if (count_out)
        *count_out = res[1];
    return CHGPU_OK;
}

extern "C" int chgpu_expr_filter_execute(chgpu_ctx * ctx, const chgpu_expr * e, uint32_t n_cols, const chgpu_col * const * cols, uint32_t filter_node,
                                         uint32_t n_outputs, const uint32_t * out_nodes, chgpu_col ** outs, uint64_t * rows_out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && e && cols && outs && rows_out && out_nodes, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(n_outputs >= 1 && n_outputs <= 7, CHGPU_ERR_BAD_ARGUMENTS, "1..7 outputs");
    CHGPU_REQUIRE(filter_node < e->types.size() && chgpu_type_is_int(e->types[filter_node]), CHGPU_ERR_BAD_ARGUMENTS,
                  "Illegal type for filter: the WHERE node must be an integer (FilterDescription.cpp:86-92)");
    u64 rows = 0;
    CHGPU_TRY(check_spec(e, n_cols, cols, &rows));
    KernelSpec ks;
    ks.filter_node = (int)filter_node;
    for (uint32_t o = 0; o < n_outputs; ++o)
    {
        CHGPU_REQUIRE(out_nodes[o] < e->types.size(), CHGPU_ERR_BAD_ARGUMENTS, "bad output node");
        ks.out_nodes.push_back(out_nodes[o]);
    }
    std::vector<chgpu_col *> res(n_outputs, nullptr);
    auto fail = [&](int rc) {
        for (chgpu_col * c : res)
            if (c)
                chgpu_col_free(c);
        return rc;
    };
    u64 total = 0;
    if (rows)
    {
        hipModule_t mod = nullptr;
        CHGPU_TRY(jit_module(ctx, gen_filter_source(e, ks), &mod));
        hipFunction_t fcount = nullptr, femit = nullptr;
        CHGPU_HIP(hipModuleGetFunction(&fcount, mod, "k_fcount"));
        CHGPU_HIP(hipModuleGetFunction(&femit, mod, "k_femit"));
        const u64 n_chunks = (rows + FE_CHUNK - 1) / FE_CHUNK;
        auto al = [](size_t b) { return (b + 255) / 256 * 256; };
        const size_t b_cnt = al(n_chunks * 4), b_off = al(n_chunks * 8), b_tmp = chgpu_scan_tmp_bytes(n_chunks);
        void * scratch = nullptr;
        CHGPU_TRY(chgpu_scratch(ctx, b_cnt + b_off + 256 + b_tmp, &scratch));
        u32 * counts = (u32 *)scratch;
        u64 * offsets = (u64 *)((char *)scratch + b_cnt);
        u64 * total_dev = (u64 *)((char *)scratch + b_cnt + b_off);
        void * tmp = (char *)scratch + b_cnt + b_off + 256;
        JitArgs a;
        memset(&a, 0, sizeof(a));
        for (size_t j = 0; j < e->input_types.size(); ++j)
            a.in[j] = e->input_types[j] >= 0 ? cols[j]->data : nullptr;
        a.n = rows;
        a.part = (u64 *)counts;
        const u32 grid = chgpu_grid_for(ctx, n_chunks * 64, 256, 8);
        void * params[] = {&a};
        CHGPU_HIP(hipModuleLaunchKernel(fcount, grid, 1, 1, 256, 1, 1, 0, ctx->stream, params, nullptr));
        CHGPU_TRY(chgpu_scan_exclusive_u32_u64(ctx, counts, offsets, n_chunks, total_dev, tmp, b_tmp));
        CHGPU_TRY(chgpu_read_back(ctx, total_dev, &total, sizeof(total)));
        for (uint32_t o = 0; o < n_outputs; ++o)
        {
            const int rc = chgpu_col_new(ctx, e->types[out_nodes[o]], total, &res[o]);
            if (rc != CHGPU_OK)
                return fail(rc);
            a.out[o] = res[o]->data;
        }
        if (total)
        {
            a.part = offsets;
            if (hipModuleLaunchKernel(femit, grid, 1, 1, 256, 1, 1, 0, ctx->stream, params, nullptr) != hipSuccess)
                return fail(chgpu_set_error(CHGPU_ERR_DEVICE, "k_femit launch failed"));
        }
        ctx->counters[6] += 2;
        ctx->counters[0] += total; // FilterTransformPassedRows
    }
    else
        for (uint32_t o = 0; o < n_outputs; ++o)
        {
            const int rc = chgpu_col_new(ctx, e->types[out_nodes[o]], 0, &res[o]);
            if (rc != CHGPU_OK)
                return fail(rc);
        }
    for (uint32_t o = 0; o < n_outputs; ++o)
        outs[o] = res[o];
    *rows_out = total;
    return CHGPU_OK;
}

extern "C" int chgpu_expr_filter_minmax_node(chgpu_ctx * ctx, const chgpu_expr * e, uint32_t n_cols, const chgpu_col * const * cols, int filter_node,
                                             uint32_t value_node, int * value_type_out, void * min_out, void * max_out, uint64_t * count_out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && e && cols, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(value_node < e->types.size() && filter_node < (int)e->types.size(), CHGPU_ERR_BAD_ARGUMENTS, "bad node");
    CHGPU_REQUIRE(filter_node < 0 || chgpu_type_is_int(e->types[filter_node]), CHGPU_ERR_BAD_ARGUMENTS,
                  "Illegal type for filter: the WHERE node must be an integer (FilterDescription.cpp:86-92)");
    const int vt = e->types[value_node];
    CHGPU_REQUIRE(chgpu_type_is_int(vt), CHGPU_ERR_NOT_IMPLEMENTED,
                  "min / max of a Float column keep a NaN that arrives first (SingleValueDataFixed::setIfSmaller): order-dependent, CPU path");
    u64 rows = 0;
    CHGPU_TRY(check_spec(e, n_cols, cols, &rows));
    if (value_type_out)
        *value_type_out = vt;
    u64 res[3] = {~0ull, 0, 0};
    if (rows)
    {
        hipModule_t mod = nullptr;
        CHGPU_TRY(jit_module(ctx, gen_minmax_source(e, filter_node, (int)value_node), &mod));
        hipFunction_t fn = nullptr, fin = nullptr;
        CHGPU_HIP(hipModuleGetFunction(&fn, mod, "k_mm"));
        CHGPU_HIP(hipModuleGetFunction(&fin, mod, "k_mm_fin"));
        const u32 grid = chgpu_grid_for(ctx, (rows + 3) / 4, 256, 8);
        void * scratch = nullptr;
        CHGPU_TRY(chgpu_scratch(ctx, ((size_t)grid + 1) * 3 * sizeof(u64), &scratch));
        JitArgs a;
        memset(&a, 0, sizeof(a));
        for (size_t j = 0; j < e->input_types.size(); ++j)
            a.in[j] = e->input_types[j] >= 0 ? cols[j]->data : nullptr;
        a.n = rows;
        a.part = (u64 *)scratch;
        a.n_parts = grid;
        void * params[] = {&a};
        CHGPU_HIP(hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, ctx->stream, params, nullptr));
        CHGPU_HIP(hipModuleLaunchKernel(fin, 1, 1, 1, 64, 1, 1, 0, ctx->stream, params, nullptr));
        ctx->counters[6] += 2;
        CHGPU_TRY(chgpu_read_back(ctx, a.part + 3 * (size_t)grid, res, sizeof(res)));
    }
    // no row passed: the aggregate of an empty set is the type's default (AggregateFunctionMin on non-Nullable arguments: 0)
    const u64 sign = chgpu_type_is_signed(vt) ? 0x8000000000000000ull : 0;
    const u64 vmin = res[2] ? (res[0] ^ sign) : 0, vmax = res[2] ? (res[1] ^ sign) : 0;
    const size_t es = chgpu_type_size(vt);
    if (min_out)
        memcpy(min_out, &vmin, es); // little endian: the low bytes are the value in its own width
    if (max_out)
        memcpy(max_out, &vmax, es);
    if (count_out)
        *count_out = res[2];
    return CHGPU_OK;
}
