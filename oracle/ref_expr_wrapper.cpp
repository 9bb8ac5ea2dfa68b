/*
 * ref_expr_wrapper.cpp — thin extern "C" exports over the REFERENCE's own src/Core/AccurateComparison.h (accurate::lessOp ...) and
 * src/DataTypes/NumberTraits.h (result types of the arithmetic functions), compiled in place from /root/reference (never copied)
 * into oracle/_ref/libchref_expr.so by oracle/Makefile.  TEST INFRASTRUCTURE ONLY: pins oracle/expr_dag.py's comparison semantics and
 * result-type rules -- and through it the product's -- against the real thing.  Both headers compile without the reference's absent
 * submodules; src/Functions/DivisionUtils.h (intDiv / modulo) does not (it needs fmt through Common/Exception.h).
 * One limit: the reference's Int8 is `signed _BitInt(8)` and its is_signed rests on std::is_signed_v, which this image's libstdc++ answers
 * false for _BitInt (the reference builds against libc++).  So ref_result_type must not be asked about Int8 operands (the tests skip
 * them); ref_compare is fed plain int8_t, whose values and order are the same, and covers Int8 fully.
 */
#include <Core/AccurateComparison.h>
#include <DataTypes/NumberTraits.h>

#include <cstddef>
#include <cstdint>
#include <type_traits>

using namespace DB;

namespace
{
/* element type tags of include/chgpu.h */
template <typename F>
bool with_type(int tag, F && f)
{
    switch (tag)
    {
        case 0: f(Int64{}); return true;
        case 1: f(UInt32{}); return true;
        case 2: f(UInt64{}); return true;
        case 3: f(Float64{}); return true;
        case 4: f(UInt8{}); return true;
        case 5: f(Int32{}); return true;
        case 6: f(UInt16{}); return true;
        case 7: f(Int16{}); return true;
        case 8: f(int8_t{}); return true; /* the reference's Int8 is _BitInt(8); plain int8_t has the same values and order */
        case 9: f(Float32{}); return true;
        default: return false;
    }
}

template <typename T>
constexpr int tag_of()
{
    if constexpr (std::is_same_v<T, Int64>) return 0;
    else if constexpr (std::is_same_v<T, UInt32>) return 1;
    else if constexpr (std::is_same_v<T, UInt64>) return 2;
    else if constexpr (std::is_same_v<T, Float64>) return 3;
    else if constexpr (std::is_same_v<T, UInt8>) return 4;
    else if constexpr (std::is_same_v<T, Int32>) return 5;
    else if constexpr (std::is_same_v<T, UInt16>) return 6;
    else if constexpr (std::is_same_v<T, Int16>) return 7;
    else if constexpr (std::is_same_v<T, Int8>) return 8;
    else if constexpr (std::is_same_v<T, Float32>) return 9;
    else return -1; /* NumberTraits::Error, 128/256-bit integers, BFloat16: not carried by the hot path */
}
}

extern "C" {

/* fn: 10 plus, 11 minus, 12 multiply, 13 divide, 14 negate (b ignored), 15 intDiv, 16 modulo, 40 bitAnd/bitOr/bitXor, 30 if(cond, a, b) */
int ref_result_type(int fn, int ta, int tb)
{
    int res = -2;
    with_type(ta, [&](auto a) {
        using A = std::conditional_t<std::is_same_v<decltype(a), int8_t>, Int8, decltype(a)>; /* the type rules take the reference's Int8 */
        if (fn == 14)
        {
            res = tag_of<typename NumberTraits::ResultOfNegate<A>::Type>();
            return;
        }
        with_type(tb, [&](auto b) {
            using B = std::conditional_t<std::is_same_v<decltype(b), int8_t>, Int8, decltype(b)>;
            switch (fn)
            {
                case 10: case 12: res = tag_of<typename NumberTraits::ResultOfAdditionMultiplication<A, B>::Type>(); break;
                case 11: res = tag_of<typename NumberTraits::ResultOfSubtraction<A, B>::Type>(); break;
                case 13: res = tag_of<typename NumberTraits::ResultOfFloatingPointDivision<A, B>::Type>(); break;
                case 15: res = tag_of<typename NumberTraits::ResultOfIntegerDivision<A, B>::Type>(); break;
                case 16: res = tag_of<typename NumberTraits::ResultOfModulo<A, B>::Type>(); break;
                case 40: res = tag_of<typename NumberTraits::ResultOfBit<A, B>::Type>(); break;
                case 30: res = tag_of<typename NumberTraits::ResultOfIf<A, B>::Type>(); break;
                default: break;
            }
        });
    });
    return res;
}

/* op: 0 equals, 1 notEquals, 2 less, 3 greater, 4 lessOrEquals, 5 greaterOrEquals (accurate::*Op, AccurateComparison.h:20-204) */
int ref_compare(int op, int ta, int tb, const void * pa, const void * pb, size_t n, uint8_t * out)
{
    bool ok = false;
    with_type(ta, [&](auto a0) {
        using A = decltype(a0);
        ok = with_type(tb, [&](auto b0) {
            using B = decltype(b0);
            const A * a = static_cast<const A *>(pa);
            const B * b = static_cast<const B *>(pb);
            for (size_t i = 0; i < n; ++i)
            {
                bool r = false;
                switch (op)
                {
                    case 0: r = accurate::equalsOp(a[i], b[i]); break;
                    case 1: r = accurate::notEqualsOp(a[i], b[i]); break;
                    case 2: r = accurate::lessOp(a[i], b[i]); break;
                    case 3: r = accurate::greaterOp(a[i], b[i]); break;
                    case 4: r = accurate::lessOrEqualsOp(a[i], b[i]); break;
                    default: r = accurate::greaterOrEqualsOp(a[i], b[i]); break;
                }
                out[i] = r ? 1 : 0;
            }
        });
    });
    return ok ? 0 : -1;
}
}
