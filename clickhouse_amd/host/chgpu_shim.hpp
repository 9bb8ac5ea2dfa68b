// chgpu_shim.hpp — C++ host side above the C ABI (include/chgpu.h), shaped like the reference's plugin surfaces for
// the hot path so the GPU path is a drop-in under the existing query pipeline:
//
//   reference interface (file:line)                                            mirror here
//   -------------------------------------------------------------------------  ---------------------------------------
//   DB::Exception + ErrorCodes                                                 chgpu::Exception
//   IColumn / ColumnVector<T>   src/Columns/IColumn.h:80, ColumnVector.h:28    chgpu::ColumnVector (HBM-resident)
//   IColumn::Filter             src/Columns/FilterDescription.h:14             chgpu::ColumnVector of UInt8
//   Chunk                       src/Processors/Chunk.h:55-128                  chgpu::Chunk
//   IProcessor / ISimpleTransform  src/Processors/IProcessor.h:119-376,        chgpu::IProcessor, chgpu::ISimpleTransform
//                               ISimpleTransform.h:42
//   FilterTransform             src/Processors/Transforms/FilterTransform.cpp:136-256   chgpu::GpuFilterTransform
//   IFunction (less, ...)       src/Functions/IFunction.h:426-434              chgpu::FunctionComparisonConst
//   ActionsDAG / ExpressionActions  src/Interpreters/ActionsDAG.h, ExpressionActions.h:75-134   chgpu::ActionsDAG, chgpu::ExpressionActions (run-time compiled)
//   ExpressionTransform         src/Processors/Transforms/ExpressionTransform.cpp:22-30      chgpu::GpuExpressionTransform, GpuExpressionFilterTransform
//   ColumnLowCardinality + low_cardinality_key* methods   src/Columns/ColumnLowCardinality.h:27-69, ColumnsHashing.h:82-260   chgpu::ColumnLowCardinality, LowCardinalityDictionary
//   ColumnString as a key       src/Columns/ColumnString.h:40-49, ColumnUnique.h:520-620     chgpu::ColumnString::dictionaryEncode
//   sortBlock / SortDescription src/Interpreters/sortBlock.cpp:240-330, Core/SortDescription.h   chgpu::sortBlock
//   CompressedReadBuffer        src/Compression/CompressedReadBufferBase.cpp:175-222         chgpu::readCompressedColumn
//   IAggregateFunction          src/AggregateFunctions/IAggregateFunction.h:55-399      chgpu::AggregateDescription (closed POD set)
//   Aggregator                  src/Interpreters/Aggregator.h:179-265          chgpu::GpuAggregator
//   AggregatingTransform        src/Processors/Transforms/AggregatingTransform.cpp:640-840   chgpu::GpuAggregatingTransform
//   IJoin / HashJoin            src/Interpreters/IJoin.h:80-142                chgpu::GpuHashJoin
//   JoiningTransform            src/Processors/Transforms/JoiningTransform.cpp:176-260       chgpu::GpuJoiningTransform
//
// Host Blocks (65 409 rows) are far too small per kernel launch, so host-side sources batch many Blocks into one HBM
// stripe (StripeBuilder) before the device operators run; device-resident Chunks flow between the GPU transforms.
// Header-only, C++17, no dependency besides include/chgpu.h.  Errors from the C ABI become chgpu::Exception with the
// reference's error-code meaning; CHGPU_ERR_NOT_IMPLEMENTED is the signal to fall back to the CPU operator.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <array>
#include <atomic>
#include <exception>
#include <utility>
#include <vector>

#include "../../include/chgpu.h"

namespace chgpu
{

class Exception : public std::runtime_error
{
public:
    Exception(int code_, const std::string & msg) : std::runtime_error(msg), code_v(code_) {}
    int code() const { return code_v; }
    bool isNotImplemented() const { return code_v == CHGPU_ERR_NOT_IMPLEMENTED; }

private:
    int code_v;
};

inline void check(int rc)
{
    if (rc != CHGPU_OK)
        throw Exception(rc, chgpu_last_error());
}

template <typename T> struct TypeTag;
template <> struct TypeTag<int64_t> { static constexpr int value = CHGPU_I64; };
template <> struct TypeTag<uint64_t> { static constexpr int value = CHGPU_U64; };
template <> struct TypeTag<uint32_t> { static constexpr int value = CHGPU_U32; };
template <> struct TypeTag<int32_t> { static constexpr int value = CHGPU_I32; };
template <> struct TypeTag<uint8_t> { static constexpr int value = CHGPU_U8; };
template <> struct TypeTag<uint16_t> { static constexpr int value = CHGPU_U16; }; // also Date
template <> struct TypeTag<int16_t> { static constexpr int value = CHGPU_I16; };
template <> struct TypeTag<int8_t> { static constexpr int value = CHGPU_I8; };
template <> struct TypeTag<double> { static constexpr int value = CHGPU_F64; };
template <> struct TypeTag<float> { static constexpr int value = CHGPU_F32; };

/// One device + one HIP stream: create one per pipeline thread (IProcessor::work() of different processors runs
/// concurrently, src/Processors/IProcessor.h:176-193).
class Context
{
public:
    explicit Context(int device = 0, void * hip_stream = nullptr) { check(chgpu_ctx_create(device, hip_stream, &h)); }
    ~Context() { chgpu_ctx_destroy(h); }
    Context(const Context &) = delete;
    Context & operator=(const Context &) = delete;
    chgpu_ctx * get() const { return h; }
    void synchronize() const { check(chgpu_ctx_synchronize(h)); }

private:
    chgpu_ctx * h = nullptr;
};
using ContextPtr = std::shared_ptr<Context>;

class ColumnVector;
using ColumnPtr = std::shared_ptr<const ColumnVector>;
using MutableColumnPtr = std::shared_ptr<ColumnVector>;
using Columns = std::vector<ColumnPtr>;

/// ColumnVector<T> pinned into HBM.  Immutable once shared, like the reference's COW columns.
class ColumnVector
{
public:
    ColumnVector(ContextPtr ctx_, chgpu_col * h_, std::shared_ptr<const void> keepalive_ = nullptr)
        : ctx(std::move(ctx_)), h(h_), keepalive(std::move(keepalive_)) {}
    ~ColumnVector() { chgpu_col_free(h); }
    ColumnVector(const ColumnVector &) = delete;

    template <typename T>
    static MutableColumnPtr fromHost(const ContextPtr & ctx, const T * data, size_t rows)
    {
        chgpu_col * c = nullptr;
        check(chgpu_col_upload(ctx->get(), TypeTag<T>::value, data, rows, &c));
        return std::make_shared<ColumnVector>(ctx, c);
    }

    /// ColumnConst (src/Columns/ColumnConst.h): one stored value standing for `rows` equal rows; the data column keeps its single row in HBM
    static MutableColumnPtr createConst(const ColumnPtr & one_row, size_t rows)
    {
        if (one_row->size() != 1)
            throw Exception(CHGPU_ERR_BAD_ARGUMENTS, "ColumnConst: the data column must have one row");
        auto c = std::make_shared<ColumnVector>(one_row->ctx, nullptr, one_row);
        c->const_data = one_row;
        c->const_rows = rows;
        return c;
    }
    bool isConst() const { return const_data != nullptr; }
    const ColumnPtr & getDataColumnPtr() const { return const_data; }
    /// ColumnConst::cut / cloneResized: a constant of another length shares the stored value (FilterTransform.cpp:248-249)
    ColumnPtr cutConst(size_t length) const { return createConst(const_data, length); }

    size_t size() const { return const_data ? const_rows : chgpu_col_rows(h); }
    int getDataType() const { return chgpu_col_type(const_data ? const_data->h : h); }
    chgpu_col * handle() const { return h; }
    const ContextPtr & context() const { return ctx; }

    template <typename T>
    std::vector<T> getData() const
    {
        if (TypeTag<T>::value != getDataType())
            throw Exception(CHGPU_ERR_BAD_ARGUMENTS, "ColumnVector::getData: type mismatch");
        std::vector<T> out(size());
        check(chgpu_col_download(ctx->get(), h, out.data(), out.size()));
        return out;
    }

    /// IColumn::filter (IColumn.h:313-314)
    ColumnPtr filter(const ColumnVector & filt, ssize_t result_size_hint) const
    {
        chgpu_col * out = nullptr;
        uint64_t n = 0;
        check(chgpu_filter(ctx->get(), h, filt.h, result_size_hint, &out, &n));
        return std::make_shared<ColumnVector>(ctx, out);
    }
    /// IColumn::cut (IColumn.h:118-121); the view keeps its parent alive
    ColumnPtr cut(size_t start, size_t length, const ColumnPtr & self) const
    {
        chgpu_col * out = nullptr;
        check(chgpu_col_slice(ctx->get(), h, start, length, &out));
        return std::make_shared<ColumnVector>(ctx, out, self);
    }
    /// IColumn::index (IColumn.h:331)
    ColumnPtr index(const ColumnVector & indexes, size_t limit, bool default_for_missing = false) const
    {
        chgpu_col * out = nullptr;
        check(chgpu_index(ctx->get(), h, indexes.h, limit, default_for_missing ? 1 : 0, &out));
        return std::make_shared<ColumnVector>(ctx, out);
    }
    /// IColumn::replicate (IColumn.h:440)
    ColumnPtr replicate(const ColumnVector & offsets) const
    {
        chgpu_col * out = nullptr;
        check(chgpu_replicate(ctx->get(), h, offsets.h, &out));
        return std::make_shared<ColumnVector>(ctx, out);
    }
    /// IColumn::scatter (IColumn.h:448)
    std::vector<ColumnPtr> scatter(uint32_t num_columns, const ColumnVector & selector) const
    {
        std::vector<chgpu_col *> outs(num_columns, nullptr);
        check(chgpu_scatter(ctx->get(), h, selector.h, num_columns, outs.data()));
        std::vector<ColumnPtr> res;
        for (auto * c : outs)
            res.push_back(std::make_shared<ColumnVector>(ctx, c));
        return res;
    }

private:
    ContextPtr ctx;
    chgpu_col * h;
    std::shared_ptr<const void> keepalive;
    ColumnPtr const_data;  // set: this is a ColumnConst of const_rows rows
    size_t const_rows = 0;
};

/// ConstantFilterDescription (src/Columns/FilterDescription.cpp:20-48): a constant UInt8 filter column keeps every row or none; the one
/// stored byte is read back once.
struct ConstantFilterDescription
{
    bool always_false = false;
    bool always_true = false;
    ConstantFilterDescription() = default;
    explicit ConstantFilterDescription(const ColumnVector & column)
    {
        if (!column.isConst())
            return;
        if (column.getDataType() != CHGPU_U8)
            throw Exception(CHGPU_ERR_BAD_ARGUMENTS, "Illegal type of column for constant filter. Must be UInt8 or Nullable(UInt8).");
        if (column.getDataColumnPtr()->getData<uint8_t>().at(0))
            always_true = true;
        else
            always_false = true;
    }
};

/// Chunk (src/Processors/Chunk.h:55-128): columns + row count, move-only in spirit.
struct Chunk
{
    Columns columns;
    size_t num_rows = 0;
    bool empty() const { return num_rows == 0; }
    void clear() { columns.clear(); num_rows = 0; }
    explicit operator bool() const { return !columns.empty() || num_rows; }
};

/// Batches host Blocks into one HBM stripe per column: a 65 409-row Block is ~0.5 MB, far below what fills 256 CUs.
/// Two page-locked staging buffers: flush() queues the stripe's upload on the context's copy stream and returns at once, so the
/// host fills the other buffer -- and the device runs the previous stripe's kernels -- while the copy is in flight
/// (chgpu_col_upload_async).  A buffer is refilled only after the upload that read it has finished.
template <typename T>
class StripeBuilder
{
public:
    explicit StripeBuilder(ContextPtr ctx_, size_t stripe_rows_ = size_t(16) << 20) : ctx(std::move(ctx_)), stripe_rows(stripe_rows_ ? stripe_rows_ : 1)
    {
        for (auto & b : buf)
        {
            void * p = nullptr;
            check(chgpu_host_alloc(stripe_rows * sizeof(T), &p));
            b = static_cast<T *>(p);
        }
    }
    ~StripeBuilder()
    {
        for (int k = 0; k < 2; ++k)
        {
            chgpu_upload_wait(ctx->get(), ticket[k]);
            chgpu_host_free(buf[k]);
        }
    }
    StripeBuilder(const StripeBuilder &) = delete;
    /// rows that still fit into the current stripe
    size_t room() const { return stripe_rows - filled; }
    size_t rows() const { return filled; }
    /// appends at most room() rows of the Block and returns how many it took: the caller flushes and appends the rest
    size_t appendBlock(const T * data, size_t n)
    {
        if (filled == 0)
            check(chgpu_upload_wait(ctx->get(), ticket[cur])); // the upload that last read this buffer
        const size_t take = n < room() ? n : room();
        std::memcpy(buf[cur] + filled, data, take * sizeof(T));
        filled += take;
        return take;
    }
    /// the stripe as a device column; the copy runs asynchronously, ordered before anything launched on the context afterwards
    ColumnPtr flush()
    {
        chgpu_col * col = nullptr;
        check(chgpu_col_upload_async(ctx->get(), TypeTag<T>::value, buf[cur], filled, &col, &ticket[cur]));
        filled = 0;
        cur ^= 1;
        return std::make_shared<ColumnVector>(ctx, col);
    }

private:
    ContextPtr ctx;
    size_t stripe_rows;
    T * buf[2] = {nullptr, nullptr};
    uint64_t ticket[2] = {0, 0};
    int cur = 0;
    size_t filled = 0;
};

/// IFunction for `col <op> constant` (FunctionsComparison.h:204-245): executeImpl -> UInt8 column.
class FunctionComparisonConst
{
public:
    template <typename S>
    FunctionComparisonConst(int op_, S scalar_) : op(op_), scalar_type(TypeTag<S>::value)
    {
        static_assert(sizeof(S) <= 8);
        std::memcpy(scalar, &scalar_, sizeof(S));
    }
    ColumnPtr executeImpl(const ColumnVector & arg) const
    {
        chgpu_col * out = nullptr;
        check(chgpu_cmp_const(arg.context()->get(), arg.handle(), op, scalar_type, scalar, &out));
        return std::make_shared<ColumnVector>(arg.context(), out);
    }
    int op;
    int scalar_type;
    unsigned char scalar[8] = {0};
};

/// IProcessor (src/Processors/IProcessor.h:119-376), reduced to the synchronous part the hot path needs.
class IProcessor
{
public:
    enum class Status { NeedData, PortFull, Finished, Ready };
    virtual ~IProcessor() = default;
    virtual std::string getName() const = 0;
    /// IProcessor::prepare (IProcessor.h:170-220): what the executor may do next -- nothing here blocks or computes
    virtual Status prepare() { return Status::Ready; }
    virtual void work() = 0;
    uint64_t elapsed_ns = 0; // IProcessor.h:359
};

/// ISimpleTransform (ISimpleTransform.h:42): transform(Chunk &) consumes one chunk, may produce one.
class ISimpleTransform : public IProcessor
{
public:
    void setInput(Chunk c) { input = std::move(c); has_input = true; }
    bool hasOutput() const { return has_output; }
    Chunk pullOutput()
    {
        has_output = false;
        if (output_exception)
        {
            auto e = output_exception;
            output_exception = nullptr;
            std::rethrow_exception(e);
        }
        return std::move(output);
    }
    /// ISimpleTransform::work (ISimpleTransform.cpp:76-108): transform() runs inside try/catch; a thrown exception does NOT unwind the
    /// executor thread -- it is parked in the output slot (output_data.exception, :88-99) and rethrown by whoever pulls the port.
    void work() override
    {
        if (!has_input)
            return;
        Chunk c = std::move(input);
        has_input = false;
        try
        {
            transform(c);
        }
        catch (...)
        {
            output.clear();
            output_exception = std::current_exception(); // ISimpleTransform.cpp:90-98
            has_output = true;
            return;
        }
        if (c.num_rows != 0) // ISimpleTransform.cpp:101-107: empty chunks are skipped
        {
            output = std::move(c);
            has_output = true;
        }
    }
    /// the consumer's side of the port: an exception travels in the data slot and surfaces on pull (Port.h: pullData rethrows)
    bool outputHasException() const { return static_cast<bool>(output_exception); }

    /// ISimpleTransform::prepare (ISimpleTransform.cpp:30-74), with the two one-chunk ports this mirror has: the consumer closed the output
    /// -> Finished; a chunk is still waiting in the output -> PortFull; a chunk is waiting in the input -> Ready (work() may run); the
    /// producer finished -> Finished; otherwise NeedData.
    Status prepare() override
    {
        if (output_closed)
            return Status::Finished;
        if (has_output)
            return Status::PortFull;
        if (has_input)
            return Status::Ready;
        if (input_finished)
            return Status::Finished;
        return Status::NeedData;
    }
    void finishInput() { input_finished = true; }  ///< InputPort::isFinished on the producer's side
    void closeOutput() { output_closed = true; }   ///< OutputPort::isFinished: the consumer needs no more (LIMIT reached)
    bool needsInput() const { return !has_input && !has_output && !input_finished && !output_closed; }

protected:
    virtual void transform(Chunk & chunk) = 0;
    Chunk input, output;
    bool has_input = false, has_output = false, input_finished = false, output_closed = false;
    std::exception_ptr output_exception;
};

/// A linear pipeline source -> t[0] -> ... -> t[k-1] -> sink driven the way PipelineExecutor drives a graph (ExecutingGraph::updateNode,
/// src/Processors/Executors/ExecutingGraph.cpp): every processor is asked prepare(); Ready -> work(); PortFull -> the chunk moves to the
/// next processor's input when that one reports NeedData; NeedData at the head pulls the source; Finished propagates downstream.
/// source(chunk) returns false when it is exhausted.
template <typename Source, typename Sink>
inline void executeChain(Source && source, const std::vector<ISimpleTransform *> & chain, Sink && sink)
{
    bool source_done = false;
    for (;;)
    {
        bool progressed = false;
        for (size_t k = chain.size(); k-- > 0;) // downstream first: ports drain before they are filled again
        {
            ISimpleTransform * t = chain[k];
            switch (t->prepare())
            {
                case IProcessor::Status::Ready:
                    t->work();
                    progressed = true;
                    break;
                case IProcessor::Status::PortFull:
                    if (k + 1 == chain.size())
                    {
                        sink(t->pullOutput());
                        progressed = true;
                    }
                    else if (chain[k + 1]->needsInput())
                    {
                        chain[k + 1]->setInput(t->pullOutput());
                        progressed = true;
                    }
                    break;
                case IProcessor::Status::NeedData:
                    if (k == 0)
                    {
                        Chunk c;
                        if (!source_done && source(c))
                            t->setInput(std::move(c));
                        else
                        {
                            source_done = true;
                            t->finishInput();
                        }
                        progressed = true;
                    }
                    else if (chain[k - 1]->prepare() == IProcessor::Status::Finished)
                    {
                        t->finishInput();
                        progressed = true;
                    }
                    break;
                case IProcessor::Status::Finished:
                    break;
            }
        }
        if (chain.empty() || chain.back()->prepare() == IProcessor::Status::Finished)
            return;
        if (!progressed)
            throw Exception(CHGPU_ERR_LOGICAL, "Pipeline stuck");
    }
}

/// FilterTransform::doTransform (FilterTransform.cpp:136-256): expression -> filter column -> filter every column.
/// Every column of a chunk filtered by one mask: the loop `for (auto & col : columns) col = col->filter(filter, hint)` of
/// FilterTransform::doTransform (FilterTransform.cpp:238-252) and of joinBlock (HashJoinMethodsImpl.h:122-123) as ONE call --
/// the mask is counted and scanned once on the device, one host synchronisation for the whole chunk.
inline void filterColumns(Columns & columns, const ColumnVector & filter, ssize_t result_size_hint)
{
    if (columns.empty())
        return;
    std::vector<const chgpu_col *> in;
    for (auto & c : columns)
        in.push_back(c->handle());
    std::vector<chgpu_col *> out(columns.size(), nullptr);
    uint64_t rows = 0;
    check(chgpu_filter_columns(filter.context()->get(), static_cast<uint32_t>(in.size()), in.data(), filter.handle(), result_size_hint, out.data(), &rows));
    for (size_t k = 0; k < columns.size(); ++k)
        columns[k] = std::make_shared<ColumnVector>(filter.context(), out[k]);
}

/// joinBlock's `columns[i] = columns[i]->replicate(offsets)` over every left column (HashJoinMethodsImpl.h:186-194) as ONE call: the total
/// is read back once, columns of one element width share a kernel.
inline void replicateColumns(Columns & columns, const ColumnVector & offsets)
{
    if (columns.empty())
        return;
    std::vector<const chgpu_col *> in;
    for (auto & c : columns)
        in.push_back(c->handle());
    std::vector<chgpu_col *> out(columns.size(), nullptr);
    check(chgpu_replicate_columns(offsets.context()->get(), static_cast<uint32_t>(in.size()), in.data(), offsets.handle(), out.data()));
    for (size_t k = 0; k < columns.size(); ++k)
        columns[k] = std::make_shared<ColumnVector>(offsets.context(), out[k]);
}

class GpuFilterTransform : public ISimpleTransform
{
public:
    GpuFilterTransform(size_t filter_arg_position_, FunctionComparisonConst predicate_)
        : filter_arg_position(filter_arg_position_), predicate(std::move(predicate_)) {}
    /// the filter column is already in the chunk (an ExpressionTransform in front computed it, or the planner folded it to a constant)
    GpuFilterTransform(size_t filter_column_position_, bool remove_filter_column_)
        : filter_arg_position(filter_column_position_), has_filter_column(true), remove_filter_column(remove_filter_column_) {}
    std::string getName() const override { return "GpuFilterTransform"; }
    uint64_t passed_rows = 0; // ProfileEvents::FilterTransformPassedRows

protected:
    void removeFilterIfNeed(Columns & columns) const
    {
        if (has_filter_column && remove_filter_column)
            columns.erase(columns.begin() + static_cast<std::ptrdiff_t>(filter_arg_position));
    }
    /// FilterTransform::doTransform over a filter column of the chunk (FilterTransform.cpp:153-252)
    void transformWithFilterColumn(Chunk & chunk)
    {
        const ColumnPtr filter_column = chunk.columns.at(filter_arg_position);
        // :168-176 -- a column that turned out constant: every row or none, no kernel
        const ConstantFilterDescription constant_filter_description(*filter_column);
        if (constant_filter_description.always_true)
        {
            passed_rows += chunk.num_rows;
            removeFilterIfNeed(chunk.columns);
            return;
        }
        if (constant_filter_description.always_false)
        {
            chunk.clear(); // "Will finish at next prepare call"
            return;
        }
        if (filter_column->getDataType() != CHGPU_U8)
            throw Exception(CHGPU_ERR_BAD_ARGUMENTS, "Illegal type of column for filter. Must be UInt8 or Nullable(UInt8) or Const variants of them.");
        uint64_t num_filtered_rows = 0;
        check(chgpu_count_bytes_in_filter(filter_column->context()->get(), filter_column->handle(), &num_filtered_rows)); // :192-216
        if (num_filtered_rows == 0)
        {
            chunk.clear(); // :221-226
            return;
        }
        if (num_filtered_rows != chunk.num_rows)
        {
            // :238-252 -- constant columns are cut, the others filtered (in one call: one count + scan for the whole chunk)
            Columns to_filter;
            std::vector<size_t> where;
            for (size_t i = 0; i < chunk.columns.size(); ++i)
            {
                if (i == filter_arg_position && remove_filter_column)
                    continue;
                if (chunk.columns[i]->isConst())
                    chunk.columns[i] = chunk.columns[i]->cutConst(num_filtered_rows);
                else
                {
                    to_filter.push_back(chunk.columns[i]);
                    where.push_back(i);
                }
            }
            filterColumns(to_filter, *filter_column, static_cast<ssize_t>(num_filtered_rows));
            for (size_t k = 0; k < where.size(); ++k)
                chunk.columns[where[k]] = to_filter[k];
            chunk.num_rows = num_filtered_rows;
        }
        passed_rows += num_filtered_rows;
        removeFilterIfNeed(chunk.columns);
    }

    void transform(Chunk & chunk) override
    {
        if (has_filter_column)
        {
            transformWithFilterColumn(chunk);
            return;
        }
        const auto & arg = *chunk.columns.at(filter_arg_position);
        auto mask = predicate.executeImpl(arg); // :146-147
        uint64_t num_filtered_rows = 0;
        check(chgpu_count_bytes_in_filter(arg.context()->get(), mask->handle(), &num_filtered_rows)); // :192-216
        if (num_filtered_rows == 0)
        {
            chunk.clear(); // :221-226: the chunk is dropped
            return;
        }
        if (num_filtered_rows == chunk.num_rows) // :229-235: all rows pass, columns untouched
        {
            passed_rows += num_filtered_rows;
            return;
        }
        filterColumns(chunk.columns, *mask, static_cast<ssize_t>(num_filtered_rows)); // :238-252
        chunk.num_rows = num_filtered_rows;
        passed_rows += num_filtered_rows;
    }

private:
    size_t filter_arg_position;
    FunctionComparisonConst predicate{CHGPU_NE, uint8_t(0)};
    bool has_filter_column = false;
    bool remove_filter_column = false;
};

/// ActionsDAG (src/Interpreters/ActionsDAG.h): INPUT / COLUMN (constant) / FUNCTION nodes under the reference's function
/// names; compile() is ExpressionActions' constructor (ExpressionActions.cpp:71-110) plus the run-time compiler the reference
/// runs under compile_expressions (src/Interpreters/JIT/compileFunction.cpp): the whole DAG becomes one kernel.
class ActionsDAG
{
public:
    using Node = uint32_t;
    Node addInput(size_t position, int type)
    {
        chgpu_expr_node n{CHGPU_EX_INPUT, static_cast<int32_t>(position), type, {-1, -1, -1}, 0};
        nodes.push_back(n);
        return static_cast<Node>(nodes.size() - 1);
    }
    template <typename T>
    Node addColumn(T constant)
    {
        chgpu_expr_node n{CHGPU_EX_CONST, 0, TypeTag<T>::value, {-1, -1, -1}, 0};
        std::memcpy(&n.bits, &constant, sizeof(T));
        nodes.push_back(n);
        return static_cast<Node>(nodes.size() - 1);
    }
    Node addFunction(const std::string & name, std::initializer_list<Node> args)
    {
        chgpu_expr_node n{CHGPU_EX_FUNC, functionCode(name), 0, {-1, -1, -1}, 0};
        size_t j = 0;
        for (Node a : args)
            n.args[j++] = static_cast<int32_t>(a);
        nodes.push_back(n);
        return static_cast<Node>(nodes.size() - 1);
    }
    /// FunctionFactory::get(name): unknown names are the caller's cue to keep its CPU actions
    static int functionCode(const std::string & name)
    {
        static const std::pair<const char *, int> table[] = {
            {"equals", CHGPU_FN_EQUALS}, {"notEquals", CHGPU_FN_NOT_EQUALS}, {"less", CHGPU_FN_LESS}, {"greater", CHGPU_FN_GREATER},
            {"lessOrEquals", CHGPU_FN_LESS_OR_EQUALS}, {"greaterOrEquals", CHGPU_FN_GREATER_OR_EQUALS}, {"plus", CHGPU_FN_PLUS},
            {"minus", CHGPU_FN_MINUS}, {"multiply", CHGPU_FN_MULTIPLY}, {"divide", CHGPU_FN_DIVIDE}, {"negate", CHGPU_FN_NEGATE}, {"intDiv", CHGPU_FN_INT_DIV}, {"modulo", CHGPU_FN_MODULO},
            {"and", CHGPU_FN_AND}, {"or", CHGPU_FN_OR}, {"xor", CHGPU_FN_XOR}, {"not", CHGPU_FN_NOT}, {"if", CHGPU_FN_IF},
            {"bitAnd", CHGPU_FN_BIT_AND}, {"bitOr", CHGPU_FN_BIT_OR}, {"bitXor", CHGPU_FN_BIT_XOR}, {"toYear", CHGPU_FN_TO_YEAR},
            {"toMonth", CHGPU_FN_TO_MONTH}, {"toDayOfMonth", CHGPU_FN_TO_DAY_OF_MONTH}, {"toYYYYMM", CHGPU_FN_TO_YYYYMM},
            {"toYYYYMMDD", CHGPU_FN_TO_YYYYMMDD}, {"toDayOfWeek", CHGPU_FN_TO_DAY_OF_WEEK}, {"toQuarter", CHGPU_FN_TO_QUARTER},
            {"toStartOfMonth", CHGPU_FN_TO_START_OF_MONTH},
            {"toInt64", CHGPU_FN_CAST + CHGPU_I64}, {"toUInt64", CHGPU_FN_CAST + CHGPU_U64}, {"toInt32", CHGPU_FN_CAST + CHGPU_I32},
            {"toUInt32", CHGPU_FN_CAST + CHGPU_U32}, {"toInt16", CHGPU_FN_CAST + CHGPU_I16}, {"toUInt16", CHGPU_FN_CAST + CHGPU_U16},
            {"toInt8", CHGPU_FN_CAST + CHGPU_I8}, {"toUInt8", CHGPU_FN_CAST + CHGPU_U8}, {"toFloat64", CHGPU_FN_CAST + CHGPU_F64},
            {"toFloat32", CHGPU_FN_CAST + CHGPU_F32}};
        for (const auto & e : table)
            if (name == e.first)
                return e.second;
        throw Exception(CHGPU_ERR_NOT_IMPLEMENTED, "Unknown function " + name);
    }
    std::vector<chgpu_expr_node> nodes;
};

/// ExpressionActions (ExpressionActions.h:75-134): execute(Block &) appends the result columns of the requested nodes.
class ExpressionActions
{
public:
    explicit ExpressionActions(const ActionsDAG & dag) { check(chgpu_expr_compile(static_cast<uint32_t>(dag.nodes.size()), dag.nodes.data(), &h)); }
    ~ExpressionActions() { chgpu_expr_free(h); }
    ExpressionActions(const ExpressionActions &) = delete;
    int resultType(ActionsDAG::Node node) const
    {
        int t = 0;
        check(chgpu_expr_node_type(h, node, &t));
        return t;
    }
    /// columns[position] feeds INPUT(position); one new column per entry of `outputs`
    Columns execute(const Columns & columns, const std::vector<ActionsDAG::Node> & outputs) const
    {
        std::vector<const chgpu_col *> in;
        ContextPtr ctx;
        for (auto & c : columns)
        {
            in.push_back(c ? c->handle() : nullptr);
            if (c && !ctx)
                ctx = c->context();
        }
        std::vector<chgpu_col *> out(outputs.size(), nullptr);
        check(chgpu_expr_execute(ctx->get(), h, static_cast<uint32_t>(in.size()), in.data(), static_cast<uint32_t>(outputs.size()), outputs.data(), out.data()));
        Columns res;
        for (chgpu_col * o : out)
            res.push_back(std::make_shared<ColumnVector>(ctx, o));
        return res;
    }
    /// SELECT sum(value), count() WHERE filter over one chunk, nothing materialised; returns the raw 8 state bytes and the count
    std::pair<uint64_t, uint64_t> filterSum(const Columns & columns, int filter_node, int value_node, int * result_type = nullptr) const
    {
        std::vector<const chgpu_col *> in;
        ContextPtr ctx;
        for (auto & c : columns)
        {
            in.push_back(c ? c->handle() : nullptr);
            if (c && !ctx)
                ctx = c->context();
        }
        uint64_t bits = 0, count = 0;
        check(chgpu_expr_filter_sum_node(ctx->get(), h, static_cast<uint32_t>(in.size()), in.data(), filter_node, value_node, result_type, &bits, &count));
        return {bits, count};
    }

    /// SELECT min(value), max(value), count() WHERE filter over one chunk (integer value nodes); the bits are the value in its own width
    struct MinMax { uint64_t min_bits = 0, max_bits = 0, count = 0; int type = 0; };
    MinMax filterMinMax(const Columns & columns, int filter_node, ActionsDAG::Node value_node) const
    {
        std::vector<const chgpu_col *> in;
        ContextPtr ctx;
        for (auto & c : columns)
        {
            in.push_back(c ? c->handle() : nullptr);
            if (c && !ctx)
                ctx = c->context();
        }
        MinMax r;
        check(chgpu_expr_filter_minmax_node(ctx->get(), h, static_cast<uint32_t>(in.size()), in.data(), filter_node, value_node, &r.type, &r.min_bits, &r.max_bits, &r.count));
        return r;
    }

private:
    chgpu_expr * h = nullptr;
};

/// ExpressionTransform (src/Processors/Transforms/ExpressionTransform.cpp:22-30): the chunk gains the result columns.
class GpuExpressionTransform : public ISimpleTransform
{
public:
    GpuExpressionTransform(std::shared_ptr<const ExpressionActions> actions_, std::vector<ActionsDAG::Node> outputs_)
        : actions(std::move(actions_)), outputs(std::move(outputs_)) {}
    std::string getName() const override { return "GpuExpressionTransform"; }

protected:
    void transform(Chunk & chunk) override
    {
        for (auto & c : actions->execute(chunk.columns, outputs))
            chunk.columns.push_back(std::move(c));
    }

private:
    std::shared_ptr<const ExpressionActions> actions;
    std::vector<ActionsDAG::Node> outputs;
};

/// FilterTransform with a whole WHERE expression (FilterTransform.cpp:136-256): the DAG's filter node is evaluated by one
/// kernel into the UInt8 filter column, then the chunk's columns are filtered as in GpuFilterTransform.
class GpuExpressionFilterTransform : public ISimpleTransform
{
public:
    GpuExpressionFilterTransform(std::shared_ptr<const ExpressionActions> actions_, ActionsDAG::Node filter_node_)
        : actions(std::move(actions_)), filter_node(filter_node_) {}
    std::string getName() const override { return "GpuExpressionFilterTransform"; }
    uint64_t passed_rows = 0;

protected:
    void transform(Chunk & chunk) override
    {
        auto mask = actions->execute(chunk.columns, {filter_node}).at(0);
        uint64_t kept = 0;
        check(chgpu_count_bytes_in_filter(mask->context()->get(), mask->handle(), &kept));
        if (kept == 0)
        {
            chunk.clear();
            return;
        }
        passed_rows += kept;
        if (kept == chunk.num_rows)
            return;
        filterColumns(chunk.columns, *mask, static_cast<ssize_t>(kept));
        chunk.num_rows = kept;
    }

private:
    std::shared_ptr<const ExpressionActions> actions;
    ActionsDAG::Node filter_node;
};

/// ColumnLowCardinality (src/Columns/ColumnLowCardinality.h:27-69): the dictionary stays on the host (a few thousand values,
/// low_cardinality_max_dictionary_size = 8192), the index column lives in HBM.
struct ColumnLowCardinality
{
    std::shared_ptr<const std::vector<std::string>> dictionary; // position -> value (ColumnUnique's nested column)
    ColumnPtr indexes;                                          // UInt8 / UInt16 / UInt32 / UInt64
};

/// ColumnString (src/Columns/ColumnString.h:40-49) pinned into HBM: chars (every value followed by a zero byte) + cumulative offsets.
/// dictionaryEncode() turns it into a ColumnLowCardinality on the device (ids by first appearance, ColumnUnique.h:520-620); the
/// dictionary's strings are read from the caller's own Block at the returned first rows, so no string leaves the device.
struct ColumnString
{
    ColumnPtr offsets; // UInt64
    ColumnPtr chars;   // UInt8

    /// value_of(row) -> the string at `row` of the host Block this column was pinned from
    template <typename ValueOf>
    ColumnLowCardinality dictionaryEncode(ValueOf && value_of) const
    {
        chgpu_col * ids = nullptr, * rows = nullptr;
        uint64_t n = 0;
        check(chgpu_string_dictionary_encode(offsets->context()->get(), offsets->handle(), chars->handle(), &ids, &rows, &n));
        ColumnVector first(offsets->context(), rows);
        auto first_rows = first.getData<uint64_t>();
        auto dict = std::make_shared<std::vector<std::string>>();
        dict->reserve(first_rows.size());
        for (uint64_t r : first_rows)
            dict->push_back(value_of(r));
        return ColumnLowCardinality{dict, std::make_shared<ColumnVector>(offsets->context(), ids)};
    }
};

/// The query-wide dictionary of one LowCardinality key: what the low_cardinality_key* aggregation methods
/// (AggregatedDataVariants.h:119-127) achieve by emplacing each Block's dictionary entries once and walking the rows through
/// HashMethodSingleLowCardinalityColumn's per-position cache (ColumnsHashing.h:82-260).  mapBlock() returns an ordinary UInt32
/// key column (dense global ids) for GpuAggregator / GpuHashJoin; decode() turns result keys back into values.
class LowCardinalityDictionary
{
public:
    explicit LowCardinalityDictionary(ContextPtr ctx_) : ctx(std::move(ctx_)) {}
    ColumnPtr mapBlock(const ColumnLowCardinality & col)
    {
        if (col.dictionary != cached_dictionary) // Blocks of one part share their dictionary: resolve it once
        {
            std::vector<uint32_t> remap(col.dictionary->size());
            for (size_t pos = 0; pos < remap.size(); ++pos)
            {
                auto it = ids.find((*col.dictionary)[pos]);
                if (it == ids.end())
                {
                    it = ids.emplace((*col.dictionary)[pos], static_cast<uint32_t>(values.size())).first;
                    values.push_back((*col.dictionary)[pos]);
                }
                remap[pos] = it->second;
            }
            cached_remap = ColumnVector::fromHost<uint32_t>(ctx, remap.data(), remap.size());
            cached_dictionary = col.dictionary;
        }
        chgpu_col * out = nullptr;
        check(chgpu_lc_remap(ctx->get(), col.indexes->handle(), cached_remap->handle(), &out));
        return std::make_shared<ColumnVector>(ctx, out);
    }
    const std::string & decode(uint32_t id) const { return values.at(id); }
    size_t size() const { return values.size(); }

private:
    ContextPtr ctx;
    std::vector<std::string> values;
    std::unordered_map<std::string, uint32_t> ids;
    std::shared_ptr<const std::vector<std::string>> cached_dictionary;
    ColumnPtr cached_remap;
};

/// AggregateDescription (src/Interpreters/AggregateDescription.h): function + argument position.
struct AggregateDescription
{
    int kind;              // CHGPU_AGG_*
    int argument_type;     // CHGPU_* (ignored for count)
    size_t argument = 0;   // position in the chunk
};

/// Aggregator + AggregatedDataVariants (Aggregator.h:179-265) for one numeric key or no key.
class GpuAggregator
{
public:
    GpuAggregator(ContextPtr ctx_, int key_type_, std::vector<AggregateDescription> aggregates_, uint64_t size_hint = 0)
        : ctx(std::move(ctx_)), key_type(key_type_), aggregates(std::move(aggregates_))
    {
        std::vector<int> kinds, types;
        for (auto & a : aggregates)
        {
            kinds.push_back(a.kind);
            types.push_back(a.argument_type);
        }
        check(chgpu_agg_create(ctx->get(), key_type, static_cast<uint32_t>(aggregates.size()), kinds.data(), types.data(), size_hint, &h));
    }
    ~GpuAggregator() { chgpu_agg_free(h); }
    GpuAggregator(const GpuAggregator &) = delete;

    /// Aggregator::executeOnBlock(columns, row_begin, row_end, ...): returns false to stop reading (never here)
    bool executeOnBlock(const Columns & columns, size_t row_begin, size_t row_end, std::optional<size_t> key_position)
    {
        std::vector<const chgpu_col *> args;
        for (auto & a : aggregates)
            args.push_back(a.kind == CHGPU_AGG_COUNT ? nullptr : columns.at(a.argument)->handle());
        const chgpu_col * key = key_position ? columns.at(*key_position)->handle() : nullptr;
        check(chgpu_agg_add_block(h, key, args.data(), row_begin, row_end));
        return true;
    }
    /// mergeDataImpl
    void merge(const GpuAggregator & other) { check(chgpu_agg_merge(h, other.h)); }
    /// convertToBlocks(final = true): [key column,] one column per aggregate
    Chunk convertToBlock() const
    {
        chgpu_col * keys = nullptr;
        std::vector<chgpu_col *> res(aggregates.size(), nullptr);
        uint64_t groups = 0;
        check(chgpu_agg_finalize(h, &keys, res.data(), &groups));
        Chunk out;
        out.num_rows = groups;
        if (keys)
            out.columns.push_back(std::make_shared<ColumnVector>(ctx, keys));
        for (auto * r : res)
            out.columns.push_back(std::make_shared<ColumnVector>(ctx, r));
        return out;
    }
    size_t size() const
    {
        uint64_t n = 0;
        check(chgpu_agg_size(h, &n));
        return n;
    }

private:
    ContextPtr ctx;
    int key_type;
    std::vector<AggregateDescription> aggregates;
    chgpu_agg * h = nullptr;
};

/// ManyAggregatedData (AggregatingTransform.h:74-100): one AggregatedDataVariants per pipeline stream, shared by the streams'
/// AggregatingTransforms; the stream that finishes LAST merges them (num_finished, AggregatingTransform.cpp:728-744).
struct ManyAggregatedData
{
    std::vector<std::shared_ptr<GpuAggregator>> variants;
    std::atomic<uint32_t> num_finished{0};
    explicit ManyAggregatedData(std::vector<std::shared_ptr<GpuAggregator>> variants_) : variants(std::move(variants_)) {}
};
using ManyAggregatedDataPtr = std::shared_ptr<ManyAggregatedData>;

/// AggregatingTransform (AggregatingTransform.cpp:523-840): consume() per chunk on its own stream's variant; when the input ends, work()
/// runs initGenerate: every stream but the last only reports in, the last one folds all variants into the first
/// (Aggregator::mergeDataImpl: insert-or-merge, chgpu_agg_merge) and becomes the generating transform.
class GpuAggregatingTransform : public IProcessor
{
public:
    /// one stream, one variant (nothing to merge)
    GpuAggregatingTransform(std::shared_ptr<GpuAggregator> aggregator_, std::optional<size_t> key_position_)
        : many_data(std::make_shared<ManyAggregatedData>(std::vector<std::shared_ptr<GpuAggregator>>{std::move(aggregator_)})), current_variant(0),
          key_position(key_position_) {}
    /// stream `current_variant_` of `many_data_->variants.size()` streams
    GpuAggregatingTransform(ManyAggregatedDataPtr many_data_, size_t current_variant_, std::optional<size_t> key_position_)
        : many_data(std::move(many_data_)), current_variant(current_variant_), key_position(key_position_) {}
    std::string getName() const override { return "GpuAggregatingTransform"; }
    void consume(Chunk chunk)
    {
        if (chunk.num_rows == 0)
            return;
        src_rows += chunk.num_rows;
        many_data->variants.at(current_variant)->executeOnBlock(chunk.columns, 0, chunk.num_rows, key_position); // :664-693
    }
    /// the input port is finished: initGenerate (:695-744)
    void work() override
    {
        if (is_generate_initialized)
            return;
        is_generate_initialized = true;
        if (many_data->num_finished.fetch_add(1) + 1 < many_data->variants.size())
            return; // not the last stream: its variant stays in many_data for the one that is
        auto & variants = many_data->variants;
        for (size_t i = 1; i < variants.size(); ++i) // mergeDataImpl into the first (prepareVariantsToMerge keeps the largest first; sizes need a read-back each)
            variants[0]->merge(*variants[i]);
        is_last = true;
    }
    /// true on exactly one of the streams' transforms after every stream's work(): the one that generates
    bool isGenerating() const { return is_last; }
    Chunk generate()
    {
        if (!is_generate_initialized)
            work();
        if (!is_last)
            return Chunk{};
        return many_data->variants[0]->convertToBlock();
    }
    uint64_t src_rows = 0;

private:
    ManyAggregatedDataPtr many_data;
    size_t current_variant;
    std::optional<size_t> key_position;
    bool is_generate_initialized = false;
    bool is_last = false;
};

/// FilterTransform + AggregatingTransform without key, fused: `SELECT sum(val), count() WHERE pred <op> constant` in one pass over the
/// stripe (chgpu_filter_sum: FilterTransform.cpp:136-256 -> Aggregator::executeWithoutKeyImpl, Aggregator.cpp:1276-1321) -- no mask, no
/// filtered column.  Integer sums (wrap-around, SumSimple's UInt64 / Int64 state).
class GpuFilterSumTransform : public IProcessor
{
public:
    GpuFilterSumTransform(size_t pred_position_, FunctionComparisonConst predicate_, size_t value_position_)
        : pred_position(pred_position_), predicate(std::move(predicate_)), value_position(value_position_) {}
    std::string getName() const override { return "GpuFilterSumTransform"; }
    void consume(const Chunk & chunk)
    {
        if (chunk.num_rows == 0)
            return;
        const auto & pred = *chunk.columns.at(pred_position);
        uint64_t s = 0, c = 0;
        check(chgpu_filter_sum(pred.context()->get(), pred.handle(), predicate.op, predicate.scalar_type, predicate.scalar, chunk.columns.at(value_position)->handle(), &s, &c));
        sum += s;
        count += c;
        src_rows += chunk.num_rows;
    }
    void work() override {}
    uint64_t sum = 0, count = 0, src_rows = 0;

private:
    size_t pred_position;
    FunctionComparisonConst predicate;
    size_t value_position;
};

/// IJoin for an ASOF join (JoinStrictness::Asof: INNER / LEFT; RowRefs.cpp SortedLookupVector, HashJoinMethodsImpl.h:462-478): equality on one
/// fixed-width integer key, `left.asof <inequality> right.asof` on one numeric column.  The right Blocks are glued into one before they
/// are inserted, so a right row id is the row's ordinal in the glued payload columns.
class GpuAsofJoin
{
public:
    GpuAsofJoin(ContextPtr ctx_, int key_type, int asof_type, int kind_, int inequality = CHGPU_ASOF_GREATER_OR_EQUALS) : ctx(std::move(ctx_)), kind(kind_)
    {
        check(chgpu_asof_create(ctx->get(), key_type, asof_type, kind, inequality, &h));
    }
    ~GpuAsofJoin() { chgpu_asof_free(h); }
    GpuAsofJoin(const GpuAsofJoin &) = delete;

    bool addBlockToJoin(const Chunk & block, size_t key_position_, size_t asof_position_)
    {
        key_position = key_position_, asof_position = asof_position_;
        right_blocks.push_back(block);
        return true;
    }
    void onBuildPhaseFinish()
    {
        if (built)
            return;
        built = true;
        if (right_blocks.empty())
            return;
        for (size_t c = 0; c < right_blocks[0].columns.size(); ++c)
        {
            if (right_blocks.size() == 1)
            {
                right_payload.push_back(right_blocks[0].columns[c]);
                continue;
            }
            std::vector<const chgpu_col *> parts;
            for (auto & b : right_blocks)
                parts.push_back(b.columns.at(c)->handle());
            chgpu_col * cat = nullptr;
            check(chgpu_col_concat(ctx->get(), static_cast<uint32_t>(parts.size()), parts.data(), &cat));
            right_payload.push_back(std::make_shared<ColumnVector>(ctx, cat));
        }
        check(chgpu_asof_add_block(h, right_payload.at(key_position)->handle(), right_payload.at(asof_position)->handle(), nullptr, nullptr, nullptr));
    }
    /// joinBlock: [left columns..., right columns...]; INNER keeps the left rows that found a partner, LEFT keeps all (default right values)
    void joinBlock(Chunk & block, size_t left_key_position, size_t left_asof_position)
    {
        onBuildPhaseFinish();
        chgpu_col *filter = nullptr, *rowid = nullptr;
        uint64_t n_out = 0;
        check(chgpu_asof_probe(h, block.columns.at(left_key_position)->handle(), block.columns.at(left_asof_position)->handle(), nullptr, &filter, &rowid, &n_out));
        auto filter_c = std::make_shared<ColumnVector>(ctx, filter);
        auto rowid_c = std::make_shared<ColumnVector>(ctx, rowid);
        Chunk res;
        res.columns = block.columns;
        if (kind == CHGPU_JOIN_INNER)
            filterColumns(res.columns, *filter_c, static_cast<ssize_t>(n_out));
        for (auto & col : right_payload)
        {
            chgpu_col * out = nullptr;
            check(chgpu_index(ctx->get(), col->handle(), rowid_c->handle(), 0, /*default_for_missing*/ 1, &out));
            res.columns.push_back(std::make_shared<ColumnVector>(ctx, out));
        }
        res.num_rows = n_out;
        block = std::move(res);
    }

private:
    ContextPtr ctx;
    int kind;
    chgpu_asof * h = nullptr;
    std::vector<Chunk> right_blocks;
    Columns right_payload;
    size_t key_position = 0, asof_position = 0;
    bool built = false;
};

/// IJoin (IJoin.h:80-142) for HashJoin key64.
class GpuHashJoin
{
public:
    GpuHashJoin(ContextPtr ctx_, int key_type, int kind_, int strictness_, bool any_take_last_row = false)
        : ctx(std::move(ctx_)), kind(kind_), strictness(strictness_)
    {
        check(chgpu_join_create(ctx->get(), key_type, kind, strictness, any_take_last_row ? 1 : 0, 0, &h));
    }
    ~GpuHashJoin() { chgpu_join_free(h); }
    GpuHashJoin(const GpuHashJoin &) = delete;

    /// addBlockToJoin(block): key column at key_position; the other columns are kept as the right payload
    bool addBlockToJoin(const Chunk & block, size_t key_position)
    {
        uint32_t idx = 0;
        check(chgpu_join_add_block(h, block.columns.at(key_position)->handle(), nullptr, nullptr, &idx));
        right_blocks.push_back(block);
        return true;
    }
    void onBuildPhaseFinish()
    {
        check(chgpu_join_finish_build(h));
        // glue the payload columns of all right Blocks once; probes gather from them by running row ordinal
        right_payload.clear();
        if (right_blocks.empty())
            return;
        for (size_t c = 0; c < right_blocks[0].columns.size(); ++c)
        {
            if (right_blocks.size() == 1)
            {
                right_payload.push_back(right_blocks[0].columns[c]);
                continue;
            }
            std::vector<const chgpu_col *> parts;
            for (auto & b : right_blocks)
                parts.push_back(b.columns.at(c)->handle());
            chgpu_col * cat = nullptr;
            check(chgpu_col_concat(ctx->get(), static_cast<uint32_t>(parts.size()), parts.data(), &cat));
            right_payload.push_back(std::make_shared<ColumnVector>(ctx, cat));
        }
    }
    size_t getTotalRowCount() const
    {
        uint64_t r = 0;
        check(chgpu_join_total_rows(h, &r, nullptr));
        return r;
    }
    /// JoinFeatures.h:28,31
    bool needReplication() const
    {
        return strictness == CHGPU_STRICT_ALL || (kind == CHGPU_JOIN_RIGHT && (strictness == CHGPU_STRICT_ANY || strictness == CHGPU_STRICT_SEMI));
    }
    bool needFilter() const
    {
        return !needReplication() && (kind == CHGPU_JOIN_INNER || kind == CHGPU_JOIN_RIGHT || strictness == CHGPU_STRICT_SEMI || strictness == CHGPU_STRICT_ANTI);
    }

    /// getNonJoinedBlocks (IJoin.h:133-134; NotJoinedBlocks::nextImpl, src/Interpreters/NotJoinedBlocks.cpp) for RIGHT / FULL joins, after
    /// the last joinBlock: the right payload columns of the build rows no left row matched, in insertion order.  The caller prepends
    /// default-filled left columns of `num_rows` rows (insertManyDefaults).
    Chunk getNonJoinedBlock()
    {
        if (right_payload.empty() && !right_blocks.empty())
            onBuildPhaseFinish();
        chgpu_col * rowid = nullptr;
        uint64_t n = 0;
        check(chgpu_join_non_joined_rows(h, &rowid, &n));
        ColumnPtr ids = std::make_shared<ColumnVector>(ctx, rowid);
        if (right_blocks.size() > 1)
        {
            chgpu_col * flat = nullptr;
            check(chgpu_join_flatten_rowids(h, ids->handle(), &flat));
            ids = std::make_shared<ColumnVector>(ctx, flat);
        }
        Chunk res;
        res.num_rows = n;
        for (auto & col : right_payload)
        {
            chgpu_col * out = nullptr;
            check(chgpu_index(ctx->get(), col->handle(), ids->handle(), 0, 0, &out));
            res.columns.push_back(std::make_shared<ColumnVector>(ctx, out));
        }
        return res;
    }

    /// joinBlock + `SELECT count(), sum(right payload column)` fused (chgpu_join_probe_agg): nothing per left row is materialised.
    /// -> {count, sum bits typed like SumSimple(payload type)}
    std::pair<uint64_t, uint64_t> probeCountSum(const ColumnVector & left_keys, size_t right_payload_position)
    {
        if (right_payload.empty() && !right_blocks.empty())
            onBuildPhaseFinish();
        uint64_t count = 0, sum_bits = 0;
        check(chgpu_join_probe_agg(h, left_keys.handle(), nullptr, right_payload.at(right_payload_position)->handle(), &count, &sum_bits));
        return {count, sum_bits};
    }

    /// joinBlock(block, not_processed): the left chunk is replaced by [left columns..., right payload columns...];
    /// the unprocessed tail (max_joined_block_rows) comes back in not_processed (HashJoin.cpp:1090-1093).
    /// Right payloads are gathered on the device (fillFromBlocksAndRowNumbers, IColumn.cpp:515-526) from the glued columns.
    void joinBlock(Chunk & block, size_t key_position, std::shared_ptr<Chunk> & not_processed, uint64_t max_joined_block_rows = 0)
    {
        if (right_payload.empty() && !right_blocks.empty())
            onBuildPhaseFinish();
        chgpu_col *filter = nullptr, *offsets = nullptr, *rowid = nullptr;
        uint64_t n_out = 0, consumed = 0;
        // LEFT SEMI / LEFT ANTI with no right column requested (an empty AddedColumns): the filter-only probe
        const bool filter_only = right_payload.empty() && (strictness == CHGPU_STRICT_SEMI || strictness == CHGPU_STRICT_ANTI);
        check(chgpu_join_probe(h, block.columns.at(key_position)->handle(), nullptr, max_joined_block_rows, &filter, &offsets,
                               filter_only ? nullptr : &rowid, &n_out, &consumed));
        auto filter_c = filter ? std::make_shared<ColumnVector>(ctx, filter) : nullptr;
        auto offsets_c = offsets ? std::make_shared<ColumnVector>(ctx, offsets) : nullptr;
        auto rowid_c = rowid ? std::make_shared<ColumnVector>(ctx, rowid) : nullptr;
        if (rowid && right_blocks.size() > 1)
        {
            chgpu_col * flat = nullptr; // (block, row) -> ordinal in the concatenated payload
            check(chgpu_join_flatten_rowids(h, rowid, &flat));
            rowid_c = std::make_shared<ColumnVector>(ctx, flat);
        }
        not_processed.reset();
        if (consumed < block.num_rows)
        {
            not_processed = std::make_shared<Chunk>();
            for (auto & col : block.columns)
                not_processed->columns.push_back(col->cut(consumed, block.num_rows - consumed, col));
            not_processed->num_rows = block.num_rows - consumed;
        }
        Chunk res;
        for (auto & col : block.columns)
            res.columns.push_back(consumed < block.num_rows ? col->cut(0, consumed, col) : col);
        if (offsets_c)
            replicateColumns(res.columns, *offsets_c); // HashJoinMethodsImpl.h:182-197, every left column in one call
        else if (filter_c)
            filterColumns(res.columns, *filter_c, -1); // :122-123
        for (auto & payload : right_payload)
            res.columns.push_back(payload->index(*rowid_c, 0, /*default_for_missing*/ true));
        res.num_rows = n_out;
        block = std::move(res);
    }

    chgpu_join * handle() const { return h; }
    const Columns & rightPayload()
    {
        if (right_payload.empty() && !right_blocks.empty())
            onBuildPhaseFinish();
        return right_payload;
    }
    size_t rightBlocks() const { return right_blocks.size(); }
    int getKind() const { return kind; }
    int getStrictness() const { return strictness; }

private:
    ContextPtr ctx;
    int kind, strictness;
    chgpu_join * h = nullptr;
    std::vector<Chunk> right_blocks; // data->blocks (HashJoin.cpp:656-658)
    Columns right_payload;           // the same columns glued end to end for the device gather
};

/// A chain of JoiningTransforms whose joins are all of the filter form (LEFT SEMI / LEFT ANTI / ALL over unique build keys): what
/// N x (joinBlock -> joinRightColumns -> block.filter(filter) over every left column -> AddedColumns gather; HashJoinMethodsImpl.h:68-202)
/// produce, computed before any left column is copied (chgpu_join_probe_chain): one sweep over the left KEY columns answers which rows
/// survive every join, then only the survivors are gathered -- their left columns once (IColumn::index), every join's right payload by the
/// matched build row (fillFromBlocksAndRowNumbers).  Output layout = the per-join chain's: [left columns..., payload of join 0..., payload
/// of join 1 ...].  Joins the entry does not take (INNER ANY, RIGHT / FULL, ALL over duplicate keys) make it throw NOT_IMPLEMENTED: the
/// caller keeps its JoiningTransform chain.
struct JoinChainStep
{
    std::shared_ptr<GpuHashJoin> join;
    size_t left_key_position;
};

inline void joinBlockChain(const std::vector<JoinChainStep> & steps, Chunk & block)
{
    if (steps.empty())
        return;
    const size_t n = steps.size();
    std::vector<chgpu_join *> joins;
    std::vector<const chgpu_col *> keys;
    std::vector<int> want(n, 0);
    for (size_t s = 0; s < n; ++s)
    {
        joins.push_back(steps[s].join->handle());
        keys.push_back(block.columns.at(steps[s].left_key_position)->handle());
        want[s] = steps[s].join->rightPayload().empty() ? 0 : 1;
    }
    std::vector<const chgpu_col *> carry;
    for (auto & c : block.columns)
        carry.push_back(c->handle());
    std::vector<chgpu_col *> rowids(n, nullptr), carried(carry.size(), nullptr);
    uint64_t kept = 0;
    check(chgpu_join_probe_chain(static_cast<uint32_t>(n), joins.data(), keys.data(), nullptr, want.data(), static_cast<uint32_t>(carry.size()), carry.data(), nullptr,
                                 rowids.data(), carried.data(), nullptr, &kept));
    const ContextPtr & ctx = block.columns.at(0)->context();
    Chunk res;
    res.num_rows = kept;
    for (auto * c : carried)
        res.columns.push_back(std::make_shared<ColumnVector>(ctx, c));
    for (size_t s = 0; s < n; ++s)
    {
        if (!rowids[s])
            continue;
        ColumnPtr ids = std::make_shared<ColumnVector>(ctx, rowids[s]);
        if (steps[s].join->rightBlocks() > 1)
        {
            chgpu_col * flat = nullptr;
            check(chgpu_join_flatten_rowids(steps[s].join->handle(), ids->handle(), &flat));
            ids = std::make_shared<ColumnVector>(ctx, flat);
        }
        for (auto & payload : steps[s].join->rightPayload())
            res.columns.push_back(payload->index(*ids, 0, /*default_for_missing*/ true));
    }
    block = std::move(res);
}

/// the same as ONE processor standing where the chain of JoiningTransforms stood
class GpuJoinChainTransform : public ISimpleTransform
{
public:
    explicit GpuJoinChainTransform(std::vector<JoinChainStep> steps_) : steps(std::move(steps_)) {}
    std::string getName() const override { return "GpuJoinChainTransform"; }

protected:
    void transform(Chunk & chunk) override
    {
        joinBlockChain(steps, chunk);
        if (chunk.num_rows == 0)
            chunk.clear(); // (an empty chunk is not forwarded: ISimpleTransform.cpp:101-107)
    }

private:
    std::vector<JoinChainStep> steps;
};

/// FillingRightJoinSideTransform + JoiningTransform (JoiningTransform.cpp:176-260, 347-357)
class GpuJoiningTransform : public ISimpleTransform
{
public:
    GpuJoiningTransform(std::shared_ptr<GpuHashJoin> join_, size_t left_key_position_, uint64_t max_joined_block_rows_ = 0)
        : join(std::move(join_)), left_key_position(left_key_position_), max_joined_block_rows(max_joined_block_rows_) {}
    std::string getName() const override { return "GpuJoiningTransform"; }
    /// readExecute: keeps resubmitting the not_processed tail; outputs are collected in order
    std::vector<Chunk> transformAll(Chunk chunk)
    {
        std::vector<Chunk> out;
        std::shared_ptr<Chunk> rest;
        for (;;)
        {
            join->joinBlock(chunk, left_key_position, rest, max_joined_block_rows);
            if (chunk.num_rows)
                out.push_back(std::move(chunk));
            if (!rest)
                break;
            chunk = std::move(*rest);
        }
        return out;
    }

protected:
    void transform(Chunk & chunk) override
    {
        std::shared_ptr<Chunk> rest;
        join->joinBlock(chunk, left_key_position, rest, 0);
    }

private:
    std::shared_ptr<GpuHashJoin> join;
    size_t left_key_position;
    uint64_t max_joined_block_rows;
};

/// One rank of the node's exchange: one process per GPU, RCCL over xGMI behind the C ABI (chgpu_comm_*).  The 128-byte id made by
/// rank 0 (uniqueId) reaches the other ranks through the host pipeline's own control plane.
class Communicator
{
public:
    using UniqueId = std::array<uint8_t, CHGPU_UNIQUE_ID_BYTES>;
    static UniqueId uniqueId()
    {
        UniqueId id;
        check(chgpu_comm_unique_id(id.data()));
        return id;
    }
    Communicator(ContextPtr ctx_, int rank, int world, const UniqueId & id) : ctx(std::move(ctx_)) { check(chgpu_comm_init(ctx->get(), rank, world, id.data(), &h)); }
    ~Communicator() { chgpu_comm_destroy(h); }
    Communicator(const Communicator &) = delete;
    int rank() const { return chgpu_comm_rank(h); }
    int world() const { return chgpu_comm_world(h); }
    const ContextPtr & context() const { return ctx; }
    std::vector<uint64_t> allToAllCounts(const std::vector<uint64_t> & send_counts) const
    {
        std::vector<uint64_t> recv(send_counts.size(), 0);
        check(chgpu_all_to_all_counts(h, send_counts.data(), recv.data()));
        return recv;
    }
    ColumnPtr allToAll(const ColumnVector & send, const std::vector<uint64_t> & send_counts, const std::vector<uint64_t> & recv_counts) const;
    /// every column of one partitioned Block in ONE exchange (one count exchange + one grouped send / recv): -> received columns, rows per peer
    Columns allToAllMulti(const Columns & send, const std::vector<uint64_t> & send_counts, std::vector<uint64_t> & recv_counts) const;
    void allReduce(std::vector<uint64_t> & values) const { check(chgpu_all_reduce_u64_host(h, values.data(), static_cast<uint32_t>(values.size()))); }
    void barrier() const { check(chgpu_comm_barrier(h)); }

private:
    ContextPtr ctx;
    chgpu_comm * h = nullptr;
};
using CommunicatorPtr = std::shared_ptr<Communicator>;

inline ColumnPtr Communicator::allToAll(const ColumnVector & send, const std::vector<uint64_t> & send_counts, const std::vector<uint64_t> & recv_counts) const
{
    chgpu_col * out = nullptr;
    check(chgpu_all_to_all(h, send.handle(), send_counts.data(), recv_counts.data(), &out));
    return std::make_shared<ColumnVector>(ctx, out);
}

inline Columns Communicator::allToAllMulti(const Columns & send, const std::vector<uint64_t> & send_counts, std::vector<uint64_t> & recv_counts) const
{
    std::vector<const chgpu_col *> in;
    for (auto & c : send)
        in.push_back(c->handle());
    std::vector<chgpu_col *> out(in.size(), nullptr);
    recv_counts.assign(send_counts.size(), 0);
    check(chgpu_all_to_all_multi(h, static_cast<uint32_t>(in.size()), in.data(), send_counts.data(), recv_counts.data(), out.data()));
    Columns res;
    for (auto * o : out)
        res.push_back(std::make_shared<ColumnVector>(ctx, o));
    return res;
}

/// ConcurrentHashJoin::dispatchBlock (ConcurrentHashJoin.cpp:538-565: hashToSelector :426-440 + scatterBlocksWithSelector :518-536) with
/// the slots living on different GPUs: the rows of `block` are split by shard = bucket(key) & (world - 1) and every shard travels
/// to its owner in ONE exchange for the whole Block; the result is the chunk of rows THIS rank owns (its own shard + what the peers sent).
inline Chunk dispatchBlock(const Communicator & comm, const Chunk & block, size_t key_position)
{
    const uint32_t world = static_cast<uint32_t>(comm.world());
    if (world == 1)
        return block;
    const auto & ctx = comm.context();
    std::vector<const chgpu_col *> in;
    for (auto & c : block.columns)
        in.push_back(c->handle());
    std::vector<chgpu_col *> parts(in.size(), nullptr);
    std::vector<uint64_t> counts(world, 0);
    check(chgpu_partition_by_hash(ctx->get(), block.columns.at(key_position)->handle(), world, static_cast<uint32_t>(in.size()), in.data(), parts.data(), counts.data()));
    Columns sends;
    for (auto * p : parts)
        sends.push_back(std::make_shared<ColumnVector>(ctx, p));
    std::vector<uint64_t> recv_counts;
    Chunk mine;
    mine.columns = comm.allToAllMulti(sends, counts, recv_counts);
    for (auto r : recv_counts)
        mine.num_rows += r;
    return mine;
}

/// GROUP BY over the GPUs of a node.  Every rank aggregates its own rows (one AggregatedDataVariants per stream, as
/// AggregatingTransform does per thread); the partial states are then handed to their owners by two-level bucket -- the parallel merge of
/// AggregatingTransform.cpp:120-136 / Aggregator::mergeBucketImpl (Aggregator.cpp:2691-2725) with the buckets owned by ranks instead of
/// claimed by threads -- with ONE all-to-all of (key, state words); owners merge and finalise their shard.
class GpuShardedAggregator
{
public:
    GpuShardedAggregator(ContextPtr ctx_, CommunicatorPtr comm_, int key_type_, std::vector<AggregateDescription> aggregates_, uint64_t size_hint_ = 0)
        : ctx(std::move(ctx_)), comm(std::move(comm_)), key_type(key_type_), aggregates(std::move(aggregates_)), size_hint(size_hint_)
    {
        for (auto & a : aggregates)
        {
            kinds.push_back(a.kind);
            types.push_back(a.argument_type);
            n_words += (a.kind == CHGPU_AGG_AVG || a.kind == CHGPU_AGG_ANY) ? 2 : 1; // avg: numerator + denominator; any: claim + value
        }
        check(chgpu_agg_create(ctx->get(), key_type, static_cast<uint32_t>(aggregates.size()), kinds.data(), types.data(), size_hint, &local));
    }
    ~GpuShardedAggregator()
    {
        chgpu_agg_free(local);
        chgpu_agg_free(owner);
    }
    GpuShardedAggregator(const GpuShardedAggregator &) = delete;

    bool executeOnBlock(const Columns & columns, size_t row_begin, size_t row_end, size_t key_position)
    {
        std::vector<const chgpu_col *> args;
        for (auto & a : aggregates)
            args.push_back(a.kind == CHGPU_AGG_COUNT ? nullptr : columns.at(a.argument)->handle());
        check(chgpu_agg_add_block(local, columns.at(key_position)->handle(), args.data(), row_begin, row_end));
        return true;
    }

    /// mergeAndConvertToBlocks: the final block of the groups THIS rank owns (the result stays sharded, like the reference's buckets)
    Chunk convertToBlock()
    {
        chgpu_agg * final_agg = local;
        if (comm->world() > 1)
        {
            chgpu_col * keys = nullptr;
            std::vector<chgpu_col *> words(n_words, nullptr);
            uint64_t groups = 0;
            check(chgpu_agg_export_states(local, &keys, words.data(), &groups)); // convertToBlockImplNotFinal
            Chunk states;
            states.num_rows = groups;
            states.columns.push_back(std::make_shared<ColumnVector>(ctx, keys));
            for (auto * w : words)
                states.columns.push_back(std::make_shared<ColumnVector>(ctx, w));
            Chunk mine = dispatchBlock(*comm, states, 0);
            if (!owner)
                check(chgpu_agg_create(ctx->get(), key_type, static_cast<uint32_t>(aggregates.size()), kinds.data(), types.data(), mine.num_rows, &owner));
            std::vector<const chgpu_col *> sc;
            for (size_t w = 0; w < n_words; ++w)
                sc.push_back(mine.columns[1 + w]->handle());
            check(chgpu_agg_merge_states(owner, mine.columns[0]->handle(), sc.data(), mine.num_rows)); // mergeBucketImpl on the owner
            final_agg = owner;
        }
        chgpu_col * keys = nullptr;
        std::vector<chgpu_col *> res(aggregates.size(), nullptr);
        uint64_t groups = 0;
        check(chgpu_agg_finalize(final_agg, &keys, res.data(), &groups));
        Chunk out;
        out.num_rows = groups;
        out.columns.push_back(std::make_shared<ColumnVector>(ctx, keys));
        for (auto * r : res)
            out.columns.push_back(std::make_shared<ColumnVector>(ctx, r));
        return out;
    }

private:
    ContextPtr ctx;
    CommunicatorPtr comm;
    int key_type;
    std::vector<AggregateDescription> aggregates;
    uint64_t size_hint;
    std::vector<int> kinds, types;
    size_t n_words = 0;
    chgpu_agg * local = nullptr;
    chgpu_agg * owner = nullptr;
};

/// A block of (partially) aggregated data with its BlockInfo (src/Core/BlockInfo.h:21-29): columns[0] = keys, then the state words
/// (not final: sum / count one word, avg numerator and denominator) or the result columns (final).
struct AggregatedChunk
{
    Chunk chunk;
    int32_t bucket_num = -1;
    bool is_overflows = false;
};

/// MergingAggregatedMemoryEfficientTransform (src/Processors/Transforms/MergingAggregatedMemoryEfficientTransform.h:17-57): the
/// initiator's side of a distributed GROUP BY.  Every source hands over either one unsplit block (bucket_num -1) or split two-level
/// blocks in increasing bucket_num, optionally one block of overflows; the blocks of one bucket from all sources are grouped
/// (GroupingAggregatedTransform, .cpp:33-330), merged (MergingAggregatedBucketTransform -> Aggregator::mergeBlocks) and emitted in
/// increasing bucket_num (SortingAggregatedTransform).  The contract is the reference's, the work is done in bulk: the blocks of every
/// bucket that has become complete go into ONE device aggregator (buckets are disjoint key sets) and the result is cut into its
/// buckets by the bucket hash (chgpu_partition_by_hash = (crc32c(key) >> 24) & 0xFF, TwoLevelHashTable.h:53); an unsplit block that
/// meets split ones is split the same way (Aggregator::convertBlockToTwoLevel, Aggregator.cpp:3300-3409).
class GpuMergingAggregatedTransform
{
public:
    static constexpr int32_t NUM_BUCKETS = 256;

    GpuMergingAggregatedTransform(ContextPtr ctx_, int key_type_, std::vector<AggregateDescription> aggregates_, size_t num_inputs, bool final_ = true)
        : ctx(std::move(ctx_)), key_type(key_type_), aggregates(std::move(aggregates_)), final_result(final_), last_bucket_number(num_inputs, -1), finished(num_inputs, false)
    {
        for (auto & a : aggregates)
        {
            kinds.push_back(a.kind);
            types.push_back(a.argument_type);
            n_words += (a.kind == CHGPU_AGG_AVG || a.kind == CHGPU_AGG_ANY) ? 2 : 1; // avg: numerator + denominator; any: claim + value
        }
    }

    /// GroupingAggregatedTransform::addChunk (.cpp:258-291)
    void addChunk(size_t input, AggregatedChunk c)
    {
        if (finished.at(input))
            throw Exception(CHGPU_ERR_LOGICAL, "GpuMergingAggregatedTransform: input sent a block after it had finished");
        if (c.chunk.columns.size() != 1 + n_words)
            throw Exception(CHGPU_ERR_BAD_ARGUMENTS, "GpuMergingAggregatedTransform: a block needs the key column and one column per state word");
        if (c.chunk.num_rows == 0)
            return;
        if (c.is_overflows)
            overflow_chunks.push_back(std::move(c.chunk));
        else if (c.bucket_num < 0)
            single_level_chunks.push_back(std::move(c.chunk));
        else
        {
            if (c.bucket_num >= NUM_BUCKETS || c.bucket_num < last_bucket_number[input] || c.bucket_num < next_bucket_to_push)
                throw Exception(CHGPU_ERR_LOGICAL, "GpuMergingAggregatedTransform: split blocks must arrive in the order of bucket_num");
            chunks_map[c.bucket_num].push_back(std::move(c.chunk));
            has_two_level = true;
            last_bucket_number[input] = c.bucket_num;
        }
    }
    void finishInput(size_t input) { finished.at(input) = true; }

    /// the merged blocks that are complete now, in increasing bucket_num; once every input has finished: the rest, then the unsplit
    /// result (only if no source was two-level), then the overflows (tryPushTwoLevelData / SingleLevel / Overflow, .cpp:33-92)
    std::vector<AggregatedChunk> pull()
    {
        std::vector<AggregatedChunk> out;
        if (done)
            return out;
        const bool all_finished = std::all_of(finished.begin(), finished.end(), [](bool f) { return f; });
        if (has_two_level && !single_level_chunks.empty())
        {
            // work() (.cpp:293-318): unsplit blocks become split ones
            for (auto & c : single_level_chunks)
                for (auto & part : splitByBucket(c))
                {
                    if (part.bucket_num < next_bucket_to_push)
                        throw Exception(CHGPU_ERR_LOGICAL, "GpuMergingAggregatedTransform: an unsplit block holds a bucket that was already pushed");
                    chunks_map[part.bucket_num].push_back(std::move(part.chunk));
                }
            single_level_chunks.clear();
        }
        if (has_two_level)
        {
            // a bucket is complete when no unfinished source can still send a block of it; sources may cut a bucket into several blocks
            // (expect_several_chunks_for_single_bucket_per_source, .cpp:117-123), so the bucket a source is AT is not complete yet
            int32_t current = NUM_BUCKETS;
            if (!all_finished)
                for (size_t i = 0; i < finished.size(); ++i)
                    if (!finished[i])
                        current = std::min(current, last_bucket_number[i]);
            std::vector<Chunk> ready;
            std::vector<bool> is_ready(NUM_BUCKETS, false);
            for (auto it = chunks_map.begin(); it != chunks_map.end() && it->first < current;)
            {
                is_ready[it->first] = true;
                for (auto & c : it->second)
                    ready.push_back(std::move(c));
                it = chunks_map.erase(it);
            }
            if (!ready.empty())
                for (auto & part : splitByBucket(merge(ready)))
                {
                    if (!is_ready[part.bucket_num])
                        throw Exception(CHGPU_ERR_LOGICAL, "GpuMergingAggregatedTransform: a block holds keys of a bucket it was not declared as");
                    out.push_back(std::move(part));
                }
            next_bucket_to_push = std::max(next_bucket_to_push, std::min(current, NUM_BUCKETS));
        }
        if (all_finished)
        {
            if (!has_two_level && !single_level_chunks.empty())
            {
                out.push_back(AggregatedChunk{merge(single_level_chunks), -1, false});
                single_level_chunks.clear();
            }
            if (!overflow_chunks.empty())
            {
                out.push_back(AggregatedChunk{merge(overflow_chunks), -1, true});
                overflow_chunks.clear();
            }
            done = true;
        }
        return out;
    }

private:
    /// MergingAggregatedBucketTransform::transform -> Aggregator::mergeBlocks(blocks, final)
    Chunk merge(const std::vector<Chunk> & chunks)
    {
        uint64_t hint = 0;
        for (auto & c : chunks)
            hint += c.num_rows;
        chgpu_agg * agg = nullptr;
        check(chgpu_agg_create(ctx->get(), key_type, static_cast<uint32_t>(aggregates.size()), kinds.data(), types.data(), hint, &agg));
        struct Free { void operator()(chgpu_agg * a) const { chgpu_agg_free(a); } };
        std::unique_ptr<chgpu_agg, Free> guard(agg);
        for (auto & c : chunks)
        {
            std::vector<const chgpu_col *> sc;
            for (size_t w = 0; w < n_words; ++w)
                sc.push_back(c.columns[1 + w]->handle());
            check(chgpu_agg_merge_states(agg, c.columns[0]->handle(), sc.data(), c.num_rows));
        }
        chgpu_col * keys = nullptr;
        std::vector<chgpu_col *> cols(final_result ? aggregates.size() : n_words, nullptr);
        uint64_t groups = 0;
        check(final_result ? chgpu_agg_finalize(agg, &keys, cols.data(), &groups) : chgpu_agg_export_states(agg, &keys, cols.data(), &groups));
        Chunk out;
        out.num_rows = groups;
        out.columns.push_back(std::make_shared<ColumnVector>(ctx, keys));
        for (auto * c : cols)
            out.columns.push_back(std::make_shared<ColumnVector>(ctx, c));
        return out;
    }

    /// the rows of `c` cut into their two-level buckets (empty buckets are left out), bucket_num ascending
    std::vector<AggregatedChunk> splitByBucket(const Chunk & c)
    {
        std::vector<const chgpu_col *> ins;
        for (auto & col : c.columns)
            ins.push_back(col->handle());
        std::vector<chgpu_col *> outs(ins.size(), nullptr);
        uint64_t counts[NUM_BUCKETS] = {};
        check(chgpu_partition_by_hash(ctx->get(), ins[0], NUM_BUCKETS, static_cast<uint32_t>(ins.size()), ins.data(), outs.data(), counts));
        Columns whole;
        for (auto * o : outs)
            whole.push_back(std::make_shared<ColumnVector>(ctx, o));
        std::vector<AggregatedChunk> parts;
        uint64_t begin = 0;
        for (int32_t b = 0; b < NUM_BUCKETS; ++b)
        {
            if (counts[b])
            {
                AggregatedChunk p;
                p.bucket_num = b;
                p.chunk.num_rows = counts[b];
                for (auto & col : whole)
                    p.chunk.columns.push_back(col->cut(begin, counts[b], col));
                parts.push_back(std::move(p));
            }
            begin += counts[b];
        }
        return parts;
    }

    ContextPtr ctx;
    int key_type;
    std::vector<AggregateDescription> aggregates;
    bool final_result;
    std::vector<int> kinds, types;
    size_t n_words = 0;
    std::vector<int32_t> last_bucket_number; // GroupingAggregatedTransform::last_bucket_number
    std::vector<bool> finished;
    std::map<int32_t, std::vector<Chunk>> chunks_map;
    std::vector<Chunk> single_level_chunks, overflow_chunks;
    bool has_two_level = false, done = false;
    int32_t next_bucket_to_push = 0;
};

/// ConcurrentHashJoin (`parallel_hash`, src/Interpreters/ConcurrentHashJoin.h:25-39, .cpp) with one slot per GPU: build rows are
/// dispatched to the slot that owns their key (addBlockToJoin -> dispatchBlock, ConcurrentHashJoin.cpp:214-263), every slot builds
/// its HashJoin; probe rows are dispatched by the same rule and joined where they land (joinBlock, :265-310).  The joined rows stay on
/// the owner rank -- the next operator (a sharded GROUP BY) consumes them there.
class GpuConcurrentHashJoin
{
public:
    GpuConcurrentHashJoin(ContextPtr ctx_, CommunicatorPtr comm_, int key_type, int kind, int strictness)
        : ctx(ctx_), comm(std::move(comm_)), local(std::make_shared<GpuHashJoin>(ctx_, key_type, kind, strictness)) {}

    bool addBlockToJoin(const Chunk & block, size_t key_position) { return local->addBlockToJoin(dispatchBlock(*comm, block, key_position), key_position); }
    void onBuildPhaseFinish() { local->onBuildPhaseFinish(); }
    size_t getTotalRowCount() const { return local->getTotalRowCount(); } // rows of THIS slot
    /// joinBlock: `block` (this rank's left rows) is replaced by the joined rows this rank OWNS
    void joinBlock(Chunk & block, size_t key_position, std::shared_ptr<Chunk> & not_processed, uint64_t max_joined_block_rows = 0)
    {
        Chunk mine = dispatchBlock(*comm, block, key_position);
        local->joinBlock(mine, key_position, not_processed, max_joined_block_rows);
        block = std::move(mine);
    }
    /// the join with a keyless aggregation fused behind it: SELECT count(), sum(right column) over the WHOLE join (all ranks):
    /// dispatch the left keys, chgpu_join_probe_agg on the owner, one 16-byte all-reduce (mergeWithoutKeyDataImpl).
    /// Integer payloads only across ranks (a Float64 sum would depend on the rank order of the reduction).
    std::pair<uint64_t, uint64_t> joinCountSum(const Chunk & block, size_t key_position, size_t right_payload_position)
    {
        Chunk keys_only;
        keys_only.columns.push_back(block.columns.at(key_position));
        keys_only.num_rows = block.num_rows;
        Chunk mine = dispatchBlock(*comm, keys_only, 0);
        auto r = local->probeCountSum(*mine.columns[0], right_payload_position);
        std::vector<uint64_t> v{r.first, r.second};
        comm->allReduce(v);
        return {v[0], v[1]};
    }

private:
    ContextPtr ctx;
    CommunicatorPtr comm;
    std::shared_ptr<GpuHashJoin> local;
};

/// SortColumnDescription (src/Core/SortDescription.h:26-60): column position, direction (+1 ASC / -1 DESC), nulls_direction (NaN counts
/// as greater than every number when +1: ASC NULLS LAST, DESC NULLS FIRST).
struct SortColumnDescription
{
    size_t column_number;
    int direction = 1;
    int nulls_direction = 1;
};
using SortDescription = std::vector<SortColumnDescription>;

/// sortBlock (src/Interpreters/sortBlock.cpp:240-330): the permutation of the whole description (stable radix sorts from the least
/// significant column to the most; one column + limit takes the sampled-threshold path), applied to every column (IColumn::permute).
inline void sortBlock(Chunk & block, const SortDescription & description, uint64_t limit = 0)
{
    if (description.empty() || block.num_rows == 0)
        return;
    ContextPtr ctx = block.columns.at(description[0].column_number)->context();
    chgpu_col * perm = nullptr;
    if (description.size() == 1 && limit)
    {
        const auto & d = description[0];
        check(chgpu_sort_permutation_limit(ctx->get(), block.columns.at(d.column_number)->handle(), d.direction < 0, d.nulls_direction, limit, &perm));
    }
    else
    {
        for (size_t k = description.size(); k-- > 0;)
        {
            const auto & d = description[k];
            chgpu_col * next = nullptr;
            const int rc = chgpu_sort_permutation(ctx->get(), block.columns.at(d.column_number)->handle(), perm, d.direction < 0, d.nulls_direction, &next);
            if (perm)
                chgpu_col_free(perm);
            check(rc);
            perm = next;
        }
    }
    ColumnVector permutation(ctx, perm);
    const uint64_t take = limit && limit < block.num_rows ? limit : 0;
    for (auto & col : block.columns)
    {
        chgpu_col * out = nullptr;
        check(chgpu_index(ctx->get(), col->handle(), permutation.handle(), take, 0, &out));
        col = std::make_shared<ColumnVector>(ctx, out);
    }
    if (take)
        block.num_rows = take;
}

/// CompressedReadBuffer + SerializationNumber::deserializeBinaryBulk for one numeric column file (MergeTree `<column>.bin`): the
/// host walks the frame headers (CompressedReadBufferBase.cpp:175-222), the compressed bytes cross PCIe once, the frames are decoded
/// in HBM.  LZ4, NONE and CODEC(Delta, LZ4); any other codec throws NOT_IMPLEMENTED (the caller decompresses on the CPU as before).
inline ColumnPtr readCompressedColumn(const ContextPtr & ctx, const unsigned char * file, size_t size, int type, bool verify_checksums = true)
{
    // frame walk + CityHash128 verification + size caps on the host (CompressedReadBufferBase.cpp:49-127,163-172), decode on the device
    chgpu_col * out = nullptr;
    check(chgpu_read_compressed_column(ctx->get(), file, size, type, verify_checksums ? 1 : 0, &out));
    return std::make_shared<ColumnVector>(ctx, out);
}

} // namespace chgpu
