// pipeline_demo.cpp — drives the C++ shim (chgpu_shim.hpp) the way the reference's pipeline drives its processors:
// host Blocks of DEFAULT_BLOCK_SIZE rows -> HBM stripes -> GpuFilterTransform -> GpuAggregatingTransform, a GROUP BY and
// a hash join, each checked against a straightforward host loop.  Exit code 0 == every check passed.
// build: g++ -std=c++17 -O2 pipeline_demo.cpp -L.. -lchgpu -Wl,-rpath,'$ORIGIN/..' -o pipeline_demo
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <algorithm>
#include <climits>
#include <cstdint>
#include <map>
#include <thread>
#include <unordered_map>

#include "chgpu_shim.hpp"

using namespace chgpu;

static constexpr size_t DEFAULT_BLOCK_SIZE = 65409; // src/Core/Defines.h:31-32

#define REQUIRE(cond)                                                        \
    do                                                                       \
    {                                                                        \
        if (!(cond))                                                         \
        {                                                                    \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

// pipeline_demo --bench-host-blocks <rows> [streams] [stripe_rows]: `SELECT sum(a), count() WHERE a < 214748365` over Int64 rows that START
// IN HOST MEMORY as Blocks of 65 409 rows -- every pipeline stream (thread, own Context) glues its Blocks into pinned stripes, uploads them
// asynchronously and runs the fused filter + sum on each stripe.  Prints one JSON line: the PCIe-inclusive rows/s.
static int bench_host_blocks(size_t n, int streams, size_t stripe_rows)
{
    std::vector<int64_t> a(n);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i)
    {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        a[i] = static_cast<int64_t>(x & 0x7FFFFFFF);
    }
    const int64_t thr = 214748365;
    uint64_t want_sum = 0, want_cnt = 0;
    for (size_t i = 0; i < n; ++i)
        if (a[i] < thr)
        {
            want_sum += static_cast<uint64_t>(a[i]);
            ++want_cnt;
        }
    // per-stream state made once, outside the timed region: a Context (its stream and copy stream), the two pinned stripe buffers
    std::vector<ContextPtr> ctxs;
    std::vector<std::unique_ptr<StripeBuilder<int64_t>>> builders;
    for (int t = 0; t < streams; ++t)
    {
        ctxs.push_back(std::make_shared<Context>(0));
        builders.push_back(std::make_unique<StripeBuilder<int64_t>>(ctxs.back(), stripe_rows));
    }
    double best = 0;
    for (int rep = 0; rep < 4; ++rep)
    {
        std::vector<uint64_t> sums(streams, 0), cnts(streams, 0);
        std::atomic<int> bad{0};
        std::vector<std::thread> pool;
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < streams; ++t)
            pool.emplace_back([&, t] {
                try
                {
                    StripeBuilder<int64_t> & sb = *builders[t];
                    GpuFilterSumTransform fs(0, FunctionComparisonConst(CHGPU_LT, thr), 0);
                    const size_t n_blocks = (n + DEFAULT_BLOCK_SIZE - 1) / DEFAULT_BLOCK_SIZE;
                    auto run_stripe = [&] {
                        Chunk c;
                        c.num_rows = sb.rows();
                        c.columns = {sb.flush()};
                        fs.consume(c);
                    };
                    for (size_t bi = n_blocks * t / streams; bi < n_blocks * (t + 1) / streams; ++bi)
                    {
                        const size_t b = bi * DEFAULT_BLOCK_SIZE, rows = std::min(DEFAULT_BLOCK_SIZE, n - b);
                        size_t done = 0;
                        while (done < rows)
                        {
                            done += sb.appendBlock(a.data() + b + done, rows - done);
                            if (sb.room() == 0)
                                run_stripe();
                        }
                    }
                    if (sb.rows())
                        run_stripe();
                    sums[t] = fs.sum;
                    cnts[t] = fs.count;
                }
                catch (...)
                {
                    ++bad;
                }
            });
        for (auto & th : pool)
            th.join();
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        uint64_t s = 0, c = 0;
        for (int t = 0; t < streams; ++t)
        {
            s += sums[t];
            c += cnts[t];
        }
        if (bad || s != want_sum || c != want_cnt)
        {
            std::fprintf(stderr, "bench-host-blocks: wrong result\n");
            return 1;
        }
        best = std::max(best, double(n) / secs);
    }
    std::printf("{\"rows\": %zu, \"streams\": %d, \"stripe_rows\": %zu, \"block_rows\": %zu, \"rows_per_s_pcie_inclusive\": %.6g, \"host_GBps\": %.4g}\n", n, streams,
                stripe_rows, size_t(DEFAULT_BLOCK_SIZE), best, best * 8 / 1e9);
    return 0;
}

int main(int argc, char ** argv)
{
    if (argc > 2 && std::string(argv[1]) == "--bench-host-blocks")
    {
        try
        {
            return bench_host_blocks(std::strtoull(argv[2], nullptr, 10), argc > 3 ? std::atoi(argv[3]) : 4, argc > 4 ? std::strtoull(argv[4], nullptr, 10) : (size_t(8) << 20));
        }
        catch (const Exception & e)
        {
            std::fprintf(stderr, "chgpu::Exception %d: %s\n", e.code(), e.what());
            return 1;
        }
    }
    const size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 2000003;
    try
    {
        auto ctx = std::make_shared<Context>(0);

        // ---- synthetic table in host Blocks -------------------------------------------------------------
        std::vector<int64_t> a(n);
        std::vector<uint32_t> k(n);
        uint64_t x = 88172645463325252ull;
        for (size_t i = 0; i < n; ++i)
        {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            a[i] = static_cast<int64_t>(x & 0x7FFFFFFF);
            k[i] = static_cast<uint32_t>((x >> 33) % 1000);
        }
        StripeBuilder<int64_t> sa(ctx, n);
        StripeBuilder<uint32_t> sk(ctx, n);
        for (size_t b = 0; b < n; b += DEFAULT_BLOCK_SIZE)
        {
            size_t rows = std::min(DEFAULT_BLOCK_SIZE, n - b);
            REQUIRE(sa.appendBlock(a.data() + b, rows) == rows); // many Blocks -> one HBM stripe
            REQUIRE(sk.appendBlock(k.data() + b, rows) == rows);
        }
        Chunk stripe;
        stripe.columns = {sa.flush(), sk.flush()};
        stripe.num_rows = n;

        // ---- SELECT sum(a), count() WHERE a < 214748365 ---------------------------------------------------
        const int64_t thr = 214748365;
        GpuFilterTransform filter(0, FunctionComparisonConst(CHGPU_LT, thr));
        auto agg0 = std::make_shared<GpuAggregator>(ctx, -1, std::vector<AggregateDescription>{{CHGPU_AGG_SUM, CHGPU_I64, 0}, {CHGPU_AGG_COUNT, CHGPU_U64, 0}});
        GpuAggregatingTransform aggregating(agg0, std::nullopt);
        filter.setInput(stripe);
        filter.work();
        if (filter.hasOutput()) // a chunk with no passing row is dropped (ISimpleTransform.cpp:101-107)
            aggregating.consume(filter.pullOutput());
        Chunk r0 = aggregating.generate();
        uint64_t want_sum = 0, want_cnt = 0;
        for (size_t i = 0; i < n; ++i)
            if (a[i] < thr)
            {
                want_sum += static_cast<uint64_t>(a[i]);
                ++want_cnt;
            }
        REQUIRE(r0.num_rows == 1);
        REQUIRE(static_cast<uint64_t>(r0.columns[0]->getData<int64_t>()[0]) == want_sum);
        REQUIRE(r0.columns[1]->getData<uint64_t>()[0] == want_cnt);
        REQUIRE(filter.passed_rows == want_cnt);

        // the same plan with a second WHERE conjunct as its own transform, run the executor's way: three stripes through
        // FilterTransform -> FilterTransform -> the aggregating sink, every step decided by prepare() (executeChain)
        {
            GpuFilterTransform f1(0, FunctionComparisonConst(CHGPU_LT, thr)), f2(0, FunctionComparisonConst(CHGPU_GE, int64_t(1000)));
            auto aggx = std::make_shared<GpuAggregator>(ctx, -1, std::vector<AggregateDescription>{{CHGPU_AGG_SUM, CHGPU_I64, 0}, {CHGPU_AGG_COUNT, CHGPU_U64, 0}});
            GpuAggregatingTransform sink_agg(aggx, std::nullopt);
            REQUIRE(f1.prepare() == IProcessor::Status::NeedData);
            int fed = 0;
            executeChain([&](Chunk & c) { if (fed == 3) return false; c = stripe; ++fed; return true; }, {&f1, &f2},
                         [&](Chunk c) { sink_agg.consume(std::move(c)); });
            REQUIRE(f1.prepare() == IProcessor::Status::Finished && f2.prepare() == IProcessor::Status::Finished);
            Chunk rx = sink_agg.generate();
            uint64_t s2 = 0, c2 = 0;
            for (size_t i = 0; i < n; ++i)
                if (a[i] < thr && a[i] >= 1000)
                {
                    s2 += static_cast<uint64_t>(a[i]);
                    ++c2;
                }
            REQUIRE(static_cast<uint64_t>(rx.columns[0]->getData<int64_t>()[0]) == 3 * s2 && rx.columns[1]->getData<uint64_t>()[0] == 3 * c2);
        }

        // a chunk in which nothing passes is dropped, not forwarded empty (FilterTransform.cpp:221-226)
        GpuFilterTransform none(0, FunctionComparisonConst(CHGPU_LT, int64_t(0)));
        none.setInput(stripe);
        none.work();
        REQUIRE(!none.hasOutput());

        // ---- SELECT k, sum(a), count() GROUP BY k ---------------------------------------------------------
        auto agg1 = std::make_shared<GpuAggregator>(ctx, CHGPU_U32, std::vector<AggregateDescription>{{CHGPU_AGG_SUM, CHGPU_I64, 0}, {CHGPU_AGG_COUNT, CHGPU_U64, 0}});
        GpuAggregatingTransform gb(agg1, 1);
        gb.consume(stripe);
        Chunk r1 = gb.generate();
        std::map<uint32_t, std::pair<uint64_t, uint64_t>> want;
        for (size_t i = 0; i < n; ++i)
        {
            want[k[i]].first += static_cast<uint64_t>(a[i]);
            want[k[i]].second += 1;
        }
        REQUIRE(r1.num_rows == want.size());
        auto gk = r1.columns[0]->getData<uint32_t>();
        auto gs = r1.columns[1]->getData<int64_t>();
        auto gc = r1.columns[2]->getData<uint64_t>();
        for (size_t i = 0; i < gk.size(); ++i)
        {
            auto it = want.find(gk[i]);
            REQUIRE(it != want.end());
            REQUIRE(static_cast<uint64_t>(gs[i]) == it->second.first && gc[i] == it->second.second);
        }

        // ---- SELECT pk, bv FROM probe ALL INNER JOIN build ON pk = bk -----------------------------------------
        const size_t nb = 50000;
        std::vector<uint64_t> bk(nb);
        std::vector<int64_t> bv(nb);
        for (size_t i = 0; i < nb; ++i)
        {
            bk[i] = (i / 2) * 7 + 1; // every key twice
            bv[i] = static_cast<int64_t>(i) * 3 - 11;
        }
        std::vector<uint64_t> pk(200000);
        for (size_t i = 0; i < pk.size(); ++i)
            pk[i] = (i * 2654435761ull) % (nb * 7);
        auto join = std::make_shared<GpuHashJoin>(ctx, CHGPU_U64, CHGPU_JOIN_INNER, CHGPU_STRICT_ALL);
        for (size_t b = 0; b < nb; b += 16384) // FillingRightJoinSideTransform: the build side arrives Block by Block
        {
            const size_t rows = std::min<size_t>(16384, nb - b);
            Chunk right;
            right.columns = {ColumnVector::fromHost<uint64_t>(ctx, bk.data() + b, rows), ColumnVector::fromHost<int64_t>(ctx, bv.data() + b, rows)};
            right.num_rows = rows;
            join->addBlockToJoin(right, 0);
        }
        join->onBuildPhaseFinish();
        REQUIRE(join->getTotalRowCount() == nb);
        Chunk left;
        left.columns = {ColumnVector::fromHost<uint64_t>(ctx, pk.data(), pk.size())};
        left.num_rows = pk.size();
        GpuJoiningTransform joining(join, 0, /*max_joined_block_rows*/ 4096);
        auto outs = joining.transformAll(left); // resubmits the not_processed tail like JoiningTransform::readExecute
        std::unordered_multimap<uint64_t, int64_t> build;
        for (size_t i = 0; i < nb; ++i)
            build.emplace(bk[i], bv[i]);
        uint64_t want_rows = 0;
        int64_t want_chk = 0;
        for (auto key : pk)
        {
            auto range = build.equal_range(key);
            for (auto it = range.first; it != range.second; ++it)
            {
                ++want_rows;
                want_chk += static_cast<int64_t>(key % 1000003) * 31 + it->second;
            }
        }
        uint64_t got_rows = 0;
        int64_t got_chk = 0;
        REQUIRE(outs.size() > 1); // the row limit forced several output blocks
        for (auto & c : outs)
        {
            auto opk = c.columns[0]->getData<uint64_t>();
            auto obk = c.columns[1]->getData<uint64_t>();
            auto obv = c.columns[2]->getData<int64_t>();
            REQUIRE(opk.size() == c.num_rows && obv.size() == c.num_rows);
            for (size_t i = 0; i < opk.size(); ++i)
            {
                REQUIRE(opk[i] == obk[i]);
                got_chk += static_cast<int64_t>(opk[i] % 1000003) * 31 + obv[i];
            }
            got_rows += c.num_rows;
        }
        REQUIRE(got_rows == want_rows && got_chk == want_chk);

        // ---- WHERE a < thr AND k % ... as an expression DAG: ExpressionActions compiled at run time into one kernel --------
        {
            ActionsDAG dag;
            auto ia = dag.addInput(0, CHGPU_I64), ik = dag.addInput(1, CHGPU_U32);
            auto p = dag.addFunction("and", {dag.addFunction("less", {ia, dag.addColumn<int64_t>(thr)}),
                                             dag.addFunction("greaterOrEquals", {ik, dag.addColumn<uint16_t>(500)})});
            auto v = dag.addFunction("minus", {dag.addFunction("multiply", {ia, ik}), dag.addColumn<uint8_t>(7)});
            auto actions = std::make_shared<ExpressionActions>(dag);
            REQUIRE(actions->resultType(p) == CHGPU_U8 && actions->resultType(v) == CHGPU_I64);
            uint64_t ws = 0, wc = 0;
            for (size_t i = 0; i < n; ++i)
                if (a[i] < thr && k[i] >= 500)
                {
                    ws += static_cast<uint64_t>(a[i]) * k[i] - 7;
                    ++wc;
                }
            int rt = -1;
            auto fused = actions->filterSum(stripe.columns, static_cast<int>(p), static_cast<int>(v), &rt); // one pass, nothing materialised
            REQUIRE(rt == CHGPU_I64 && fused.first == ws && fused.second == wc);
            auto mm = actions->filterMinMax(stripe.columns, static_cast<int>(p), v);
            int64_t wmin = INT64_MAX, wmax = INT64_MIN;
            for (size_t i = 0; i < n; ++i)
                if (a[i] < thr && k[i] >= 500)
                {
                    const int64_t x = static_cast<int64_t>(static_cast<uint64_t>(a[i]) * k[i] - 7);
                    wmin = std::min(wmin, x), wmax = std::max(wmax, x);
                }
            REQUIRE(mm.count == wc && mm.type == CHGPU_I64);
            REQUIRE(wc == 0 || (static_cast<int64_t>(mm.min_bits) == wmin && static_cast<int64_t>(mm.max_bits) == wmax));
            // the same through the processors: ExpressionTransform appends v, FilterTransform on the DAG's filter node, sum
            GpuExpressionTransform project(actions, {v});
            GpuExpressionFilterTransform where(actions, p);
            auto ag = std::make_shared<GpuAggregator>(ctx, -1, std::vector<AggregateDescription>{{CHGPU_AGG_SUM, CHGPU_I64, 2}, {CHGPU_AGG_COUNT, CHGPU_U64, 0}});
            GpuAggregatingTransform at(ag, std::nullopt);
            project.setInput(stripe);
            project.work();
            where.setInput(project.pullOutput());
            where.work();
            if (where.hasOutput())
                at.consume(where.pullOutput());
            Chunk r = at.generate();
            REQUIRE(static_cast<uint64_t>(r.columns[0]->getData<int64_t>()[0]) == ws && r.columns[1]->getData<uint64_t>()[0] == wc);
            REQUIRE(where.passed_rows == wc);
            bool unknown = false;
            try { dag.addFunction("cityHash64", {ia}); } catch (const Exception & e) { unknown = e.isNotImplemented(); }
            REQUIRE(unknown);
        }

        // ---- GROUP BY a LowCardinality(String) key: two Blocks with different dictionaries ------------------------------
        {
            auto d1 = std::make_shared<const std::vector<std::string>>(std::vector<std::string>{"ASIA", "EUROPE", "AFRICA"});
            auto d2 = std::make_shared<const std::vector<std::string>>(std::vector<std::string>{"EUROPE", "AMERICA", "ASIA"});
            const size_t m = 100003;
            std::vector<uint8_t> i1(m), i2(m);
            std::vector<int64_t> v1(m), v2(m);
            std::map<std::string, std::pair<int64_t, uint64_t>> want_lc;
            uint64_t y = 2463534242ull;
            for (size_t i = 0; i < m; ++i)
            {
                y ^= y << 13; y ^= y >> 7; y ^= y << 17;
                i1[i] = static_cast<uint8_t>(y % 3), i2[i] = static_cast<uint8_t>((y >> 20) % 3);
                v1[i] = static_cast<int64_t>((y >> 8) % 1000), v2[i] = -static_cast<int64_t>((y >> 30) % 777);
                auto & w1 = want_lc[(*d1)[i1[i]]];
                w1.first += v1[i], ++w1.second;
                auto & w2 = want_lc[(*d2)[i2[i]]];
                w2.first += v2[i], ++w2.second;
            }
            // a String key column: dictionary-encoded on the device, then the same path
            {
                std::vector<std::string> names = {"FRANCE", "", "PERU", "FRANCE", "PERU", "FRANCE", std::string("a\0b", 3), "PERU"};
                std::vector<uint8_t> chars;
                std::vector<uint64_t> offs;
                for (auto & v : names)
                {
                    chars.insert(chars.end(), v.begin(), v.end());
                    chars.push_back(0);
                    offs.push_back(chars.size());
                }
                ColumnString col{ColumnVector::fromHost<uint64_t>(ctx, offs.data(), offs.size()), ColumnVector::fromHost<uint8_t>(ctx, chars.data(), chars.size())};
                ColumnLowCardinality enc = col.dictionaryEncode([&](uint64_t r) { return names[r]; });
                auto got_ids = enc.indexes->getData<uint32_t>();
                REQUIRE((*enc.dictionary == std::vector<std::string>{"FRANCE", "", "PERU", std::string("a\0b", 3)}));
                REQUIRE((got_ids == std::vector<uint32_t>{0, 1, 2, 0, 2, 0, 3, 2}));
            }
            LowCardinalityDictionary dict(ctx);
            GpuAggregator lc_agg(ctx, CHGPU_U32, {{CHGPU_AGG_SUM, CHGPU_I64, 1}, {CHGPU_AGG_COUNT, CHGPU_U64, 0}});
            Columns b1 = {dict.mapBlock({d1, ColumnVector::fromHost<uint8_t>(ctx, i1.data(), m)}), ColumnVector::fromHost<int64_t>(ctx, v1.data(), m)};
            Columns b2 = {dict.mapBlock({d2, ColumnVector::fromHost<uint8_t>(ctx, i2.data(), m)}), ColumnVector::fromHost<int64_t>(ctx, v2.data(), m)};
            lc_agg.executeOnBlock(b1, 0, m, 0);
            lc_agg.executeOnBlock(b2, 0, m, 0);
            Chunk r = lc_agg.convertToBlock();
            auto ids = r.columns[0]->getData<uint32_t>();
            auto sums = r.columns[1]->getData<int64_t>();
            auto cnts = r.columns[2]->getData<uint64_t>();
            REQUIRE(ids.size() == want_lc.size() && dict.size() == 4);
            for (size_t g = 0; g < ids.size(); ++g)
            {
                const auto & w = want_lc.at(dict.decode(ids[g]));
                REQUIRE(sums[g] == w.first && cnts[g] == w.second);
            }
        }

        // ---- threading contract: work() of different processors runs concurrently, each thread with its own Context ----
        {
            std::atomic<int> bad{0};
            std::vector<std::thread> pool;
            for (int t = 0; t < 4; ++t)
                pool.emplace_back([&, t] {
                    try
                    {
                        auto tctx = std::make_shared<Context>(0);
                        const size_t lo = n * t / 4, hi = n * (t + 1) / 4;
                        Chunk part;
                        part.columns = {ColumnVector::fromHost<int64_t>(tctx, a.data() + lo, hi - lo)};
                        part.num_rows = hi - lo;
                        GpuFilterTransform f(0, FunctionComparisonConst(CHGPU_LT, thr));
                        auto ag = std::make_shared<GpuAggregator>(tctx, -1, std::vector<AggregateDescription>{{CHGPU_AGG_SUM, CHGPU_I64, 0}, {CHGPU_AGG_COUNT, CHGPU_U64, 0}});
                        GpuAggregatingTransform at(ag, std::nullopt);
                        for (int rep = 0; rep < 5; ++rep)
                        {
                            f.setInput(part);
                            f.work();
                            if (f.hasOutput())
                                at.consume(f.pullOutput());
                        }
                        Chunk r = at.generate();
                        uint64_t ws = 0, wc = 0;
                        for (size_t i = lo; i < hi; ++i)
                            if (a[i] < thr)
                            {
                                ws += static_cast<uint64_t>(a[i]);
                                ++wc;
                            }
                        if (static_cast<uint64_t>(r.columns[0]->getData<int64_t>()[0]) != 5 * ws || r.columns[1]->getData<uint64_t>()[0] != 5 * wc)
                            ++bad;
                    }
                    catch (...)
                    {
                        ++bad;
                    }
                });
            for (auto & th : pool)
                th.join();
            REQUIRE(bad == 0);
        }

        // ---- many streams -> one result: every pipeline stream aggregates into its own variant on its own Context; the stream that
        //      finishes last merges them (ManyAggregatedData, AggregatingTransform.cpp:728-744 -> Aggregator::mergeDataImpl) ----
        {
            constexpr int STREAMS = 4;
            std::vector<ContextPtr> tctx;
            std::vector<std::shared_ptr<GpuAggregator>> variants;
            for (int t = 0; t < STREAMS; ++t)
            {
                tctx.push_back(std::make_shared<Context>(0));
                variants.push_back(std::make_shared<GpuAggregator>(tctx.back(), CHGPU_U32, std::vector<AggregateDescription>{{CHGPU_AGG_SUM, CHGPU_I64, 0}, {CHGPU_AGG_COUNT, CHGPU_U64, 0}}));
            }
            auto many = std::make_shared<ManyAggregatedData>(variants);
            std::vector<std::unique_ptr<GpuAggregatingTransform>> transforms;
            for (int t = 0; t < STREAMS; ++t)
                transforms.push_back(std::make_unique<GpuAggregatingTransform>(many, t, 1));
            std::atomic<int> bad{0};
            std::vector<std::thread> pool;
            for (int t = 0; t < STREAMS; ++t)
                pool.emplace_back([&, t] {
                    try
                    {
                        const size_t lo = n * t / STREAMS, hi = n * (t + 1) / STREAMS;
                        for (size_t b = lo; b < hi; b += 300000) // the stream's input arrives chunk by chunk
                        {
                            const size_t rows = std::min<size_t>(300000, hi - b);
                            Chunk part;
                            part.columns = {ColumnVector::fromHost<int64_t>(tctx[t], a.data() + b, rows), ColumnVector::fromHost<uint32_t>(tctx[t], k.data() + b, rows)};
                            part.num_rows = rows;
                            transforms[t]->consume(std::move(part));
                        }
                        transforms[t]->work(); // input finished: the last stream to arrive here merges
                    }
                    catch (...)
                    {
                        ++bad;
                    }
                });
            for (auto & th : pool)
                th.join();
            REQUIRE(bad == 0);
            int generating = 0;
            for (auto & tr : transforms)
                if (tr->isGenerating())
                {
                    ++generating;
                    Chunk r = tr->generate();
                    REQUIRE(r.num_rows == want.size());
                    auto gk = r.columns[0]->getData<uint32_t>();
                    auto gs = r.columns[1]->getData<int64_t>();
                    auto gc = r.columns[2]->getData<uint64_t>();
                    for (size_t g = 0; g < gk.size(); ++g)
                    {
                        const auto & w = want.at(gk[g]);
                        REQUIRE(static_cast<uint64_t>(gs[g]) == w.first && gc[g] == w.second);
                    }
                }
            REQUIRE(generating == 1);
        }

        // ---- constant filter columns: ConstantFilterDescription (FilterDescription.cpp:20-48, FilterTransform.cpp:153-176, :248-249) ----
        {
            const uint8_t one = 1, zero = 0;
            auto c1 = ColumnVector::createConst(ColumnVector::fromHost<uint8_t>(ctx, &one, 1), n);
            auto c0 = ColumnVector::createConst(ColumnVector::fromHost<uint8_t>(ctx, &zero, 1), n);
            REQUIRE(ConstantFilterDescription(*c1).always_true && !ConstantFilterDescription(*c1).always_false);
            REQUIRE(ConstantFilterDescription(*c0).always_false && !ConstantFilterDescription(*stripe.columns[0]).always_true);
            Chunk in;
            in.columns = {stripe.columns[0], c1};
            in.num_rows = n;
            GpuFilterTransform keep_all(1, true); // WHERE 1: columns untouched, the filter column removed
            keep_all.setInput(in);
            keep_all.work();
            REQUIRE(keep_all.hasOutput());
            Chunk out_all = keep_all.pullOutput();
            REQUIRE(out_all.num_rows == n && out_all.columns.size() == 1 && out_all.columns[0].get() == stripe.columns[0].get());
            in.columns = {stripe.columns[0], c0};
            GpuFilterTransform keep_none(1, true); // WHERE 0: the chunk is dropped
            keep_none.setInput(in);
            keep_none.work();
            REQUIRE(!keep_none.hasOutput());
            // an ordinary filter column next to a constant column: the constant is cut, not filtered
            if (want_cnt != 0 && want_cnt != n) // (a chunk in which every or no row passes never reaches IColumn::filter)
            {
            auto mask = FunctionComparisonConst(CHGPU_LT, thr).executeImpl(*stripe.columns[0]);
            const int64_t seven = 7;
            in.columns = {stripe.columns[0], mask, ColumnVector::createConst(ColumnVector::fromHost<int64_t>(ctx, &seven, 1), n)};
            GpuFilterTransform some(1, false);
            some.setInput(in);
            some.work();
            REQUIRE(some.hasOutput());
            Chunk out_some = some.pullOutput();
            REQUIRE(out_some.num_rows == want_cnt && out_some.columns.size() == 3);
            REQUIRE(out_some.columns[0]->size() == want_cnt && out_some.columns[1]->size() == want_cnt);
            REQUIRE(out_some.columns[2]->isConst() && out_some.columns[2]->size() == want_cnt && out_some.columns[2]->getDataColumnPtr()->getData<int64_t>()[0] == 7);
            uint64_t s2 = 0;
            for (auto v : out_some.columns[0]->getData<int64_t>())
                s2 += static_cast<uint64_t>(v);
            REQUIRE(s2 == want_sum);
            }
        }

        // ---- a star join (SSB Q4.1's shape in small): fact rows through two semi joins and two payload joins, once as the chain of
        //      JoiningTransforms (one joinBlock + block.filter per join) and once as ONE GpuJoinChainTransform (late materialisation);
        //      both feed the same GROUP BY and must agree row for row -------------------------------------------------------------
        {
            const size_t fact_rows = std::max<size_t>(n, 1500000), D1 = 20000, D2 = 30000, D3 = 300000, D4 = 400;
            std::vector<uint32_t> f1(fact_rows), f2(fact_rows), f3(fact_rows), f4(fact_rows), rev(fact_rows);
            uint64_t y = 0x9E3779B97F4A7C15ull;
            auto next = [&] { y ^= y << 13; y ^= y >> 7; y ^= y << 17; return y; };
            for (size_t i = 0; i < fact_rows; ++i)
            {
                f1[i] = 1 + next() % D1; f2[i] = 1 + next() % D2; f3[i] = 1 + next() % D3; f4[i] = 19920101 + next() % D4; rev[i] = next() % 1000000;
            }
            auto dim_keys = [&](size_t domain, uint32_t base, uint32_t keep_of, uint32_t keep) {
                std::vector<uint32_t> kk;
                for (uint32_t v = 0; v < domain; ++v)
                    if ((v * 2654435761u >> 7) % keep_of < keep)
                        kk.push_back(base + v);
                return kk;
            };
            const auto k1 = dim_keys(D1, 1, 5, 1), k2 = dim_keys(D2, 1, 5, 2), k3 = dim_keys(D3, 1, 5, 1), k4 = dim_keys(D4, 19920101, 1, 1);
            std::vector<uint8_t> nation(k3.size());
            std::vector<uint32_t> year(k4.size());
            for (size_t i = 0; i < k3.size(); ++i) nation[i] = static_cast<uint8_t>(k3[i] % 25);
            for (size_t i = 0; i < k4.size(); ++i) year[i] = 1992 + static_cast<uint32_t>(i / 366);
            // (addBlockToJoin keeps every column of the right Block as the right payload, the key included)
            auto build = [&](int kind, int strictness, const std::vector<uint32_t> & kk, ColumnPtr payload) {
                auto j = std::make_shared<GpuHashJoin>(ctx, CHGPU_U32, kind, strictness);
                Chunk b;
                b.columns = {ColumnVector::fromHost<uint32_t>(ctx, kk.data(), kk.size())};
                if (payload)
                    b.columns.push_back(payload);
                b.num_rows = kk.size();
                j->addBlockToJoin(b, 0);
                j->onBuildPhaseFinish();
                return j;
            };
            Chunk fact;
            fact.columns = {ColumnVector::fromHost<uint32_t>(ctx, f1.data(), fact_rows), ColumnVector::fromHost<uint32_t>(ctx, f2.data(), fact_rows),
                            ColumnVector::fromHost<uint32_t>(ctx, f3.data(), fact_rows), ColumnVector::fromHost<uint32_t>(ctx, f4.data(), fact_rows),
                            ColumnVector::fromHost<uint32_t>(ctx, rev.data(), fact_rows)};
            fact.num_rows = fact_rows;
            auto run = [&](bool chain) {
                auto j1 = build(CHGPU_JOIN_LEFT, CHGPU_STRICT_SEMI, k1, nullptr);
                auto j2 = build(CHGPU_JOIN_LEFT, CHGPU_STRICT_SEMI, k2, nullptr);
                auto j3 = build(CHGPU_JOIN_INNER, CHGPU_STRICT_ALL, k3, ColumnVector::fromHost<uint8_t>(ctx, nation.data(), nation.size()));
                auto j4 = build(CHGPU_JOIN_INNER, CHGPU_STRICT_ALL, k4, ColumnVector::fromHost<uint32_t>(ctx, year.data(), year.size()));
                Chunk c = fact;
                if (chain)
                {
                    GpuJoinChainTransform t({{j1, 0}, {j2, 1}, {j3, 2}, {j4, 3}});
                    t.setInput(c);
                    t.work();
                    c = t.hasOutput() ? t.pullOutput() : Chunk{};
                }
                else
                {
                    std::shared_ptr<Chunk> rest;
                    j1->joinBlock(c, 0, rest);
                    j2->joinBlock(c, 1, rest);
                    j3->joinBlock(c, 2, rest);
                    j4->joinBlock(c, 3, rest);
                }
                return c;
            };
            Chunk a_chain = run(true), a_each = run(false);
            REQUIRE(a_chain.num_rows == a_each.num_rows && a_chain.num_rows > 0);
            REQUIRE(a_chain.columns.size() == a_each.columns.size());
            // SEMI joins built through addBlockToJoin carry their key column as payload too; every column must agree
            for (size_t c = 0; c < a_chain.columns.size(); ++c)
            {
                REQUIRE(a_chain.columns[c]->getDataType() == a_each.columns[c]->getDataType());
                if (a_chain.columns[c]->getDataType() == CHGPU_U32)
                    REQUIRE(a_chain.columns[c]->getData<uint32_t>() == a_each.columns[c]->getData<uint32_t>());
                else
                    REQUIRE(a_chain.columns[c]->getData<uint8_t>() == a_each.columns[c]->getData<uint8_t>());
            }
            // and against the host: the rows whose four keys are all present
            auto has = [](const std::vector<uint32_t> & kk, uint32_t v) { return std::binary_search(kk.begin(), kk.end(), v); };
            uint64_t want_rows = 0, want_rev = 0;
            for (size_t i = 0; i < fact_rows; ++i)
                if (has(k1, f1[i]) && has(k2, f2[i]) && has(k3, f3[i]) && has(k4, f4[i]))
                {
                    ++want_rows;
                    want_rev += rev[i];
                }
            REQUIRE(a_chain.num_rows == want_rows);
            uint64_t got_rev = 0;
            for (auto v : a_chain.columns[4]->getData<uint32_t>())
                got_rev += v;
            REQUIRE(got_rev == want_rev);
        }

        // ---- FULL JOIN: the LEFT probe + used flags, then getNonJoinedBlocks ------------------------------------------------
        {
            auto full = std::make_shared<GpuHashJoin>(ctx, CHGPU_U64, CHGPU_JOIN_FULL, CHGPU_STRICT_ALL);
            std::vector<uint64_t> rk1 = {1, 2, 3}, rk2 = {2, 9};
            std::vector<int64_t> rv1 = {10, 20, 30}, rv2 = {21, 90};
            Chunk r1, r2;
            r1.columns = {ColumnVector::fromHost<uint64_t>(ctx, rk1.data(), 3), ColumnVector::fromHost<int64_t>(ctx, rv1.data(), 3)};
            r1.num_rows = 3;
            r2.columns = {ColumnVector::fromHost<uint64_t>(ctx, rk2.data(), 2), ColumnVector::fromHost<int64_t>(ctx, rv2.data(), 2)};
            r2.num_rows = 2;
            full->addBlockToJoin(r1, 0);
            full->addBlockToJoin(r2, 0);
            full->onBuildPhaseFinish();
            std::vector<uint64_t> lk = {2, 5};
            Chunk l;
            l.columns = {ColumnVector::fromHost<uint64_t>(ctx, lk.data(), 2)};
            l.num_rows = 2;
            std::shared_ptr<Chunk> rest;
            full->joinBlock(l, 0, rest);
            REQUIRE(l.num_rows == 3); // (2,20) (2,21) (5,default)
            int64_t joined_sum = 0;
            for (int64_t v : l.columns.back()->getData<int64_t>())
                joined_sum += v;
            REQUIRE(joined_sum == 41);
            Chunk nj = full->getNonJoinedBlock(); // right rows 1, 3 (block 0) and 9 (block 1), in insertion order
            REQUIRE(nj.num_rows == 3);
            REQUIRE((nj.columns[0]->getData<uint64_t>() == std::vector<uint64_t>{1, 3, 9}));
            REQUIRE((nj.columns[1]->getData<int64_t>() == std::vector<int64_t>{10, 30, 90}));
        }

        // ---- ORDER BY k DESC, a ASC LIMIT 5 over the first stripe rows; and a NONE-compressed column file read back ------------
        {
            const size_t m = std::min<size_t>(n, 50000);
            Chunk blk;
            blk.columns = {ColumnVector::fromHost<int64_t>(ctx, a.data(), m), ColumnVector::fromHost<uint32_t>(ctx, k.data(), m)};
            blk.num_rows = m;
            sortBlock(blk, {{1, -1, 1}, {0, 1, 1}}, 5);
            std::vector<size_t> order(m);
            for (size_t i = 0; i < m; ++i)
                order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return k[x] != k[y] ? k[x] > k[y] : a[x] < a[y]; });
            auto sa2 = blk.columns[0]->getData<int64_t>();
            auto sk2 = blk.columns[1]->getData<uint32_t>();
            REQUIRE(blk.num_rows == std::min<size_t>(5, m));
            for (size_t i = 0; i < blk.num_rows; ++i)
                REQUIRE(sa2[i] == a[order[i]] && sk2[i] == k[order[i]]);
            // a column file of one uncompressed (method NONE) frame: checksum(16) | 0x02 | compressed size | decompressed size | payload
            std::vector<unsigned char> file(16 + 9 + m * 8, 0);
            file[16] = 0x02;
            const uint32_t csz = static_cast<uint32_t>(9 + m * 8), dsz = static_cast<uint32_t>(m * 8);
            std::memcpy(&file[17], &csz, 4);
            std::memcpy(&file[21], &dsz, 4);
            std::memcpy(&file[25], a.data(), m * 8);
            uint64_t cksum[2];
            check(chgpu_city_hash128(&file[16], csz, cksum)); // CompressedWriteBuffer: CityHash128 over header + payload, {low64, high64}
            std::memcpy(&file[0], cksum, 16);
            auto col = readCompressedColumn(ctx, file.data(), file.size(), CHGPU_I64);
            auto back = col->getData<int64_t>();
            REQUIRE(back.size() == m && std::equal(back.begin(), back.end(), a.begin()));
            file[file.size() - 1] ^= 0x10; // one flipped payload bit: refused, not decoded into a wrong column
            bool refused = false;
            try { readCompressedColumn(ctx, file.data(), file.size(), CHGPU_I64); } catch (const Exception & e) { refused = std::string(e.what()).find("Checksum") != std::string::npos; }
            REQUIRE(refused);
        }

        // ---- ISimpleTransform::work never unwinds the executor thread: a chunk whose columns disagree in length (a broken upstream)
        //      makes IColumn::filter throw SIZES_OF_COLUMNS_DOESNT_MATCH inside transform(); the exception travels in the output slot
        //      (ISimpleTransform.cpp:88-99) and surfaces where the port is pulled ---------------------------------------------
        if (want_cnt != 0 && want_cnt != n) // (a chunk in which every or no row passes never reaches IColumn::filter)
        {
            Chunk broken;
            broken.columns = {stripe.columns[0], stripe.columns[1]->cut(0, n / 2, stripe.columns[1])};
            broken.num_rows = n;
            GpuFilterTransform f(0, FunctionComparisonConst(CHGPU_LT, thr));
            f.setInput(broken);
            bool unwound = false;
            try { f.work(); } catch (...) { unwound = true; }
            REQUIRE(!unwound && f.hasOutput() && f.outputHasException());
            int code = 0;
            try { f.pullOutput(); } catch (const Exception & e) { code = e.code(); }
            REQUIRE(code == CHGPU_ERR_SIZES_MISMATCH);
            f.setInput(stripe); // the processor is still usable afterwards
            f.work();
            REQUIRE(f.hasOutput() && !f.outputHasException() && f.pullOutput().num_rows == want_cnt);
        }

        // ---- stripes streamed through pinned double buffers: the upload of stripe s+1 overlaps the kernels of stripe s; the
        //      PCIe-inclusive rate of `SELECT sum(a), count() WHERE a < thr` fed from host Blocks ------------------------------
        {
            const size_t stripe_rows = std::min<size_t>(n, size_t(4) << 20), reps = n >= (size_t(1) << 22) ? 4 : 1;
            StripeBuilder<int64_t> sb(ctx, stripe_rows);
            auto agg = std::make_shared<GpuAggregator>(ctx, -1, std::vector<AggregateDescription>{{CHGPU_AGG_SUM, CHGPU_I64, 0}, {CHGPU_AGG_COUNT, CHGPU_U64, 0}});
            GpuAggregatingTransform at(agg, std::nullopt);
            GpuFilterTransform f(0, FunctionComparisonConst(CHGPU_LT, thr));
            auto run_stripe = [&] {
                Chunk c;
                c.num_rows = sb.rows();
                c.columns = {sb.flush()};
                f.setInput(std::move(c));
                f.work();
                if (f.hasOutput())
                    at.consume(f.pullOutput());
            };
            const auto t0 = std::chrono::steady_clock::now();
            for (size_t rep = 0; rep < reps; ++rep)
                for (size_t b = 0; b < n; b += DEFAULT_BLOCK_SIZE)
                {
                    size_t rows = std::min(DEFAULT_BLOCK_SIZE, n - b), done = 0;
                    while (done < rows)
                    {
                        done += sb.appendBlock(a.data() + b + done, rows - done);
                        if (sb.room() == 0)
                            run_stripe();
                    }
                }
            if (sb.rows())
                run_stripe();
            Chunk r = at.generate();
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            REQUIRE(static_cast<uint64_t>(r.columns[0]->getData<int64_t>()[0]) == reps * want_sum && r.columns[1]->getData<uint64_t>()[0] == reps * want_cnt);
            std::printf("striped upload + filter + sum: %.3g rows/s PCIe-inclusive (%zu rows, stripes of %zu)\n", double(n * reps) / secs, n * reps, stripe_rows);
        }

        // ---- the sharded operators at world size 1: RCCL is loaded and initialised, dispatchBlock degenerates to "all rows are mine";
        //      results must equal the single-GPU operators' (the exchange itself runs in the multi-rank tests) -------------------
        {
            auto comm = std::make_shared<Communicator>(ctx, 0, 1, Communicator::uniqueId());
            REQUIRE(comm->rank() == 0 && comm->world() == 1);
            GpuShardedAggregator sagg(ctx, comm, CHGPU_U32, {{CHGPU_AGG_SUM, CHGPU_I64, 0}, {CHGPU_AGG_COUNT, CHGPU_U64, 0}}, 1000);
            sagg.executeOnBlock(stripe.columns, 0, n, 1);
            Chunk rs = sagg.convertToBlock();
            REQUIRE(rs.num_rows == want.size());
            auto sk2 = rs.columns[0]->getData<uint32_t>();
            auto ss2 = rs.columns[1]->getData<int64_t>();
            auto sc2 = rs.columns[2]->getData<uint64_t>();
            for (size_t i = 0; i < sk2.size(); ++i)
                REQUIRE(static_cast<uint64_t>(ss2[i]) == want.at(sk2[i]).first && sc2[i] == want.at(sk2[i]).second);
            GpuConcurrentHashJoin cj(ctx, comm, CHGPU_U64, CHGPU_JOIN_INNER, CHGPU_STRICT_ALL);
            Chunk right;
            right.columns = {ColumnVector::fromHost<uint64_t>(ctx, bk.data(), nb), ColumnVector::fromHost<int64_t>(ctx, bv.data(), nb)};
            right.num_rows = nb;
            cj.addBlockToJoin(right, 0);
            cj.onBuildPhaseFinish();
            auto cs = cj.joinCountSum(left, 0, 1);
            int64_t want_bv = 0;
            for (auto key : pk)
            {
                auto range = build.equal_range(key);
                for (auto it = range.first; it != range.second; ++it)
                    want_bv += it->second;
            }
            REQUIRE(cs.first == want_rows && static_cast<int64_t>(cs.second) == want_bv);
            std::vector<uint64_t> v{7, 9};
            comm->allReduce(v);
            comm->barrier();
            REQUIRE(v[0] == 7 && v[1] == 9);
        }

        // ---- MergingAggregatedMemoryEfficientTransform: two sources' partial states (one as split two-level blocks bucket by bucket,
        //      one unsplit) merged on the initiator; blocks come out in bucket order and add up to the aggregation of all rows ---------
        {
            const std::vector<AggregateDescription> descr{{CHGPU_AGG_SUM, CHGPU_I64, 0}, {CHGPU_AGG_COUNT, CHGPU_U64, 0}};
            const int kinds[2] = {CHGPU_AGG_SUM, CHGPU_AGG_COUNT}, types[2] = {CHGPU_I64, CHGPU_U64};
            const size_t half = n / 2;
            chgpu_agg * src[2] = {nullptr, nullptr};
            for (int sidx = 0; sidx < 2; ++sidx)
            {
                check(chgpu_agg_create(ctx->get(), CHGPU_U32, 2, kinds, types, 1000, &src[sidx]));
                const chgpu_col * args[2] = {stripe.columns[0]->handle(), nullptr};
                check(chgpu_agg_add_block(src[sidx], stripe.columns[1]->handle(), args, sidx == 0 ? 0 : half, sidx == 0 ? half : n));
            }
            GpuMergingAggregatedTransform merging(ctx, CHGPU_U32, descr, 2);
            auto as_chunk = [&](chgpu_col * k, chgpu_col * w0, chgpu_col * w1, uint64_t rows) {
                Chunk c;
                c.num_rows = rows;
                c.columns = {std::make_shared<ColumnVector>(ctx, k), std::make_shared<ColumnVector>(ctx, w0), std::make_shared<ColumnVector>(ctx, w1)};
                return c;
            };
            {
                chgpu_col * k = nullptr, * w[2] = {nullptr, nullptr};
                uint64_t groups = 0;
                check(chgpu_agg_export_states(src[1], &k, w, &groups));
                merging.addChunk(1, AggregatedChunk{as_chunk(k, w[0], w[1], groups), -1, false}); // unsplit
                merging.finishInput(1);
            }
            std::vector<AggregatedChunk> merged;
            if (half)
            {
                chgpu_col * k = nullptr, * w[2] = {nullptr, nullptr};
                uint64_t groups = 0, counts[256] = {};
                check(chgpu_agg_export_states_two_level(src[0], &k, w, &groups, counts));
                Chunk whole = as_chunk(k, w[0], w[1], groups);
                uint64_t begin = 0;
                for (int32_t b = 0; b < 256; ++b)
                {
                    if (counts[b])
                    {
                        AggregatedChunk c;
                        c.bucket_num = b;
                        c.chunk.num_rows = counts[b];
                        for (auto & col : whole.columns)
                            c.chunk.columns.push_back(col->cut(begin, counts[b], col));
                        merging.addChunk(0, std::move(c));
                    }
                    begin += counts[b];
                    if (b == 100)
                        for (auto & m : merging.pull())
                        {
                            REQUIRE(m.bucket_num < 100);
                            merged.push_back(std::move(m));
                        }
                }
            }
            merging.finishInput(0);
            for (auto & m : merging.pull())
                merged.push_back(std::move(m));
            REQUIRE(merging.pull().empty());
            size_t groups_out = 0;
            int32_t last_bucket = -2;
            for (auto & m : merged)
            {
                REQUIRE(!m.is_overflows && (m.bucket_num > last_bucket || (m.bucket_num == -1 && merged.size() == 1)));
                last_bucket = m.bucket_num;
                auto mk = m.chunk.columns[0]->getData<uint32_t>();
                auto ms = m.chunk.columns[1]->getData<int64_t>();
                auto mc = m.chunk.columns[2]->getData<uint64_t>();
                for (size_t i = 0; i < mk.size(); ++i)
                    REQUIRE(static_cast<uint64_t>(ms[i]) == want.at(mk[i]).first && mc[i] == want.at(mk[i]).second);
                groups_out += mk.size();
            }
            REQUIRE(groups_out == want.size());
            chgpu_agg_free(src[0]);
            chgpu_agg_free(src[1]);
        }

        // ---- ASOF LEFT JOIN: trades(k, t) against quotes(k, t, v) inserted as two Blocks: every trade gets the latest quote at or before it ----
        {
            const size_t nq = 2000, nt = 3000;
            std::vector<uint32_t> qk(nq), qt(nq), tk(nt), tt(nt);
            std::vector<int64_t> qv(nq);
            for (size_t i = 0; i < nq; ++i)
            {
                qk[i] = static_cast<uint32_t>(i % 40);
                qt[i] = static_cast<uint32_t>((i / 40) * 7 + 3); // per key: times 3, 10, 17, ...
                qv[i] = static_cast<int64_t>(qk[i]) * 1000000 + qt[i];
            }
            for (size_t i = 0; i < nt; ++i)
            {
                tk[i] = static_cast<uint32_t>((i * 13) % 44); // keys 40..43 have no quotes
                tt[i] = static_cast<uint32_t>((i * 29) % 400);
            }
            GpuAsofJoin asof(ctx, CHGPU_U32, CHGPU_U32, CHGPU_JOIN_LEFT);
            for (size_t part = 0; part < 2; ++part)
            {
                const size_t lo = part * nq / 2, cnt = nq / 2;
                Chunk q;
                q.columns = {ColumnVector::fromHost<uint32_t>(ctx, qk.data() + lo, cnt), ColumnVector::fromHost<uint32_t>(ctx, qt.data() + lo, cnt),
                             ColumnVector::fromHost<int64_t>(ctx, qv.data() + lo, cnt)};
                q.num_rows = cnt;
                asof.addBlockToJoin(q, 0, 1);
            }
            Chunk t;
            t.columns = {ColumnVector::fromHost<uint32_t>(ctx, tk.data(), nt), ColumnVector::fromHost<uint32_t>(ctx, tt.data(), nt)};
            t.num_rows = nt;
            asof.joinBlock(t, 0, 1);
            REQUIRE(t.num_rows == nt && t.columns.size() == 5);
            auto got_v = t.columns[4]->getData<int64_t>();
            for (size_t i = 0; i < nt; ++i)
            {
                int64_t want_v = 0; // the default row
                if (tk[i] < 40 && tt[i] >= 3)
                {
                    const uint32_t best = (tt[i] - 3) / 7 * 7 + 3; // the greatest quote time <= the trade's
                    want_v = static_cast<int64_t>(tk[i]) * 1000000 + (best > 49 * 7 + 3 ? 49 * 7 + 3 : best);
                }
                REQUIRE(got_v[i] == want_v);
            }
        }

        // unsupported surface -> NOT_IMPLEMENTED (CPU fallback signal), not a crash
        bool fell_back = false;
        try
        {
            GpuHashJoin full_any(ctx, CHGPU_U64, CHGPU_JOIN_FULL, CHGPU_STRICT_ANY); // FULL ANY: a TODO in the reference too (HashJoinMethodsImpl.h:511-514)
        }
        catch (const Exception & e)
        {
            fell_back = e.isNotImplemented();
        }
        REQUIRE(fell_back);

        std::printf("pipeline_demo OK: rows=%zu filter+sum=(%llu,%llu) groups=%zu join_rows=%llu\n", n, (unsigned long long)want_sum,
                    (unsigned long long)want_cnt, want.size(), (unsigned long long)got_rows);
        return 0;
    }
    catch (const Exception & e)
    {
        std::fprintf(stderr, "chgpu::Exception %d: %s\n", e.code(), e.what());
        return 2;
    }
}
