// Host-only check of the prelude's file readers (no GPU call): prints what each golden file parses to;
// tests/test_io_formats.py compares the lines with the Python readers of graph_amd.prelude.
#include <cstdio>
#include <string>

#include "graph_prelude.hpp"

static void report(const char *name, const graph::EdgeData &d)
{
    unsigned long long hs = 1469598103934665603ull; // FNV-1a over (src, dst) pairs
    for (size_t i = 0; i < d.src.size(); ++i) {
        hs = (hs ^ d.src[i]) * 1099511628211ull;
        hs = (hs ^ d.dst[i]) * 1099511628211ull;
    }
    double wsum = 0.0;
    for (float v : d.values)
        wsum += v;
    std::printf("%s nodes=%llu edges=%zu values=%zu hash=%llu wsum=%.9g\n", name, (unsigned long long)d.node_count,
                d.src.size(), d.values.size(), hs, wsum);
}

// the reference's unit tests of greedy_node_map_partition (crates/builder/src/graph_ops.rs:673-708) print their ranges
template <class F> static void partition_line(const char *name, F node_map, size_t n, uint64_t batch, size_t max_batches)
{
    std::vector<uint64_t> prefix(n + 1, 0);
    for (size_t v = 0; v < n; ++v)
        prefix[v + 1] = prefix[v] + node_map(v);
    std::printf("%s ranges=", name);
    const auto parts = graph::greedy_node_map_partition<uint32_t>(prefix, batch, max_batches);
    for (size_t i = 0; i < parts.size(); ++i)
        std::printf("%s%u-%u", i ? "," : "", parts[i].first, parts[i].second);
    std::printf("\n");
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : ".";
    using namespace graph::prelude;
    try {
        report("example.el", EdgeListInput{false}.read(dir + "/example.el"));
        report("example.wel", EdgeListInput{true}.read(dir + "/example.wel"));
        report("windows.el", EdgeListInput{false}.read(dir + "/windows.el"));
        report("scale_8.graph500", Graph500Input{}.read(dir + "/scale_8.graph500"));
        partition_line("partition_1_part", [](size_t) { return (uint64_t)1; }, 10, 10, 99999);
        partition_line("partition_2_parts", [](size_t x) { return (uint64_t)(x % 2); }, 10, 4, 99999);
        partition_line("partition_6_parts", [](size_t x) { return (uint64_t)x; }, 10, 6, 99999);
        partition_line("partition_max_batches", [](size_t x) { return (uint64_t)x; }, 10, 6, 3);
        partition_line("partition_empty", [](size_t) { return (uint64_t)0; }, 0, 3, 2);
        bool threw = false;
        try {
            EdgeListInput{false}.read(dir + "/does-not-exist.el");
        } catch (const graph::Error &) {
            threw = true;
        }
        std::printf("missing-file-throws=%d\n", threw ? 1 : 0);
    } catch (const std::exception &e) {
        std::printf("FAILED: %s\n", e.what());
        return 1;
    }
    return 0;
}
