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

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : ".";
    using namespace graph::prelude;
    try {
        report("example.el", EdgeListInput{false}.read(dir + "/example.el"));
        report("example.wel", EdgeListInput{true}.read(dir + "/example.wel"));
        report("windows.el", EdgeListInput{false}.read(dir + "/windows.el"));
        report("scale_8.graph500", Graph500Input{}.read(dir + "/scale_8.graph500"));
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
