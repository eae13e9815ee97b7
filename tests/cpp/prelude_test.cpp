// prelude_test.cpp — the reference's own unit tests restated against the C++ prelude mirror
// (include/graph_prelude.hpp) over the C ABI.  Built by __graft_entry__.build(); run on a GPU by
// tests/test_gpu_cpp_prelude.py.  Each block cites the reference test it mirrors.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "graph_prelude.hpp"

using namespace graph::prelude;

static int failures = 0;
#define EXPECT(cond)                                                              \
    do {                                                                          \
        if (!(cond)) {                                                            \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);           \
            ++failures;                                                           \
        }                                                                         \
    } while (0)

static bool bits_equal(const std::vector<float> &a, const std::vector<float> &b)
{
    return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0;
}

int main(int argc, char **argv)
{
    const std::string golden = argc > 1 ? argv[1] : "";
    try {
        { // crates/algos/src/lib.rs:92-141 (README doc-test): bit-exact scores, 10 iterations
            auto g = GraphBuilder()
                         .edges({{1, 2}, {2, 1}, {4, 0}, {4, 1}, {5, 4}, {5, 1}, {5, 6}, {6, 1}, {6, 5}, {7, 1},
                                 {7, 5}, {8, 1}, {8, 5}, {9, 1}, {9, 5}, {10, 1}, {10, 5}, {11, 5}, {12, 5}})
                         .build<DirectedCsrGraph<uint32_t>>();
            auto [scores, iterations, error] = page_rank(g, PageRankConfig{10, 1e-4, 0.85f});
            std::vector<float> expected = {0.024064068f, 0.3145448f,  0.27890152f, 0.01153846f, 0.029471997f,
                                           0.06329483f,  0.029471997f, 0.01153846f, 0.01153846f, 0.01153846f,
                                           0.01153846f,  0.01153846f,  0.01153846f};
            EXPECT(iterations == 10);
            EXPECT(bits_equal(scores, expected));
            (void)error;
        }
        { // crates/algos/src/page_rank.rs:175-197
            auto g = GraphBuilder()
                         .csr_layout(CsrLayout::Sorted)
                         .edges({{0, 1}, {1, 2}, {0, 2}, {3, 4}, {4, 5}, {3, 5}})
                         .build<DirectedCsrGraph<uint32_t>>();
            auto [scores, iterations, error] = page_rank(g);
            std::vector<float> expected = {0.024999997f, 0.035624996f, 0.06590624f,
                                           0.024999997f, 0.035624996f, 0.06590624f};
            EXPECT(bits_equal(scores, expected));
            EXPECT(g.out_degree(0) == 2 && g.in_degree(2) == 2 && g.out_neighbors(0)[1] == 2);
            auto ug = g.to_undirected(CsrLayout::Deduplicated); // two directed triangles -> 2 triangles
            EXPECT(ug.edge_count() == 6 && global_triangle_count(ug) == 2);
            (void)iterations;
            (void)error;
        }
        { // the same call split over GPUs through the C ABI (RCCL from the compiled host): a communicator of the
          // size this box has, and three virtual ranks on device 0
            auto g = GraphBuilder()
                         .csr_layout(CsrLayout::Sorted)
                         .edges({{0, 1}, {1, 2}, {0, 2}, {3, 4}, {4, 5}, {3, 5}, {5, 0}, {2, 3}})
                         .build<DirectedCsrGraph<uint32_t>>();
            auto [one, it1, e1] = page_rank(g, PageRankConfig{12, 0.0, 0.85f}, GM_PR_JACOBI);
            for (const std::vector<int> &devs : {std::vector<int>{0}, std::vector<int>{0, 0, 0}}) {
                auto [many, it, e] = page_rank_multi(g, devs, PageRankConfig{12, 0.0, 0.85f});
                EXPECT(it == it1 && it == 12);
                for (size_t i = 0; i < one.size(); ++i)
                    EXPECT(std::fabs(many[i] - one[i]) <= 2e-7f * one[i]);
                (void)e;
            }
            // what the handles parked (plan, call state, the multi-GPU state) released: the next calls rebuild and agree
            g.release_device_caches();
            EXPECT(gm_trim(-1) == GM_OK);
            auto [again, it2, e2] = page_rank(g, PageRankConfig{12, 0.0, 0.85f}, GM_PR_JACOBI);
            EXPECT(it2 == it1 && again == one);
            (void)e2;
            (void)e1;
        }
        { // crates/algos/src/wcc.rs:307-329
            auto g = GraphBuilder().edges({{0, 1}, {2, 3}}).build<DirectedCsrGraph<uint32_t>>();
            for (int k = 0; k < 3; ++k) {
                auto res = k == 0 ? wcc_afforest(g) : k == 1 ? wcc_afforest_dss(g) : wcc_baseline(g);
                EXPECT(res.component(0) == res.component(1));
                EXPECT(res.component(2) == res.component(3));
                EXPECT(res.component(1) != res.component(2));
            }
        }
        { // crates/algos/src/sssp.rs:282-313
            auto g = GraphBuilder()
                         .csr_layout(CsrLayout::Deduplicated)
                         .edges_with_values({{0, 1, 4.0f}, {0, 2, 2.0f}, {1, 2, 5.0f}, {1, 3, 10.0f}, {2, 4, 3.0f},
                                             {3, 5, 11.0f}, {4, 3, 4.0f}})
                         .build<DirectedCsrGraph<uint32_t>>();
            auto dist = delta_stepping(g, DeltaSteppingConfig{0, 3.0f});
            EXPECT(bits_equal(dist, std::vector<float>{0.0f, 4.0f, 2.0f, 9.0f, 5.0f, 20.0f}));
            bool threw = false;
            try {
                delta_stepping(g, DeltaSteppingConfig{6, 3.0f}); // reference: index-out-of-bounds panic (sssp.rs:52)
            } catch (const graph::Error &e) {
                threw = e.status == GM_ERR_RANGE;
            }
            EXPECT(threw);
        }
        { // crates/algos/src/triangle_count.rs:93-130
            auto g = GraphBuilder()
                         .csr_layout(CsrLayout::Deduplicated)
                         .edges({{0, 1}, {1, 2}, {2, 0}, {3, 4}, {4, 5}, {5, 3}})
                         .build<UndirectedCsrGraph<uint32_t>>();
            EXPECT(global_triangle_count(g) == 2);
            relabel_graph(g);
            EXPECT(global_triangle_count(g) == 2);
            EXPECT(g.edge_count() == 6 && g.degree(0) == 2);
        }
        if (!golden.empty()) { // crates/builder/tests/builder.rs:448-491: the Graph500 fixture through the file reader
            auto g = GraphBuilder()
                         .csr_layout(CsrLayout::Sorted)
                         .file_format(Graph500Input{})
                         .path(golden + "/scale_8.graph500")
                         .build<DirectedCsrGraph<uint32_t>>();
            EXPECT(g.node_count() == 256 && g.edge_count() == 4096);
            auto o = g.out_neighbors(0);
            EXPECT(o.size() == 2 && o[0] == 37 && o[1] == 157);
            EXPECT(g.in_neighbors(0).size() == 14);
            auto ug = g.to_undirected<uint32_t>(CsrLayout::Sorted);
            relabel_graph(ug);
            EXPECT(global_triangle_count(ug) == 227874); // mate/tests/triangle_count_test.py:5-9 (after relabel)
            auto gw = GraphBuilder()
                          .csr_layout(CsrLayout::Deduplicated)
                          .file_format(EdgeListInput{true})
                          .path(golden + "/example.wel")
                          .build<DirectedCsrGraph<uint32_t>>();
            EXPECT(gw.node_count() == 4 && gw.edge_count() == 5);
        }
    } catch (const std::exception &e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
    return failures ? 1 : 0;
}
