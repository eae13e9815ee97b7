// graph_prelude.hpp — C++17 host-side mirror of the reference's `graph::prelude` over the C ABI of
// graph_mi355x.h (header-only; link with -lgraph_mi355x).
//
// The reference is Rust (crates/algos/src/prelude.rs:1-7 re-exporting crates/builder/src/prelude.rs);
// no Rust toolchain exists in the build image, so the compiled-language host layer is restated in C++
// with the reference's names, argument meaning and error behaviour (a reference panic = a thrown
// graph::Error).  The Rust shim a maintainer would add is bindings/rust/ (see INTEGRATION.md).
//
//   using namespace graph::prelude;
//   auto g = GraphBuilder().csr_layout(CsrLayout::Sorted).edges({{0,1},{1,2}}).build<DirectedCsrGraph<uint32_t>>();
//   auto [scores, iterations, error] = page_rank(g, PageRankConfig{10, 1e-4, 0.85f});
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "graph_mi355x.h"

namespace graph {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &msg) : std::runtime_error(msg), status(s) {}
};

namespace detail {
inline void check(int status)
{
    if (status != GM_OK)
        throw Error(status, gm_last_error());
}
struct CsrDeleter {
    void operator()(gm_csr *c) const { gm_csr_free(c); }
};
using CsrPtr = std::unique_ptr<gm_csr, CsrDeleter>;

struct HostCsr {
    std::vector<uint32_t> offsets, targets;
    std::vector<float> weights;
    bool loaded = false;
};

inline void load(const gm_csr *c, HostCsr &h)
{
    if (h.loaded)
        return;
    h.offsets.resize(gm_csr_node_count(c) + 1);
    h.targets.resize(gm_csr_edge_count(c));
    const bool weighted = gm_csr_weights_ptr(c) != 0;
    if (weighted)
        h.weights.resize(h.targets.size());
    check(gm_csr_download(c, h.offsets.data(), h.targets.empty() ? nullptr : h.targets.data(),
                          weighted && !h.weights.empty() ? h.weights.data() : nullptr));
    h.loaded = true;
}
} // namespace detail

// crates/builder/src/graph/csr.rs:34-45
enum class CsrLayout { Unsorted = 0, Sorted = 1, Deduplicated = 2 };

template <class T> struct Slice { // what `slice::Iter<NI>` gives the reference's callers
    const T *ptr;
    size_t len;
    const T *begin() const { return ptr; }
    const T *end() const { return ptr + len; }
    size_t size() const { return len; }
    const T &operator[](size_t i) const { return ptr[i]; }
};

// ---- greedy range partitions (crates/builder/src/graph_ops.rs:17-50 the traits, 331-440 their implementations) ----------
// A node range [first, second): the reference's std::ops::Range<NI>.
template <class NI> using Range = std::pair<NI, NI>;

// graph_ops.rs:479-509 over node_map(v) = prefix[v + 1] - prefix[v]: one pass over the nodes; a range is closed as soon as its
// sum reaches batch_size while fewer than max_batches - 1 ranges exist, the last range ends at the last node.
template <class NI, class Off>
inline std::vector<Range<NI>> greedy_node_map_partition(const std::vector<Off> &prefix, uint64_t batch_size, size_t max_batches)
{
    if (max_batches < 1)
        throw Error(GM_ERR_INVALID, "greedy_node_map_partition: max_batches must be at least 1");
    std::vector<Range<NI>> parts;
    if (prefix.size() < 2)
        return parts;
    const size_t n = prefix.size() - 1;
    uint64_t size = 0;
    size_t start = 0;
    for (size_t node = 0; node < n; ++node) {
        size += (uint64_t)(prefix[node + 1] - prefix[node]);
        if ((parts.size() < max_batches - 1 && size >= batch_size) || node == n - 1) {
            parts.emplace_back((NI)start, (NI)(node + 1));
            size = 0;
            start = node + 1;
        }
    }
    return parts;
}

namespace detail {
// batch = ceil(total / concurrency) (graph_ops.rs:358, 395, 432), then the greedy walk over the CSR's offsets
template <class NI> inline std::vector<Range<NI>> degree_partition(const std::vector<uint32_t> &offsets, uint64_t total, size_t concurrency)
{
    if (concurrency < 1)
        throw Error(GM_ERR_INVALID, "degree partition: concurrency must be at least 1");
    return greedy_node_map_partition<NI>(offsets, (total + concurrency - 1) / concurrency, concurrency);
}
} // namespace detail

// DirectedCsrGraph = csr_out + csr_inc (crates/builder/src/graph/csr.rs:364-520), resident in HBM.
template <class NI = uint32_t> class DirectedCsrGraph {
public:
    DirectedCsrGraph(detail::CsrPtr out, detail::CsrPtr inc, CsrLayout layout)
        : out_(std::move(out)), inc_(std::move(inc)), layout_(layout) {}
    NI node_count() const { return (NI)gm_csr_node_count(out_.get()); }
    NI edge_count() const { return (NI)gm_csr_edge_count(out_.get()); }
    NI out_degree(NI u) const { const auto &h = host_out(); check_node(u); return (NI)(h.offsets[u + 1] - h.offsets[u]); }
    NI in_degree(NI u) const { const auto &h = host_inc(); check_node(u); return (NI)(h.offsets[u + 1] - h.offsets[u]); }
    Slice<uint32_t> out_neighbors(NI u) const { const auto &h = host_out(); check_node(u); return {h.targets.data() + h.offsets[u], h.offsets[u + 1] - h.offsets[u]}; }
    Slice<uint32_t> in_neighbors(NI u) const { const auto &h = host_inc(); check_node(u); return {h.targets.data() + h.offsets[u], h.offsets[u + 1] - h.offsets[u]}; }
    // OutDegreePartitionOp / InDegreePartitionOp (graph_ops.rs:29-50, 368-440): at most `concurrency` ranges of roughly equal
    // total out- / in-degree (the in-degree ranges are what the multi-GPU PageRank shards its rows by)
    std::vector<Range<NI>> out_degree_partition(size_t concurrency) const { return detail::degree_partition<NI>(host_out().offsets, edge_count(), concurrency); }
    std::vector<Range<NI>> in_degree_partition(size_t concurrency) const { return detail::degree_partition<NI>(host_inc().offsets, edge_count(), concurrency); }
    const gm_csr *csr_out() const { return out_.get(); }
    const gm_csr *csr_inc() const { return inc_.get(); }
    CsrLayout layout() const { return layout_; }
    // not in the reference: releases what the device handles parked for later calls (PageRank plan and call state, SSSP / WCC
    // working sets, the multi-GPU state); the graph stays resident, the next call rebuilds what it needs
    void release_device_caches() const
    {
        detail::check(gm_csr_trim(out_.get()));
        detail::check(gm_csr_trim(inc_.get()));
    }
    // ToUndirectedOp::to_undirected (crates/builder/src/graph_ops.rs:176-230); defined after UndirectedCsrGraph
    template <class U = NI> auto to_undirected(CsrLayout layout) const;

private:
    void check_node(NI u) const { if ((uint64_t)u >= gm_csr_node_count(out_.get())) throw Error(GM_ERR_RANGE, "node id out of range"); }
    const detail::HostCsr &host_out() const { detail::load(out_.get(), hout_); return hout_; }
    const detail::HostCsr &host_inc() const { detail::load(inc_.get(), hinc_); return hinc_; }
    detail::CsrPtr out_, inc_;
    CsrLayout layout_;
    mutable detail::HostCsr hout_, hinc_;
};

// UndirectedCsrGraph = one symmetrised CSR (crates/builder/src/graph/csr.rs:658-732)
template <class NI = uint32_t> class UndirectedCsrGraph {
public:
    UndirectedCsrGraph(detail::CsrPtr csr, CsrLayout layout) : csr_(std::move(csr)), layout_(layout) {}
    NI node_count() const { return (NI)gm_csr_node_count(csr_.get()); }
    NI edge_count() const { return (NI)(gm_csr_edge_count(csr_.get()) / 2); } // csr.rs:687-689
    NI degree(NI u) const { const auto &h = host(); return (NI)(h.offsets[u + 1] - h.offsets[u]); }
    Slice<uint32_t> neighbors(NI u) const { const auto &h = host(); return {h.targets.data() + h.offsets[u], h.offsets[u + 1] - h.offsets[u]}; }
    // DegreePartitionOp (graph_ops.rs:17-26, 331-366): batch = ceil(2 edge_count / concurrency), every edge counts at both ends
    std::vector<Range<NI>> degree_partition(size_t concurrency) const { return detail::degree_partition<NI>(host().offsets, 2 * (uint64_t)edge_count(), concurrency); }
    // RelabelByDegreeOp::make_degree_ordered (crates/builder/src/graph_ops.rs:240-253, 511-638)
    void make_degree_ordered()
    {
        gm_csr *fresh = nullptr;
        detail::check(gm_csr_relabel_by_degree(csr_.get(), &fresh, nullptr));
        csr_.reset(fresh);
        host_ = detail::HostCsr{};
    }
    const gm_csr *csr() const { return csr_.get(); }
    void release_device_caches() const { detail::check(gm_csr_trim(csr_.get())); } // the triangle count's DAG, WCC working set

private:
    const detail::HostCsr &host() const { detail::load(csr_.get(), host_); return host_; }
    detail::CsrPtr csr_;
    CsrLayout layout_;
    mutable detail::HostCsr host_;
};

template <class NI> template <class U> auto DirectedCsrGraph<NI>::to_undirected(CsrLayout layout) const
{
    gm_csr *c = nullptr;
    detail::check(gm_csr_to_undirected(out_.get(), (int)layout, &c));
    return UndirectedCsrGraph<U>(detail::CsrPtr(c), layout);
}

// ---- on-disk inputs (host-side parsing; the CSR is then built on the device) --------------------------
// What a file holds: edges as u64 ids (narrowed to the u32 device id type at build time), optional f32
// values, and the node count the reference derives for that format.
struct EdgeData {
    std::vector<uint64_t> src, dst;
    std::vector<float> values; // empty unless the format carries them
    uint64_t node_count = 0;
};

namespace detail {
inline std::vector<char> read_file(const std::string &path)
{
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f)
        throw Error(GM_ERR_INVALID, "cannot open " + path);
    std::vector<char> buf;
    char chunk[1 << 16];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof(chunk), f)) > 0)
        buf.insert(buf.end(), chunk, chunk + got);
    std::fclose(f);
    return buf;
}
} // namespace detail

// `source target[ value]` per line, \n or \r\n (crates/builder/src/input/edgelist.rs:181-265);
// node_count = largest id + 1 (csr.rs:530)
struct EdgeListInput {
    bool weighted = false;
    EdgeData read(const std::string &path) const
    {
        std::vector<char> buf = detail::read_file(path);
        buf.push_back('\0');
        EdgeData out;
        const char *p = buf.data();
        const char *end = buf.data() + buf.size() - 1;
        auto skip_ws = [&] {
            while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n'))
                ++p;
        };
        for (;;) {
            skip_ws();
            if (p >= end)
                break;
            char *q = nullptr;
            const unsigned long long s = std::strtoull(p, &q, 10);
            if (q == p)
                throw Error(GM_ERR_INVALID, path + ": malformed edge list");
            p = q;
            skip_ws();
            const unsigned long long t = std::strtoull(p, &q, 10);
            if (q == p)
                throw Error(GM_ERR_INVALID, path + ": malformed edge list");
            p = q;
            out.src.push_back(s);
            out.dst.push_back(t);
            if (weighted) {
                skip_ws();
                const float v = std::strtof(p, &q);
                if (q == p)
                    throw Error(GM_ERR_INVALID, path + ": edge without a value");
                p = q;
                out.values.push_back(v);
            }
            if (s + 1 > out.node_count) out.node_count = s + 1;
            if (t + 1 > out.node_count) out.node_count = t + 1;
        }
        return out;
    }
};

// 12-byte packed edges {v0_low: u32, v1_low: u32, high: u32}: source = v0_low | (high & 0xFFFF) << 32,
// target = v1_low | (high >> 16) << 32 (crates/builder/src/input/graph500.rs:111-127);
// node_count = edge_count / 16 (:74)
struct Graph500Input {
    EdgeData read(const std::string &path) const
    {
        const std::vector<char> buf = detail::read_file(path);
        if (buf.size() % 12)
            throw Error(GM_ERR_INVALID, path + ": size is not a multiple of 12 bytes");
        EdgeData out;
        const size_t m = buf.size() / 12;
        out.src.resize(m);
        out.dst.resize(m);
        const unsigned char *b = reinterpret_cast<const unsigned char *>(buf.data());
        auto u32le = [](const unsigned char *x) {
            return (uint64_t)x[0] | (uint64_t)x[1] << 8 | (uint64_t)x[2] << 16 | (uint64_t)x[3] << 24;
        };
        for (size_t i = 0; i < m; ++i) {
            const uint64_t lo0 = u32le(b + 12 * i), lo1 = u32le(b + 12 * i + 4), hi = u32le(b + 12 * i + 8);
            out.src[i] = lo0 | (hi & 0xFFFFull) << 32;
            out.dst[i] = lo1 | (hi >> 16) << 32;
        }
        out.node_count = m / 16;
        return out;
    }
};

// GraphBuilder::new().csr_layout(..).edges(..) | .file_format(..).path(..) -> build()
// (crates/builder/src/builder.rs:123-540)
class GraphBuilder {
public:
    GraphBuilder &csr_layout(CsrLayout l) { layout_ = l; return *this; }
    GraphBuilder &device(int d) { device_ = d; return *this; }
    GraphBuilder &edges(const std::vector<std::pair<uint64_t, uint64_t>> &e)
    {
        src_.clear(); dst_.clear(); w_.clear(); weighted_ = false;
        for (auto &p : e) { push(p.first, p.second); }
        return *this;
    }
    GraphBuilder &edges_with_values(const std::vector<std::tuple<uint64_t, uint64_t, float>> &e)
    {
        src_.clear(); dst_.clear(); w_.clear(); weighted_ = true;
        for (auto &t : e) { push(std::get<0>(t), std::get<1>(t)); w_.push_back(std::get<2>(t)); }
        return *this;
    }
    // .file_format(EdgeListInput{..} | Graph500Input{}).path("...")   (builder.rs:330-420)
    template <class Format> GraphBuilder &file_format(const Format &f)
    {
        reader_ = [f](const std::string &path) { return f.read(path); };
        return *this;
    }
    GraphBuilder &path(const std::string &file)
    {
        if (!reader_)
            throw Error(GM_ERR_INVALID, "GraphBuilder::path before file_format");
        const EdgeData data = reader_(file);
        src_.clear(); dst_.clear(); w_.clear();
        weighted_ = !data.values.empty();
        n_ = 0;
        for (size_t i = 0; i < data.src.size(); ++i)
            push(data.src[i], data.dst[i]);
        w_ = data.values;
        n_ = data.node_count; // the format's own rule (Graph500: edge_count / 16)
        return *this;
    }
    template <class G> G build() const { return build_impl(static_cast<G *>(nullptr)); }

private:
    void push(uint64_t s, uint64_t t)
    {
        if (s >= (1ull << 32) || t >= (1ull << 32))
            throw Error(GM_ERR_RANGE, "node id does not fit the u32 device id type");
        src_.push_back((uint32_t)s); dst_.push_back((uint32_t)t);
        if (s + 1 > n_) n_ = s + 1;
        if (t + 1 > n_) n_ = t + 1; // node_count = max id + 1 (csr.rs:530)
    }
    detail::CsrPtr make(int direction) const
    {
        gm_csr *c = nullptr;
        detail::check(gm_csr_build_host(n_, src_.size(), src_.data(), dst_.data(), weighted_ ? w_.data() : nullptr, direction,
                                        (int)layout_, device_, &c));
        return detail::CsrPtr(c);
    }
    template <class NI> DirectedCsrGraph<NI> build_impl(DirectedCsrGraph<NI> *) const
    {
        return DirectedCsrGraph<NI>(make(GM_DIR_OUTGOING), make(GM_DIR_INCOMING), layout_);
    }
    template <class NI> UndirectedCsrGraph<NI> build_impl(UndirectedCsrGraph<NI> *) const
    {
        return UndirectedCsrGraph<NI>(make(GM_DIR_UNDIRECTED), layout_);
    }
    CsrLayout layout_ = CsrLayout::Unsorted;
    int device_ = 0;
    uint64_t n_ = 0;
    bool weighted_ = false;
    std::vector<uint32_t> src_, dst_;
    std::vector<float> w_;
    std::function<EdgeData(const std::string &)> reader_;
};

// ---- algorithms ------------------------------------------------------------------------------------
// crates/algos/src/page_rank.rs:14-56
struct PageRankConfig {
    size_t max_iterations = 20;
    double tolerance = 1e-4;
    float damping_factor = 0.85f;
};

// page_rank(&graph, config) -> (scores, iterations, error)   crates/algos/src/page_rank.rs:58-111
template <class NI>
std::tuple<std::vector<float>, size_t, double> page_rank(const DirectedCsrGraph<NI> &g, PageRankConfig config = {},
                                                         int mode = GM_PR_AUTO)
{
    const uint64_t n = gm_csr_node_count(g.csr_inc());
    std::vector<float> scores(n);
    uint64_t iterations = 0;
    double error = 0.0;
    // both CSRs are resident: out-degrees come from the out-CSR's offsets on the device
    detail::check(gm_page_rank_directed(g.csr_out(), g.csr_inc(), config.max_iterations, config.tolerance,
                                        config.damping_factor, mode, scores.data(), &iterations, &error));
    return {std::move(scores), (size_t)iterations, error};
}

// The same call on a graph split over `devices` GPUs of this node (gm_page_rank_multi: 1-D in-degree-balanced row
// ranges, RCCL all-gather of out_scores per sweep, one host thread).  An ordinal listed twice = virtual ranks.
template <class NI>
std::tuple<std::vector<float>, size_t, double> page_rank_multi(const DirectedCsrGraph<NI> &g, const std::vector<int> &devices,
                                                               PageRankConfig config = {})
{
    const uint64_t n = gm_csr_node_count(g.csr_inc());
    std::vector<float> scores(n);
    uint64_t iterations = 0;
    double error = 0.0;
    detail::check(gm_page_rank_multi(g.csr_out(), g.csr_inc(), devices.data(), (uint32_t)devices.size(), config.max_iterations,
                                     config.tolerance, config.damping_factor, scores.data(), &iterations, &error));
    return {std::move(scores), (size_t)iterations, error};
}

// crates/algos/src/wcc.rs:43-79 (chunk_size is a CPU scheduling knob)
struct WccConfig {
    size_t chunk_size = 16384, neighbor_rounds = 2, sampling_size = 1024;
};

// Components<NI> (wcc.rs:95-99)
template <class NI> class Components {
public:
    explicit Components(std::vector<uint32_t> labels) : labels_(std::move(labels)) {}
    NI component(NI node) const { return (NI)labels_.at(node); }
    std::vector<uint32_t> to_vec() && { return std::move(labels_); }
    const std::vector<uint32_t> &labels() const { return labels_; }

private:
    std::vector<uint32_t> labels_;
};

template <class NI> Components<NI> wcc_afforest(const DirectedCsrGraph<NI> &g, WccConfig config = {})
{
    std::vector<uint32_t> labels(gm_csr_node_count(g.csr_out()));
    detail::check(gm_wcc_afforest(g.csr_out(), g.csr_inc(), config.neighbor_rounds, config.sampling_size, labels.data()));
    return Components<NI>(std::move(labels));
}
template <class NI> Components<NI> wcc_afforest_dss(const DirectedCsrGraph<NI> &g, WccConfig config = {})
{
    return wcc_afforest(g, config); // same component(u) (the root = minimum id); the backend is a CPU detail
}
template <class NI> Components<NI> wcc_baseline(const DirectedCsrGraph<NI> &g, WccConfig = {})
{
    std::vector<uint32_t> labels(gm_csr_node_count(g.csr_out()));
    detail::check(gm_wcc_baseline(g.csr_out(), labels.data()));
    return Components<NI>(std::move(labels));
}

// crates/algos/src/sssp.rs:18-36
struct DeltaSteppingConfig {
    size_t start_node;
    float delta;
};
// delta_stepping -> distances, f32::MAX = unreachable   crates/algos/src/sssp.rs:38-102
template <class NI> std::vector<float> delta_stepping(const DirectedCsrGraph<NI> &g, DeltaSteppingConfig config)
{
    std::vector<float> dist(gm_csr_node_count(g.csr_out()));
    detail::check(gm_sssp_delta_stepping(g.csr_out(), config.start_node, config.delta, dist.data()));
    return dist;
}

// crates/algos/src/triangle_count.rs:12-86
template <class NI> uint64_t global_triangle_count(const UndirectedCsrGraph<NI> &g)
{
    uint64_t t = 0;
    detail::check(gm_triangle_count(g.csr(), &t));
    return t;
}
template <class NI> void relabel_graph(UndirectedCsrGraph<NI> &g) { g.make_degree_ordered(); }

namespace prelude {
using graph::Components;
using graph::CsrLayout;
using graph::delta_stepping;
using graph::DeltaSteppingConfig;
using graph::DirectedCsrGraph;
using graph::EdgeListInput;
using graph::global_triangle_count;
using graph::Graph500Input;
using graph::GraphBuilder;
using graph::page_rank;
using graph::page_rank_multi;
using graph::PageRankConfig;
using graph::relabel_graph;
using graph::UndirectedCsrGraph;
using graph::wcc_afforest;
using graph::wcc_afforest_dss;
using graph::wcc_baseline;
using graph::WccConfig;
} // namespace prelude

} // namespace graph
