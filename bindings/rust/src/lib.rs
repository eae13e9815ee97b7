//! `graph::prelude` algorithms on MI355X: the same public names, generic bounds, config structs and
//! return types as `crates/algos` (prelude.rs:1-7), with bodies that call the C ABI of
//! `include/graph_mi355x.h`.  A caller swaps `use graph::prelude::*` for `use graph_mi355x::prelude::*`.
//!
//! NOT COMPILED IN THIS REPOSITORY'S ENVIRONMENT (no rustc / cargo in the image).  The same mirror exists,
//! compiled and tested on the GPU, as C++ (`include/graph_prelude.hpp`) and Python (`graph_amd/prelude.py`).
//!
//! Signatures replaced (reference file:line):
//!   page_rank<NI, G>(&G, PageRankConfig) -> (Vec<f32>, usize, f64)        crates/algos/src/page_rank.rs:58-62
//!   wcc_afforest / wcc_afforest_dss / wcc_baseline -> impl Components<NI>  crates/algos/src/wcc.rs:103,127,144
//!   delta_stepping<NI, G>(&G, DeltaSteppingConfig) -> Vec<AtomicF32>       crates/algos/src/sssp.rs:38-42
//!   global_triangle_count<NI, G>(&G) -> u64                                crates/algos/src/triangle_count.rs:22-26
//!
//! The free functions have EXACTLY the reference's bounds: any graph type that compiles against `graph::prelude` — the
//! CSR graphs, the adjacency-list graphs (crates/builder/src/graph/adj_list.rs), a caller's own type — compiles here.
//!
//! How the graph reaches the GPU.  The traits only promise iterators (`NeighborsIterator<'a>: Iterator<Item =
//! &'a NI>`, crates/builder/src/lib.rs:336-412), so every list is flattened through them, node by node, into one
//! array: `Vec::extend(iter_of_refs)`.  For the CSR graph types that iterator is `std::slice::Iter`, for which the
//! standard library's `extend` is a `memcpy` of the node's slice (monomorphisation picks it, no trait of ours is
//! involved) — no per-element work, and no pointer is ever used outside the borrow it came from.  (Rounds 2-3 uploaded
//! the CSR's target array "from where it lies" by stretching node 0's slice over the whole array: that relies on the
//! crate-private layout of `Csr` and is undefined behaviour under strict provenance.  A one-line public accessor
//! upstream — INTEGRATION.md — would make the zero-copy path sound; this crate does not pretend to have it.)
//! `Target<NI, f32>` records (AoS, crates/builder/src/graph/mod.rs:5-10) are split into targets + weights.
//! Ids of 4 bytes go through `gm_csr_upload_u32`, wider ones through `gm_csr_upload_u64` (narrowed with a range check).
//!
//! Who owns the device copy.  NOT a process-wide cache keyed by the graph's address (a graph dropped and rebuilt at
//! the same address with the same counts would silently meet its predecessor's data, and nothing would ever be
//! evicted).  The free functions keep no copy: every call flattens, uploads, computes and frees — always correct,
//! and O(n + m) of host work + PCIe per call (a loop over `page_rank(&graph, ..)` the way the reference's app runs it
//! re-pays that and the plan build every time).  `OnDevice<G>` is the "offsets/targets uploaded once to HBM" of the
//! design: it owns a graph together with its device copies and offers the same algorithms as methods
//! (`on_device.page_rank(cfg)`, ...), with the same bounds on `G`; it hands out `&G` (Deref), the copies are dropped
//! the moment `get_mut()` is taken (the one door to in-place changes such as `make_degree_ordered`) and when the
//! wrapper is dropped.
use std::collections::HashMap;
use std::ffi::{c_char, c_int, CStr};
use std::hash::Hash;
use std::ops::Deref;
use std::sync::{Arc, Mutex};

use atomic_float::AtomicF32;
use graph_builder::prelude::*;

pub mod prelude {
    pub use super::{
        delta_stepping, global_triangle_count, page_rank, relabel_graph, wcc_afforest, wcc_afforest_dss, wcc_baseline,
        Components, DeltaSteppingConfig, OnDevice, PageRankConfig, WccConfig,
    };
    pub use graph_builder::prelude::*;
}

// ------------------------------------------------------------------------------------------------
// C ABI (include/graph_mi355x.h)
// ------------------------------------------------------------------------------------------------
#[repr(C)]
pub struct GmCsr {
    _private: [u8; 0],
}

extern "C" {
    fn gm_last_error() -> *const c_char;
    fn gm_csr_upload_u32(offsets: *const u32, targets: *const u32, weights: *const f32, n: u64, m: u64,
                         device: c_int, out: *mut *mut GmCsr) -> c_int;
    fn gm_csr_upload_u64(offsets: *const u64, targets: *const u64, weights: *const f32, n: u64, m: u64,
                         device: c_int, out: *mut *mut GmCsr) -> c_int;
    fn gm_csr_free(csr: *mut GmCsr);
    fn gm_page_rank_directed(out_csr: *const GmCsr, in_csr: *const GmCsr, max_iterations: u64, tolerance: f64,
                             damping_factor: f32, mode: c_int, scores_out: *mut f32, iterations_out: *mut u64,
                             error_out: *mut f64) -> c_int;
    fn gm_page_rank_multi(out_csr: *const GmCsr, in_csr: *const GmCsr, devices: *const c_int, n_devices: u32,
                          max_iterations: u64, tolerance: f64, damping_factor: f32, scores_out: *mut f32,
                          iterations_out: *mut u64, error_out: *mut f64) -> c_int;
    fn gm_wcc_afforest(out_csr: *const GmCsr, in_csr: *const GmCsr, neighbor_rounds: u64, sampling_size: u64,
                       components_out: *mut u32) -> c_int;
    fn gm_wcc_baseline(out_csr: *const GmCsr, components_out: *mut u32) -> c_int;
    fn gm_sssp_delta_stepping(out_csr: *const GmCsr, start_node: u64, delta: f32, distances_out: *mut f32) -> c_int;
    fn gm_triangle_count(undirected_csr: *const GmCsr, triangles_out: *mut u64) -> c_int;
}

/// The reference's functions are infallible and panic on bad input (start node out of range, empty sample
/// set, ...): the shim keeps that contract and carries the library's message into the panic.
fn check(status: c_int) {
    if status != 0 {
        let msg = unsafe { CStr::from_ptr(gm_last_error()) }.to_string_lossy().into_owned();
        panic!("graph_mi355x status {status}: {msg}");
    }
}

struct DeviceCsr(*mut GmCsr);
unsafe impl Send for DeviceCsr {}
unsafe impl Sync for DeviceCsr {} // a gm_csr is immutable after creation (header: "may be used from several host threads")
impl Drop for DeviceCsr {
    fn drop(&mut self) {
        unsafe { gm_csr_free(self.0) }
    }
}

/// The lists of nodes 0 .. node_count flattened into (offsets, targets) and uploaded.  `extend_list(u, &mut targets)`
/// appends node u's list: `targets.extend(graph.out_neighbors(u))` — a memcpy per node for slice iterators.
fn upload_lists<NI: Idx>(node_count: usize, edge_hint: usize, mut extend_list: impl FnMut(NI, &mut Vec<NI>)) -> DeviceCsr {
    let mut tgt: Vec<NI> = Vec::with_capacity(edge_hint);
    let mut off: Vec<u64> = Vec::with_capacity(node_count + 1);
    off.push(0);
    for u in 0..node_count {
        extend_list(NI::new(u), &mut tgt);
        off.push(tgt.len() as u64);
    }
    upload_flat::<NI>(node_count, &off, &tgt, None)
}

/// ... with `Target<NI, f32>` records split into targets and weights
fn upload_weighted_lists<NI: Idx>(node_count: usize, mut each: impl FnMut(NI, &mut dyn FnMut(NI, f32))) -> DeviceCsr {
    let (mut tgt, mut weights): (Vec<NI>, Vec<f32>) = (Vec::new(), Vec::new());
    let mut off: Vec<u64> = Vec::with_capacity(node_count + 1);
    off.push(0);
    for u in 0..node_count {
        each(NI::new(u), &mut |v, w| {
            tgt.push(v);
            weights.push(w);
        });
        off.push(tgt.len() as u64);
    }
    upload_flat::<NI>(node_count, &off, &tgt, Some(&weights))
}

fn upload_flat<NI: Idx>(node_count: usize, off: &[u64], tgt: &[NI], weights: Option<&[f32]>) -> DeviceCsr {
    let mut out = std::ptr::null_mut();
    let wp = weights.map_or(std::ptr::null(), |w| w.as_ptr());
    let (n, m) = (node_count as u64, tgt.len() as u64);
    match std::mem::size_of::<NI>() {
        // Idx is implemented for the primitive integers: a 4-byte NI has u32's layout, an 8-byte one u64's
        4 => {
            assert!(tgt.len() < u32::MAX as usize, "more than 2^32 - 1 target entries: beyond the device id type");
            let off32: Vec<u32> = off.iter().map(|&o| o as u32).collect();
            check(unsafe { gm_csr_upload_u32(off32.as_ptr(), tgt.as_ptr() as *const u32, wp, n, m, 0, &mut out) });
        }
        8 => {
            // gm_csr_upload_u64 rejects n or m >= 2^32 and ids beyond u32 (GM_ERR_RANGE)
            check(unsafe { gm_csr_upload_u64(off.as_ptr(), tgt.as_ptr() as *const u64, wp, n, m, 0, &mut out) });
        }
        _ => {
            let wide: Vec<u64> = tgt.iter().map(|v| v.index() as u64).collect();
            check(unsafe { gm_csr_upload_u64(off.as_ptr(), wide.as_ptr(), wp, n, m, 0, &mut out) });
        }
    }
    DeviceCsr(out)
}

// ------------------------------------------------------------------------------------------------
// who keeps the device copies: nobody (plain graphs) or the OnDevice wrapper
// ------------------------------------------------------------------------------------------------
#[derive(Clone, Copy, PartialEq, Eq, Hash)]
enum Kind {
    Directed,         // out + in lists
    DirectedWeighted, // out lists with f32 values
    OutOnly,          // wcc_baseline needs nothing else
    Undirected,
}

struct Resident {
    out: Option<DeviceCsr>,
    inc: Option<DeviceCsr>,
}

/// The device copies one graph keeps (at most one per `Kind`).
#[derive(Default)]
struct Copies(Mutex<HashMap<Kind, Arc<Resident>>>);

/// A graph together with its copies in HBM: the algorithms as methods (below, next to their free functions), `&G` through
/// Deref; the copies die with the wrapper or when `get_mut()` opens the graph for changes.
pub struct OnDevice<G> {
    graph: G,
    copies: Copies,
}

impl<G> OnDevice<G> {
    pub fn new(graph: G) -> Self {
        Self { graph, copies: Copies::default() }
    }
    /// The one way to `&mut G`: whatever is done through it, no device copy of the old content survives it.
    pub fn get_mut(&mut self) -> &mut G {
        self.copies.0.lock().unwrap().clear();
        &mut self.graph
    }
    pub fn into_inner(self) -> G {
        self.graph
    }
}

impl<G> Deref for OnDevice<G> {
    type Target = G;
    fn deref(&self) -> &G {
        &self.graph
    }
}

/// The copy of `kind`: from `copies` (an `OnDevice` graph: built once, kept), or this call's own, freed when it returns.
fn resident(copies: Option<&Copies>, kind: Kind, build: impl FnOnce() -> Resident) -> Arc<Resident> {
    let Some(copies) = copies else {
        return Arc::new(build());
    };
    if let Some(r) = copies.0.lock().unwrap().get(&kind) {
        return r.clone();
    }
    let r = Arc::new(build()); // outside the lock: the four kinds of one graph may be uploaded side by side
    copies.0.lock().unwrap().entry(kind).or_insert(r).clone()
}

// ------------------------------------------------------------------------------------------------
// PageRank — crates/algos/src/page_rank.rs:14-111
// ------------------------------------------------------------------------------------------------
#[derive(Copy, Clone, Debug)]
pub struct PageRankConfig {
    pub max_iterations: usize,
    pub tolerance: f64,
    pub damping_factor: f32,
}

impl PageRankConfig {
    pub const DEFAULT_MAX_ITERATIONS: usize = 20;
    pub const DEFAULT_TOLERANCE: f64 = 1E-4;
    pub const DEFAULT_DAMPING_FACTOR: f32 = 0.85;

    pub fn new(max_iterations: usize, tolerance: f64, damping_factor: f32) -> Self {
        Self { max_iterations, tolerance, damping_factor }
    }
}

impl Default for PageRankConfig {
    fn default() -> Self {
        Self::new(Self::DEFAULT_MAX_ITERATIONS, Self::DEFAULT_TOLERANCE, Self::DEFAULT_DAMPING_FACTOR)
    }
}

fn directed<NI, G>(graph: &G, copies: Option<&Copies>) -> Arc<Resident>
where
    NI: Idx,
    G: Graph<NI> + DirectedDegrees<NI> + DirectedNeighbors<NI> + Sync,
{
    let (n, m) = (graph.node_count().index(), graph.edge_count().index());
    resident(copies, Kind::Directed, || Resident {
        out: Some(upload_lists::<NI>(n, m, |u, tgt| tgt.extend(graph.out_neighbors(u)))),
        inc: Some(upload_lists::<NI>(n, m, |u, tgt| tgt.extend(graph.in_neighbors(u)))),
    })
}

/// crates/algos/src/page_rank.rs:58-62, the same bounds.  `GM_DEVICES=k` (k > 1) runs the call 1-D partitioned over the
/// first k GPUs of the node (`gm_page_rank_multi`: RCCL all-gather of out_scores per sweep); default: one GPU.
pub fn page_rank<NI, G>(graph: &G, config: PageRankConfig) -> (Vec<f32>, usize, f64)
where
    NI: Idx,
    G: Graph<NI> + DirectedDegrees<NI> + DirectedNeighbors<NI> + Sync,
{
    page_rank_on(graph, None, config)
}

fn page_rank_on<NI, G>(graph: &G, copies: Option<&Copies>, config: PageRankConfig) -> (Vec<f32>, usize, f64)
where
    NI: Idx,
    G: Graph<NI> + DirectedDegrees<NI> + DirectedNeighbors<NI> + Sync,
{
    let PageRankConfig { max_iterations, tolerance, damping_factor } = config;
    let g = directed(graph, copies);
    let mut scores = vec![0f32; graph.node_count().index()];
    let (mut iterations, mut error) = (0u64, 0f64);
    let (out, inc) = (g.out.as_ref().unwrap().0, g.inc.as_ref().unwrap().0);
    let devices: u32 = std::env::var("GM_DEVICES").ok().and_then(|v| v.parse().ok()).unwrap_or(1);
    check(unsafe {
        if devices > 1 {
            gm_page_rank_multi(out, inc, std::ptr::null(), devices, max_iterations as u64, tolerance, damping_factor,
                               scores.as_mut_ptr(), &mut iterations, &mut error)
        } else {
            // mode 0 = GM_PR_AUTO: n <= 16384 runs the reference's exact in-place order; beyond that block-Gauss-Seidel sweeps on the
            // propagation-blocking engine (the in-place update of page_rank.rs:142-160 at block granularity), synchronous ones on the pull engine
            gm_page_rank_directed(out, inc, max_iterations as u64, tolerance, damping_factor, 0, scores.as_mut_ptr(),
                                  &mut iterations, &mut error)
        }
    });
    (scores, iterations as usize, error)
}

// ------------------------------------------------------------------------------------------------
// WCC — crates/algos/src/wcc.rs:43-156
// ------------------------------------------------------------------------------------------------
#[derive(Copy, Clone, Debug)]
pub struct WccConfig {
    pub chunk_size: usize, // a CPU scheduling knob (rayon chunks): no meaning on the device
    pub neighbor_rounds: usize,
    pub sampling_size: usize,
}

impl WccConfig {
    pub const DEFAULT_CHUNK_SIZE: usize = 16384;
    pub const DEFAULT_NEIGHBOR_ROUNDS: usize = 2;
    pub const DEFAULT_SAMPLING_SIZE: usize = 1024;

    pub fn new(chunk_size: usize, neighbor_rounds: usize, sampling_size: usize) -> Self {
        Self { chunk_size, neighbor_rounds, sampling_size }
    }
}

impl Default for WccConfig {
    fn default() -> Self {
        Self::new(Self::DEFAULT_CHUNK_SIZE, Self::DEFAULT_NEIGHBOR_ROUNDS, Self::DEFAULT_SAMPLING_SIZE)
    }
}

pub trait Components<NI> {
    fn component(&self, node: NI) -> NI;

    fn to_vec(self) -> Vec<NI>;
}

/// component(u) = minimum node id of u's weakly connected component — what `Afforest::find` /
/// `DisjointSetStruct::find` return after the final compress (afforest.rs:22-56, dss.rs:38-116).
struct DeviceComponents(Vec<u32>);

impl<NI: Idx> Components<NI> for DeviceComponents {
    fn component(&self, node: NI) -> NI {
        NI::new(self.0[node.index()] as usize)
    }

    fn to_vec(self) -> Vec<NI> {
        self.0.into_iter().map(|c| NI::new(c as usize)).collect()
    }
}

pub fn wcc_afforest<NI, G>(graph: &G, config: WccConfig) -> impl Components<NI>
where
    NI: Idx + Hash,
    G: Graph<NI> + DirectedDegrees<NI> + DirectedNeighbors<NI> + Sync,
{
    wcc_afforest_on(graph, None, config)
}

fn wcc_afforest_on<NI, G>(graph: &G, copies: Option<&Copies>, config: WccConfig) -> DeviceComponents
where
    NI: Idx,
    G: Graph<NI> + DirectedDegrees<NI> + DirectedNeighbors<NI> + Sync,
{
    let g = directed(graph, copies);
    let mut comp = vec![0u32; graph.node_count().index()];
    check(unsafe {
        gm_wcc_afforest(g.out.as_ref().unwrap().0, g.inc.as_ref().unwrap().0, config.neighbor_rounds as u64,
                        config.sampling_size as u64, comp.as_mut_ptr())
    });
    DeviceComponents(comp)
}

/// The union-find backend is a CPU data-structure choice; `component()` is the same minimum id.
pub fn wcc_afforest_dss<NI, G>(graph: &G, config: WccConfig) -> impl Components<NI>
where
    NI: Idx + Hash,
    G: Graph<NI> + DirectedDegrees<NI> + DirectedNeighbors<NI> + Sync,
{
    wcc_afforest_on(graph, None, config)
}

pub fn wcc_baseline<NI, G>(graph: &G, config: WccConfig) -> impl Components<NI>
where
    NI: Idx,
    G: Graph<NI> + DirectedNeighbors<NI> + Sync,
{
    wcc_baseline_on(graph, None, config)
}

fn wcc_baseline_on<NI, G>(graph: &G, copies: Option<&Copies>, _config: WccConfig) -> DeviceComponents
where
    NI: Idx,
    G: Graph<NI> + DirectedNeighbors<NI> + Sync,
{
    let (n, m) = (graph.node_count().index(), graph.edge_count().index());
    let g = resident(copies, Kind::OutOnly, || Resident {
        out: Some(upload_lists::<NI>(n, m, |u, tgt| tgt.extend(graph.out_neighbors(u)))),
        inc: None,
    });
    let mut comp = vec![0u32; n];
    check(unsafe { gm_wcc_baseline(g.out.as_ref().unwrap().0, comp.as_mut_ptr()) });
    DeviceComponents(comp)
}

// ------------------------------------------------------------------------------------------------
// SSSP — crates/algos/src/sssp.rs:18-102
// ------------------------------------------------------------------------------------------------
#[derive(Copy, Clone, Debug)]
pub struct DeltaSteppingConfig {
    pub start_node: usize,
    pub delta: f32,
}

impl DeltaSteppingConfig {
    pub fn new(start_node: usize, delta: f32) -> Self {
        Self { start_node, delta }
    }
}

/// Unreachable nodes hold `f32::MAX` (sssp.rs:12), not infinity; an out-of-range start node panics (:52).
pub fn delta_stepping<NI, G>(graph: &G, config: DeltaSteppingConfig) -> Vec<AtomicF32>
where
    NI: Idx,
    G: Graph<NI> + DirectedNeighborsWithValues<NI, f32> + Sync,
{
    delta_stepping_on(graph, None, config)
}

fn delta_stepping_on<NI, G>(graph: &G, copies: Option<&Copies>, config: DeltaSteppingConfig) -> Vec<AtomicF32>
where
    NI: Idx,
    G: Graph<NI> + DirectedNeighborsWithValues<NI, f32> + Sync,
{
    let n = graph.node_count().index();
    let g = resident(copies, Kind::DirectedWeighted, || Resident {
        // Target<NI, f32> is an 8-byte AoS record on the host; the device streams targets and weights apart
        out: Some(upload_weighted_lists::<NI>(n, |u, push| {
            graph.out_neighbors_with_values(u).for_each(|t| push(t.target, t.value))
        })),
        inc: None,
    });
    let mut dist = vec![0f32; n];
    check(unsafe {
        gm_sssp_delta_stepping(g.out.as_ref().unwrap().0, config.start_node as u64, config.delta, dist.as_mut_ptr())
    });
    dist.into_iter().map(AtomicF32::new).collect()
}

// ------------------------------------------------------------------------------------------------
// Triangle count — crates/algos/src/triangle_count.rs:12-86
// ------------------------------------------------------------------------------------------------
/// Lists must be sorted (`CsrLayout::Sorted` / `Deduplicated`); unsorted lists are rejected (status -5 ->
/// panic) where the reference silently returns a meaningless number.
pub fn global_triangle_count<NI, G>(graph: &G) -> u64
where
    NI: Idx,
    G: Graph<NI> + UndirectedNeighbors<NI> + Sync,
{
    global_triangle_count_on(graph, None)
}

fn global_triangle_count_on<NI, G>(graph: &G, copies: Option<&Copies>) -> u64
where
    NI: Idx,
    G: Graph<NI> + UndirectedNeighbors<NI> + Sync,
{
    // Graph::edge_count() of an undirected CSR is half its target entries (csr.rs:687-689)
    let (n, m) = (graph.node_count().index(), 2 * graph.edge_count().index());
    let g = resident(copies, Kind::Undirected, || Resident {
        out: Some(upload_lists::<NI>(n, m, |u, tgt| tgt.extend(graph.neighbors(u)))),
        inc: None,
    });
    let mut triangles = 0u64;
    check(unsafe { gm_triangle_count(g.out.as_ref().unwrap().0, &mut triangles) });
    triangles
}

/// `make_degree_ordered` rewrites the host graph in place (graph_ops.rs:511-638).  A plain graph keeps no device copy,
/// so there is nothing to invalidate; for a graph that lives on the device too use `OnDevice::relabel`.
pub fn relabel_graph<NI, G, EV>(graph: &mut G)
where
    NI: Idx,
    G: RelabelByDegreeOp<NI, EV>,
{
    graph.make_degree_ordered();
}

// ------------------------------------------------------------------------------------------------
// the same algorithms on a graph that keeps its device copies: same bounds on G, same results
// ------------------------------------------------------------------------------------------------
impl<G> OnDevice<G> {
    pub fn page_rank<NI>(&self, config: PageRankConfig) -> (Vec<f32>, usize, f64)
    where
        NI: Idx,
        G: Graph<NI> + DirectedDegrees<NI> + DirectedNeighbors<NI> + Sync,
    {
        page_rank_on(&self.graph, Some(&self.copies), config)
    }

    pub fn wcc_afforest<NI>(&self, config: WccConfig) -> impl Components<NI>
    where
        NI: Idx + Hash,
        G: Graph<NI> + DirectedDegrees<NI> + DirectedNeighbors<NI> + Sync,
    {
        wcc_afforest_on(&self.graph, Some(&self.copies), config)
    }

    pub fn wcc_afforest_dss<NI>(&self, config: WccConfig) -> impl Components<NI>
    where
        NI: Idx + Hash,
        G: Graph<NI> + DirectedDegrees<NI> + DirectedNeighbors<NI> + Sync,
    {
        wcc_afforest_on(&self.graph, Some(&self.copies), config)
    }

    pub fn wcc_baseline<NI>(&self, config: WccConfig) -> impl Components<NI>
    where
        NI: Idx,
        G: Graph<NI> + DirectedNeighbors<NI> + Sync,
    {
        wcc_baseline_on(&self.graph, Some(&self.copies), config)
    }

    pub fn delta_stepping<NI>(&self, config: DeltaSteppingConfig) -> Vec<AtomicF32>
    where
        NI: Idx,
        G: Graph<NI> + DirectedNeighborsWithValues<NI, f32> + Sync,
    {
        delta_stepping_on(&self.graph, Some(&self.copies), config)
    }

    pub fn global_triangle_count<NI>(&self) -> u64
    where
        NI: Idx,
        G: Graph<NI> + UndirectedNeighbors<NI> + Sync,
    {
        global_triangle_count_on(&self.graph, Some(&self.copies))
    }

    /// `relabel_graph` for a graph with device copies: they are dropped first (`get_mut`), the next algorithm call
    /// uploads the relabelled lists.
    pub fn relabel<NI: Idx, EV>(&mut self)
    where
        G: RelabelByDegreeOp<NI, EV>,
    {
        self.get_mut().make_degree_ordered();
    }
}
