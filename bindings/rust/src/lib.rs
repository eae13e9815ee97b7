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
//! How the graph reaches the GPU.  The traits only promise iterators (`NeighborsIterator<'a>: Iterator<Item =
//! &'a NI>`, crates/builder/src/lib.rs:336-412), so the generic path walks them once — O(n + m) on the host —
//! into u32 (NI up to 32 bits) or u64 (`u64` / `usize`, narrowed with a range check by `gm_csr_upload_u64`)
//! arrays and splits `Target<NI, f32>` (AoS, crates/builder/src/graph/mod.rs:5-10) into targets + weights.  The
//! CSR graph types store their targets as ONE slice in node order, which their `std::slice::Iter` hands out
//! (`out_neighbors(0).as_slice().as_ptr()`): for them only the n + 1 offsets are rebuilt from the degrees and the
//! target array is uploaded from where it lies (`Residency::out_targets` ...).
//!
//! Who owns the device copy.  NOT a process-wide cache keyed by the graph's address (a graph dropped and rebuilt at
//! the same address with the same counts would silently meet its predecessor's data, and nothing would ever be
//! evicted).  A plain graph keeps no copy: every call uploads, computes and frees — always correct.  `OnDevice<G>`
//! wraps a graph together with its copies ("offsets/targets uploaded once to HBM"): it hands out `&G` only, the
//! copies are dropped the moment `get_mut()` is taken (the one door to in-place changes such as
//! `make_degree_ordered`) and when the wrapper is dropped.  The algorithms take either: `page_rank(&graph, cfg)` or
//! `page_rank(&on_device, cfg)`.
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
        Components, DeltaSteppingConfig, OnDevice, PageRankConfig, Residency, WccConfig,
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

/// One neighbour list walked into flat arrays; `wide` = ids do not fit u32 by type (u64 / usize / i64 ...).
fn upload<NI: Idx>(node_count: usize, lists: impl Fn(NI, &mut dyn FnMut(NI, Option<f32>))) -> DeviceCsr {
    let wide = std::mem::size_of::<NI>() > 4;
    let mut weights: Vec<f32> = Vec::new();
    let mut weighted = false;
    let mut out = std::ptr::null_mut();
    if wide {
        let (mut off, mut tgt) = (Vec::<u64>::with_capacity(node_count + 1), Vec::<u64>::new());
        off.push(0);
        for u in 0..node_count {
            lists(NI::new(u), &mut |v, w| {
                tgt.push(v.index() as u64);
                if let Some(w) = w {
                    weighted = true;
                    weights.push(w);
                }
            });
            off.push(tgt.len() as u64); // u64: no overflow; gm_csr_upload_u64 rejects n or m >= 2^32 (GM_ERR_RANGE)
        }
        let wp = if weighted { weights.as_ptr() } else { std::ptr::null() };
        check(unsafe { gm_csr_upload_u64(off.as_ptr(), tgt.as_ptr(), wp, node_count as u64, tgt.len() as u64, 0, &mut out) });
    } else {
        let (mut off, mut tgt) = (Vec::<u32>::with_capacity(node_count + 1), Vec::<u32>::new());
        off.push(0);
        for u in 0..node_count {
            lists(NI::new(u), &mut |v, w| {
                tgt.push(v.index() as u32);
                if let Some(w) = w {
                    weighted = true;
                    weights.push(w);
                }
            });
            assert!(tgt.len() < u32::MAX as usize, "more than 2^32 - 1 target entries: beyond the device id type");
            off.push(tgt.len() as u32);
        }
        let wp = if weighted { weights.as_ptr() } else { std::ptr::null() };
        check(unsafe { gm_csr_upload_u32(off.as_ptr(), tgt.as_ptr(), wp, node_count as u64, tgt.len() as u64, 0, &mut out) });
    }
    DeviceCsr(out)
}

// ------------------------------------------------------------------------------------------------
// who keeps the device copies: nobody (plain graphs) or the OnDevice wrapper
// ------------------------------------------------------------------------------------------------
#[derive(Clone, Copy, PartialEq, Eq, Hash)]
pub enum Kind {
    Directed,         // out + in lists
    DirectedWeighted, // out lists with f32 values
    OutOnly,          // wcc_baseline needs nothing else
    Undirected,
}

pub struct Resident {
    out: Option<DeviceCsr>,
    inc: Option<DeviceCsr>,
}

/// The device copies one graph keeps (at most one per `Kind`).
#[derive(Default)]
pub struct Copies(Mutex<HashMap<Kind, Arc<Resident>>>);

/// What the algorithms ask of a graph besides the reference's traits.  Implemented for the reference's CSR graph
/// types (no copies kept, contiguous target arrays exposed) and for `OnDevice<G>` (copies kept).
pub trait Residency<NI: Idx> {
    /// Where this graph keeps its device copies; `None`: nowhere, every call uploads and frees its own.
    fn copies(&self) -> Option<&Copies> {
        None
    }
    /// The whole target array of the out / in / undirected lists, when the graph stores it as one slice in node
    /// order (CSR): uploaded from where it lies instead of walked.
    fn out_targets(&self) -> Option<&[NI]> {
        None
    }
    fn in_targets(&self) -> Option<&[NI]> {
        None
    }
    fn undirected_targets(&self) -> Option<&[NI]> {
        None
    }
}

/// `first` is node 0's list of a CSR whose lists lie back to back in one allocation of `len` entries.
unsafe fn whole<NI>(first: &[NI], len: usize) -> &[NI] {
    std::slice::from_raw_parts(first.as_ptr(), len)
}

impl<NI: Idx, NV> Residency<NI> for DirectedCsrGraph<NI, NV, ()> {
    fn out_targets(&self) -> Option<&[NI]> {
        (self.node_count().index() > 0)
            .then(|| unsafe { whole(self.out_neighbors(NI::zero()).as_slice(), self.edge_count().index()) })
    }
    fn in_targets(&self) -> Option<&[NI]> {
        (self.node_count().index() > 0)
            .then(|| unsafe { whole(self.in_neighbors(NI::zero()).as_slice(), self.edge_count().index()) })
    }
}

impl<NI: Idx, NV> Residency<NI> for DirectedCsrGraph<NI, NV, f32> {} // Target<NI, f32> records: walked and split

impl<NI: Idx, NV> Residency<NI> for UndirectedCsrGraph<NI, NV, ()> {
    fn undirected_targets(&self) -> Option<&[NI]> {
        // Graph::edge_count() of an undirected CSR is half its target entries (csr.rs:687-689)
        (self.node_count().index() > 0)
            .then(|| unsafe { whole(self.neighbors(NI::zero()).as_slice(), 2 * self.edge_count().index()) })
    }
}

/// A graph together with its copies in HBM.  `&OnDevice<G>` goes wherever `&G` goes (Deref + the reference's traits
/// delegated below); the copies die with the wrapper or when `get_mut()` opens the graph for changes.
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

impl<NI: Idx, G: Residency<NI>> Residency<NI> for OnDevice<G> {
    fn copies(&self) -> Option<&Copies> {
        Some(&self.copies)
    }
    fn out_targets(&self) -> Option<&[NI]> {
        self.graph.out_targets()
    }
    fn in_targets(&self) -> Option<&[NI]> {
        self.graph.in_targets()
    }
    fn undirected_targets(&self) -> Option<&[NI]> {
        self.graph.undirected_targets()
    }
}

impl<NI: Idx, G: Graph<NI>> Graph<NI> for OnDevice<G> {
    fn node_count(&self) -> NI {
        self.graph.node_count()
    }
    fn edge_count(&self) -> NI {
        self.graph.edge_count()
    }
}

impl<NI: Idx, G: DirectedDegrees<NI>> DirectedDegrees<NI> for OnDevice<G> {
    fn out_degree(&self, node: NI) -> NI {
        self.graph.out_degree(node)
    }
    fn in_degree(&self, node: NI) -> NI {
        self.graph.in_degree(node)
    }
}

impl<NI: Idx, G: UndirectedDegrees<NI>> UndirectedDegrees<NI> for OnDevice<G> {
    fn degree(&self, node: NI) -> NI {
        self.graph.degree(node)
    }
}

impl<NI: Idx, G: DirectedNeighbors<NI>> DirectedNeighbors<NI> for OnDevice<G> {
    type NeighborsIterator<'a>
        = G::NeighborsIterator<'a>
    where
        Self: 'a;
    fn out_neighbors(&self, node: NI) -> Self::NeighborsIterator<'_> {
        self.graph.out_neighbors(node)
    }
    fn in_neighbors(&self, node: NI) -> Self::NeighborsIterator<'_> {
        self.graph.in_neighbors(node)
    }
}

impl<NI: Idx, G: DirectedNeighborsWithValues<NI, f32>> DirectedNeighborsWithValues<NI, f32> for OnDevice<G> {
    type NeighborsIterator<'a>
        = G::NeighborsIterator<'a>
    where
        Self: 'a;
    fn out_neighbors_with_values(&self, node: NI) -> Self::NeighborsIterator<'_> {
        self.graph.out_neighbors_with_values(node)
    }
    fn in_neighbors_with_values(&self, node: NI) -> Self::NeighborsIterator<'_> {
        self.graph.in_neighbors_with_values(node)
    }
}

impl<NI: Idx, G: UndirectedNeighbors<NI>> UndirectedNeighbors<NI> for OnDevice<G> {
    type NeighborsIterator<'a>
        = G::NeighborsIterator<'a>
    where
        Self: 'a;
    fn neighbors(&self, node: NI) -> Self::NeighborsIterator<'_> {
        self.graph.neighbors(node)
    }
}

fn resident<NI: Idx, G: Residency<NI>>(graph: &G, kind: Kind, build: impl FnOnce() -> Resident) -> Arc<Resident> {
    let Some(copies) = graph.copies() else {
        return Arc::new(build()); // a plain graph: this call's own copy, freed when the call returns
    };
    if let Some(r) = copies.0.lock().unwrap().get(&kind) {
        return r.clone();
    }
    let r = Arc::new(build()); // outside the lock: the four kinds of one graph may be uploaded side by side
    copies.0.lock().unwrap().entry(kind).or_insert(r).clone()
}

/// One CSR from its degrees and its target array where it lies (4- or 8-byte ids; `None` for other widths).
fn upload_contiguous<NI: Idx>(node_count: usize, degree: impl Fn(NI) -> usize, targets: &[NI]) -> Option<DeviceCsr> {
    let mut out = std::ptr::null_mut();
    match std::mem::size_of::<NI>() {
        4 => {
            assert!(targets.len() < u32::MAX as usize, "more than 2^32 - 1 target entries: beyond the device id type");
            let mut off = Vec::<u32>::with_capacity(node_count + 1);
            let mut at = 0u32;
            off.push(0);
            for u in 0..node_count {
                at += degree(NI::new(u)) as u32;
                off.push(at);
            }
            assert_eq!(at as usize, targets.len(), "degrees do not add up to the target array");
            check(unsafe {
                gm_csr_upload_u32(off.as_ptr(), targets.as_ptr() as *const u32, std::ptr::null(), node_count as u64,
                                  targets.len() as u64, 0, &mut out)
            });
        }
        8 => {
            let mut off = Vec::<u64>::with_capacity(node_count + 1);
            let mut at = 0u64;
            off.push(0);
            for u in 0..node_count {
                at += degree(NI::new(u)) as u64;
                off.push(at);
            }
            assert_eq!(at as usize, targets.len(), "degrees do not add up to the target array");
            check(unsafe {
                gm_csr_upload_u64(off.as_ptr(), targets.as_ptr() as *const u64, std::ptr::null(), node_count as u64,
                                  targets.len() as u64, 0, &mut out)
            });
        }
        _ => return None,
    }
    Some(DeviceCsr(out))
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

fn directed<NI, G>(graph: &G) -> Arc<Resident>
where
    NI: Idx,
    G: Graph<NI> + DirectedDegrees<NI> + DirectedNeighbors<NI> + Residency<NI> + Sync,
{
    let n = graph.node_count().index();
    resident(graph, Kind::Directed, || Resident {
        out: Some(
            graph
                .out_targets()
                .and_then(|t| upload_contiguous::<NI>(n, |u| graph.out_degree(u).index(), t))
                .unwrap_or_else(|| upload::<NI>(n, |u, push| graph.out_neighbors(u).for_each(|v| push(*v, None)))),
        ),
        inc: Some(
            graph
                .in_targets()
                .and_then(|t| upload_contiguous::<NI>(n, |u| graph.in_degree(u).index(), t))
                .unwrap_or_else(|| upload::<NI>(n, |u, push| graph.in_neighbors(u).for_each(|v| push(*v, None)))),
        ),
    })
}

/// The reference's signature plus `Residency` (its CSR graph types and `OnDevice<_>` have it).  `GM_DEVICES=k` (k > 1) runs the call 1-D partitioned over the first k
/// GPUs of the node (`gm_page_rank_multi`: RCCL all-gather of out_scores per sweep); default: one GPU.
pub fn page_rank<NI, G>(graph: &G, config: PageRankConfig) -> (Vec<f32>, usize, f64)
where
    NI: Idx,
    G: Graph<NI> + DirectedDegrees<NI> + DirectedNeighbors<NI> + Residency<NI> + Sync,
{
    let PageRankConfig { max_iterations, tolerance, damping_factor } = config;
    let g = directed(graph);
    let mut scores = vec![0f32; graph.node_count().index()];
    let (mut iterations, mut error) = (0u64, 0f64);
    let (out, inc) = (g.out.as_ref().unwrap().0, g.inc.as_ref().unwrap().0);
    let devices: u32 = std::env::var("GM_DEVICES").ok().and_then(|v| v.parse().ok()).unwrap_or(1);
    check(unsafe {
        if devices > 1 {
            gm_page_rank_multi(out, inc, std::ptr::null(), devices, max_iterations as u64, tolerance, damping_factor,
                               scores.as_mut_ptr(), &mut iterations, &mut error)
        } else {
            // mode 0 = GM_PR_AUTO: n <= 16384 runs the reference's exact in-place order, else synchronous sweeps
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
    G: Graph<NI> + DirectedDegrees<NI> + DirectedNeighbors<NI> + Residency<NI> + Sync,
{
    let g = directed(graph);
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
    G: Graph<NI> + DirectedDegrees<NI> + DirectedNeighbors<NI> + Residency<NI> + Sync,
{
    wcc_afforest(graph, config)
}

pub fn wcc_baseline<NI, G>(graph: &G, _config: WccConfig) -> impl Components<NI>
where
    NI: Idx,
    G: Graph<NI> + DirectedNeighbors<NI> + Residency<NI> + Sync,
{
    let n = graph.node_count().index();
    let g = resident(graph, Kind::OutOnly, || Resident {
        // no degrees in this function's bounds: the lists are walked (wcc_afforest's copy has the fast path)
        out: Some(upload::<NI>(n, |u, push| graph.out_neighbors(u).for_each(|v| push(*v, None)))),
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
    G: Graph<NI> + DirectedNeighborsWithValues<NI, f32> + Residency<NI> + Sync,
{
    let n = graph.node_count().index();
    let g = resident(graph, Kind::DirectedWeighted, || Resident {
        // Target<NI, f32> is an 8-byte AoS record on the host; the device streams targets and weights apart
        out: Some(upload::<NI>(n, |u, push| {
            graph.out_neighbors_with_values(u).for_each(|t| push(t.target, Some(t.value)))
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
    G: Graph<NI> + UndirectedNeighbors<NI> + Residency<NI> + Sync,
{
    let n = graph.node_count().index();
    let g = resident(graph, Kind::Undirected, || Resident {
        out: Some(
            graph
                .undirected_targets()
                .and_then(|t| upload_contiguous::<NI>(n, |u| graph.neighbors(u).count(), t)) // O(1) on slice iterators
                .unwrap_or_else(|| upload::<NI>(n, |u, push| graph.neighbors(u).for_each(|v| push(*v, None)))),
        ),
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

impl<G> OnDevice<G> {
    /// `relabel_graph` for a graph with device copies: they are dropped first (`get_mut`), the next algorithm call
    /// uploads the relabelled lists.
    pub fn relabel<NI: Idx, EV>(&mut self)
    where
        G: RelabelByDegreeOp<NI, EV>,
    {
        self.get_mut().make_degree_ordered();
    }
}
