//! Drop-in for the hot path of `graph::prelude` (crates/algos/src/prelude.rs:1-7) on MI355X.
//!
//! NOT COMPILED HERE: the build image has no Rust toolchain.  This file is the binding a
//! maintainer adds; every `extern "C"` item is declared in include/graph_mi355x.h.
//!
//! The CSR arrays of the upstream `DirectedCsrGraph<u32>` are reachable from outside the crate
//! because the neighbour lists are laid out back to back (csr.rs:87-92,146-172):
//! `g.in_neighbors(0).as_slice().as_ptr()` is the base of `targets`, offsets are the prefix sum
//! of the degrees.  Inside the upstream workspace one would read `Csr::offsets/targets` directly.
use std::ffi::{c_char, c_int, c_void, CStr};

use atomic_float::AtomicF32;
use graph_builder::prelude::*;

#[repr(C)]
pub struct GmCsr {
    _private: [u8; 0],
}

extern "C" {
    fn gm_last_error() -> *const c_char;
    fn gm_csr_upload_u32(offsets: *const u32, targets: *const u32, weights: *const f32, n: u64, m: u64,
                         device: c_int, out: *mut *mut GmCsr) -> c_int;
    fn gm_csr_free(csr: *mut GmCsr);
    fn gm_page_rank(in_csr: *const GmCsr, out_degree: *const u32, max_iterations: u64, tolerance: f64,
                    damping_factor: f32, mode: c_int, scores_out: *mut f32, iterations_out: *mut u64,
                    error_out: *mut f64) -> c_int;
    fn gm_page_rank_directed(out_csr: *const GmCsr, in_csr: *const GmCsr, max_iterations: u64, tolerance: f64,
                             damping_factor: f32, mode: c_int, scores_out: *mut f32, iterations_out: *mut u64,
                             error_out: *mut f64) -> c_int;
    fn gm_wcc_afforest(out_csr: *const GmCsr, in_csr: *const GmCsr, neighbor_rounds: u64, sampling_size: u64,
                       components_out: *mut u32) -> c_int;
    fn gm_sssp_delta_stepping(out_csr: *const GmCsr, start_node: u64, delta: f32, distances_out: *mut f32) -> c_int;
    fn gm_triangle_count(undirected_csr: *const GmCsr, triangles_out: *mut u64) -> c_int;
}

fn check(status: c_int) {
    if status != 0 {
        // the reference's functions are infallible and panic on bad input; keep that contract
        let msg = unsafe { CStr::from_ptr(gm_last_error()) }.to_string_lossy().into_owned();
        panic!("graph_mi355x status {status}: {msg}");
    }
}

/// One device-resident CSR, uploaded once and cached by the caller next to the graph.
pub struct DeviceCsr(*mut GmCsr);
unsafe impl Send for DeviceCsr {}
unsafe impl Sync for DeviceCsr {}
impl Drop for DeviceCsr {
    fn drop(&mut self) {
        unsafe { gm_csr_free(self.0) }
    }
}

impl DeviceCsr {
    /// `degree(u)` and `neighbors(u)` are the graph's accessors for one direction.
    pub fn upload(n: u32, degree: impl Fn(u32) -> u32, first_list: *const u32) -> Self {
        let mut offsets = Vec::with_capacity(n as usize + 1);
        let mut acc = 0u32;
        offsets.push(0);
        for u in 0..n {
            acc += degree(u);
            offsets.push(acc);
        }
        let mut out = std::ptr::null_mut();
        check(unsafe { gm_csr_upload_u32(offsets.as_ptr(), first_list, std::ptr::null(), n as u64, acc as u64, 0, &mut out) });
        DeviceCsr(out)
    }
}

/// Device mirror of a `DirectedCsrGraph<u32>` (build once, reuse for every algorithm call).
pub struct DeviceDirected {
    pub out: DeviceCsr,
    pub inc: DeviceCsr,
    pub node_count: u32,
}

impl DeviceDirected {
    pub fn new(g: &DirectedCsrGraph<u32>) -> Self {
        let n = g.node_count();
        let out = DeviceCsr::upload(n, |u| g.out_degree(u), g.out_neighbors(0).as_slice().as_ptr());
        let inc = DeviceCsr::upload(n, |u| g.in_degree(u), g.in_neighbors(0).as_slice().as_ptr());
        Self { out, inc, node_count: n }
    }
}

/// `page_rank(&graph, config) -> (Vec<f32>, usize, f64)` — crates/algos/src/page_rank.rs:58-62
pub fn page_rank(g: &DeviceDirected, max_iterations: usize, tolerance: f64, damping_factor: f32) -> (Vec<f32>, usize, f64) {
    let mut scores = vec![0f32; g.node_count as usize];
    let (mut iterations, mut error) = (0u64, 0f64);
    check(unsafe {
        // both CSRs are resident: out-degrees come from the out-CSR's offsets on the device
        gm_page_rank_directed(g.out.0, g.inc.0, max_iterations as u64, tolerance, damping_factor, 0,
                              scores.as_mut_ptr(), &mut iterations, &mut error)
    });
    (scores, iterations as usize, error)
}

/// `wcc_afforest(&graph, config) -> impl Components<u32>` — crates/algos/src/wcc.rs:127-141
pub fn wcc_afforest(g: &DeviceDirected, neighbor_rounds: usize, sampling_size: usize) -> Vec<u32> {
    let mut comp = vec![0u32; g.node_count as usize];
    check(unsafe { gm_wcc_afforest(g.out.0, g.inc.0, neighbor_rounds as u64, sampling_size as u64, comp.as_mut_ptr()) });
    comp
}

/// `delta_stepping(&graph, config) -> Vec<AtomicF32>` — crates/algos/src/sssp.rs:38-42
/// (`out_weighted` uploaded with the f32 values of `out_neighbors_with_values`).
pub fn delta_stepping(out_weighted: &DeviceCsr, node_count: usize, start_node: usize, delta: f32) -> Vec<AtomicF32> {
    let mut dist = vec![0f32; node_count];
    check(unsafe { gm_sssp_delta_stepping(out_weighted.0, start_node as u64, delta, dist.as_mut_ptr()) });
    dist.into_iter().map(AtomicF32::new).collect()
}

/// `global_triangle_count(&graph) -> u64` — crates/algos/src/triangle_count.rs:22-26
pub fn global_triangle_count(undirected: &DeviceCsr) -> u64 {
    let mut t = 0u64;
    check(unsafe { gm_triangle_count(undirected.0, &mut t) });
    t
}

#[allow(dead_code)]
fn _unused(_: *mut c_void) {}
