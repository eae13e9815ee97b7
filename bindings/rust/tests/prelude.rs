//! The reference's own unit tests for the hot path, restated against this crate's prelude (edge lists
//! instead of GDL strings, which are out of scope): crates/algos/src/page_rank.rs:175-197, lib.rs:92-141,
//! wcc.rs:307-329, sssp.rs:282-313, triangle_count.rs:93-130.  Needs an MI355X and libgraph_mi355x.so
//! (`GRAPH_MI355X_LIB_DIR`, see build.rs); not run in this repository's build image (no rustc).
use std::sync::atomic::Ordering;

use graph_mi355x::prelude::*;

#[test]
fn pr_two_components() {
    // "(a)-->()-->()<--(a),(b)-->()-->()<--(b)"
    let graph: DirectedCsrGraph<usize> = GraphBuilder::new()
        .csr_layout(CsrLayout::Sorted)
        .edges(vec![(0, 1), (1, 2), (0, 2), (3, 4), (4, 5), (3, 5)])
        .build();
    let (scores, _, _) = page_rank(&graph, PageRankConfig::default());
    let expected: Vec<f32> = vec![0.024999997, 0.035624996, 0.06590624, 0.024999997, 0.035624996, 0.06590624];
    assert_eq!(scores, expected);
}

#[test]
fn pr_readme_graph() {
    let graph: DirectedCsrGraph<usize> = GraphBuilder::new()
        .edges(vec![
            (1, 2), (2, 1), (4, 0), (4, 1), (5, 4), (5, 1), (5, 6), (6, 1), (6, 5), (7, 1), (7, 5), (8, 1), (8, 5),
            (9, 1), (9, 5), (10, 1), (10, 5), (11, 5), (12, 5),
        ])
        .build();
    let (ranks, iterations, _) = page_rank(&graph, PageRankConfig::new(10, 1E-4, 0.85));
    assert_eq!(iterations, 10);
    let expected = vec![
        0.024064068, 0.3145448, 0.27890152, 0.01153846, 0.029471997, 0.06329483, 0.029471997, 0.01153846, 0.01153846,
        0.01153846, 0.01153846, 0.01153846, 0.01153846,
    ];
    assert_eq!(ranks, expected);
}

#[test]
fn two_components_afforest_and_dss_and_baseline() {
    let graph: DirectedCsrGraph<usize> = GraphBuilder::new().edges(vec![(0, 1), (2, 3)]).build();
    let res = wcc_afforest_dss(&graph, WccConfig::default());
    assert_eq!(res.component(0), res.component(1));
    assert_eq!(res.component(2), res.component(3));
    assert_ne!(res.component(1), res.component(2));
    let res = wcc_afforest(&graph, WccConfig::default());
    assert_eq!(res.component(0), res.component(1));
    assert_eq!(res.component(2), res.component(3));
    assert_ne!(res.component(1), res.component(2));
    assert_eq!(wcc_baseline(&graph, WccConfig::default()).to_vec(), vec![0usize, 0, 2, 2]);
}

#[test]
fn sssp_six_nodes() {
    let graph: DirectedCsrGraph<usize, (), f32> = GraphBuilder::new()
        .csr_layout(CsrLayout::Deduplicated)
        .edges_with_values(vec![
            (0, 1, 4.0), (0, 2, 2.0), (1, 2, 5.0), (1, 3, 10.0), (2, 4, 3.0), (3, 5, 11.0), (4, 3, 4.0),
        ])
        .build();
    let actual: Vec<f32> =
        delta_stepping(&graph, DeltaSteppingConfig::new(0, 3.0)).into_iter().map(|d| d.load(Ordering::Relaxed)).collect();
    assert_eq!(actual, vec![0.0, 4.0, 2.0, 9.0, 5.0, 20.0]);
}

#[test]
fn tc_three_shapes() {
    for edges in [
        vec![(0, 1), (1, 2), (0, 2), (3, 4), (4, 5), (3, 5)], // two components
        vec![(0, 1), (1, 2), (0, 2), (0, 3), (3, 4), (0, 4)], // two triangles sharing a node
        vec![(0, 1), (1, 2), (0, 2), (1, 3), (2, 3)],         // diamond
    ] {
        let graph: UndirectedCsrGraph<u32> = GraphBuilder::new().csr_layout(CsrLayout::Deduplicated).edges(edges).build();
        assert_eq!(global_triangle_count(&graph), 2);
    }
}

#[test]
fn on_device_keeps_one_copy_and_plain_graphs_keep_none() {
    let graph: DirectedCsrGraph<u32> = GraphBuilder::new().csr_layout(CsrLayout::Sorted).edges(vec![(0, 1), (1, 2), (2, 0)]).build();
    let a = page_rank(&graph, PageRankConfig::default()); // a plain graph: uploaded for this call, freed after it
    let resident = OnDevice::new(graph);
    let b = resident.page_rank(PageRankConfig::default()); // uploads once ...
    let c = resident.page_rank(PageRankConfig::default()); // ... and runs on the same copy
    assert_eq!(a.0, b.0);
    assert_eq!(b.0, c.0);
    assert_eq!(resident.wcc_afforest(WccConfig::default()).to_vec(), vec![0u32, 0, 0]); // the same Directed copy
    // &OnDevice<G> derefs to &G: the free functions take it like any graph (and upload for the call)
    assert_eq!(page_rank(&*resident, PageRankConfig::default()).0, a.0);
}

/// A graph type of the caller's own that implements only the reference's traits (here: adjacency lists in plain Vecs):
/// the free functions have the reference's bounds, so it compiles and runs without any trait of this crate.
struct VecGraph {
    out: Vec<Vec<u32>>,
    inc: Vec<Vec<u32>>,
}

impl Graph<u32> for VecGraph {
    fn node_count(&self) -> u32 {
        self.out.len() as u32
    }
    fn edge_count(&self) -> u32 {
        self.out.iter().map(|l| l.len() as u32).sum()
    }
}

impl DirectedDegrees<u32> for VecGraph {
    fn out_degree(&self, node: u32) -> u32 {
        self.out[node as usize].len() as u32
    }
    fn in_degree(&self, node: u32) -> u32 {
        self.inc[node as usize].len() as u32
    }
}

impl DirectedNeighbors<u32> for VecGraph {
    type NeighborsIterator<'a> = std::slice::Iter<'a, u32>;
    fn out_neighbors(&self, node: u32) -> Self::NeighborsIterator<'_> {
        self.out[node as usize].iter()
    }
    fn in_neighbors(&self, node: u32) -> Self::NeighborsIterator<'_> {
        self.inc[node as usize].iter()
    }
}

#[test]
fn a_graph_type_that_only_knows_the_reference_traits_is_accepted() {
    let g = VecGraph { out: vec![vec![1], vec![2], vec![0], vec![]], inc: vec![vec![2], vec![0], vec![1], vec![]] };
    let csr: DirectedCsrGraph<u32> = GraphBuilder::new().csr_layout(CsrLayout::Sorted).edges(vec![(0, 1), (1, 2), (2, 0), (3, 3)]).build();
    let _ = csr; // (3, 3) only makes node 3 exist in the CSR build; VecGraph's node 3 is isolated
    let (scores, iterations, _) = page_rank(&g, PageRankConfig::default());
    assert_eq!(scores.len(), 4);
    assert!(iterations >= 1);
    assert_eq!(wcc_afforest(&g, WccConfig::default()).to_vec(), vec![0u32, 0, 0, 3]);
    assert_eq!(wcc_baseline(&g, WccConfig::default()).to_vec(), vec![0u32, 0, 0, 3]);
}

#[test]
fn graphs_rebuilt_at_one_address_with_equal_counts_are_not_confused() {
    // the failure of an address-keyed cache: same stack slot, same node and edge counts, other edges
    let mut results = Vec::new();
    for edges in [vec![(0u32, 1u32), (1, 2), (2, 0), (3, 0)], vec![(0, 1), (1, 0), (2, 3), (3, 2)]] {
        let graph: DirectedCsrGraph<u32> = GraphBuilder::new().csr_layout(CsrLayout::Sorted).edges(edges).build();
        results.push(wcc_afforest(&graph, WccConfig::default()).to_vec());
    }
    assert_eq!(results[0], vec![0u32, 0, 0, 0]);
    assert_eq!(results[1], vec![0u32, 0, 2, 2]);
}

#[test]
fn relabel_through_on_device_drops_the_stale_copy() {
    let graph: UndirectedCsrGraph<u32> =
        GraphBuilder::new().csr_layout(CsrLayout::Deduplicated).edges(vec![(0, 1), (1, 2), (0, 2), (2, 3)]).build();
    let mut resident = OnDevice::new(graph);
    assert_eq!(resident.global_triangle_count(), 1);
    resident.relabel(); // make_degree_ordered through get_mut(): the device copy of the old ids is gone
    assert_eq!(resident.global_triangle_count(), 1);
    assert_eq!(resident.degree(0), 3); // the old node 2 is node 0 now
}
