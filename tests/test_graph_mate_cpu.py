"""Host-only pieces of the graph_mate-shaped front end (graph_amd/graph_mate.py): no GPU needed."""
import pytest


@pytest.mark.parametrize("micros,text", [
    (0, "0ns"), (1, "1µs"), (12, "12µs"), (999, "999µs"), (1000, "1ms"), (1234, "1.234ms"), (1500, "1.5ms"),
    (999999, "999.999ms"), (1000000, "1s"), (1234567, "1.234567s"), (2500000, "2.5s"), (61000001, "61.000001s"),
])
def test_load_took_prints_like_rusts_duration_debug(micros, text):
    """the reference's __repr__ shows `load_took: {:?}` of Duration::from_micros (crates/mate/src/graphs/mod.rs:374-383):
    largest unit with an integer part, fraction without trailing zeros"""
    from graph_amd import graph_mate

    assert graph_mate._duration_debug(micros) == text


def test_timed_adds_to_what_the_graph_cost_so_far():
    """crates/mate/src/graphs/mod.rs:407-432"""
    from graph_amd import graph_mate

    out, micros = graph_mate._timed(1000, lambda: "x")
    assert out == "x" and micros >= 1000


def test_the_surface_of_the_reference_module_is_there():
    """graph_mate.pyi: classes, their methods and the read-only load_micros"""
    from graph_amd import graph_mate as M

    for name in ("load", "from_numpy", "from_pandas", "node_count", "edge_count", "out_degree", "in_degree", "out_neighbors",
                 "in_neighbors", "copy_out_neighbors", "copy_in_neighbors", "to_undirected", "page_rank", "wcc", "load_micros",
                 "__repr__"):
        assert hasattr(M.DiGraph, name), name
    for name in ("load", "from_numpy", "from_pandas", "node_count", "edge_count", "degree", "neighbors", "copy_neighbors",
                 "make_degree_ordered", "global_triangle_count", "load_micros", "__repr__"):
        assert hasattr(M.Graph, name), name
    assert isinstance(M.DiGraph.load_micros, property) and M.DiGraph.load_micros.fset is None
    assert [m.name for m in M.Layout] == ["Sorted", "Unsorted", "Deduplicated"]
    assert [m.name for m in M.FileFormat] == ["Graph500", "EdgeList"]
