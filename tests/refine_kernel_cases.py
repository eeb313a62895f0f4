"""Seeded inputs of the refinement-kernel parity cases (shared by the golden generator and the tests)."""
import numpy as np

import cases

TREE_CASE = "sh4_d6"
N_NEW, SPC = 37, 5
VARIANTS = {"plain": (False, -1), "viewdir_embedding": (True, 3)}


def options(mnv, need_viewdir, embedding):
    opt = mnv.RenderOptions.cli_defaults()
    opt.samples_per_corner, opt.need_viewdir, opt.appearance_embedding = SPC, need_viewdir, embedding
    return opt, 3 + (3 if need_viewdir else 0) + (1 if embedding != -1 else 0)


def grid(mnv):
    g = mnv.ClusterGrid()
    g.grid_dim[0], g.grid_dim[1] = 3, 2
    for i, (lo, rng) in enumerate([(-1.0, 2.0), (-1.1, 2.2), (-0.9, 1.8)]):
        g.min_position[i], g.range[i] = lo, rng
    return g


def add_children_inputs(mnv, variant):
    tree = cases.make_tree(mnv, cases.CASES[TREE_CASE]["tree"])
    _, child, parent = tree.host_arrays()
    cap = tree.capacity
    opt, dim = options(mnv, *VARIANTS[variant])
    rng = np.random.default_rng(5)
    leaves = np.argwhere(child == 0)
    parent_nodes = np.ascontiguousarray(leaves[rng.choice(len(leaves), N_NEW, replace=False)], dtype=np.int32)
    visited = np.zeros(cap + N_NEW, np.int32)
    visited[:cap:2] = 1
    samples = rng.uniform(0, 1, (N_NEW * 8, SPC, dim)).astype(np.float32)
    return tree, opt, dim, parent_nodes, visited, samples


def generate_samples_inputs(mnv, variant):
    tree = cases.make_tree(mnv, cases.CASES[TREE_CASE]["tree"])
    _, child, _ = tree.host_arrays()
    opt, dim = options(mnv, *VARIANTS[variant])
    rng = np.random.default_rng(6)
    nodes = np.ascontiguousarray(np.argwhere(child == 0)[::17][:64], dtype=np.int32)  # argwhere slices are not C-contiguous
    samples = rng.uniform(0, 1, (len(nodes), SPC, dim)).astype(np.float32)
    return tree, opt, dim, nodes, samples


def adjust_inputs(mnv, orc):
    """Visit marks of a sparse frame decide what is deleted (descendants of a deleted chunk are deleted too)."""
    spec = cases.CASES[TREE_CASE]
    tree = cases.make_tree(mnv, spec["tree"])
    cam = cases.make_camera(mnv, dict(spec["camera"], width=40, height=32, fx=140.0))
    opt = cases.make_options(mnv, spec["options"])
    v = tree.host_view()
    visited = np.zeros(v.capacity, np.int32)
    orc.render(orc.tree_from_view(v), cam.c, opt, visited=visited, track_visit=True)
    to_delete = (visited == 0).astype(np.uint8)
    return tree, to_delete, np.cumsum(to_delete, dtype=np.int32)
