"""Test helper: replaces the BVH section of every model of a built scene by a deliberately lopsided
tree, so that traversal stacks get deeper than any tree the host builder makes from a small mesh.

The streams keep the reference's layout (src/TraceableObject.cpp:202-228): per model
[light identifiers L x 5][nodes N x 9][identifiers T x 5][vertices V x 8], node = min.xyz, max.xyz,
left, right, isLeaf; leaves name a range of identifiers, which are rewritten in leaf order; children
follow their parent (pre-order, left subtree first).  Checker and backend read the same streams, so
the comparison between them stays exact on such a tree.
"""

import numpy as np


def _build(order, centroid, lo_box, hi_box, frac, overlapping):
    """Returns (nodes[N, 9] as float32 words, leaf order, depth)."""
    nodes, leaf_order = [], []
    max_depth = 0
    # explicit stack: (triangle indices, axis, parent, is_right, depth)
    todo = [(order, 0, -1, False, 1)]
    while todo:
        tris, axis, parent, is_right, depth = todo.pop()
        me = len(nodes)
        max_depth = max(max_depth, depth)
        if parent >= 0:
            nodes[parent][7 if is_right else 6] = me
        # overlapping: every node carries the bounds of the whole model (loose but valid boxes).  Both
        # children of a node are then entered at the same distance, the walk takes the right one
        # first and stacks the left one (src/RayScene.cpp:722-731), and since the larger part of
        # each cut is the right child the stack holds depth - 1 entries at the bottom of the tree.
        inside = order if overlapping else tris
        box_lo = (lo_box[inside].min(axis=0) - np.float32(1e-6)).astype(np.float32)
        box_hi = (hi_box[inside].max(axis=0) + np.float32(1e-6)).astype(np.float32)
        if len(tris) <= 2:
            first = len(leaf_order)
            leaf_order.extend(int(t) for t in tris)
            nodes.append([*box_lo, *box_hi, first, len(leaf_order), True])
            continue
        axis = (axis + 1) % 3
        tris = tris[np.argsort(centroid[tris, axis], kind="stable")]
        cut = min(max(int(len(tris) * frac), 1), len(tris) - 1)
        nodes.append([*box_lo, *box_hi, -1, -1, False])
        todo.append((tris[cut:], axis, me, True, depth + 1))
        todo.append((tris[:cut], axis, me, False, depth + 1))   # popped first: the left subtree follows its parent
    words = np.zeros((len(nodes), 9), np.float32)
    ints = words.view(np.int32)
    for k, n in enumerate(nodes):
        words[k, :6] = n[:6]
        ints[k, 6], ints[k, 7] = n[6], n[7]
        words[k, 8] = 1.0 if n[8] else 0.0
    return words, np.asarray(leaf_order, np.int64), max_depth


def rebuild_trees(scene, frac, overlapping=False):
    """Rewrites scene.streams[3] and [4] in place; returns the depth of every model's new tree
    (root = 1, the figure the traversal stack must hold)."""
    info = scene.streams[3].view(np.int32).reshape(-1, 4).copy()
    mdl = scene.streams[4]
    starts = sorted(int(info[m, 3]) for m in range(scene.models_count)) + [len(mdl)]
    blobs, depths = [], []
    for m in range(scene.models_count):
        T, L, N, loc = (int(x) for x in info[m])
        end = starts[starts.index(loc) + 1]
        lights = mdl[loc:loc + L * 5]
        ids = mdl[loc + L * 5 + N * 9:loc + L * 5 + N * 9 + T * 5].reshape(T, 5)
        verts = mdl[loc + L * 5 + N * 9 + T * 5:end]
        if T == 0 or N == 0:
            blobs.append((lights, mdl[loc + L * 5:loc + L * 5 + N * 9].reshape(N, 9), ids, verts))
            depths.append(0)
            continue
        corner = verts.reshape(-1, 8)[ids[:, :3].view(np.int32)][..., :3]          # [T, 3 corners, xyz]
        lo_box, hi_box = corner.min(axis=1), corner.max(axis=1)
        centroid = corner.astype(np.float64).mean(axis=1)
        nodes, leaf_order, depth = _build(np.arange(T), centroid, lo_box, hi_box, frac, overlapping)
        blobs.append((lights, nodes, ids[leaf_order], verts))
        depths.append(depth)
    out, at = [], 0
    for m, (lights, nodes, ids, verts) in enumerate(blobs):
        info[m, 2] = len(nodes)
        info[m, 3] = at
        for part in (lights, nodes, ids, verts):
            out.append(np.ascontiguousarray(part, np.float32).reshape(-1))
            at += out[-1].size
    scene.streams[3] = info.reshape(-1).view(np.float32).copy()
    scene.streams[4] = np.concatenate(out) if out else np.zeros(0, np.float32)
    return depths


def with_depth(make_scene, lo, hi, overlapping=False):
    """Builds the scene with make_scene() and re-trees it so that the deepest tree has lo..hi levels
    (bisection on the split ratio: the more lopsided the cut, the deeper the tree)."""
    f_lo, f_hi = 0.02, 0.5
    depths = None
    for _ in range(24):
        frac = 0.5 * (f_lo + f_hi)
        scene = make_scene()
        depths = rebuild_trees(scene, frac, overlapping)
        if lo <= max(depths) <= hi:
            return scene, depths
        if max(depths) > hi:
            f_lo = frac
        else:
            f_hi = frac
    raise AssertionError(f"no split ratio gives a tree {lo}..{hi} levels deep (last: {depths})")
