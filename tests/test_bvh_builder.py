"""CPU tests of the hierarchy builder behind rtmi_upload_scene (rtmi_bvh_build needs no device)."""
import ctypes as C

import numpy as np
import pytest

LEAF = 0x80000000
EMPTY = 0xFFFFFFFF


def build(rtmi, boxes, leaf_base=0):
    lib = rtmi.abi.load()
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
    n = len(boxes)
    nodes = np.zeros((max(1, n), 16), dtype=np.uint32)
    order = np.zeros(max(1, n), dtype=np.uint32)
    root = C.c_uint32()
    rc = lib.rtmi_bvh_build(boxes.ctypes.data, n, leaf_base, nodes.ctypes.data, nodes.nbytes, order.ctypes.data,
                            C.byref(root))
    assert rc >= 0
    return nodes[:rc], order[:n], root.value


def walk(nodes, order, boxes, ref, leaf_base, seen, depth=1):
    """Returns (lo, hi, depth) of the subtree; checks every stored child box against its contents."""
    if ref & LEAF:
        first, count = (ref & 0x0FFFFFFF) - leaf_base, (ref >> 28) & 7
        assert 1 <= count <= 2
        ids = order[first:first + count]
        seen.extend(int(i) for i in ids)
        b = boxes[ids]
        return np.nanmin(b[:, :3], axis=0), np.nanmax(b[:, 3:], axis=0), depth
    f = nodes[ref].view(np.float32)
    out = []
    for child, lo, hi in ((nodes[ref][3], f[0:3], f[4:7]), (nodes[ref][7], f[8:11], f[12:15])):
        clo, chi, d = walk(nodes, order, boxes, int(child), leaf_base, seen, depth + 1)
        assert (lo <= clo).all() and (hi >= chi).all()
        assert (lo == clo).all() and (hi == chi).all()     # boxes are tight, margins belong to the traversal
        out.append((clo, chi, d))
    return np.minimum(out[0][0], out[1][0]), np.maximum(out[0][1], out[1][1]), max(out[0][2], out[1][2])


# 70000: large enough for the builder to split the work over several threads
@pytest.mark.parametrize("n", [1, 2, 4, 5, 9, 1000, 4097, 70000])
def test_tree_covers_every_box_once(rtmi, n):
    rs = np.random.RandomState(n)
    c = rs.uniform(-10, 10, size=(n, 3))
    r = rs.uniform(0.01, 0.5, size=(n, 3))
    boxes = np.concatenate([c - r, c + r], axis=1).astype(np.float32)
    leaf_base = 12345
    nodes, order, root = build(rtmi, boxes, leaf_base)
    assert sorted(order.tolist()) == list(range(n))
    seen = []
    _, _, depth = walk(nodes, order, boxes, root, leaf_base, seen)
    assert sorted(seen) == list(range(n))
    assert depth <= 25    # 24 levels of inner nodes and the leaves: what the traversal stack is sized for


def test_threads_do_not_change_the_tree(rtmi):
    """Nodes are numbered in pre-order whichever thread built them: every inner child reference points
    forward, a left inner child is the next node, and two builds give the same bytes."""
    rs = np.random.RandomState(5)
    c = rs.uniform(-10, 10, size=(150000, 3))
    boxes = np.concatenate([c - 0.05, c + 0.05], axis=1).astype(np.float32)
    nodes, order, root = build(rtmi, boxes)
    nodes2, order2, root2 = build(rtmi, boxes)
    assert root == root2 == 0 and (nodes == nodes2).all() and (order == order2).all()
    left, right = nodes[:, 3], nodes[:, 7]
    idx = np.arange(len(nodes), dtype=np.uint32)
    inner_l, inner_r = (left & LEAF) == 0, (right & LEAF) == 0
    assert (left[inner_l] == idx[inner_l] + 1).all()
    assert (right[inner_r] > idx[inner_r]).all() and (right[inner_r] < len(nodes)).all()
    # every node except the root is referenced exactly once
    refs = np.concatenate([left[inner_l], right[inner_r]])
    assert len(refs) == len(nodes) - 1 and len(np.unique(refs)) == len(refs)


def test_empty_and_identical_boxes(rtmi):
    nodes, order, root = build(rtmi, np.zeros((0, 6), dtype=np.float32))
    assert len(nodes) == 0 and root == EMPTY
    boxes = np.tile(np.array([[0, 0, 0, 1, 1, 1]], dtype=np.float32), (100, 1))
    nodes, order, root = build(rtmi, boxes)
    seen = []
    walk(nodes, order, boxes, root, 0, seen)
    assert sorted(seen) == list(range(100))
    assert order.tolist() == list(range(100))      # equal Morton codes keep the input order


def test_order_is_deterministic_and_spatial(rtmi):
    rs = np.random.RandomState(3)
    c = rs.uniform(0, 1, size=(5000, 3))
    boxes = np.concatenate([c, c], axis=1).astype(np.float32)
    a = build(rtmi, boxes)
    b = build(rtmi, boxes)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]) and a[2] == b[2]
    # neighbours in leaf order are close in space: mean step far below the mean distance of random pairs
    step = np.linalg.norm(np.diff(c[a[1]], axis=0), axis=1).mean()
    assert step < 0.15


def test_depth_limit_on_adversarial_input(rtmi):
    """Exponentially spaced points make every surface-area split peel off a few primitives; the builder
    has to fall back to halving early enough to stay within the traversal stack."""
    n = 20000
    x = 2.0 ** np.linspace(-60, 60, n)
    boxes = np.zeros((n, 6), dtype=np.float32)
    boxes[:, 0] = boxes[:, 3] = x
    nodes, order, root = build(rtmi, boxes)
    seen = []
    _, _, depth = walk(nodes, order, boxes, root, 0, seen)
    assert sorted(seen) == list(range(n)) and depth <= 25


def test_margin_floor_is_the_smallest_below(rtmi):
    """Words 11 and 15 of a node: smallest margin floor below the left / right child (0 through this
    entry point, which passes none)."""
    rs = np.random.RandomState(9)
    c = rs.uniform(-1, 1, size=(64, 3))
    boxes = np.concatenate([c - 0.1, c + 0.1], axis=1).astype(np.float32)
    nodes, _, _ = build(rtmi, boxes)
    f = nodes.view(np.float32)
    assert (f[:, 11] == 0).all() and (f[:, 15] == 0).all()


def quantize(rtmi, nodes):
    lib = rtmi.abi.load()
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32).reshape(-1, 16)
    q = np.zeros((max(1, len(nodes)), 8), dtype=np.uint32)
    frame = np.zeros(6, dtype=np.float32)
    rc = lib.rtmi_bvh_quantize(nodes.ctypes.data, len(nodes), q.ctypes.data, q.nbytes, frame.ctypes.data)
    assert rc == len(nodes)
    return q[:len(nodes)], frame[:3].astype(np.float64), frame[3:].astype(np.float64)


def _child_boxes(node_words):
    f = node_words.view(np.float32)
    return (f[0:3], f[4:7]), (f[8:11], f[12:15])


def test_quantised_boxes_enclose_the_float_boxes(rtmi):
    rs = np.random.RandomState(4)
    c = rs.uniform(-40, 60, size=(3000, 3))
    r = rs.uniform(0.01, 0.5, size=(3000, 1))
    boxes = np.concatenate([c - r, c + r], axis=1).astype(np.float32)
    nodes, _, _ = build(rtmi, boxes)
    q, origin, step = quantize(rtmi, nodes)
    assert (q[:, 0] & 0x87FFFFFF == nodes[:, 3] & 0x87FFFFFF).all() and (q[:, 1] & 0x87FFFFFF == nodes[:, 7] & 0x87FFFFFF).all()
    assert not (q[:, :2] & 0x08000000).any()                       # every box finite and inside the frame
    for i in range(len(nodes)):
        for side, (lo, hi) in enumerate(_child_boxes(nodes[i])):
            w = q[i, 2 + 3 * side: 5 + 3 * side]
            qlo = origin + (w & 0xFFFF).astype(np.float64) * step
            qhi = origin + (w >> 16).astype(np.float64) * step
            assert (qlo <= lo.astype(np.float64)).all() and (qhi >= hi.astype(np.float64)).all()
            assert ((lo.astype(np.float64) - qlo) <= 2.5 * step).all() and ((qhi - hi.astype(np.float64)) <= 2.5 * step).all()


def test_quantise_infinite_inverted_and_degenerate_boxes(rtmi):
    """Boxes with infinite bounds (an infinite-radius sphere), an inverted box (a child without finite primitives),
    a NaN bound (which the builder never stores and the quantiser reads as "no primitives"), a point box and a box
    of astronomic extent: none of them may reach an integer conversion unchecked (this entry point runs under UBSan
    in the sanitizer build), children with an infinite plane are flagged as always visited, and the finite ones
    still enclose."""
    inf, nan = np.inf, np.nan
    node = np.zeros((4, 16), dtype=np.uint32)
    f = node.view(np.float32)

    def put(i, l_lo, l_hi, r_lo, r_hi):
        f[i, 0:3], f[i, 4:7], f[i, 8:11], f[i, 12:15] = l_lo, l_hi, r_lo, r_hi
        node[i, 3], node[i, 7] = 0x90000000 | (2 * i), 0x90000000 | (2 * i + 1)   # leaves of one primitive

    put(0, (-1, -1, -1), (1, 1, 1), (-inf, -inf, -inf), (inf, inf, inf))           # finite | infinite
    put(1, (2, 2, 2), (2, 2, 2), (3.4e38, 0, 0), (-3.4e38, 0, 0))                  # a point | inverted
    put(2, (nan, 0, 0), (1, 1, 1), (-5, -5, -5), (5, 5, 5))                        # NaN bound | finite
    put(3, (-1e30, 0, 0), (1e30, 1, 1), (0, 0, 0), (0.5, inf, 0.5))                # huge but finite | one infinite plane
    q, origin, step = quantize(rtmi, node)
    flag = 0x08000000
    assert not q[0, 0] & flag and q[0, 1] & flag
    assert not q[1, 0] & flag
    assert not q[2, 0] & flag and not q[2, 1] & flag
    for w in list(q[1, 5:8]) + list(q[2, 2:5]):                    # inverted and NaN children: lower 0xFFFF, upper 0
        assert int(w) == 0x0000FFFF
    assert q[3, 1] & flag
    assert np.isfinite(origin).all() and np.isfinite(step).all() and (step > 0).all()
    for i, side in ((0, 0), (1, 0), (2, 1), (3, 0)):
        lo, hi = _child_boxes(node[i])[side]
        w = q[i, 2 + 3 * side: 5 + 3 * side]
        qlo = origin + (w & 0xFFFF).astype(np.float64) * step
        qhi = origin + (w >> 16).astype(np.float64) * step
        assert (qlo <= lo.astype(np.float64)).all() and (qhi >= hi.astype(np.float64)).all()
