"""Property tests of the rank-independent pieces of the multi-GPU path
(safeopt_amd/dist.py): the row partition and the merges every rank computes
identically from all-gathered per-rank results."""
import os
import sys

import numpy as np
from hypothesis import given, settings, strategies as st

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safeopt_amd.dist import merge_argmax, merge_topk, shard_range  # noqa: E402


@given(st.integers(0, 10 ** 7), st.integers(1, 16))
def test_shard_range_is_a_balanced_contiguous_partition(N, world):
    blocks = [shard_range(N, r, world) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == N
    assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
    sizes = [hi - lo for lo, hi in blocks]
    assert min(sizes) >= 0 and max(sizes) - min(sizes) <= 1
    assert sizes == sorted(sizes, reverse=True)      # the longer blocks come first


widths = st.sampled_from([0.0, 0.25, 0.5, 0.5, 1.0, 1.5])   # few values: many exact ties


@settings(max_examples=200, deadline=None)
@given(st.lists(st.lists(widths, min_size=0, max_size=6), min_size=1, max_size=5),
       st.integers(1, 8), st.booleans())
def test_merge_topk_is_the_global_visiting_order(per_rank, k, by_index):
    """Every rank contributes its own next-k in visiting order (padded with
    index -1); the merge must be the next-k of the union."""
    ws, idxs, allw, alli, off = [], [], [], [], 0
    for vals in per_rank:
        w = np.asarray(vals, dtype=float)
        i = off + np.arange(w.size, dtype=np.int64)
        off += w.size + 3                             # gaps: rows that are no candidates
        order = np.argsort(i) if by_index else np.lexsort((-i, -w))
        w, i = w[order][:k], i[order][:k]
        allw.append(w); alli.append(i)
        ws.append(np.concatenate([w, np.full(k - w.size, -np.inf)]))
        idxs.append(np.concatenate([i, np.full(k - i.size, -1, dtype=np.int64)]))
    w_m, i_m = merge_topk(ws, idxs, k, by_index=by_index)
    W, I = np.concatenate(allw), np.concatenate(alli)
    order = np.argsort(I) if by_index else np.lexsort((-I, -W))
    assert np.array_equal(i_m, I[order][:k]) and np.array_equal(w_m, W[order][:k])


@given(st.lists(st.tuples(st.sampled_from([-1.0, 0.0, 0.5, 0.5, 2.0]),
                          st.integers(-1, 40)), min_size=1, max_size=8))
def test_merge_argmax_picks_the_largest_value_lowest_index(pairs):
    vals = np.array([p[0] for p in pairs]); idx = np.array([p[1] for p in pairs], dtype=np.int64)
    v, i = merge_argmax(vals, idx)
    keep = idx >= 0
    if not keep.any():
        assert i == -1 and v == -np.inf
        return
    best = vals[keep].max()
    assert v == best and i == idx[keep][vals[keep] == best].min()
