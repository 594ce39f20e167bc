"""np_walk_k's deal (csrc/narrow_walk.inc, "THE DEAL"): the list is dealt in batches of 21 queries, as one list or as eight
interleaved sub-lists of segments of 2^seg_shift batches with the rest on sub-list 0.  The formulas of run_walk() (seg_shift,
rounds) and of the kernel (my_len, the batch a sub-list ticket maps to) are restated here and checked to be a bijection of the
list for many lengths and grid sizes -- every query dealt exactly once whatever block serves which sub-list."""
import numpy as np
import pytest

NW_BATCH = 21


def host_params(n, blocks):  # run_walk()
    n_batches = (n + NW_BATCH - 1) // NW_BATCH
    seg_shift, rounds = 0, 0
    if blocks >= 8 and n_batches >= 32:
        while seg_shift < 5 and (n_batches >> (seg_shift + 1 + 3)) >= 4:
            seg_shift += 1
        rounds = n_batches >> (seg_shift + 3)
    return n_batches, seg_shift, rounds


def deal(n, blocks):  # np_walk_k: every (sub-list, ticket) -> (first query, count)
    n_batches, seg_shift, rounds = host_params(n, blocks)
    out = []
    for sub in range(8 if rounds else 1):
        inter = rounds << seg_shift
        my_len = inter + (n_batches - (inter << 3) if sub == 0 else 0)
        for t in range(my_len):
            if t < inter:
                g = ((((t >> seg_shift) << 3) + sub) << seg_shift) | (t & ((1 << seg_shift) - 1))
            else:
                g = (inter << 3) + (t - inter)
            pos = g * NW_BATCH
            out.append((pos, min(NW_BATCH, n - pos)))
    return out


@pytest.mark.parametrize("blocks", [1, 4, 7, 8, 9, 100, 1536])
def test_every_query_is_dealt_exactly_once(blocks):
    lengths = {1, 20, 21, 22, 671, 672, 673, 1000, 5375, 5376, 5377, 65536, 100001, 630000}
    for nb in (31, 32, 33, 255, 256, 257, 1023, 1024, 1025):
        lengths |= {NW_BATCH * nb - 1, NW_BATCH * nb, NW_BATCH * nb + 1}
    for n in sorted(lengths):
        seen = np.zeros(n, np.int32)
        for pos, count in deal(n, blocks):
            assert 0 <= pos < n and count >= 1, (n, blocks, pos, count)
            seen[pos:pos + count] += 1
        assert (seen == 1).all(), (n, blocks)


def test_sub_lists_are_balanced_and_need_eight_blocks():
    n = 5_063_853
    n_batches, seg_shift, rounds = host_params(n, 1536)
    assert seg_shift == 5 and rounds == n_batches >> 8
    inter = rounds << seg_shift
    assert n_batches - 8 * inter < 8 << seg_shift  # what sub-list 0 takes on top is less than one round
    assert host_params(n, 7)[2] == 0               # fewer than eight blocks: one list, one ticket word
    assert host_params(31 * NW_BATCH, 1536)[2] == 0
