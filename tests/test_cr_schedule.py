"""Host-side arithmetic of the cyclic-reduction launches (no GPU): the level / task decode that
gpmp2_amd/csrc/wide_cr.h (k_cr_level_wide, wcr_forward / wcr_backward, k_finish_trial_wide) and dense_kernels.hip
(k_dense_cr_level, k_dense_cr_back) share with their launchers, restated and checked for every trajectory length:
each block is eliminated exactly once, after both neighbours it absorbs, and back-substituted after the blocks it
reads."""
import pytest


def hfinal_of(N):
    h = 1
    while h <= N:
        h <<= 1
    return h


def forward_tasks(N, h, first_level):
    """(kind, block) of every task of forward level h; level `first_level` has no U tasks (nothing to absorb)"""
    final = h == hfinal_of(N)
    countE = 1 if final else ((N // h) + 1) // 2
    countU = 0 if (final or h == first_level == 1) else (N // (2 * h)) + 1
    out = []
    for idx in range(countE + countU):
        elim = idx < countE
        j = (0 if final else h * (2 * idx + 1)) if elim else 2 * h * (idx - countE)
        out.append(("E" if elim else "U", j))
    return out


@pytest.mark.parametrize("first_level", [1, 2])
def test_every_block_is_eliminated_once_and_in_order(first_level):
    # first_level 1: dense path (level 1 is a launch); 2: tile paths (level 1 happens in the assemble kernel)
    for N in range(0 if first_level == 1 else 1, 260):   # (the tile kernels need total_step >= 1)
        hf = hfinal_of(N)
        eliminated_at = {}
        if first_level == 2:
            for j in range(1, N + 1, 2):
                eliminated_at[j] = 1
        h = first_level
        while h <= hf:
            tasks = forward_tasks(N, h, first_level)
            blocks = [j for _, j in tasks]
            assert len(set(blocks)) == len(blocks) and all(0 <= j <= N for j in blocks), (N, h)
            for kind, j in tasks:
                assert j % h == 0 or h == hf
                for jn in (j - h // 2, j + h // 2):          # neighbours absorbed at this level
                    if h > 1 and 0 <= jn <= N:
                        assert eliminated_at.get(jn) == h // 2, (N, h, j, jn)
                if kind == "E":
                    assert j not in eliminated_at, (N, h, j)
                    eliminated_at[j] = h
            h <<= 1
        assert sorted(eliminated_at) == list(range(N + 1)), N
        # backward: level h solves the blocks eliminated at level h from x_{j-h}, x_{j+h}
        solved = set()
        h = hf
        while h >= 1:
            final = h == hf
            count = 1 if final else ((N // h) + 1) // 2
            for idx in range(count):
                j = 0 if final else h * (2 * idx + 1)
                assert eliminated_at[j] == h
                if not final:
                    for jn in (j - h, j + h):
                        if 0 <= jn <= N:
                            assert jn in solved, (N, h, j, jn)
                solved.add(j)
            h >>= 1
        assert len(solved) == N + 1


def test_wide_split_tail_groups_cover_every_block():
    """k_finish_step / k_finish_trial / k_finish_trial_wide: groups of 8 blocks; the multiples of 8 come from the solve
    kernel (levels >= 8), block
    8q+4 needs x_{8q}, x_{8q+8}, then 8q+2 / 8q+6, then the odd ones -- the neighbours always sit in slots 0..8"""
    for N in range(16, 260):
        groups = (N + 8) // 8
        seen = set()
        for q in range(groups):
            have = {8 * q} | ({8 * q + 8} if 8 * q + 8 <= N else set())
            for h, waves in ((4, (4,)), (2, (2, 6)), (1, (1, 3, 5, 7))):
                new = set()
                for wv in waves:
                    i = 8 * q + wv
                    if i > N:
                        continue
                    for jn in (i - h, i + h):
                        if 0 <= jn <= N:
                            assert jn in have and 0 <= jn - 8 * q <= 8, (N, q, i, jn)
                    new.add(i)
                have |= new
            seen |= {i for i in have if i // 8 == q}
        assert seen == set(range(N + 1)), N
