"""Host logic of the round schedule (csrc/mcpc_api.hip: setup_rounds), mirrored in Python and checked for the properties the kernel
launches rely on -- no GPU needed.

A shard of U 16-chain units on C < U CUs runs cycles of k launches of q steps; launch i holds groups i .. i+m-1 (mod k) of the k
groups the units are dealt into.  What must hold for ANY (U, C):
  * every launch fits the CUs;
  * every unit takes part in exactly m launches of a cycle, so that all units have advanced m q steps after it;
  * the number of launches of the cycle a unit has ALREADY taken part in (`rel`, from which a workgroup derives its first step)
    counts 0, 1, .. m-1 in launch order: a unit's steps are done in order and none twice;
  * (k, m) is the smallest k within 3 % of the best k / m over k <= 16, and no worse than ceil(U / C) plain rounds.
"""
import math

import pytest


def plan(units, n_cu=256):
    fit = {}
    for k in range(2, 17):
        sizes = [(g + 1) * units // k - g * units // k for g in range(k)]
        for m in range(k - 1, 0, -1):
            if max(sum(sizes[(i + j) % k] for j in range(m)) for i in range(k)) <= n_cu:
                fit[k] = m
                break
    if not fit:
        return ((units + n_cu - 1) // n_cu, 1)
    bk = min(fit, key=lambda k: (k / fit[k], k))
    for k in sorted(fit):
        if k < bk and 100.0 * k * fit[bk] <= 103.0 * bk * fit[k]:
            return (k, fit[k])
    return (bk, fit[bk])


def tables(units, k, m):
    """Per launch of a cycle: [(unit, rel)], as setup_rounds builds them."""
    done = [0] * k
    out = []
    for i in range(k):
        rows = []
        for j in range(m):
            g = (i + j) % k
            rows += [(u, done[g]) for u in range(g * units // k, (g + 1) * units // k)]
        for j in range(m):
            done[(i + j) % k] += 1
        out.append(rows)
    return out


def test_known_cases():
    assert plan(375) == (3, 2)            # 6000 chains on 256 CUs: 250 workgroups per launch, 1.5 launch times per step
    assert plan(512) == (2, 1) and plan(1024) == (4, 1) and plan(3000) == (12, 1)
    assert plan(300) == (6, 5)            # (13, 11) would be 1.5 % better with launches of 5 steps inside a Hebbian segment
    assert plan(258) == (12, 11) and plan(438) == (7, 4) and plan(313) == (5, 4)


@pytest.mark.parametrize("n_cu", [256, 304, 64])
def test_cycle_properties_for_every_shard_size(n_cu):
    for units in list(range(n_cu + 1, 3 * n_cu + 2)) + [5 * n_cu + 3, 12 * n_cu - 1, 16 * n_cu, 16 * n_cu + 1, 40 * n_cu + 7]:
        k, m = plan(units, n_cu)
        assert 1 <= m < k or (m == 1 and k >= 2)
        launches = tables(units, k, m)
        seen = {u: [] for u in range(units)}
        for rows in launches:
            assert 0 < len(rows) <= n_cu, (units, k, m, len(rows))
            assert len({u for u, _ in rows}) == len(rows)                 # a unit at most once per launch
            for u, rel in rows:
                seen[u].append(rel)
        for u, rels in seen.items():
            assert rels == list(range(m)), (units, k, m, u, rels)       # m launches, in order, each step range once
        # never worse than plain rounds of at most n_cu units, and within 3 % of the best k <= 16 can do
        assert k / m <= math.ceil(units / n_cu) + 1e-9
        best = min((kk / mm for kk in range(2, 17) for mm in range(1, kk)
                    if max(sum(((g + 1) * units // kk - g * units // kk) for g in [(i + j) % kk for j in range(mm)]) for i in range(kk)) <= n_cu),
                   default=None)
        if best is not None:
            assert k / m <= 1.03 * best + 1e-9, (units, k, m, best)
