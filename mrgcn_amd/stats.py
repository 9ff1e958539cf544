"""Which path did a step take?  The fast paths of this package are chosen by notes that travel on tensor objects
(functional._grad_meta, `_mrgcn_rows`, ...): a `.clone()` or `.to()` in user code silently drops a note and the step
falls back to a slower, equally correct path.  Every such decision point counts here, so that a test (or a user) can
assert that the path it expects was the one that ran:

    mrgcn_amd.reset_stats(); train_step(...); assert mrgcn_amd.stats()["backward.support"] == 2
"""
from collections import Counter

_COUNTS: Counter = Counter()


def bump(key: str, n: int = 1) -> None:
    _COUNTS[key] += n


def stats() -> dict:
    """Counts since the last `reset_stats()`: `backward.support` / `backward.marking` / `backward.general` /
    `backward.wide_input` (which backward a fused layer ran), `weight_I.fused_rows` / `weight_I.rows` / `weight_I.dense`
    (how the node table's gradient left the layer), `loss.sparse_rows` / `loss.flagged` / `loss.plain` (how much the
    cross-entropy knew about its rows), `discovered_rows` (plain dense gradients whose rows were looked up),
    `adam.list` / `adam.rows_fused` / `adam.rows` (row-sparse optimizer updates)."""
    return dict(_COUNTS)


def reset_stats() -> None:
    _COUNTS.clear()
