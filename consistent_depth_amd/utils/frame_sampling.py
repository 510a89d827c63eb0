"""Which frame pairs to fine-tune on.

Public names follow the reference's module of the same name (/root/reference/utils/frame_sampling.py:12-155 --
`SamplePairsMode`, `SamplePairsOptions`, `Pair`, `SamplePairs.sample / to_one_way`, `to_in_range`), because
`video_dataset` / `params` / the pair store address it by those names.  The sampling itself is written as plain pair
generators over frame *indices*:

  dyadic(n, level, stride)   (s, s + 2^level) for s = 0, stride, 2*stride, ... < n
  hierarchical               levels with min_dist <= 2^level <= max_dist, stride 2^level
  hierarchical2              same levels, stride 2^(level-1): every window also starts at its mid point
  consecutive                level 0 only;   exhausted: all i < j (the reference's version raises TypeError)

244 frames -> 715 one-way pairs with hierarchical2, 1000 frames -> 2979."""
from __future__ import annotations

import enum
from collections import namedtuple
from typing import Any, Dict, Iterable, Iterator, NamedTuple, Optional, Set, Tuple

from .frame_range import FrameRange

Pair = namedtuple("Pair", ["first", "second"])
Pairs_t = Set[Pair]


# ---------------------------------------------------------------------------- generators over frame indices
def _dyadic(n: int, level: int, stride: int, both_ways: bool) -> Iterator[Tuple[int, int]]:
    reach = 1 << level
    for s in range(0, n, stride):
        if s + reach < n:
            yield s, s + reach
        if both_ways and s - reach >= 0:
            yield s, s - reach


def _dyadic_levels(lo: int, hi: int) -> Iterator[int]:
    if lo < 1:
        raise ValueError("min_dist must be >= 1")
    level = max(0, (lo - 1).bit_length())          # smallest level with 2^level >= lo
    while (1 << level) <= hi:
        yield level
        level += 1


def _hierarchy(n: int, both_ways: bool, lo: int, hi: Optional[int], mid_points: bool) -> Pairs_t:
    hi = n - 1 if hi is None else hi
    found: Pairs_t = set()
    for level in _dyadic_levels(lo, hi):
        stride = 1 << (level - 1 if (mid_points and level > 0) else level)
        found.update(Pair(a, b) for a, b in _dyadic(n, level, stride, both_ways))
    return found


def _all_pairs(n: int, both_ways: bool) -> Pairs_t:
    return {Pair(i, j) for i in range(n) for j in range(n) if (i < j or (both_ways and i != j))}


# ---------------------------------------------------------------------------- the reference's surface
@enum.unique
class SamplePairsMode(enum.Enum):
    EXHAUSTED = 0
    CONSECUTIVE = 1
    HIERARCHICAL = 2
    HIERARCHICAL2 = 3

    @classmethod
    def name_mode_map(cls) -> Dict[str, "SamplePairsMode"]:
        return {member.name.lower(): member for member in cls}

    @classmethod
    def names(cls):
        return list(cls.name_mode_map())


class SamplePairsOptions(NamedTuple):
    mode: SamplePairsMode
    params: Dict[str, Any] = {}


class SamplePairs:
    """Facade with the reference's method names; every method returns a set of index `Pair`s."""

    sample_exhausted = staticmethod(_all_pairs)

    @staticmethod
    def sample_hierarchical(num_frames, two_way, min_dist=1, max_dist=None, include_mid_point=False) -> Pairs_t:
        return _hierarchy(num_frames, two_way, min_dist, max_dist, include_mid_point)

    @staticmethod
    def sample_hierarchical2(num_frames, two_way, min_dist=1, max_dist=None) -> Pairs_t:
        return _hierarchy(num_frames, two_way, min_dist, max_dist, True)

    @staticmethod
    def sample_consecutive(num_frames, two_way) -> Pairs_t:
        return _hierarchy(num_frames, two_way, 1, 1, False)

    @classmethod
    def factory(cls, num_frames: int, opt: SamplePairsOptions, two_way: bool) -> Pairs_t:
        return getattr(cls, "sample_" + opt.mode.name.lower())(num_frames, two_way, **opt.params)

    @classmethod
    def sample(cls, opts: Iterable[SamplePairsOptions], frame_range: FrameRange, two_way=False) -> Pairs_t:
        """Union of the requested samplings over len(frame_range) indices, mapped to frame numbers; a pair is kept when
        at least one of its frames is in the range."""
        by_index: Pairs_t = set()
        for opt in opts:
            by_index |= cls.factory(len(frame_range), opt, two_way)
        number, inside = frame_range.index_to_frame, set(frame_range.frames())
        return {Pair(number[a], number[b]) for a, b in by_index if number[a] in inside or number[b] in inside}

    @staticmethod
    def to_one_way(pairs) -> Pairs_t:
        return {Pair(min(p), max(p)) for p in pairs}


def to_in_range(pairs, frame_range=None):
    if frame_range is None:
        return pairs
    first, stop = frame_range
    return [p for p in pairs if first <= min(p) and max(p) < stop]


def sample_pairs(frame_range: FrameRange, flow_ops):
    """flow_ops names ("hierarchical2", ...) -> two-way pair set, as video.py:18-28 builds it."""
    known = SamplePairsMode.name_mode_map()
    return SamplePairs.sample([SamplePairsOptions(mode=known[name]) for name in flow_ops], frame_range, two_way=True)
