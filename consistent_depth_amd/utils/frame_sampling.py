"""Frame-pair sampling (consecutive / hierarchical / hierarchical2 / exhausted).
Same public surface as /root/reference/utils/frame_sampling.py:12-155 (SamplePairsMode,
SamplePairsOptions, Pair, SamplePairs.sample/to_one_way, to_in_range).

hierarchical: for every level l with 2^l in [min_dist, max_dist], pair (s, s + 2^l) for s stepping
by 2^l; hierarchical2 ("include_mid_point") steps by 2^(l-1) instead.  244 frames -> 715 one-way
pairs, 1000 frames -> 2979.  `sample_exhausted` is broken in the
reference (TypeError); here it works.
"""
from __future__ import annotations

from collections import namedtuple
from enum import Enum, auto, unique
from typing import Any, Dict, Iterable, NamedTuple, Set

from .frame_range import FrameRange


@unique
class SamplePairsMode(Enum):
    EXHAUSTED = 0
    CONSECUTIVE = auto()
    HIERARCHICAL = auto()
    HIERARCHICAL2 = auto()

    @classmethod
    def name_mode_map(cls):
        return {m.name.lower(): m for m in cls}

    @classmethod
    def names(cls):
        return [m.name.lower() for m in cls]


class SamplePairsOptions(NamedTuple):
    mode: SamplePairsMode
    params: Dict[str, Any] = {}


Pair = namedtuple("Pair", ["first", "second"])
Pairs_t = Set[Pair]


def _levels(min_dist: int, max_dist: int):
    level = 0
    while (1 << level) < min_dist:
        level += 1
    while (1 << level) <= max_dist:
        yield level
        level += 1


class SamplePairs:
    @classmethod
    def sample(cls, opts: Iterable[SamplePairsOptions], frame_range: FrameRange, two_way=False) -> Pairs_t:
        n = len(frame_range)
        rel = set()
        for opt in opts:
            rel |= cls.factory(n, opt, two_way)
        wanted = set(frame_range.frames())
        to_frame = frame_range.index_to_frame
        return {Pair(to_frame[a], to_frame[b]) for a, b in rel if to_frame[a] in wanted or to_frame[b] in wanted}

    @classmethod
    def factory(cls, num_frames: int, opt: SamplePairsOptions, two_way: bool) -> Pairs_t:
        table = {
            SamplePairsMode.EXHAUSTED: cls.sample_exhausted,
            SamplePairsMode.CONSECUTIVE: cls.sample_consecutive,
            SamplePairsMode.HIERARCHICAL: cls.sample_hierarchical,
            SamplePairsMode.HIERARCHICAL2: cls.sample_hierarchical2,
        }
        return table[opt.mode](num_frames, two_way, **opt.params)

    @staticmethod
    def sample_hierarchical(num_frames: int, two_way: bool, min_dist=1, max_dist=None,
                            include_mid_point=False) -> Pairs_t:
        if min_dist < 1:
            raise ValueError("min_dist must be >= 1")
        if max_dist is None:
            max_dist = num_frames - 1
        pairs = set()
        for level in _levels(min_dist, max_dist):
            dist = 1 << level
            step = 1 << (max(0, level - 1) if include_mid_point else level)
            for start in range(0, num_frames, step):
                for end in ((start - dist, start + dist) if two_way else (start + dist,)):
                    if 0 <= end < num_frames:
                        pairs.add(Pair(start, end))
        return pairs

    @classmethod
    def sample_hierarchical2(cls, num_frames: int, two_way: bool, min_dist=1, max_dist=None) -> Pairs_t:
        return cls.sample_hierarchical(num_frames, two_way, min_dist, max_dist, include_mid_point=True)

    @classmethod
    def sample_consecutive(cls, num_frames: int, two_way: bool) -> Pairs_t:
        return cls.sample_hierarchical(num_frames, two_way, min_dist=1, max_dist=1)

    @staticmethod
    def sample_exhausted(num_frames: int, two_way: bool) -> Pairs_t:
        return {Pair(i, j) for i in range(num_frames) for j in range(num_frames)
                if i != j and (two_way or i < j)}

    @classmethod
    def to_one_way(cls, pairs) -> Pairs_t:
        return {Pair(*sorted(p)) for p in pairs}


def to_in_range(pairs, frame_range=None):
    if frame_range is None:
        return pairs
    lo, hi = frame_range
    return [p for p in pairs if all(lo <= i < hi for i in p)]


def sample_pairs(frame_range: FrameRange, flow_ops):
    """flow_ops names -> two-way pair set (reference: video.py:18-28)."""
    modes = SamplePairsMode.name_mode_map()
    return SamplePairs.sample([SamplePairsOptions(mode=modes[op]) for op in flow_ops], frame_range, two_way=True)
