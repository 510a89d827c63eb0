"""Frame-range parsing ("0,2-10,21-40") and index<->frame mapping.
Same public names as /root/reference/utils/frame_range.py:9-107 (NamedOptionalSet, OptionalSet,
FrameRange, parse_frame_range) because params.py and the pair sampler consume them."""
from __future__ import annotations

from collections import namedtuple
from typing import Optional, Set

NamedOptionalSet = namedtuple("NamedOptionalSet", ["name", "set"])


class OptionalSet:
    """A set, or None meaning "everything"."""

    def __init__(self, set: Optional[Set] = None):
        self.set = set

    def intersection(self, other: "OptionalSet") -> "OptionalSet":
        if self.set is None:
            return other
        if other.set is None:
            return self
        return OptionalSet(self.set & other.set)

    def __str__(self):
        return str(self.set)


class FrameRange:
    def __init__(self, frame_range: OptionalSet, num_frames: int = None):
        everything = OptionalSet(set(range(num_frames)) if num_frames is not None else None)
        self.update(frame_range.intersection(everything))

    def intersection(self, other: OptionalSet) -> "FrameRange":
        return FrameRange(self.frame_range.intersection(other))

    def update(self, frame_range: OptionalSet):
        if frame_range.set is None:
            raise ValueError("FrameRange needs a finite set of frames")
        self.frame_range = frame_range
        self.index_to_frame = dict(enumerate(sorted(frame_range.set)))

    def frames(self):
        return sorted(self.index_to_frame.values())

    def __len__(self):
        return len(self.index_to_frame)


def _compact_name(frames) -> str:
    """sorted unique ints -> "0,2-6,8-10"."""
    runs, start, prev = [], None, None
    for f in frames:
        if start is None:
            start = prev = f
        elif f == prev + 1:
            prev = f
        else:
            runs.append((start, prev))
            start = prev = f
    runs.append((start, prev))
    return ",".join(str(a) if a == b else f"{a}-{b}" for a, b in runs)


def parse_frame_range(frame_range_str: str) -> NamedOptionalSet:
    if not frame_range_str:
        return NamedOptionalSet(name=frame_range_str, set=OptionalSet())
    frames = set()
    for part in frame_range_str.split(","):
        bounds = [int(s) for s in part.split("-", 1)]
        if len(bounds) == 1:
            frames.add(bounds[0])
        else:
            lo, hi = bounds
            if lo > hi:
                raise ValueError(f"bad sub-range '{part}'")
            frames.update(range(lo, hi + 1))
    if min(frames) < 0:
        raise ValueError("Frame indices must be positive.")
    return NamedOptionalSet(name=_compact_name(sorted(frames)), set=OptionalSet(frames))
