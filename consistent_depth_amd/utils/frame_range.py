"""Which frames of a clip take part: parsing of the `--frame_range` flag and the index <-> frame-number mapping.

The names `NamedOptionalSet`, `OptionalSet`, `FrameRange` and `parse_frame_range` are the ones the reference's module of
the same name exports (/root/reference/utils/frame_range.py:9-107) -- `params`, `process` and the pair sampler address
them.  A flag like "0,2-6,8,21-40" selects frames 0, 2..6, 8, 21..40; the empty string selects every frame.  The parser
validates with a regular expression and canonicalises the name with `itertools.groupby` (consecutive runs collapse to
"a-b", duplicates and order do not matter: "6,5,8,0,2-4,5-6" -> "0,2-6,8")."""
from __future__ import annotations

import itertools
import re
from collections import namedtuple
from typing import Iterable, Optional, Set

NamedOptionalSet = namedtuple("NamedOptionalSet", ["name", "set"])

_PIECE = re.compile(r"^\s*(\d+)\s*(?:-\s*(\d+)\s*)?$")


class OptionalSet:
    """A finite set of frame numbers, or `None` for "all frames" (the neutral element of `intersection`)."""

    __slots__ = ("set",)

    def __init__(self, set: Optional[Set[int]] = None):
        self.set = set

    def intersection(self, other: "OptionalSet") -> "OptionalSet":
        unbounded = [s for s in (self, other) if s.set is None]
        if unbounded:                                   # all & x = x
            return other if self.set is None else self
        return OptionalSet(self.set & other.set)

    def __str__(self) -> str:
        return "all" if self.set is None else str(self.set)

    __repr__ = __str__


class FrameRange:
    """The selected frames in ascending order; position i in that order is "index i" of the pair sampler."""

    def __init__(self, frame_range: OptionalSet, num_frames: Optional[int] = None):
        whole_clip = OptionalSet(None if num_frames is None else set(range(num_frames)))
        self.update(whole_clip.intersection(frame_range))

    def update(self, frame_range: OptionalSet) -> None:
        if frame_range.set is None:
            raise ValueError("FrameRange needs a finite set of frames (give num_frames or an explicit range)")
        self.frame_range = frame_range
        self._ordered = tuple(sorted(frame_range.set))

    @property
    def index_to_frame(self):
        """index -> frame number (a tuple: indexable like the reference's dict)."""
        return self._ordered

    def intersection(self, other: OptionalSet) -> "FrameRange":
        return FrameRange(self.frame_range.intersection(other))

    def frames(self):
        return list(self._ordered)

    def __len__(self) -> int:
        return len(self._ordered)


def _expand(spec: str) -> Iterable[int]:
    for piece in spec.split(","):
        m = _PIECE.match(piece)
        if m is None:
            raise ValueError(f"bad frame range piece '{piece}' in '{spec}' (expected N or A-B)" if "-" not in piece.strip()[:1]
                             else "Frame indices must be positive.")
        first = int(m.group(1))
        last = first if m.group(2) is None else int(m.group(2))
        if last < first:
            raise ValueError(f"bad sub-range '{piece}': end before start")
        yield from range(first, last + 1)


def _canonical(frames: Iterable[int]) -> str:
    ordered = sorted(set(frames))
    runs = []
    for _, grp in itertools.groupby(enumerate(ordered), key=lambda t: t[1] - t[0]):   # equal (value - position) = one run
        run = [v for _, v in grp]
        runs.append(str(run[0]) if len(run) == 1 else f"{run[0]}-{run[-1]}")
    return ",".join(runs)


def parse_frame_range(frame_range_str: str) -> NamedOptionalSet:
    if not frame_range_str:
        return NamedOptionalSet(name=frame_range_str, set=OptionalSet())
    frames = set(_expand(frame_range_str))
    return NamedOptionalSet(name=_canonical(frames), set=OptionalSet(frames))
