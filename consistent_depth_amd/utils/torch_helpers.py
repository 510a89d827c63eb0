"""Device plumbing (mirrors /root/reference/utils/torch_helpers.py:7-23: module-level `_device`,
recursive `to_device` that mutates dicts/lists in place).  There is no CPU branch: the engine
only runs on the HIP device."""
import torch


def _current_device():
    if not torch.cuda.is_available():
        raise RuntimeError("consistent_depth_amd needs a HIP device (torch.cuda.is_available() is False)")
    return torch.device("cuda", torch.cuda.current_device())


class _LazyDevice:
    """Resolves to the current HIP device at use time (the reference freezes it at import)."""

    def __call__(self):
        return _current_device()


_device = _LazyDevice()


def to_device(data):
    if isinstance(data, torch.Tensor):
        return data.to(_current_device(), non_blocking=True)
    if isinstance(data, dict):
        for k in data:
            data[k] = to_device(data[k])
        return data
    if isinstance(data, (list, tuple)):
        moved = [to_device(v) for v in data]
        if isinstance(data, list):
            data[:] = moved
            return data
        return type(data)(moved)
    return data
