"""`.raw` float32 image codec -- the on-disk contract of the hot path's inputs (colour, flow)
and outputs (inverse depth).  Same byte format as /root/reference/utils/image_io.py:101-169:

    int32 h | int32 w | int32 cv_type = 5 + ((d-1) << 3) | uint64 pixel_size = 4*d | h*w*d float32 (row-major HWC)

Pinned by tests/golden/raw_codec.npz (bytes written by the reference itself).
"""
from __future__ import annotations

import struct

import numpy as np

_HEADER = struct.Struct("<iiiQ")  # 20 bytes, no padding
_CV_32F, _CV_CN_SHIFT, _CV_CN_MAX = 5, 3, 512


def load_raw_float32_image(file_name: str) -> np.ndarray:
    with open(file_name, "rb") as f:
        head = f.read(_HEADER.size)
        if len(head) != _HEADER.size:
            raise IOError(f"{file_name}: truncated header")
        h, w, cv_type, pixel_size = _HEADER.unpack(head)
        d = ((cv_type - _CV_32F) >> _CV_CN_SHIFT) + 1
        if d < 1 or d > _CV_CN_MAX or pixel_size != 4 * d:
            raise ValueError(f"{file_name}: inconsistent header (cv_type={cv_type}, pixel_size={pixel_size})")
        data = np.fromfile(f, dtype=np.float32, count=h * w * d)
    if data.size != h * w * d:
        raise IOError(f"{file_name}: expected {h * w * d} floats, found {data.size}")
    return data.reshape((h, w) if d == 1 else (h, w, d))


def save_raw_float32_image(file_name: str, image: np.ndarray) -> None:
    image = np.asarray(image)
    if image.ndim == 2:
        h, w, d = image.shape[0], image.shape[1], 1
    elif image.ndim == 3:
        h, w, d = image.shape
    else:
        raise ValueError(f"expected (H,W) or (H,W,C), got {image.shape}")
    if d > _CV_CN_MAX:
        raise ValueError("more than 512 channels")
    with open(file_name, "wb") as f:
        f.write(_HEADER.pack(h, w, _CV_32F + ((d - 1) << _CV_CN_SHIFT), 4 * d))
        np.ascontiguousarray(image, dtype=np.float32).tofile(f)


def load_mask_png(file_name: str) -> np.ndarray:
    """8-bit mask PNG -> (H,W) bool, `> 0` like loaders/video_dataset.py:71-77."""
    from PIL import Image
    with Image.open(file_name) as im:
        a = np.asarray(im)
    if a.ndim == 3:
        a = a[..., 0]
    return a > 0


def save_mask_png(file_name: str, mask: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray((np.asarray(mask) > 0).astype(np.uint8) * 255).save(file_name)


class AsyncRawWriter:
    """Writes device tensors as `.raw` float32 images OFF the critical path: `submit` enqueues a device-to-pinned-host copy on
    a side stream (ordered after the producing stream by an event) and returns at once; a background thread waits for each
    copy and writes the file.  `close()` (or leaving the `with` block) drains the queue and re-raises a writer error.
    Replaces the reference's blocking per-frame `.cpu().numpy()` + write (depth_fine_tuning.py:185-199)."""

    def __init__(self, device=None, max_pending: int = 64):
        import queue
        import threading
        import torch
        self._torch = torch
        self._stream = torch.cuda.Stream(device=device)
        self._q = queue.Queue(maxsize=max_pending)
        self._err = None
        self._thread = threading.Thread(target=self._run, name="cd-raw-writer", daemon=True)
        self._thread.start()

    def submit(self, file_name: str, image) -> None:
        torch = self._torch
        if self._err is not None:
            raise self._err
        src = image.detach()
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(src.device))
        host = torch.empty(src.shape, dtype=torch.float32, pin_memory=True)
        with torch.cuda.stream(self._stream):
            self._stream.wait_event(ready)
            host.copy_(src, non_blocking=True)
            src.record_stream(self._stream)      # keep the device tensor alive until the copy has run
            done = torch.cuda.Event()
            done.record(self._stream)
        self._q.put((file_name, host, done))

    def _run(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            try:
                file_name, host, done = item
                done.synchronize()
                save_raw_float32_image(file_name, host.numpy())
            except Exception as e:   # noqa: BLE001 -- surfaced by submit / close
                self._err = e

    def close(self):
        self._q.put(None)
        self._thread.join()
        if self._err is not None:
            raise self._err

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
