"""Off-critical-path `save_last_stitch` (reference stitcher.py:168-174 writes blend_N.png with
cv2.imwrite synchronously, which stalls a live stream for the D2H copy and the PNG encode of the whole
mosaic).

AsyncImageWriter.submit() snapshots a resident mosaic into pinned host memory on a side HIP stream (the
stitching stream only records an event) and hands it to ONE worker thread that waits for the copy, applies
the user's blend_processor and encodes the file.  One worker = files appear in submission order; flush()
returns when everything submitted so far is on disk.  Resident mosaics are blended in place by later stitches,
so the stitching stream is made to wait (GPU side) for the snapshot copy: the file equals what the synchronous
path would have written.
"""
import atexit
import os
import queue
import threading

import numpy as np
import torch


def write_image(filepath, array):
    """cv2.imwrite(filepath, array) (BGR) when cv2 is importable, else the same file through PIL."""
    try:
        import cv2
        cv2.imwrite(filepath, array)
    except ImportError:
        from PIL import Image
        a = np.asarray(array)
        Image.fromarray(a[..., ::-1] if a.ndim == 3 else a).save(filepath)


class AsyncImageWriter:
    def __init__(self, device):
        self.device = device
        self._stream = torch.cuda.Stream(device=device)
        self._q = queue.Queue()
        self._free = []                # pinned staging buffers (flat uint8), reused: hipHostMalloc is slow
        self._lock = threading.Lock()
        self._errors = []
        self._worker = threading.Thread(target=self._run, name="isx-async-save", daemon=True)
        self._worker.start()
        atexit.register(self.flush)

    def _staging(self, nbytes):
        with self._lock:
            for i, b in enumerate(self._free):
                if b.numel() >= nbytes:
                    return self._free.pop(i)
        return torch.empty((max(nbytes, 1 << 20),), dtype=torch.uint8, pin_memory=True)

    def submit(self, image, filepath, processor=None):
        """image: DeviceImage / uint8 device tensor / ndarray.  Returns at once."""
        t = getattr(image, "t", image)
        if isinstance(t, torch.Tensor) and t.is_cuda:
            cur = torch.cuda.current_stream(t.device)
            self._stream.wait_stream(cur)
            buf = self._staging(t.numel())
            with torch.cuda.stream(self._stream):
                host = buf[:t.numel()].view(t.shape)
                host.copy_(t, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._stream)
            t.record_stream(self._stream)
            # resident mosaics are blended in place by the next stitch: later work on the stitching stream
            # waits (on the GPU, not the host) until the snapshot has been taken
            cur.wait_stream(self._stream)
            self._q.put((ev, buf, host, filepath, processor))
        else:
            self._q.put((None, None, np.array(np.asarray(t), copy=True), filepath, processor))

    def _run(self):
        while True:
            ev, buf, host, filepath, processor = self._q.get()
            try:
                if ev is not None:
                    ev.synchronize()
                    host = host.numpy()
                out = host if processor is None else processor(host)
                d = os.path.dirname(filepath)
                if d and not os.path.exists(d):
                    os.makedirs(d, exist_ok=True)
                write_image(filepath, out)
            except Exception as e:  # surfaced by flush()
                self._errors.append((filepath, e))
            finally:
                if buf is not None:
                    with self._lock:
                        self._free.append(buf)
                self._q.task_done()

    def flush(self):
        self._q.join()
        if self._errors:
            filepath, e = self._errors.pop(0)
            self._errors.clear()
            raise RuntimeError(f"asynchronous save of {filepath} failed") from e
