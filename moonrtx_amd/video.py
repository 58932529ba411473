"""Video export without NVENC: Motion-JPEG in an AVI (RIFF) container, written frame by frame.

The reference exports time-lapse videos through PlotOptiX's NVENC H.264 encoder (renderer_video.py:219-260: `encoder_create(fps,
bitrate)`, `encoder_start(filename, n_frames)`, one frame captured per finished accumulation cycle, `encoder_stop()`).  There is
no NVENC on an MI355X node and no FFmpeg in this image, so the facade's encoder (tkoptix.TkOptiX.encoder_*) writes the one video
format that needs nothing but a JPEG coder: every frame is an independent JPEG (`00dc` chunk), the index (`idx1`) is appended on
close -- mpv, VLC, FFmpeg and every NLE open it.  The file is larger than an H.264 stream of the same quality (no inter-frame
prediction); `bitrate` is honoured as a per-frame byte budget as far as the JPEG quality range allows (see `pick_quality`).

Not on the hot path ("further out"): host code only, Pillow for the JPEG coding; fails loudly without it.
"""
import io
import os
import struct

import numpy as np

MAX_RIFF = (1 << 32) - (1 << 20)      # a RIFF chunk holds 4 GiB; OpenDML extensions are not written, the writer stops before that


def _jpeg(rgb, quality):
    try:
        from PIL import Image
    except ImportError as e:                                   # pragma: no cover - Pillow is part of the image
        raise RuntimeError("video export needs Pillow for the JPEG coding (no NVENC / FFmpeg on this backend)") from e
    h, w, ch = rgb.shape
    if ch == 4:       # the facade's RGBA8 image as it is: Pillow drops the fourth byte in C (a strided numpy copy costs 3x a frame's coding)
        im = Image.frombuffer("RGBX", (w, h), rgb, "raw", "RGBX", 0, 1).convert("RGB")
    else:
        im = Image.frombuffer("RGB", (w, h), rgb, "raw", "RGB", 0, 1)
    buf = io.BytesIO()
    im.save(buf, format="JPEG", quality=int(quality), subsampling="4:2:0", optimize=False)
    return buf.getvalue()


def pick_quality(rgb, budget_bytes, q_min=70, q_max=95):
    """The highest JPEG quality in [q_min, q_max] whose coding of `rgb` fits `budget_bytes` (bisection, <= 5 codings); q_min when
    even that does not fit -- an intra-only stream cannot reach an H.264 bitrate, and a blocky Moon is worth less than a larger file."""
    if len(_jpeg(rgb, q_max)) <= budget_bytes:
        return q_max
    lo, hi = q_min, q_max                 # invariant: hi does not fit
    if len(_jpeg(rgb, lo)) > budget_bytes:
        return q_min
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if len(_jpeg(rgb, mid)) <= budget_bytes:
            lo = mid
        else:
            hi = mid
    return lo


class MjpegAviWriter:
    """AVI 1.0, one `vids`/`MJPG` stream.  add_frame(rgb_or_rgba uint8 [H, W, 3|4]) appends; close() patches the headers and writes
    the index.  `n_frames` > 0 closes the file by itself after that many frames (PlotOptiX's encoder_start contract)."""

    def __init__(self, path, width, height, fps, bitrate_mbps=16.0, n_frames=0, q_min=None, q_max=95, workers=None):
        """`workers` JPEG coders run beside the caller (Pillow releases the GIL while it codes; default MOONRT_MJPEG_WORKERS or 3;
        0 = code inside add_frame): a 4K frame takes ~30 ms to code against ~21 ms to render at 64 spp, so add_frame only copies
        the RGB planes and hands them over; chunks are written in frame order, at most 2 x workers frames are in flight."""
        if width <= 0 or height <= 0 or fps <= 0:
            raise ValueError("width, height and fps must be positive")
        self.path, self.width, self.height = path, int(width), int(height)
        self.rate, self.scale = (int(fps), 1) if float(fps).is_integer() else (int(round(fps * 1000)), 1000)
        self.budget = max(1, int(bitrate_mbps * 1e6 / 8.0 * self.scale / self.rate))
        self.q_min = int(os.environ.get("MOONRT_MJPEG_MIN_QUALITY", "70")) if q_min is None else int(q_min)
        self.q_max = max(self.q_min, int(q_max))
        self.quality = None
        self.limit = int(n_frames)
        self.index = []                    # (offset from the 'movi' fourcc, size) per frame
        self.max_chunk = 0
        self.f = open(path, "wb")
        self._write_headers(0)
        self.movi_at = self.f.tell() - 4   # position of the 'movi' fourcc
        self.open = True
        self.submitted = 0
        self.pending = []                  # futures of frames not yet written, in frame order
        n_workers = int(os.environ.get("MOONRT_MJPEG_WORKERS", "3")) if workers is None else int(workers)
        self.pool = None
        self.max_pending = 0
        if n_workers > 0:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=n_workers, thread_name_prefix="moonrt-mjpeg")
            self.max_pending = 2 * n_workers

    # ---- layout: RIFF('AVI ' LIST('hdrl' avih LIST('strl' strh strf)) LIST('movi' 00dc...) idx1)
    def _write_headers(self, frames, riff_size=0, movi_size=4):
        usec = int(round(1e6 * self.scale / self.rate))
        avih = struct.pack("<14I", usec, self.max_chunk * self.rate // self.scale, 0, 0x10, frames, 0, 1, self.max_chunk,
                           self.width, self.height, 0, 0, 0, 0)
        strh = struct.pack("<4s4sIHHIIIIIIII4h", b"vids", b"MJPG", 0, 0, 0, 0, self.scale, self.rate, 0, frames, self.max_chunk,
                           0xFFFFFFFF, 0, 0, 0, self.width, self.height)
        strf = struct.pack("<IiiHH4sIiiII", 40, self.width, self.height, 1, 24, b"MJPG", self.width * self.height * 3, 0, 0, 0, 0)
        strl = b"strl" + b"strh" + struct.pack("<I", len(strh)) + strh + b"strf" + struct.pack("<I", len(strf)) + strf
        hdrl = b"hdrl" + b"avih" + struct.pack("<I", len(avih)) + avih + b"LIST" + struct.pack("<I", len(strl)) + strl
        self.f.seek(0)
        self.f.write(b"RIFF" + struct.pack("<I", riff_size) + b"AVI ")
        self.f.write(b"LIST" + struct.pack("<I", len(hdrl)) + hdrl)
        self.f.write(b"LIST" + struct.pack("<I", movi_size) + b"movi")

    def add_frame(self, image, copy=True):
        """`copy=False`: the caller hands the array over and does not touch it until max_pending + 1 further frames have been added
        (the facade rotates its read-back buffers instead of copying 33 MB per 4K frame)."""
        if not self.open:
            raise RuntimeError("encoder is closed")
        a = np.asarray(image)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[0] != self.height or a.shape[1] != self.width or a.shape[2] not in (3, 4):
            raise ValueError(f"frame must be uint8 [{self.height}, {self.width}, 3|4], got {a.dtype} {a.shape}")
        rgb = a.copy() if (copy or not a.flags.c_contiguous) else a
        if self.quality is None:
            self.quality = pick_quality(rgb, self.budget, self.q_min, self.q_max)
        self.submitted += 1
        if self.pool is None:
            self._write_chunk(_jpeg(rgb, self.quality))
        else:
            self.pending.append(self.pool.submit(_jpeg, rgb, self.quality))
            self._drain(block_above=self.max_pending)
        if self.limit > 0 and self.submitted >= self.limit:
            self.close()

    def _write_chunk(self, data):
        pad = len(data) & 1
        if self.f.tell() + 8 + len(data) + pad + 16 * (len(self.index) + 2) > MAX_RIFF:
            for fut in self.pending:
                fut.cancel()
            self.pending = []
            self._finish()
            raise RuntimeError("AVI file would exceed 4 GiB: closed after %d frames" % len(self.index))
        self.index.append((self.f.tell() - self.movi_at, len(data)))
        self.f.write(b"00dc" + struct.pack("<I", len(data)) + data + b"\0" * pad)
        self.max_chunk = max(self.max_chunk, len(data))

    def _drain(self, block_above=0):
        """Write the coded frames at the head of the queue; wait while more than `block_above` are in flight."""
        while self.pending and (self.pending[0].done() or len(self.pending) > block_above):
            self._write_chunk(self.pending.pop(0).result())

    @property
    def frames(self):
        """Frames accepted so far (PlotOptiX's encoded_frames); all of them are in the file once close() has returned."""
        return self.submitted

    def close(self):
        if not self.open:
            return
        self._drain(block_above=0)
        self._finish()

    def _finish(self):
        if not self.open:
            return
        self.open = False
        if self.pool is not None:
            self.pool.shutdown(wait=True)
        end_movi = self.f.tell()
        idx = b"".join(struct.pack("<4sIII", b"00dc", 0x10, off, size) for off, size in self.index)
        self.f.write(b"idx1" + struct.pack("<I", len(idx)) + idx)
        total = self.f.tell()
        self._write_headers(len(self.index), riff_size=total - 8, movi_size=end_movi - self.movi_at)
        self.f.close()


def read_avi_frames(path):
    """Minimal reader for the files MjpegAviWriter writes (tests and tools): header fields + the JPEG payloads, via the index."""
    b = open(path, "rb").read()
    if b[:4] != b"RIFF" or b[8:12] != b"AVI ":
        raise ValueError("not an AVI file")
    info, pos, movi = {"riff_size": struct.unpack_from("<I", b, 4)[0]}, 12, None
    frames = []
    while pos + 8 <= len(b):
        cc, size = b[pos:pos + 4], struct.unpack_from("<I", b, pos + 4)[0]
        if cc == b"LIST" and b[pos + 8:pos + 12] == b"hdrl":
            avih = struct.unpack_from("<14I", b, pos + 20)
            info.update(usec_per_frame=avih[0], total_frames=avih[4], streams=avih[6], width=avih[8], height=avih[9])
            strh = pos + 20 + 56 + 12 + 8
            info.update(handler=b[strh + 4:strh + 8], scale=struct.unpack_from("<I", b, strh + 20)[0],
                        rate=struct.unpack_from("<I", b, strh + 24)[0], length=struct.unpack_from("<I", b, strh + 32)[0])
        elif cc == b"LIST" and b[pos + 8:pos + 12] == b"movi":
            movi = pos + 8
        elif cc == b"idx1":
            for i in range(size // 16):
                ck, flags, off, sz = struct.unpack_from("<4sIII", b, pos + 8 + 16 * i)
                at = movi + off
                if b[at:at + 4] != ck or struct.unpack_from("<I", b, at + 4)[0] != sz:
                    raise ValueError("index entry %d does not point at its chunk" % i)
                frames.append(b[at + 8:at + 8 + sz])
        pos += 8 + size + (size & 1)
    info["file_size"] = len(b)
    return info, frames
