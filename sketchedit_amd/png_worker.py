"""Encoder side of the I/O pipeline (sketchedit_amd/pipeline.py): the PNG writers, and the encoder PROCESS
(`python -m sketchedit_amd.png_worker`).  Deliberately light -- numpy + PIL + zlib, no torch, no HIP -- so that a worker starts
in a fraction of a second.  A worker never receives pixel data through a pipe: the device-to-host copy lands in a ring of
files on /dev/shm that the workers map read-only; a job is one JSON line on stdin, its completion one line on stdout.

Two writers (`--png_writer` of test.py):
  pil   PIL's Image.save defaults (adaptive row filters, zlib level 6): the files this repo has always written.
  fast  what the REFERENCE's writer does -- /root/reference/test.py:37 calls cv2.imwrite, whose PNG defaults (OpenCV 4.5,
        environment.yml:16) are the SUB row filter, zlib level 1 (Z_BEST_SPEED) and the Z_RLE strategy -- written out directly
        (zlib + numpy): 2.5 ms instead of 18 ms per 256x256 RGB image.  Same pixels on decode, valid PNG, files 13 % larger.
"""
import json
import os
import struct
import sys
import time
import zlib

import numpy as np
from PIL import Image

_SIG = b"\x89PNG\r\n\x1a\n"


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def png_bytes_fast(a):
    """(H,W,3) RGB or (H,W) grey uint8 -> PNG file bytes: filter type 1 (Sub) on every row, deflate level 1, strategy RLE"""
    a = np.ascontiguousarray(a, dtype=np.uint8)
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    if c not in (1, 3):
        raise ValueError("png_bytes_fast: grey or RGB only")
    rows = a.reshape(h, w * c)
    f = np.empty((h, 1 + w * c), np.uint8)
    f[:, 0] = 1
    f[:, 1:1 + c] = rows[:, :c]
    np.subtract(rows[:, c:], rows[:, :-c], out=f[:, 1 + c:])          # uint8 arithmetic wraps modulo 256, as the filter is defined
    z = zlib.compressobj(1, zlib.DEFLATED, 15, 8, zlib.Z_RLE)
    data = z.compress(f.tobytes()) + z.flush()
    return _SIG + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0 if c == 1 else 2, 0, 0, 0)) + _chunk(b"IDAT", data) + _chunk(b"IEND", b"")


def save_png(array, path, writer="pil"):
    """what /root/reference/test.py:35-37 does for one image (RGB order on disk, as cv2.imwrite(output[:, :, ::-1]) stores it)"""
    if writer == "fast":
        with open(path, "wb") as f:
            f.write(png_bytes_fast(array))
    else:
        Image.fromarray(array).save(path)


_maps = {}


def _ring(path, shape):
    key = (path, tuple(shape))
    m = _maps.get(key)
    if m is None:
        m = _maps[key] = np.memmap(path, dtype=np.uint8, mode="r", shape=tuple(shape))
    return m


def encode_from_ring(job):
    """job: dict(rgb_ring, rgb_shape, mask_ring, mask_shape, slot, first, paths, out_dir, mask_dir, writer) -> seconds spent"""
    t0 = time.perf_counter()
    rgb = _ring(job["rgb_ring"], job["rgb_shape"])[job["slot"]]
    m8 = _ring(job["mask_ring"], job["mask_shape"])[job["slot"]] if job.get("mask_ring") else None
    for j, p in enumerate(job["paths"]):
        save_png(np.asarray(rgb[job["first"] + j]), os.path.join(job["out_dir"], p), job.get("writer", "pil"))
        if m8 is not None:
            save_png(np.asarray(m8[job["first"] + j]), os.path.join(job["mask_dir"], p), job.get("writer", "pil"))
    return time.perf_counter() - t0


_wmaps = {}


def _ring_rw(path, shape):
    key = (path, tuple(shape))
    m = _wmaps.get(key)
    if m is None:
        m = _wmaps[key] = np.memmap(path, dtype=np.uint8, mode="r+", shape=tuple(shape))
    return m


def decode_into_ring(job):
    """job: dict(img_ring, img_shape, sk_ring, sk_shape, slot, first, image_paths, mask_paths): decode pairs as the dataset does
    (/root/reference/data/testimage_dataset.py:89-111: RGB; sketch 'L' resized to the image) straight into slot `slot` of the
    page-locked input rings, images first .. first + n - 1 -> seconds spent.  The ring fixes the batch's image size."""
    t0 = time.perf_counter()
    img = _ring_rw(job["img_ring"], job["img_shape"])[job["slot"]]
    sk = _ring_rw(job["sk_ring"], job["sk_shape"])[job["slot"]]
    H, W = img.shape[1:3]
    for j, (ip, mp) in enumerate(zip(job["image_paths"], job["mask_paths"])):
        image = Image.open(ip).convert("RGB")
        w, h = image.size
        if (h, w) != (H, W):
            raise ValueError("%s is %dx%d, the batch is %dx%d (images of one batch must have one size)" % (ip, h, w, H, W))
        img[job["first"] + j] = np.asarray(image, dtype=np.uint8)
        sk[job["first"] + j] = np.asarray(Image.open(mp).convert("L").resize((w, h)), dtype=np.uint8)
    return time.perf_counter() - t0


def main():
    out = sys.stdout
    out.write("ready %d\n" % os.getpid())
    out.flush()
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        job = json.loads(line)
        try:
            out.write("done %d %.6f\n" % (job["id"], decode_into_ring(job) if job.get("kind") == "decode" else encode_from_ring(job)))
        except Exception as e:      # noqa: BLE001  (reported to the parent, which raises it in the caller's thread)
            out.write("fail %d %s\n" % (job["id"], json.dumps("%s: %s" % (type(e).__name__, e))))
        out.flush()


if __name__ == "__main__":
    main()
