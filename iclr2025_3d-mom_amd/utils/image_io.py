"""PNG output without torchvision (absent from the image): save_image() quantises exactly like torchvision.utils.save_image
for a single CHW image -- x * 255 + 0.5, clamped, truncated to uint8 -- and writes it with PIL."""
import numpy as np
import torch


def to_uint8_hwc(img):
    """[C,H,W] float tensor (any device) -> [H,W,C] uint8 tensor on the same device."""
    return img.detach().mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8)


def _png_parts(arr_hwc, compress_level):
    """(signature + IHDR, IDAT header, compressed scanlines, IDAT crc, IEND) of an 8-bit grey / RGB / RGBA PNG with filter type 0 on
    every row.  Everything heavy -- zlib.compress, zlib.crc32 -- is C code that releases the GIL, so many frames encode in parallel
    on a thread pool (PIL's encoder holds it); what holds the GIL is one strided copy into the scanline layout: the scanlines go to
    zlib through the buffer protocol and the parts to the file one by one, not through tobytes() and concatenations (three more
    copies of the frame with the GIL held; removing them did not move the asynchronous writer's 700-1470 frames/s, which follow the box's
    file system and cores, but there is no reason to keep them)."""
    import struct
    import zlib
    a = np.ascontiguousarray(arr_hwc, dtype=np.uint8)
    H, W, C = a.shape
    color_type = {1: 0, 3: 2, 4: 6}[C]
    rows = np.empty((H, 1 + W * C), dtype=np.uint8)
    rows[:, 0] = 0
    rows[:, 1:] = a.reshape(H, W * C)
    comp = zlib.compress(rows, compress_level)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(data, zlib.crc32(tag)) & 0xFFFFFFFF)
    head = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, color_type, 0, 0, 0))
    return (head, struct.pack(">I", len(comp)) + b"IDAT", comp, struct.pack(">I", zlib.crc32(comp, zlib.crc32(b"IDAT")) & 0xFFFFFFFF),
            chunk(b"IEND", b""))


def encode_png(arr_hwc, compress_level=6):
    """A complete PNG file as bytes (see _png_parts)."""
    return b"".join(_png_parts(arr_hwc, compress_level))


def save_uint8(arr_hwc, path, compress_level=6):
    parts = _png_parts(arr_hwc, compress_level)
    with open(path, "wb") as fh:
        for part in parts:
            fh.write(part)


def save_image(img, path):
    save_uint8(to_uint8_hwc(img).cpu().numpy(), path)
