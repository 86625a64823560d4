"""PNG output without torchvision (absent from the image): save_image() quantises exactly like torchvision.utils.save_image
for a single CHW image -- x * 255 + 0.5, clamped, truncated to uint8 -- and writes it with PIL."""
import numpy as np
import torch


def to_uint8_hwc(img):
    """[C,H,W] float tensor (any device) -> [H,W,C] uint8 tensor on the same device."""
    return img.detach().mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8)


def encode_png(arr_hwc, compress_level=6):
    """A complete PNG file (8-bit grey / RGB / RGBA, filter type 0 on every row) as bytes.  Everything heavy -- zlib.compress,
    zlib.crc32 -- is C code that releases the GIL, so many frames encode in parallel on a thread pool (PIL's encoder holds it)."""
    import struct
    import zlib
    a = np.ascontiguousarray(arr_hwc, dtype=np.uint8)
    H, W, C = a.shape
    color_type = {1: 0, 3: 2, 4: 6}[C]
    rows = np.empty((H, 1 + W * C), dtype=np.uint8)
    rows[:, 0] = 0
    rows[:, 1:] = a.reshape(H, W * C)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(data, zlib.crc32(tag)) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, color_type, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(rows.tobytes(), compress_level)) + chunk(b"IEND", b""))


def save_uint8(arr_hwc, path, compress_level=6):
    with open(path, "wb") as fh:
        fh.write(encode_png(arr_hwc, compress_level))


def save_image(img, path):
    save_uint8(to_uint8_hwc(img).cpu().numpy(), path)
