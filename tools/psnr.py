#!/usr/bin/env python3
"""PSNR between two images written by Renderer::save / acr_app (.pfm float RGB, or .ppm 8-bit):
    python tools/psnr.py a.pfm b.pfm      -> PSNR in dB against a peak of 1.0 (inf for identical images), and the
                                            number of differing pixels"""
import sys
import numpy as np


def read(path):
    with open(path, "rb") as f:
        magic = f.readline().strip()
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        if magic == b"PF":
            img = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, 3)[::-1]
            return img.astype(np.float64)
        if magic == b"P6":
            return np.frombuffer(f.read(), dtype=np.uint8).reshape(h, w, 3).astype(np.float64) / 255.0
    raise SystemExit(f"{path}: not a PF or P6 image")


def psnr(a, b):
    mse = float(np.mean((a - b) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(1.0 / mse)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    a, b = read(sys.argv[1]), read(sys.argv[2])
    if a.shape != b.shape:
        raise SystemExit(f"sizes differ: {a.shape} vs {b.shape}")
    print("PSNR %.2f dB, %d of %d pixels differ" % (psnr(a, b), int((a != b).any(axis=-1).sum()), a.shape[0] * a.shape[1]))
