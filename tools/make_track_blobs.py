#!/usr/bin/env python3
"""Derive track blobs (wall bitmap + 100-point centre-line) from a template directory.

Usage: python tools/make_track_blobs.py /path/to/template [names...]
Writes ft_grandprix_amd/assets/<name>.npz.  Only derived data is stored (packed
wall bits, sampled centre-line, frame constants) -- no PNG/SVG text is copied.
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ft_grandprix_amd.track import load_track_from_template, bundled_track_path

def main():
    template = sys.argv[1]
    names = sys.argv[2:] or ["track", "circle", "small-circle", "inkscape"]
    for n in names:
        t0 = time.time()
        t = load_track_from_template(template, n)
        out = bundled_track_path(n)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        t.save_npz(out)
        print(f"{n}: {t.width}x{t.height} wall_px={int(t.wall_mask().sum())} chunks={len(t.chunks)} "
              f"-> {out} ({os.path.getsize(out)} B, {time.time()-t0:.1f}s)")

if __name__ == "__main__":
    main()
