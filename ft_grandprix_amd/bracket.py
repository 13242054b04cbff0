"""Roster tooling of the reference's ``ft_grandprix.bracket`` (bracket.py:5-46): every ``<name>.py`` of a drivers directory
becomes one roster entry ``{"driver": "drivers.<name>", "name": ..., "primary": ..., "secondary": ..., "icon": "white.png"}``
whose two colours are picked from a sorted palette by a small string hash of the module path, and is written next to the
driver as ``<name>.json`` (the per-car files ``template/cars/*.json`` rosters are assembled from).

The palette is an argument: the reference hard-wires its own colour table (GUI cosmetics, out of scope here); by default
the CSS colour names Pillow knows are used, sorted by name like the reference sorts its table."""
from __future__ import annotations

import json
import os
from typing import List, Optional, Sequence


class Hasher:
    """bracket.py:5-10: h("") = 1, h(c + rest) = (((seed + 101 * ord(c)) % 2003) * h(rest)) % 1009 -- evaluated right to left."""

    def __init__(self, seed: int):
        self.seed = seed

    def hash(self, string: str) -> int:
        h = 1
        for ch in reversed(string):
            h = (((self.seed + 101 * ord(ch)) % 2003) * h) % 1009
        return h


def default_palette() -> List[List[int]]:
    from PIL import ImageColor
    return [list(ImageColor.getrgb(v)) for _, v in sorted(ImageColor.colormap.items())]


def compute_driver_files(drivers_path: str, silent: bool = False, palette: Optional[Sequence] = None,
                         output_dir: Optional[str] = None, module_prefix: str = "drivers") -> List[dict]:
    """bracket.py:12-46.  Returns the entries; writes ``<output_dir>/<name>.json`` (default: into ``drivers_path``)."""
    colors = list(palette) if palette is not None else default_palette()
    out_dir = output_dir if output_dir is not None else drivers_path
    hasher = Hasher(10)
    items = []
    for file in sorted(os.listdir(drivers_path)):
        if ".py" not in file or "__" in file:
            continue
        stripped = file[:-3]
        if not silent:
            print(f"Found candidate driver '{stripped}'")
        module = f"{module_prefix}.{stripped}"
        item = dict(driver=module, name=module,
                    primary=colors[hasher.hash(module) % len(colors)],
                    secondary=colors[hasher.hash(module + ".") % len(colors)],
                    icon="white.png")
        output_path = os.path.join(out_dir, f"{stripped}.json")
        if not silent:
            print(f"- Writing driver config to '{output_path}'")
        with open(output_path, "w") as f:
            json.dump(item, f)
        items.append(item)
    if not silent:
        print("Collection of all items")
        for item in items:
            print(f"- {item}")
    return items


if __name__ == "__main__":
    compute_driver_files("drivers")
