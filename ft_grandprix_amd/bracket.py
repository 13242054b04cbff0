"""Roster tooling: what ``python -m ft_grandprix.bracket`` does for a drivers directory (bracket.py:5-46).

Every file of the directory whose name contains ".py" and no "__" becomes one roster entry

    {"driver": "drivers.<stem>", "name": "drivers.<stem>", "primary": rgb, "secondary": rgb, "icon": "white.png"}

with ``<stem>`` = the file name minus its last three characters (so ``x.pyc`` yields the stem ``x.``, as in the reference).  The
two colours come out of a palette indexed by a small multiplicative string hash of the module path (``primary``) and of the
module path plus "." (``secondary``).  Each entry is written as ``<stem>.json`` next to the drivers: the per-car files that
rosters like ``template/cars/cars.json`` are assembled from.  Fixture G7 (tests/golden/g7_bracket.json) holds the reference's
own output for a generated directory.

The default palette is the table the reference indexes (140 RGB triples in colour-name order), shipped as data in
``assets/bracket_palette.json``; ``palette=`` overrides it.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Callable, Iterator, List, Optional, Sequence, Tuple

ICON = "white.png"
HASH_SEED = 10                      # bracket.py:15


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
    with open(Path(__file__).with_name("assets") / "bracket_palette.json") as f:
        return json.load(f)["rgb"]


def is_candidate(file_name: str) -> bool:
    """The reference's filter (bracket.py:18-19): substring tests, not a suffix test."""
    return ".py" in file_name and "__" not in file_name


def roster_entries(file_names: Sequence[str], palette: Sequence, module_prefix: str = "drivers") -> Iterator[Tuple[str, dict]]:
    """(stem, entry) for every candidate, in sorted file order."""
    pick = Hasher(HASH_SEED).hash
    n = len(palette)
    for file_name in sorted(file_names):
        if not is_candidate(file_name):
            continue
        stem = file_name[:-3]
        module = f"{module_prefix}.{stem}"
        yield stem, {"driver": module, "name": module, "primary": palette[pick(module) % n],
                     "secondary": palette[pick(module + ".") % n], "icon": ICON}


def compute_driver_files(drivers_path: str, silent: bool = False, palette: Optional[Sequence] = None,
                         output_dir: Optional[str] = None, module_prefix: str = "drivers",
                         report: Callable[[str], None] = print) -> List[dict]:
    """Writes ``<output_dir or drivers_path>/<stem>.json`` for every candidate and returns the entries."""
    src = Path(drivers_path)
    dst = Path(output_dir) if output_dir is not None else src
    say = (lambda _msg: None) if silent else report
    table = list(palette) if palette is not None else default_palette()
    entries = []
    for stem, entry in roster_entries([p.name for p in src.iterdir()], table, module_prefix):
        target = dst / f"{stem}.json"
        say(f"driver candidate '{stem}' -> {target}")
        target.write_text(json.dumps(entry))
        entries.append(entry)
    say(f"{len(entries)} roster entries:")
    for entry in entries:
        say(f"  {entry}")
    return entries


if __name__ == "__main__":
    compute_driver_files("drivers")
