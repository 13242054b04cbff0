#!/usr/bin/env python3
"""Evidence hygiene: every log or counter file that goes under profiles/roundN says which kernel sources it was measured on.

    python3 tools/evidence.py sha                      print the hash of the kernel sources (the one bench.py compares with)
    python3 tools/evidence.py stamp FILE...            put `# kernel_source_sha=<sha> <date>` in front of each log
    python3 tools/evidence.py publish DST FILE...      copy into DST (profiles/roundN) -- REFUSES a file whose stamp (or, for a
                                                       .json, whose "kernel_source_sha") is not the current sources' hash
    python3 tools/evidence.py meta [lib.so]            resource usage of the step kernels from the code object's metadata
"""
import hashlib, json, os, re, shutil, subprocess, sys, tempfile, time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
SOURCES = ("ft_grandprix_amd/csrc/ftgp_kernels.hip", "ft_grandprix_amd/csrc/ftgp_march.h", "ft_grandprix_amd/csrc/ftgp_device.h",
           "ft_grandprix_amd/csrc/ftgp_api.hip", "ft_grandprix_amd/csrc/diag/ftgp_diag.inc", "include/ftgp.h")
LLVM = "/opt/rocm/lib/llvm/bin"


def sha():
    h = hashlib.sha256()
    for rel in SOURCES:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


def stamp(path):
    text = open(path, errors="replace").read()
    if text.startswith("# kernel_source_sha="):
        return                                         # stamped when it was measured (e.g. by tools/ab_pmc.sh): never re-stamp after the fact
    open(path, "w").write(f"# kernel_source_sha={sha()} {time.strftime('%Y-%m-%d %H:%M:%S')} {os.path.basename(path)}\n" + text)


def stamp_of(path):
    if path.endswith(".json"):
        try:
            first = open(path).read().lstrip()
            d = json.loads(first if not first.startswith("{\"metric") else first.splitlines()[0])
        except ValueError:
            return None
        if "kernel_source_sha" in d:
            return d["kernel_source_sha"]
        side = path + ".sha"                            # a bench line carries the hashes of its counter files, not its own: sidecar
        return open(side).read().split()[0] if os.path.exists(side) else None
    head = open(path, errors="replace").readline()
    m = re.match(r"# kernel_source_sha=([0-9a-f]{16})", head)
    if m:
        return m.group(1)
    side = path + ".sha"                                # files that cannot carry a comment line (csv): a sidecar written at collection time
    return open(side).read().split()[0] if os.path.exists(side) else None


def publish(dst, files):
    os.makedirs(dst, exist_ok=True)
    cur, bad = sha(), []
    for f in files:
        got = stamp_of(f)
        if got != cur:
            bad.append((f, got))
            continue
        shutil.copy(f, os.path.join(dst, os.path.basename(f)))
        if os.path.exists(f + ".sha"):
            shutil.copy(f + ".sha", os.path.join(dst, os.path.basename(f) + ".sha"))
    for f, got in bad:
        print(f"REFUSED {f}: measured on sources {got}, the tree is at {cur}", file=sys.stderr)
    return 1 if bad else 0


def code_object_meta(lib=None):
    """.vgpr_count / .sgpr_count / spills / scratch of every ftgp_step_kernel instantiation, from the gfx950 code object inside the
    library (rocprofv3's VGPR_Count / LDS_Block_Size columns read 32 / 0 for a 63-register kernel with 73 KB of dynamic LDS)."""
    lib = lib or os.path.join(ROOT, "ft_grandprix_amd", "lib", "libftgp.so")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(td, "copy.so")])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    for block in notes.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name or "ftgp_step_kernel" not in name.group(1):
            continue
        flags = re.search(r"ftgp_step_kernelILb(\d)ELb(\d)ELb(\d)E", name.group(1))
        key = "ftgp_step_kernel<%s, %s, %s>" % tuple("true" if f == "1" else "false" for f in flags.groups()) if flags else name.group(1)
        get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
        out[key] = {"vgpr_count": get("vgpr_count"), "sgpr_count": get("sgpr_count"), "sgpr_spill_count": get("sgpr_spill_count"),
                    "vgpr_spill_count": get("vgpr_spill_count"), "scratch_bytes_per_lane": get("private_segment_fixed_size"),
                    "static_lds_bytes": get("group_segment_fixed_size")}
    return out


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "sha"
    if cmd == "sha":
        print(sha())
    elif cmd == "stamp":
        for f in sys.argv[2:]:
            stamp(f)
    elif cmd == "publish":
        raise SystemExit(publish(sys.argv[2], sys.argv[3:]))
    elif cmd == "meta":
        print(json.dumps(code_object_meta(sys.argv[2] if len(sys.argv) > 2 else None), indent=1))
    else:
        raise SystemExit(__doc__)
