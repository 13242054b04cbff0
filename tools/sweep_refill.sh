#!/bin/bash
# diagnostic: build libftgp with several -DFTGP_REFILL values on the GPU box and time the headline config
set -e
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize -DFTGP_REFILL=${n%%:*} -DFTGP_SLOTS=${n##*:} -o gpurun_out/libftgp_refill${n%%:*}_${n##*:}.so ft_grandprix_amd/csrc/ftgp_api.hip -ldl
done
python3 - "$@" <<'PY'
import os, sys
sys.path.insert(0, ".")
from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
t = load_track("track")
for n in sys.argv[1:]:
    lib = capi.CLib("gpurun_out/libftgp_refill%s_%s.so" % tuple(n.split(":")), "ftgp_")
    out = []
    for policy, cars, steps in (("fast", 1, 300), ("nidc", 1, 300), ("fast", 4, 100)):
        with capi.Env(lib, t, n_envs=4096, cars_per_env=cars, n_rays=1080, spawn_mode=1, seed=1234) as e:
            e.rollout(policy, 50); e.last_kernel_ms(); best = 1e9
            for _ in range(3):
                e.rollout(policy, steps); best = min(best, e.last_kernel_ms())
        out.append(best * 1e3 / steps)
    print(f"REFILL:SLOTS {n:>5s}: fast {out[0]:7.2f}  nidc {out[1]:7.2f}  4-car {out[2]:7.2f} us/step", flush=True)
PY
