#!/bin/bash
# A variant library for same-box A/B (tools/ab_libs.sh): the kernel sources of a git revision -- or of the work tree, with extra -D
# flags -- built into ft_grandprix_amd/lib/variants/libftgp_<name>.so (git-ignored like every .so; travels to the GPU box with the tree).
#   tools/build_variant.sh <name> [rev|WORK] [extra hipcc flags ...]
set -e
cd "$(dirname "$0")/.."
name=$1; rev=${2:-WORK}; shift; shift || true
out=ft_grandprix_amd/lib/variants/libftgp_$name.so
mkdir -p ft_grandprix_amd/lib/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -w -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize"
if [ "$rev" = WORK ]; then
  /opt/rocm/bin/hipcc $FLAGS "$@" -o $out ft_grandprix_amd/csrc/ftgp_api.hip -ldl
else
  tmp=$(mktemp -d)
  git archive "$rev" ft_grandprix_amd/csrc include | tar -x -C "$tmp"
  /opt/rocm/bin/hipcc $FLAGS "$@" -o $out "$tmp/ft_grandprix_amd/csrc/ftgp_api.hip" -ldl
  rm -rf "$tmp"
fi
echo "built $out ($rev $*)"
