cd "${GRAFT_REPO_ROOT:-.}"
q() { echo -n "$1: "; shift; env "$@" QUICK_CASES=1 python3 tools/quick_perf.py | cut -c12-; }
for i in 1 2; do
q "512 envs: 1 car    " QUICK_ENVS=512
q "512 envs: 2 cars   " QUICK_ENVS=512 FTGP_CARS_PER_BLOCK=2
q "512 envs: 4 cars   " QUICK_ENVS=512 FTGP_CARS_PER_BLOCK=4
q "2048 envs: 4 cars  " QUICK_ENVS=2048
q "2048 envs: 8 cars  " QUICK_ENVS=2048 FTGP_CARS_PER_BLOCK=8
q "2048 envs: 6 cars  " QUICK_ENVS=2048 FTGP_CARS_PER_BLOCK=6
q "256 envs: 1 car    " QUICK_ENVS=256
q "256 envs: 2 cars   " QUICK_ENVS=256 FTGP_CARS_PER_BLOCK=2
q "1536 envs: 3 cars  " QUICK_ENVS=1536
q "1536 envs: 6 cars  " QUICK_ENVS=1536 FTGP_CARS_PER_BLOCK=6
q "1536 envs: 4 cars  " QUICK_ENVS=1536 FTGP_CARS_PER_BLOCK=4
done
