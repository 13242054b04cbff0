cd "${GRAFT_REPO_ROOT:-.}"
q() { echo -n "$1: "; shift; env "$@" QUICK_SHORT=1 QUICK_CASES=1 python3 tools/quick_perf.py | cut -c12-; }
for i in 1 2; do
q "default            " A=1
q "no pairs           " FTGP_NO_PAIRS=1
q "pair tail 4        " FTGP_PAIR_TAIL=4
q "pair tail 8        " FTGP_PAIR_TAIL=8
q "index order        " FTGP_GROUP_ORDER_PLAIN=1
q "1 car x 8 waves    " FTGP_CARS_PER_BLOCK=1 FTGP_WAVES_PER_BLOCK=8
q "4 cars x 16 waves  " FTGP_CARS_PER_BLOCK=4
q "32 sectors         " FTGP_SECTORS_RT=32
q "nopairs+32 sectors " FTGP_NO_PAIRS=1 FTGP_SECTORS_RT=32
done
