cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2; do
echo "== default";                       python3 tools/launch_fixed.py fast 4096 | grep wall
echo "== ROC_ACTIVE_WAIT_TIMEOUT=1000";  ROC_ACTIVE_WAIT_TIMEOUT=1000 python3 tools/launch_fixed.py fast 4096 | grep wall
echo "== HSA_ENABLE_INTERRUPT=0";        HSA_ENABLE_INTERRUPT=0 python3 tools/launch_fixed.py fast 4096 | grep wall
echo "== both";                          HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=1000 python3 tools/launch_fixed.py fast 4096 | grep wall
done
