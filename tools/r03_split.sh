set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for P in 10 20 30 40 50; do echo "split $P"; RT_SPLIT_PCT=$P RT_STREAMS=2 timeout -k 10 300 python3 -u tools/trace_driver.py 8 2>&1 | grep "^size"; done
