#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the kernels beside the loop (tools/aux_kernels.py), each counter in its own pass.
#   bash tools/aux_counters.sh <tag>     -> gpurun_out/aux_<tag>/{aux_kernels.json, counters_aux.json}
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/aux_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/tools/aux_kernels.py 5 > $O/aux_kernels.json 2>$O/aux.err || exit 5
for grp in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${grp%%:*}; ctr=${grp#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_aux_$name -o runc --output-format csv -- python3 $R/tools/aux_kernels.py 3 > $O/pmc_aux_$name.log 2>&1 || exit 5
done
cd $R
python tools/collect_counters.py "" $O/pmc_aux_fetch $O/pmc_aux_write > $O/counters_aux.json || exit 6
rm -rf $O/pmc_*/
echo done
