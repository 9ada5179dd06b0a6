#!/bin/bash
# rocprofv3 --pmc passes (one per counter set, no trace domains) for one command:
#   tools/counters.sh OUTDIR "SET1;SET2;..." -- python3 /abs/path/script.py      (sets: space-separated counter names)
# then: python3 tools/sq_summary.py OUTDIR [kernel-name filter]
out=$1; sets=$2; shift; shift; shift
cd /tmp && export TMPDIR=/tmp
i=0
IFS=';' read -ra arr <<< "$sets"
for set in "${arr[@]}"; do
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/pass$i -o p -- "$@" > $out.pass$i.log 2>&1
  rc=$?
  echo "pass $i ($set) rc=$rc"
  if [ $rc -ne 0 ]; then exit $rc; fi        # a failed GPU step starts no further one
  i=$((i+1))
done
